// Lock-in sweep kernels for gfx950.
//
// A lock-in is  sf = ifft2( fft2(img * cx (x) cy) * Gx (x) Gy ).  Carrier and
// Gaussian are both separable, so it is computed as two 1-D circular filters
//     pass A (x axis, column tiles):  T  = Cx( img * cx )                   (complex, to HBM)
//     pass B (y axis, rows)        :  sf = Cy( cy * T )  -> per-pixel best-of-K  (stays in registers)
// each filter being  forward FFT -> table multiply -> inverse FFT  entirely in
// registers + LDS.  HBM sees the real image once, T once each way, and the winning
// lock-in once: 4 reals per lock-in per pixel instead of the 8 a 2-D FFT pair costs.
//
// The k-vector loop runs INSIDE both kernels: pass A keeps its image tile in
// registers for all B lock-ins, pass B keeps the running best of a peak in
// registers across its K candidates (strict '>' in list order == the reference's
// first-maximum-wins rule, geometric_phase_analysis.py:679-684).
#include "gpa_internal.h"

namespace gpa {

thread_local KernelProfiler* g_kprof = nullptr;

// ---------------------------------------------------------------------------
// carrier / compensation tables, computed in double on the device
// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ cpx<T> unit_phasor(double cycles) {
  double fr = cycles - rint(cycles);
  double s, c;
  sincospi(2.0 * fr, &s, &c);
  return {(T)c, (T)s};
}

template <class T>
__global__ void tables_kernel(const double* __restrict__ kl, const double* __restrict__ kr,
                              const double* __restrict__ pw, int B, int Bx, int n0, int n1, int L0, int L1,
                              cpx<T>* cxb, cpx<T>* sx, cpx<T>* wxw, cpx<T>* cyb, cpx<T>* sy, cpx<T>* wyw,
                              cpx<T>* dx, cpx<T>* dy) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < Bx) {   // x carrier of x-plane b
    const double wx = pw[b];
    const int tpf = L0 / 16;
    if (j < tpf) cxb[(size_t)b * tpf + j] = unit_phasor<T>(wx * j);
    if (j < 16) sx[b * 16 + j] = unit_phasor<T>(wx * (double)tpf * j);
    if (j == 0) wxw[b] = unit_phasor<T>(-wx * (double)(L0 - n0));
  }
  if (b < B) {    // y carrier and compensation phasors of candidate b
    const double wx = kl[2 * b], wy = kl[2 * b + 1], kx = kr[2 * b], ky = kr[2 * b + 1];
    const int tpf = L1 / 16;
    if (j < tpf) cyb[(size_t)b * tpf + j] = unit_phasor<T>(wy * j);
    if (j < 16) sy[b * 16 + j] = unit_phasor<T>(wy * (double)tpf * j);
    if (j == 0) wyw[b] = unit_phasor<T>(-wy * (double)(L1 - n1));
    if (j < n1) dy[(size_t)b * n1 + j] = unit_phasor<T>(-(wy - ky) * j);
    if (j < n0) dx[(size_t)b * n0 + j] = unit_phasor<T>(-(wx - kx) * j);
  }
}

hipError_t launch_tables(int dtype, const Axis& a0, const Axis& a1, const double* kl,
                         const double* kr, int B, const double* pw, int Bx, const SweepTables& tb,
                         hipStream_t s) {
  int len = a0.n > a1.n ? a0.n : a1.n;
  if (len < a0.L / 16) len = a0.L / 16;
  if (len < a1.L / 16) len = a1.L / 16;
  if (len < 16) len = 16;
  dim3 grid((len + 255) / 256, B > Bx ? B : Bx);
  if (dtype == 0)
    tables_kernel<float><<<grid, 256, 0, s>>>(kl, kr, pw, B, Bx, a0.n, a1.n, a0.L, a1.L, (cpx<float>*)tb.cxb,
                                              (cpx<float>*)tb.sx, (cpx<float>*)tb.wxw, (cpx<float>*)tb.cyb,
                                              (cpx<float>*)tb.sy, (cpx<float>*)tb.wyw, (cpx<float>*)tb.dx,
                                              (cpx<float>*)tb.dy);
  else
    tables_kernel<double><<<grid, 256, 0, s>>>(kl, kr, pw, B, Bx, a0.n, a1.n, a0.L, a1.L, (cpx<double>*)tb.cxb,
                                               (cpx<double>*)tb.sx, (cpx<double>*)tb.wxw, (cpx<double>*)tb.cyb,
                                               (cpx<double>*)tb.sy, (cpx<double>*)tb.wyw, (cpx<double>*)tb.dx,
                                               (cpx<double>*)tb.dy);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// image mean (deterministic two-stage reduction in double)
// ---------------------------------------------------------------------------
template <class T>
__global__ void mean_partial_kernel(const T* __restrict__ img, size_t count, double* partial) {
  __shared__ double sh[256];
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    acc += (double)img[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
template <class T>
__global__ void mean_final_kernel(const double* partial, int nparts, size_t count, T* mean_out) {
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *mean_out = (T)(sh[0] / (double)count);
}

hipError_t launch_mean(int dtype, const void* image, size_t count, double* scratch,
                       void* mean_out, hipStream_t s) {
  const int nparts = 1024;
  GPA_PROF("mean_kernels", s);
  if (dtype == 0) {
    mean_partial_kernel<float><<<nparts, 256, 0, s>>>((const float*)image, count, scratch);
    mean_final_kernel<float><<<1, 256, 0, s>>>(scratch, nparts, count, (float*)mean_out);
  } else {
    mean_partial_kernel<double><<<nparts, 256, 0, s>>>((const double*)image, count, scratch);
    mean_final_kernel<double><<<1, 256, 0, s>>>(scratch, nparts, count, (double*)mean_out);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// geometry helpers
// ---------------------------------------------------------------------------
template <class T, int LG>
struct PassAGeom {
  using F = WgFFT<T, LG>;
  // columns per workgroup: as many as fit 1024 threads and the 160 KiB LDS
  static constexpr int cols() {
    int c = 16;
    while (c > 1 && (c * F::TPF > 1024 || (size_t)c * (F::LDS_ELEMS + 32) * sizeof(cpx<T>) > 160 * 1024)) c /= 2;
    return c;
  }
  static constexpr int C = cols();
  // transforms per thread: two f32 columns per thread halve the barriers per transform
  // and leave 256 VGPRs per lane (8 waves per CU) instead of forcing 128 on 16 waves
  static constexpr int NT = (sizeof(T) == 4 && C >= 2) ? 2 : 1;
  static constexpr int CT = C / NT;   // columns side by side in the thread index
  // LDS: one region per transform-of-a-thread (n), inside which the CT columns that sit
  // side by side in the thread index are interleaved element by element
  static constexpr int REGION = CT * F::LDS_ELEMS;
  static constexpr int THREADS = CT * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NT * REGION * sizeof(cpx<T>);
  static constexpr bool FITS = (size_t)(F::LDS_ELEMS + 32) * sizeof(cpx<T>) <= 160 * 1024;
};

template <class T, int LG>
struct PassBGeom {
  using F = WgFFT<T, LG>;
  static constexpr int NF = F::TPF >= 256 ? 1 : 256 / F::TPF;   // rows per workgroup
  static constexpr int RS = F::LDS_ELEMS + (NF > 1 ? (F::TPF < 32 ? F::TPF : 0) : 0);
  static constexpr int THREADS = NF * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NF * RS * sizeof(cpx<T>);
};

template <bool PADDED, class T>
struct HType { using type = T; };
template <class T>
struct HType<true, T> { using type = cpx<T>; };

template <class T> __device__ __forceinline__ cpx<T> hmul(cpx<T> a, T h) { return {a.x * h, a.y * h}; }
template <class T> __device__ __forceinline__ cpx<T> hmul(cpx<T> a, cpx<T> h) { return cmul(a, h); }

// ---------------------------------------------------------------------------
// pass A: x-axis filter on column tiles
// ---------------------------------------------------------------------------
template <class T, int LG, bool PADDED>
__global__ __launch_bounds__((PassAGeom<T, LG>::THREADS)) void passA_kernel(
    const T* __restrict__ image, const T* __restrict__ mean, int n0, int n1,
    const cpx<T>* __restrict__ cxb, const cpx<T>* __restrict__ sx, const cpx<T>* __restrict__ wxw,
    const typename HType<PADDED, T>::type* __restrict__ H,
    const cpx<T>* __restrict__ twtab, cpx<T>* __restrict__ Tout, int B, int bchunk) {
  using F = WgFFT<T, LG>;
  using G = PassAGeom<T, LG>;
  constexpr int CT = G::CT, NT = G::NT, TPF = F::TPF, L = F::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // thread -> (column slot c, transform thread t); transform n of the thread handles
  // column  tile*C + c*NT + n  (a thread's columns are adjacent: one 16-byte store per row)
  const int c = threadIdx.x % CT, t = threadIdx.x / CT;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + c;
  constexpr int LDS_NSTRIDE = G::REGION;
  // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (round-robin
  // dispatch), so give each XCD a contiguous run of column tiles -- the tiles that
  // share a 128-byte line of the output then meet in one L2 and leave it as whole lines.
  int tile = blockIdx.x;
  if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int y0 = tile * G::C + c * NT;
  const T m = mean ? *mean : T(0);

  T val[NT][16];
  unsigned wrapmask = 0;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int y = y0 + n;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int slot = t + TPF * i;
      const int xs = axis_src(slot, n0, L, PADDED);
      val[n][i] = (y < n1 && xs >= 0) ? image[(size_t)xs * n1 + y] - m : T(0);
      if (PADDED && slot >= n0) wrapmask |= 1u << i;
    }
  }
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, t);

  const int b0 = blockIdx.y * bchunk;
  const int b1 = (b0 + bchunk < B) ? b0 + bchunk : B;
  for (int b = b0; b < b1; ++b) {
    const cpx<T> base = cxb[(size_t)b * TPF + t];
    cpx<T> x[NT][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      cpx<T> ph = cmul(base, sx[b * 16 + i]);
      if (PADDED && ((wrapmask >> i) & 1)) ph = cmul(ph, wxw[b]);
#pragma unroll
      for (int n = 0; n < NT; ++n) x[n][i] = {val[n][i] * ph.x, val[n][i] * ph.y};
    }
    F::template forward_multi<NT, CT>(x, lds, LDS_NSTRIDE, t, tw);
    {
      // keep the filter table out of the registers across the lock-in loop: it is
      // re-read from L1/L2 every iteration (the asm makes the pointer opaque to LICM)
      const typename HType<PADDED, T>::type* Hb = H;
      asm volatile("" : "+s"(Hb));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const auto h = Hb[i * TPF + t];
#pragma unroll
        for (int n = 0; n < NT; ++n) x[n][i] = hmul(x[n][i], h);
      }
    }
    F::template inverse_multi<NT, CT>(x, lds, LDS_NSTRIDE, t, tw);
    bool paired = false;
    if constexpr (NT == 2) {
      // two adjacent columns of one row go out as a single 2-complex (16 B for f32) store
      // when the row pitch keeps that store naturally aligned
      if (y0 + 1 < n1 && (n1 & 1) == 0) {
        paired = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int slot = t + TPF * i;
          if (!PADDED || slot < n0) {
            struct alignas(2 * sizeof(cpx<T>)) Pair { cpx<T> a, b; };
            Pair pr = {x[0][i], x[1][i]};
            *reinterpret_cast<Pair*>(&Tout[((size_t)b * n0 + slot) * n1 + y0]) = pr;
          }
        }
      }
    }
    if (!paired) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int y = y0 + n;
        if (y < n1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int slot = t + TPF * i;
            if (!PADDED || slot < n0) Tout[((size_t)b * n0 + slot) * n1 + y] = x[n][i];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pass B: y-axis filter on rows, best-of-K select
// ---------------------------------------------------------------------------
template <class T, int LG, bool PADDED, bool SELECT>
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64: cap at 256 VGPRs (2 waves/SIMD) instead of 299 at 1 wave: pass B 7.5 -> 5.9 ms
#endif
__global__ __launch_bounds__((PassBGeom<T, LG>::THREADS), (sizeof(T) == 8 ? GPA_F64_WAVES : 1)) void passB_kernel(
    const cpx<T>* __restrict__ Tin, int n0, int n1,
    const typename HType<PADDED, T>::type* __restrict__ H, const cpx<T>* __restrict__ twtab,
    const int* __restrict__ planeof, const cpx<T>* __restrict__ cyb, const cpx<T>* __restrict__ sy,
    const cpx<T>* __restrict__ wyw, const cpx<T>* __restrict__ dx, const cpx<T>* __restrict__ dy, int K,
    cpx<T>* __restrict__ out, int32_t* __restrict__ kidx) {
  using F = WgFFT<T, LG>;
  using G = PassBGeom<T, LG>;
  constexpr int TPF = F::TPF, L = F::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int row = blockIdx.x * G::NF + f;
  const bool valid = row < n0;
  const int p = blockIdx.y;

  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);

  cpx<T> best[16];
  int bidx[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { best[i] = {T(0), T(0)}; bidx[i] = -1; }

  const int nk = SELECT ? K : 1;
  for (int k = 0; k < nk; ++k) {
    const int b = SELECT ? p * K + k : p;
    // the x-plane of this candidate (shared by every candidate with the same wx: re-reads hit L2)
    const cpx<T>* src = Tin + ((size_t)planeof[b] * n0 + (valid ? row : 0)) * n1;
    const cpx<T> cbase = cyb[(size_t)b * TPF + tid];
    cpx<T> x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      cpx<T> ph = cmul(cbase, sy[b * 16 + i]);   // exp(2 pi i wy y) at y = tid + TPF*i
      if constexpr (PADDED) {
        const int slot = tid + TPF * i;
        const int ys = axis_src(slot, n1, L, true);
        if (slot >= n1) ph = cmul(ph, wyw[b]);
        x[i] = ys >= 0 ? cmul(src[ys], ph) : cpx<T>{T(0), T(0)};
      } else {
        x[i] = cmul(src[tid + TPF * i], ph);   // rows past the image reuse row 0; their results are dropped
      }
    }
    F::forward(x, lds, tid, tw);
    {
      const typename HType<PADDED, T>::type* Hb = H;
      asm volatile("" : "+s"(Hb));   // re-read the filter table per candidate instead of pinning 16+ VGPRs
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = hmul(x[i], Hb[i * TPF + tid]);
    }
    F::inverse(x, lds, tid, tw);
    if constexpr (SELECT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // |sf|^2 of the kept candidate is recomputed rather than kept in a register
        const T a = x[i].x * x[i].x + x[i].y * x[i].y;
        const T ab = best[i].x * best[i].x + best[i].y * best[i].y;
        if (a > ab) { best[i] = x[i]; bidx[i] = k; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) best[i] = x[i];
    }
  }
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int yy = tid + TPF * i;
    if (!PADDED || yy < n1) {
      const size_t o = ((size_t)p * n0 + row) * n1 + yy;
      if constexpr (SELECT) {
        cpx<T> v = {T(0), T(0)};
        if (bidx[i] >= 0) {
          const size_t bb = (size_t)p * K + bidx[i];
          v = cmul(best[i], cmul(dx[bb * n0 + row], dy[bb * n1 + yy]));
        }
        out[o] = v;
        if (kidx) kidx[o] = bidx[i];
      } else {
        out[o] = best[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// a4: select + phase-gradient of the winning candidate (wfr2_grad_opt,
// geometric_phase_analysis.py:803-812).  Input: all K lock-ins of one peak, sf[k][x][y].
// grad = np.gradient(-angle(sf_winner)) (central differences, one-sided at the borders)
//        + 2 pi (w - kref), finally wrapToPi(2 g) / 2.
// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T neg_angle(cpx<T> v) { return -atan2(v.y, v.x); }

template <class T>
__global__ __launch_bounds__(256) void gradselect_kernel(const cpx<T>* __restrict__ sf, int K, int n0, int n1,
                                                        const double* __restrict__ kl, const double* __restrict__ kr,
                                                        const cpx<T>* __restrict__ dx, const cpx<T>* __restrict__ dy,
                                                        cpx<T>* __restrict__ lockin, int32_t* __restrict__ kidx,
                                                        T* __restrict__ grad) {
  const int y = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y;
  if (y >= n1) return;
  const size_t npx = (size_t)n0 * n1, o = (size_t)x * n1 + y;
  T ba = T(0);
  int bi = -1;
  for (int k = 0; k < K; ++k) {
    const cpx<T> v = sf[k * npx + o];
    const T a = v.x * v.x + v.y * v.y;
    if (a > ba) { ba = a; bi = k; }
  }
  cpx<T> out = {T(0), T(0)};
  T g0 = T(0), g1 = T(0);
  if (bi >= 0) {
    const cpx<T>* pl = sf + bi * npx;
    out = cmul(pl[o], cmul(dx[(size_t)bi * n0 + x], dy[(size_t)bi * n1 + y]));
    const T c = neg_angle(pl[o]);
    const T xm = x > 0 ? neg_angle(pl[o - n1]) : c, xp = x + 1 < n0 ? neg_angle(pl[o + n1]) : c;
    const T ym = y > 0 ? neg_angle(pl[o - 1]) : c, yp = y + 1 < n1 ? neg_angle(pl[o + 1]) : c;
    g0 = (x > 0 && x + 1 < n0) ? T(0.5) * (xp - xm) : (xp - xm);
    g1 = (y > 0 && y + 1 < n1) ? T(0.5) * (yp - ym) : (yp - ym);
    const T two_pi = T(6.28318530717958647692), pi = T(3.14159265358979323846);
    g0 += (T)(6.28318530717958647692 * (kl[2 * bi] - kr[2 * bi]));
    g1 += (T)(6.28318530717958647692 * (kl[2 * bi + 1] - kr[2 * bi + 1]));
    // wrapToPi(2 g) / 2 with the floored modulo of mathtools.py:72-75
    T t0 = T(2) * g0 + pi, t1 = T(2) * g1 + pi;
    g0 = T(0.5) * (t0 - two_pi * floor(t0 / two_pi) - pi);
    g1 = T(0.5) * (t1 - two_pi * floor(t1 / two_pi) - pi);
  }
  lockin[o] = out;
  if (kidx) kidx[o] = bi;
  grad[2 * o] = g0;
  grad[2 * o + 1] = g1;
}

hipError_t launch_gradselect(int dtype, const void* sf, int K, int n0, int n1, const double* kl, const double* kr,
                             const SweepTables& tb, void* lockin, int32_t* kidx, void* grad, hipStream_t s) {
  dim3 grid((n1 + 255) / 256, n0);
  if (dtype == 0)
    gradselect_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)sf, K, n0, n1, kl, kr, (const cpx<float>*)tb.dx,
                                                  (const cpx<float>*)tb.dy, (cpx<float>*)lockin, kidx, (float*)grad);
  else
    gradselect_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)sf, K, n0, n1, kl, kr, (const cpx<double>*)tb.dx,
                                                   (const cpx<double>*)tb.dy, (cpx<double>*)lockin, kidx, (double*)grad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <class T, int LG, bool PADDED>
static hipError_t run_passA(const Axis& a0, int n1, const void* image, const void* mean,
                            const SweepTables& tb, const void* Hx, const void* tw0, void* Tbuf,
                            int B, hipStream_t s) {
  using G = PassAGeom<T, LG>;
  if constexpr (!G::FITS) {
    return hipErrorInvalidValue;
  } else {
    auto kern = passA_kernel<T, LG, PADDED>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int tiles = (n1 + G::C - 1) / G::C;
    // split the lock-in loop over grid.y when there are too few column tiles to fill 256 CUs
    int ysplit = 1;
    while (tiles * ysplit < 512 && ysplit < B) ysplit *= 2;
    if (ysplit > B) ysplit = B;
    const int bchunk = (B + ysplit - 1) / ysplit;
    dim3 grid(tiles, (B + bchunk - 1) / bchunk);
    GPA_PROF("passA_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>(
        (const T*)image, (const T*)mean, a0.n, n1, (const cpx<T>*)tb.cxb, (const cpx<T>*)tb.sx,
        (const cpx<T>*)tb.wxw,
        (const typename HType<PADDED, T>::type*)Hx, (const cpx<T>*)tw0, (cpx<T>*)Tbuf, B, bchunk);
    return hipGetLastError();
  }
}

template <class T, int LG, bool PADDED, bool SELECT>
static hipError_t run_passB(const Axis& a1, int n0, const void* Tbuf, const void* Hy,
                            const void* tw1, const SweepTables& tb, int P, int K, void* out,
                            int32_t* kidx, hipStream_t s) {
  using G = PassBGeom<T, LG>;
  if constexpr (G::LDS_BYTES > 160 * 1024) {
    return hipErrorInvalidValue;
  } else {
    auto kern = passB_kernel<T, LG, PADDED, SELECT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
    dim3 grid((n0 + G::NF - 1) / G::NF, P);
    GPA_PROF("passB_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>(
        (const cpx<T>*)Tbuf, n0, a1.n, (const typename HType<PADDED, T>::type*)Hy,
        (const cpx<T>*)tw1, tb.planeof, (const cpx<T>*)tb.cyb, (const cpx<T>*)tb.sy, (const cpx<T>*)tb.wyw,
        (const cpx<T>*)tb.dx, (const cpx<T>*)tb.dy, K, (cpx<T>*)out, kidx);
    return hipGetLastError();
  }
}

#define GPA_FOR_LG(X) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)

hipError_t launch_passA(int dtype, const Axis& a0, int n1, const void* image, const void* mean,
                        const SweepTables& tb, const void* Hx, const void* tw0, void* Tbuf,
                        int B, hipStream_t s) {
#define CASE_A(LG)                                                                                \
  case LG:                                                                                        \
    if (dtype == 0)                                                                               \
      return a0.padded ? run_passA<float, LG, true>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s) \
                       : run_passA<float, LG, false>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s); \
    else                                                                                          \
      return a0.padded ? run_passA<double, LG, true>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s) \
                       : run_passA<double, LG, false>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s);
  switch (a0.lg) { GPA_FOR_LG(CASE_A) }
#undef CASE_A
  return hipErrorInvalidValue;
}

hipError_t launch_passB(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy,
                        const void* tw1, const SweepTables& tb, int P, int K, bool select,
                        void* out, int32_t* kidx, hipStream_t s) {
#define CALL_B(T, LG, PD)                                                                   \
  (select ? run_passB<T, LG, PD, true>(a1, n0, Tbuf, Hy, tw1, tb, P, K, out, kidx, s)       \
          : run_passB<T, LG, PD, false>(a1, n0, Tbuf, Hy, tw1, tb, P, K, out, kidx, s))
#define CASE_B(LG)                                                           \
  case LG:                                                                   \
    if (dtype == 0) return a1.padded ? CALL_B(float, LG, true) : CALL_B(float, LG, false); \
    else return a1.padded ? CALL_B(double, LG, true) : CALL_B(double, LG, false);
  switch (a1.lg) { GPA_FOR_LG(CASE_B) }
#undef CASE_B
#undef CALL_B
  return hipErrorInvalidValue;
}

int passA_cols(int dtype, int lg) {
#define CASE_C(LG) \
  case LG: return dtype == 0 ? PassAGeom<float, LG>::C : PassAGeom<double, LG>::C;
  switch (lg) { GPA_FOR_LG(CASE_C) }
#undef CASE_C
  return 0;
}

int spec_index_rt(int lg, int tid, int reg) {
#define CASE_S(LG) \
  case LG: return WgFFT<float, LG>::spec_index(tid, reg);
  switch (lg) { GPA_FOR_LG(CASE_S) }
#undef CASE_S
  return -1;
}

}  // namespace gpa
