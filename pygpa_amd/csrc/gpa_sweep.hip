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
#include "gpa_passb.h"

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
                              const double* __restrict__ pw, int B, int Bx, int n0, int n1, int L0, int L1, int tpf0, int tpf1,
                              cpx<T>* cxb, cpx<T>* sx, cpx<T>* wxw, cpx<T>* wxr, cpx<T>* cyb, cpx<T>* sy, cpx<T>* wyw,
                              cpx<T>* wyr,
                              cpx<T>* dx, cpx<T>* dy) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < Bx) {   // x carrier of x-plane b
    const double wx = pw[b];
    const int tpf = tpf0;
    if (j < tpf) cxb[(size_t)b * tpf + j] = unit_phasor<T>(wx * j);
    if (j < 16) sx[b * 16 + j] = unit_phasor<T>(wx * (double)tpf * j);
    if (j == 0) wxw[b] = unit_phasor<T>(-wx * (double)(L0 - n0));
    if (j == 1) wxr[b] = unit_phasor<T>(-wx * (double)n0);
  }
  if (b < B) {    // y carrier and compensation phasors of candidate b
    const double wx = kl[2 * b], wy = kl[2 * b + 1], kx = kr[2 * b], ky = kr[2 * b + 1];
    const int tpf = tpf1;
    if (j < tpf) cyb[(size_t)b * tpf + j] = unit_phasor<T>(wy * j);
    if (j < 16) sy[b * 16 + j] = unit_phasor<T>(wy * (double)tpf * j);
    if (j == 0) wyw[b] = unit_phasor<T>(-wy * (double)(L1 - n1));
    if (j == 1) wyr[b] = unit_phasor<T>(-wy * (double)n1);
    if (j < n1) dy[(size_t)b * n1 + j] = unit_phasor<T>(-(wy - ky) * j);
    if (j < n0) dx[(size_t)b * n0 + j] = unit_phasor<T>(-(wx - kx) * j);
  }
}

hipError_t launch_tables(int dtype, const Axis& a0, const Axis& a1, const double* kl,
                         const double* kr, int B, const double* pw, int Bx, const SweepTables& tb,
                         hipStream_t s) {
  const int tpf0 = axis_tpf(a0), tpf1 = axis_tpf(a1);
  int len = a0.n > a1.n ? a0.n : a1.n;
  if (len < tpf0) len = tpf0;
  if (len < tpf1) len = tpf1;
  if (len < 16) len = 16;
  dim3 grid((len + 255) / 256, B > Bx ? B : Bx);
  if (dtype == 0)
    tables_kernel<float><<<grid, 256, 0, s>>>(kl, kr, pw, B, Bx, a0.n, a1.n, a0.L, a1.L, tpf0, tpf1, (cpx<float>*)tb.cxb,
                                              (cpx<float>*)tb.sx, (cpx<float>*)tb.wxw, (cpx<float>*)tb.wxr, (cpx<float>*)tb.cyb,
                                              (cpx<float>*)tb.sy, (cpx<float>*)tb.wyw, (cpx<float>*)tb.wyr, (cpx<float>*)tb.dx,
                                              (cpx<float>*)tb.dy);
  else
    tables_kernel<double><<<grid, 256, 0, s>>>(kl, kr, pw, B, Bx, a0.n, a1.n, a0.L, a1.L, tpf0, tpf1, (cpx<double>*)tb.cxb,
                                               (cpx<double>*)tb.sx, (cpx<double>*)tb.wxw, (cpx<double>*)tb.wxr, (cpx<double>*)tb.cyb,
                                               (cpx<double>*)tb.sy, (cpx<double>*)tb.wyw, (cpx<double>*)tb.wyr, (cpx<double>*)tb.dx,
                                               (cpx<double>*)tb.dy);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// image mean (deterministic two-stage reduction in double)
// ---------------------------------------------------------------------------
template <class T> struct alignas(16) MeanVec4 { T v[4]; };
template <class T>
__global__ void mean_partial_kernel(const T* __restrict__ img, size_t count, double* partial) {
  img += (size_t)blockIdx.y * count;        // (image stacks: blockIdx.y = image)
  partial += (size_t)blockIdx.y * gridDim.x;
  __shared__ double sh[256];
  double acc = 0;
  if ((count & 3) == 0 && (reinterpret_cast<size_t>(img) & 15) == 0) {
    // 16-byte loads, four independent chains per thread (the scalar loop was one dependent chain of 4-byte loads:
    // 33 us for a 4096^2 image, a third of HBM rate); fixed order, so the sum is reproducible
    const MeanVec4<T>* v = reinterpret_cast<const MeanVec4<T>*>(img);
    const size_t c4 = count >> 2, step = (size_t)gridDim.x * 256;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + step < c4; i += 2 * step) {
      const MeanVec4<T> u = v[i], w = v[i + step];
      a0 += (double)u.v[0] + (double)w.v[0];
      a1 += (double)u.v[1] + (double)w.v[1];
      a2 += (double)u.v[2] + (double)w.v[2];
      a3 += (double)u.v[3] + (double)w.v[3];
    }
    if (i < c4) {
      const MeanVec4<T> u = v[i];
      a0 += (double)u.v[0]; a1 += (double)u.v[1]; a2 += (double)u.v[2]; a3 += (double)u.v[3];
    }
    acc = (a0 + a1) + (a2 + a3);
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
      acc += (double)img[i];
  }
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
  partial += (size_t)blockIdx.y * nparts;
  mean_out += blockIdx.y;
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

// nimg > 1: the means of nimg images of `count` pixels, one after the other in memory (scratch: 1024 doubles per
// image, mean_out: nimg values); every image's sum is formed exactly as by a single-image call
hipError_t launch_mean(int dtype, const void* image, size_t count, double* scratch,
                       void* mean_out, hipStream_t s, int nimg) {
  // one workgroup per 4096 pixels, 32 ... 1024 of them (a 512^2 image was reduced by 1024 workgroups, most of them idle)
  const size_t want = count / 4096;
  const int nparts = want > 1024 ? 1024 : (want < 32 ? 32 : (int)want);
  GPA_PROF("mean_kernels", s);
  if (dtype == 0) {
    mean_partial_kernel<float><<<dim3(nparts, nimg), 256, 0, s>>>((const float*)image, count, scratch);
    mean_final_kernel<float><<<dim3(1, nimg), 256, 0, s>>>(scratch, nparts, count, (float*)mean_out);
  } else {
    mean_partial_kernel<double><<<dim3(nparts, nimg), 256, 0, s>>>((const double*)image, count, scratch);
    mean_final_kernel<double><<<dim3(1, nimg), 256, 0, s>>>(scratch, nparts, count, (double*)mean_out);
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
#ifndef GPA_PA_C10
#define GPA_PA_C10 8
#endif
#ifndef GPA_PA_C9
#define GPA_PA_C9 16
#endif
#ifndef GPA_PA_C11
#define GPA_PA_C11 4
#endif
#ifndef GPA_PA_C12
#define GPA_PA_C12 16
#endif
    int c = LG == 10 ? GPA_PA_C10 : (LG == 9 ? GPA_PA_C9 : (LG == 11 ? GPA_PA_C11 : (LG == 12 ? GPA_PA_C12 : 16)));
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

// ---------------------------------------------------------------------------
// pass A: x-axis filter on column tiles
// ---------------------------------------------------------------------------
template <class T, int LG, bool PADDED>
__global__ __launch_bounds__((PassAGeom<T, LG>::THREADS)) void passA_kernel(
    const T* __restrict__ image, const T* __restrict__ mean, int n0, int n1,
    const cpx<T>* __restrict__ cxb, const cpx<T>* __restrict__ sx, const cpx<T>* __restrict__ wxw,
    const cpx<T>* __restrict__ wxr, int extL, int extR,
    const typename HType<PADDED, T>::type* __restrict__ H,
    const cpx<T>* __restrict__ twtab, cpx<T>* __restrict__ Tout, int B, int bchunk, int stag_phases, int stag_ticks,
    int stag_first, int rot_mul) {
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
  int tile = xcd_tile(blockIdx.x, gridDim.x);
  if (stag_phases < 0) {
    // experiment (PA_STAG=-q): groups of q lines' worth of tiles dealt round-robin over the XCDs instead of one contiguous run
    // of tiles per XCD (the column strip an XCD writes then spans the whole row instead of 1 KB of it)
    constexpr int TPL0 = (128 / (int)(G::C * sizeof(cpx<T>))) > 0 ? (128 / (int)(G::C * sizeof(cpx<T>))) : 1;
    const int g = TPL0 * -stag_phases, x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (gridDim.x % (8 * g) == 0) tile = ((idx / g) * 8 + x) * g + idx % g;
  }
  const int y0 = tile * G::C + c * NT;
  // image stacks: blockIdx.z = image, its B planes behind those of the image before
  image += (size_t)blockIdx.z * n0 * n1;
  Tout += (size_t)blockIdx.z * B * n0 * n1;
  const T m = mean ? mean[blockIdx.z] : T(0);

  T val[NT][16];
  unsigned wrapmask = 0, rightmask = 0;   // slots of the left / right periodic extension (padded mode)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int y = y0 + n;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int slot = t + TPF * i;
      const int xs = axis_src(slot, n0, L, PADDED, extL, extR);
      val[n][i] = (y < n1 && xs >= 0) ? image[(size_t)xs * n1 + y] - m : T(0);
      if (PADDED && slot >= n0) { if (slot < n0 + extR) rightmask |= 1u << i; else wrapmask |= 1u << i; }
    }
  }
  typename F::KTw tw;
  F::load_twiddles(tw, twtab, t);
  if (stag_phases > 1 && (int)blockIdx.x < stag_first) {
    // phase stagger (PA_STAG): the workgroups that assemble one 128-byte line of an x-plane stay in step, but the lines'
    // owners start their plane loops 1/phases of a plane period apart, so that the drain of one group of CUs meets the
    // transforms of another instead of the whole chip storing -- and then computing -- at once.  Only the first wave of
    // workgroups waits; the later ones inherit the phase of the workgroup whose CU they take over.
    constexpr int TPL = (128 / (int)(G::C * sizeof(cpx<T>))) > 0 ? (128 / (int)(G::C * sizeof(cpx<T>))) : 1;
    const int ph = (tile / TPL) % stag_phases;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ph) {
      const unsigned long long t0 = wall_clock64(), wait = (unsigned long long)ph * (unsigned)stag_ticks;
      while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
  }

  const int b0 = blockIdx.y * bchunk;
  const int b1 = (b0 + bchunk < B) ? b0 + bchunk : B;
  // plane rotation (PA_ROT=<m>, default 1 for transforms of 4096 points and more, 0 = off): the owners of line q start their
  // plane loop at plane (q m) mod nb.  The workgroups of this kernel run in step (one per CU, same work), so without it the whole
  // chip stores 32-byte row pieces of ONE plane at a time, 32 KB apart; spread over the planes (128 MB apart) the same stores
  // drain in 3.5 instead of 6.2 us per plane: 0.81 -> 0.68 ms at 4096^2 f32, 1.77 -> 1.61 f64 (profiles/r06_passA_rotation.txt).
  // Every plane is computed exactly as before; only the order differs.
  const int nb = b1 - b0;
  int b = b0 + (nb > 0 ? (int)(((unsigned)(tile * G::C * (int)sizeof(cpx<T>) / 128) * (unsigned)rot_mul) % (unsigned)nb) : 0);
  for (int j = 0; j < nb; ++j, b = (b + 1 < b1 ? b + 1 : b0)) {
    F::refresh(tw);
    const cpx<T> base = cxb[(size_t)b * TPF + t];
    cpx<T> x[NT][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      cpx<T> ph = cmul(base, sx[b * 16 + i]);
      if (PADDED && ((wrapmask >> i) & 1)) ph = cmul(ph, wxw[b]);
      if (PADDED && ((rightmask >> i) & 1)) ph = cmul(ph, wxr[b]);
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
        const auto h = hload<PADDED, T>(Hb, i * TPF + t);
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
        struct alignas(2 * sizeof(cpx<T>)) Pair { cpx<T> a, b; };
        auto put = [&](int i) {
          const int slot = t + TPF * i;
          if (!PADDED || slot < n0) {
            Pair pr = {x[0][i], x[1][i]};
            *reinterpret_cast<Pair*>(&Tout[((size_t)b * n0 + slot) * n1 + y0]) = pr;
          }
        };
        {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
#ifdef GPA_PA_NOSTORE
            if (i > 0) continue;   // diagnosis only
#endif
            put(i);
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
// launchers
// ---------------------------------------------------------------------------
template <class T, int LG, bool PADDED>
static hipError_t run_passA(const Axis& a0, int n1, const void* image, const void* mean,
                            const SweepTables& tb, const void* Hx, const void* tw0, void* Tbuf,
                            int B, hipStream_t s, int nimg) {
  using G = PassAGeom<T, LG>;
  if constexpr (!G::FITS) {
    return hipErrorInvalidValue;
  } else {
    auto kern = passA_kernel<T, LG, PADDED>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int tiles = (n1 + G::C - 1) / G::C;
    // split the lock-in loop over grid.y when there are too few column tiles to fill 256 CUs
    int ysplit = 1;
    while (tiles * ysplit < 512 && ysplit < B) ysplit *= 2;
    if (ysplit > B) ysplit = B;
    const int bchunk = (B + ysplit - 1) / ysplit;
    dim3 grid(tiles, (B + bchunk - 1) / bchunk, nimg);
    // PA_STAG=<phases> [PA_STAG_TICKS=<10-ns ticks per phase step>]: see the kernel
    const int stag = opt_set(OPT_PA_STAG) ? (int)opt(OPT_PA_STAG).num : 0;
    const int stag_ticks = opt_set(OPT_PA_STAG_TICKS) ? (int)opt(OPT_PA_STAG_TICKS).num : 800 / (stag > 1 ? stag : 1);
    GPA_PROF("passA_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>(
        (const T*)image, (const T*)mean, a0.n, n1, (const cpx<T>*)tb.cxb, (const cpx<T>*)tb.sx,
        (const cpx<T>*)tb.wxw, (const cpx<T>*)tb.wxr, a0.extL, a0.extR,
        (const typename HType<PADDED, T>::type*)Hx, (const cpx<T>*)tw0, (cpx<T>*)Tbuf, B, bchunk, stag,
        stag_ticks, device_cus(), opt_set(OPT_PA_ROT) ? (int)opt(OPT_PA_ROT).num : (LG >= 12 ? 1 : 0));
    return hipGetLastError();
  }
}


hipError_t launch_passA(int dtype, const Axis& a0, int n1, const void* image, const void* mean,
                        const SweepTables& tb, const void* Hx, const void* tw0, void* Tbuf,
                        int B, hipStream_t s, int nimg) {
#define CASE_A(LG)                                                                                \
  case LG:                                                                                        \
    if (dtype == 0)                                                                               \
      return a0.padded ? run_passA<float, LG, true>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s, nimg) \
                       : run_passA<float, LG, false>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s, nimg); \
    else                                                                                          \
      return a0.padded ? run_passA<double, LG, true>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s, nimg) \
                       : run_passA<double, LG, false>(a0, n1, image, mean, tb, Hx, tw0, Tbuf, B, s, nimg);
  switch (a0.lg) { GPA_FOR_LG(CASE_A) }
#undef CASE_A
  return hipErrorInvalidValue;
}

hipError_t launch_passB(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy,
                        const void* tw1, const SweepTables& tb, int P, int K, bool select,
                        void* out, int32_t* kidx, hipStream_t s, int nimg, int Bx) {
#define CALL_B(T, LG, PD)                                                                                                   \
  (select ? run_passB<T, LG, PD, PB_SELECT>(a1, n0, Tbuf, Hy, tw1, tb, P, K, out, kidx, nullptr, nullptr, s, 1, nimg, Bx)    \
          : run_passB<T, LG, PD, PB_ALL>(a1, n0, Tbuf, Hy, tw1, tb, P, K, out, kidx, nullptr, nullptr, s, 1, nimg, Bx))
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

int spec_index_rt(int lg, int tid, int reg, int elems) {
  if (elems == 8) {
#define CASE_S8(LG) \
  case LG: return WgFFT<float, LG, 8>::spec_index(tid, reg);
    switch (lg) { CASE_S8(6) CASE_S8(7) CASE_S8(8) CASE_S8(9) CASE_S8(10) CASE_S8(11) CASE_S8(12) }
#undef CASE_S8
    return -1;
  }
#define CASE_S(LG) \
  case LG: return WgFFT<float, LG>::spec_index(tid, reg);
  switch (lg) { GPA_FOR_LG(CASE_S) }
#undef CASE_S
  return -1;
}

}  // namespace gpa
