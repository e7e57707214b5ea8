// The lock-in sweep on an axis whose length is NOT a power of two, at that length.
//
// The power-of-two engine evaluates the reference's circular filter of length n (geometric_phase_analysis.py:72-75:
// ifft2(fft2(img * carrier) * G)) on a periodically extended row of L >= n + 2E samples, L a power of two: 500 -> 1024,
// 1000 -> 2048, 2000 -> 4096 -- twice the samples per transform, and for rows from the 2048 class the shared-forward
// kernel in its zero-padded mode.  A smooth n = 2^a 3^b 5^c 7^d 11^e 13^f has a transform of its own (gpa_mrfft.h, the
// Stockham engine of the unwrap's generic path, resident in LDS): these kernels are pass A (gpa_sweep.hip) and the
// per-candidate pass B (gpa_passb.h) on that engine, in periodic mode at length n -- the reference's arithmetic
// literally: carrier multiply, DFT_n, Gaussian bin by bin, inverse DFT_n.  No extension, no end fix, no wrap factors.
//
//   pass A   a workgroup = nf adjacent columns x T threads; the image column stays in registers for all x-planes;
//            per plane: carrier -> LDS -> forward -> H[k] / n, conjugate -> forward -> conjugate -> T[plane][x][y]
//            (the inverse transform is conj(DFT(conj .)): one engine, one twiddle table)
//   pass B   a workgroup = nf rows x T threads; per candidate the same chain on the row of its x-plane, then the
//            selection of gpa_passb.h in registers -- every mode of that kernel (all lock-ins, best-of-K, the gated
//            selection of wfr4, selection + per-candidate phases, candidates split over grid.z) with the same rules.
// A thread owns the samples t, t + T, t + 2T, ... of its transform on both sides of the transform (natural order in
// and out), so a pass result is read and the next input written by the same thread in the same LDS slots: no barrier
// between candidates beyond the engine's own.
#include "gpa_internal.h"
#include "gpa_passb.h"

namespace gpa {
namespace {

constexpr int MRS_LDS_MAX = 160 * 1024 - 512;

// ---------------------------------------------------------------------------
// pass A
// ---------------------------------------------------------------------------
template <class T, int MAXT, int E>
__global__ __launch_bounds__(MAXT) void passA_mr_kernel(
    const T* __restrict__ image, const T* __restrict__ mean, int n0, int n1, const cpx<T>* __restrict__ cxb,
    const cpx<T>* __restrict__ sx, const T* __restrict__ H, const cpx<T>* __restrict__ W, const MrPlan pl, int rs,
    cpx<T>* __restrict__ Tout, int B, int bchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tn = pl.T, nf = blockDim.x / Tn;
  // column fastest in the thread index: neighbouring lanes read / write neighbouring columns of one row
  const int f = threadIdx.x % nf, t = threadIdx.x / nf;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * rs;
  const int y = xcd_tile(blockIdx.x, gridDim.x) * nf + f;
  const bool vy = y < n1;
  image += (size_t)blockIdx.z * n0 * n1;
  Tout += (size_t)blockIdx.z * B * n0 * n1;
  const T m = mean ? mean[blockIdx.z] : T(0);
  T val[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int x = t + Tn * i;
    val[i] = (x < n0 && vy) ? image[(size_t)x * n1 + y] - m : T(0);
  }
  const int b0 = blockIdx.y * bchunk;
  const int b1 = (b0 + bchunk < B) ? b0 + bchunk : B;
  for (int b = b0; b < b1; ++b) {
    const cpx<T> base = cxb[(size_t)b * Tn + t];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int x = t + Tn * i;
      if (x < n0) {
        const cpx<T> ph = cmul(base, sx[b * 16 + i]);   // exp(2 pi i wx x) at x = t + Tn i
        lds[mr_pad(x)] = {val[i] * ph.x, val[i] * ph.y};
      }
    }
    __syncthreads();
    mr_run<MAXT>(lds, pl, W, t);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = t + Tn * i;
      if (k < n0) {
        const cpx<T> v = lds[mr_pad(k)];
        const T h = H[k];
        lds[mr_pad(k)] = {v.x * h, -v.y * h};
      }
    }
    __syncthreads();
    mr_run<MAXT>(lds, pl, W, t);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int x = t + Tn * i;
      if (x < n0 && vy) {
        const cpx<T> v = lds[mr_pad(x)];
        Tout[((size_t)b * n0 + x) * n1 + y] = {v.x, -v.y};
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pass B
// ---------------------------------------------------------------------------
template <class T, int E, int MODE>
__global__ __launch_bounds__(256) void passB_mr_kernel(
    const cpx<T>* __restrict__ Tin, int n0, int n1, const T* __restrict__ H, const cpx<T>* __restrict__ W, const MrPlan pl,
    int rs, const int* __restrict__ planeof, const cpx<T>* __restrict__ cyb, const cpx<T>* __restrict__ sy,
    const cpx<T>* __restrict__ dx, const cpx<T>* __restrict__ dy, int K, cpx<T>* __restrict__ out,
    int32_t* __restrict__ kidx, const uint8_t* __restrict__ gate, T* __restrict__ psi, int P, int Bx) {
  constexpr bool SELECT = MODE != PB_ALL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tn = pl.T;
  const int tid = threadIdx.x % Tn, f = threadIdx.x / Tn;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * rs;
  const int nf = blockDim.x / Tn;
  const int row = blockIdx.x * nf + f;
  const bool valid = row < n0;
  // image stacks: blockIdx.y = image * P + peak (see passB_kernel)
  const int p = blockIdx.y, pt = p % P, img = p / P;

  cpx<T> best[E];
  int bidx[E];
#pragma unroll
  for (int i = 0; i < E; ++i) { best[i] = {T(0), T(0)}; bidx[i] = -1; }
  int k0 = 0, nk = SELECT ? K : 1;
  if constexpr (MODE == PB_PART) {
    const int kc = (K + (int)gridDim.z - 1) / (int)gridDim.z;
    k0 = (int)blockIdx.z * kc;
    nk = k0 + kc < K ? k0 + kc : K;
  }
  for (int k = k0; k < nk; ++k) {
    const int b = SELECT ? pt * K + k : pt;
    const cpx<T>* src = Tin + (((size_t)img * Bx + planeof[b]) * n0 + (valid ? row : 0)) * n1;
    const cpx<T> cbase = cyb[(size_t)b * Tn + tid];
    {
      cpx<T> x[E];
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int yy = tid + Tn * i;
        x[i] = src[yy < n1 ? yy : 0];
      }
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int yy = tid + Tn * i;
        if (yy < n1) lds[mr_pad(yy)] = cmul(x[i], cmul(cbase, sy[b * 16 + i]));   // exp(2 pi i wy y) at y = tid + Tn i
      }
    }
    __syncthreads();
    mr_run<256>(lds, pl, W, tid);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int kk = tid + Tn * i;
      if (kk < n1) {
        const cpx<T> v = lds[mr_pad(kk)];
        const T h = H[kk];
        lds[mr_pad(kk)] = {v.x * h, -v.y * h};
      }
    }
    __syncthreads();
    mr_run<256>(lds, pl, W, tid);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int yy = tid + Tn * i;
      cpx<T> v = lds[mr_pad(yy < n1 ? yy : 0)];
      v.y = -v.y;
      if constexpr (SELECT) {
        const T a = v.x * v.x + v.y * v.y;
        const T ab = best[i].x * best[i].x + best[i].y * best[i].y;
        if constexpr (MODE == PB_GATED) {
          const int j = bidx[i] < 0 ? 0 : bidx[i];   // wfr4: klist[0] until something is accepted (see passB_kernel)
          if (yy < n1 && a > ab && gate[(size_t)j * K + k]) { best[i] = v; bidx[i] = k; }
        } else {
          if (yy < n1 && a > ab) { best[i] = v; bidx[i] = k; }
        }
        if constexpr (MODE == PB_PHASES) {
          if (valid && yy < n1) psi[((size_t)(p * K + k) * n0 + row) * n1 + yy] = -atan2(v.y, v.x);
        }
      } else {
        best[i] = v;
      }
    }
  }
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int yy = tid + Tn * i;
    if (yy < n1) {
      const size_t o = ((size_t)p * n0 + row) * n1 + yy;
      if constexpr (MODE == PB_PART) {
        const size_t slab = (size_t)gridDim.y * n0 * n1;
        out[(size_t)blockIdx.z * slab + o] = best[i];
        kidx[(size_t)blockIdx.z * slab + o] = bidx[i];
      } else if constexpr (SELECT) {
        cpx<T> v = {T(0), T(0)};
        if (bidx[i] >= 0) {
          const size_t bb = (size_t)pt * K + bidx[i];
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

template <class T, int MAXT, int E>
hipError_t run_passA_mr(const Axis& a0, int n1, const void* image, const void* mean, const SweepTables& tb, void* Tbuf, int B,
                        hipStream_t s, int nimg, int nf) {
  auto kern = passA_mr_kernel<T, MAXT, E>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), MRS_LDS_MAX, lds_set);
  if (e != hipSuccess) return e;
  const int Tn = a0.pl.T;
  // region stride = 16 (mod 32) elements: the nf columns of a wavefront then fall on disjoint bank groups
  int rs = mr_lds_elems(a0.n);
  if (nf > 1) rs += (16 - (rs & 31) + 32) & 31;
  const size_t lds = (size_t)nf * rs * sizeof(cpx<T>);
  const int tiles = (n1 + nf - 1) / nf;
  int ysplit = 1;
  while (tiles * ysplit < 512 && ysplit < B) ysplit *= 2;
  if (ysplit > B) ysplit = B;
  const int bchunk = (B + ysplit - 1) / ysplit;
  dim3 grid(tiles, (B + bchunk - 1) / bchunk, nimg);
  GPA_PROF("passA_kernel", s);
  kern<<<grid, nf * Tn, lds, s>>>((const T*)image, (const T*)mean, a0.n, n1, (const cpx<T>*)tb.cxb, (const cpx<T>*)tb.sx,
                                  (const T*)a0.natH, (const cpx<T>*)a0.natW, a0.pl, rs, (cpx<T>*)Tbuf, B, bchunk);
  return hipGetLastError();
}

template <class T, int E, int MODE>
hipError_t run_passB_mr(const Axis& a1, int n0, const void* Tbuf, const SweepTables& tb, int P, int K, void* out, int32_t* kidx,
                        const uint8_t* gate, void* psi, hipStream_t s, int ksplit, int nimg, int Bx) {
  auto kern = passB_mr_kernel<T, E, MODE>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), MRS_LDS_MAX, lds_set);
  if (e != hipSuccess) return e;
  const int Tn = a1.pl.T, rs = mr_lds_elems(a1.n);
  int nf = 256 / Tn;
  if (nf < 1) nf = 1;
  // (small images: one row per workgroup rather than fewer workgroups than twice the CUs)
  while (nf > 1 && ((n0 + nf - 1) / nf) * P * nimg * ksplit < 512) nf /= 2;
  while (nf > 1 && (size_t)nf * rs * sizeof(cpx<T>) > (size_t)MRS_LDS_MAX) nf /= 2;
  const size_t lds = (size_t)nf * rs * sizeof(cpx<T>);
  dim3 grid((n0 + nf - 1) / nf, P * nimg, ksplit);
  GPA_PROF("passB_kernel", s);
  kern<<<grid, nf * Tn, lds, s>>>((const cpx<T>*)Tbuf, n0, a1.n, (const T*)a1.natH, (const cpx<T>*)a1.natW, a1.pl, rs,
                                  tb.planeof, (const cpx<T>*)tb.cyb, (const cpx<T>*)tb.sy, (const cpx<T>*)tb.dx,
                                  (const cpx<T>*)tb.dy, K, (cpx<T>*)out, kidx, gate, (T*)psi, P, Bx);
  return hipGetLastError();
}

template <class T, int E>
hipError_t passB_mr_mode(int mode, const Axis& a1, int n0, const void* Tbuf, const SweepTables& tb, int P, int K, void* out,
                         int32_t* kidx, const uint8_t* gate, void* psi, hipStream_t s, int ksplit, int nimg, int Bx) {
  switch (mode) {
    case PB_ALL: return run_passB_mr<T, E, PB_ALL>(a1, n0, Tbuf, tb, P, K, out, kidx, gate, psi, s, 1, nimg, Bx);
    case PB_SELECT: return run_passB_mr<T, E, PB_SELECT>(a1, n0, Tbuf, tb, P, K, out, kidx, gate, psi, s, 1, nimg, Bx);
    case PB_GATED: return run_passB_mr<T, E, PB_GATED>(a1, n0, Tbuf, tb, P, K, out, kidx, gate, psi, s, 1, nimg, Bx);
    case PB_PHASES: return run_passB_mr<T, E, PB_PHASES>(a1, n0, Tbuf, tb, P, K, out, kidx, gate, psi, s, 1, nimg, Bx);
    case PB_PART: return run_passB_mr<T, E, PB_PART>(a1, n0, Tbuf, tb, P, K, out, kidx, gate, psi, s, ksplit, nimg, Bx);
  }
  return hipErrorInvalidValue;
}

}  // namespace

// elements per thread the kernels are instantiated for
static int mr_elems(const Axis& a) {
  const int e = (a.n + a.pl.T - 1) / a.pl.T;
  return e <= 4 ? 4 : (e <= 8 ? 8 : 16);
}

hipError_t launch_passA_mr(int dtype, const Axis& a0, int n1, const void* image, const void* mean, const SweepTables& tb,
                           void* Tbuf, int Bx, hipStream_t s, int nimg) {
  if (!a0.native || !a0.natW || !a0.natH || a0.pl.T > 256) return hipErrorInvalidValue;
  const int Tn = a0.pl.T, E = mr_elems(a0);
  const size_t csz = dtype == 0 ? 8 : 16;
  // columns per workgroup: 8 (64-byte pieces of a row of T in f32) where 1024 threads and LDS allow, fewer while the
  // grid would not cover the chip twice
  // (1024-thread workgroups only where the kernel fits their 128 registers: f32 with at most 8 samples per thread)
  const int maxthreads = (dtype == 0 && E <= 8) ? 1024 : 512;
  int nf = maxthreads / Tn;
  if (nf > 8) nf = 8;
  if (nf < 1) nf = 1;
  while (nf > 1 && (size_t)nf * (mr_lds_elems(a0.n) + 32) * csz > (size_t)MRS_LDS_MAX) nf /= 2;
  while (nf > 2 && ((n1 + nf - 1) / nf) * Bx * nimg < 512) nf /= 2;
  const int threads = nf * Tn;
#define CALL_A(T, MAXT, EE) run_passA_mr<T, MAXT, EE>(a0, n1, image, mean, tb, Tbuf, Bx, s, nimg, nf)
#define PICK_A(T, EE) (threads <= 256 ? CALL_A(T, 256, EE) : (threads <= 512 ? CALL_A(T, 512, EE) : CALL_A(T, 1024, EE)))
  if (dtype == 0) return E == 4 ? PICK_A(float, 4) : (E == 8 ? PICK_A(float, 8) : PICK_A(float, 16));
  return E == 4 ? PICK_A(double, 4) : (E == 8 ? PICK_A(double, 8) : PICK_A(double, 16));
#undef PICK_A
#undef CALL_A
}

hipError_t launch_passB_mr(int dtype, const Axis& a1, int n0, const void* Tbuf, const SweepTables& tb, int P, int K, int mode,
                           void* out, int32_t* kidx, const uint8_t* gate, void* psi, hipStream_t s, int ksplit, int nimg,
                           int Bx) {
  if (!a1.native || !a1.natW || !a1.natH || a1.pl.T > 256) return hipErrorInvalidValue;
  const int E = mr_elems(a1);
#define CALL_B(T, EE) passB_mr_mode<T, EE>(mode, a1, n0, Tbuf, tb, P, K, out, kidx, gate, psi, s, ksplit, nimg, Bx)
  if (dtype == 0) return E == 4 ? CALL_B(float, 4) : (E == 8 ? CALL_B(float, 8) : CALL_B(float, 16));
  return E == 4 ? CALL_B(double, 4) : (E == 8 ? CALL_B(double, 8) : CALL_B(double, 16));
#undef CALL_B
}

}  // namespace gpa
