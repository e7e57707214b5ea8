// Plain 2-D DFTs of any image size and the kernels of a9 -- Moisan's periodic + smooth decomposition, `moisan2011.per`
// (call site geometric_phase_analysis.py:429) -- that run on them.  Interface and the three engines: gpa_dft.h.
// Entry points: gpa_per_dft / gpa_per / gpa_find_peaks / gpa_gaussian_deconvolve (gpa_api_spectral.hip).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "gpa_dct.h"
#include "gpa_dft.h"
#include "gpa_internal.h"

namespace gpa {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// DFT_POW2: the register engine at the axis' own length
// ---------------------------------------------------------------------------------------------------------------------
// rows: NF transforms per workgroup (GenGeom: 256 threads up to 4096 points, one row per workgroup above), lanes along the row.
// REAL_IN: transform g packs the real rows 2g and 2g + 1 as z = a + i b; Z_k = A_k + i B_k with A, B Hermitian, so
//   A_k = (Z_k + conj Z_{-k}) / 2,  B_k = (Z_k - conj Z_{-k}) / 2i  -- both full rows leave from one transform.
// The forward transform leaves the spectrum digit-scrambled in registers; one more trip through LDS (scatter to the bin's own
// slot, read back in the natural layout) makes the stores coalesced and gives every thread the partner bin -k.
// HALF (real input only): bins 0 ... L/2 of every row, row pitch `opitch` -- the other half of a real image's spectrum is its
// mirror image, U^[-q, -r] = conj U^[q, r], and a consumer that wants |U^| (f-3) never needs it.
template <class T, int LG, bool REAL_IN, bool HALF = false>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void p2_rows_kernel(const void* __restrict__ in_, cpx<T>* __restrict__ out,
                                                                          int nrows, const cpx<T>* __restrict__ twtab,
                                                                          size_t opitch) {
  using F = WgFFT<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF, L = F::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int g = blockIdx.x * G::NF + f;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
  if constexpr (REAL_IN) {
    const T* in = static_cast<const T*>(in_);
    const int ra = 2 * g, rb = 2 * g + 1;
    const T* pa = in + (size_t)(ra < nrows ? ra : 0) * L;
    const T* pb = in + (size_t)(rb < nrows ? rb : 0) * L;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int slot = tid + TPF * i;
      x[i] = {ra < nrows ? pa[slot] : T(0), rb < nrows ? pb[slot] : T(0)};
    }
  } else {
    const cpx<T>* in = static_cast<const cpx<T>*>(in_) + (size_t)(g < nrows ? g : 0) * L;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = g < nrows ? in[tid + TPF * i] : cpx<T>{T(0), T(0)};
  }
  F::forward(x, lds, tid, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) lds[F::pad(F::spec_index(tid, i))] = x[i];
  __syncthreads();
  if constexpr (REAL_IN) {
    const int ra = 2 * g, rb = 2 * g + 1;
    if (ra >= nrows) return;
    cpx<T>* oa = out + (size_t)ra * opitch;
    cpx<T>* ob = out + (size_t)(rb < nrows ? rb : ra) * opitch;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = tid + TPF * i;
      if (HALF && k > L / 2) continue;
      const cpx<T> zk = lds[F::pad(k)], zm = lds[F::pad((L - k) & (L - 1))];
      oa[k] = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};
      if (rb < nrows) ob[k] = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};
    }
  } else {
    if (g >= nrows) return;
    cpx<T>* o = out + (size_t)g * opitch;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[tid + TPF * i] = lds[F::pad(tid + TPF * i)];
  }
}

// columns: tiles of C adjacent columns (as many as 1024 threads and the LDS hold, at most 16), two columns per thread in f32
// so that a row piece of a thread is one 16-byte access; XCD-aware tile order (neighbouring tiles share 128-byte lines).
template <class T, int LG>
struct ColGeom {
  using F = WgFFT<T, LG>;
  static constexpr int cols() {
    int c = 16;
    while (c > 1 && (c * F::TPF > 1024 || (size_t)c * F::LDS_ELEMS * sizeof(cpx<T>) > 140 * 1024)) c /= 2;
    return c;
  }
  static constexpr int C = cols();
  static constexpr int NT = (sizeof(T) == 4 && C >= 2) ? 2 : 1;   // columns per thread
  static constexpr int CT = C / NT;                               // columns side by side in the thread index
  static constexpr int REGION = CT * F::LDS_ELEMS;                // one LDS region per column-of-a-thread, CT columns interleaved
  static constexpr int THREADS = CT * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NT * REGION * sizeof(cpx<T>);
  static constexpr bool FITS = (size_t)F::LDS_ELEMS * sizeof(cpx<T>) <= 140 * 1024;
};

template <class T, int LG>
__global__ __launch_bounds__((ColGeom<T, LG>::THREADS)) void p2_cols_kernel(cpx<T>* __restrict__ Z, int n1, size_t pitch,
                                                                          const cpx<T>* __restrict__ twtab) {
  using F = WgFFT<T, LG>;
  using G = ColGeom<T, LG>;
  constexpr int CT = G::CT, NT = G::NT, TPF = F::TPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = threadIdx.x % CT, t = threadIdx.x / CT;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + c;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int y0 = tile * G::C + c * NT;
  struct alignas(2 * sizeof(cpx<T>)) Pair { cpx<T> a, b; };
  const bool paired = NT == 2 && (pitch & 1) == 0 && y0 + 1 < n1 && (reinterpret_cast<size_t>(Z) & (2 * sizeof(cpx<T>) - 1)) == 0;
  cpx<T> x[NT][16];
  if (paired) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const Pair pr = *reinterpret_cast<const Pair*>(&Z[(size_t)(t + TPF * i) * pitch + y0]);
      x[0][i] = pr.a;
      x[NT - 1][i] = pr.b;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        x[n][i] = y0 + n < n1 ? Z[(size_t)(t + TPF * i) * pitch + y0 + n] : cpx<T>{T(0), T(0)};
  }
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, t);
  F::template forward_multi<NT, CT>(x, lds, G::REGION, t, tw);
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[n * G::REGION + CT * F::pad(F::spec_index(t, i))] = x[n][i];
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) x[n][i] = lds[n * G::REGION + CT * F::pad(t + TPF * i)];
  if (paired) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      Pair pr = {x[0][i], x[NT - 1][i]};
      *reinterpret_cast<Pair*>(&Z[(size_t)(t + TPF * i) * pitch + y0]) = pr;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n)
      if (y0 + n < n1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Z[(size_t)(t + TPF * i) * pitch + y0 + n] = x[n][i];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// DFT_BLUE: chirp-z in one workgroup (round 1's kernels)
// ---------------------------------------------------------------------------------------------------------------------
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_rowdft_kernel(
    cpx<T>* __restrict__ Z, int n0, int n, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ chirp,
    const cpx<T>* __restrict__ bspec) {
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int row = blockIdx.x * G::NF + f;
  const bool valid = row < n0;
  cpx<T>* zr = Z + (size_t)(valid ? row : 0) * n;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    x[i] = slot < n ? zr[slot] : cpx<T>{T(0), T(0)};
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) zr[slot] = x[i];
  }
}

template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_coldft_kernel(
    cpx<T>* __restrict__ Z, int n, int n1, size_t pitch, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ chirp,
    const cpx<T>* __restrict__ bspec) {
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF, NF = G::NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = threadIdx.x % NF, tid = threadIdx.x / NF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int col = blockIdx.x * NF + f;
  const bool valid = col < n1;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    x[i] = (valid && slot < n) ? Z[(size_t)slot * pitch + col] : cpx<T>{T(0), T(0)};
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) Z[(size_t)slot * pitch + col] = x[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// DFT_BIG: chirp-z on a two-level transform through HBM.  L = M1 M2, sample m = l1 M2 + l2, bin k = k1 + M1 k2:
//   X[k1 + M1 k2] = sum_l2 w_M2^(l2 k2) [ w_L^(l2 k1) sum_l1 x[l1 M2 + l2] w_M1^(l1 k1) ]
// big_fwdA: the bracket (strided sub-transforms of length M1 over l1, chirp on load, twiddle on store) -> W
// big_mid : per k1 the transform over l2, the table FFT_L(b) / L, and straight back (inverse over k2): the spectrum is never
//           stored, so its order never matters -- sub-bin k1 stays in the slot the register engine left it in
// big_invA: conj twiddle, inverse over k1, chirp -> the n outputs of the line
// "lines" are rows (COLS = false: lanes along l2, line = blockIdx.y) or columns (COLS = true: lanes along the lines).
// ---------------------------------------------------------------------------------------------------------------------
template <class T, int LG>
struct TileGeom {
  using F = WgFFT<T, LG>;
  static constexpr int NF = 256 / F::TPF;                         // lines (or l2 values) side by side: 64 .. 8 for 64 .. 512 points
  static constexpr size_t LDS_BYTES = (size_t)NF * F::LDS_ELEMS * sizeof(cpx<T>);
};

template <class T, int LG1, bool COLS>
__global__ __launch_bounds__(256) void big_fwdA_kernel(const cpx<T>* __restrict__ Z, size_t zpitch, int n, int nlines,
                                                       cpx<T>* __restrict__ W, size_t wpitch, int lg2,
                                                       const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ twL,
                                                       const cpx<T>* __restrict__ tw1) {
  using F = WgFFT<T, LG1>;
  constexpr int TPF = F::TPF, NF = TileGeom<T, LG1>::NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = threadIdx.x % NF, tid = threadIdx.x / NF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f;
  const int line = COLS ? blockIdx.x * NF + f : blockIdx.y;
  const int l2 = COLS ? blockIdx.y : blockIdx.x * NF + f;
  const bool valid = line < nlines;
  typename F::Twiddles tw;
  F::load_twiddles(tw, tw1, tid);
  cpx<T> x[1][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = ((tid + TPF * i) << lg2) + l2;
    x[0][i] = {T(0), T(0)};
    if (valid && m < n) x[0][i] = cmulc(COLS ? Z[(size_t)m * zpitch + line] : Z[(size_t)line * zpitch + m], chirp[m]);
  }
  F::template forward_multi<1, NF>(x, lds, 0, tid, tw);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i, k1 = F::spec_index(tid, i);
    const cpx<T> v = cmul(x[0][i], twL[k1 * l2]);
    const size_t pos = ((size_t)slot << lg2) + l2;
    if (COLS) W[pos * wpitch + line] = v;
    else W[(size_t)line * wpitch + pos] = v;
  }
}

template <class T, int LG1, bool COLS>
__global__ __launch_bounds__(256) void big_invA_kernel(cpx<T>* __restrict__ Z, size_t zpitch, int n, int nlines,
                                                       const cpx<T>* __restrict__ W, size_t wpitch, int lg2,
                                                       const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ twL,
                                                       const cpx<T>* __restrict__ tw1) {
  using F = WgFFT<T, LG1>;
  constexpr int TPF = F::TPF, NF = TileGeom<T, LG1>::NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = threadIdx.x % NF, tid = threadIdx.x / NF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f;
  const int line = COLS ? blockIdx.x * NF + f : blockIdx.y;
  const int l2 = COLS ? blockIdx.y : blockIdx.x * NF + f;
  const bool valid = line < nlines;
  typename F::Twiddles tw;
  F::load_twiddles(tw, tw1, tid);
  cpx<T> x[1][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i, k1 = F::spec_index(tid, i);
    const size_t pos = ((size_t)slot << lg2) + l2;
    x[0][i] = {T(0), T(0)};
    if (valid) x[0][i] = cmulc(COLS ? W[pos * wpitch + line] : W[(size_t)line * wpitch + pos], twL[k1 * l2]);
  }
  F::template inverse_multi<1, NF>(x, lds, 0, tid, tw);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = ((tid + TPF * i) << lg2) + l2;
    if (m < n) {
      const cpx<T> v = cmulc(x[0][i], chirp[m]);
      if (COLS) Z[(size_t)m * zpitch + line] = v;
      else Z[(size_t)line * zpitch + m] = v;
    }
  }
}

// ROWS: the M2 samples of (line, slot) are contiguous: transform index g = line M1 + slot, lanes along the piece, NF pieces per
// workgroup with their own LDS regions.  COLS: lanes along the lines, blockIdx.y = slot, the NF lines interleaved in LDS.
template <class T, int LG2, bool COLS>
__global__ __launch_bounds__(256) void big_mid_kernel(cpx<T>* __restrict__ W, size_t wpitch, int nlines, int lg1,
                                                      const cpx<T>* __restrict__ tab, const cpx<T>* __restrict__ tw2) {
  using F = WgFFT<T, LG2>;
  constexpr int TPF = F::TPF, NF = TileGeom<T, LG2>::NF, M2 = F::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cpx<T> x[1][16];
  typename F::Twiddles tw;
  if constexpr (COLS) {
    const int f = threadIdx.x % NF, tid = threadIdx.x / NF;
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f;
    const int line = blockIdx.x * NF + f, slot = blockIdx.y;
    const bool valid = line < nlines;
    F::load_twiddles(tw, tw2, tid);
    cpx<T>* base = W + (size_t)slot * M2 * wpitch + (valid ? line : 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[0][i] = valid ? base[(size_t)(tid + TPF * i) * wpitch] : cpx<T>{T(0), T(0)};
    F::template forward_multi<1, NF>(x, lds, 0, tid, tw);
    const cpx<T>* tb = tab + (size_t)slot * M2;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[0][i] = cmul(x[0][i], tb[i * TPF + tid]);
    F::template inverse_multi<1, NF>(x, lds, 0, tid, tw);   // (first writes go where this thread last read: no barrier)
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) base[(size_t)(tid + TPF * i) * wpitch] = x[0][i];
  } else {
    const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * F::LDS_ELEMS;
    const size_t g = (size_t)blockIdx.x * NF + f, total = (size_t)nlines << lg1;
    const bool valid = g < total;
    const int slot = (int)(g & ((size_t(1) << lg1) - 1));
    F::load_twiddles(tw, tw2, tid);
    cpx<T>* base = W + (valid ? g : 0) * M2;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[0][i] = valid ? base[tid + TPF * i] : cpx<T>{T(0), T(0)};
    F::template forward_multi<1, 1>(x, lds, 0, tid, tw);
    const cpx<T>* tb = tab + (size_t)slot * M2;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[0][i] = cmul(x[0][i], tb[i * TPF + tid]);
    F::template inverse_multi<1, 1>(x, lds, 0, tid, tw);
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) base[tid + TPF * i] = x[0][i];
  }
}

// real image -> complex array (engines other than POW2 transform complex rows in place)
template <class T>
__global__ __launch_bounds__(256) void real_to_complex_kernel(const T* __restrict__ u, cpx<T>* __restrict__ Z, size_t count) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) Z[i] = {u[i], T(0)};
}

// a9.  v (the border-jump image) is non-zero on the four borders only, so its 2-D DFT
// factorises:  v^[q,r] = D0[r] (1 - e^{2 pi i q/n0}) + D1[q] (1 - e^{2 pi i r/n1}),
// D0 = DFT(u[n0-1,:] - u[0,:]), D1 = DFT(u[:,n1-1] - u[:,0]): two 1-D DFTs instead of a
// second 2-D one, and no cancellation against the (much larger) image spectrum.
template <class T>
__global__ __launch_bounds__(256) void per_borders_kernel(const T* __restrict__ u, int n0, int n1, cpx<T>* __restrict__ d0,
                                                         cpx<T>* __restrict__ d1) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n1) d0[j] = {u[(size_t)(n0 - 1) * n1 + j] - u[j], T(0)};
  if (j < n0) d1[j] = {u[(size_t)j * n1 + n1 - 1] - u[(size_t)j * n1], T(0)};
}

// per-axis factors of the border image's spectrum, in double once per plan: e[j] = 1 - exp(2 pi i j / n) (real part as
// 2 sin^2(pi j / n): no cancellation), s[j] = sin^2(pi j / n).  Layout of the table: s0[n0], s1[n1] doubles, e0[n0], e1[n1] complex.
template <class T>
__global__ __launch_bounds__(256) void per_tables_kernel(int n0, int n1, double* __restrict__ s0, double* __restrict__ s1,
                                                        cpx<T>* __restrict__ e0, cpx<T>* __restrict__ e1) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  for (int ax = 0; ax < 2; ++ax) {
    const int n = ax ? n1 : n0;
    if (j >= n) continue;
    const double sh = sinpi((double)j / n);
    double sn, cs;
    sincospi(2.0 * (double)j / n, &sn, &cs);
    (ax ? s1 : s0)[j] = sh * sh;
    (ax ? e1 : e0)[j] = {(T)(2.0 * sh * sh), (T)(-sn)};
  }
}

// P^ = U^ - V^ / (2 cos(2 pi q / n0) + 2 cos(2 pi r / n1) - 4), 2 cos a + 2 cos b - 4 = -4 (sin^2(a/2) + sin^2(b/2)).
// ABS: what f-3 wants of it, |fftshift(P^)| with the DC bin exactly 0 (geometric_phase_analysis.py:427-430), in ONE pass:
// the thread of output pixel (si, sj) reads bin (si - n0/2, sj - n1/2) mod (n0, n1).
template <class T, bool ABS>
__global__ __launch_bounds__(256) void per_combine_kernel(const cpx<T>* __restrict__ Uh, const cpx<T>* __restrict__ D0,
                                                         const cpx<T>* __restrict__ D1, int n0, int n1,
                                                         const double* __restrict__ s0, const double* __restrict__ s1,
                                                         const cpx<T>* __restrict__ e0, const cpx<T>* __restrict__ e1,
                                                         void* __restrict__ out_, size_t upitch, int half) {
  const int oj = blockIdx.x * 256 + threadIdx.x, oi = blockIdx.y;
  if (oj >= n1) return;
  int q = oi, r = oj;
  if constexpr (ABS) {
    q = oi + n0 - n0 / 2;
    q = q >= n0 ? q - n0 : q;
    r = oj + n1 - n1 / 2;
    r = r >= n1 ? r - n1 : r;
    // u_hat holds columns 0 ... n1/2 only: |P^[q, r]| = |P^[-q, -r]|
    if (half && r > n1 / 2) { r = n1 - r; q = q ? n0 - q : 0; }
  }
  const cpx<T> U = Uh[(size_t)q * upitch + r];
  cpx<T> P = {T(0), T(0)};
  if (q != 0 || r != 0) {
    const cpx<T> V = cmul(D0[r], e0[q]) + cmul(D1[q], e1[r]);
    const T inv = T(-0.25) / (T)(s0[q] + s1[r]);
    P = {U.x - V.x * inv, U.y - V.y * inv};
  } else if (!ABS) {
    P = U;
  }
  if constexpr (ABS) {
    // (f32: the bins of a 4096^2 image stay below 1e8, far from where x^2 + y^2 leaves the f32 range)
    if constexpr (sizeof(T) == 4) static_cast<T*>(out_)[(size_t)oi * n1 + oj] = sqrtf(P.x * P.x + P.y * P.y);
    else static_cast<T*>(out_)[(size_t)oi * n1 + oj] = hypot(P.x, P.y);
  }
  else static_cast<cpx<T>*>(out_)[(size_t)oi * n1 + oj] = P;
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
template <class T, int LG, bool REAL_IN, bool HALF = false>
hipError_t run_p2_rows(const DftAxis& a, int nrows, const void* in, void* out, hipStream_t s, size_t opitch = 0) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = p2_rows_kernel<T, LG, REAL_IN, HALF>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int ntr = REAL_IN ? (nrows + 1) / 2 : nrows;
    GPA_PROF(REAL_IN ? "dft_rows_r2c_kernel" : "dft_rows_kernel", s);
    kern<<<(ntr + G::NF - 1) / G::NF, G::THREADS, G::LDS_BYTES, s>>>(in, (cpx<T>*)out, nrows, (const cpx<T>*)a.tw,
                                                                     opitch ? opitch : (size_t)a.n);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_p2_cols(const DftAxis& a, int n1, size_t pitch, void* Z, hipStream_t s) {
  using G = ColGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = p2_cols_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    GPA_PROF("dft_cols_kernel", s);
    kern<<<(n1 + G::C - 1) / G::C, G::THREADS, G::LDS_BYTES, s>>>((cpx<T>*)Z, n1, pitch, (const cpx<T>*)a.tw);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_rowdft(const DftAxis& a, int n0, void* Z, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_rowdft_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    GPA_PROF("dft_rows_chirpz_kernel", s);
    kern<<<(n0 + G::NF - 1) / G::NF, G::THREADS, G::LDS_BYTES, s>>>((cpx<T>*)Z, n0, a.n, (const cpx<T>*)a.tw,
                                                                     (const cpx<T>*)a.chirp, (const cpx<T>*)a.bspec);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_coldft(const DftAxis& a, int n1, size_t pitch, void* Z, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_coldft_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    GPA_PROF("dft_cols_chirpz_kernel", s);
    kern<<<(n1 + G::NF - 1) / G::NF, G::THREADS, G::LDS_BYTES, s>>>((cpx<T>*)Z, a.n, n1, pitch, (const cpx<T>*)a.tw,
                                                                     (const cpx<T>*)a.chirp, (const cpx<T>*)a.bspec);
    return hipGetLastError();
  }
}

#define GPA_FOR_SUBLG(X) X(6) X(7) X(8) X(9)

#define GPA_BIG_LAUNCH(KERN, LG, GRID, ...)                                                                         \
  case LG: {                                                                                                        \
    auto kern = KERN<T, LG, COLS>;                                                                                  \
    static unsigned lds_set = 0;                                                                                    \
    if ((e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)TileGeom<T, LG>::LDS_BYTES, lds_set)) != hipSuccess) return e; \
    kern<<<GRID, 256, TileGeom<T, LG>::LDS_BYTES, s>>>(__VA_ARGS__);                                                \
  } break;

template <class T, bool COLS>
hipError_t big_launch(const DftAxis& a, cpx<T>* Z, size_t zpitch, int nlines, cpx<T>* W, size_t wpitch, hipStream_t s) {
  const int M1 = 1 << a.lg1, M2 = 1 << a.lg2;
  const int nf1 = 4096 >> a.lg1, nf2 = 4096 >> a.lg2;   // TileGeom::NF of the two levels
  const dim3 gridA = COLS ? dim3((nlines + nf1 - 1) / nf1, M2) : dim3(M2 / nf1, nlines);
  const dim3 gridM = COLS ? dim3((nlines + nf2 - 1) / nf2, M1) : dim3((unsigned)((((size_t)nlines << a.lg1) + nf2 - 1) / nf2), 1);
  const cpx<T>*chirp = (const cpx<T>*)a.chirp, *twL = (const cpx<T>*)a.tw, *tw1 = (const cpx<T>*)a.tw1, *tw2 = (const cpx<T>*)a.tw2;
  hipError_t e;
  {
    GPA_PROF("dft_big_fwd_kernel", s);
#define CASE(LG) GPA_BIG_LAUNCH(big_fwdA_kernel, LG, gridA, Z, zpitch, a.n, nlines, W, wpitch, a.lg2, chirp, twL, tw1)
    switch (a.lg1) { GPA_FOR_SUBLG(CASE) default: return hipErrorInvalidValue; }
#undef CASE
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  {
    GPA_PROF("dft_big_mid_kernel", s);
#define CASE(LG) GPA_BIG_LAUNCH(big_mid_kernel, LG, gridM, W, wpitch, nlines, a.lg1, (const cpx<T>*)a.bspec, tw2)
    switch (a.lg2) { GPA_FOR_SUBLG(CASE) default: return hipErrorInvalidValue; }
#undef CASE
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  {
    GPA_PROF("dft_big_inv_kernel", s);
#define CASE(LG) GPA_BIG_LAUNCH(big_invA_kernel, LG, gridA, Z, zpitch, a.n, nlines, W, wpitch, a.lg2, chirp, twL, tw1)
    switch (a.lg1) { GPA_FOR_SUBLG(CASE) default: return hipErrorInvalidValue; }
#undef CASE
  }
  return hipGetLastError();
}

hipError_t work_reserve(DftWork* w, size_t bytes, hipStream_t s) {
  if (!w) return hipErrorInvalidValue;
  if (w->cap >= bytes) return hipSuccess;
  hipError_t e;
  if (w->buf) {
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    (void)hipFree(w->buf);
    if (w->counted) *w->counted -= w->cap;
    w->buf = nullptr;
    w->cap = 0;
  }
  if ((e = hipMalloc(&w->buf, bytes)) != hipSuccess) return e;
  w->cap = bytes;
  if (w->counted) *w->counted += bytes;
  return hipSuccess;
}

// lines of length a.n: COLS -> `nlines` columns of a row-major array with row pitch `pitch`; rows -> `nlines` rows
template <class T>
hipError_t big_run(const DftAxis& a, bool cols, cpx<T>* Z, size_t pitch, int nlines, DftWork* w, hipStream_t s) {
  const size_t L = (size_t)1 << a.lg;
  // scratch of at most ~1 GiB: chunks of lines (columns: multiples of 64 so that the chunk's rows stay 512-byte aligned)
  size_t chunk = ((size_t)1 << 30) / (L * sizeof(cpx<T>));
  if (cols) chunk &= ~(size_t)63;
  if (chunk < (cols ? 64 : 1)) chunk = cols ? 64 : 1;
  if (chunk > (size_t)nlines) chunk = nlines;
  if (!cols && chunk > 32768) chunk = 32768;   // (grid.y)
  hipError_t e = work_reserve(w, chunk * L * sizeof(cpx<T>), s);
  if (e != hipSuccess) return e;
  cpx<T>* W = (cpx<T>*)w->buf;
  for (size_t l0 = 0; l0 < (size_t)nlines; l0 += chunk) {
    const int nl = (int)(l0 + chunk <= (size_t)nlines ? chunk : nlines - l0);
    if (cols) e = big_launch<T, true>(a, Z + l0, pitch, nl, W, (size_t)nl, s);
    else e = big_launch<T, false>(a, Z + l0 * pitch, pitch, nl, W, L, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t rows_inplace(int dtype, const DftAxis& a, int rows, void* Z, DftWork* w, hipStream_t s) {
  hipError_t e = hipErrorInvalidValue;
  switch (a.kind) {
    case DFT_POW2:
#define CASE(LG) case LG: e = dtype == 0 ? run_p2_rows<float, LG, false>(a, rows, Z, Z, s) : run_p2_rows<double, LG, false>(a, rows, Z, Z, s); break;
      switch (a.lg) { GPA_FOR_LG(CASE) }
#undef CASE
      break;
    case DFT_BLUE:
#define CASE(LG) case LG: e = dtype == 0 ? run_rowdft<float, LG>(a, rows, Z, s) : run_rowdft<double, LG>(a, rows, Z, s); break;
      switch (a.lg) { GPA_FOR_LG(CASE) }
#undef CASE
      break;
    case DFT_BIG:
      e = dtype == 0 ? big_run<float>(a, false, (cpx<float>*)Z, (size_t)a.n, rows, w, s)
                     : big_run<double>(a, false, (cpx<double>*)Z, (size_t)a.n, rows, w, s);
      break;
  }
  return e;
}

// `n1` columns of a row-major array with row pitch `pitch` (complex elements)
hipError_t cols_inplace(int dtype, const DftAxis& a, int n1, size_t pitch, void* Z, DftWork* w, hipStream_t s) {
  hipError_t e = hipErrorInvalidValue;
  switch (a.kind) {
    case DFT_POW2:
#define CASE(LG) case LG: e = dtype == 0 ? run_p2_cols<float, LG>(a, n1, pitch, Z, s) : run_p2_cols<double, LG>(a, n1, pitch, Z, s); break;
      switch (a.lg) { GPA_FOR_LG(CASE) }
#undef CASE
      break;
    case DFT_BLUE:
#define CASE(LG) case LG: e = dtype == 0 ? run_coldft<float, LG>(a, n1, pitch, Z, s) : run_coldft<double, LG>(a, n1, pitch, Z, s); break;
      switch (a.lg) { GPA_FOR_LG(CASE) }
#undef CASE
      break;
    case DFT_BIG:
      e = dtype == 0 ? big_run<float>(a, true, (cpx<float>*)Z, pitch, n1, w, s)
                     : big_run<double>(a, true, (cpx<double>*)Z, pitch, n1, w, s);
      break;
  }
  return e;
}

// ---- host tables ----------------------------------------------------------------------------------------------------
void host_fft_inplace(std::vector<double>& re, std::vector<double>& im) {
  const size_t n = re.size();
  int lg = 0;
  while ((size_t(1) << lg) < n) ++lg;
  std::vector<double> wr(n / 2), wi(n / 2);
  for (size_t j = 0; j < n / 2; ++j) { wr[j] = cos(-2.0 * M_PI * (double)j / (double)n); wi[j] = sin(-2.0 * M_PI * (double)j / (double)n); }
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
    if (r > i) { std::swap(re[i], re[r]); std::swap(im[i], im[r]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t step = n / len;
    for (size_t s0 = 0; s0 < n; s0 += len)
      for (size_t j = 0; j < len / 2; ++j) {
        const double cr = wr[j * step], ci = wi[j * step];
        const size_t a = s0 + j, b = s0 + j + len / 2;
        const double vr = re[b] * cr - im[b] * ci, vi = re[b] * ci + im[b] * cr;
        re[b] = re[a] - vr; im[b] = im[a] - vi;
        re[a] += vr; im[a] += vi;
      }
  }
}

template <class T>
hipError_t upload_vec(void** dst, const std::vector<double>& v, size_t* bytes, hipStream_t s) {
  std::vector<T> tmp(v.begin(), v.end());
  hipError_t e = hipMalloc(dst, tmp.size() * sizeof(T) + 16);
  if (e != hipSuccess) return e;
  *bytes += tmp.size() * sizeof(T);
  e = hipMemcpyAsync(*dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}
hipError_t upload(int dtype, void** dst, const std::vector<double>& v, size_t* bytes, hipStream_t s) {
  return dtype == 0 ? upload_vec<float>(dst, v, bytes, s) : upload_vec<double>(dst, v, bytes, s);
}
std::vector<double> twiddle_table(int L) {
  std::vector<double> t((size_t)2 * L);
  for (int k = 0; k < L; ++k) { t[2 * (size_t)k] = cos(-2.0 * M_PI * k / L); t[2 * (size_t)k + 1] = sin(-2.0 * M_PI * k / L); }
  return t;
}
// one workgroup holds a row of 2^lg points: rows (GenGeom) and columns (ColGeom) agree on the limit
bool fits_workgroup(int dtype, int lg) { return lg >= 6 && lg <= (dtype == 0 ? 14 : 13); }

}  // namespace

const char* dft_kind_name(int kind) {
  return kind == DFT_POW2 ? "pow2" : kind == DFT_BLUE ? "chirpz" : kind == DFT_BIG ? "chirpz2" : "none";
}

hipError_t dft_axis_create(int dtype, int n, hipStream_t s, DftAxis* out, size_t* bytes, int force) {
  if (n < 1 || n > 65536) return hipErrorInvalidValue;
  int lgn = 0;
  while ((1 << lgn) < n) ++lgn;
  int lgb = 6;
  while ((1 << lgb) < 2 * n - 1) ++lgb;
  int kind;
  if (force == DFT_BIG) kind = DFT_BIG;
  else if (force == DFT_BLUE) kind = fits_workgroup(dtype, lgb) ? DFT_BLUE : DFT_BIG;
  else if ((1 << lgn) == n && fits_workgroup(dtype, lgn)) kind = DFT_POW2;
  else kind = fits_workgroup(dtype, lgb) ? DFT_BLUE : DFT_BIG;
  *out = DftAxis{};
  out->n = n;
  out->kind = kind;
  size_t b = 0;
  hipError_t e;
  if (kind == DFT_POW2) {
    out->lg = lgn;
    if ((e = upload(dtype, &out->tw, twiddle_table(1 << lgn), &b, s)) != hipSuccess) return e;
    if (bytes) *bytes += b;
    return hipSuccess;
  }
  if (kind == DFT_BIG && lgb < 12) lgb = 12;
  if (lgb > 18) return hipErrorInvalidValue;
  const int L = 1 << lgb;
  out->lg = lgb;
  std::vector<double> ch((size_t)2 * n), bre((size_t)L, 0.0), bim((size_t)L, 0.0);
  for (int m = 0; m < n; ++m) {
    const long long mm = ((long long)m * m) % (2LL * n);
    const double cr = cos(M_PI * (double)mm / n), ci = sin(M_PI * (double)mm / n);
    ch[2 * (size_t)m] = cr; ch[2 * (size_t)m + 1] = ci;
    bre[m] = cr; bim[m] = ci;
    if (m > 0) { bre[L - m] = cr; bim[L - m] = ci; }
  }
  host_fft_inplace(bre, bim);
  std::vector<double> bs((size_t)2 * L);
  if (kind == DFT_BLUE) {
    const int tpf = L / 16;
    for (int i = 0; i < 16; ++i)
      for (int tt = 0; tt < tpf; ++tt) {
        const int k = spec_index_rt(lgb, tt, i);
        bs[2 * ((size_t)i * tpf + tt)] = bre[k] / L;
        bs[2 * ((size_t)i * tpf + tt) + 1] = bim[k] / L;
      }
  } else {
    const int lg1 = (lgb + 1) / 2, lg2 = lgb / 2, M1 = 1 << lg1, M2 = 1 << lg2, tpf1 = M1 / 16, tpf2 = M2 / 16;
    out->lg1 = lg1;
    out->lg2 = lg2;
    for (int i1 = 0; i1 < 16; ++i1)
      for (int t1 = 0; t1 < tpf1; ++t1) {
        const int slot = t1 + tpf1 * i1, k1 = spec_index_rt(lg1, t1, i1);
        for (int i2 = 0; i2 < 16; ++i2)
          for (int t2 = 0; t2 < tpf2; ++t2) {
            const size_t k = (size_t)k1 + (size_t)M1 * spec_index_rt(lg2, t2, i2);
            const size_t o = (size_t)slot * M2 + (size_t)i2 * tpf2 + t2;
            bs[2 * o] = bre[k] / L;
            bs[2 * o + 1] = bim[k] / L;
          }
      }
    if ((e = upload(dtype, &out->tw1, twiddle_table(M1), &b, s)) != hipSuccess) return e;
    if ((e = upload(dtype, &out->tw2, twiddle_table(M2), &b, s)) != hipSuccess) return e;
  }
  if ((e = upload(dtype, &out->tw, twiddle_table(L), &b, s)) != hipSuccess) return e;
  if ((e = upload(dtype, &out->chirp, ch, &b, s)) != hipSuccess) return e;
  if ((e = upload(dtype, &out->bspec, bs, &b, s)) != hipSuccess) return e;
  if (bytes) *bytes += b;
  return hipSuccess;
}

void dft_axis_destroy(DftAxis* a) {
  for (void* p : {a->tw, a->chirp, a->bspec, a->tw1, a->tw2})
    if (p) (void)hipFree(p);
  *a = DftAxis{};
}

void dft_work_free(DftWork* w) {
  if (w->buf) {
    (void)hipFree(w->buf);
    if (w->counted) *w->counted -= w->cap;
  }
  w->buf = nullptr;
  w->cap = 0;
}

hipError_t dft2_inplace(int dtype, const DftAxis& a0, const DftAxis& a1, void* Z, DftWork* w, hipStream_t s) {
  hipError_t e = rows_inplace(dtype, a1, a0.n, Z, w, s);
  if (e != hipSuccess) return e;
  return cols_inplace(dtype, a0, a1.n, (size_t)a1.n, Z, w, s);
}

hipError_t dft_rows_inplace(int dtype, const DftAxis& a, int rows, void* Z, DftWork* w, hipStream_t s) {
  return rows_inplace(dtype, a, rows, Z, w, s);
}

hipError_t dft2_forward_real(int dtype, const DftAxis& a0, const DftAxis& a1, const void* image, void* Z, DftWork* w,
                             hipStream_t s) {
  hipError_t e = hipErrorInvalidValue;
  if (a1.kind == DFT_POW2) {
#define CASE(LG) case LG: e = dtype == 0 ? run_p2_rows<float, LG, true>(a1, a0.n, image, Z, s) : run_p2_rows<double, LG, true>(a1, a0.n, image, Z, s); break;
    switch (a1.lg) { GPA_FOR_LG(CASE) }
#undef CASE
  } else {
    const size_t count = (size_t)a0.n * a1.n;
    const unsigned grid = (unsigned)((count + 255) / 256);
    {
      GPA_PROF("dft_pack_kernel", s);
      if (dtype == 0) real_to_complex_kernel<float><<<grid, 256, 0, s>>>((const float*)image, (cpx<float>*)Z, count);
      else real_to_complex_kernel<double><<<grid, 256, 0, s>>>((const double*)image, (cpx<double>*)Z, count);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = rows_inplace(dtype, a1, a0.n, Z, w, s);
  }
  if (e != hipSuccess) return e;
  return cols_inplace(dtype, a0, a1.n, (size_t)a1.n, Z, w, s);
}

// bins [0, n1/2] of every row only (row pitch dft2_half_pitch): half the column transforms and half the traffic
size_t dft2_half_pitch(const DftAxis& a1) { return a1.kind == DFT_POW2 ? (size_t)(a1.n / 2 + 2) : 0; }
hipError_t dft2_forward_real_half(int dtype, const DftAxis& a0, const DftAxis& a1, const void* image, void* Z, DftWork* w,
                                  hipStream_t s) {
  const size_t hp = dft2_half_pitch(a1);
  if (!hp) return hipErrorInvalidValue;
  hipError_t e = hipErrorInvalidValue;
#define CASE(LG) case LG: e = dtype == 0 ? run_p2_rows<float, LG, true, true>(a1, a0.n, image, Z, s, hp) : run_p2_rows<double, LG, true, true>(a1, a0.n, image, Z, s, hp); break;
  switch (a1.lg) { GPA_FOR_LG(CASE) }
#undef CASE
  if (e != hipSuccess) return e;
  return cols_inplace(dtype, a0, a1.n / 2 + 1, hp, Z, w, s);
}

hipError_t per_borders(int dtype, const void* image, int n0, int n1, void* d0, void* d1, hipStream_t s) {
  const int len = n0 > n1 ? n0 : n1;
  GPA_PROF("per_borders_kernel", s);
  if (dtype == 0)
    per_borders_kernel<float><<<(len + 255) / 256, 256, 0, s>>>((const float*)image, n0, n1, (cpx<float>*)d0, (cpx<float>*)d1);
  else
    per_borders_kernel<double><<<(len + 255) / 256, 256, 0, s>>>((const double*)image, n0, n1, (cpx<double>*)d0, (cpx<double>*)d1);
  return hipGetLastError();
}
// the rest of moisan2011.per: s_hat = u_hat - p_hat, and the two components in real space through ONE more DFT:
// p = Re(ifft2(p_hat)) = Re(fft2(conj(p_hat))) / (n0 n1), s = u - p
namespace {
template <class T>
__global__ __launch_bounds__(256) void per_diff_kernel(cpx<T>* __restrict__ Uh, const cpx<T>* __restrict__ Ph, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) Uh[i] = {Uh[i].x - Ph[i].x, Uh[i].y - Ph[i].y};
}
template <class T>
__global__ __launch_bounds__(256) void per_conj_kernel(cpx<T>* __restrict__ Z, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) Z[i].y = -Z[i].y;
}
template <class T>
__global__ __launch_bounds__(256) void per_real_kernel(const cpx<T>* __restrict__ Z, const T* __restrict__ u, double scale,
                                                       T* __restrict__ pout, T* __restrict__ sout, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const T pv = (T)((double)Z[i].x * scale);
    pout[i] = pv;
    sout[i] = u[i] - pv;
  }
}
}  // namespace
hipError_t per_smooth_hat(int dtype, void* Uhat_inout, const void* Phat, size_t n, hipStream_t s) {
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == 0) per_diff_kernel<float><<<grid, 256, 0, s>>>((cpx<float>*)Uhat_inout, (const cpx<float>*)Phat, n);
  else per_diff_kernel<double><<<grid, 256, 0, s>>>((cpx<double>*)Uhat_inout, (const cpx<double>*)Phat, n);
  return hipGetLastError();
}
hipError_t per_components(int dtype, const DftAxis& a0, const DftAxis& a1, void* Phat_destroyed, const void* image,
                          void* p_out, void* s_out, DftWork* w, hipStream_t s) {
  const size_t n = (size_t)a0.n * a1.n;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == 0) per_conj_kernel<float><<<grid, 256, 0, s>>>((cpx<float>*)Phat_destroyed, n);
  else per_conj_kernel<double><<<grid, 256, 0, s>>>((cpx<double>*)Phat_destroyed, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = dft2_inplace(dtype, a0, a1, Phat_destroyed, w, s);
  if (e != hipSuccess) return e;
  const double scale = 1.0 / (double)n;
  if (dtype == 0)
    per_real_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)Phat_destroyed, (const float*)image, scale, (float*)p_out, (float*)s_out, n);
  else
    per_real_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)Phat_destroyed, (const double*)image, scale, (double*)p_out, (double*)s_out, n);
  return hipGetLastError();
}
size_t per_tables_bytes(int dtype, int n0, int n1) { return (size_t)(n0 + n1) * (sizeof(double) + (dtype == 0 ? 8 : 16)); }
hipError_t per_tables_fill(int dtype, int n0, int n1, void* tab, hipStream_t s) {
  double* s0 = static_cast<double*>(tab);
  double* s1 = s0 + n0;
  void* e0 = s1 + n1;
  const int len = n0 > n1 ? n0 : n1;
  if (dtype == 0) per_tables_kernel<float><<<(len + 255) / 256, 256, 0, s>>>(n0, n1, s0, s1, (cpx<float>*)e0, (cpx<float>*)e0 + n0);
  else per_tables_kernel<double><<<(len + 255) / 256, 256, 0, s>>>(n0, n1, s0, s1, (cpx<double>*)e0, (cpx<double>*)e0 + n0);
  return hipGetLastError();
}
hipError_t per_combine(int dtype, const void* Uhat, const void* D0, const void* D1, int n0, int n1, const void* tab,
                       bool abs_shift, void* out, hipStream_t s, size_t half_pitch) {
  if (half_pitch && !abs_shift) return hipErrorInvalidValue;
  dim3 grid((n1 + 255) / 256, n0);
  const double* s0 = static_cast<const double*>(tab);
  const double* s1 = s0 + n0;
  const void* e0 = s1 + n1;
  GPA_PROF(abs_shift ? "per_absshift_kernel" : "per_combine_kernel", s);
#define CALL(T, ABS) per_combine_kernel<T, ABS><<<grid, 256, 0, s>>>((const cpx<T>*)Uhat, (const cpx<T>*)D0, (const cpx<T>*)D1, n0, n1, \
                                                                      s0, s1, (const cpx<T>*)e0, (const cpx<T>*)e0 + n0, out, \
                                                                      half_pitch ? half_pitch : (size_t)n1, half_pitch ? 1 : 0)
  if (dtype == 0) { if (abs_shift) CALL(float, true); else CALL(float, false); }
  else { if (abs_shift) CALL(double, true); else CALL(double, false); }
#undef CALL
  return hipGetLastError();
}

}  // namespace gpa
