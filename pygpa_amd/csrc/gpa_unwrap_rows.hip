// a7, row kernels of the fused PCG iteration on power-of-two rows (phase_unwrap.py:326-349; preconditioner :95-115):
//   rowdct_fused : R -= alpha DCT-II_rows(q)   the residual is kept as its row spectrum R; ||r||^2 by Parseval
//   rowidct_p    : Z -> p = z + beta p_prev    row DCT-III straight into the new search direction
//   rowidct_pq   : the same and the stencil in one launch (one image, rows up to 512 pixels)
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

#ifndef GPA_ROW_TWLDS
#define GPA_ROW_TWLDS 1   // 16-element three-pass row transforms: pass-1 base twiddles from a small LDS table (12 VGPRs less in f32)
#endif
template <class T, int LG, bool LAT = false>
struct RowGeom {
  using F = WgFFT<T, LG, unwrap_elems(LG, sizeof(T))>;
  static constexpr bool TWLDS = GPA_ROW_TWLDS && F::E == 16 && F::P == 3;
  using TW = typename std::conditional<TWLDS, typename F::TwiddlesP1Lds, typename F::Twiddles>::type;
  static constexpr int T1N = TWLDS ? F::P1_SETS * 6 : 1;
  using D = WgDCT<T, LG, unwrap_elems(LG, sizeof(T))>;
  // threads per workgroup: 256; the latency-tuned kernels of ONE image with rows up to 512 pixels take 128 (twice the
  // workgroups on a GPU that such an image leaves mostly empty: 512^2 893 -> 935 Mpix/s; stacks prefer 256)
#ifndef GPA_ROW_THREADS_LAT
#define GPA_ROW_THREADS_LAT 128
#endif
#ifndef GPA_ROW_THREADS
#define GPA_ROW_THREADS 256
#endif
  // (rows up to 256 pixels: one wavefront per workgroup, 256^2 280 -> 290 Mpix/s; at 512 that loses 8 %)
  static constexpr int WGT = (LAT && LG <= 8) ? 64 : (LAT && LG == 9) ? GPA_ROW_THREADS_LAT : GPA_ROW_THREADS;
  static constexpr int NF = F::TPF >= WGT ? 1 : WGT / F::TPF;   // row PAIRS per workgroup
  static constexpr int RS = F::LDS_ELEMS + (NF > 1 ? (F::TPF < 32 ? F::TPF : 0) : 0);
  static constexpr int THREADS = NF * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NF * RS * sizeof(cpx<T>);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024;
};
// fused path: apply the pending update of the previous iteration (alpha from the pq
// kernel's partial sums), then DCT-II along axis 1 of the new residual
//   r -= alpha q;  phi += alpha p;  partial ||r||^2;  Z = DCT(r)
#ifndef GPA_DCTF_WAVES
#define GPA_DCTF_WAVES 1
#endif
#ifndef GPA_EARLY16
#define GPA_EARLY16 0   // experiment: the every-input-first kernel variants also for 16-element transforms
#endif
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64 row kernels: 2 waves/SIMD (256 VGPRs) beat 1 wave with AGPR spill-over
#endif
// LAT: the latency-tuned variant (one image per call, axes up to 1024) -- same arithmetic, same results
template <class T, int LG, bool LAT = false>
__global__ __launch_bounds__((RowGeom<T, LG, LAT>::THREADS), (sizeof(T) == 8 ? GPA_F64_WAVES : GPA_DCTF_WAVES)) void rowdct_fused_kernel(
    T* __restrict__ r, const T* __restrict__ q, int n0, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ wk, int* flags, const double* part_pq, int npq, double* part_norm,
    double* scal, int it, int ring, int init, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    r += pb * pimg;
    q += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_pq += pb * PART_N;
    part_norm += pb * PART_N;
  }
  // The fused iteration keeps the residual as its row spectrum R = DCT-II_rows(r) (the only consumers of r
  // are this transform, ||r|| and <r,z>, and the last two follow from the spectra by Parseval):
  //   it == 0: r (spatial, from the set-up) -> R, in place;
  //   it  > 0: R -= alpha DCT-II_rows(q)    (linearity; phase_unwrap.py:345), partial ||r||^2 from R.
  // So the update reads q and R and writes R: three arrays instead of r, q in and r, Z out.
  using G = RowGeom<T, LG, LAT>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = F::TPF, N = F::L, E = F::E;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowGeom<T, LG, LAT>::THREADS];
  // init (first iteration of a solve on prepared residuals): part_pq / npq are the producer's partial norms of r0
  // EARLY (short transforms): every input of an update -- flags, q, the kept spectrum, w_k, partial sums, rho -- is
  // requested before anything waits, so the kernel pays one memory round trip instead of five in a row
  constexpr bool EARLY = LAT && (E == 8 || GPA_EARLY16);
  const bool early = EARLY && it > 0;
  int stop = 0;
  if (init) { if (!solve_init(part_pq, npq, scal, flags, sh)) return; }
  else if (early) stop = flags[1];
  else if (flags[1]) return;
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const bool valid = 2 * pr + 1 < n0;
  const size_t oa = (size_t)(valid ? 2 * pr : 0) * N, ob = oa + N;
  // (register twiddles by default: the LDS table of rowidct_p_kernel made this kernel's allocation worse, 156 -> 160 VGPRs)
#ifndef GPA_DCTF_TWLDS
#define GPA_DCTF_TWLDS 0
#endif
  constexpr bool TWL = GPA_DCTF_TWLDS && G::TWLDS;
  typename std::conditional<TWL, typename F::TwiddlesP1Lds, typename F::Twiddles>::type tw;
  __shared__ cpx<T> t1s[TWL ? G::T1N : 1];
  if constexpr (TWL) {
    F::fill_pass1_table(t1s, twtab, threadIdx.x, G::THREADS);
    __syncthreads();
    F::load_twiddles(tw, twtab, tid, t1s);
  } else {
    F::load_twiddles(tw, twtab, tid);
  }
  cpx<T> x[E];
  cpx<T> rk[E];
  cpx<T> wkv[EARLY ? E : 1];
  T alpha = T(0);
  if (early) {
    // up to 1024 points the even/odd-permuted DCT input is fetched directly (stride-2 accesses: these sizes are
    // latency-, not bandwidth-bound, and the detour through LDS costs two barriers); 2048 points: 16-byte loads
    constexpr bool DIRECTQ = LG <= 10;
    constexpr int NQ = N / (4 * TPF);   // 16-byte vectors of q per thread and row
    Vec4<T> qa[NQ], qb[NQ];
    if constexpr (DIRECTQ) {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int src = makhoul_src(tid + TPF * i, N);
        x[i] = {q[oa + src], q[ob + src]};
      }
    } else {
#pragma unroll
      for (int v = 0; v < NQ; ++v) {
        const int c0 = 4 * (tid + TPF * v);
        qa[v] = *reinterpret_cast<const Vec4<T>*>(q + oa + c0);
        qb[v] = *reinterpret_cast<const Vec4<T>*>(q + ob + c0);
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      rk[i] = {r[oa + tid + TPF * i], r[ob + tid + TPF * i]};
      wkv[EARLY ? i : 0] = wk[tid + TPF * i];
    }
    const double pq_part = load_partials(part_pq, npq);
    const double rho = scal[8 + ((it - 1) & 1)];
    if (stop) return;
    const double pq = block_sum(pq_part, sh);
    const double alpha_d = rho / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
    if constexpr (!DIRECTQ) {
#pragma unroll
      for (int v = 0; v < NQ; ++v) {
        const int c0 = 4 * (tid + TPF * v);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[F::pad(c0 + j)] = {qa[v].v[j], qb[v].v[j]};
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < E; ++i) x[i] = lds[F::pad(makhoul_src(tid + TPF * i, N))];
      __syncthreads();
    }
  } else if (it > 0) {
    const double pq = reduce_partials(part_pq, npq, sh);
    const double alpha_d = scal[8 + ((it - 1) & 1)] / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    // phi += alpha p is not applied here: alpha is filed for phi_flush_kernel, which adds the kept
    // search directions of up to `ring` iterations in one pass over phi
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
    // q comes in with coalesced 16-byte accesses and is parked in LDS, so that the even/odd-permuted
    // DCT input does not have to be fetched with stride-2 accesses
    for (int c0 = 4 * tid; c0 < N; c0 += 4 * TPF) {
      const Vec4<T> qa = *reinterpret_cast<const Vec4<T>*>(q + oa + c0), qb = *reinterpret_cast<const Vec4<T>*>(q + ob + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[F::pad(c0 + j)] = {qa.v[j], qb.v[j]};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = lds[F::pad(makhoul_src(tid + TPF * i, N))];
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int src = makhoul_src(tid + TPF * i, N);
      x[i] = {r[oa + src], r[ob + src]};
    }
    __syncthreads();   // in place: every sample of the two rows is in registers before any bin is written
  }
#ifndef GPA_DCTF_LATE_RK
#define GPA_DCTF_LATE_RK 0
#endif
  // the kept spectrum is requested before the transform so that its latency hides behind it
  // (GPA_DCTF_LATE_RK: after it instead -- 32 registers less across the transform, one more wave per SIMD)
  if (it > 0 && !early && !GPA_DCTF_LATE_RK) {
#pragma unroll
    for (int i = 0; i < E; ++i) rk[i] = {r[oa + tid + TPF * i], r[ob + tid + TPF * i]};
  }
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::fwd_scatter(x, lds, tid);
  __syncthreads();
  if constexpr (EARLY) { if (early) D::fwd_gather(x, lds, tid, wkv); else D::fwd_gather(x, lds, tid, wk); }
  else D::fwd_gather(x, lds, tid, wk);
  if (it > 0 && GPA_DCTF_LATE_RK) {
#pragma unroll
    for (int i = 0; i < E; ++i) rk[i] = {r[oa + tid + TPF * i], r[ob + tid + TPF * i]};
  }
  double sq = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    T ra = x[i].x, rb = x[i].y;
    if (it > 0) {
      ra = rk[i].x - alpha * ra;
      rb = rk[i].y - alpha * rb;
      // sum_n r^2 = (1 / 2N) sum_k c_k R_k^2, c_0 = 1/2 (SciPy's unnormalised DCT-II)
      const double t = (double)ra * (double)ra + (double)rb * (double)rb;
      sq += k == 0 ? 0.5 * t : t;
    }
    if (valid) {
      r[oa + k] = ra;
      r[ob + k] = rb;
    }
  }
  if (it > 0) {
    if (!valid) sq = 0;
    const double tot = block_sum(sq, sh);
    if (threadIdx.x == 0) part_norm[blockIdx.x] = tot / (2.0 * N);
  }
}
// fused path: rows Z -> z = DCT-III along axis 1, and straight on to the new search direction
// p = z + beta p_prev (phase_unwrap.py:336-340) -- z itself never goes to HBM.  beta = rho / rho_prev with
// rho from the column kernel's Parseval partial sums.
// f32, 4096-point rows: 4 waves per SIMD (<= 128 VGPRs; the unconstrained allocation takes 130 and runs at 3):
// 49 -> 43 us.  Other lengths would spill under that cap (2048: 13 -> 16 us) and keep the default.
template <class T, int LG, bool LAT = false>
#ifndef GPA_IDCTP_COND
#define GPA_IDCTP_COND (sizeof(T) == 4 && (LG == 12 || LG == 13))   // (8192 points: 28 B of scratch buy a second workgroup per CU, -12 %)
#endif
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64 row kernels: 2 waves/SIMD (256 VGPRs) beat 1 wave with AGPR spill-over
#endif
__global__ __launch_bounds__((RowGeom<T, LG, LAT>::THREADS), ((GPA_IDCTP_COND && !LAT) ? 4 : (sizeof(T) == 8 ? GPA_F64_WAVES : 1))) void rowidct_p_kernel(
    const T* __restrict__ Z, const T* __restrict__ pin, T* __restrict__ pout, int n0,
    const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ wk, const int* flags, const double* part_rho,
    int nrho, double* scal, int it, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    pin += pb * pimg;
    pout += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_rho += pb * PART_N;
  }
  // Every input of the kernel is requested before anything waits (one memory round trip for the flags, the
  // spectrum, the previous search direction, the tables and the partial sums together -- a 512-point kernel is
  // little more than its chain of dependent round trips), then the early exit, then the arithmetic.
  const int stop = flags[1];
  using G = RowGeom<T, LG, LAT>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = F::TPF, N = F::L, E = F::E;
  // (short transforms only: the long ones are bandwidth-bound, hide latency behind other workgroups and have no
  //  registers to spare for 2 E more values)
  constexpr bool EARLY = LAT && (E == 8 || GPA_EARLY16);
  if (!EARLY && stop) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowGeom<T, LG, LAT>::THREADS];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const bool valid = 2 * pr + 1 < n0;
  const size_t oa = (size_t)(valid ? 2 * pr : 0) * N, ob = oa + N;
  typename G::TW tw;
  __shared__ cpx<T> t1s[G::T1N];
  if constexpr (G::TWLDS) {
    F::fill_pass1_table(t1s, twtab, threadIdx.x, G::THREADS);
    __syncthreads();
    F::load_twiddles(tw, twtab, tid, t1s);
  } else {
    F::load_twiddles(tw, twtab, tid);
  }
  cpx<T> x[E], xm[E], wkv[EARLY ? E : 1], pv[EARLY ? E : 1];
  const bool first = it == 0;                        // first iteration: p = z (pin is uninitialised)
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    x[i] = {Z[oa + k], Z[ob + k]};
    xm[i] = k == 0 ? cpx<T>{T(0), T(0)} : cpx<T>{Z[oa + N - k], Z[ob + N - k]};
    if constexpr (EARLY) {
      wkv[i] = wk[k];
      pv[i] = first ? cpx<T>{T(0), T(0)} : cpx<T>{pin[oa + k], pin[ob + k]};
    }
  }
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  if constexpr (EARLY) D::inv_prepare(x, xm, wkv); else D::inv_prepare(x, xm, tid, wk);
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::inv_scatter(x, lds, tid, T(1) / T(N));
  __syncthreads();
  D::inv_gather(x, lds, tid);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int c = tid + TPF * i;
    T pa = x[i].x, pb = x[i].y;
    if (!first) {
      if constexpr (EARLY) {
        pa += beta * pv[i].x;
        pb += beta * pv[i].y;
      } else {
        pa += beta * pin[oa + c];
        pb += beta * pin[ob + c];
      }
    }
    pout[oa + c] = pa;
    pout[ob + c] = pb;
  }
}

// One image, rows of at most 512 pixels: rowidct_p_kernel and the stencil kernel in ONE launch.  At these sizes a
// launch costs more than the work of either (an empty kernel: 3.6 us; the stencil kernel: 4.0), so the row kernel
// also transforms the row pair above and the one below its own NF pairs, keeps all 2 NF + 4 rows of the new search
// direction in LDS and applies q = A^T W^2 A p to its own rows there.  (NF + 2) / NF of the transforms instead of one
// more launch per iteration; the GPU is far from full at these sizes.  Same formulas as the two kernels.
// (1024-pixel rows: two own pairs per workgroup, i.e. twice the transforms -- measured slower, 1612 -> 1530 Mpix/s.)
template <class T, int LG>
struct RowPqGeom {
  using G = RowGeom<T, LG, true>;   // (this kernel only serves one small image: the latency-tuned geometry)
  static constexpr int NFH = G::NF + 2;                       // transform groups: own pairs + one halo pair each side
  static constexpr int THREADS = NFH * G::F::TPF;
  static constexpr size_t FFT_BYTES = (size_t)NFH * G::RS * sizeof(cpx<T>);
  static constexpr int PROWS = 2 * NFH;                       // rows of p kept for the stencil
  static constexpr int PPITCH = G::F::L + 4;                  // (a pad of 4 keeps 16-byte row alignment)
  static constexpr size_t LDS_BYTES = FFT_BYTES + (size_t)PROWS * PPITCH * sizeof(T);
};
template <class T, int LG>
__global__ __launch_bounds__((RowPqGeom<T, LG>::THREADS)) void rowidct_pq_kernel(
    const T* __restrict__ Z, const T* __restrict__ pin, T* __restrict__ pout, const T* __restrict__ wgt,
    T* __restrict__ q, int n0, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ wk, const int* flags,
    const double* part_rho, int nrho, double* part_pq, double* scal, int it, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    pin += pb * pimg;
    pout += pb * pimg;
    q += pb * pimg;
    if (wgt) wgt += (pb >> 1) * pimg;   // the two components of an image share its weight
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_rho += pb * PART_N;
    part_pq += pb * PART_N;
  }
  using G = RowGeom<T, LG, true>;
  using H = RowPqGeom<T, LG>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = F::TPF, N = F::L, E = F::E, NF = G::NF;
  static_assert(E == 8, "latency-tuned kernels use the 8-element transforms");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[H::THREADS];
  const int stop = flags[1];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  T* prow = reinterpret_cast<T*>(smem + H::FFT_BYTES);
  const int npairs = n0 / 2;
  const int pr = (int)blockIdx.x * NF + f - 1;        // group 0 / NF + 1: the halo pairs
  const bool valid = pr >= 0 && pr < npairs;
  const bool own = valid && f >= 1 && f <= NF;
  const size_t oa = (size_t)(valid ? 2 * pr : 0) * N, ob = oa + N;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[E], xm[E], wkv[E], pv[E];
  const bool first = it == 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    x[i] = {Z[oa + k], Z[ob + k]};
    xm[i] = k == 0 ? cpx<T>{T(0), T(0)} : cpx<T>{Z[oa + N - k], Z[ob + N - k]};
    wkv[i] = wk[k];
    pv[i] = first ? cpx<T>{T(0), T(0)} : cpx<T>{pin[oa + k], pin[ob + k]};
  }
  // f32: the stencil's weights are requested here too, with everything else (f64 has no registers to spare for them)
  const int xbase = 2 * (int)blockIdx.x * NF;                // first own image row; LDS row of image row x: x - xbase + 2
  constexpr int VPR = N / 4;                                  // 4-pixel items per row
  constexpr int NITEM = (2 * NF * VPR + H::THREADS - 1) / H::THREADS;
  constexpr bool WPRE = sizeof(T) == 4;
  Vec4<T> wcv[WPRE ? NITEM : 1], wuv[WPRE ? NITEM : 1], wdv[WPRE ? NITEM : 1];
  T wlv[WPRE ? NITEM : 1], wrv[WPRE ? NITEM : 1];
  if constexpr (WPRE) {
#pragma unroll
    for (int t = 0; t < NITEM; ++t) {
      const int item = threadIdx.x + t * H::THREADS;
      const int rl = item / VPR, c0 = (item % VPR) * 4;
      int xg = xbase + rl;
      const bool act = item < 2 * NF * VPR && xg < n0;
      xg = act ? xg : 0;
      const bool up = xg > 0, dn = xg + 1 < n0, hasl = c0 > 0, hasr = c0 + 4 < N;
      if (wgt) {
        const T* wp = wgt + (size_t)xg * N + c0;
        wcv[t] = *reinterpret_cast<const Vec4<T>*>(wp);
        wuv[t] = *reinterpret_cast<const Vec4<T>*>(up ? wp - N : wp);
        wdv[t] = *reinterpret_cast<const Vec4<T>*>(dn ? wp + N : wp);
        wlv[t] = wp[hasl ? -1 : 0];
        wrv[t] = wp[hasr ? 4 : 0];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) wcv[t].v[j] = wuv[t].v[j] = wdv[t].v[j] = T(1);
        wlv[t] = wrv[t] = T(1);
      }
    }
  }
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  D::inv_prepare(x, xm, wkv);
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::inv_scatter(x, lds, tid, T(1) / T(N));
  __syncthreads();
  D::inv_gather(x, lds, tid);
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int c = tid + TPF * i;
    T pa = x[i].x, pb = x[i].y;
    if (!first) {
      pa += beta * pv[i].x;
      pb += beta * pv[i].y;
    }
    if (own) {
      pout[oa + c] = pa;
      pout[ob + c] = pb;
    }
    prow[(2 * f) * H::PPITCH + c] = pa;
    prow[(2 * f + 1) * H::PPITCH + c] = pb;
  }
  __syncthreads();
  // ---- q = A^T W^2 A p on the 2 NF own rows, 4 pixels per item (as pq_kernel: min of the squared weights per edge)
  double pq = 0;
#pragma unroll
  for (int t = 0; t < NITEM; ++t) {
    const int item = threadIdx.x + t * H::THREADS;
    if (item >= 2 * NF * VPR) continue;
    const int rl = item / VPR, c0 = (item % VPR) * 4;
    const int xg = xbase + rl;
    if (xg >= n0) continue;
    const bool up = xg > 0, dn = xg + 1 < n0, hasl = c0 > 0, hasr = c0 + 4 < N;
    const T* pc = prow + (rl + 2) * H::PPITCH + c0;
    const Vec4<T> vc = *reinterpret_cast<const Vec4<T>*>(pc);
    const Vec4<T> vu = *reinterpret_cast<const Vec4<T>*>(pc - H::PPITCH), vd = *reinterpret_cast<const Vec4<T>*>(pc + H::PPITCH);
    const T pl = hasl ? pc[-1] : T(0), prr = hasr ? pc[4] : T(0);
    Vec4<T> wc, wu, wd;
    T wl = T(1), wr = T(1);
    if constexpr (WPRE) {
      wc = wcv[t]; wu = wuv[t]; wd = wdv[t]; wl = wlv[t]; wr = wrv[t];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) wc.v[j] = wu.v[j] = wd.v[j] = T(1);
      if (wgt) {
        const T* wp = wgt + (size_t)xg * N + c0;
        wc = *reinterpret_cast<const Vec4<T>*>(wp);
        wu = *reinterpret_cast<const Vec4<T>*>(up ? wp - N : wp);
        wd = *reinterpret_cast<const Vec4<T>*>(dn ? wp + N : wp);
        wl = hasl ? wp[-1] : T(1);
        wr = hasr ? wp[4] : T(1);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { wc.v[j] *= wc.v[j]; wu.v[j] *= wu.v[j]; wd.v[j] *= wd.v[j]; }
    wl *= wl;
    wr *= wr;
    Vec4<T> qv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T c = vc.v[j], wj = wc.v[j];
      T acc = T(0);
      if (j < 3) { const T wn = wc.v[j + 1]; acc += (wn < wj ? wn : wj) * (vc.v[j + 1] - c); }
      else if (hasr) acc += (wr < wj ? wr : wj) * (prr - c);
      if (j > 0) { const T wn = wc.v[j - 1]; acc += (wn < wj ? wn : wj) * (vc.v[j - 1] - c); }
      else if (hasl) acc += (wl < wj ? wl : wj) * (pl - c);
      if (dn) { const T wn = wd.v[j]; acc += (wn < wj ? wn : wj) * (vd.v[j] - c); }
      if (up) { const T wn = wu.v[j]; acc += (wn < wj ? wn : wj) * (vu.v[j] - c); }
      qv.v[j] = acc;
      pq += (double)c * (double)acc;
    }
    *reinterpret_cast<Vec4<T>*>(q + (size_t)xg * N + c0) = qv;
  }
  const double tot = block_sum(pq, sh);
  if (threadIdx.x == 0) part_pq[blockIdx.x] = tot;
}
template <class T, int LG>
hipError_t run_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq,
                            double* part_norm, int it, int* nnorm, int init, hipStream_t s) {
  if constexpr (!RowGeom<T, LG>::FITS) return hipErrorInvalidValue;
  else {
    const bool lat = unwrap_latency_tuned(w, LG);
    auto launch = [&](auto latc) -> hipError_t {
      constexpr bool LATC = decltype(latc)::value;
      using G = RowGeom<T, LG, LATC>;
      auto kern = rowdct_fused_kernel<T, LG, LATC>;
      static unsigned lds_set = 0;   // one flag word per instantiation of this lambda
      hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
      if (e != hipSuccess) return e;
      const int npairs = w->n0 / 2, grid = (npairs + G::NF - 1) / G::NF;
      *nnorm = grid;
      GPA_PROF("rowdct_fused_kernel", s);
      kern<<<dim3(grid, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((T*)w->r, (const T*)q, w->n0, (const cpx<T>*)w->tw1,
                                                   (const cpx<T>*)w->wk1, w->flags, part_pq, npq, part_norm, w->scal, it,
                                                   ring, init, (size_t)w->n0 * w->n1);
      return hipGetLastError();
    };
    if constexpr (LG <= GPA_UNWRAP_LAT_MAXLG) { if (lat) return launch(std::true_type{}); }
    return launch(std::false_type{});
  }
}

template <class T, int LG>
hipError_t run_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                         hipStream_t s) {
  if constexpr (!RowGeom<T, LG>::FITS) return hipErrorInvalidValue;
  else {
    const bool lat = unwrap_latency_tuned(w, LG);
    auto launch = [&](auto latc) -> hipError_t {
      constexpr bool LATC = decltype(latc)::value;
      using G = RowGeom<T, LG, LATC>;
      auto kern = rowidct_p_kernel<T, LG, LATC>;
      static unsigned lds_set = 0;
      hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
      if (e != hipSuccess) return e;
      const int npairs = w->n0 / 2, grid = (npairs + G::NF - 1) / G::NF;
      GPA_PROF("rowidct_p_kernel", s);
      kern<<<dim3(grid, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((const T*)w->z, (const T*)pin, (T*)pout, w->n0, (const cpx<T>*)w->tw1,
                                                   (const cpx<T>*)w->wk1, w->flags, part_rho, nrho, w->scal, it, (size_t)w->n0 * w->n1);
      return hipGetLastError();
    };
    if constexpr (LG <= GPA_UNWRAP_LAT_MAXLG) { if (lat) return launch(std::true_type{}); }
    return launch(std::false_type{});
  }
}
// the row kernel and the stencil in one launch (one image, rows up to 512 pixels); *npq_out = partial sums of <p, q>
template <class T, int LG>
hipError_t run_rowidct_pq(const Impl* w, const void* pin, void* pout, const void* weight, const double* part_rho,
                          int nrho, double* part_pq, int* npq_out, int it, hipStream_t s) {
  if constexpr (LG > GPA_ROWPQ_MAXLG || unwrap_elems(LG, sizeof(T)) != 8) return hipErrorInvalidValue;
  else {
    using G = RowGeom<T, LG, true>;
    using H = RowPqGeom<T, LG>;
    auto kern = rowidct_pq_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)H::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = w->n0 / 2, grid = (npairs + G::NF - 1) / G::NF;
    if (grid > MAXPART) return hipErrorInvalidValue;
    *npq_out = grid;
    GPA_PROF("rowidct_pq_kernel", s);
    kern<<<dim3(grid, 1, w->nprob), H::THREADS, H::LDS_BYTES, s>>>((const T*)w->z, (const T*)pin, (T*)pout, (const T*)weight,
                                                 (T*)w->q, w->n0, (const cpx<T>*)w->tw1, (const cpx<T>*)w->wk1, w->flags,
                                                 part_rho, nrho, part_pq, w->scal, it, (size_t)w->n0 * w->n1);
    return hipGetLastError();
  }
}

// ---------------------------------------------------------------------------
// Long rows (8192 and 16384 points): ONE row per complex transform of HALF the row length.
//
// The kernels above pack two rows into one N-point complex transform.  At 8192 / 16384 points that transform needs 512 /
// 1024 threads and 64 / 128 KiB of LDS: one workgroup per CU, and at 1024 threads a 128-register budget that the
// fused kernels overrun by 70-150 bytes of scratch per lane (16384^2: 0.12-0.15 of the HBM rate).  A real sequence of N
// points needs only an N/2-point complex transform: with v the Makhoul-permuted row (gpa_dct.h),
//     t[n] = v[2n] + i v[2n+1],  T = FFT_(N/2)(t),  Ve_k = (T_k + conj T_(N/2-k)) / 2,  Vo_k = -i (T_k - conj T_(N/2-k)) / 2,
//     V_k = Ve_k + e^(-2 pi i k / N) Vo_k = DFT_N(v)_k,  U_k = w_k V_k,   X_k = 2 Re U_k,  X_(N-k) = -2 Im U_k   (0 < k < N/2),
//     X_0 = 2 (Re T_0 + Im T_0),  X_(N/2) = sqrt 2 (Re T_0 - Im T_0)
// -- the same numbers as the packed form (SciPy's unnormalised DCT-II, phase_unwrap.py:84-103), half the LDS and half
// the threads per workgroup, no scratch, two or more workgroups per CU.  In samples: t[j] = (x[4j], x[4j+2]) and
// t[N/2-1-j] = (x[4j+3], x[4j+1]), so a row is staged (and written back) with 16-byte accesses.  The inverse runs the
// chain backwards (conj-forward-conj for the inverse transform, as everywhere).
// ---------------------------------------------------------------------------
template <class T, int LG>
struct RowHalfGeom {
  using F = WgFFT<T, LG - 1, 16>;
  static constexpr int N = 1 << LG, HN = N / 2, TPF = F::TPF, THREADS = F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)F::LDS_ELEMS * sizeof(cpx<T>);
};

// (f32 rows of 16384 points: two 512-thread workgroups per CU, i.e. 4 waves per SIMD and at most 128 registers)
#ifndef GPA_ROWHALF14_WAVES
#define GPA_ROWHALF14_WAVES 4
#endif
template <class T, int LG>
__global__ __launch_bounds__((RowHalfGeom<T, LG>::THREADS), (sizeof(T) == 4 && LG == 14 ? GPA_ROWHALF14_WAVES : 1)) void rowdct_half_kernel(
    T* __restrict__ r, const T* __restrict__ q, const cpx<T>* __restrict__ twh, const cpx<T>* __restrict__ twn,
    const cpx<T>* __restrict__ wk, int* flags, const double* part_pq, int npq, double* part_norm, double* scal, int it,
    int ring, int init, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    r += pb * pimg;
    q += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_pq += pb * PART_N;
    part_norm += pb * PART_N;
  }
  // same contract as rowdct_fused_kernel: it == 0: r (spatial) -> R in place; it > 0: R -= alpha DCT-II_rows(q), partial ||r||^2
  using G = RowHalfGeom<T, LG>;
  using F = typename G::F;
  constexpr int N = G::N, HN = G::HN, TPF = G::TPF, E = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowHalfGeom<T, LG>::THREADS];
  if (init) { if (!solve_init(part_pq, npq, scal, flags, sh)) return; }
  else if (flags[1]) return;
  const int tid = threadIdx.x;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem);
  const size_t o = (size_t)blockIdx.x * N;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twh, tid);
  T alpha = T(0);
  if (it > 0) {
    const double pq = reduce_partials(part_pq, npq, sh);
    const double alpha_d = scal[8 + ((it - 1) & 1)] / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
  }
  const T* src = (it > 0 ? q : r) + o;
  // the row comes in with 16-byte accesses and is parked in LDS as the half-length transform's input
  constexpr int NV = N / (4 * TPF);
  Vec4<T> stage[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) stage[v] = *reinterpret_cast<const Vec4<T>*>(src + 4 * (tid + TPF * v));
  // the kept spectrum is requested before the transform so that its latency hides behind it -- except for f32 rows of
  // 16384 points, whose 128-register budget (two workgroups per CU) it would overrun: there it is requested after
  // the transform, and the other workgroup of the CU covers the wait
#ifndef GPA_ROWHALF_LATE_R
#define GPA_ROWHALF_LATE_R (sizeof(T) == 4 && LG == 14)
#endif
  constexpr bool LATE_R = GPA_ROWHALF_LATE_R;
  T rlo[E], rhi[E];
  auto load_kept = [&]() {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tid + TPF * i;
      rlo[i] = r[o + k];
      rhi[i] = r[o + (k == 0 ? HN : N - k)];
    }
  };
  if (it > 0 && !LATE_R) load_kept();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int j = tid + TPF * v;
    lds[F::pad(j)] = {stage[v].v[0], stage[v].v[2]};
    lds[F::pad(HN - 1 - j)] = {stage[v].v[3], stage[v].v[1]};
  }
  __syncthreads();
  cpx<T> x[E];
#pragma unroll
  for (int i = 0; i < E; ++i) x[i] = lds[F::pad(tid + TPF * i)];
  __syncthreads();
  F::forward(x, lds, tid, tw);
  if (it > 0 && LATE_R) load_kept();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < E; ++i) lds[F::pad(F::spec_index(tid, i))] = x[i];
  __syncthreads();
  double sq = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    const cpx<T> zk = lds[F::pad(k)], zm = lds[F::pad((HN - k) & (HN - 1))];
    T xlo, xhi;
    if (k == 0) {
      xlo = T(2) * (zk.x + zk.y);
      xhi = T(1.41421356237309504880) * (zk.x - zk.y);
    } else {
      const cpx<T> ve = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};   // (T_k + conj T_m) / 2
      const cpx<T> vo = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};  // -i (T_k - conj T_m) / 2
      const cpx<T> V = ve + cmul(twn[k], vo);
      const cpx<T> U = cmul(wk[k], V);
      xlo = T(2) * U.x;
      xhi = T(-2) * U.y;
    }
    if (it > 0) {
      xlo = rlo[i] - alpha * xlo;
      xhi = rhi[i] - alpha * xhi;
      // sum_n r^2 = (1 / 2N) sum_k c_k R_k^2, c_0 = 1/2 (SciPy's unnormalised DCT-II)
      sq += (k == 0 ? 0.5 : 1.0) * (double)xlo * (double)xlo + (double)xhi * (double)xhi;
    }
    r[o + k] = xlo;
    r[o + (k == 0 ? HN : N - k)] = xhi;
  }
  if (it > 0) {
    const double tot = block_sum(sq, sh);
    if (threadIdx.x == 0) part_norm[blockIdx.x] = tot / (2.0 * N);
  }
}

template <class T, int LG>
__global__ __launch_bounds__((RowHalfGeom<T, LG>::THREADS)) void rowidct_p_half_kernel(
    const T* __restrict__ Z, const T* __restrict__ pin, T* __restrict__ pout, const cpx<T>* __restrict__ twh,
    const cpx<T>* __restrict__ twn, const cpx<T>* __restrict__ wk, const int* flags, const double* part_rho, int nrho,
    double* scal, int it, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    pin += pb * pimg;
    pout += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_rho += pb * PART_N;
  }
  // same contract as rowidct_p_kernel: row DCT-III of Z straight into p = z + beta p_prev (phase_unwrap.py:336-340)
  using G = RowHalfGeom<T, LG>;
  using F = typename G::F;
  constexpr int N = G::N, HN = G::HN, TPF = G::TPF, E = 16;
  const int stop = flags[1];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowHalfGeom<T, LG>::THREADS];
  const int tid = threadIdx.x;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem);
  const size_t o = (size_t)blockIdx.x * N;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twh, tid);
  const bool first = it == 0;
  // every input is requested before anything waits
  T zlo[E], zhi[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    zlo[i] = Z[o + k];
    zhi[i] = Z[o + (k == 0 ? HN : N - k)];
  }
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  // V_k = conj(w_k) (X_k - i X_(N-k)) / 2;  V_0 = X_0 / 2 and V_(N/2) = X_(N/2) / sqrt 2 are real
  cpx<T> x[E];
  T vh = T(0);
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    if (k == 0) {
      x[i] = {T(0.5) * zlo[i], T(0)};
      vh = T(0.70710678118654752440) * zhi[i];
    } else {
      x[i] = cmulc(cpx<T>{T(0.5) * zlo[i], T(-0.5) * zhi[i]}, wk[k]);
    }
    lds[F::pad(k)] = x[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    cpx<T> Tk;
    if (k == 0) {
      Tk = {T(0.5) * (x[i].x + vh), T(0.5) * (x[i].x - vh)};
    } else {
      const cpx<T> vm = lds[F::pad(HN - k)];
      const cpx<T> ve = {T(0.5) * (x[i].x + vm.x), T(0.5) * (x[i].y - vm.y)};       // (V_k + conj V_m) / 2
      const cpx<T> d = {x[i].x - vm.x, x[i].y + vm.y};                              // V_k - conj V_m
      const cpx<T> vo = cscale(cmulc(d, twn[k]), T(0.5));                           // conj(E_k) (.) / 2
      Tk = {ve.x - vo.y, ve.y + vo.x};                                              // Ve + i Vo
    }
    x[i] = {Tk.x, -Tk.y};                                                           // IFFT = conj(FFT(conj .))
  }
  __syncthreads();
  F::forward(x, lds, tid, tw);
  __syncthreads();
  const T inv = T(1) / T(HN);
#pragma unroll
  for (int i = 0; i < E; ++i) lds[F::pad(F::spec_index(tid, i))] = {x[i].x * inv, -x[i].y * inv};
  __syncthreads();
  constexpr int NV = N / (4 * TPF);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int j = tid + TPF * v;
    const cpx<T> a = lds[F::pad(j)], b = lds[F::pad(HN - 1 - j)];
    Vec4<T> out = {{a.x, b.y, a.y, b.x}};
    if (!first) {
      const Vec4<T> pv = *reinterpret_cast<const Vec4<T>*>(pin + o + 4 * j);
#pragma unroll
      for (int c = 0; c < 4; ++c) out.v[c] += beta * pv.v[c];
    }
    *reinterpret_cast<Vec4<T>*>(pout + o + 4 * j) = out;
  }
}

template <class T, int LG>
hipError_t run_rowdct_half(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm, int it,
                           int* nnorm, int init, hipStream_t s) {
  using G = RowHalfGeom<T, LG>;
  auto kern = rowdct_half_kernel<T, LG>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
  if (e != hipSuccess) return e;
  if (w->n0 > MAXPART) return hipErrorInvalidValue;
  *nnorm = w->n0;
  GPA_PROF("rowdct_fused_kernel", s);
  kern<<<dim3(w->n0, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((T*)w->r, (const T*)q, (const cpx<T>*)w->tw1h, (const cpx<T>*)w->tw1,
                                                                (const cpx<T>*)w->wk1, w->flags, part_pq, npq, part_norm, w->scal, it,
                                                                ring, init, (size_t)w->n0 * w->n1);
  return hipGetLastError();
}
template <class T, int LG>
hipError_t run_rowidct_p_half(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it, hipStream_t s) {
  using G = RowHalfGeom<T, LG>;
  auto kern = rowidct_p_half_kernel<T, LG>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
  if (e != hipSuccess) return e;
  GPA_PROF("rowidct_p_kernel", s);
  kern<<<dim3(w->n0, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((const T*)w->z, (const T*)pin, (T*)pout, (const cpx<T>*)w->tw1h,
                                                                (const cpx<T>*)w->tw1, (const cpx<T>*)w->wk1, w->flags, part_rho, nrho,
                                                                w->scal, it, (size_t)w->n0 * w->n1);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Stencil and row transform in ONE launch (rows of 2048 / 4096 points, streamed column solve): D = DCT-II_rows(A^T W^2 A p).
//
// q = A^T W^2 A p (phase_unwrap.py:118-132) has a single consumer, the row transform of the next iteration's residual
// update r -= alpha q, and by linearity that update is R -= alpha DCT_rows(q) on the kept row spectrum.  alpha = rho / <p, q>
// needs the whole image's <p, q>, so the transform cannot finish the update -- but it need not: it writes D = DCT_rows(q)
// where q used to go, and the chunk-sum kernel of the streamed column solve, which reads R anyway, applies R -= alpha D
// on the fly (colstream_agg_kernel<..., UPDATE>).  q never reaches HBM; the iteration is
//     pqdct (p, w in, D out) -> colstream agg (R, D in, R out) -> scan -> apply (R in, Z out) -> rowidct_p (Z, p in, p out)
// five launches and 44 bytes per pixel instead of six and 48.  A workgroup owns a row pair (2 pr, 2 pr + 1): it reads
// the four rows pr*2 - 1 .. pr*2 + 2 of p and w with 16-byte accesses (the halo rows are its neighbours' own rows: L2),
// forms q of its two rows in registers exactly as pq_kernel does (same edge order: right, left, down, up), parks them
// in LDS as the packed pair's transform input, and runs rowdct_fused_kernel's transform.
// ---------------------------------------------------------------------------
#ifndef GPA_PQDCT_WAVES
#define GPA_PQDCT_WAVES 4
#endif
template <class T, int LG>
__global__ __launch_bounds__((RowGeom<T, LG>::THREADS), (sizeof(T) == 8 ? GPA_F64_WAVES : GPA_PQDCT_WAVES)) void pqdct_kernel(
    const T* __restrict__ p, const T* __restrict__ wgt, T* __restrict__ Dout, int n0, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ wk, const int* flags, double* part_pq, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    p += pb * pimg;
    Dout += pb * pimg;
    if (wgt) wgt += (pb >> 1) * pimg;   // the two components of an image share its weight
    flags += pb * FLAGS_N;
    part_pq += pb * PART_N;
  }
  using G = RowGeom<T, LG>;
  using F = typename G::F;
  using D = typename G::D;
  static_assert(G::NF == 1, "one row pair per workgroup");
  constexpr int TPF = F::TPF, N = F::L, E = F::E;
  const int stop = flags[1];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowGeom<T, LG>::THREADS];
  const int tid = threadIdx.x, lane = tid & 63;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem);
  // XCD-aware order (gpa_internal.h): the workgroups that share an XCD, and with it an L2, own CONSECUTIVE row pairs, so the
  // halo rows a workgroup reads are its neighbours' own rows in the same L2 (round robin would fetch every row twice
  // from HBM: 67 -> 5x us at 4096^2)
  const int xa = 2 * xcd_tile((int)blockIdx.x, (int)gridDim.x), xb = xa + 1;
  const bool up = xa > 0, dn = xb + 1 < n0;
  const size_t oa = (size_t)xa * N, ob = oa + N;
  constexpr int NQ = N / (4 * TPF);   // 16-byte vectors per thread and row
  // The row pair's columns are worked through in NPH phases of VP vectors.  Every load of a phase -- four rows of p, four
  // rows of w, the neighbour pixels of the two lanes at the ends of a wavefront -- is requested before anything of the
  // phase is used, and the second phase's rows of p are requested before the first phase computes: two memory round
  // trips per workgroup (the first cut of this kernel waited once per vector and row of w: five).
  constexpr int NPH = NQ >= 2 ? 2 : 1, VP = NQ / NPH;
  struct Rows { Vec4<T> u[VP], a[VP], b[VP], d[VP]; };
  struct Edge { T la[VP], ra[VP], lb[VP], rb[VP]; };
  auto load_rows = [&](const T* base, int ph, Rows& r) {
#pragma unroll
    for (int v = 0; v < VP; ++v) {
      const int c0 = 4 * (tid + TPF * (ph * VP + v));
      // (halo rows outside the image are never used: their address is clamped to a row of the pair)
      r.u[v] = *reinterpret_cast<const Vec4<T>*>(base + (up ? oa - N : oa) + c0);
      r.a[v] = *reinterpret_cast<const Vec4<T>*>(base + oa + c0);
      r.b[v] = *reinterpret_cast<const Vec4<T>*>(base + ob + c0);
      r.d[v] = *reinterpret_cast<const Vec4<T>*>(base + (dn ? ob + N : ob) + c0);
    }
  };
  // left / right neighbours of a thread's four pixels come from the adjacent lanes; the two lanes at the ends of a
  // wavefront go to memory (requested here, used in compute())
  auto load_edges = [&](const T* base, int ph, Edge& e) {
#pragma unroll
    for (int v = 0; v < VP; ++v) {
      const int c0 = 4 * (tid + TPF * (ph * VP + v));
      e.la[v] = e.lb[v] = e.ra[v] = e.rb[v] = T(1);
      if (lane == 0 && c0 > 0) { e.la[v] = base[oa + c0 - 1]; e.lb[v] = base[ob + c0 - 1]; }
      if (lane == 63 && c0 + 4 < N) { e.ra[v] = base[oa + c0 + 4]; e.rb[v] = base[ob + c0 + 4]; }
    }
  };
  auto ones = [&](Rows& r, Edge& e) {
#pragma unroll
    for (int v = 0; v < VP; ++v) {
#pragma unroll
      for (int j = 0; j < 4; ++j) r.u[v].v[j] = r.a[v].v[j] = r.b[v].v[j] = r.d[v].v[j] = T(1);
      e.la[v] = e.lb[v] = e.ra[v] = e.rb[v] = T(1);
    }
  };
  double pq = 0;
  auto compute = [&](int ph, const Rows& P, Rows& W, const Edge& EP, Edge& EW) {
#pragma unroll
    for (int v = 0; v < VP; ++v) {
      const int c0 = 4 * (tid + TPF * (ph * VP + v));
      const bool hasl = c0 > 0, hasr = c0 + 4 < N;
      if (wgt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { W.u[v].v[j] *= W.u[v].v[j]; W.a[v].v[j] *= W.a[v].v[j]; W.b[v].v[j] *= W.b[v].v[j]; W.d[v].v[j] *= W.d[v].v[j]; }
        EW.la[v] *= EW.la[v]; EW.lb[v] *= EW.lb[v]; EW.ra[v] *= EW.ra[v]; EW.rb[v] *= EW.rb[v];
      }
      T pla = __shfl_up(P.a[v].v[3], 1), pra = __shfl_down(P.a[v].v[0], 1), plb = __shfl_up(P.b[v].v[3], 1), prb = __shfl_down(P.b[v].v[0], 1);
      T wla = __shfl_up(W.a[v].v[3], 1), wra = __shfl_down(W.a[v].v[0], 1), wlb = __shfl_up(W.b[v].v[3], 1), wrb = __shfl_down(W.b[v].v[0], 1);
      if (lane == 0) { pla = EP.la[v]; plb = EP.lb[v]; wla = EW.la[v]; wlb = EW.lb[v]; }
      if (lane == 63) { pra = EP.ra[v]; prb = EP.rb[v]; wra = EW.ra[v]; wrb = EW.rb[v]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // q = sum over the 4 edges of min(w^2, w_nb^2) * (p_nb - p)   (phase_unwrap.py:118-132), row a then row b
        T qa, qb;
        {
          const T c = P.a[v].v[j], wj = W.a[v].v[j];
          T acc = T(0);
          if (j < 3) { const T wn = W.a[v].v[j + 1]; acc += (wn < wj ? wn : wj) * (P.a[v].v[j + 1] - c); }
          else if (hasr) acc += (wra < wj ? wra : wj) * (pra - c);
          if (j > 0) { const T wn = W.a[v].v[j - 1]; acc += (wn < wj ? wn : wj) * (P.a[v].v[j - 1] - c); }
          else if (hasl) acc += (wla < wj ? wla : wj) * (pla - c);
          { const T wn = W.b[v].v[j]; acc += (wn < wj ? wn : wj) * (P.b[v].v[j] - c); }
          if (up) { const T wn = W.u[v].v[j]; acc += (wn < wj ? wn : wj) * (P.u[v].v[j] - c); }
          qa = acc;
          pq += (double)c * (double)acc;
        }
        {
          const T c = P.b[v].v[j], wj = W.b[v].v[j];
          T acc = T(0);
          if (j < 3) { const T wn = W.b[v].v[j + 1]; acc += (wn < wj ? wn : wj) * (P.b[v].v[j + 1] - c); }
          else if (hasr) acc += (wrb < wj ? wrb : wj) * (prb - c);
          if (j > 0) { const T wn = W.b[v].v[j - 1]; acc += (wn < wj ? wn : wj) * (P.b[v].v[j - 1] - c); }
          else if (hasl) acc += (wlb < wj ? wlb : wj) * (plb - c);
          if (dn) { const T wn = W.d[v].v[j]; acc += (wn < wj ? wn : wj) * (P.d[v].v[j] - c); }
          { const T wn = W.a[v].v[j]; acc += (wn < wj ? wn : wj) * (P.a[v].v[j] - c); }
          qb = acc;
          pq += (double)c * (double)acc;
        }
        lds[F::pad(c0 + j)] = {qa, qb};
      }
    }
  };
  Rows pA, wA;
  Edge epA, ewA;
  load_rows(p, 0, pA);
  load_edges(p, 0, epA);
  if (wgt) { load_rows(wgt, 0, wA); load_edges(wgt, 0, ewA); } else ones(wA, ewA);
  if (stop) return;
  compute(0, pA, wA, epA, ewA);
  if constexpr (NPH == 2) {
    // (the second phase reuses the first one's registers: requesting its rows of p ahead of the first phase's arithmetic
    //  overruns the 128 registers of four waves per SIMD by 100 bytes of scratch)
    asm volatile("" ::: "memory");   // (the compiler must not hoist these loads into the first phase either)
    __builtin_amdgcn_sched_barrier(0);
    load_rows(p, 1, pA);
    load_edges(p, 1, epA);
    if (wgt) { load_rows(wgt, 1, wA); load_edges(wgt, 1, ewA); } else ones(wA, ewA);
    compute(1, pA, wA, epA, ewA);
  }
  // (the transform's twiddles only now: requested before the stencil they would hold 12-24 registers through it)
  typename G::TW tw;
  __shared__ cpx<T> t1s[G::T1N];
  if constexpr (G::TWLDS) {
    F::fill_pass1_table(t1s, twtab, threadIdx.x, G::THREADS);
    __syncthreads();
    F::load_twiddles(tw, twtab, tid, t1s);
  } else {
    F::load_twiddles(tw, twtab, tid);
  }
  __syncthreads();
  cpx<T> x[E];
#pragma unroll
  for (int i = 0; i < E; ++i) x[i] = lds[F::pad(makhoul_src(tid + TPF * i, N))];
  __syncthreads();
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::fwd_scatter(x, lds, tid);
  __syncthreads();
  D::fwd_gather(x, lds, tid, wk);
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    Dout[oa + k] = x[i].x;
    Dout[ob + k] = x[i].y;
  }
  const double tot = block_sum(pq, sh);
  if (threadIdx.x == 0) part_pq[blockIdx.x] = tot;
}

template <class T, int LG>
hipError_t run_pqdct(const Impl* w, const void* p, const void* weight, double* part_pq, int* npq, hipStream_t s) {
  using G = RowGeom<T, LG>;
  auto kern = pqdct_kernel<T, LG>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
  if (e != hipSuccess) return e;
  const int grid = w->n0 / 2;
  if (grid > MAXPART) return hipErrorInvalidValue;
  *npq = grid;
  GPA_PROF("pqdct_kernel", s);
  kern<<<dim3(grid, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((const T*)p, (const T*)weight, (T*)w->q, w->n0, (const cpx<T>*)w->tw1,
                                                               (const cpx<T>*)w->wk1, w->flags, part_pq, (size_t)w->n0 * w->n1);
  return hipGetLastError();
}
// rows of 8192 points and more take the half-length kernels (NO_ROWHALF: the packed ones, for tests and measurements)
inline bool use_row_half(const Impl* w) {
  const int minlg = opt_set(OPT_ROWHALF_MINLG) ? (int)opt(OPT_ROWHALF_MINLG).num : GPA_ROWHALF_MINLG;
  return w->lg1 >= minlg && w->lg1 >= 12 && w->tw1h && !opt_set(OPT_NO_ROWHALF);
}
}  // namespace

bool pow2_pqdct_offered(const Impl* w) { return !w->generic && (w->lg1 == 11 || w->lg1 == 12) && (w->n0 % 2) == 0; }
hipError_t pow2_pqdct(const Impl* w, const void* p, const void* weight, double* part_pq, int* npq, hipStream_t s) {
  if (pow2_rowpers_offered(w) && w->n0 >= 16 && !opt_set(OPT_NO_PQPERS)) return pow2_pqdct_pers(w, p, weight, part_pq, npq, s);
  if (w->lg1 == 11) return w->dtype == 0 ? run_pqdct<float, 11>(w, p, weight, part_pq, npq, s) : run_pqdct<double, 11>(w, p, weight, part_pq, npq, s);
  if (w->lg1 == 12) return w->dtype == 0 ? run_pqdct<float, 12>(w, p, weight, part_pq, npq, s) : run_pqdct<double, 12>(w, p, weight, part_pq, npq, s);
  return hipErrorInvalidValue;
}
hipError_t pow2_rowidct_pq(const Impl* w, const void* pin, void* pout, const void* weight, const double* part_rho,
                           int nrho, double* part_pq, int* npq_out, int it, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowidct_pq<float, LG>(w, pin, pout, weight, part_rho, nrho, part_pq, npq_out, it, s) \
                                               : run_rowidct_pq<double, LG>(w, pin, pout, weight, part_rho, nrho, part_pq, npq_out, it, s);
  switch (w->lg1) { CASE(6) CASE(7) CASE(8) CASE(9) }
#undef CASE
  return hipErrorInvalidValue;
}
hipError_t pow2_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                          hipStream_t s) {
  if (use_row_half(w)) {
    if (w->lg1 == 12) return w->dtype == 0 ? run_rowidct_p_half<float, 12>(w, pin, pout, part_rho, nrho, it, s)
                                            : run_rowidct_p_half<double, 12>(w, pin, pout, part_rho, nrho, it, s);
    if (w->lg1 == 13) return w->dtype == 0 ? run_rowidct_p_half<float, 13>(w, pin, pout, part_rho, nrho, it, s)
                                            : run_rowidct_p_half<double, 13>(w, pin, pout, part_rho, nrho, it, s);
    if (w->lg1 == 14 && w->dtype == 0) return run_rowidct_p_half<float, 14>(w, pin, pout, part_rho, nrho, it, s);
  }
  if (pow2_rowpers_offered(w)) return pow2_rowidct_p_pers(w, pin, pout, part_rho, nrho, it, s);
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowidct_p<float, LG>(w, pin, pout, part_rho, nrho, it, s) \
                                               : run_rowidct_p<double, LG>(w, pin, pout, part_rho, nrho, it, s);
  switch (w->lg1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
// init: first iteration of a solve on prepared residuals -- part_pq / npq are then the producer's partial norms
hipError_t pow2_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm,
                             int it, int* nnorm, int init, hipStream_t s) {
  // f64 rows of 4096 points: the forward kernel alone gains from the half-length form (116 -> 99 us per launch; the inverse
  // loses, 95 -> 126, and stays packed)
  if (w->dtype == 1 && w->lg1 == 12 && w->tw1h && !opt_set(OPT_NO_ROWHALF) && !opt_set(OPT_ROWHALF_MINLG))
    return run_rowdct_half<double, 12>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  if (use_row_half(w)) {
    if (w->lg1 == 12) return w->dtype == 0 ? run_rowdct_half<float, 12>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                                            : run_rowdct_half<double, 12>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
    if (w->lg1 == 13) return w->dtype == 0 ? run_rowdct_half<float, 13>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                                            : run_rowdct_half<double, 13>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
    if (w->lg1 == 14 && w->dtype == 0) return run_rowdct_half<float, 14>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  }
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowdct_fused<float, LG>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s) \
                                               : run_rowdct_fused<double, LG>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  switch (w->lg1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}

}  // namespace gpa
