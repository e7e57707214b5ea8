// a7, row kernels of the fused PCG iteration on power-of-two rows (phase_unwrap.py:326-349; preconditioner :95-115):
//   rowdct_fused : R -= alpha DCT-II_rows(q)   the residual is kept as its row spectrum R; ||r||^2 by Parseval
//   rowidct_p    : Z -> p = z + beta p_prev    row DCT-III straight into the new search direction
//   rowidct_pq   : the same and the stencil in one launch (one image, rows up to 512 pixels)
#include "gpa_unwrap_rowgeom.h"

namespace gpa {
namespace {

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

// rows of 8192 points and more take the half-length kernels (NO_ROWHALF: the packed ones, for tests and measurements)
inline bool use_row_half(const Impl* w) {
  const int minlg = opt_set(OPT_ROWHALF_MINLG) ? (int)opt(OPT_ROWHALF_MINLG).num : GPA_ROWHALF_MINLG;
  return w->lg1 >= minlg && w->lg1 >= 12 && w->tw1h && !opt_set(OPT_NO_ROWHALF);
}
}  // namespace

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
  if (use_row_half(w) && rowhalf_offered(w))
    return rowhalfpers_offered(w) ? rowhalfpers_rowidct_p(w, pin, pout, part_rho, nrho, it, s) : rowhalf_rowidct_p(w, pin, pout, part_rho, nrho, it, s);
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
    return rowhalf_rowdct(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  if (use_row_half(w) && rowhalf_offered(w))
    return rowhalfpers_offered(w) ? rowhalfpers_rowdct(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                                  : rowhalf_rowdct(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowdct_fused<float, LG>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s) \
                                               : run_rowdct_fused<double, LG>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  switch (w->lg1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}

}  // namespace gpa
