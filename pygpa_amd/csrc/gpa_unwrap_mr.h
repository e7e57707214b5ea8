// Fused PCG kernels for image sizes that are not powers of two, on the mixed-radix LDS-resident FFT
// (gpa_mrfft.h).  Included by gpa_unwrap_generic.hip only (Impl, reductions and scalar conventions: gpa_unwrap_impl.h).
//
// Same iteration as the power-of-two fused path (rowdct_fused -> colsolve -> rowidct_p -> pq, the residual kept
// as its row spectrum, rho and ||r|| by Parseval), same scalars and flags, so run_pcg() drives both with one
// loop.  What differs is where a transform lives: here the packed pair of rows / columns sits in LDS in the
// Makhoul order (slot m holds sample makhoul_src(m)), the FFT passes run in place, and the DCT pre / post
// processing reads bins k and n - k from LDS.  Global accesses are lane-consecutive along the rows and batched
// (every load of a thread is issued before the first use), V pixels per access: V = 4 where rows are whole 16-byte
// vectors, V = 1 otherwise; the column kernel takes NF adjacent column pairs per workgroup (NF * 2 * sizeof(T)
// contiguous bytes per row).  A DFT is mr_dft(): the mixed-radix transform of length n, or chirp-z on a smooth
// L >= 2n - 1 when n has a prime factor > 13.  blockIdx.z selects the problem of a batched solve (see gpa_unwrap.hip).
//
// Replaces, per PCG iteration, the Bluestein kernels g_rowdct / g_colsolve / g_rowidct (4 FFTs of length
// >= 2n - 1 per DCT pair instead of one of length n) and the separate pupdate / applyq / update / scal_* kernels:
// 4 launches per iteration instead of 9.  (phase_unwrap.py:84-115 preconditioner, :326-349 iteration.)
#pragma once
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

// LDS slot of sample c in the Makhoul order (inverse of makhoul_src)
__device__ __forceinline__ int mr_slot_of(int c, int n) { return (c & 1) ? n - 1 - (c >> 1) : (c >> 1); }

// rows: R = DCT-II_rows(r) (it == 0, in place) or R -= alpha DCT-II_rows(q) with the partial ||r||^2 (it > 0)
template <class T, int MAXT, int V>
__global__ __launch_bounds__(MAXT) void mr_rowdct_fused_kernel(
    T* __restrict__ r, const T* __restrict__ q, int n0, const MrDft d, int rs, const cpx<T>* __restrict__ W,
    const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec,
    const cpx<T>* __restrict__ wk, int* flags, const double* part_pq, int npq, double* part_norm, double* scal,
    int it, int ring, int init, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    r += pb * pimg;
    q += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_pq += pb * PART_N;
    part_norm += pb * PART_N;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[32];
  // init (first iteration of a solve on prepared residuals): part_pq / npq are the producer's partial norms of r0
  int stop = 0;
  double pq_part = 0, rho_it = 0;
  if (init) { if (!solve_init(part_pq, npq, scal, flags, sh)) return; }
  else {
    // flags, partial sums and rho in one round trip; the early exit after it
    stop = flags[1];
    if (it > 0) {
      pq_part = load_partials(part_pq, npq);
      rho_it = scal[8 + ((it - 1) & 1)];
    }
    if (stop) return;
  }
  const int n = d.n, Tn = d.pl.T;
  const int tid = threadIdx.x % Tn, f = threadIdx.x / Tn, nf = blockDim.x / Tn;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * rs;
  const int pr = blockIdx.x * nf + f;
  const int xa = 2 * pr, xb = xa + 1;
  const bool va = xa < n0, vb = xb < n0;
  const size_t oa = (size_t)(va ? xa : 0) * n, ob = (size_t)(vb ? xb : 0) * n;
  T alpha = T(0);
  if (it > 0) {
    const double pq = block_sum(pq_part, sh);
    const double alpha_d = rho_it / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
  }
  const T* src = it > 0 ? q : r;
  // all loads of a thread are issued before the first use (a loop of load -> LDS store pays one memory latency per
  // trip); rows are whole 4-pixel vectors, n / (4 Tn) <= 4 vectors per thread and row
  constexpr int NV = MR_REGS / V;
  VecN<T, V> zero4;
#pragma unroll
  for (int j = 0; j < V; ++j) zero4.v[j] = T(0);
  {
    VecN<T, V> qa[NV], qb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      // (loads are unconditional from a clamped address and zeroed by a select afterwards: a load inside a branch
      //  sits in its own basic block and is waited for on its own, see pass B of the sweep)
      const int c0 = V * (tid + Tn * i), cc = c0 < n ? c0 : 0;
      qa[i] = *reinterpret_cast<const VecN<T, V>*>(src + oa + cc);
      qb[i] = *reinterpret_cast<const VecN<T, V>*>(src + ob + cc);
      if (!(c0 < n && va)) qa[i] = zero4;
      if (!(c0 < n && vb)) qb[i] = zero4;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = V * (tid + Tn * i);
      if (c0 < n) {
#pragma unroll
        for (int j = 0; j < V; ++j) lds[mr_pad(mr_slot_of(c0 + j, n))] = {qa[i].v[j], qb[i].v[j]};
      }
    }
  }
  __syncthreads();   // (it == 0, in place: both rows are in LDS before any bin is written)
  mr_dft<MAXT>(lds, d, W, chirp, bspec, tid);
  double sq = 0;
  {
    VecN<T, V> ra[NV], rb[NV];
    struct alignas(V * sizeof(cpx<T>)) W4 { cpx<T> v[V]; };
    W4 w4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k0 = V * (tid + Tn * i), kc = k0 < n ? k0 : 0;
      ra[i] = *reinterpret_cast<const VecN<T, V>*>(r + oa + kc);
      rb[i] = *reinterpret_cast<const VecN<T, V>*>(r + ob + kc);
      if (!(it > 0 && k0 < n && va)) ra[i] = zero4;
      if (!(it > 0 && k0 < n && vb)) rb[i] = zero4;
      w4[i] = *reinterpret_cast<const W4*>(wk + kc);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k0 = V * (tid + Tn * i);
      if (k0 < n) {
        VecN<T, V> oa4, ob4;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const int k = k0 + j;
          const cpx<T> zk = lds[mr_pad(k)], zm = lds[mr_pad(k == 0 ? 0 : n - k)];
          const cpx<T> w = w4[i].v[j];
          const cpx<T> X = cmul(w, zk) + cmulc(zm, w);
          T xa = X.x, xb = X.y;
          if (it > 0) {
            xa = ra[i].v[j] - alpha * xa;
            xb = rb[i].v[j] - alpha * xb;
            // sum_n r^2 = (1 / 2N) sum_k c_k R_k^2, c_0 = 1/2 (SciPy's unnormalised DCT-II)
            const double t = (double)xa * (double)xa + (double)xb * (double)xb;
            sq += k == 0 ? 0.5 * t : t;
          }
          oa4.v[j] = xa;
          ob4.v[j] = xb;
        }
        if (va) *reinterpret_cast<VecN<T, V>*>(r + oa + k0) = oa4;
        if (vb) *reinterpret_cast<VecN<T, V>*>(r + ob + k0) = ob4;
      }
    }
  }
  if (it > 0) {
    const double tot = block_sum(sq, sh);
    if (threadIdx.x == 0) part_norm[blockIdx.x] = tot / (2.0 * n);
  }
}

// rows: z = DCT-III_rows(Z), p = z + beta p_prev -> the ring slot of this iteration
template <class T, int MAXT, int V>
__global__ __launch_bounds__(MAXT) void mr_rowidct_p_kernel(
    const T* __restrict__ Z, const T* __restrict__ pin, T* __restrict__ pout, int n0, const MrDft d, int rs,
    const cpx<T>* __restrict__ W, const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec,
    const cpx<T>* __restrict__ wk, const int* flags, const double* part_rho, int nrho,
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
  // the flags, the partial sums and rho are requested with the spectra; the early exit follows that one round trip
  const int stop = flags[1];
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[32];
  const int n = d.n, Tn = d.pl.T;
  const int tid = threadIdx.x % Tn, f = threadIdx.x / Tn, nf = blockDim.x / Tn;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * rs;
  const int pr = blockIdx.x * nf + f;
  const int xa = 2 * pr, xb = xa + 1;
  const bool va = xa < n0, vb = xb < n0;
  const size_t oa = (size_t)(va ? xa : 0) * n, ob = (size_t)(vb ? xb : 0) * n;
  constexpr int NV = MR_REGS / V;
  VecN<T, V> zero4;
#pragma unroll
  for (int j = 0; j < V; ++j) zero4.v[j] = T(0);
  // the two spectra come in with batched 16-byte loads and are parked in LDS in natural order: bin n - k, which the
  // DCT-III needs next to bin k, would otherwise be a second, misaligned pass over the rows
  {
    VecN<T, V> za[NV], zb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k0 = V * (tid + Tn * i), kc = k0 < n ? k0 : 0;
      za[i] = *reinterpret_cast<const VecN<T, V>*>(Z + oa + kc);
      zb[i] = *reinterpret_cast<const VecN<T, V>*>(Z + ob + kc);
      if (!(k0 < n && va)) za[i] = zero4;
      if (!(k0 < n && vb)) zb[i] = zero4;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k0 = V * (tid + Tn * i);
      if (k0 < n) {
#pragma unroll
        for (int j = 0; j < V; ++j) lds[mr_pad(k0 + j)] = {za[i].v[j], zb[i].v[j]};
      }
    }
  }
  if (stop) return;
  const double rho = block_sum(rho_part, sh);               // (contains the barrier the LDS image needs)
  const bool first = it == 0;                               // first iteration: p = z (pin is uninitialised)
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  {
    cpx<T> pv[MR_REGS];
#pragma unroll
    for (int i = 0; i < MR_REGS; ++i) {
      const int k = tid + Tn * i;
      if (k < n) {
        const cpx<T> X = lds[mr_pad(k)];
        const cpx<T> Xm = k == 0 ? cpx<T>{T(0), T(0)} : lds[mr_pad(n - k)];
        const cpx<T> d = {X.x + Xm.y, X.y - Xm.x};     // X_k - i X_(n-k)
        const cpx<T> v = cmulc(d, wk[k]);              // V_k = conj(w_k) (.) / 2
        pv[i] = {T(0.5) * v.x, T(-0.5) * v.y};         // conj(V_k): IDFT = conj(DFT(conj .))
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MR_REGS; ++i) {
      const int k = tid + Tn * i;
      if (k < n) lds[mr_pad(k)] = pv[i];
    }
  }
  __syncthreads();
  mr_dft<MAXT>(lds, d, W, chirp, bspec, tid);
  const T inv_n = T(1) / T(n);
  {
    VecN<T, V> pa4[NV], pb4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = V * (tid + Tn * i), cc = c0 < n ? c0 : 0;
      pa4[i] = *reinterpret_cast<const VecN<T, V>*>(pin + oa + cc);   // (a ring slot: allocated even before it holds a p)
      pb4[i] = *reinterpret_cast<const VecN<T, V>*>(pin + ob + cc);
      if (!(!first && c0 < n && va)) pa4[i] = zero4;
      if (!(!first && c0 < n && vb)) pb4[i] = zero4;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = V * (tid + Tn * i);
      if (c0 < n) {
        VecN<T, V> oa4, ob4;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const cpx<T> v = lds[mr_pad(mr_slot_of(c0 + j, n))];
          oa4.v[j] = v.x * inv_n + beta * pa4[i].v[j];
          ob4.v[j] = -v.y * inv_n + beta * pb4[i].v[j];
        }
        if (va) *reinterpret_cast<VecN<T, V>*>(pout + oa + c0) = oa4;
        if (vb) *reinterpret_cast<VecN<T, V>*>(pout + ob + c0) = ob4;
      }
    }
  }
}

// columns: Z = DCT-III_cols( DCT-II_cols(R) / eigenvalues ), the stopping test on the update the row kernel has
// just applied, partial rho = <r, z> from the packed spectra (see WgDCT::solve_combine for the Parseval argument)
template <class T, int MAXT, int V>   // (V unused: the columns are read pair by pair)
__global__ __launch_bounds__(MAXT) void mr_colsolve_kernel(
    const T* __restrict__ Zin, T* __restrict__ Z, int n1, const MrDft d, int rs, const cpx<T>* __restrict__ W,
    const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec,
    const cpx<T>* __restrict__ wk, const T* __restrict__ ha, const T* __restrict__ ham, const T* __restrict__ hb,
    int* flags, const double* part_norm, int nnorm, int it, double eps, double* scal, double* part_rho, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Zin += pb * pimg;
    Z += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_norm += pb * PART_N;
    part_rho += pb * PART_N;
  }
  if (flags[1]) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[32];
  const int n = d.n, Tn = d.pl.T, nf = blockDim.x / Tn;
  // column pair fastest in the thread index: neighbouring lanes read neighbouring columns
  const int f = threadIdx.x % nf, tid = threadIdx.x / nf;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * rs;
  // (neighbouring column groups on one XCD: they share the 128-byte lines of every row, see xcd_tile())
  const int ya = (xcd_tile(blockIdx.x, gridDim.x) * nf + f) * 2, yb = ya + 1;
  const bool va = ya < n1, vb = yb < n1;
  struct alignas(2 * sizeof(T)) Pair { T a, b; };
  const bool vec = vb && (n1 & 1) == 0;   // the pair is one aligned 2-element access
  {
    // (all of a thread's loads first, then the LDS stores: one memory latency instead of one per row)
    cpx<T> v[MR_REGS];
#pragma unroll
    for (int i = 0; i < MR_REGS; ++i) {
      const int row = tid + Tn * i;
      const size_t o = (size_t)(row < n ? row : 0) * n1;   // (clamped: the load is unconditional, see the row kernels)
      if (vec) {
        const Pair pz = *reinterpret_cast<const Pair*>(Zin + o + ya);
        v[i] = {pz.a, pz.b};
      } else {
        v[i] = {Zin[o + (va ? ya : 0)], Zin[o + (vb ? yb : 0)]};
        if (!va) v[i].x = T(0);
        if (!vb) v[i].y = T(0);
      }
      if (row >= n) v[i] = {T(0), T(0)};
    }
#pragma unroll
    for (int i = 0; i < MR_REGS; ++i) {
      const int row = tid + Tn * i;
      if (row < n) lds[mr_pad(mr_slot_of(row, n))] = v[i];
    }
  }
  if (it > 0) {
    // the reference's stopping test (phase_unwrap.py:348), evaluated identically by every workgroup
    const double tot = reduce_partials(part_norm, nnorm, sh);
    const double best = scal[10 + ((it - 1) & 1)];
    double stall;
    const bool stop = sqrt(tot) < eps * sqrt(scal[5]) || tot == 0.0 || pcg_breakdown(tot, best, scal[5], sizeof(T) == 4, scal[SC_STALL + ((it - 1) & 1)], &stall, scal[SC_STALL_LIMIT]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[0] = it;
      scal[6] = tot;
      scal[10 + (it & 1)] = tot < best ? tot : best;
      scal[SC_STALL + (it & 1)] = stall;
      if (stop) flags[1] = 1;
    }
    if (stop) return;
  }
  __syncthreads();
  mr_dft<MAXT>(lds, d, W, chirp, bspec, tid);
  // bins k and n - k of the packed spectrum into registers, then (after a barrier) the solve in place
  cpx<T> zk[MR_REGS], zm[MR_REGS];
#pragma unroll
  for (int i = 0; i < MR_REGS; ++i) {
    const int k = tid + Tn * i;
    if (k < n) {
      zk[i] = lds[mr_pad(k)];
      zm[i] = lds[mr_pad(k == 0 ? 0 : n - k)];
    }
  }
  __syncthreads();
  const T inv_n = T(1) / T(n), cn = T(-0.5) * inv_n;   // 1 / (2 (cos + cos - 2)) / n = cn / (ha + hb)
  const T hba = va ? hb[ya] : T(1), hbb = vb ? hb[yb] : T(1);
  const bool first_a = ya == 0;
  double packed = 0.0, corr = 0.0;
#pragma unroll
  for (int i = 0; i < MR_REGS; ++i) {
    const int k = tid + Tn * i;
    if (k < n) {
      const cpx<T> w = wk[k];
      const T h = ha[k], hm = ham[k];
      const cpx<T> qa = {T(0.5) * (zk[i].x + zm[i].x), T(0.5) * (zk[i].y - zm[i].y)};
      const cpx<T> qb = {T(0.5) * (zk[i].y + zm[i].y), T(-0.5) * (zk[i].x - zm[i].x)};
      const cpx<T> ua = cmul(w, qa), ub = cmul(w, qb);
      T sa = cn * fast_recip(h + hba), sb = cn * fast_recip(h + hbb);
      T sam = cn * fast_recip(hm + hba), sbm = cn * fast_recip(hm + hbb);
      if (k == 0) {
        sam = T(0);
        sbm = T(0);
        if (first_a) sa = inv_n;   // the DC bin is divided by 1 (phase_unwrap.py:110-114)
      }
      const cpx<T> pa = cmulc(cpx<T>{sa * ua.x, sam * ua.y}, w);
      const cpx<T> pb = cmulc(cpx<T>{sb * ub.x, sbm * ub.y}, w);
      const cpx<T> xn = {pa.x - pb.y, pa.y + pb.x};
      lds[mr_pad(k)] = {xn.x, -xn.y};   // conjugated for the conj-DFT-conj inverse
      packed += (double)zk[i].x * (double)xn.x + (double)zk[i].y * (double)xn.y;
      if (first_a) corr += (k == 0 ? 0.5 : 1.0) * (double)ua.x * (double)ua.x * (double)sa;
    }
  }
  __syncthreads();
  mr_dft<MAXT>(lds, d, W, chirp, bspec, tid);
  for (int row = tid; row < n; row += Tn) {
    const size_t o = (size_t)row * n1;
    const cpx<T> v = lds[mr_pad(mr_slot_of(row, n))];
    if (vec) {
      *reinterpret_cast<Pair*>(Z + o + ya) = Pair{v.x, -v.y};
    } else {
      if (va) Z[o + ya] = v.x;
      if (vb) Z[o + yb] = -v.y;
    }
  }
  const double tot = block_sum(va ? 0.5 * (packed - corr) : 0.0, sh);
  if (threadIdx.x == 0) part_rho[blockIdx.x] = tot / (double)n1;
}

// dynamic LDS a kernel may ask for (the kernels also hold 256 bytes of static LDS for their reductions)
constexpr int MR_LDS_MAX = 160 * 1024 - 1024;

// transforms per workgroup: as many as `cap` threads allow, but not so many that the grid falls under 256 workgroups
inline int mr_pick_nf(int pairs, int T, int cap, size_t lds_per_transform) {
  int nf = cap / T;
  if (nf > 4) nf = 4;
  if (nf < 1) nf = 1;
  while (nf > 1 && ((pairs + nf - 1) / nf < 256 || nf * lds_per_transform > (size_t)MR_LDS_MAX)) nf /= 2;
  return nf;
}

#define GPA_MR_LAUNCH(KERNEL, VV, threads, ...)                                                                    \
  do {                                                                                                         \
    if ((threads) <= 256) { GPA_MR_LAUNCH1(KERNEL, VV, 256, __VA_ARGS__); }                                    \
    else if ((threads) <= 512) { GPA_MR_LAUNCH1(KERNEL, VV, 512, __VA_ARGS__); }                               \
    else { GPA_MR_LAUNCH1(KERNEL, VV, 1024, __VA_ARGS__); }                                                     \
  } while (0)
#define GPA_MR_LAUNCH1(KERNEL, VV, MAXT, ...)                                                                    \
  do {                                                                                                         \
    auto kern = KERNEL<T, MAXT, VV>;                                                                           \
    static unsigned lds_set = 0;                                                                               \
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), MR_LDS_MAX, lds_set);               \
    if (e != hipSuccess) return e;                                                                             \
    kern<<<dim3(grid, 1, w->nprob), threads, lds, s>>>(__VA_ARGS__);                                                              \
  } while (0)

template <class T>
hipError_t run_mr_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq,
                               double* part_norm, int it, int* nnorm, int init, hipStream_t s) {
  const MrDft& d = w->mr1;
  const MrPlan& pl = d.pl;
  const int rs = mr_lds_elems(pl.n), pairs = (w->n0 + 1) / 2;
  const int nf = mr_pick_nf(pairs, pl.T, 256, (size_t)rs * sizeof(cpx<T>));
  const int grid = (pairs + nf - 1) / nf, threads = nf * pl.T;
  const size_t lds = (size_t)nf * rs * sizeof(cpx<T>);
  *nnorm = grid;
  GPA_PROF("rowdct_fused_kernel", s);
#define GPA_MR_ARGS                                                                                                    \
  (T*)w->r, (const T*)q, w->n0, d, rs, (const cpx<T>*)w->mrW1, (const cpx<T>*)w->chirp1, (const cpx<T>*)w->mrB1,       \
      (const cpx<T>*)w->gwk1, w->flags, part_pq, npq, part_norm, w->scal, it, ring, init, (size_t)w->n0 * w->n1
  if ((w->n1 % 4) == 0) GPA_MR_LAUNCH(mr_rowdct_fused_kernel, 4, threads, GPA_MR_ARGS);
  else GPA_MR_LAUNCH(mr_rowdct_fused_kernel, 1, threads, GPA_MR_ARGS);
#undef GPA_MR_ARGS
  return hipGetLastError();
}

template <class T>
hipError_t run_mr_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                            hipStream_t s) {
  const MrDft& d = w->mr1;
  const MrPlan& pl = d.pl;
  const int rs = mr_lds_elems(pl.n), pairs = (w->n0 + 1) / 2;
  const int nf = mr_pick_nf(pairs, pl.T, 256, (size_t)rs * sizeof(cpx<T>));
  const int grid = (pairs + nf - 1) / nf, threads = nf * pl.T;
  const size_t lds = (size_t)nf * rs * sizeof(cpx<T>);
  GPA_PROF("rowidct_p_kernel", s);
#define GPA_MR_ARGS                                                                                                   \
  (const T*)w->z, (const T*)pin, (T*)pout, w->n0, d, rs, (const cpx<T>*)w->mrW1, (const cpx<T>*)w->chirp1,            \
      (const cpx<T>*)w->mrB1, (const cpx<T>*)w->gwk1, w->flags, part_rho, nrho, w->scal, it, (size_t)w->n0 * w->n1
  if ((w->n1 % 4) == 0) GPA_MR_LAUNCH(mr_rowidct_p_kernel, 4, threads, GPA_MR_ARGS);
  else GPA_MR_LAUNCH(mr_rowidct_p_kernel, 1, threads, GPA_MR_ARGS);
#undef GPA_MR_ARGS
  return hipGetLastError();
}

template <class T>
hipError_t run_mr_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                           double eps, double* part_rho, int* nrho, const void* zin) {
  const MrDft& d = w->mr0;
  const MrPlan& pl = d.pl;
  const int rs = mr_lds_elems(pl.n), pairs = (w->n1 + 1) / 2;
  // f64: 512 threads at most, so that the 16 + 16 bins a thread holds in the solve fit 256 registers
  const int nf = mr_pick_nf(pairs, pl.T, sizeof(T) == 4 ? 1024 : 512, (size_t)rs * sizeof(cpx<T>));
  const int grid = (pairs + nf - 1) / nf, threads = nf * pl.T;
  const size_t lds = (size_t)nf * rs * sizeof(cpx<T>);
  if (nrho) *nrho = grid;
  GPA_PROF("colsolve_kernel", s);
  GPA_MR_LAUNCH(mr_colsolve_kernel, 4, threads, (const T*)(zin ? zin : w->z), (T*)w->z, w->n1, d, rs,
                (const cpx<T>*)w->mrW0, (const cpx<T>*)w->chirp0, (const cpx<T>*)w->mrB0, (const cpx<T>*)w->gwk0, (const T*)w->gha0[compat], (const T*)w->gham0[compat],
                (const T*)w->hb1[compat], w->flags, part_norm, nnorm, it, eps, w->scal, part_rho, (size_t)w->n0 * w->n1);
  return hipGetLastError();
}

}  // namespace
}  // namespace gpa
