// a7, row kernels for LONG rows of the fused PCG iteration: one row per complex transform of half the row length
// (split off gpa_unwrap_rows.hip in round 5; phase_unwrap.py:84-103, :326-349).
// (compiled with -ffp-contract=off: pygpa_amd/build.py, EXTRA_FLAGS, says why)
#include "gpa_unwrap_rowgeom.h"

namespace gpa {
namespace {

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

// four reals at the alignment of ONE (global memory takes 16-byte accesses at 4-byte alignment): the partner bins N - k
template <class T> struct alignas(sizeof(T)) Vec4U {
  T v[4];
  __device__ __forceinline__ operator Vec4<T>() const { Vec4<T> r; r.v[0] = v[0]; r.v[1] = v[1]; r.v[2] = v[2]; r.v[3] = v[3]; return r; }
  __device__ __forceinline__ Vec4U& operator=(const Vec4<T>& o) { v[0] = o.v[0]; v[1] = o.v[1]; v[2] = o.v[2]; v[3] = o.v[3]; return *this; }
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
  // The post-processing owns its bins in BLOCKS OF FOUR: thread t, block v holds k = 4 (t + TPF v) + e, e = 0 .. 3, and their
  // partners N - k -- so the kept spectrum comes in and goes out as 16-byte accesses (the partners' vector starts at N - k0 - 3:
  // 4-byte aligned, which global memory takes) and the two tables as 32 bytes per block.  (Rounds 3-5 owned k = t + TPF i:
  // 64 dword accesses per thread and row where there are 16 vectors now; the memory pipeline issued four times the
  // instructions for the same bytes.)  Bin 0's partner is bin N/2, stored at HN: block 0 of thread 0 goes element by element.
  constexpr int NB = E / 4;
  Vec4<T> rlo[NB], rhi[NB];     // rhi[v].v[3 - e] = R[N - k0 - e]
  auto load_kept = [&]() {
#pragma unroll
    for (int v = 0; v < NB; ++v) {
      const int k0 = 4 * (tid + TPF * v);
      rlo[v] = *reinterpret_cast<const Vec4<T>*>(r + o + k0);
      if (k0 == 0) {
        rhi[v].v[3] = r[o + HN];
        rhi[v].v[2] = r[o + N - 1];
        rhi[v].v[1] = r[o + N - 2];
        rhi[v].v[0] = r[o + N - 3];
      } else {
        rhi[v] = *reinterpret_cast<const Vec4U<T>*>(r + o + N - k0 - 3);
      }
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
  for (int v = 0; v < NB; ++v) {
    const int k0 = 4 * (tid + TPF * v);
    struct alignas(16) C2 { cpx<T> a, b; };
    const C2 tn01 = *reinterpret_cast<const C2*>(twn + k0), tn23 = *reinterpret_cast<const C2*>(twn + k0 + 2);
    const C2 wk01 = *reinterpret_cast<const C2*>(wk + k0), wk23 = *reinterpret_cast<const C2*>(wk + k0 + 2);
    const cpx<T> tn[4] = {tn01.a, tn01.b, tn23.a, tn23.b}, wv[4] = {wk01.a, wk01.b, wk23.a, wk23.b};
    Vec4<T> olo, ohi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      const cpx<T> zk = lds[F::pad(k)], zm = lds[F::pad((HN - k) & (HN - 1))];
      T xlo, xhi;
      if (k == 0) {
        xlo = T(2) * (zk.x + zk.y);
        xhi = T(1.41421356237309504880) * (zk.x - zk.y);
      } else {
        const cpx<T> ve = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};   // (T_k + conj T_m) / 2
        const cpx<T> vo = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};  // -i (T_k - conj T_m) / 2
        const cpx<T> V = ve + cmul(tn[e], vo);
        const cpx<T> U = cmul(wv[e], V);
        xlo = T(2) * U.x;
        xhi = T(-2) * U.y;
      }
      if (it > 0) {
        xlo = rlo[v].v[e] - alpha * xlo;
        xhi = rhi[v].v[3 - e] - alpha * xhi;
        // sum_n r^2 = (1 / 2N) sum_k c_k R_k^2, c_0 = 1/2 (SciPy's unnormalised DCT-II)
        sq += (k == 0 ? 0.5 : 1.0) * (double)xlo * (double)xlo + (double)xhi * (double)xhi;
      }
      olo.v[e] = xlo;
      ohi.v[3 - e] = xhi;
    }
    *reinterpret_cast<Vec4<T>*>(r + o + k0) = olo;
    if (k0 == 0) {
      r[o + HN] = ohi.v[3];
      r[o + N - 1] = ohi.v[2];
      r[o + N - 2] = ohi.v[1];
      r[o + N - 3] = ohi.v[0];
    } else {
      *reinterpret_cast<Vec4U<T>*>(r + o + N - k0 - 3) = ohi;
    }
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

}  // namespace

// dispatch for gpa_unwrap_rows.hip: hipErrorInvalidValue where no half-length instantiation exists
hipError_t rowhalf_rowdct(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm, int it,
                          int* nnorm, int init, hipStream_t s) {
  if (w->lg1 == 12) return w->dtype == 0 ? run_rowdct_half<float, 12>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                                          : run_rowdct_half<double, 12>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  if (w->lg1 == 13) return w->dtype == 0 ? run_rowdct_half<float, 13>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                                          : run_rowdct_half<double, 13>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  if (w->lg1 == 14 && w->dtype == 0) return run_rowdct_half<float, 14>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  return hipErrorInvalidValue;
}
hipError_t rowhalf_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it, hipStream_t s) {
  if (w->lg1 == 12) return w->dtype == 0 ? run_rowidct_p_half<float, 12>(w, pin, pout, part_rho, nrho, it, s)
                                          : run_rowidct_p_half<double, 12>(w, pin, pout, part_rho, nrho, it, s);
  if (w->lg1 == 13) return w->dtype == 0 ? run_rowidct_p_half<float, 13>(w, pin, pout, part_rho, nrho, it, s)
                                          : run_rowidct_p_half<double, 13>(w, pin, pout, part_rho, nrho, it, s);
  if (w->lg1 == 14 && w->dtype == 0) return run_rowidct_p_half<float, 14>(w, pin, pout, part_rho, nrho, it, s);
  return hipErrorInvalidValue;
}
bool rowhalf_offered(const Impl* w) { return (w->lg1 == 12 || w->lg1 == 13 || (w->lg1 == 14 && w->dtype == 0)) && w->tw1h != nullptr; }

}  // namespace gpa
