// a7, stencil + forward row transform of the fused PCG iteration in one launch (split off gpa_unwrap_rows.hip in round 5;
// phase_unwrap.py:118-132, :84-92).
#include "gpa_unwrap_rowgeom.h"

namespace gpa {
namespace {

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
// the value the neighbouring lane holds (lane - 1 / lane + 1 of the wavefront; lanes 0 / 63 get 0 and take theirs from
// memory): one DPP move per dword instead of the ds_bpermute of __shfl_up / __shfl_down (an LDS-crossbar round trip and
// its address arithmetic, 16 times per row pair)
#ifndef GPA_PQ_PLAIN
__device__ __forceinline__ float from_lane_below(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ double from_lane_below(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_lane_above(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// min of two weights: one v_min instead of compare + select (the same value for everything but NaN)
__device__ __forceinline__ float wmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double wmin(double a, double b) { return fmin(a, b); }
#else
template <class T> __device__ __forceinline__ T from_lane_below(T x) { return __shfl_up(x, 1); }
template <class T> __device__ __forceinline__ T from_lane_above(T x) { return __shfl_down(x, 1); }
template <class T> __device__ __forceinline__ T wmin(T a, T b) { return a < b ? a : b; }
#endif
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
      T pla = from_lane_below(P.a[v].v[3]), pra = from_lane_above(P.a[v].v[0]), plb = from_lane_below(P.b[v].v[3]), prb = from_lane_above(P.b[v].v[0]);
      T wla = from_lane_below(W.a[v].v[3]), wra = from_lane_above(W.a[v].v[0]), wlb = from_lane_below(W.b[v].v[3]), wrb = from_lane_above(W.b[v].v[0]);
      if (lane == 0) { pla = EP.la[v]; plb = EP.lb[v]; wla = EW.la[v]; wlb = EW.lb[v]; }
      if (lane == 63) { pra = EP.ra[v]; prb = EP.rb[v]; wra = EW.ra[v]; wrb = EW.rb[v]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // q = sum over the 4 edges of min(w^2, w_nb^2) * (p_nb - p)   (phase_unwrap.py:118-132), row a then row b
        T qa, qb;
        {
          const T c = P.a[v].v[j], wj = W.a[v].v[j];
          T acc = T(0);
          if (j < 3) { const T wn = W.a[v].v[j + 1]; acc += wmin(wn, wj) * (P.a[v].v[j + 1] - c); }
          else if (hasr) acc += wmin(wra, wj) * (pra - c);
          if (j > 0) { const T wn = W.a[v].v[j - 1]; acc += wmin(wn, wj) * (P.a[v].v[j - 1] - c); }
          else if (hasl) acc += wmin(wla, wj) * (pla - c);
          { const T wn = W.b[v].v[j]; acc += wmin(wn, wj) * (P.b[v].v[j] - c); }
          if (up) { const T wn = W.u[v].v[j]; acc += wmin(wn, wj) * (P.u[v].v[j] - c); }
          qa = acc;
          pq += (double)c * (double)acc;
        }
        {
          const T c = P.b[v].v[j], wj = W.b[v].v[j];
          T acc = T(0);
          if (j < 3) { const T wn = W.b[v].v[j + 1]; acc += wmin(wn, wj) * (P.b[v].v[j + 1] - c); }
          else if (hasr) acc += wmin(wrb, wj) * (prb - c);
          if (j > 0) { const T wn = W.b[v].v[j - 1]; acc += wmin(wn, wj) * (P.b[v].v[j - 1] - c); }
          else if (hasl) acc += wmin(wlb, wj) * (plb - c);
          if (dn) { const T wn = W.d[v].v[j]; acc += wmin(wn, wj) * (P.d[v].v[j] - c); }
          { const T wn = W.a[v].v[j]; acc += wmin(wn, wj) * (P.a[v].v[j] - c); }
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
}  // namespace

bool pow2_pqdct_offered(const Impl* w) { return !w->generic && (w->lg1 == 11 || w->lg1 == 12) && (w->n0 % 2) == 0; }
hipError_t pow2_pqdct(const Impl* w, const void* p, const void* weight, double* part_pq, int* npq, hipStream_t s) {
  if (w->lg1 == 11) return w->dtype == 0 ? run_pqdct<float, 11>(w, p, weight, part_pq, npq, s) : run_pqdct<double, 11>(w, p, weight, part_pq, npq, s);
  if (w->lg1 == 12) return w->dtype == 0 ? run_pqdct<float, 12>(w, p, weight, part_pq, npq, s) : run_pqdct<double, 12>(w, p, weight, part_pq, npq, s);
  return hipErrorInvalidValue;
}
}  // namespace gpa
