// a7 for image sizes that are not powers of two: the Bluestein DCT kernels of the plain scheme (g_*) and, through
// gpa_unwrap_mr.h, the fused iteration on the mixed-radix engine (phase_unwrap.py:95-115, :326-349).
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

// ---------------------------------------------------------------------------
// generic-size DCT kernels (Bluestein).  Same data flow as the power-of-two kernels,
// everything in the natural layout; 4 FFTs of length L >= 2n-1 per column instead of 2 of
// length n, so roughly 4-8x the arithmetic -- the price of accepting any image size.
// ---------------------------------------------------------------------------

// rows: r (n0 x n) -> Z = DCT-II along axis 1; two rows per complex transform
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_rowdct_kernel(
    const T* __restrict__ r, int n0, int n, T* __restrict__ Z, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec, const cpx<T>* __restrict__ wk,
    const int* flags) {
  if (flags[1]) return;
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const int xa = 2 * pr, xb = 2 * pr + 1;
  const bool va = xa < n0, vb = xb < n0;
  const T* ra = r + (size_t)(va ? xa : 0) * n;
  const T* rb = r + (size_t)(vb ? xb : 0) * n;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) {
      const int src = makhoul_src(slot, n);
      x[i] = {va ? ra[src] : T(0), vb ? rb[src] : T(0)};
    } else {
      x[i] = {T(0), T(0)};
    }
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) lds[F::pad(slot)] = x[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = tid + TPF * i;
    if (k < n) {
      const cpx<T> zm = lds[F::pad(k == 0 ? 0 : n - k)];
      const cpx<T> w = wk[k];
      const cpx<T> X = cmul(w, x[i]) + cmulc(zm, w);
      if (va) Z[(size_t)xa * n + k] = X.x;
      if (vb) Z[(size_t)xb * n + k] = X.y;
    }
  }
}

// rows: Z -> z = DCT-III along axis 1 (in place), partial <r, z>
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_rowidct_kernel(
    T* __restrict__ Z, const T* __restrict__ r, int n0, int n, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec, const cpx<T>* __restrict__ wk,
    double* part, const int* flags) {
  if (flags[1]) return;
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[1024];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const int xa = 2 * pr, xb = 2 * pr + 1;
  const bool va = xa < n0, vb = xb < n0;
  T* za = Z + (size_t)(va ? xa : 0) * n;
  T* zb = Z + (size_t)(vb ? xb : 0) * n;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
  const T inv_n = T(1) / T(n);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = tid + TPF * i;
    x[i] = {T(0), T(0)};
    if (k < n) {
      const cpx<T> X = {va ? za[k] : T(0), vb ? zb[k] : T(0)};
      const cpx<T> Xm = k == 0 ? cpx<T>{T(0), T(0)} : cpx<T>{va ? za[n - k] : T(0), vb ? zb[n - k] : T(0)};
      const cpx<T> d = {X.x + Xm.y, X.y - Xm.x};     // X_k - i X_{n-k}
      const cpx<T> v = cmulc(d, wk[k]);              // V_k = conj(w_k) (.) / 2
      x[i] = {T(0.5) * v.x, T(-0.5) * v.y};          // conj(V_k): IDFT = conj(DFT(conj .))
    }
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = tid + TPF * i;
    if (m < n) lds[F::pad(makhoul_src(m, n))] = {x[i].x * inv_n, -x[i].y * inv_n};
  }
  __syncthreads();
  double dot = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = tid + TPF * i;
    if (c < n) {
      const cpx<T> v = lds[F::pad(c)];
      if (va) { za[c] = v.x; dot += (double)r[(size_t)xa * n + c] * (double)v.x; }
      if (vb) { zb[c] = v.y; dot += (double)r[(size_t)xb * n + c] * (double)v.y; }
    }
  }
  const double tot = block_sum(dot, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// columns: DCT-II along axis 0 -> divide by eigenvalues -> DCT-III along axis 0, in place;
// two adjacent columns per complex transform, NF transforms per workgroup
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_colsolve_kernel(
    T* __restrict__ Z, int n, int n1, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ chirp,
    const cpx<T>* __restrict__ bspec, const cpx<T>* __restrict__ wk, const T* __restrict__ ha,
    const T* __restrict__ ham, const T* __restrict__ hb, const int* flags) {
  if (flags[1]) return;
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF, NF = G::NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // column-pair index fastest in the thread index: neighbouring lanes read neighbouring columns
  const int f = threadIdx.x % NF, tid = threadIdx.x / NF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int ya = (blockIdx.x * NF + f) * 2, yb = ya + 1;
  const bool va = ya < n1, vb = yb < n1;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    x[i] = {T(0), T(0)};
    if (slot < n) {
      const size_t row = (size_t)makhoul_src(slot, n) * n1;
      x[i] = {va ? Z[row + ya] : T(0), vb ? Z[row + yb] : T(0)};
    }
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) lds[F::pad(slot)] = x[i];
  }
  __syncthreads();
  const T inv_n = T(1) / T(n);
  const T hba = va ? hb[ya] : T(1), hbb = vb ? hb[yb] : T(1);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = tid + TPF * i;
    if (k < n) {
      const cpx<T> zk = x[i], zm = lds[F::pad(k == 0 ? 0 : n - k)];
      const cpx<T> w = wk[k];
      const T h = ha[k], hm = ham[k];
      const cpx<T> qa = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};
      const cpx<T> qb = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};
      const cpx<T> ua = cmul(w, qa), ub = cmul(w, qb);
      T sa = inv_n / (T(-2) * (h + hba)), sb = inv_n / (T(-2) * (h + hbb));
      T sam = inv_n / (T(-2) * (hm + hba)), sbm = inv_n / (T(-2) * (hm + hbb));
      if (k == 0) {
        sam = T(0);
        sbm = T(0);
        if (ya == 0) sa = inv_n;
      }
      const cpx<T> pa = cmulc(cpx<T>{sa * ua.x, sam * ua.y}, w);
      const cpx<T> pb = cmulc(cpx<T>{sb * ub.x, sbm * ub.y}, w);
      // V'_k of the packed pair, conjugated for the conj-DFT-conj inverse
      x[i] = {pa.x - pb.y, -(pa.y + pb.x)};
    } else {
      x[i] = {T(0), T(0)};
    }
  }
  __syncthreads();
  B::dft(x, lds, tid, n, chirp, bspec, tw);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) {
      const size_t row = (size_t)makhoul_src(slot, n) * n1;
      if (va) Z[row + ya] = x[i].x;
      if (vb) Z[row + yb] = -x[i].y;
    }
  }
}
template <class T, int LG>
hipError_t run_g_rowdct(const Impl* w, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_rowdct_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = (w->n0 + 1) / 2, grid = (npairs + G::NF - 1) / G::NF;
    GPA_PROF("g_rowdct_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>((const T*)w->r, w->n0, w->n1, (T*)w->z, (const cpx<T>*)w->btw1,
                                                 (const cpx<T>*)w->chirp1, (const cpx<T>*)w->bspec1,
                                                 (const cpx<T>*)w->gwk1, w->flags);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_g_rowidct(const Impl* w, int* nparts, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_rowidct_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = (w->n0 + 1) / 2, grid = (npairs + G::NF - 1) / G::NF;
    *nparts = grid;
    GPA_PROF("g_rowidct_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>((T*)w->z, (const T*)w->r, w->n0, w->n1, (const cpx<T>*)w->btw1,
                                                 (const cpx<T>*)w->chirp1, (const cpx<T>*)w->bspec1,
                                                 (const cpx<T>*)w->gwk1, w->part, w->flags);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_g_colsolve(const Impl* w, int compat, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_colsolve_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = (w->n1 + 1) / 2, grid = (npairs + G::NF - 1) / G::NF;
    GPA_PROF("g_colsolve_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>((T*)w->z, w->n0, w->n1, (const cpx<T>*)w->btw0,
                                                 (const cpx<T>*)w->chirp0, (const cpx<T>*)w->bspec0,
                                                 (const cpx<T>*)w->gwk0, (const T*)w->gha0[compat],
                                                 (const T*)w->gham0[compat], (const T*)w->hb1[compat], w->flags);
    return hipGetLastError();
  }
}
}  // namespace
hipError_t dispatch_g_rowdct(const Impl* w, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_g_rowdct<float, LG>(w, s) : run_g_rowdct<double, LG>(w, s);
  switch (w->lgb1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
hipError_t dispatch_g_rowidct(const Impl* w, int* nparts, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_g_rowidct<float, LG>(w, nparts, s) : run_g_rowidct<double, LG>(w, nparts, s);
  switch (w->lgb1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
hipError_t dispatch_g_colsolve(const Impl* w, int compat, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_g_colsolve<float, LG>(w, compat, s) : run_g_colsolve<double, LG>(w, compat, s);
  switch (w->lgb0) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
}  // namespace gpa
#include "gpa_unwrap_mr.h"
namespace gpa {
hipError_t mr_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm,
                           int it, int* nnorm, int init, hipStream_t s) {
  return w->dtype == 0 ? run_mr_rowdct_fused<float>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                       : run_mr_rowdct_fused<double>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
}
hipError_t mr_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                        hipStream_t s) {
  return w->dtype == 0 ? run_mr_rowidct_p<float>(w, pin, pout, part_rho, nrho, it, s)
                       : run_mr_rowidct_p<double>(w, pin, pout, part_rho, nrho, it, s);
}
hipError_t mr_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it, double eps,
                       double* part_rho, int* nrho, const void* zin) {
  return w->dtype == 0 ? run_mr_colsolve<float>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
                       : run_mr_colsolve<double>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
}

}  // namespace gpa
