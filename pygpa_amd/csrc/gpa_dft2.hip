// Arbitrary-size 2-D DFT (Bluestein on the register FFT engine) and the kernels of a9 -- Moisan's periodic + smooth
// decomposition, `moisan2011.per` (call site geometry_phase_analysis.py:429) -- that run on it.  Tables (chirp, the
// transformed chirp in the spectral layout) are built by blue_axis_create() in gpa_unwrap.hip beside the other host-side
// table code; entry points: gpa_per_dft / gpa_per / gpa_find_peaks / gpa_gaussian_deconvolve (gpa_api.hip).
#include <hip/hip_runtime.h>

#include "gpa_dct.h"
#include "gpa_internal.h"
#include "gpa_unwrap.h"

namespace gpa {
namespace {
// ---------------------------------------------------------------------------
// plain forward DFTs of arbitrary length (a9, smooth + periodic decomposition)
// ---------------------------------------------------------------------------
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
    cpx<T>* __restrict__ Z, int n, int n1, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ chirp,
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
    x[i] = (valid && slot < n) ? Z[(size_t)slot * n1 + col] : cpx<T>{T(0), T(0)};
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) Z[(size_t)slot * n1 + col] = x[i];
  }
}

// a9.  v (the border-jump image) is non-zero on the four borders only, so its 2-D DFT
// factorises:  v^[q,r] = D0[r] (1 - e^{2 pi i q/n0}) + D1[q] (1 - e^{2 pi i r/n1}),
// D0 = DFT(u[n0-1,:] - u[0,:]), D1 = DFT(u[:,n1-1] - u[:,0]): two 1-D DFTs instead of a
// second 2-D one, and no cancellation against the (much larger) image spectrum.
template <class T>
__global__ __launch_bounds__(256) void per_pack_kernel(const T* __restrict__ u, int n0, int n1, cpx<T>* __restrict__ Z,
                                                      cpx<T>* __restrict__ d0, cpx<T>* __restrict__ d1) {
  const int y = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y;
  if (y >= n1) return;
  Z[(size_t)x * n1 + y] = {u[(size_t)x * n1 + y], T(0)};
  if (x == 0) d0[y] = {u[(size_t)(n0 - 1) * n1 + y] - u[y], T(0)};
  if (y == 0) d1[x] = {u[(size_t)x * n1 + n1 - 1] - u[(size_t)x * n1], T(0)};
}

template <class T>
__global__ __launch_bounds__(256) void per_combine_kernel(const cpx<T>* __restrict__ Uh, const cpx<T>* __restrict__ D0,
                                                         const cpx<T>* __restrict__ D1, int n0, int n1,
                                                         cpx<T>* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
  if (r >= n1) return;
  const cpx<T> U = Uh[(size_t)q * n1 + r];
  cpx<T> S = {T(0), T(0)};
  if (q != 0 || r != 0) {
    double sq, cq, sr, cr;
    sincospi(2.0 * (double)q / n0, &sq, &cq);
    sincospi(2.0 * (double)r / n1, &sr, &cr);
    const cpx<T> fq = {(T)(1.0 - cq), (T)(-sq)}, fr = {(T)(1.0 - cr), (T)(-sr)};
    const cpx<T> V = cmul(D0[r], fq) + cmul(D1[q], fr);
    // 2 cos a + 2 cos b - 4 = -4 (sin^2(a/2) + sin^2(b/2))
    const double sa = sinpi((double)q / n0), sb = sinpi((double)r / n1);
    const T inv = (T)(1.0 / (-4.0 * (sa * sa + sb * sb)));
    S = {V.x * inv, V.y * inv};
  }
  out[(size_t)q * n1 + r] = {U.x - S.x, U.y - S.y};
}

}  // namespace

namespace {
template <class T, int LG>
hipError_t run_rowdft(const BlueAxis& a, int n0, void* Z, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_rowdft_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    kern<<<(n0 + G::NF - 1) / G::NF, G::THREADS, G::LDS_BYTES, s>>>((cpx<T>*)Z, n0, a.n, (const cpx<T>*)a.tw,
                                                                     (const cpx<T>*)a.chirp, (const cpx<T>*)a.bspec);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_coldft(const BlueAxis& a, int n1, void* Z, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_coldft_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    kern<<<(n1 + G::NF - 1) / G::NF, G::THREADS, G::LDS_BYTES, s>>>((cpx<T>*)Z, a.n, n1, (const cpx<T>*)a.tw,
                                                                     (const cpx<T>*)a.chirp, (const cpx<T>*)a.bspec);
    return hipGetLastError();
  }
}
}  // namespace

hipError_t dft2_inplace(int dtype, const BlueAxis& a0, const BlueAxis& a1, void* Z, hipStream_t s) {
  hipError_t e = hipErrorInvalidValue;
#define CASE(LG) case LG: e = dtype == 0 ? run_rowdft<float, LG>(a1, a0.n, Z, s) : run_rowdft<double, LG>(a1, a0.n, Z, s); break;
  switch (a1.lg) { GPA_FOR_LG(CASE) }
#undef CASE
  if (e != hipSuccess) return e;
  e = hipErrorInvalidValue;
#define CASE(LG) case LG: e = dtype == 0 ? run_coldft<float, LG>(a0, a1.n, Z, s) : run_coldft<double, LG>(a0, a1.n, Z, s); break;
  switch (a0.lg) { GPA_FOR_LG(CASE) }
#undef CASE
  return e;
}

// forward DFT of `rows` contiguous complex rows of length a.n, in place
hipError_t dft_rows_inplace(int dtype, const BlueAxis& a, int rows, void* Z, hipStream_t s) {
  hipError_t e = hipErrorInvalidValue;
#define CASE(LG) case LG: e = dtype == 0 ? run_rowdft<float, LG>(a, rows, Z, s) : run_rowdft<double, LG>(a, rows, Z, s); break;
  switch (a.lg) { GPA_FOR_LG(CASE) }
#undef CASE
  return e;
}

hipError_t per_pack(int dtype, const void* image, int n0, int n1, void* Z, void* d0, void* d1, hipStream_t s) {
  dim3 grid((n1 + 255) / 256, n0);
  if (dtype == 0)
    per_pack_kernel<float><<<grid, 256, 0, s>>>((const float*)image, n0, n1, (cpx<float>*)Z, (cpx<float>*)d0, (cpx<float>*)d1);
  else
    per_pack_kernel<double><<<grid, 256, 0, s>>>((const double*)image, n0, n1, (cpx<double>*)Z, (cpx<double>*)d0, (cpx<double>*)d1);
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
hipError_t per_components(int dtype, const BlueAxis& a0, const BlueAxis& a1, void* Phat_destroyed, const void* image,
                          void* p_out, void* s_out, hipStream_t s) {
  const size_t n = (size_t)a0.n * a1.n;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == 0) per_conj_kernel<float><<<grid, 256, 0, s>>>((cpx<float>*)Phat_destroyed, n);
  else per_conj_kernel<double><<<grid, 256, 0, s>>>((cpx<double>*)Phat_destroyed, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = dft2_inplace(dtype, a0, a1, Phat_destroyed, s);
  if (e != hipSuccess) return e;
  const double scale = 1.0 / (double)n;
  if (dtype == 0)
    per_real_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)Phat_destroyed, (const float*)image, scale, (float*)p_out, (float*)s_out, n);
  else
    per_real_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)Phat_destroyed, (const double*)image, scale, (double*)p_out, (double*)s_out, n);
  return hipGetLastError();
}
hipError_t per_combine(int dtype, const void* Uhat, const void* D0, const void* D1, int n0, int n1, void* out,
                       hipStream_t s) {
  dim3 grid((n1 + 255) / 256, n0);
  if (dtype == 0)
    per_combine_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)Uhat, (const cpx<float>*)D0, (const cpx<float>*)D1, n0, n1, (cpx<float>*)out);
  else
    per_combine_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)Uhat, (const cpx<double>*)D0, (const cpx<double>*)D1, n0, n1, (cpx<double>*)out);
  return hipGetLastError();
}

}  // namespace gpa
