// a5 + a6: lock-ins -> wrapped phase differences -> per-pixel weighted 2x2 solve.
//
// Streaming kernel, one thread per pixel.  The reference solves a P x 2 weighted
// least-squares problem per pixel with LAPACK (myweighed_lstsq,
// geometric_phase_analysis.py:97-113); with K = 2 pi kvecs this is the 2x2 normal
// system (K^T W^2 K) x = K^T W^2 b, solved here in registers (min-norm answer at
// rank-deficient pixels, 0 where every weight is 0).  MFMA has nothing to offer
// a 3x2 problem per pixel; the kernel is HBM-bound.
#include "gpa_internal.h"

namespace gpa {

template <class T> struct Consts;
template <> struct Consts<float> {
  static constexpr float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
};
template <> struct Consts<double> {
  static constexpr double pi = 3.14159265358979323846, two_pi = 6.28318530717958647692;
};

// (x + pi) mod 2 pi - pi with floored mod: [-pi, pi), +pi -> -pi (mathtools.py:72-75)
template <class T>
__device__ __forceinline__ T wrap_to_pi(T x) {
  const T t = x + Consts<T>::pi;
  return t - Consts<T>::two_pi * floor(t / Consts<T>::two_pi) - Consts<T>::pi;
}

template <class T>
__device__ __forceinline__ void solve2(T a00, T a01, T a11, T r0, T r1, T& x0, T& x1) {
  const T det = a00 * a11 - a01 * a01, tr = a00 + a11;
  const T tiny = sizeof(T) == 4 ? T(1e-12) : T(1e-28);
  if (det > tiny * tr * tr) {
    const T inv = T(1) / det;
    x0 = (a11 * r0 - a01 * r1) * inv;
    x1 = (a00 * r1 - a01 * r0) * inv;
  } else if (tr > T(0)) {
    x0 = r0 / tr;
    x1 = r1 / tr;
  } else {
    x0 = T(0);
    x1 = T(0);
  }
}

constexpr int MAXP = 8;

template <class T>
__global__ __launch_bounds__(256) void reconstruct_kernel(const cpx<T>* __restrict__ lockin,
                                                         const double* __restrict__ kmat, int P,
                                                         int n0, int n1, int border,
                                                         T* __restrict__ dudx, T* __restrict__ dudy,
                                                         T* __restrict__ wnorm) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  const int x = blockIdx.y;
  if (y >= n1) return;
  const size_t npx = (size_t)n0 * n1;
  const size_t o = (size_t)x * n1 + y;
  const bool inside = x >= border && x < n0 - border && y >= border && y < n1 - border;
  const T mfac = (inside ? T(1) : T(0)) + T(1e-6);
  const bool has_r = y + 1 < n1, has_d = x + 1 < n0;

  T w[MAXP], bx[MAXP], by[MAXP];
  T wmax = T(0), wsq = T(0);
  for (int p = 0; p < P; ++p) {
    const cpx<T> c = lockin[p * npx + o];
    const T ph = atan2(c.y, c.x);
    const T amp = sqrt(c.x * c.x + c.y * c.y);
    w[p] = amp * mfac;
    wsq += w[p] * w[p];
    wmax = w[p] > wmax ? w[p] : wmax;
    bx[p] = T(0);
    by[p] = T(0);
    if (has_r) {
      const cpx<T> r = lockin[p * npx + o + 1];
      bx[p] = wrap_to_pi(atan2(r.y, r.x) - ph);
    }
    if (has_d) {
      const cpx<T> d = lockin[p * npx + o + n1];
      by[p] = wrap_to_pi(atan2(d.y, d.x) - ph);
    }
  }
  if (wnorm) wnorm[o] = sqrt(wsq);
  // normalise the weights per pixel: the solution is scale invariant and this keeps
  // w^4 away from the f32 underflow range outside the mask (weights ~ 1e-6 |lockin|)
  const T ws = wmax > T(0) ? T(1) / wmax : T(0);
  T a00 = 0, a01 = 0, a11 = 0, rx0 = 0, rx1 = 0, ry0 = 0, ry1 = 0;
  for (int p = 0; p < P; ++p) {
    const T k0 = (T)kmat[2 * p], k1 = (T)kmat[2 * p + 1];
    const T wn = w[p] * ws, ww = wn * wn;
    a00 += ww * k0 * k0;
    a01 += ww * k0 * k1;
    a11 += ww * k1 * k1;
    rx0 += ww * k0 * bx[p];
    rx1 += ww * k1 * bx[p];
    ry0 += ww * k0 * by[p];
    ry1 += ww * k1 * by[p];
  }
  if (has_r) {
    T s0, s1;
    solve2(a00, a01, a11, rx0, rx1, s0, s1);
    const size_t ox = (size_t)x * (n1 - 1) + y, plane = (size_t)n0 * (n1 - 1);
    dudx[ox] = s0;
    dudx[plane + ox] = s1;
  }
  if (has_d) {
    T s0, s1;
    solve2(a00, a01, a11, ry0, ry1, s0, s1);
    const size_t plane = (size_t)(n0 - 1) * n1;
    dudy[o] = s0;
    dudy[plane + o] = s1;
  }
}

// per-pixel weighted least squares on given right-hand sides b (P x n0 x n1), the weighted
// branch of reconstruct_u_inv (geometric_phase_analysis.py:188 -> myweighed_lstsq :97-113)
template <class T>
__global__ __launch_bounds__(256) void wlstsq_kernel(const T* __restrict__ b, const T* __restrict__ w,
                                                    const double* __restrict__ kmat, int P, size_t npx,
                                                    T* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= npx) return;
  T wv[MAXP], wmax = T(0);
  for (int p = 0; p < P; ++p) { wv[p] = w[p * npx + o]; const T a = wv[p] < T(0) ? -wv[p] : wv[p]; wmax = a > wmax ? a : wmax; }
  const T ws = wmax > T(0) ? T(1) / wmax : T(0);
  T a00 = 0, a01 = 0, a11 = 0, r0 = 0, r1 = 0;
  for (int p = 0; p < P; ++p) {
    const T k0 = (T)kmat[2 * p], k1 = (T)kmat[2 * p + 1], wn = wv[p] * ws, ww = wn * wn, bv = b[p * npx + o];
    a00 += ww * k0 * k0; a01 += ww * k0 * k1; a11 += ww * k1 * k1;
    r0 += ww * k0 * bv; r1 += ww * k1 * bv;
  }
  T s0, s1;
  solve2(a00, a01, a11, r0, r1, s0, s1);
  out[o] = s0;
  out[npx + o] = s1;
}

hipError_t launch_wlstsq(int dtype, const void* b, const void* w, const double* kmat, int P, size_t npx, void* out,
                         hipStream_t s) {
  if (P > MAXP) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)((npx + 255) / 256);
  if (dtype == 0)
    wlstsq_kernel<float><<<grid, 256, 0, s>>>((const float*)b, (const float*)w, kmat, P, npx, (float*)out);
  else
    wlstsq_kernel<double><<<grid, 256, 0, s>>>((const double*)b, (const double*)w, kmat, P, npx, (double*)out);
  return hipGetLastError();
}

hipError_t launch_reconstruct(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1,
                              int border, void* dudx, void* dudy, void* wnorm, hipStream_t s) {
  if (P > MAXP) return hipErrorInvalidValue;
  dim3 grid((n1 + 255) / 256, n0);
  if (dtype == 0)
    reconstruct_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)lockin, kmat, P, n0, n1, border,
                                                   (float*)dudx, (float*)dudy, (float*)wnorm);
  else
    reconstruct_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)lockin, kmat, P, n0, n1, border,
                                                    (double*)dudx, (double*)dudy, (double*)wnorm);
  return hipGetLastError();
}

}  // namespace gpa
