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
constexpr int REC_ROWS = 16;   // rows per workgroup band

// One thread per column sliding down a band of REC_ROWS rows: the phase of every lock-in
// sample is evaluated once (atan2 is the expensive part) -- the right neighbour's phase comes
// from the next lane, the lower neighbour's from the next row, which becomes the current row
// of the following step.
template <class T, int P>
__global__ __launch_bounds__(256) void reconstruct_kernel(const cpx<T>* __restrict__ lockin,
                                                         const double* __restrict__ kmat, int n0, int n1,
                                                         int border, T* __restrict__ dudx, T* __restrict__ dudy,
                                                         T* __restrict__ wnorm) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  const int x0 = blockIdx.y * REC_ROWS;
  const int x1 = x0 + REC_ROWS < n0 ? x0 + REC_ROWS : n0;
  const int lane = threadIdx.x & 63;
  const bool act = y < n1;
  const int yc = act ? y : n1 - 1;
  const size_t npx = (size_t)n0 * n1;
  const bool has_r = act && yc + 1 < n1;
  T k0[P], k1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { k0[p] = (T)kmat[2 * p]; k1[p] = (T)kmat[2 * p + 1]; }
  T phc[P], ampc[P], phn[P], ampn[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const cpx<T> c = lockin[p * npx + (size_t)x0 * n1 + yc];
    phc[p] = atan2(c.y, c.x);
    ampc[p] = sqrt(c.x * c.x + c.y * c.y);
    phn[p] = T(0);
    ampn[p] = T(0);
  }
  for (int x = x0; x < x1; ++x) {
    const size_t o = (size_t)x * n1 + yc;
    const bool has_d = x + 1 < n0;
    if (has_d) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const cpx<T> c = lockin[p * npx + o + n1];
        phn[p] = atan2(c.y, c.x);
        ampn[p] = sqrt(c.x * c.x + c.y * c.y);
      }
    }
    const bool inside = x >= border && x < n0 - border && yc >= border && yc < n1 - border;
    const T mfac = (inside ? T(1) : T(0)) + T(1e-6);
    T w[P], bx[P], by[P];
    T wmax = T(0), wsq = T(0);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      T phr = __shfl_down(phc[p], 1);
      if (lane == 63 && has_r) {
        const cpx<T> c = lockin[p * npx + o + 1];
        phr = atan2(c.y, c.x);
      }
      w[p] = ampc[p] * mfac;
      wsq += w[p] * w[p];
      wmax = w[p] > wmax ? w[p] : wmax;
      bx[p] = has_r ? wrap_to_pi(phr - phc[p]) : T(0);
      by[p] = has_d ? wrap_to_pi(phn[p] - phc[p]) : T(0);
    }
    if (act) {
      if (wnorm) wnorm[o] = sqrt(wsq);
      // normalise the weights per pixel: the solution is scale invariant and this keeps
      // w^4 away from the f32 underflow range outside the mask (weights ~ 1e-6 |lockin|)
      const T ws = wmax > T(0) ? T(1) / wmax : T(0);
      T a00 = 0, a01 = 0, a11 = 0, rx0 = 0, rx1 = 0, ry0 = 0, ry1 = 0;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const T wn = w[p] * ws, ww = wn * wn;
        a00 += ww * k0[p] * k0[p];
        a01 += ww * k0[p] * k1[p];
        a11 += ww * k1[p] * k1[p];
        rx0 += ww * k0[p] * bx[p];
        rx1 += ww * k1[p] * bx[p];
        ry0 += ww * k0[p] * by[p];
        ry1 += ww * k1[p] * by[p];
      }
      if (has_r) {
        T s0, s1;
        solve2(a00, a01, a11, rx0, rx1, s0, s1);
        const size_t ox = (size_t)x * (n1 - 1) + yc, plane = (size_t)n0 * (n1 - 1);
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
#pragma unroll
    for (int p = 0; p < P; ++p) { phc[p] = phn[p]; ampc[p] = ampn[p]; }
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

template <class T>
static hipError_t launch_reconstruct_t(const void* lockin, const double* kmat, int P, int n0, int n1, int border,
                                       void* dudx, void* dudy, void* wnorm, hipStream_t s) {
  dim3 grid((n1 + 255) / 256, (n0 + REC_ROWS - 1) / REC_ROWS);
#define REC_CASE(PP)                                                                                          \
  case PP:                                                                                                    \
    reconstruct_kernel<T, PP><<<grid, 256, 0, s>>>((const cpx<T>*)lockin, kmat, n0, n1, border, (T*)dudx, (T*)dudy, \
                                                   (T*)wnorm);                                                \
    break;
  switch (P) {
    REC_CASE(2) REC_CASE(3) REC_CASE(4) REC_CASE(5) REC_CASE(6) REC_CASE(7) REC_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef REC_CASE
  return hipGetLastError();
}

hipError_t launch_reconstruct(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1,
                              int border, void* dudx, void* dudy, void* wnorm, hipStream_t s) {
  if (P > MAXP || P < 2) return hipErrorInvalidValue;
  return dtype == 0 ? launch_reconstruct_t<float>(lockin, kmat, P, n0, n1, border, dudx, dudy, wnorm, s)
                    : launch_reconstruct_t<double>(lockin, kmat, P, n0, n1, border, dudx, dudy, wnorm, s);
}

}  // namespace gpa
