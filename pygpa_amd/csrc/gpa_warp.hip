// f-1: Lawler-Fujita undistortion -- invert_u_overlap / undistort_image
// (geometric_phase_analysis.py:262-300, :935-974), i.e. scipy.ndimage.map_coordinates
// (order 3) restated for the GPU.  SciPy's conventions, pinned numerically to 4e-15 against
// SciPy itself before this was written:
//   prefilter   : cubic B-spline coefficients = two-sided exponential filter
//                 h_k = (-6 z / (1 - z^2)) z^|k|, z = sqrt(3) - 2, along both axes;
//                 evaluated here as a 2*KT+1 tap FIR (|z|^KT < 1e-18) instead of SciPy's
//                 causal/anticausal recursion: same numbers, but every output is independent.
//   mode nearest : pad the image by 12 edge-replicated samples, filter with HALF-sample
//                  symmetric extension, interpolate with the coordinate left unclamped and
//                  the 4 tap indices clamped to the padded array.
//   mode constant: no padding, WHOLE-sample symmetric (mirror) extension for filter and taps,
//                  cval where the coordinate leaves [0, n-1].
// The fixed-point inversion u_it(r) <- u(r + u_it(r)) is independent per pixel, so all its
// rounds run inside one kernel launch.
#include <math.h>

#include "gpa_internal.h"

namespace gpa {

namespace {

constexpr int KT = 32;       // taps on each side of the prefilter
constexpr int NPAD = 12;     // SciPy's pre-padding for mode='nearest'

enum Ext { EXT_REFLECT = 0, EXT_MIRROR = 1 };

__device__ __forceinline__ int ext_index(int i, int n, int ext) {
  if (n == 1) return 0;
  if (ext == EXT_REFLECT) {          // half-sample symmetric: -1 -> 0, n -> n-1
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
  }
  const int p = 2 * n - 2;           // whole-sample symmetric: -1 -> 1, n -> n-2
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}

template <class T>
__global__ __launch_bounds__(256) void pad_edge_kernel(const T* __restrict__ in, int n0, int n1, int npad, T scale,
                                                      T* __restrict__ out) {
  const int m1 = n1 + 2 * npad;
  const int y = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y;
  if (y >= m1) return;
  int sx = x - npad, sy = y - npad;
  sx = sx < 0 ? 0 : (sx >= n0 ? n0 - 1 : sx);
  sy = sy < 0 ? 0 : (sy >= n1 ? n1 - 1 : sy);
  out[(size_t)x * m1 + y] = scale * in[(size_t)sx * n1 + sy];
}

// FIR along rows (axis 1): one workgroup = 256 consecutive outputs of one row
template <class T>
__global__ __launch_bounds__(256) void fir_rows_kernel(const T* __restrict__ in, int m0, int m1, int ext,
                                                      const T* __restrict__ h, T* __restrict__ out) {
  __shared__ T tile[256 + 2 * KT];
  __shared__ T hs[2 * KT + 1];
  const int x = blockIdx.y, y0 = blockIdx.x * 256;
  for (int i = threadIdx.x; i < 256 + 2 * KT; i += 256) tile[i] = in[(size_t)x * m1 + ext_index(y0 - KT + i, m1, ext)];
  if (threadIdx.x < 2 * KT + 1) hs[threadIdx.x] = h[threadIdx.x];
  __syncthreads();
  const int y = y0 + threadIdx.x;
  if (y >= m1) return;
  T acc = T(0);
#pragma unroll 5
  for (int k = 0; k <= 2 * KT; ++k) acc += hs[k] * tile[threadIdx.x + k];
  out[(size_t)x * m1 + y] = acc;
}

// FIR along columns (axis 0): one workgroup = 32 rows x 64 columns of outputs
template <class T>
__global__ __launch_bounds__(256) void fir_cols_kernel(const T* __restrict__ in, int m0, int m1, int ext,
                                                      const T* __restrict__ h, T* __restrict__ out) {
  __shared__ T tile[(32 + 2 * KT) * 64];
  __shared__ T hs[2 * KT + 1];
  const int y0 = blockIdx.x * 64, x0 = blockIdx.y * 32;
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int yc = y0 + c < m1 ? y0 + c : m1 - 1;
  for (int r = g; r < 32 + 2 * KT; r += 4) tile[r * 64 + c] = in[(size_t)ext_index(x0 - KT + r, m0, ext) * m1 + yc];
  if (threadIdx.x < 2 * KT + 1) hs[threadIdx.x] = h[threadIdx.x];
  __syncthreads();
  if (y0 + c >= m1) return;
  for (int r = g * 8; r < g * 8 + 8; ++r) {
    if (x0 + r >= m0) break;
    T acc = T(0);
#pragma unroll 5
    for (int k = 0; k <= 2 * KT; ++k) acc += hs[k] * tile[(r + k) * 64 + c];
    out[(size_t)(x0 + r) * m1 + y0 + c] = acc;
  }
}

template <class T>
__device__ __forceinline__ void bspline_weights(T t, T (&w)[4]) {
  const T z = T(1) - t;
  w[1] = (t * t * (t - T(2)) * T(3) + T(4)) / T(6);
  w[2] = (z * z * (z - T(2)) * T(3) + T(4)) / T(6);
  w[0] = z * z * z / T(6);
  w[3] = T(1) - w[0] - w[1] - w[2];
}

// mode='nearest': coordinate (already shifted by npad) unclamped, tap indices clamped
template <class T, int NC>
__device__ __forceinline__ void interp_nearest(const T* const (&coef)[NC], int m0, int m1, T x, T y, T (&out)[NC]) {
  // keep floor() finite for wild coordinates: everything beyond one sample outside reads the edge
  x = x < T(-2) ? T(-2) : (x > T(m0 + 1) ? T(m0 + 1) : x);
  y = y < T(-2) ? T(-2) : (y > T(m1 + 1) ? T(m1 + 1) : y);
  const T fx = floor(x), fy = floor(y);
  T wx[4], wy[4];
  bspline_weights(x - fx, wx);
  bspline_weights(y - fy, wy);
  const int ix = (int)fx - 1, iy = (int)fy - 1;
  int cy[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { const int j = iy + b; cy[b] = j < 0 ? 0 : (j >= m1 ? m1 - 1 : j); }
#pragma unroll
  for (int n = 0; n < NC; ++n) out[n] = T(0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int i = ix + a;
    i = i < 0 ? 0 : (i >= m0 ? m0 - 1 : i);
    const size_t row = (size_t)i * m1;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const T* cr = coef[n] + row;
      out[n] += wx[a] * (wy[0] * cr[cy[0]] + wy[1] * cr[cy[1]] + wy[2] * cr[cy[2]] + wy[3] * cr[cy[3]]);
    }
  }
}

// mode='constant': whole-sample mirrored taps, `cval` where the coordinate leaves [0, n-1] (NaN coordinates too)
template <class T, int NC>
__device__ __forceinline__ void interp_constant(const T* const (&coef)[NC], int n0, int n1, T x, T y, T cval, T (&out)[NC]) {
  if (!(x >= T(0) && x <= T(n0 - 1) && y >= T(0) && y <= T(n1 - 1))) {
#pragma unroll
    for (int n = 0; n < NC; ++n) out[n] = cval;
    return;
  }
  const T fx = floor(x), fy = floor(y);
  T wx[4], wy[4];
  bspline_weights(x - fx, wx);
  bspline_weights(y - fy, wy);
  const int ix = (int)fx - 1, iy = (int)fy - 1;
  int cy[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) cy[b] = ext_index(iy + b, n1, EXT_MIRROR);
#pragma unroll
  for (int n = 0; n < NC; ++n) out[n] = T(0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const size_t row = (size_t)ext_index(ix + a, n0, EXT_MIRROR) * n1;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const T* cr = coef[n] + row;
      out[n] += wx[a] * (wy[0] * cr[cy[0]] + wy[1] * cr[cy[1]] + wy[2] * cr[cy[2]] + wy[3] * cr[cy[3]]);
    }
  }
}

// the same fixed point with scipy's mode='constant' (geometric_phase_analysis.py:248, :262 `mode=`): coefficients of the
// unpadded field, 0 outside it in every round but the last of the overlap variant, which passes cval=nan (:297-299)
template <class T>
__global__ __launch_bounds__(256) void invert_constant_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int n0, int n1,
                                                             int edge, int shift, int iters, int nan_last,
                                                             T* __restrict__ out) {
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= o1) return;
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge), yb = T(j - edge);
  T v[2];
  interp_constant<T, 2>(coef, n0, n1, xb, yb, T(0), v);
  const T xs = xb - T(shift), ys = yb - T(shift);
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    const T cval = (nan_last && it == iters - 1) ? (T)__builtin_nan("") : T(0);
    interp_constant<T, 2>(coef, n0, n1, xs + v[0], ys + v[1], cval, nv);
    v[0] = nv[0];
    v[1] = nv[1];
  }
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// u_it(r) <- u(r + u_it(r)), all rounds for one pixel (geometric_phase_analysis.py:291-299)
template <class T>
__global__ __launch_bounds__(256) void invert_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int m0, int m1,
                                                    int n0, int n1, int edge, int shift, int iters, T* __restrict__ out) {
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= o1) return;
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge + NPAD), yb = T(j - edge + NPAD);
  T v[2];
  interp_nearest<T, 2>(coef, m0, m1, xb, yb, v);
  // (shift != 0: invert_u, which samples every later round at r + u_it - shift, geometric_phase_analysis.py:258)
  const T xs = xb - T(shift), ys = yb - T(shift);
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    interp_nearest<T, 2>(coef, m0, m1, xs + v[0], ys + v[1], nv);
    v[0] = nv[0];
    v[1] = nv[1];
  }
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// final resampling, map_coordinates defaults: order 3, mode='constant', cval = 0
template <class T>
__global__ __launch_bounds__(256) void warp_constant_kernel(const T* __restrict__ coef, int n0, int n1,
                                                           const T* __restrict__ uinv, T cval, T* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= n1) return;
  const size_t o = (size_t)i * n1 + j, npx = (size_t)n0 * n1;
  const T x = T(i) + uinv[o], y = T(j) + uinv[npx + o];
  if (!(x >= T(0) && x <= T(n0 - 1) && y >= T(0) && y <= T(n1 - 1))) {   // also catches NaN
    out[o] = cval;
    return;
  }
  const T fx = floor(x), fy = floor(y);
  T wx[4], wy[4];
  bspline_weights(x - fx, wx);
  bspline_weights(y - fy, wy);
  const int ix = (int)fx - 1, iy = (int)fy - 1;
  T acc = T(0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const T* cr = coef + (size_t)ext_index(ix + a, n0, EXT_MIRROR) * n1;
    T r = T(0);
#pragma unroll
    for (int b = 0; b < 4; ++b) r += wy[b] * cr[ext_index(iy + b, n1, EXT_MIRROR)];
    acc += wx[a] * r;
  }
  out[o] = acc;
}

template <class T>
hipError_t build_taps(T** d_h, hipStream_t s) {
  const double z = sqrt(3.0) - 2.0;
  T h[2 * KT + 1];
  for (int k = -KT; k <= KT; ++k) h[k + KT] = (T)((-6.0 * z / (1.0 - z * z)) * pow(z, abs(k)));
  hipError_t e = hipMalloc((void**)d_h, sizeof(h));
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(*d_h, h, sizeof(h), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);   // h lives on this stack frame
}

// coefficients of `in` (m0 x m1, already padded if the mode wants it); tmp: same size
template <class T>
hipError_t prefilter(const T* in, int m0, int m1, int ext, const T* d_h, T* tmp, T* out, hipStream_t s) {
  fir_rows_kernel<T><<<dim3((m1 + 255) / 256, m0), 256, 0, s>>>(in, m0, m1, ext, d_h, tmp);
  fir_cols_kernel<T><<<dim3((m1 + 63) / 64, (m0 + 31) / 32), 256, 0, s>>>(tmp, m0, m1, ext, d_h, out);
  return hipGetLastError();
}

template <class T>
hipError_t invert_constant_t(const T* d_u, int n0, int n1, T scale, int iters, int edge, int shift, int nan_last, T* d_out,
                             hipStream_t s) {
  const size_t npx = (size_t)n0 * n1;
  T *buf = nullptr, *d_h = nullptr;
  hipError_t e = hipMalloc((void**)&buf, 4 * npx * sizeof(T));   // scaled copy, tmp, coef0, coef1
  if (e != hipSuccess) return e;
  e = build_taps<T>(&d_h, s);
  T *cp = buf, *tmp = buf + npx, *c0 = buf + 2 * npx, *c1 = buf + 3 * npx;
  for (int c = 0; c < 2 && e == hipSuccess; ++c) {
    pad_edge_kernel<T><<<dim3((n1 + 255) / 256, n0), 256, 0, s>>>(d_u + (size_t)c * npx, n0, n1, 0, scale, cp);
    e = prefilter<T>(cp, n0, n1, EXT_MIRROR, d_h, tmp, c == 0 ? c0 : c1, s);
  }
  if (e == hipSuccess) {
    const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
    // nan_last: invert_u_overlap ends on a cval=nan round (geometric_phase_analysis.py:296-299); invert_u never passes
    // cval (:255-258) and keeps 0 outside -- said by the caller, not guessed from the geometry (invert_u with its
    // default edge = 0 has shift == 0 too)
    invert_constant_kernel<T><<<dim3((o1 + 255) / 256, o0), 256, 0, s>>>(c0, c1, n0, n1, edge, shift, iters, nan_last ? 1 : 0, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(buf);
  if (d_h) hipFree(d_h);
  return e;
}

template <class T>
hipError_t invert_t(const T* d_u, int n0, int n1, T scale, int iters, int edge, int shift, T* d_out, hipStream_t s) {
  const int m0 = n0 + 2 * NPAD, m1 = n1 + 2 * NPAD;
  const size_t mp = (size_t)m0 * m1;
  T *buf = nullptr, *d_h = nullptr;
  hipError_t e = hipMalloc((void**)&buf, 4 * mp * sizeof(T));   // padded, tmp, coef0, coef1
  if (e != hipSuccess) return e;
  e = build_taps<T>(&d_h, s);
  T *pad = buf, *tmp = buf + mp, *c0 = buf + 2 * mp, *c1 = buf + 3 * mp;
  for (int c = 0; c < 2 && e == hipSuccess; ++c) {
    pad_edge_kernel<T><<<dim3((m1 + 255) / 256, m0), 256, 0, s>>>(d_u + (size_t)c * n0 * n1, n0, n1, NPAD, scale, pad);
    e = prefilter<T>(pad, m0, m1, EXT_REFLECT, d_h, tmp, c == 0 ? c0 : c1, s);
  }
  if (e == hipSuccess) {
    const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
    invert_kernel<T><<<dim3((o1 + 255) / 256, o0), 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(buf);
  if (d_h) hipFree(d_h);
  return e;
}

template <class T>
hipError_t warp_t(const T* d_img, const T* d_uinv, int n0, int n1, T* d_out, hipStream_t s) {
  const size_t npx = (size_t)n0 * n1;
  T *buf = nullptr, *d_h = nullptr;
  hipError_t e = hipMalloc((void**)&buf, 2 * npx * sizeof(T));
  if (e != hipSuccess) return e;
  e = build_taps<T>(&d_h, s);
  if (e == hipSuccess) e = prefilter<T>(d_img, n0, n1, EXT_MIRROR, d_h, buf, buf + npx, s);
  if (e == hipSuccess) {
    warp_constant_kernel<T><<<dim3((n1 + 255) / 256, n0), 256, 0, s>>>(buf + npx, n0, n1, d_uinv, T(0), d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(buf);
  if (d_h) hipFree(d_h);
  return e;
}

}  // namespace

// d_u: 2 x n0 x n1 (device); the field that is inverted is scale * u; d_out: 2 x (n0+2e) x (n1+2e)
hipError_t warp_invert_u(int dtype, const void* d_u, int n0, int n1, double scale, int iters, int edge, int shift,
                         void* d_out, hipStream_t s, int mode, int nan_last) {
  if (mode == 1)
    return dtype == 0 ? invert_constant_t<float>((const float*)d_u, n0, n1, (float)scale, iters, edge, shift, nan_last, (float*)d_out, s)
                      : invert_constant_t<double>((const double*)d_u, n0, n1, scale, iters, edge, shift, nan_last, (double*)d_out, s);
  return dtype == 0 ? invert_t<float>((const float*)d_u, n0, n1, (float)scale, iters, edge, shift, (float*)d_out, s)
                    : invert_t<double>((const double*)d_u, n0, n1, scale, iters, edge, shift, (double*)d_out, s);
}
// resample d_img (n0 x n1) at r + u_inv(r), order 3, mode='constant', cval=0
hipError_t warp_image(int dtype, const void* d_img, const void* d_uinv, int n0, int n1, void* d_out, hipStream_t s) {
  return dtype == 0 ? warp_t<float>((const float*)d_img, (const float*)d_uinv, n0, n1, (float*)d_out, s)
                    : warp_t<double>((const double*)d_img, (const double*)d_uinv, n0, n1, (double*)d_out, s);
}

}  // namespace gpa
