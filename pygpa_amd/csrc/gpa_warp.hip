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

// FIR along rows (axis 1): one workgroup = FR_OUT consecutive outputs of one row, a thread computes FPT consecutive ones
// from a register window of FPT + 2 KT inputs (read from the LDS tile once: ~10 LDS reads per output where the
// one-output-per-thread form of rounds 1-4 made 130; the taps are uniform and come through the scalar cache).  Same
// sum, same order of additions per output.
constexpr int FPT = 8;                 // outputs per thread
constexpr int FR_OUT = 256 * FPT;      // outputs per workgroup (rows)
template <class T>
__global__ __launch_bounds__(256) void fir_rows_kernel(const T* __restrict__ in, int m0, int m1, int ext,
                                                      const T* __restrict__ h, T* __restrict__ out) {
  __shared__ T tile[FR_OUT + 2 * KT];
  const int x = blockIdx.y, y0 = blockIdx.x * FR_OUT;
  const T* row = in + (size_t)x * m1;
  const bool interior = y0 - KT >= 0 && y0 + FR_OUT + KT <= m1;
  if (interior) {
    for (int i = threadIdx.x; i < FR_OUT + 2 * KT; i += 256) tile[i] = row[y0 - KT + i];
  } else {
    for (int i = threadIdx.x; i < FR_OUT + 2 * KT; i += 256) tile[i] = row[ext_index(y0 - KT + i, m1, ext)];
  }
  __syncthreads();
  const int t0 = threadIdx.x * FPT;
  if (y0 + t0 >= m1) return;
  T win[FPT + 2 * KT];
#pragma unroll
  for (int i = 0; i < FPT + 2 * KT; ++i) win[i] = tile[t0 + i];
  T acc[FPT];
#pragma unroll
  for (int o = 0; o < FPT; ++o) acc[o] = T(0);
#pragma unroll
  for (int k = 0; k <= 2 * KT; ++k) {
    const T hk = h[k];
#pragma unroll
    for (int o = 0; o < FPT; ++o) acc[o] += hk * win[o + k];
  }
  T* orow = out + (size_t)x * m1 + y0 + t0;
#pragma unroll
  for (int o = 0; o < FPT; ++o)
    if (y0 + t0 + o < m1) orow[o] = acc[o];
}

// FIR along columns (axis 0): one workgroup = 32 rows x 64 columns of outputs; a thread computes 8 consecutive rows of
// one column from a register window of 8 + 2 KT inputs (LDS column reads: conflict-free, 64 lanes = 64 banks)
template <class T>
__global__ __launch_bounds__(256) void fir_cols_kernel(const T* __restrict__ in, int m0, int m1, int ext,
                                                      const T* __restrict__ h, T* __restrict__ out) {
  __shared__ T tile[(32 + 2 * KT) * 64];
  const int y0 = blockIdx.x * 64, x0 = blockIdx.y * 32;
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int yc = y0 + c < m1 ? y0 + c : m1 - 1;
  for (int r = g; r < 32 + 2 * KT; r += 4) tile[r * 64 + c] = in[(size_t)ext_index(x0 - KT + r, m0, ext) * m1 + yc];
  __syncthreads();
  if (y0 + c >= m1) return;
  T win[8 + 2 * KT];
#pragma unroll
  for (int i = 0; i < 8 + 2 * KT; ++i) win[i] = tile[(g * 8 + i) * 64 + c];
  T acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = T(0);
#pragma unroll
  for (int k = 0; k <= 2 * KT; ++k) {
    const T hk = h[k];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] += hk * win[o + k];
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int r = x0 + g * 8 + o;
    if (r < m0) out[(size_t)r * m1 + y0 + c] = acc[o];
  }
}

// cubic B-spline weights as scipy.ndimage evaluates them (ni_splines.c: get_spline_interpolation_weights, order 3).
// f64: the divisions by 6 as written (pinned to SciPy at 4e-15); f32: times 1/6 -- one instruction where an IEEE
// division takes ten, six times per round of the fixed point, and 0.5 ulp of f32 either way
template <class T>
__device__ __forceinline__ T sixth(T v) {
  if constexpr (sizeof(T) == 4) return v * T(0.16666666666666666);
  else return v / T(6);
}
template <class T>
__device__ __forceinline__ void bspline_weights(T t, T (&w)[4]) {
  const T z = T(1) - t;
  w[1] = sixth(t * t * (t - T(2)) * T(3) + T(4));
  w[2] = sixth(z * z * (z - T(2)) * T(3) + T(4));
  w[0] = sixth(z * z * z);
  w[3] = T(1) - w[0] - w[1] - w[2];
}
// index type of the coefficient gathers: 32-bit element offsets from a uniform base (one address instruction per tap,
// shared by the two components) while the field is below 2^32 bytes, 64-bit beyond
template <bool WIDE> struct GatherIdx { typedef unsigned type; };
template <> struct GatherIdx<true> { typedef size_t type; };
// element `o` of a field: as base + 32-bit BYTE offset (the form the scalar-base global loads take) or a 64-bit index
template <class T> __device__ __forceinline__ T gather(const T* __restrict__ base, unsigned o) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (unsigned)(o * (unsigned)sizeof(T)));
}
template <class T> __device__ __forceinline__ T gather(const T* __restrict__ base, size_t o) { return base[o]; }

// mode='nearest': coordinate (already shifted by npad) unclamped, tap indices clamped
template <class T, int NC, bool WIDE = false>
__device__ __forceinline__ void interp_nearest(const T* const (&coef)[NC], int m0, int m1, T x, T y, T (&out)[NC]) {
  typedef typename GatherIdx<WIDE>::type I;
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
    const I row = (I)i * (I)m1;
    const I o0 = row + (I)cy[0], o1 = row + (I)cy[1], o2 = row + (I)cy[2], o3 = row + (I)cy[3];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const T* cr = coef[n];
      out[n] += wx[a] * (wy[0] * gather(cr, o0) + wy[1] * gather(cr, o1) + wy[2] * gather(cr, o2) + wy[3] * gather(cr, o3));
    }
  }
}

// mode='constant': whole-sample mirrored taps, `cval` where the coordinate leaves [0, n-1] (NaN coordinates too)
template <class T, int NC, bool WIDE = false>
__device__ __forceinline__ void interp_constant(const T* const (&coef)[NC], int n0, int n1, T x, T y, T cval, T (&out)[NC]) {
  typedef typename GatherIdx<WIDE>::type I;
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
    const I row = (I)ext_index(ix + a, n0, EXT_MIRROR) * (I)n1;
    const I o0 = row + (I)cy[0], o1 = row + (I)cy[1], o2 = row + (I)cy[2], o3 = row + (I)cy[3];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const T* cr = coef[n];
      out[n] += wx[a] * (wy[0] * gather(cr, o0) + wy[1] * gather(cr, o1) + wy[2] * gather(cr, o2) + wy[3] * gather(cr, o3));
    }
  }
}

// the same fixed point with scipy's mode='constant' (geometric_phase_analysis.py:248, :262 `mode=`): coefficients of the
// unpadded field, 0 outside it in every round but the last of the overlap variant, which passes cval=nan (:297-299)
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void invert_constant_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int n0, int n1,
                                                             int edge, int shift, int iters, int nan_last,
                                                             T* __restrict__ out, int wr0, int wc0, int wc1) {
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int j = wc0 + blockIdx.x * 256 + threadIdx.x, i = wr0 + blockIdx.y;
  if (j >= wc1) return;
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge), yb = T(j - edge);
  T v[2];
  interp_constant<T, 2, WIDE>(coef, n0, n1, xb, yb, T(0), v);
  const T xs = xb - T(shift), ys = yb - T(shift);
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    const bool last_nan = nan_last && it == iters - 1;
    const T cval = last_nan ? (T)__builtin_nan("") : T(0);
    interp_constant<T, 2, WIDE>(coef, n0, n1, xs + v[0], ys + v[1], cval, nv);
    // a round that reproduces its input bit for bit is a fixed point: every later round returns the same numbers, so the
    // wavefront leaves once all its pixels are there (the same result as running all rounds; f32 fields settle after
    // ~15 of the reference's 36 rounds).  The cval = NaN round of invert_u_overlap is a different function: still run.
    const bool fixed = nv[0] == v[0] && nv[1] == v[1];
    v[0] = nv[0];
    v[1] = nv[1];
    if (__all(fixed)) {
      if (nan_last && !last_nan) interp_constant<T, 2, WIDE>(coef, n0, n1, xs + v[0], ys + v[1], (T)__builtin_nan(""), v);
      break;
    }
  }
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// u_it(r) <- u(r + u_it(r)), all rounds for one pixel (geometric_phase_analysis.py:291-299)
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void invert_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int m0, int m1,
                                                    int n0, int n1, int edge, int shift, int iters, T* __restrict__ out,
                                                    int wr0, int wc0, int wc1) {
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int j = wc0 + blockIdx.x * 256 + threadIdx.x, i = wr0 + blockIdx.y;
  if (j >= wc1) return;
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge + NPAD), yb = T(j - edge + NPAD);
  T v[2];
  interp_nearest<T, 2, WIDE>(coef, m0, m1, xb, yb, v);
  // (shift != 0: invert_u, which samples every later round at r + u_it - shift, geometric_phase_analysis.py:258)
  const T xs = xb - T(shift), ys = yb - T(shift);
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    interp_nearest<T, 2, WIDE>(coef, m0, m1, xs + v[0], ys + v[1], nv);
    // bitwise fixed point of every pixel of the wavefront: all later rounds return the same numbers (see above)
    const bool fixed = nv[0] == v[0] && nv[1] == v[1];
    v[0] = nv[0];
    v[1] = nv[1];
    if (__all(fixed)) break;
  }
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// final resampling, map_coordinates defaults: order 3, mode='constant', cval = 0
template <class T>
__global__ __launch_bounds__(256) void warp_constant_kernel(const T* __restrict__ coef, int n0, int n1,
                                                           const T* __restrict__ uinv, T cval, T* __restrict__ out,
                                                           int wr0, int wc0, int wc1) {
  const int j = wc0 + blockIdx.x * 256 + threadIdx.x, i = wr0 + blockIdx.y;
  if (j >= wc1) return;
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

// the prefilter's taps, uploaded once per workspace and precision (pinned staging is not worth it: 65 values)
template <class T>
hipError_t ensure_taps(WarpWs* ws, hipStream_t s) {
  const int want = sizeof(T) == 4 ? 0 : 1;
  if (ws->taps && ws->taps_dtype == want) return hipSuccess;
  const double z = sqrt(3.0) - 2.0;
  T h[2 * KT + 1];
  for (int k = -KT; k <= KT; ++k) h[k + KT] = (T)((-6.0 * z / (1.0 - z * z)) * pow(z, abs(k)));
  hipError_t e = hipSuccess;
  if (!ws->taps) e = hipMalloc(&ws->taps, (2 * KT + 1) * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(ws->taps, h, sizeof(h), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  ws->taps_dtype = want;
  return hipStreamSynchronize(s);   // h lives on this stack frame (once per workspace)
}
// scratch of at least `bytes`: grown (stream drained, old buffer freed) only when a call needs more than any before it
hipError_t reserve(WarpWs* ws, size_t bytes, hipStream_t s) {
  if (ws->cap >= bytes) return hipSuccess;
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  if (ws->buf) (void)hipFree(ws->buf);
  ws->buf = nullptr;
  ws->cap = 0;
  e = hipMalloc(&ws->buf, bytes);
  if (e != hipSuccess) return e;
  ws->cap = bytes;
  return hipSuccess;
}

// coefficients of `in` (m0 x m1, already padded if the mode wants it); tmp: same size
template <class T>
hipError_t prefilter(const T* in, int m0, int m1, int ext, const T* d_h, T* tmp, T* out, hipStream_t s) {
  {
    GPA_PROF("fir_rows_kernel", s);
    fir_rows_kernel<T><<<dim3((m1 + FR_OUT - 1) / FR_OUT, m0), 256, 0, s>>>(in, m0, m1, ext, d_h, tmp);
  }
  {
    GPA_PROF("fir_cols_kernel", s);
    fir_cols_kernel<T><<<dim3((m1 + 63) / 64, (m0 + 31) / 32), 256, 0, s>>>(tmp, m0, m1, ext, d_h, out);
  }
  return hipGetLastError();
}

// a window {r0, c0, h, w} of the output grid (o0 x o1), or null for all of it; nrect windows are worked through one
// after the other behind ONE prefilter
struct Win { int r0, c0, h, w; };
inline Win window(const int* rect, int o0, int o1) {
  if (!rect) return {0, 0, o0, o1};
  Win v{rect[0], rect[1], rect[2], rect[3]};
  if (v.r0 < 0) { v.h += v.r0; v.r0 = 0; }
  if (v.c0 < 0) { v.w += v.c0; v.c0 = 0; }
  if (v.r0 + v.h > o0) v.h = o0 - v.r0;
  if (v.c0 + v.w > o1) v.w = o1 - v.c0;
  return v;
}

template <class T>
hipError_t invert_constant_t(const T* d_u, int n0, int n1, T scale, int iters, int edge, int shift, int nan_last, T* d_out,
                             hipStream_t s, WarpWs* ws, const int* rects, int nrect) {
  const size_t npx = (size_t)n0 * n1;
  hipError_t e = reserve(ws, 4 * npx * sizeof(T), s);   // scaled copy, tmp, coef0, coef1
  if (e == hipSuccess) e = ensure_taps<T>(ws, s);
  if (e != hipSuccess) return e;
  T* buf = (T*)ws->buf;
  const T* d_h = (const T*)ws->taps;
  T *cp = buf, *tmp = buf + npx, *c0 = buf + 2 * npx, *c1 = buf + 3 * npx;
  for (int c = 0; c < 2 && e == hipSuccess; ++c) {
    pad_edge_kernel<T><<<dim3((n1 + 255) / 256, n0), 256, 0, s>>>(d_u + (size_t)c * npx, n0, n1, 0, scale, cp);
    e = prefilter<T>(cp, n0, n1, EXT_MIRROR, d_h, tmp, c == 0 ? c0 : c1, s);
  }
  if (e == hipSuccess) {
    const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
    for (int q = 0; q < (nrect > 0 ? nrect : 1); ++q) {
    const Win v = window(nrect > 0 ? rects + 4 * q : nullptr, o0, o1);
    if (v.h <= 0 || v.w <= 0) continue;
    // nan_last: invert_u_overlap ends on a cval=nan round (geometric_phase_analysis.py:296-299); invert_u never passes
    // cval (:255-258) and keeps 0 outside -- said by the caller, not guessed from the geometry (invert_u with its
    // default edge = 0 has shift == 0 too)
    GPA_PROF("invert_kernel", s);
    if (npx * sizeof(T) < ((size_t)1 << 32))
      invert_constant_kernel<T, false><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, n0, n1, edge, shift, iters, nan_last ? 1 : 0, d_out,
                                                                                   v.r0, v.c0, v.c0 + v.w);
    else
      invert_constant_kernel<T, true><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, n0, n1, edge, shift, iters, nan_last ? 1 : 0, d_out,
                                                                                  v.r0, v.c0, v.c0 + v.w);
    }
    e = hipGetLastError();
  }
  return e;
}

template <class T>
hipError_t invert_t(const T* d_u, int n0, int n1, T scale, int iters, int edge, int shift, T* d_out, hipStream_t s, WarpWs* ws,
                    const int* rects, int nrect) {
  const int m0 = n0 + 2 * NPAD, m1 = n1 + 2 * NPAD;
  const size_t mp = (size_t)m0 * m1;
  hipError_t e = reserve(ws, 4 * mp * sizeof(T), s);   // padded, tmp, coef0, coef1
  if (e == hipSuccess) e = ensure_taps<T>(ws, s);
  if (e != hipSuccess) return e;
  T* buf = (T*)ws->buf;
  const T* d_h = (const T*)ws->taps;
  T *pad = buf, *tmp = buf + mp, *c0 = buf + 2 * mp, *c1 = buf + 3 * mp;
  for (int c = 0; c < 2 && e == hipSuccess; ++c) {
    {
      GPA_PROF("pad_edge_kernel", s);
      pad_edge_kernel<T><<<dim3((m1 + 255) / 256, m0), 256, 0, s>>>(d_u + (size_t)c * n0 * n1, n0, n1, NPAD, scale, pad);
    }
    e = prefilter<T>(pad, m0, m1, EXT_REFLECT, d_h, tmp, c == 0 ? c0 : c1, s);
  }
  if (e == hipSuccess) {
    const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
    for (int q = 0; q < (nrect > 0 ? nrect : 1); ++q) {
      const Win v = window(nrect > 0 ? rects + 4 * q : nullptr, o0, o1);
      if (v.h <= 0 || v.w <= 0) continue;
      GPA_PROF("invert_kernel", s);
      if (mp * sizeof(T) < ((size_t)1 << 32))
        invert_kernel<T, false><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out, v.r0, v.c0, v.c0 + v.w);
      else
        invert_kernel<T, true><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out, v.r0, v.c0, v.c0 + v.w);
    }
    e = hipGetLastError();
  }
  return e;
}

template <class T>
hipError_t warp_t(const T* d_img, const T* d_uinv, int n0, int n1, T* d_out, hipStream_t s, WarpWs* ws, const int* rects, int nrect) {
  const size_t npx = (size_t)n0 * n1;
  // (its own scratch region BEHIND what an inversion of this shape uses: undistort_image inverts and warps back to back on
  //  one stream, and a second workspace user must not shrink the first)
  const size_t inv_bytes = 4 * (size_t)(n0 + 2 * NPAD) * (n1 + 2 * NPAD) * sizeof(T);
  hipError_t e = reserve(ws, inv_bytes + 2 * npx * sizeof(T), s);
  if (e == hipSuccess) e = ensure_taps<T>(ws, s);
  if (e != hipSuccess) return e;
  T* buf = (T*)((char*)ws->buf + inv_bytes);
  e = prefilter<T>(d_img, n0, n1, EXT_MIRROR, (const T*)ws->taps, buf, buf + npx, s);
  if (e == hipSuccess) {
    for (int q = 0; q < (nrect > 0 ? nrect : 1); ++q) {
      const Win v = window(nrect > 0 ? rects + 4 * q : nullptr, n0, n1);
      if (v.h <= 0 || v.w <= 0) continue;
      GPA_PROF("warp_constant_kernel", s);
      warp_constant_kernel<T><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(buf + npx, n0, n1, d_uinv, T(0), d_out, v.r0, v.c0, v.c0 + v.w);
    }
    e = hipGetLastError();
  }
  return e;
}

}  // namespace

void warp_ws_free(WarpWs* ws) {
  if (ws->buf) (void)hipFree(ws->buf);
  if (ws->taps) (void)hipFree(ws->taps);
  *ws = WarpWs{};
}

// d_u: 2 x n0 x n1 (device); the field that is inverted is scale * u; d_out: 2 x (n0+2e) x (n1+2e).  Enqueued on s, no
// host synchronisation (the workspace grows with one, the first time a shape needs more).
hipError_t warp_invert_u(int dtype, const void* d_u, int n0, int n1, double scale, int iters, int edge, int shift,
                         void* d_out, hipStream_t s, int mode, int nan_last, WarpWs* ws, const int* rects, int nrect) {
  if (mode == 1)
    return dtype == 0 ? invert_constant_t<float>((const float*)d_u, n0, n1, (float)scale, iters, edge, shift, nan_last, (float*)d_out, s, ws, rects, nrect)
                      : invert_constant_t<double>((const double*)d_u, n0, n1, scale, iters, edge, shift, nan_last, (double*)d_out, s, ws, rects, nrect);
  return dtype == 0 ? invert_t<float>((const float*)d_u, n0, n1, (float)scale, iters, edge, shift, (float*)d_out, s, ws, rects, nrect)
                    : invert_t<double>((const double*)d_u, n0, n1, scale, iters, edge, shift, (double*)d_out, s, ws, rects, nrect);
}
// resample d_img (n0 x n1) at r + u_inv(r), order 3, mode='constant', cval=0
hipError_t warp_image(int dtype, const void* d_img, const void* d_uinv, int n0, int n1, void* d_out, hipStream_t s, WarpWs* ws,
                      const int* rects, int nrect) {
  return dtype == 0 ? warp_t<float>((const float*)d_img, (const float*)d_uinv, n0, n1, (float*)d_out, s, ws, rects, nrect)
                    : warp_t<double>((const double*)d_img, (const double*)d_uinv, n0, n1, (double*)d_out, s, ws, rects, nrect);
}

}  // namespace gpa
