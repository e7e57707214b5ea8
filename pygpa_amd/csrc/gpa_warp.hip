// f-1: Lawler-Fujita undistortion -- invert_u_overlap / undistort_image
// (geometric_phase_analysis.py:262-300, :935-974), i.e. scipy.ndimage.map_coordinates
// (order 3) restated for the GPU.  SciPy's conventions, pinned numerically to 4e-15 against
// SciPy itself before this was written:
//   prefilter   : cubic B-spline coefficients = two-sided exponential filter
//                 h_k = (-6 z / (1 - z^2)) z^|k|, z = sqrt(3) - 2, along both axes;
//                 evaluated here as a 2*KT+1 tap FIR (f64: KT = 32, |z|^KT < 1e-18; f32: KT = 16, 7e-10) instead of SciPy's
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

// taps on each side of the prefilter: the pole is z = sqrt(3) - 2, |z|^k falls below the precision's rounding of the
// centre tap at k = 16 in f32 (|z|^16 = 7e-10 against 2^-24 = 6e-8) and at k = 28 in f64 (1e-16; 32 kept: |z|^32 = 5e-19)
template <class T> struct TapHalf { static constexpr int value = sizeof(T) == 4 ? 16 : 32; };
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

// FIR along rows (axis 1): one workgroup = FR_OUT consecutive outputs of one row, a thread computes FPT consecutive ones
// from a register window of FPT + 2 KT inputs (read from the LDS tile once: ~10 LDS reads per output where the
// one-output-per-thread form of rounds 1-4 made 130; the taps are uniform and come through the scalar cache).  Same
// sum, same order of additions per output.
constexpr int FPT = 8;                 // outputs per thread
constexpr int FR_OUT = 256 * FPT;      // outputs per workgroup (rows)
// PADSRC: `in` is the UNPADDED sn0 x sn1 field and the m0 x m1 array this pass filters is its edge-replicated padding by npad
// samples, scaled: element (x, y) = scale * in[clamp(x - npad)][clamp(y - npad)] is formed while the tile is loaded -- what
// pad_edge_kernel used to write to memory and this kernel to read back (2 x 1.2 ms at 16384^2)
template <class T, bool PADSRC>
__global__ __launch_bounds__(256) void fir_rows_kernel(const T* __restrict__ in, int m0, int m1, int ext,
                                                      const T* __restrict__ h, T* __restrict__ out, int npad, int sn0, int sn1,
                                                      T scale) {
  constexpr int KT = TapHalf<T>::value;
  __shared__ T tile[FR_OUT + 2 * KT];
  const int x = blockIdx.y, y0 = blockIdx.x * FR_OUT;
  if constexpr (PADSRC) {
    int sx = x - npad;
    sx = sx < 0 ? 0 : (sx >= sn0 ? sn0 - 1 : sx);
    const T* row = in + (size_t)sx * sn1;
    for (int i = threadIdx.x; i < FR_OUT + 2 * KT; i += 256) {
      int sy = ext_index(y0 - KT + i, m1, ext) - npad;
      sy = sy < 0 ? 0 : (sy >= sn1 ? sn1 - 1 : sy);
      tile[i] = scale * row[sy];
    }
  } else {
    const T* row = in + (size_t)x * m1;
    const bool interior = y0 - KT >= 0 && y0 + FR_OUT + KT <= m1;
    if (interior) {
      for (int i = threadIdx.x; i < FR_OUT + 2 * KT; i += 256) tile[i] = row[y0 - KT + i];
    } else {
      for (int i = threadIdx.x; i < FR_OUT + 2 * KT; i += 256) tile[i] = row[ext_index(y0 - KT + i, m1, ext)];
    }
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
  if ((m1 & 3) == 0 && y0 + t0 + FPT <= m1) {
    // whole 16-byte (f32: two 16-byte) stores: eight scalar stores per thread, 32 bytes apart between lanes, used a quarter of
    // every line they touched
    struct alignas(16) V { T v[16 / sizeof(T)]; };
    constexpr int PER = 16 / sizeof(T);
#pragma unroll
    for (int o = 0; o < FPT; o += PER) {
      V q;
#pragma unroll
      for (int e = 0; e < PER; ++e) q.v[e] = acc[o + e];
      *reinterpret_cast<V*>(orow + o) = q;
    }
    return;
  }
#pragma unroll
  for (int o = 0; o < FPT; ++o)
    if (y0 + t0 + o < m1) orow[o] = acc[o];
}

// FIR along columns (axis 0): one workgroup = 32 rows x 64 columns of outputs; a thread computes 8 consecutive rows of
// one column from a register window of 8 + 2 KT inputs (LDS column reads: conflict-free, 64 lanes = 64 banks)
template <class T>
__global__ __launch_bounds__(256) void fir_cols_kernel(const T* __restrict__ in, int m0, int m1, int ext,
                                                      const T* __restrict__ h, T* __restrict__ out) {
  constexpr int KT = TapHalf<T>::value;
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
#pragma clang fp contract(off)   // (the same bits at every call site; SciPy's C evaluates these without fused operations too)
  const T z = T(1) - t;
  w[1] = sixth(t * t * (t - T(2)) * T(3) + T(4));
  w[2] = sixth(z * z * (z - T(2)) * T(3) + T(4));
  w[0] = sixth(z * z * z);
  w[3] = T(1) - w[0] - w[1] - w[2];
}
// one row of the 4 x 4 tap sum, and its accumulation: explicit fused multiply-adds in ONE fixed order, so that the same taps
// give the same bits whichever kernel or code path gathers them (from L1 / L2, or from the LDS window of invert_tile_kernel)
template <class T>
__device__ __forceinline__ T tap_row(const T (&wy)[4], T t0, T t1, T t2, T t3) {
#pragma clang fp contract(off)
  T r = wy[0] * t0;
  r = fma(wy[1], t1, r);
  r = fma(wy[2], t2, r);
  r = fma(wy[3], t3, r);
  return r;
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
// four ADJACENT elements from element `o` on: one 16-byte (f64: 32-byte) load -- global memory takes it at 4-byte alignment --
// where four scalar gathers cost the texture-address unit four wave-instructions (the gathers of a round are bound by that
// unit, not by bytes)
template <class T> struct Tap4 { T v[4]; };
template <class T> __device__ __forceinline__ Tap4<T> gather4(const T* __restrict__ base, unsigned o) {
  return *reinterpret_cast<const Tap4<T>*>(reinterpret_cast<const char*>(base) + (unsigned)(o * (unsigned)sizeof(T)));
}
template <class T> __device__ __forceinline__ Tap4<T> gather4(const T* __restrict__ base, size_t o) {
  return *reinterpret_cast<const Tap4<T>*>(base + o);
}

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
  // (four scalar gathers per tap row, not one 16-byte load as interp_constant's interior path: measured, the fixed point
  //  got slower with it -- 13.3 -> 14.5 ms at 16384^2 -- its rounds are bound by instruction issue, not by the gathers)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int i = ix + a;
    i = i < 0 ? 0 : (i >= m0 ? m0 - 1 : i);
    const I row = (I)i * (I)m1;
    const I o0 = row + (I)cy[0], o1 = row + (I)cy[1], o2 = row + (I)cy[2], o3 = row + (I)cy[3];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const T* cr = coef[n];
      out[n] = fma(wx[a], tap_row(wy, gather(cr, o0), gather(cr, o1), gather(cr, o2), gather(cr, o3)), out[n]);
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
#pragma unroll
  for (int n = 0; n < NC; ++n) out[n] = T(0);
  if (ix >= 0 && iy >= 0 && ix + 3 < n0 && iy + 3 < n1) {
    // the 4 x 4 footprint inside the field -- all but a frame of pixels: no mirror arithmetic (eight integer remainders), the
    // four taps of a row in one load
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const I o = (I)(ix + a) * (I)n1 + (I)iy;
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const Tap4<T> q = gather4(coef[n], o);
        out[n] = fma(wx[a], tap_row(wy, q.v[0], q.v[1], q.v[2], q.v[3]), out[n]);
      }
    }
    return;
  }
  int cy[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) cy[b] = ext_index(iy + b, n1, EXT_MIRROR);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const I row = (I)ext_index(ix + a, n0, EXT_MIRROR) * (I)n1;
    const I o0 = row + (I)cy[0], o1 = row + (I)cy[1], o2 = row + (I)cy[2], o3 = row + (I)cy[3];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const T* cr = coef[n];
      out[n] = fma(wx[a], tap_row(wy, gather(cr, o0), gather(cr, o1), gather(cr, o2), gather(cr, o3)), out[n]);
    }
  }
}

// the same fixed point with scipy's mode='constant' (geometric_phase_analysis.py:248, :262 `mode=`): coefficients of the
// unpadded field, 0 outside it in every round but the last of the overlap variant, which passes cval=nan (:297-299)
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void invert_constant_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int n0, int n1,
                                                             int edge, int shift, int iters, int nan_last,
                                                             T* __restrict__ out, int wr0, int wc0, int wc1, int all_rounds) {
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int j = wc0 + blockIdx.x * 256 + threadIdx.x, i = wr0 + blockIdx.y;
  if (j >= wc1) return;
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge), yb = T(j - edge);
  T v[2];
  interp_constant<T, 2, WIDE>(coef, n0, n1, xb, yb, T(0), v);
  const T xs = xb - T(shift), ys = yb - T(shift);
  T pv[2] = {(T)__builtin_nan(""), (T)__builtin_nan("")};   // the iterate before v
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    const bool last_nan = nan_last && it == iters - 1;
    const T cval = last_nan ? (T)__builtin_nan("") : T(0);
    interp_constant<T, 2, WIDE>(coef, n0, n1, xs + v[0], ys + v[1], cval, nv);
    // a round that reproduces its input bit for bit is a fixed point: every later round returns the same numbers, so the
    // wavefront leaves once all its pixels are there (the same result as running all rounds; f32 fields settle after
    // ~15 of the reference's 36 rounds).  So is a CYCLE OF TWO: a few per cent of the pixels of an f32 field never reach a
    // bitwise fixed point but alternate between two neighbouring values (x_(k+1) = x_(k-1) bit for bit: the map is
    // deterministic, so the sequence alternates from there on), and one such pixel kept its whole wavefront in the loop
    // for all rounds -- 43 % of the wavefronts at 4096^2 (tools/lf_rounds.py).  A wavefront leaves once every pixel is at a
    // fixed point or in a cycle of two; the member of the cycle that the full count of rounds ends on follows from the
    // parity of the rounds left.  Longer cycles run to the end.  The cval = NaN round of invert_u_overlap is a
    // different function: still run, on the iterate the full count of rounds would hand it.
    const bool fixed = nv[0] == v[0] && nv[1] == v[1];
    const bool cyc2 = nv[0] == pv[0] && nv[1] == pv[1];
    if (!last_nan && !all_rounds && __all(fixed || cyc2)) {
      const int plain_left = (nan_last ? iters - 2 : iters - 1) - it;   // plain rounds the full count would still run
      if (!(plain_left & 1)) { v[0] = nv[0]; v[1] = nv[1]; }          // (odd: the cycle's other member, v itself)
      if (nan_last) interp_constant<T, 2, WIDE>(coef, n0, n1, xs + v[0], ys + v[1], (T)__builtin_nan(""), v);
      break;
    }
    pv[0] = v[0];
    pv[1] = v[1];
    v[0] = nv[0];
    v[1] = nv[1];
  }
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// u_it(r) <- u(r + u_it(r)), all rounds for one pixel (geometric_phase_analysis.py:291-299)
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void invert_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int m0, int m1,
                                                    int n0, int n1, int edge, int shift, int iters, T* __restrict__ out,
                                                    int wr0, int wc0, int wc1, int all_rounds) {
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int j = wc0 + blockIdx.x * 256 + threadIdx.x, i = wr0 + blockIdx.y;
  if (j >= wc1) return;
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge + NPAD), yb = T(j - edge + NPAD);
  T v[2];
  interp_nearest<T, 2, WIDE>(coef, m0, m1, xb, yb, v);
  // (shift != 0: invert_u, which samples every later round at r + u_it - shift, geometric_phase_analysis.py:258)
  const T xs = xb - T(shift), ys = yb - T(shift);
  T pv[2] = {(T)__builtin_nan(""), (T)__builtin_nan("")};
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    interp_nearest<T, 2, WIDE>(coef, m0, m1, xs + v[0], ys + v[1], nv);
    // every pixel of the wavefront at a bitwise fixed point or in a cycle of two: the later rounds are known (see above)
    const bool fixed = nv[0] == v[0] && nv[1] == v[1];
    const bool cyc2 = nv[0] == pv[0] && nv[1] == pv[1];
    if (!all_rounds && __all(fixed || cyc2)) {
      if (!((iters - 1 - it) & 1)) { v[0] = nv[0]; v[1] = nv[1]; }
      break;
    }
    pv[0] = v[0];
    pv[1] = v[1];
    v[0] = nv[0];
    v[1] = nv[1];
  }
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// The same fixed point on 16 x 16 pixel tiles with the coefficient window of the LATER rounds staged in LDS (round 5).
// Every round gathers 2 x 16 coefficients per pixel around r + u_it(r); the kernel above is bound by those gathers (L1 /
// texture-address rate and the dependent chain of rounds: 200 instructions per round issue in a quarter of the time a round
// takes).  The iterates of a tile move by hundreds of pixels in the first rounds where |u| is large, but contract by
// |grad u| per round; once no pixel of the tile moved by more than LMOVE = 8 samples in a round, the bounding box of the tile's
// sample points (+ spline support + a margin for what movement is left) is loaded into LDS ONCE -- both components
// interleaved, tap indices clamped while loading, exactly as mode='nearest' clamps them -- and the remaining rounds read
// their taps with ds_read_b64 (a lane whose 4 x 4 footprint leaves the window falls back to global memory for that round).
// The arithmetic per tap and its order are interp_nearest's: the results are bit-identical to invert_kernel's.
constexpr int LT = 16;        // tile side
constexpr int LW = 48;        // window side (samples): tile stretched by up to 1.5 + footprint 4 + margin 2 x 4
constexpr int LWP = LW + 1;   // row pitch (odd: the four rows of a wavefront start on different banks)
constexpr int LMARG = 4;
// the window is staged once no pixel of the tile moved by more than LMOVE samples in a round: with the iterates contracting
// by c = |grad u| per round what is left to move is <= LMOVE c / (1 - c) -- inside the margin for c <= 1/3 (a lane that does
// leave the window gathers from global memory for that round, same bits).  2 / 4 / 8 / 16: 8.42 / 8.01 / 7.74 / 7.36 ms at
// 16384^2 on the benchmark field (one global round fewer per step); 8 keeps the margin honest for rougher fields.
#ifndef GPA_LF_LMOVE
#define GPA_LF_LMOVE 8
#endif
constexpr int LMOVE = GPA_LF_LMOVE;
template <class T> struct T2 { T a, b; };

template <class T>
__device__ __forceinline__ bool interp_window(const T2<T>* __restrict__ win, int wx0, int wy0, int m0, int m1, T x, T y, T (&out)[2]) {
  x = x < T(-2) ? T(-2) : (x > T(m0 + 1) ? T(m0 + 1) : x);
  y = y < T(-2) ? T(-2) : (y > T(m1 + 1) ? T(m1 + 1) : y);
  const T fx = floor(x), fy = floor(y);
  const int rx = (int)fx - 1 - wx0, ry = (int)fy - 1 - wy0;     // window coordinates of the first tap
  if (!(rx >= 0 && ry >= 0 && rx + 3 < LW && ry + 3 < LW)) return false;    // (NaN coordinates: false as well)
  T wx[4], wy[4];
  bspline_weights(x - fx, wx);
  bspline_weights(y - fy, wy);
  out[0] = out[1] = T(0);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const T2<T>* cr = win + (rx + a) * LWP + ry;
    const T2<T> t0 = cr[0], t1 = cr[1], t2 = cr[2], t3 = cr[3];
    out[0] = fma(wx[a], tap_row(wy, t0.a, t1.a, t2.a, t3.a), out[0]);
    out[1] = fma(wx[a], tap_row(wy, t0.b, t1.b, t2.b, t3.b), out[1]);
  }
  return true;
}

template <class T, bool WIDE>
__global__ __launch_bounds__(256) void invert_tile_kernel(const T* __restrict__ c0, const T* __restrict__ c1, int m0, int m1,
                                                         int n0, int n1, int edge, int shift, int iters, T* __restrict__ out,
                                                         int wr0, int wc0, int wr1, int wc1, int all_rounds) {
  __shared__ T2<T> win[LW * LWP];
  __shared__ int box[4][4];                                // per wavefront: min x, min y, max x, max y of the first taps
  const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  const int i_raw = wr0 + blockIdx.y * LT + ti, j_raw = wc0 + blockIdx.x * LT + tj;
  const bool valid = i_raw < wr1 && j_raw < wc1;
  const int i = i_raw < wr1 ? i_raw : wr1 - 1, j = j_raw < wc1 ? j_raw : wc1 - 1;   // (lanes beyond the window shadow its last pixel)
  const T* const coef[2] = {c0, c1};
  const T xb = T(i - edge + NPAD), yb = T(j - edge + NPAD);
  T v[2];
  interp_nearest<T, 2, WIDE>(coef, m0, m1, xb, yb, v);
  const T xs = xb - T(shift), ys = yb - T(shift);
  int state = 0;        // 0: rounds from global memory, the workgroup still deciding; 1: window staged; 2: no window (does not fit)
  int wx0 = 0, wy0 = 0;
  T pv[2] = {(T)__builtin_nan(""), (T)__builtin_nan("")};   // the iterate before v
  for (int it = 0; it < iters; ++it) {
    T nv[2];
    const T x = xs + v[0], y = ys + v[1];
    if (state != 1 || !interp_window<T>(win, wx0, wy0, m0, m1, x, y, nv)) interp_nearest<T, 2, WIDE>(coef, m0, m1, x, y, nv);
    const bool fixed = nv[0] == v[0] && nv[1] == v[1];
    const bool cyc2 = nv[0] == pv[0] && nv[1] == pv[1];
    const bool moving = !(fabs(nv[0] - v[0]) <= T(LMOVE) && fabs(nv[1] - v[1]) <= T(LMOVE));   // (NaN: moving)
    if (state != 0 && !all_rounds && __all(fixed || cyc2)) {
      // every pixel of the wavefront has settled; a cycle's member after the full count of rounds follows from the parity
      if (!((iters - 1 - it) & 1)) { v[0] = nv[0]; v[1] = nv[1]; }
#ifdef GPA_LF_ROUNDS_DEBUG
      v[0] = T(it);
      v[1] = fixed ? T(1) : T(0);
#endif
      break;
    }
    pv[0] = v[0];
    pv[1] = v[1];
    v[0] = nv[0];
    v[1] = nv[1];
    if (state == 0) {
      // workgroup-uniform decision (every wavefront is still in the loop: none leaves before the decision is made)
      if (!__syncthreads_or(moving ? 1 : 0)) {
        T xn = xs + v[0], yn = ys + v[1];
        xn = xn < T(-2) ? T(-2) : (xn > T(m0 + 1) ? T(m0 + 1) : xn);
        yn = yn < T(-2) ? T(-2) : (yn > T(m1 + 1) ? T(m1 + 1) : yn);
        const int fx = (int)floor(xn) - 1, fy = (int)floor(yn) - 1;
        // bounding box of the tile's first taps: a shuffle tree per wavefront, the four wavefronts' results through LDS.
        // (atomicMin / atomicMax on one LDS word were what hipcc turns into a loop over the 64 lanes, readlane by readlane:
        //  ~1500 scalar instructions per wavefront, more than the kernel's vector instructions -- rocprofv3 SQ_INSTS_SALU)
        int lo_x = fx, lo_y = fy, hi_x = fx, hi_y = fy;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
          lo_x = min(lo_x, __shfl_xor(lo_x, m));
          lo_y = min(lo_y, __shfl_xor(lo_y, m));
          hi_x = max(hi_x, __shfl_xor(hi_x, m));
          hi_y = max(hi_y, __shfl_xor(hi_y, m));
        }
        if ((threadIdx.x & 63) == 0) {
          int* b = box[threadIdx.x >> 6];
          b[0] = lo_x; b[1] = lo_y; b[2] = hi_x; b[3] = hi_y;
        }
        __syncthreads();
        const int bx0 = min(min(box[0][0], box[1][0]), min(box[2][0], box[3][0])), by0 = min(min(box[0][1], box[1][1]), min(box[2][1], box[3][1]));
        const int bx1 = max(max(box[0][2], box[1][2]), max(box[2][2], box[3][2])), by1 = max(max(box[0][3], box[1][3]), max(box[2][3], box[3][3]));
        if (bx1 - bx0 + 4 + 2 * LMARG <= LW && by1 - by0 + 4 + 2 * LMARG <= LW) {
          // centre the box in the window; tap indices clamped while loading = mode 'nearest' clamping them at use
          wx0 = bx0 - (LW - (bx1 - bx0 + 4)) / 2;
          wy0 = by0 - (LW - (by1 - by0 + 4)) / 2;
          // (192 of the 256 threads: a thread keeps its column and walks down the rows four at a time -- no division, one
          //  clamp and one multiply per element; the loop over e = thread + 256 k with e / LW, e % LW was a fifth of the
          //  kernel's instructions)
          if (threadIdx.x < 4 * LW) {
            const int r0w = (int)threadIdx.x / LW, c = (int)threadIdx.x - r0w * LW;
            int gj = wy0 + c;
            gj = gj < 0 ? 0 : (gj >= m1 ? m1 - 1 : gj);
#pragma unroll 4
            for (int r = r0w; r < LW; r += 4) {
              int gi = wx0 + r;
              gi = gi < 0 ? 0 : (gi >= m0 ? m0 - 1 : gi);
              const size_t o = (size_t)gi * m1 + gj;
              win[r * LWP + c] = T2<T>{c0[o], c1[o]};
            }
          }
          state = 1;
        } else {
          state = 2;
        }
        __syncthreads();
      }
      continue;      // (no wavefront leaves while the workgroup may still meet at a barrier)
    }
#ifdef GPA_LF_ROUNDS_DEBUG
    if (it == iters - 1) { v[0] = T(it + 1); v[1] = fixed ? T(1) : T(0); }
#endif
  }
  if (!valid) return;
  out[(size_t)i * o1 + j] = v[0];
  out[(size_t)o0 * o1 + (size_t)i * o1 + j] = v[1];
}

// final resampling, map_coordinates defaults: order 3, mode='constant', cval = 0 (geometric_phase_analysis.py:969-973): out(r) =
// image(r + u_inv(r)).  One row of 256 pixels per workgroup: u_inv is smooth, so a wavefront's 64 x 16 taps are four runs of
// ~67 adjacent coefficients per tap row -- coalesced as they are.  (A 16 x 16 tile form with the coefficient window staged in
// LDS, as invert_tile_kernel has it, measured 2.5 x SLOWER here: one evaluation per pixel does not pay for the window.)
template <class T, bool WIDE>
__global__ __launch_bounds__(256) void warp_constant_kernel(const T* __restrict__ coef, int n0, int n1,
                                                           const T* __restrict__ uinv, T cval, T* __restrict__ out,
                                                           int wr0, int wc0, int wc1) {
  const int j = wc0 + blockIdx.x * 256 + threadIdx.x, i = wr0 + blockIdx.y;
  if (j >= wc1) return;
  const size_t o = (size_t)i * n1 + j, npx = (size_t)n0 * n1;
  const T* const cf[1] = {coef};
  T r[1];
  interp_constant<T, 1, WIDE>(cf, n0, n1, T(i) + uinv[o], T(j) + uinv[npx + o], cval, r);
  out[o] = r[0];
}

// the prefilter's taps, uploaded once per workspace and precision (pinned staging is not worth it: 65 values)
template <class T>
hipError_t ensure_taps(WarpWs* ws, hipStream_t s) {
  const int want = sizeof(T) == 4 ? 0 : 1;
  if (ws->taps && ws->taps_dtype == want) return hipSuccess;
  constexpr int KT = TapHalf<T>::value;
  const double z = sqrt(3.0) - 2.0;
  T h[2 * KT + 1];
  for (int k = -KT; k <= KT; ++k) h[k + KT] = (T)((-6.0 * z / (1.0 - z * z)) * pow(z, abs(k)));
  hipError_t e = hipSuccess;
  if (!ws->taps) e = hipMalloc(&ws->taps, (2 * TapHalf<double>::value + 1) * sizeof(double));
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
  if (ws->counted) *ws->counted -= ws->cap;
  ws->buf = nullptr;
  ws->cap = 0;
  e = hipMalloc(&ws->buf, bytes);
  if (e != hipSuccess) return e;
  ws->cap = bytes;
  if (ws->counted) *ws->counted += bytes;
  return hipSuccess;
}

// coefficients of `in` (m0 x m1); tmp: same size.  npad >= 0: `in` is the unpadded (m0 - 2 npad) x (m1 - 2 npad) field, padded
// by edge replication and scaled on the fly (fir_rows_kernel<PADSRC>)
template <class T>
hipError_t prefilter(const T* in, int m0, int m1, int ext, const T* d_h, T* tmp, T* out, hipStream_t s, int npad = -1,
                     T scale = T(1)) {
  {
    GPA_PROF("fir_rows_kernel", s);
    const dim3 grid((m1 + FR_OUT - 1) / FR_OUT, m0);
    if (npad >= 0) fir_rows_kernel<T, true><<<grid, 256, 0, s>>>(in, m0, m1, ext, d_h, tmp, npad, m0 - 2 * npad, m1 - 2 * npad, scale);
    else fir_rows_kernel<T, false><<<grid, 256, 0, s>>>(in, m0, m1, ext, d_h, tmp, 0, m0, m1, T(1));
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
  T *tmp = buf + npx, *c0 = buf + 2 * npx, *c1 = buf + 3 * npx;
  for (int c = 0; c < 2 && e == hipSuccess; ++c)
    e = prefilter<T>(d_u + (size_t)c * npx, n0, n1, EXT_MIRROR, d_h, tmp, c == 0 ? c0 : c1, s, 0, scale);
  if (e == hipSuccess) {
    const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
    const int allr = opt_set(OPT_LF_ALL_ROUNDS) ? 1 : 0;   // (diagnostic: every round of the fixed point, no early exit)
    for (int q = 0; q < (nrect > 0 ? nrect : 1); ++q) {
    const Win v = window(nrect > 0 ? rects + 4 * q : nullptr, o0, o1);
    if (v.h <= 0 || v.w <= 0) continue;
    // nan_last: invert_u_overlap ends on a cval=nan round (geometric_phase_analysis.py:296-299); invert_u never passes
    // cval (:255-258) and keeps 0 outside -- said by the caller, not guessed from the geometry (invert_u with its
    // default edge = 0 has shift == 0 too)
    GPA_PROF("invert_kernel", s);
    if (npx * sizeof(T) < ((size_t)1 << 32))
      invert_constant_kernel<T, false><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, n0, n1, edge, shift, iters, nan_last ? 1 : 0, d_out,
                                                                                   v.r0, v.c0, v.c0 + v.w, allr);
    else
      invert_constant_kernel<T, true><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, n0, n1, edge, shift, iters, nan_last ? 1 : 0, d_out,
                                                                                  v.r0, v.c0, v.c0 + v.w, allr);
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
  T *tmp = buf + mp, *c0 = buf + 2 * mp, *c1 = buf + 3 * mp;
  // (scaling and the 12-sample edge padding of mode 'nearest' happen while the row pass loads its tiles)
  for (int c = 0; c < 2 && e == hipSuccess; ++c)
    e = prefilter<T>(d_u + (size_t)c * n0 * n1, m0, m1, EXT_REFLECT, d_h, tmp, c == 0 ? c0 : c1, s, NPAD, scale);
  if (e == hipSuccess) {
    const int o1 = n1 + 2 * edge, o0 = n0 + 2 * edge;
    const int allr = opt_set(OPT_LF_ALL_ROUNDS) ? 1 : 0;   // (diagnostic: every round of the fixed point, no early exit)
    for (int q = 0; q < (nrect > 0 ? nrect : 1); ++q) {
      const Win v = window(nrect > 0 ? rects + 4 * q : nullptr, o0, o1);
      if (v.h <= 0 || v.w <= 0) continue;
      GPA_PROF("invert_kernel", s);
      if (!opt_set(OPT_NO_LFTILE)) {
        // 16 x 16 tiles, the later rounds from an LDS window (NO_LFTILE: the row-segment kernel, every round from L1 / L2)
        const dim3 grid((v.w + LT - 1) / LT, (v.h + LT - 1) / LT);
        if (mp * sizeof(T) < ((size_t)1 << 32))
          invert_tile_kernel<T, false><<<grid, 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out, v.r0, v.c0, v.r0 + v.h, v.c0 + v.w, allr);
        else
          invert_tile_kernel<T, true><<<grid, 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out, v.r0, v.c0, v.r0 + v.h, v.c0 + v.w, allr);
      } else if (mp * sizeof(T) < ((size_t)1 << 32))
        invert_kernel<T, false><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out, v.r0, v.c0, v.c0 + v.w, allr);
      else
        invert_kernel<T, true><<<dim3((v.w + 255) / 256, v.h), 256, 0, s>>>(c0, c1, m0, m1, n0, n1, edge, shift, iters, d_out, v.r0, v.c0, v.c0 + v.w, allr);
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
      const dim3 grid((v.w + 255) / 256, v.h);
      if (npx * sizeof(T) < ((size_t)1 << 32))
        warp_constant_kernel<T, false><<<grid, 256, 0, s>>>(buf + npx, n0, n1, d_uinv, T(0), d_out, v.r0, v.c0, v.c0 + v.w);
      else
        warp_constant_kernel<T, true><<<grid, 256, 0, s>>>(buf + npx, n0, n1, d_uinv, T(0), d_out, v.r0, v.c0, v.c0 + v.w);
    }
    e = hipGetLastError();
  }
  return e;
}

}  // namespace

hipError_t warp_reserve_undistort(int dtype, int n0, int n1, WarpWs* ws, hipStream_t s) {
  const size_t rs = dtype == 0 ? 4 : 8;
  return reserve(ws, 4 * (size_t)(n0 + 2 * NPAD) * (n1 + 2 * NPAD) * rs + 2 * (size_t)n0 * n1 * rs, s);
}

void warp_ws_free(WarpWs* ws) {
  if (ws->buf) (void)hipFree(ws->buf);
  if (ws->buf && ws->counted) *ws->counted -= ws->cap;
  if (ws->taps) (void)hipFree(ws->taps);
  size_t* counted = ws->counted;
  *ws = WarpWs{};
  ws->counted = counted;
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
