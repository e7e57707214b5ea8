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

// the same for the difference of two phases, |x| < 2 pi: floor((x + pi) / 2 pi) is -1, 0 or 1 and follows from
// two comparisons -- no division, and (like NumPy's exact float modulo) no quotient rounding at the seams
template <class T>
__device__ __forceinline__ T wrap_phase_diff(T x) {
  const T t = x + Consts<T>::pi;
  const T r = t >= Consts<T>::two_pi ? t - Consts<T>::two_pi : (t < T(0) ? t + Consts<T>::two_pi : t);
  return r - Consts<T>::pi;
}

// wrap(d + step) for a phase difference d of two atan2 results and a CONSTANT step = hi + lo (raw winners, see
// reconstruct_setup_kernel), without the systematic rounding that wrap_phase_diff(d + hi) has: d is a multiple of the
// phases' own spacing (2.4e-7 for |phase| >= 2), so `d + hi + pi` would round the off-grid constant the same way at every
// pixel -- a bias of up to 1e-7 rad per column, i.e. a ramp across the image.  Here d is wrapped by comparisons (the
// subtractions are exact), the step is added where the sum is small (d ~ -step: exact or rounded at its own fine
// spacing), wrapped again the same way (only differences near +-pi get there) and the low part goes on last.
template <class T>
__device__ __forceinline__ T wrap_diff_plus_step(T d, T hi, T lo) {
  const T pi = Consts<T>::pi, two_pi = Consts<T>::two_pi;
  d = d >= pi ? d - two_pi : (d < -pi ? d + two_pi : d);
  T x = d + hi;
  x = x >= pi ? x - two_pi : (x < -pi ? x + two_pi : x);
  return x + lo;
}

// circumference of a full turn as this file's phases measure it (atan2 with the precision's own pi constants, wraps by
// Consts<T>::two_pi) over the true 2 pi: 1 + 2.78e-8 in f32, 1 in f64
template <class T> __device__ __forceinline__ double unwrap_turn_scale() {
  return (double)Consts<T>::two_pi / 6.28318530717958647692;
}

// general argument, fast path for the usual |x| < 2 pi (same arithmetic as wrap_to_pi there)
template <class T>
__device__ __forceinline__ T wrap_to_pi_fast(T x) {
  return fabs(x) < Consts<T>::two_pi ? wrap_phase_diff(x) : wrap_to_pi(x);
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

// Tile pipeline (gpa_tile_gradients_*): only the INTERIOR of a halo window is kept, and the owner of
// displacement component c is sent the block (du_c/dx, du_c/dy, weight) of every tile.  With `on` the kernel stores the
// interior pixels straight into those blocks instead of full-window fields that a copy kernel would cut the interiors
// out of (round 4: one launch and 160 MB of traffic less per 2048^2 window; the values are the same).
struct TileOut {
  int on;
  int i0, j0, t0, t1;   // interior rectangle of the window: origin and size
  int wx, hy;           // columns of the x-difference fields / rows of the y-difference fields that exist inside it
  void *dx0, *dx1, *dy0, *dy1, *w0, *w1;   // destinations of (du_0/dx, du_1/dx, du_0/dy, du_1/dy, weight, weight again or null)
  size_t dxp, dyp, wp;  // their row pitches in elements
};

// One thread per column sliding down a band of REC_ROWS rows: the phase of every lock-in
// sample is evaluated once (atan2 is the expensive part) -- the right neighbour's phase comes
// from the next lane, the lower neighbour's from the next row, which becomes the current row
// of the following step.
template <class T, int P>
__global__ __launch_bounds__(256) void reconstruct_kernel(const cpx<T>* __restrict__ lockin,
                                                         const double* __restrict__ kmat, int n0, int n1,
                                                         int border, T* __restrict__ dudx, T* __restrict__ dudy,
                                                         T* __restrict__ wnorm, const double* __restrict__ ystep,
                                                         const TileOut tile) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  const int x0 = blockIdx.y * REC_ROWS;
  const int x1 = x0 + REC_ROWS < n0 ? x0 + REC_ROWS : n0;
  const int lane = threadIdx.x & 63;
  const bool act = y < n1;
  const int yc = act ? y : n1 - 1;
  const size_t npx = (size_t)n0 * n1;
  const bool has_r = act && yc + 1 < n1;
  T k0[P], k1[P], cy[P], cyl[P];   // cy + cyl: phase step along y of a compensation phasor the lock-ins lack (see reconstruct_setup_kernel)
#pragma unroll
  for (int p = 0; p < P; ++p) {
    k0[p] = (T)kmat[2 * p];
    k1[p] = (T)kmat[2 * p + 1];
    const double c = ystep ? ystep[p] * unwrap_turn_scale<T>() : 0.0;
    cy[p] = (T)c;
    cyl[p] = (T)(c - (double)cy[p]);
  }
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
      bx[p] = has_r ? (ystep ? wrap_diff_plus_step(phr - phc[p], cy[p], cyl[p]) : wrap_phase_diff(phr - phc[p])) : T(0);
      by[p] = has_d ? wrap_phase_diff(phn[p] - phc[p]) : T(0);
    }
    // (tile mode: this pixel's place in the interior rectangle)
    const int xi = x - tile.i0, yj = yc - tile.j0;
    const bool tin = tile.on && xi >= 0 && xi < tile.t0 && yj >= 0 && yj < tile.t1;
    if (act && (!tile.on || tin)) {
      if (!tile.on) {
        if (wnorm) wnorm[o] = sqrt(wsq);
      } else {
        const T wv = sqrt(wsq);
        ((T*)tile.w0)[(size_t)xi * tile.wp + yj] = wv;
        if (tile.w1) ((T*)tile.w1)[(size_t)xi * tile.wp + yj] = wv;
      }
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
      if (has_r && (!tile.on || yj < tile.wx)) {
        T s0, s1;
        solve2(a00, a01, a11, rx0, rx1, s0, s1);
        if (!tile.on) {
          const size_t ox = (size_t)x * (n1 - 1) + yc, plane = (size_t)n0 * (n1 - 1);
          dudx[ox] = s0;
          dudx[plane + ox] = s1;
        } else {
          ((T*)tile.dx0)[(size_t)xi * tile.dxp + yj] = s0;
          ((T*)tile.dx1)[(size_t)xi * tile.dxp + yj] = s1;
        }
      }
      if (has_d && (!tile.on || xi < tile.hy)) {
        T s0, s1;
        solve2(a00, a01, a11, ry0, ry1, s0, s1);
        if (!tile.on) {
          const size_t plane = (size_t)(n0 - 1) * n1;
          dudy[o] = s0;
          dudy[plane + o] = s1;
        } else {
          ((T*)tile.dy0)[(size_t)xi * tile.dyp + yj] = s0;
          ((T*)tile.dy1)[(size_t)xi * tile.dyp + yj] = s1;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) { phc[p] = phn[p]; ampc[p] = ampn[p]; }
  }
}

// ---- fused driver: a5 + a6 + the unwrap's setup in one pass --------------------------------------
// reconstruct_kernel followed by the unwrap's setup_kernel (gpa_unwrap.hip) writes the four
// gradient fields to HBM only for the next kernel to read them back.  Here the weighted, wrapped
// gradient of every edge is consumed where it is produced: the kernel emits wnorm and, for both
// displacement components, r0 = div(min(w^2) wrap(grad u_c)) (phase_unwrap.py:326-331 on the
// weighted-prediff inputs of geometric_phase_analysis.py:234-240) plus the partial sums of ||r0||^2.
// A wavefront owns 62 columns (lane 0 / lane 63 are a left / right halo column, so every neighbour
// comes from a lane shuffle); a workgroup slides down FUSED_ROWS rows after one halo row.
// Arithmetic identical to the two separate kernels (same values, same order).
constexpr int FUSED_ROWS = 32, FUSED_COLS = 62;

template <class T, int P>
__global__ __launch_bounds__(256) void reconstruct_setup_kernel(const cpx<T>* __restrict__ lockin,
                                                               const double* __restrict__ kmat, int n0, int n1,
                                                               int border, T* __restrict__ wnorm, T* __restrict__ r0,
                                                               T* __restrict__ r1, double* __restrict__ part0,
                                                               double* __restrict__ part1, int band, size_t rstride,
                                                               size_t pstride, const double* __restrict__ ystep) {
  {
    // image stacks: blockIdx.z = image; its lock-ins / weight follow those of the image before, its two residuals and
    // partial-sum blocks lie rstride elements / pstride doubles behind the previous image's (the batched workspace)
    const size_t img = blockIdx.z, npx_ = (size_t)n0 * n1;
    lockin += img * P * npx_;
    wnorm += img * npx_;
    r0 += img * rstride;
    r1 += img * rstride;
    part0 += img * pstride;
    part1 += img * pstride;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int yw = (blockIdx.x * 4 + wave) * FUSED_COLS + lane - 1;
  const bool incol = yw >= 0 && yw < n1;
  const int yc = yw < 0 ? 0 : (yw >= n1 ? n1 - 1 : yw);
  const bool outcol = incol && lane >= 1 && lane <= FUSED_COLS;
  const int x0 = blockIdx.y * band;
  const int x1 = x0 + band < n0 ? x0 + band : n0;
  const int xs = x0 > 0 ? x0 - 1 : x0;
  const size_t npx = (size_t)n0 * n1;
  const bool has_r = incol && yw + 1 < n1;
  T k0[P], k1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { k0[p] = (T)kmat[2 * p]; k1[p] = (T)kmat[2 * p + 1]; }
  // lock-ins handed over WITHOUT the compensation phasor exp(i ystep_p y) of geometric_phase_analysis.py:683 (the shared
  // pass B in raw mode): a unit phasor that depends on y alone changes no amplitude and no phase difference along x, and
  // adds the constant ystep_p to every phase difference along y before it is wrapped
  // Three details keep this free of SYSTEMATIC f32 error (a constant bias of 1e-8 rad per column is a ramp of 1e-5 .. 1e-4 px
  // across an image after the unwrap; measured on a 512 x 2048 image, mean error of du/dy: 2e-10 px per column compensated,
  // 7e-8 with the naive wrap(d + step), 2e-8 .. 4.5e-8 without the first point, 1e-10 with all three -- tools/raw_bias_probe.py):
  //  * the raw phase turns once every 1 / (ky + s/16) columns, and a full turn of phases that atan2f and the wraps measure
  //    closes at the f32 value of 2 pi, 1.75e-7 above 2 pi: the step is scaled by that ratio (unwrap_turn_scale);
  //  * the step is held as hi + lo of the data's precision: one rounded f32 would bias every difference of a peak alike;
  //  * the constant is added AFTER the difference has been wrapped exactly (wrap_diff_plus_step).
  T cy[P], cyl[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double c = ystep ? ystep[p] * unwrap_turn_scale<T>() : 0.0;
    cy[p] = (T)c;
    cyl[p] = (T)(c - (double)cy[p]);
  }
  auto mask_fac = [&](int x, int y) {
    const bool inside = x >= border && x < n0 - border && y >= border && y < n1 - border;
    return (inside ? T(1) : T(0)) + T(1e-6);
  };
  T phc[P], ampc[P], phn[P], ampn[P];
  T wsq_c = T(0);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const cpx<T> c = lockin[p * npx + (size_t)xs * n1 + yc];
    phc[p] = atan2(c.y, c.x);
    ampc[p] = sqrt(c.x * c.x + c.y * c.y);
    phn[p] = T(0);
    ampn[p] = T(0);
    const T w = ampc[p] * mask_fac(xs, yc);
    wsq_c += w * w;
  }
  T fyu0 = T(0), fyu1 = T(0);
  double sq0 = 0, sq1 = 0;
  for (int x = xs; x < x1; ++x) {
    const size_t o = (size_t)x * n1 + yc;
    const bool has_d = x + 1 < n0, outrow = x >= x0;
    T wsq_n = T(0);
    if (has_d) {
      const T mn = mask_fac(x + 1, yc);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const cpx<T> c = lockin[p * npx + o + n1];
        phn[p] = atan2(c.y, c.x);
        ampn[p] = sqrt(c.x * c.x + c.y * c.y);
        const T w = ampn[p] * mn;
        wsq_n += w * w;
      }
    }
    // the unwrap weight is wnorm = sqrt(sum w^2) as stored; its square is what setup_kernel uses
    const T tc = sqrt(wsq_c), tn = sqrt(wsq_n);
    const T wwc = tc * tc, wwn = tn * tn;
    const T wwr = __shfl_down(wwc, 1);
    const T mfac = mask_fac(x, yc);
    T w[P], bx[P], by[P];
    T wmax = T(0);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const T phr = __shfl_down(phc[p], 1);
      w[p] = ampc[p] * mfac;
      wmax = w[p] > wmax ? w[p] : wmax;
      bx[p] = has_r ? (ystep ? wrap_diff_plus_step(phr - phc[p], cy[p], cyl[p]) : wrap_phase_diff(phr - phc[p])) : T(0);
      by[p] = has_d ? wrap_phase_diff(phn[p] - phc[p]) : T(0);
    }
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
    T fx0 = T(0), fx1 = T(0), fy0 = T(0), fy1 = T(0);
    if (outrow && has_r) {   // (the halo row only supplies its down edges)
      T s0, s1;
      solve2(a00, a01, a11, rx0, rx1, s0, s1);
      const T m = wwr < wwc ? wwr : wwc;
      fx0 = wrap_to_pi_fast(s0) * m;
      fx1 = wrap_to_pi_fast(s1) * m;
    }
    if (has_d) {
      T s0, s1;
      solve2(a00, a01, a11, ry0, ry1, s0, s1);
      const T m = wwn < wwc ? wwn : wwc;
      fy0 = wrap_to_pi_fast(s0) * m;
      fy1 = wrap_to_pi_fast(s1) * m;
    }
    const T fxl0 = __shfl_up(fx0, 1), fxl1 = __shfl_up(fx1, 1);
    if (outrow && outcol) {
      const T v0 = fx0 - fxl0 + fy0 - fyu0, v1 = fx1 - fxl1 + fy1 - fyu1;
      wnorm[o] = tc;
      r0[o] = v0;
      r1[o] = v1;
      sq0 += (double)v0 * (double)v0;
      sq1 += (double)v1 * (double)v1;
    }
    fyu0 = fy0;
    fyu1 = fy1;
    wsq_c = wsq_n;
#pragma unroll
    for (int p = 0; p < P; ++p) { phc[p] = phn[p]; ampc[p] = ampn[p]; }
  }
  // deterministic block sums (fixed shuffle tree, then fixed order over the four wavefronts)
  __shared__ double sh[8];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sq0 += __shfl_down(sq0, off); sq1 += __shfl_down(sq1, off); }
  if (lane == 0) { sh[wave] = sq0; sh[4 + wave] = sq1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    part0[blk] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    part1[blk] = ((sh[4] + sh[5]) + sh[6]) + sh[7];
  }
}

template <class T>
static hipError_t launch_reconstruct_setup_t(const void* lockin, const double* kmat, int P, int n0, int n1, int border,
                                             void* wnorm, void* r0, void* r1, double* part0, double* part1, int* nparts,
                                             hipStream_t s, int nimg, size_t rstride, size_t pstride, const double* ystep) {
  // rows per workgroup band: FUSED_ROWS for large images (one halo row per band is recomputed), fewer while the
  // grid would leave most of the 256 CUs idle -- a band is a serial loop of ~1.5 us per row
  int band = FUSED_ROWS;
  const int gx = (n1 + 4 * FUSED_COLS - 1) / (4 * FUSED_COLS);
  while (band > 2 && gx * ((n0 + band - 1) / band) < 1024) band /= 2;
  dim3 grid(gx, (n0 + band - 1) / band, nimg);
  *nparts = (int)(grid.x * grid.y);
  GPA_PROF("reconstruct_setup_kernel", s);
#define RS_CASE(PP)                                                                                                   \
  case PP:                                                                                                            \
    reconstruct_setup_kernel<T, PP><<<grid, 256, 0, s>>>((const cpx<T>*)lockin, kmat, n0, n1, border, (T*)wnorm,       \
                                                         (T*)r0, (T*)r1, part0, part1, band, rstride, pstride, ystep); \
    break;
  switch (P) {
    RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef RS_CASE
  return hipGetLastError();
}

hipError_t launch_reconstruct_setup(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1,
                                    int border, void* wnorm, void* r0, void* r1, double* part0, double* part1,
                                    int* nparts, hipStream_t s, int nimg, size_t rstride, size_t pstride, const double* ystep) {
  if (P > MAXP || P < 2) return hipErrorInvalidValue;
  return dtype == 0 ? launch_reconstruct_setup_t<float>(lockin, kmat, P, n0, n1, border, wnorm, r0, r1, part0, part1, nparts, s,
                                                        nimg, rstride, pstride, ystep)
                    : launch_reconstruct_setup_t<double>(lockin, kmat, P, n0, n1, border, wnorm, r0, r1, part0, part1, nparts, s,
                                                         nimg, rstride, pstride, ystep);
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

// reconstruct_u_inv_from_phases(pre_diff=True), geometric_phase_analysis.py:228-237: g[p][x][y][0] / [1] are the
// given phase gradients along axis 1 / axis 0; both are wrapped and solved against 2 pi kvecs with the weights of
// their own pixel, dudx keeps columns [0, n1-1), dudy rows [0, n0-1); wnorm = || weights ||_2 over the peaks
template <class T>
__global__ __launch_bounds__(256) void prediff_kernel(const T* __restrict__ g, const T* __restrict__ w,
                                                     const double* __restrict__ kmat, int P, int n0, int n1,
                                                     T* __restrict__ dudx, T* __restrict__ dudy, T* __restrict__ wnorm) {
  const int y = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y;
  if (y >= n1) return;
  const size_t npx = (size_t)n0 * n1, o = (size_t)x * n1 + y;
  T wv[MAXP], wmax = T(0), wsq = T(0);
  for (int p = 0; p < P; ++p) {
    wv[p] = w[p * npx + o];
    const T a = wv[p] < T(0) ? -wv[p] : wv[p];
    wmax = a > wmax ? a : wmax;
  }
  const T ws = wmax > T(0) ? T(1) / wmax : T(0);
  T a00 = 0, a01 = 0, a11 = 0, rx0 = 0, rx1 = 0, ry0 = 0, ry1 = 0;
  for (int p = 0; p < P; ++p) {
    const T k0 = (T)kmat[2 * p], k1 = (T)kmat[2 * p + 1], wn = wv[p] * ws, ww = wn * wn;
    const T bx = wrap_to_pi(g[2 * (p * npx + o)]), by = wrap_to_pi(g[2 * (p * npx + o) + 1]);
    wsq += wn * wn;
    a00 += ww * k0 * k0; a01 += ww * k0 * k1; a11 += ww * k1 * k1;
    rx0 += ww * k0 * bx; rx1 += ww * k1 * bx;
    ry0 += ww * k0 * by; ry1 += ww * k1 * by;
  }
  T s0, s1;
  if (y + 1 < n1) {
    solve2(a00, a01, a11, rx0, rx1, s0, s1);
    dudx[(size_t)x * (n1 - 1) + y] = s0;
    dudx[(size_t)n0 * (n1 - 1) + (size_t)x * (n1 - 1) + y] = s1;
  }
  if (x + 1 < n0) {
    solve2(a00, a01, a11, ry0, ry1, s0, s1);
    dudy[o] = s0;
    dudy[(size_t)(n0 - 1) * n1 + o] = s1;
  }
  if (wnorm) wnorm[o] = sqrt(wsq) * wmax;
}

hipError_t launch_prediff(int dtype, const void* grads, const void* w, const double* kmat, int P, int n0, int n1,
                          void* dudx, void* dudy, void* wnorm, hipStream_t s) {
  if (P > MAXP) return hipErrorInvalidValue;
  dim3 grid((n1 + 255) / 256, n0);
  if (dtype == 0)
    prediff_kernel<float><<<grid, 256, 0, s>>>((const float*)grads, (const float*)w, kmat, P, n0, n1, (float*)dudx,
                                               (float*)dudy, (float*)wnorm);
  else
    prediff_kernel<double><<<grid, 256, 0, s>>>((const double*)grads, (const double*)w, kmat, P, n0, n1, (double*)dudx,
                                                (double*)dudy, (double*)wnorm);
  return hipGetLastError();
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

// ---- f-2: phase gradients -> Jacobian -> lattice properties --------------------------------
// J[n,m,i,j] = (per-pixel weighted lstsq of grads[:, n, m, j])_i / nmperpixel: phasegradient2J with
// iso_ref=False (property_extract.py:69-101).  One thread per pixel, both solves share the normal
// matrix; grads are read as (d/dx, d/dy) pairs, J leaves as one 4-element store.
template <class T> struct alignas(2 * sizeof(T)) Pair2 { T x, y; };
template <class T> struct alignas(4 * sizeof(T)) Quad { T a, b, c, d; };

// shift.on: gradients are taken relative to the isotropic lattice kvecs + dks, g <- wrapToPi(g - 2 pi dk)
// (iso_ref=True, :87-92); kmat then already holds 2 pi (kvecs + dks).
struct GradShift { double v[2 * MAXP]; int on; };

template <class T>
__global__ __launch_bounds__(256) void jacobian_kernel(const Pair2<T>* __restrict__ grads, const T* __restrict__ w,
                                                      const double* __restrict__ kmat, int P, size_t npx,
                                                      T inv_nm, GradShift shift, Quad<T>* __restrict__ J) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= npx) return;
  T wv[MAXP], wmax = T(0);
  for (int p = 0; p < P; ++p) { wv[p] = w[p * npx + o]; const T a = wv[p] < T(0) ? -wv[p] : wv[p]; wmax = a > wmax ? a : wmax; }
  const T ws = wmax > T(0) ? T(1) / wmax : T(0);
  T a00 = 0, a01 = 0, a11 = 0, rx0 = 0, rx1 = 0, ry0 = 0, ry1 = 0;
  for (int p = 0; p < P; ++p) {
    const T k0 = (T)kmat[2 * p], k1 = (T)kmat[2 * p + 1], wn = wv[p] * ws, ww = wn * wn;
    Pair2<T> g = grads[p * npx + o];
    if (shift.on) {
      g.x = wrap_to_pi(g.x - (T)shift.v[2 * p]);
      g.y = wrap_to_pi(g.y - (T)shift.v[2 * p + 1]);
    }
    a00 += ww * k0 * k0; a01 += ww * k0 * k1; a11 += ww * k1 * k1;
    rx0 += ww * k0 * g.x; rx1 += ww * k1 * g.x;
    ry0 += ww * k0 * g.y; ry1 += ww * k1 * g.y;
  }
  T ux0, ux1, uy0, uy1;
  solve2(a00, a01, a11, rx0, rx1, ux0, ux1);   // d u_i / d axis 0
  solve2(a00, a01, a11, ry0, ry1, uy0, uy1);   // d u_i / d axis 1
  J[o] = Quad<T>{ux0 * inv_nm, uy0 * inv_nm, ux1 * inv_nm, uy1 * inv_nm};
}

// (angle, aniangle, alpha, kappa) per pixel from the 2x2 Jacobian of the lattice transformation:
// props_from_Jac (property_extract.py:137-178).  The reference takes a LAPACK SVD and normalises
// the signs of U; for a 2x2 matrix the same quantities come in closed form from
//   E,H = (a+d)/2, (c-b)/2   (rotation part),   F,G = (a-d)/2, (c+b)/2   (shear part):
//   s0,s1 = Q+R, |Q-R| with Q=|(E,H)|, R=|(F,G)|; U = rot((atan2(H,E)+atan2(G,F))/2);
//   U V^T = rot(atan2(H,E)) for det>0 and the reflection of angle -atan2(G,F) for det<0.
template <class T>
__global__ __launch_bounds__(256) void props_kernel(const Quad<T>* __restrict__ jac, size_t npx, T ident,
                                                   T refangle, T refscale, int diff, T* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= npx) return;
  const Quad<T> j = jac[o];
  T a = j.a + ident, b = j.b, c = j.c, d = j.d + ident;
  // scale by the largest entry: the squares below must not leave the range of T
  const T m = fmax(fmax(fabs(a), fabs(b)), fmax(fabs(c), fabs(d)));
  const T im = m > T(0) ? T(1) / m : T(0);
  a *= im; b *= im; c *= im; d *= im;
  const T E = T(0.5) * (a + d), F = T(0.5) * (a - d), G = T(0.5) * (c + b), H = T(0.5) * (c - b);
  const T Q = sqrt(E * E + H * H), R = sqrt(F * F + G * G);
  const T s0 = Q + R, s1 = fabs(Q - R);
  const T a1 = atan2(G, F), a2 = atan2(H, E);
  const T deg = T(180) / Consts<T>::pi;
  const T angle = (Q >= R ? a2 : -a1) * deg;
  T ani = T(-0.5) * (a1 + a2) * deg + (diff ? T(90) : T(0));
  ani -= T(180) * floor(ani / T(180));
  out[o] = angle + refangle;
  out[npx + o] = ani;
  out[2 * npx + o] = (diff ? s0 : s1) * m * refscale;
  out[3 * npx + o] = s0 / s1;
}

// |z| of a complex array (np.abs of the lock-ins: the weights of phasegradient2J), two values per thread
template <class T>
__global__ __launch_bounds__(256) void cabs_kernel(const cpx<T>* __restrict__ z, size_t count, T* __restrict__ out) {
  const size_t i = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x);
  if (i + 1 < count) {
    struct alignas(2 * sizeof(cpx<T>)) Two { cpx<T> a, b; };
    struct alignas(2 * sizeof(T)) Out { T a, b; };
    const Two v = *reinterpret_cast<const Two*>(z + i);
    *reinterpret_cast<Out*>(out + i) = Out{(T)hypot(v.a.x, v.a.y), (T)hypot(v.b.x, v.b.y)};
  } else if (i < count) {
    out[i] = (T)hypot(z[i].x, z[i].y);
  }
}
hipError_t launch_cabs(int dtype, const void* z, size_t count, void* out, hipStream_t s) {
  const unsigned grid = (unsigned)((count / 2 + 256) / 256);
  GPA_PROF("lockin_abs_kernel", s);
  if (dtype == 0) cabs_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)z, count, (float*)out);
  else cabs_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)z, count, (double*)out);
  return hipGetLastError();
}

hipError_t launch_jacobian(int dtype, const void* grads, const void* w, const double* kmat, int P, size_t npx,
                           double nmperpixel, const double* dks, void* J, hipStream_t s) {
  if (P > MAXP || P < 2) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)((npx + 255) / 256);
  GradShift shift{};
  shift.on = dks != nullptr;
  for (int i = 0; dks && i < 2 * P; ++i) shift.v[i] = 6.283185307179586476925 * dks[i];
  GPA_PROF("jacobian_kernel", s);
  if (dtype == 0)
    jacobian_kernel<float><<<grid, 256, 0, s>>>((const Pair2<float>*)grads, (const float*)w, kmat, P, npx,
                                                (float)(1.0 / nmperpixel), shift, (Quad<float>*)J);
  else
    jacobian_kernel<double><<<grid, 256, 0, s>>>((const Pair2<double>*)grads, (const double*)w, kmat, P, npx,
                                                 1.0 / nmperpixel, shift, (Quad<double>*)J);
  return hipGetLastError();
}

hipError_t launch_props(int dtype, const void* jac, size_t npx, int add_identity, double refangle, double refscale,
                        int diff, void* out, hipStream_t s) {
  const unsigned grid = (unsigned)((npx + 255) / 256);
  GPA_PROF("props_kernel", s);
  if (dtype == 0)
    props_kernel<float><<<grid, 256, 0, s>>>((const Quad<float>*)jac, npx, add_identity ? 1.f : 0.f, (float)refangle,
                                             (float)refscale, diff, (float*)out);
  else
    props_kernel<double><<<grid, 256, 0, s>>>((const Quad<double>*)jac, npx, add_identity ? 1.0 : 0.0, refangle,
                                              refscale, diff, (double*)out);
  return hipGetLastError();
}

// ---- f-4: robust plane fit (mathtools.py:30-47) -----------------------------------------------
// The reference minimises sum rho_huber(r^2) of r = image - (a0 x + a1 y + a2) with
// scipy.optimize.least_squares(loss='huber', f_scale=1).  That cost is convex; its minimiser is the
// fixed point of iteratively reweighted least squares with w = 1 for |r| <= 1 and 1/|r| beyond.
// One pass over the image per iteration accumulates the weighted normal equations in centred, scaled
// coordinates (u = (x - cx) / sx, v = (y - cy) / sy) in double; the 3x3 solve is the host's.
// sums: 0..5 = w*(uu, uv, u, vv, v, 1), 6..8 = w*(u z, v z, z), 9 = cost = sum 0.5 rho
// (round 6: workgroup b takes rows b, b + grid, ...: the row index and its coordinate are wave-uniform, no 64-bit division per
//  pixel, up to 1024 workgroups instead of 256 -- 124 -> ~20 us per pass at 4096^2; the sums keep a fixed order)
constexpr int HUBER_PARTS = 1024;
template <class T>
__global__ __launch_bounds__(256) void huber_moments_kernel(const T* __restrict__ img, int n0, int n1, double a0,
                                                           double a1, double a2, double cx, double cy, double isx,
                                                           double isy, double* __restrict__ part) {
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  for (int x = blockIdx.x; x < n0; x += gridDim.x) {
    const double u = (x - cx) * isx, pu = a0 * u + a2;
    const T* row = img + (size_t)x * n1;
    double su = 0.0, sw = 0.0, suz = 0.0;   // sums that carry the row's constant u: multiplied in once per row
    for (int y = threadIdx.x; y < n1; y += 256) {
      const double v = (y - cy) * isy, z = (double)row[y];
      const double r = z - (a1 * v + pu), ar = fabs(r);
      const double w = ar <= 1.0 ? 1.0 : 1.0 / ar;
      sw += w; su += w * v; suz += w * z;
      acc[3] += w * v * v; acc[7] += w * v * z;
      acc[9] += ar <= 1.0 ? 0.5 * r * r : ar - 0.5;
    }
    acc[0] += sw * u * u; acc[1] += su * u; acc[2] += sw * u; acc[4] += su; acc[5] += sw;
    acc[6] += suz * u; acc[8] += suz;
  }
  __shared__ double sh[256];
  for (int k = 0; k < 10; ++k) {
    sh[threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)k * gridDim.x + blockIdx.x] = sh[0];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void huber_final_kernel(const double* __restrict__ part, int nparts,
                                                         double* __restrict__ out) {
  __shared__ double sh[256];
  for (int k = 0; k < 10; ++k) {
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[(size_t)k * nparts + i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sh[0];
    __syncthreads();
  }
}

// scratch: >= 10 * HUBER_PARTS + 10 doubles; the ten sums land at scratch + 10 * HUBER_PARTS
int huber_sums_offset() { return 10 * HUBER_PARTS; }
hipError_t launch_huber_moments(int dtype, const void* img, int n0, int n1, const double* coef, double cx, double cy,
                                double sx, double sy, double* scratch, hipStream_t s) {
  const int nparts = n0 < HUBER_PARTS ? n0 : HUBER_PARTS;
  GPA_PROF("huber_moments_kernels", s);
  if (dtype == 0)
    huber_moments_kernel<float><<<nparts, 256, 0, s>>>((const float*)img, n0, n1, coef[0], coef[1], coef[2], cx, cy,
                                                       1.0 / sx, 1.0 / sy, scratch);
  else
    huber_moments_kernel<double><<<nparts, 256, 0, s>>>((const double*)img, n0, n1, coef[0], coef[1], coef[2], cx, cy,
                                                        1.0 / sx, 1.0 / sy, scratch);
  huber_final_kernel<<<1, 256, 0, s>>>(scratch, nparts, scratch + 10 * HUBER_PARTS);
  return hipGetLastError();
}

template <class T>
static hipError_t launch_reconstruct_t(const void* lockin, const double* kmat, int P, int n0, int n1, int border,
                                       void* dudx, void* dudy, void* wnorm, hipStream_t s, const double* ystep,
                                       const TileOut& tile) {
  dim3 grid((n1 + 255) / 256, (n0 + REC_ROWS - 1) / REC_ROWS);
#define REC_CASE(PP)                                                                                          \
  case PP:                                                                                                    \
    reconstruct_kernel<T, PP><<<grid, 256, 0, s>>>((const cpx<T>*)lockin, kmat, n0, n1, border, (T*)dudx, (T*)dudy, \
                                                   (T*)wnorm, ystep, tile);                                   \
    break;
  switch (P) {
    REC_CASE(2) REC_CASE(3) REC_CASE(4) REC_CASE(5) REC_CASE(6) REC_CASE(7) REC_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef REC_CASE
  return hipGetLastError();
}

hipError_t launch_reconstruct(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1,
                              int border, void* dudx, void* dudy, void* wnorm, hipStream_t s, const double* ystep) {
  if (P > MAXP || P < 2) return hipErrorInvalidValue;
  TileOut none{};
  return dtype == 0 ? launch_reconstruct_t<float>(lockin, kmat, P, n0, n1, border, dudx, dudy, wnorm, s, ystep, none)
                    : launch_reconstruct_t<double>(lockin, kmat, P, n0, n1, border, dudx, dudy, wnorm, s, ystep, none);
}

hipError_t launch_reconstruct_tile(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1, int border,
                                   int i0, int j0, int t0, int t1, void* const dx[2], size_t dx_pitch, void* const dy[2],
                                   size_t dy_pitch, void* const wn[2], size_t wn_pitch, hipStream_t s, const double* ystep) {
  if (P > MAXP || P < 2 || !dx[0] || !dx[1] || !dy[0] || !dy[1] || !wn[0]) return hipErrorInvalidValue;
  TileOut t{};
  t.on = 1;
  t.i0 = i0; t.j0 = j0; t.t0 = t0; t.t1 = t1;
  t.wx = t1 < n1 - 1 - j0 ? t1 : n1 - 1 - j0;
  t.hy = t0 < n0 - 1 - i0 ? t0 : n0 - 1 - i0;
  t.dx0 = dx[0]; t.dx1 = dx[1]; t.dy0 = dy[0]; t.dy1 = dy[1]; t.w0 = wn[0]; t.w1 = wn[1];
  t.dxp = dx_pitch; t.dyp = dy_pitch; t.wp = wn_pitch;
  GPA_PROF("reconstruct_kernel", s);
  return dtype == 0 ? launch_reconstruct_t<float>(lockin, kmat, P, n0, n1, border, nullptr, nullptr, nullptr, s, ystep, t)
                    : launch_reconstruct_t<double>(lockin, kmat, P, n0, n1, border, nullptr, nullptr, nullptr, s, ystep, t);
}

}  // namespace gpa
