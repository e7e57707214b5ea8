// The less travelled modes of pass B (see gpa_passb.h), in a translation unit of their own so that
// they compile beside the headline kernels instead of after them:
//   gated selection         -- wfr4 (geometric_phase_analysis.py:839-862)
//   selection + phases out  -- the a4 family: wfr2_grad_opt / wfr2_grad / wfr2_grad_vec (:722-836) and
//                              cuGPA.wfr2_grad_opt / wfr2_grad_single / wfr2_only_grad (cuGPA.py:41-133, :161-202)
// and the stencil kernel that turns (winner index, per-candidate phases) into the phase gradient.
#include "gpa_internal.h"
#include "gpa_passb.h"

namespace gpa {

// grad[x][y][:] of the winning candidate k = kidx[x][y] from psi_k = -angle(sf_k):
//   mode 0: np.gradient (central differences / 2, one-sided at the borders) along axis 0, axis 1
//   mode 1: forward differences along axis 0, axis 1, NaN at the last index (np.diff(..., append=nan),
//           cuGPA.py:58-62)
//   mode 2: the same with the components swapped: [..., 0] along axis 1, [..., 1] along axis 0 (the 'diff'
//           of wfr2_grad, geometric_phase_analysis.py:738-742)
// then + 2 pi (w - kref) of the matching axis and wrapToPi(2 g) / 2 (:807-812).  wfr2_grad differentiates the
// phase of the COMPENSATED lock-in and wraps per candidate; that is the same number modulo pi, i.e. the same
// result up to rounding (tests/test_oracle_golden.py::test_variants_gradient_spellings).
template <class T>
__global__ __launch_bounds__(256) void phasegrad_kernel(const T* __restrict__ psi, int K, const int32_t* __restrict__ kidx,
                                                       int n0, int n1, const double* __restrict__ kl,
                                                       const double* __restrict__ kr, int mode, T* __restrict__ grad,
                                                       const double* __restrict__ ystep) {
  const int y = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y;
  if (y >= n1) return;
  const size_t npx = (size_t)n0 * n1, o = (size_t)x * n1 + y;
  const int bi = kidx[o];
  T g0 = T(0), g1 = T(0);
  if (bi >= 0) {
    const T* pl = psi + (size_t)bi * npx;
    const T c = pl[o];
    const T nan = __builtin_nan("");
    if (mode == 0) {
      const T xm = x > 0 ? pl[o - n1] : c, xp = x + 1 < n0 ? pl[o + n1] : c;
      const T ym = y > 0 ? pl[o - 1] : c, yp = y + 1 < n1 ? pl[o + 1] : c;
      g0 = (x > 0 && x + 1 < n0) ? T(0.5) * (xp - xm) : (xp - xm);
      g1 = (y > 0 && y + 1 < n1) ? T(0.5) * (yp - ym) : (yp - ym);
    } else {
      g0 = x + 1 < n0 ? pl[o + n1] - c : nan;
      g1 = y + 1 < n1 ? pl[o + 1] - c : nan;
    }
    const T two_pi = T(6.28318530717958647692), pi = T(3.14159265358979323846);
    if (ystep) {
      // compensated form (the shared pass B's phases): psi is -angle of the lock-in compensated along x and lacking only the
      // candidate-independent phasor exp(i ystep y): the compensated phase differs by -ystep per column, and 2 pi (w - k) is in it
      g1 = (T)((double)g1 - ystep[0]);
    } else {
      g0 += (T)(6.28318530717958647692 * (kl[2 * bi] - kr[2 * bi]));
      g1 += (T)(6.28318530717958647692 * (kl[2 * bi + 1] - kr[2 * bi + 1]));
    }
    // wrapToPi(2 g) / 2 with the floored modulo of mathtools.py:72-75
    const T t0 = T(2) * g0 + pi, t1 = T(2) * g1 + pi;
    g0 = T(0.5) * (t0 - two_pi * floor(t0 / two_pi) - pi);
    g1 = T(0.5) * (t1 - two_pi * floor(t1 / two_pi) - pi);
  }
  grad[2 * o] = mode == 2 ? g1 : g0;
  grad[2 * o + 1] = mode == 2 ? g0 : g1;
}

hipError_t launch_phasegrad(int dtype, const void* psi, int K, const int32_t* kidx, int n0, int n1, const double* kl,
                            const double* kr, int mode, void* grad, hipStream_t s, const double* ystep) {
  dim3 grid((n1 + 255) / 256, n0);
  GPA_PROF("phasegrad_kernel", s);
  if (dtype == 0)
    phasegrad_kernel<float><<<grid, 256, 0, s>>>((const float*)psi, K, kidx, n0, n1, kl, kr, mode, (float*)grad, ystep);
  else
    phasegrad_kernel<double><<<grid, 256, 0, s>>>((const double*)psi, K, kidx, n0, n1, kl, kr, mode, (double*)grad, ystep);
  return hipGetLastError();
}

// ---- small images: the K candidates of a peak split over ksplit workgroups per row (PB_PART) + merge ------------
template <class T>
__global__ __launch_bounds__(256) void merge_parts_kernel(const cpx<T>* __restrict__ part, const int32_t* __restrict__ pidx,
                                                         int S, int P, int K, int n0, int n1, const cpx<T>* __restrict__ dx,
                                                         const cpx<T>* __restrict__ dy, cpx<T>* __restrict__ out,
                                                         int32_t* __restrict__ kidx) {
  const int y = blockIdx.x * 256 + threadIdx.x, x = blockIdx.y, p = blockIdx.z;
  if (y >= n1) return;
  const size_t o = ((size_t)p * n0 + x) * n1 + y, slab = (size_t)P * n0 * n1;
  cpx<T> best = {T(0), T(0)};
  int bi = -1;
  // the first four slabs (all there are today) are requested together: a loop of load -> compare pays one memory
  // round trip per slab, and this kernel runs where nothing hides them (small images)
  cpx<T> v4[4];
  int i4[4];
#pragma unroll
  for (int z = 0; z < 4; ++z) {
    const size_t zo = (size_t)(z < S ? z : 0) * slab + o;
    v4[z] = part[zo];
    i4[z] = pidx[zo];
  }
#pragma unroll
  for (int z = 0; z < 4; ++z) {
    // slabs hold increasing candidate ranges: the sequential rule "strictly larger replaces" carries over
    const T a = v4[z].x * v4[z].x + v4[z].y * v4[z].y, ab = best.x * best.x + best.y * best.y;
    if (z < S && i4[z] >= 0 && a > ab) { best = v4[z]; bi = i4[z]; }
  }
  for (int z = 4; z < S; ++z) {
    const cpx<T> v = part[(size_t)z * slab + o];
    const T a = v.x * v.x + v.y * v.y, ab = best.x * best.x + best.y * best.y;
    const int vi = pidx[(size_t)z * slab + o];
    if (vi >= 0 && a > ab) { best = v; bi = vi; }
  }
  cpx<T> r = {T(0), T(0)};
  if (bi >= 0) {
    const size_t bb = (size_t)p * K + bi;
    r = cmul(best, cmul(dx[bb * n0 + x], dy[bb * n1 + y]));
  }
  out[o] = r;
  if (kidx) kidx[o] = bi;
}

// part: ksplit * P * n0 * n1 complex, pidx: the same count of int32 (scratch of the caller)
hipError_t launch_passB_split(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy, const void* tw1,
                              const SweepTables& tb, int P, int K, int ksplit, void* part, int32_t* pidx, void* out,
                              int32_t* kidx, hipStream_t s) {
  hipError_t e = hipErrorInvalidValue;
  {
#define CALL_P(T, LG, PD) run_passB<T, LG, PD, PB_PART>(a1, n0, Tbuf, Hy, tw1, tb, P, K, part, pidx, nullptr, nullptr, s, ksplit)
#define CASE_P(LG)                                                                                  \
  case LG:                                                                                          \
    e = dtype == 0 ? (a1.padded ? CALL_P(float, LG, true) : CALL_P(float, LG, false))               \
                   : (a1.padded ? CALL_P(double, LG, true) : CALL_P(double, LG, false));            \
    break;
  switch (a1.lg) { CASE_P(6) CASE_P(7) CASE_P(8) CASE_P(9) CASE_P(10) default: return hipErrorInvalidValue; }
#undef CASE_P
#undef CALL_P
  }
  if (e != hipSuccess) return e;
  dim3 grid((a1.n + 255) / 256, n0, P);
  if (dtype == 0)
    merge_parts_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)part, pidx, ksplit, P, K, n0, a1.n, (const cpx<float>*)tb.dx,
                                                   (const cpx<float>*)tb.dy, (cpx<float>*)out, kidx);
  else
    merge_parts_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)part, pidx, ksplit, P, K, n0, a1.n, (const cpx<double>*)tb.dx,
                                                    (const cpx<double>*)tb.dy, (cpx<double>*)out, kidx);
  return hipGetLastError();
}

// one peak (grid.y = 1), K candidates; mode PB_GATED (gate: device K x K bytes) or PB_PHASES (psi: K x n0 x n1)
hipError_t launch_passB_ext(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy, const void* tw1,
                            const SweepTables& tb, int K, int mode, void* out, int32_t* kidx, const uint8_t* gate,
                            void* psi, hipStream_t s) {
#define CALL_X(T, LG, PD)                                                                                        \
  (mode == PB_GATED ? run_passB<T, LG, PD, PB_GATED>(a1, n0, Tbuf, Hy, tw1, tb, 1, K, out, kidx, gate, psi, s)   \
                    : run_passB<T, LG, PD, PB_PHASES>(a1, n0, Tbuf, Hy, tw1, tb, 1, K, out, kidx, gate, psi, s))
#define CASE_X(LG)                                                                         \
  case LG:                                                                                 \
    if (dtype == 0) return a1.padded ? CALL_X(float, LG, true) : CALL_X(float, LG, false); \
    else return a1.padded ? CALL_X(double, LG, true) : CALL_X(double, LG, false);
  if (mode != PB_GATED && mode != PB_PHASES) return hipErrorInvalidValue;
  switch (a1.lg) { GPA_FOR_LG(CASE_X) }
#undef CASE_X
#undef CALL_X
  return hipErrorInvalidValue;
}

}  // namespace gpa
