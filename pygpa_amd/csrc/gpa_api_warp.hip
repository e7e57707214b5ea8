// C ABI, f-1: Lawler-Fujita undistortion (geometric_phase_analysis.py:248-300, :935-974).
//
// The _dev entry points take device pointers, enqueue on the plan's stream and return without a host synchronisation
// (gpa_plan_sync / gpa_stream_wait_plan order later work); scratch and prefilter taps live in the plan (p->warp, grown
// on first use).  `rects` = nrect x {r0, c0, h, w} restricts the heavy kernels -- the fixed-point rounds, the final resampling --
// to windows of the output grid: the tiles a rank owns after the stitched field has been handed to it.
// The host-pointer entry points of rounds 1-4 are these plus the copies.
#include "gpa_plan.h"

// windows of the output grid: every rect must have a positive size and touch the grid (a window that lies outside entirely is a
// caller's mistake, not an empty request: ADVICE r05); windows that stick out are clipped by the kernels' launchers
static int check_rects(const int* rects, int nrect, int o0, int o1, const char* who) {
  if (nrect < 0 || (nrect > 0 && !rects)) return fail(GPA_ERR_ARG, std::string(who) + ": nrect windows need rects");
  for (int q = 0; q < nrect; ++q) {
    const int *r = rects + 4 * q;
    if (r[2] <= 0 || r[3] <= 0 || r[0] >= o0 || r[1] >= o1 || r[0] + r[2] <= 0 || r[1] + r[3] <= 0)
      return fail(GPA_ERR_ARG, std::string(who) + ": window " + std::to_string(q) + " {" + std::to_string(r[0]) + ", " + std::to_string(r[1]) +
                                   ", " + std::to_string(r[2]) + ", " + std::to_string(r[3]) + "} is empty or outside the " +
                                   std::to_string(o0) + " x " + std::to_string(o1) + " output grid");
  }
  return GPA_OK;
}

static int check_invert_args(gpa_plan* p, const void* u, void* out, int iters, int edge, int mode) {
  if (!p || !u || !out) return fail(GPA_ERR_ARG, "gpa_invert_u: null argument");
  if (iters < 1 || edge < 0) return fail(GPA_ERR_ARG, "gpa_invert_u: need iters >= 1, edge >= 0");
  if (mode != 0 && mode != 1) return fail(GPA_ERR_ARG, "gpa_invert_u_mode: mode must be 0 (nearest) or 1 (constant)");
  return GPA_OK;
}

// invert_u_overlap (overlap != 0: out is 2 x (n0 + 2 edge) x (n1 + 2 edge)) or invert_u (out 2 x n0 x n1, every round
// after the first sampled at r + u_it(r) - edge) of scale * u, all on the device
int gpa_invert_u_mode_dev(gpa_plan* p, const void* u_dev, double scale, int iters, int edge, int overlap, int mode,
                          const int* rects, int nrect, void* out_dev) {
  TRY(check_invert_args(p, u_dev, out_dev, iters, edge, mode));
  TRY(check_rects(rects, nrect, p->n0 + (overlap ? 2 * edge : 0), p->n1 + (overlap ? 2 * edge : 0), "gpa_invert_u_mode_dev"));
  HIP_TRY(hipSetDevice(p->device));
  p->warp.counted = &p->ws_bytes;
  ProfInstall prof(p);
  const hipError_t e = overlap ? warp_invert_u(p->dtype, u_dev, p->n0, p->n1, scale, iters, edge, 0, out_dev, p->stream, mode, 1, &p->warp, rects, nrect)
                               : warp_invert_u(p->dtype, u_dev, p->n0, p->n1, scale, iters, 0, edge, out_dev, p->stream, mode, 0, &p->warp, rects, nrect);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_invert_u_mode_dev: ") + hipGetErrorString(e));
  if (p->profiling) { HIP_TRY(hipStreamSynchronize(p->stream)); collect_kernel_profile(p); }
  return GPA_OK;
}

// undistort_image (geometric_phase_analysis.py:935-974): u_inv = invert_u_overlap(-u) (35 rounds, mode 'nearest', no
// edge) into uinv_dev (2 x n0 x n1; null: plan scratch), then deformed resampled at r + u_inv(r) (order 3, 'constant')
int gpa_undistort_image_dev(gpa_plan* p, const void* deformed_dev, const void* u_dev, const int* rects, int nrect,
                            void* uinv_dev, void* out_dev) {
  return gpa_undistort_image_scaled_dev(p, deformed_dev, u_dev, 1.0, rects, nrect, uinv_dev, out_dev);
}

// undistort_image(deformed, scale * u).  The reference's tests recover the true displacement as MINUS the extracted field
// (tests/test_geometric_phase_analysis.py:63) and undistort with the true one (:76): a caller that keeps the extracted field on
// the device undistorts with scale = -1 instead of negating 2 n0 n1 values first.
int gpa_undistort_image_scaled_dev(gpa_plan* p, const void* deformed_dev, const void* u_dev, double scale, const int* rects,
                                   int nrect, void* uinv_dev, void* out_dev) {
  if (!p || !deformed_dev || !u_dev || !out_dev) return fail(GPA_ERR_ARG, "gpa_undistort_image_dev: null argument");
  TRY(check_rects(rects, nrect, p->n0, p->n1, "gpa_undistort_image_dev"));
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  // one reservation for the inversion AND the resampling (the first call used to grow the scratch twice, with a stream
  // synchronisation and a free in between); counted in gpa_plan_workspace_bytes
  p->warp.counted = &p->ws_bytes;
  HIP_TRY(warp_reserve_undistort(p->dtype, p->n0, p->n1, &p->warp, p->stream));
  void* uinv = uinv_dev ? uinv_dev : p->d_dudx;   // (2 planes of n0 x n1 fit)
  HIP_TRY(warp_invert_u(p->dtype, u_dev, p->n0, p->n1, -scale, 35, 0, 0, uinv, p->stream, 0, 1, &p->warp, rects, nrect));
  HIP_TRY(warp_image(p->dtype, deformed_dev, uinv, p->n0, p->n1, out_dev, p->stream, &p->warp, rects, nrect));
  if (p->profiling) { HIP_TRY(hipStreamSynchronize(p->stream)); collect_kernel_profile(p); }
  return GPA_OK;
}

static int invert_u_host(gpa_plan* p, const void* u, int iters, int edge, int overlap, int mode, void* out) {
  TRY(check_invert_args(p, u, out, iters, edge, mode));
  HIP_TRY(hipSetDevice(p->device));
  const int e2 = overlap ? edge : 0;
  const size_t npx = (size_t)p->n0 * p->n1, nout = (size_t)(p->n0 + 2 * e2) * (p->n1 + 2 * e2);
  void* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, 2 * nout * p->rsz));
  int rc = GPA_OK;
  hipError_t e = hipMemcpyAsync(p->d_u, u, 2 * npx * p->rsz, hipMemcpyHostToDevice, p->stream);
  if (e == hipSuccess) rc = gpa_invert_u_mode_dev(p, p->d_u, 1.0, iters, edge, overlap, mode, nullptr, 0, d_out);
  if (e == hipSuccess && rc == GPA_OK) e = hipMemcpyAsync(out, d_out, 2 * nout * p->rsz, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_invert_u: ") + hipGetErrorString(e));
  return rc;
}

int gpa_invert_u_overlap(gpa_plan* p, const void* u, int iters, int edge, void* out) {
  return invert_u_host(p, u, iters, edge, 1, 0, out);
}

// invert_u (geometric_phase_analysis.py:248-259): the image's own grid, one sampling at r and then `iters` rounds
// at r + u_it(r) - edge
int gpa_invert_u(gpa_plan* p, const void* u, int iters, int edge, void* out) {
  return invert_u_host(p, u, iters, edge, 0, 0, out);
}

// the two with scipy's boundary mode as an argument: 0 = 'nearest', 1 = 'constant' (the `mode=` keyword of
// geometric_phase_analysis.py:248, :262); overlap != 0 = invert_u_overlap
int gpa_invert_u_mode(gpa_plan* p, const void* u, int iters, int edge, int overlap, int mode, void* out) {
  return invert_u_host(p, u, iters, edge, overlap ? 1 : 0, mode, out);
}

int gpa_undistort_image(gpa_plan* p, const void* deformed, const void* u, void* out) {
  if (!p || !deformed || !u || !out) return fail(GPA_ERR_ARG, "gpa_undistort_image: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_u, u, 2 * npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_image, deformed, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(gpa_undistort_image_dev(p, p->d_image, p->d_u, nullptr, 0, nullptr, p->d_wnorm));
  HIP_TRY(hipMemcpyAsync(out, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}
