// C ABI, f-1: Lawler-Fujita undistortion (geometric_phase_analysis.py:248-300, :935-974).
#include "gpa_plan.h"

int invert_u_host(gpa_plan* p, const void* u, int iters, int edge, int shift, void* out, int mode, bool overlap) {
  if (!p || !u || !out) return fail(GPA_ERR_ARG, "gpa_invert_u: null argument");
  if (iters < 1 || edge < 0) return fail(GPA_ERR_ARG, "gpa_invert_u: need iters >= 1, edge >= 0");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1, nout = (size_t)(p->n0 + 2 * edge) * (p->n1 + 2 * edge);
  void* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, 2 * nout * p->rsz));
  hipError_t e = hipMemcpyAsync(p->d_u, u, 2 * npx * p->rsz, hipMemcpyHostToDevice, p->stream);
  if (e == hipSuccess) e = warp_invert_u(p->dtype, p->d_u, p->n0, p->n1, 1.0, iters, edge, shift, d_out, p->stream, mode, overlap ? 1 : 0);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, 2 * nout * p->rsz, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  hipFree(d_out);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_invert_u: ") + hipGetErrorString(e));
  return GPA_OK;
}

int gpa_invert_u_overlap(gpa_plan* p, const void* u, int iters, int edge, void* out) {
  return invert_u_host(p, u, iters, edge, 0, out, 0, true);
}

// invert_u (geometric_phase_analysis.py:248-259): the image's own grid, one sampling at r and then `iters` rounds
// at r + u_it(r) - edge
int gpa_invert_u(gpa_plan* p, const void* u, int iters, int edge, void* out) {
  return invert_u_host(p, u, iters, 0, edge, out, 0, false);
}

// the two with scipy's boundary mode as an argument: 0 = 'nearest', 1 = 'constant' (the `mode=` keyword of
// geometric_phase_analysis.py:248, :262); overlap != 0 = invert_u_overlap
int gpa_invert_u_mode(gpa_plan* p, const void* u, int iters, int edge, int overlap, int mode, void* out) {
  if (mode != 0 && mode != 1) return fail(GPA_ERR_ARG, "gpa_invert_u_mode: mode must be 0 (nearest) or 1 (constant)");
  return overlap ? invert_u_host(p, u, iters, edge, 0, out, mode, true) : invert_u_host(p, u, iters, 0, edge, out, mode, false);
}

int gpa_undistort_image(gpa_plan* p, const void* deformed, const void* u, void* out) {
  if (!p || !deformed || !u || !out) return fail(GPA_ERR_ARG, "gpa_undistort_image: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_u, u, 2 * npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_image, deformed, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  // u_inv = invert_u_overlap(-u) (35 rounds, no overlap edge) lands in dudx (2 planes of n0*n1 fit)
  HIP_TRY(warp_invert_u(p->dtype, p->d_u, p->n0, p->n1, -1.0, 35, 0, 0, p->d_dudx, p->stream));
  HIP_TRY(warp_image(p->dtype, p->d_image, p->d_dudx, p->n0, p->n1, p->d_wnorm, p->stream));
  HIP_TRY(hipMemcpyAsync(out, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

