// C ABI, a5-a7: phases / weights / per-pixel least squares and the weighted unwrap
// (geometric_phase_analysis.py:97-113, :196-245, :922-926; phase_unwrap.py:141-350).
#include "gpa_plan.h"

// ---- a5/a6 -------------------------------------------------------------------
int gpa_reconstruct_grad_dev(gpa_plan* p, const void* lockin, const double* kvecs, int P, int mask_border,
                             void* dudx, void* dudy, void* wnorm) {
  if (!p || !lockin || !kvecs || !dudx || !dudy) return fail(GPA_ERR_ARG, "gpa_reconstruct_grad: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_reconstruct_grad: need 2 <= P <= 8");
  HIP_TRY(hipSetDevice(p->device));
  TRY(stage_kmat(p, kvecs, P));
  HIP_TRY(launch_reconstruct(p->dtype, lockin, p->d_kmat, P, p->n0, p->n1, mask_border, dudx, dudy, wnorm,
                             p->stream));
  return GPA_OK;
}

int gpa_reconstruct_grad(gpa_plan* p, const void* lockin, const double* kvecs, int P, int mask_border,
                         void* dudx, void* dudy, void* wnorm) {
  if (!p || !lockin || !kvecs || !dudx || !dudy) return fail(GPA_ERR_ARG, "gpa_reconstruct_grad: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_reconstruct_grad: need 2 <= P <= 8");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_lockin, lockin, (size_t)P * npx * p->csz, hipMemcpyHostToDevice, p->stream));
  TRY(gpa_reconstruct_grad_dev(p, p->d_lockin, kvecs, P, mask_border, p->d_dudx, p->d_dudy, p->d_wnorm));
  HIP_TRY(hipMemcpyAsync(dudx, p->d_dudx, (size_t)2 * p->n0 * (p->n1 - 1) * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(dudy, p->d_dudy, (size_t)2 * (p->n0 - 1) * p->n1 * p->rsz, hipMemcpyDeviceToHost, p->stream));
  if (wnorm) HIP_TRY(hipMemcpyAsync(wnorm, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// reconstruct_u_inv_from_phases(pre_diff=True) (geometric_phase_analysis.py:228-237): the phase gradients are given
int gpa_reconstruct_prediff(gpa_plan* p, const void* grads, const void* weights, const double* kvecs, int P, void* dudx,
                            void* dudy, void* wnorm) {
  if (!p || !grads || !weights || !kvecs || !dudx || !dudy) return fail(GPA_ERR_ARG, "gpa_reconstruct_prediff: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_reconstruct_prediff: need 2 <= P <= 8 (and P <= max_batch)");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  TRY(ensure_tbuf(p, (P + 1) / 2));
  // staging: grads (P x npx x 2 reals = P complex planes) in d_lockin, weights in Tbuf
  HIP_TRY(hipMemcpyAsync(p->d_lockin, grads, (size_t)P * npx * 2 * p->rsz, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->Tbuf, weights, (size_t)P * npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(stage_kmat(p, kvecs, P));
  HIP_TRY(launch_prediff(p->dtype, p->d_lockin, p->Tbuf, p->d_kmat, P, p->n0, p->n1, p->d_dudx, p->d_dudy, p->d_wnorm,
                         p->stream));
  HIP_TRY(hipMemcpyAsync(dudx, p->d_dudx, (size_t)2 * p->n0 * (p->n1 - 1) * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(dudy, p->d_dudy, (size_t)2 * (p->n0 - 1) * p->n1 * p->rsz, hipMemcpyDeviceToHost, p->stream));
  if (wnorm) HIP_TRY(hipMemcpyAsync(wnorm, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

int gpa_weighted_lstsq(gpa_plan* p, const void* b, const void* weights, const double* kvecs, int P, void* out) {
  if (!p || !b || !weights || !kvecs || !out) return fail(GPA_ERR_ARG, "gpa_weighted_lstsq: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_weighted_lstsq: need 2 <= P <= 8 (and P <= max_batch)");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  // staging: b in d_lockin (P complex planes hold 2P real ones), weights in Tbuf
  TRY(ensure_tbuf(p, (P + 1) / 2));
  HIP_TRY(hipMemcpyAsync(p->d_lockin, b, (size_t)P * npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->Tbuf, weights, (size_t)P * npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(stage_kmat(p, kvecs, P));
  HIP_TRY(launch_wlstsq(p->dtype, p->d_lockin, p->Tbuf, p->d_kmat, P, npx, p->d_u, p->stream));
  HIP_TRY(hipMemcpyAsync(out, p->d_u, 2 * npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// ---- a7 ----------------------------------------------------------------------
int gpa_unwrap_prediff_dev(gpa_plan* p, const void* dx, const void* dy, const void* weight, int kmax,
                           double eps, int compat, void* phi, int* iters_out) {
  if (!p || !dx || !dy || !phi) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff: null argument");
  if (kmax < 1) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff: kmax must be >= 1");
  NEED_UNWRAP(p, "gpa_unwrap_prediff");
  HIP_TRY(hipSetDevice(p->device));
  int iters = 0;
  ProfInstall prof(p);   // (gpa_set_profiling: per-kernel times of this solve through gpa_last_kernel_profile)
  hipError_t e = unwrap_run(&p->uw, dx, dy, weight, false, kmax, eps, compat != 0, phi, &iters, p->stream);
  if (e != hipSuccess) return unwrap_fail(e);
  if (p->profiling) collect_kernel_profile(p);
  if (iters_out) *iters_out = iters;
  return GPA_OK;
}

int gpa_unwrap_prediff_enqueue_dev(gpa_plan* p, const void* dx, const void* dy, const void* weight, int kmax, double eps,
                                   int compat, void* phi) {
  if (!p || !dx || !dy || !phi) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff_enqueue: null argument");
  if (kmax < 1) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff_enqueue: kmax must be >= 1");
  NEED_UNWRAP(p, "gpa_unwrap_prediff_enqueue");
  HIP_TRY(hipSetDevice(p->device));
  hipError_t e = unwrap_enqueue(&p->uw, dx, dy, weight, false, kmax, eps, compat != 0, phi, p->stream);
  if (e != hipSuccess) return unwrap_fail(e);
  return GPA_OK;
}

int gpa_unwrap_finish(gpa_plan* p, int* iters_out) {
  if (!p) return fail(GPA_ERR_ARG, "gpa_unwrap_finish: null plan");
  NEED_UNWRAP(p, "gpa_unwrap_finish");
  HIP_TRY(hipSetDevice(p->device));
  int iters = 0;
  hipError_t e = unwrap_finish(&p->uw, &iters, p->stream);
  if (e != hipSuccess) return unwrap_fail(e);
  if (iters_out) *iters_out = iters;
  return GPA_OK;
}

int gpa_unwrap_prediff(gpa_plan* p, const void* dx, const void* dy, const void* weight, int kmax, double eps,
                       int compat, void* phi, int* iters_out) {
  if (!p || !dx || !dy || !phi) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  const size_t nx = (size_t)p->n0 * (p->n1 - 1), ny = (size_t)(p->n0 - 1) * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_dudx, dx, nx * p->rsz, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_dudy, dy, ny * p->rsz, hipMemcpyHostToDevice, p->stream));
  if (weight) HIP_TRY(hipMemcpyAsync(p->d_wnorm, weight, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(gpa_unwrap_prediff_dev(p, p->d_dudx, p->d_dudy, weight ? p->d_wnorm : nullptr, kmax, eps, compat, p->d_u,
                             iters_out));
  HIP_TRY(hipMemcpyAsync(phi, p->d_u, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

int gpa_unwrap(gpa_plan* p, const void* psi, const void* weight, int kmax, double eps, int compat, void* phi,
               int* iters_out) {
  if (!p || !psi || !phi) return fail(GPA_ERR_ARG, "gpa_unwrap: null argument");
  if (kmax < 1) return fail(GPA_ERR_ARG, "gpa_unwrap: kmax must be >= 1");
  NEED_UNWRAP(p, "gpa_unwrap");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, psi, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  if (weight) HIP_TRY(hipMemcpyAsync(p->d_wnorm, weight, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  int iters = 0;
  hipError_t e = unwrap_run(&p->uw, p->d_image, nullptr, weight ? p->d_wnorm : nullptr, true, kmax, eps,
                            compat != 0, p->d_u, &iters, p->stream);
  if (e != hipSuccess) return unwrap_fail(e);
  if (iters_out) *iters_out = iters;
  HIP_TRY(hipMemcpyAsync(phi, p->d_u, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

