// C ABI, the fused driver: extract_displacement_field in one call (geometric_phase_analysis.py:907-932), its asynchronous
// and batched forms.
#include "gpa_plan.h"

// ---- fused driver --------------------------------------------------------------
// host-side preparation: filter / carrier / k-matrix tables (re-staged only when they change; these upload
// synchronously), the x-plane buffer, and the second unwrap workspace + stream
int extract_stage(gpa_plan* p, const double* kvecs, int P, const double* klists, int K, double sigma, int* Bx) {
  const int B = P * K;
  TRY(ensure_filters(p, sigma));
  std::vector<double> kr((size_t)B * 2);
  for (int pp = 0; pp < P; ++pp)
    for (int k = 0; k < K; ++k) {
      kr[2 * ((size_t)pp * K + k)] = kvecs[2 * pp];
      kr[2 * ((size_t)pp * K + k) + 1] = kvecs[2 * pp + 1];
    }
  TRY(stage_kvectors(p, klists, kr.data(), B, Bx));
  TRY(ensure_tbuf(p, *Bx));
  TRY(stage_kmat(p, kvecs, P));
  {
    // (PAIR_MAXSIDE: measurement switch for the size up to which both components share one set of launches)
    const size_t side = opt_set(OPT_PAIR_MAXSIDE) ? (size_t)opt(OPT_PAIR_MAXSIDE).num : 1024;
    p->use_pair = (size_t)p->n0 * p->n1 <= side * side && !opt_set(OPT_NO_PAIR);
  }
  if (p->use_pair && !p->have_uwp) {
    size_t bp = 0;
    hipError_t ep = unwrap_workspace_create(p->dtype, p->n0, p->n1, p->stream, &p->uwp, &bp, 2);
    if (ep != hipSuccess) {
      unwrap_workspace_destroy(&p->uwp);
      return fail(GPA_ERR_HIP, std::string("paired unwrap workspace: ") + hipGetErrorString(ep));
    }
    p->ws_bytes += bp;
    p->have_uwp = true;
  }
  if (p->use_pair && !unwrap_supports_batch(&p->uwp)) p->use_pair = false;
  if (!p->stream2) {
    // the two displacement components are independent solves: give the second one its own
    // workspace and stream so the latency-bound kernels of one fill the gaps of the other
    HIP_TRY(hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    size_t b2 = 0;
    hipError_t e2 = unwrap_workspace_create(p->dtype, p->n0, p->n1, p->stream2, &p->uw2, &b2);
    if (e2 != hipSuccess) return fail(GPA_ERR_HIP, std::string("second unwrap workspace: ") + hipGetErrorString(e2));
    p->ws_bytes += b2;
  }
  return GPA_OK;
}

// every launch of the driver, nothing else: this is what a hipGraph of the call holds
int extract_launch(gpa_plan* p, const void* image, int P, int K, int Bx, int mask_border, int kmax, void* u,
                          void* lk, int32_t* kidx, bool want_lockins) {
  const size_t npx = (size_t)p->n0 * p->n1;
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[0], p->stream));
  HIP_TRY(launch_mean(p->dtype, image, npx, p->d_scratch, p->d_mean, p->stream));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[1], p->stream));
  TRY(run_passA(p, image, p->d_mean, p->Tbuf, Bx));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[2], p->stream));
  TRY(passB_select(p, P, K, lk, kidx, !want_lockins));
  const double* ystep = p->lk_raw ? p->d_ystep : nullptr;
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[3], p->stream));
  // phases / weights / per-pixel least squares fused with the unwrap's set-up: the gradient fields never
  // go to HBM, the kernel leaves r0 of both components in the two unwrap workspaces
  int nparts = 0;
  if (p->use_pair && p->have_uwp && !p->profiling && !p->serial_unwrap) {
    HIP_TRY(launch_reconstruct_setup(p->dtype, lk, p->d_kmat, P, p->n0, p->n1, mask_border, p->d_wnorm,
                                     unwrap_residual_buffer(&p->uwp, 0), unwrap_residual_buffer(&p->uwp, 1),
                                     unwrap_partials_buffer(&p->uwp, 0), unwrap_partials_buffer(&p->uwp, 1), &nparts,
                                     p->stream, 1, 0, 0, ystep));
    hipError_t ep = unwrap_enqueue_prepared(&p->uwp, p->d_wnorm, nparts, kmax, 1e-9, true, u, p->stream);
    if (ep == hipSuccess) ep = unwrap_fetch_iters(&p->uwp, p->h_iters, p->stream);
    if (ep != hipSuccess) return unwrap_fail(ep);
    p->iters_stride = 4;
    p->iters_off = unwrap_iters_slot(&p->uwp);
    return GPA_OK;
  }
  p->iters_stride = 1;
  p->iters_off = 0;
  HIP_TRY(launch_reconstruct_setup(p->dtype, lk, p->d_kmat, P, p->n0, p->n1, mask_border, p->d_wnorm,
                                   unwrap_residual_buffer(&p->uw), unwrap_residual_buffer(&p->uw2),
                                   unwrap_partials_buffer(&p->uw), unwrap_partials_buffer(&p->uw2), &nparts, p->stream, 1, 0,
                                   0, ystep));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[4], p->stream));
  HIP_TRY(hipEventRecord(p->ev_fork, p->stream));
  HIP_TRY(hipStreamWaitEvent(p->stream2, p->ev_fork, 0));
  // the second component's launches go out from the plan's helper thread while this thread enqueues the first
  // (not while profiling: the per-kernel event pairs belong to this thread)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(p->stream, &cap);
  const bool threaded = p->use_worker && !p->profiling && !p->serial_unwrap && cap == hipStreamCaptureStatusNone;
  hipError_t e2 = hipSuccess;
  void* u1 = (char*)u + npx * p->rsz;
  auto second = [&]() {
    e2 = unwrap_enqueue_prepared(&p->uw2, p->d_wnorm, nparts, kmax, 1e-9, true, u1, p->stream2);
    if (e2 == hipSuccess) e2 = unwrap_fetch_iters(&p->uw2, &p->h_iters[1], p->stream2);
  };
  if (threaded) {
    if (!p->worker) p->worker = new EnqueueWorker(p->device);
    p->worker->submit(second);
  }
  hipError_t e = unwrap_enqueue_prepared(&p->uw, p->d_wnorm, nparts, kmax, 1e-9, true, u, p->stream);
  if (p->profiling || p->serial_unwrap) {   // per-kernel timings: run the second component after the first
    HIP_TRY(hipEventRecord(p->ev_fork, p->stream));
    HIP_TRY(hipStreamWaitEvent(p->stream2, p->ev_fork, 0));
  }
  if (e == hipSuccess) e = unwrap_fetch_iters(&p->uw, &p->h_iters[0], p->stream);
  if (threaded) p->worker->wait(); else second();
  if (e == hipSuccess) e = e2;
  if (e != hipSuccess) return unwrap_fail(e);
  HIP_TRY(hipEventRecord(p->ev_join, p->stream2));
  HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_join, 0));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[5], p->stream));
  return GPA_OK;
}

// enqueue the whole driver on the plan's streams without any host synchronisation: ~110 eager kernel launches.
// (Rounds 2-4 could capture them into a hipGraph -- USE_GRAPH=1 -- and replay it: measured NOT faster than eager launches at
// any size on MI355X / ROCm 7.2, 512^2 0.65 against 0.60 ms, 4096^2 equal, and it serialised with the copy stream of
// gpa_download_async: profiles/r02_graph_vs_eager.txt.  Removed in round 5; git history has it.)
int extract_enqueue(gpa_plan* p, const void* image, const double* kvecs, int P, const double* klists, int K,
                           double sigma, int mask_border, int kmax, void* u, void* lockins, int32_t* kidx) {
  if (!p || !image || !kvecs || !klists || !u) return fail(GPA_ERR_ARG, "gpa_extract_displacement_field: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_extract_displacement_field: need 2 <= P <= 8");
  if (K < 1 || P * K > p->max_batch) return fail(GPA_ERR_STATE, "gpa_extract_displacement_field: P*K exceeds max_batch");
  if (kmax < 1) return fail(GPA_ERR_ARG, "kmax must be >= 1");
  HIP_TRY(hipSetDevice(p->device));
  int Bx = 0;
  TRY(extract_stage(p, kvecs, P, klists, K, sigma, &Bx));
  void* lk = lockins ? lockins : p->d_lockin;
  // per-kernel event pairs while profiling (installed for this thread until the function returns)
  ProfInstall prof(p);
  return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
}

int gpa_extract_displacement_field_async(gpa_plan* p, const void* image, const double* kvecs, int P,
                                         const double* klists, int K, double sigma, int mask_border, int kmax,
                                         void* u, void* lockins, int32_t* kidx) {
  return extract_enqueue(p, image, kvecs, P, klists, K, sigma, mask_border, kmax, u, lockins, kidx);
}

// A stack of images of one shape in one call: every kernel of the driver takes an image / problem index from its
// grid, so the stack is ONE set of ~50 launches instead of ~110 per image -- a small image is bound by its chain of
// dependent launches, not by their work.
// images: B x n0 x n1, u: B x 2 x n0 x n1 (device pointers), iters_out: 2 B counts (host, may be NULL: no
// synchronisation then).  The results are those of B separate gpa_extract_displacement_field_dev calls, bit for bit.
int gpa_extract_displacement_field_batch_dev(gpa_plan* p, const void* images, int B, const double* kvecs, int P,
                                             const double* klists, int K, double sigma, int mask_border, int kmax,
                                             void* u, int* iters_out) {
  if (!p || !images || !kvecs || !klists || !u) return fail(GPA_ERR_ARG, "gpa_extract_displacement_field_batch: null argument");
  if (B < 1 || B > 4096) return fail(GPA_ERR_ARG, "gpa_extract_displacement_field_batch: need 1 <= images <= 4096");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_extract_displacement_field_batch: need 2 <= P <= 8");
  if (K < 1 || P * K > p->max_batch) return fail(GPA_ERR_STATE, "gpa_extract_displacement_field_batch: P*K exceeds max_batch");
  if (kmax < 1) return fail(GPA_ERR_ARG, "kmax must be >= 1");
  HIP_TRY(hipSetDevice(p->device));
  int Bx = 0;
  TRY(extract_stage(p, kvecs, P, klists, K, sigma, &Bx));
  const size_t npx = (size_t)p->n0 * p->n1;
  // the batched workspace is a capacity: fewer frames (a ragged last chunk, a shorter stack) reuse it
  if (B <= p->uwb_images) {
    if (!unwrap_set_active(&p->uwb, 2 * B)) return fail(GPA_ERR_STATE, "batched unwrap workspace: bad active count");
  } else {
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->uwb_images) {
      unwrap_workspace_destroy(&p->uwb);
      (void)hipFree(p->d_wnorm_b);
      (void)hipHostFree(p->h_iters_b);
      p->uwb_images = 0;
      p->d_wnorm_b = nullptr;
      p->h_iters_b = nullptr;
    }
    size_t bb = 0;
    hipError_t e = unwrap_workspace_create(p->dtype, p->n0, p->n1, p->stream, &p->uwb, &bb, 2 * B);
    if (e != hipSuccess) {
      unwrap_workspace_destroy(&p->uwb);
      return fail(GPA_ERR_HIP, std::string("batched unwrap workspace: ") + hipGetErrorString(e));
    }
    if (!unwrap_supports_batch(&p->uwb)) {
      unwrap_workspace_destroy(&p->uwb);
      return fail(GPA_ERR_STATE, "gpa_extract_displacement_field_batch: this image shape has no batched unwrap");
    }
    e = hipMalloc(&p->d_wnorm_b, (size_t)B * npx * p->rsz);
    if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_iters_b, (size_t)8 * B * sizeof(int));
    if (e != hipSuccess) {
      unwrap_workspace_destroy(&p->uwb);
      if (p->d_wnorm_b) (void)hipFree(p->d_wnorm_b);
      p->d_wnorm_b = nullptr;
      return fail(GPA_ERR_HIP, std::string("batched driver buffers: ") + hipGetErrorString(e));
    }
    p->uwb_images = B;
  }
  // the sweep and the least squares of a chunk of images are ONE set of launches too (blockIdx.z / .y = image);
  // the chunk is what fits ~3 GB of x-planes (512^2: the whole stack, 4096^2: one image at a time)
  const size_t t_img = (size_t)Bx * npx * p->csz, l_img = (size_t)P * npx * p->csz;
  int chunk = (int)std::min<size_t>((size_t)B, std::max<size_t>(1, ((size_t)3 << 30) / t_img));
  if ((size_t)chunk * t_img > p->bT_bytes || (size_t)chunk * l_img > p->bL_bytes || chunk > p->b_chunk) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (void* b : {p->bT, p->bL, p->bMean, (void*)p->bScratch})
      if (b) (void)hipFree(b);
    p->bT = p->bL = p->bMean = nullptr;
    p->bScratch = nullptr;
    p->bT_bytes = p->bL_bytes = 0;
    p->b_chunk = 0;
    hipError_t ea = hipMalloc(&p->bT, (size_t)chunk * t_img);
    if (ea == hipSuccess) ea = hipMalloc(&p->bL, (size_t)chunk * l_img);
    if (ea == hipSuccess) ea = hipMalloc(&p->bMean, (size_t)chunk * 8);
    if (ea == hipSuccess) ea = hipMalloc((void**)&p->bScratch, (size_t)chunk * 1024 * sizeof(double));
    if (ea != hipSuccess) return fail(GPA_ERR_HIP, std::string("batched sweep buffers: ") + hipGetErrorString(ea));
    p->bT_bytes = (size_t)chunk * t_img;
    p->bL_bytes = (size_t)chunk * l_img;
    p->b_chunk = chunk;
  }
  TRY(shared_prepare(p, P, K));
  int nparts = 0;
  const size_t rstride = 2 * npx;                                                     // residual slices per image
  const size_t pstride = (size_t)(unwrap_partials_buffer(&p->uwb, 2) - unwrap_partials_buffer(&p->uwb, 0));
  for (int c0 = 0; c0 < B; c0 += chunk) {
    const int nimg = std::min(chunk, B - c0);
    const void* image = (const char*)images + (size_t)c0 * npx * p->rsz;
    HIP_TRY(launch_mean(p->dtype, image, npx, p->bScratch, p->bMean, p->stream, nimg));
    TRY(run_passA(p, image, p->bMean, p->bT, Bx, nimg));
    const bool raw = p->sh_use && !opt_set(OPT_NO_RAW);   // (the stack's lock-ins are never handed out)
    if (p->sh_use)
      HIP_TRY(launch_passB_shared(p->dtype, p->ax1s, p->n0, p->bT, p->ax1s.L == p->ax1.L ? p->tw1 : p->tw1s, p->tb,
                                  p->sh, p->sh_E, p->sh_Epad, P, K, p->bL, nullptr, p->stream, nimg, Bx, p->sh_elems, p->sh_nbl,
                                  raw));
    else
      HIP_TRY(launch_passB(p->dtype, p->ax1, p->n0, p->bT, p->Hy, p->tw1, p->tb, P, K, true, p->bL, nullptr, p->stream, nimg,
                           Bx));
    HIP_TRY(launch_reconstruct_setup(p->dtype, p->bL, p->d_kmat, P, p->n0, p->n1, mask_border,
                                     (char*)p->d_wnorm_b + (size_t)c0 * npx * p->rsz,
                                     unwrap_residual_buffer(&p->uwb, 2 * c0), unwrap_residual_buffer(&p->uwb, 2 * c0 + 1),
                                     unwrap_partials_buffer(&p->uwb, 2 * c0), unwrap_partials_buffer(&p->uwb, 2 * c0 + 1),
                                     &nparts, p->stream, nimg, rstride, pstride, raw ? p->d_ystep : nullptr));
  }
  hipError_t e = unwrap_enqueue_prepared(&p->uwb, p->d_wnorm_b, nparts, kmax, 1e-9, true, u, p->stream);
  if (e == hipSuccess) e = unwrap_fetch_iters(&p->uwb, p->h_iters_b, p->stream);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("batched unwrap: ") + hipGetErrorString(e));
  if (iters_out) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int j = 0; j < 2 * B; ++j) iters_out[j] = p->h_iters_b[4 * j + unwrap_iters_slot(&p->uwb)];
  }
  return GPA_OK;
}

// whether gpa_extract_displacement_field_batch_dev can take this plan's image shape (the fused iteration covers it);
// callers with other shapes loop over gpa_extract_displacement_field_dev instead
int gpa_supports_batch(gpa_plan* p) {
  if (!p) return 0;
  return unwrap_supports_batch(&p->uw) ? 1 : 0;
}

int gpa_last_batch_iters(gpa_plan* p, int B, int* iters_out) {
  if (!p || !iters_out || B < 1 || B > p->uwb_images) return fail(GPA_ERR_ARG, "gpa_last_batch_iters: bad argument");
  HIP_TRY(hipStreamSynchronize(p->stream));
  for (int j = 0; j < 2 * B; ++j) iters_out[j] = p->h_iters_b[4 * j + unwrap_iters_slot(&p->uwb)];
  return GPA_OK;
}

int gpa_last_iters(gpa_plan* p, int* iters2) {
  if (!p || !iters2) return fail(GPA_ERR_ARG, "null argument");
  HIP_TRY(hipStreamSynchronize(p->stream));
  iters2[0] = p->h_iters[p->iters_off];
  iters2[1] = p->h_iters[p->iters_stride + p->iters_off];
  return GPA_OK;
}

int gpa_extract_displacement_field_dev(gpa_plan* p, const void* image, const double* kvecs, int P,
                                       const double* klists, int K, double sigma, int mask_border, int kmax,
                                       void* u, void* lockins, int32_t* kidx, int* iters_out) {
  TRY(extract_enqueue(p, image, kvecs, P, klists, K, sigma, mask_border, kmax, u, lockins, kidx));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->profiling) {
    for (int i = 0; i < 5; ++i) hipEventElapsedTime(&p->stage_ms[i], p->stage_ev[i], p->stage_ev[i + 1]);
    collect_kernel_profile(p);
  }
  if (iters_out) { iters_out[0] = p->h_iters[p->iters_off]; iters_out[1] = p->h_iters[p->iters_stride + p->iters_off]; }
  return GPA_OK;
}

int gpa_extract_displacement_field(gpa_plan* p, const void* image, const double* kvecs, int P,
                                   const double* klists, int K, double sigma, int mask_border, int kmax,
                                   void* u, void* lockins, int32_t* kidx, int* iters_out) {
  if (!p || !image || !u) return fail(GPA_ERR_ARG, "gpa_extract_displacement_field: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(gpa_extract_displacement_field_dev(p, p->d_image, kvecs, P, klists, K, sigma, mask_border, kmax, p->d_u,
                                         lockins ? p->d_lockin : nullptr, kidx ? p->d_kidx : nullptr, iters_out));
  HIP_TRY(hipMemcpyAsync(u, p->d_u, 2 * npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  if (lockins) HIP_TRY(hipMemcpyAsync(lockins, p->d_lockin, (size_t)P * npx * p->csz, hipMemcpyDeviceToHost, p->stream));
  if (kidx) HIP_TRY(hipMemcpyAsync(kidx, p->d_kidx, (size_t)P * npx * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

int gpa_extract_gradients(gpa_plan* p, const void* image, const double* kvecs, int P, const double* klists, int K,
                          double sigma, int mask_border, void* dudx, void* dudy, void* wnorm) {
  if (!p || !image || !kvecs || !klists || !dudx || !dudy || !wnorm)
    return fail(GPA_ERR_ARG, "gpa_extract_gradients: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_extract_gradients: need 2 <= P <= 8");
  if (K < 1 || P * K > p->max_batch) return fail(GPA_ERR_STATE, "gpa_extract_gradients: P*K exceeds max_batch");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  // no mean subtraction here: a tile must be offset by the mean of the WHOLE image
  // (geometric_phase_analysis.py:919), which only the caller knows
  TRY(sweep_peaks_dev(p, p->d_image, nullptr, kvecs, P, klists, K, sigma, p->d_lockin, nullptr, true));
  TRY(stage_kmat(p, kvecs, P));
  HIP_TRY(launch_reconstruct(p->dtype, p->d_lockin, p->d_kmat, P, p->n0, p->n1, mask_border, p->d_dudx, p->d_dudy,
                             p->d_wnorm, p->stream, p->lk_raw ? p->d_ystep : nullptr));
  HIP_TRY(hipMemcpyAsync(dudx, p->d_dudx, (size_t)2 * p->n0 * (p->n1 - 1) * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(dudy, p->d_dudy, (size_t)2 * (p->n0 - 1) * p->n1 * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(wnorm, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

