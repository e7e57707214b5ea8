// C ABI, tile sharding: means, the tile stage, stitching, stream-to-stream ordering.
#include "gpa_plan.h"

int gpa_mean_dev(gpa_plan* p, const void* data, size_t count, double* mean_out) {
  if (!p || !data || !mean_out || count == 0) return fail(GPA_ERR_ARG, "gpa_mean_dev: null argument");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(launch_mean(p->dtype, data, count, p->d_scratch, p->d_mean, p->stream));
  double buf = 0.0;
  HIP_TRY(hipMemcpyAsync(&buf, p->d_mean, p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *mean_out = p->dtype == GPA_F32 ? (double)*reinterpret_cast<float*>(&buf) : buf;
  return GPA_OK;
}

// the tile stage of one window.  mean_on_device: p->d_tile_mean holds the mean already (gpa_tile_set_mean_dev);
// otherwise `mean` is staged there (once per value).  wn_plane != 0: a second copy of the weight wn_plane elements on.
int tile_gradients_impl(gpa_plan* p, const void* image, size_t image_pitch, int r0, int c0, bool mean_on_device,
                               double mean, const double* kvecs, int P, const double* klists, int K, double sigma,
                               int mask_border, int i0, int j0, int t0, int t1, void* dx, size_t dx_pitch, size_t dx_plane,
                               void* dy, size_t dy_pitch, size_t dy_plane, void* wn, size_t wn_pitch, size_t wn_plane) {
  if (!p || !image || !kvecs || !klists || !dx || !dy || !wn)
    return fail(GPA_ERR_ARG, "gpa_tile_gradients_dev: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_tile_gradients_dev: need 2 <= P <= 8");
  if (K < 1 || P * K > p->max_batch) return fail(GPA_ERR_STATE, "gpa_tile_gradients_dev: P*K exceeds max_batch");
  const int n0 = p->n0, n1 = p->n1;
  if (r0 < 0 || c0 < 0 || (size_t)c0 + n1 > image_pitch || i0 < 0 || j0 < 0 || t0 < 1 || t1 < 1 || i0 + t0 > n0 ||
      j0 + t1 > n1)
    return fail(GPA_ERR_ARG, "gpa_tile_gradients_dev: window / interior rectangle out of range");
  HIP_TRY(hipSetDevice(p->device));
  const size_t rsz = p->rsz;
  hipStream_t st = p->stream;
  // the window: read in place when it is contiguous (the pipeline keeps its windows that way), else cut out of the
  // larger image by a copy kernel; the mean of the WHOLE image (geometric_phase_analysis.py:919) is subtracted by pass A
  const void* win = (const char*)image + ((size_t)r0 * image_pitch + c0) * rsz;
  if (image_pitch != (size_t)n1) {
    const void* src[1] = {win};
    void* dst[1] = {p->d_image};
    const size_t sp[1] = {image_pitch}, dp[1] = {(size_t)n1};
    const int rows[1] = {n0}, cols[1] = {n1};
    HIP_TRY(launch_copy_fields(p->dtype, src, dst, sp, dp, rows, cols, 1, st));
    win = p->d_image;
  }
  if (!mean_on_device && !(mean == p->tile_mean)) {   // tiles of one image share the mean: staged once (d_tile_mean is
    HIP_TRY(hipStreamSynchronize(st));                // not d_mean, which the fused driver recomputes per call)
    if (p->dtype == GPA_F32) {
      *reinterpret_cast<float*>(p->h_k) = (float)mean;
    } else {
      p->h_k[0] = mean;
    }
    // h_k is pinned and also stages k-vectors: the copy must have left it before stage_kvectors rewrites it
    HIP_TRY(hipMemcpyAsync(p->d_tile_mean, p->h_k, rsz, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    p->tile_mean = mean;
  }
  TRY(sweep_peaks_dev(p, win, p->d_tile_mean, kvecs, P, klists, K, sigma, p->d_lockin, nullptr, true));
  TRY(stage_kmat(p, kvecs, P));
  if (!opt_set(OPT_NO_TILEFUSE)) {
    // the least-squares kernel stores the interior pixels straight into the tile blocks (one launch, and neither the
    // full-window fields nor the copy that cut the interiors out of them: 0.585 -> see profiles per 2048^2 window)
    void* const dxs[2] = {dx, (char*)dx + dx_plane * rsz};
    void* const dys[2] = {dy, (char*)dy + dy_plane * rsz};
    void* const wns[2] = {wn, wn_plane ? (void*)((char*)wn + wn_plane * rsz) : nullptr};
    HIP_TRY(launch_reconstruct_tile(p->dtype, p->d_lockin, p->d_kmat, P, n0, n1, mask_border, i0, j0, t0, t1, dxs, dx_pitch,
                                    dys, dy_pitch, wns, wn_pitch, st, p->lk_raw ? p->d_ystep : nullptr));
    return GPA_OK;
  }
  HIP_TRY(launch_reconstruct(p->dtype, p->d_lockin, p->d_kmat, P, n0, n1, mask_border, p->d_dudx, p->d_dudy,
                             p->d_wnorm, st, p->lk_raw ? p->d_ystep : nullptr));
  // interiors -> destination in ONE launch; the difference fields are one column / row short of the window
  const int wx = std::min(t1, n1 - 1 - j0), hy = std::min(t0, n0 - 1 - i0);
  const void* src[6];
  void* dst[6];
  size_t sp[6], dp[6];
  int rows[6], cols[6], nf = 0;
  for (int c = 0; c < 2; ++c) {
    src[nf] = (const char*)p->d_dudx + (((size_t)c * n0 + i0) * (n1 - 1) + j0) * rsz;
    dst[nf] = (char*)dx + c * dx_plane * rsz;
    sp[nf] = (size_t)(n1 - 1); dp[nf] = dx_pitch; rows[nf] = t0; cols[nf] = wx > 0 ? wx : 0;
    ++nf;
  }
  for (int c = 0; c < 2; ++c) {
    src[nf] = (const char*)p->d_dudy + (((size_t)c * (n0 - 1) + i0) * n1 + j0) * rsz;
    dst[nf] = (char*)dy + c * dy_plane * rsz;
    sp[nf] = (size_t)n1; dp[nf] = dy_pitch; rows[nf] = hy > 0 ? hy : 0; cols[nf] = t1;
    ++nf;
  }
  for (int c = 0; c < (wn_plane ? 2 : 1); ++c) {
    src[nf] = (const char*)p->d_wnorm + ((size_t)i0 * n1 + j0) * rsz;
    dst[nf] = (char*)wn + c * wn_plane * rsz;
    sp[nf] = (size_t)n1; dp[nf] = wn_pitch; rows[nf] = t0; cols[nf] = t1;
    ++nf;
  }
  HIP_TRY(launch_copy_fields(p->dtype, src, dst, sp, dp, rows, cols, nf, st));
  return GPA_OK;
}

int gpa_tile_gradients_dev(gpa_plan* p, const void* image, size_t image_pitch, int r0, int c0, double mean,
                           const double* kvecs, int P, const double* klists, int K, double sigma, int mask_border,
                           int i0, int j0, int t0, int t1, void* dx, size_t dx_pitch, size_t dx_plane, void* dy,
                           size_t dy_pitch, size_t dy_plane, void* wn, size_t wn_pitch) {
  return tile_gradients_impl(p, image, image_pitch, r0, c0, false, mean, kvecs, P, klists, K, sigma, mask_border, i0, j0, t0,
                             t1, dx, dx_pitch, dx_plane, dy, dy_pitch, dy_plane, wn, wn_pitch, 0);
}

int gpa_tile_gradients_meandev_dev(gpa_plan* p, const void* image, size_t image_pitch, int r0, int c0,
                                   const double* kvecs, int P, const double* klists, int K, double sigma, int mask_border,
                                   int i0, int j0, int t0, int t1, void* dx, size_t dx_pitch, size_t dx_plane, void* dy,
                                   size_t dy_pitch, size_t dy_plane, void* wn, size_t wn_pitch, size_t wn_plane) {
  return tile_gradients_impl(p, image, image_pitch, r0, c0, true, 0.0, kvecs, P, klists, K, sigma, mask_border, i0, j0, t0,
                             t1, dx, dx_pitch, dx_plane, dy, dy_pitch, dy_plane, wn, wn_pitch, wn_plane);
}

int gpa_tile_sums_dev(gpa_plan* p, const void* wins, size_t win_stride, size_t win_pitch, const int* rects_dev, int ntiles,
                      int max_rows, double* sum_dev) {
  if (!p || !wins || !rects_dev || !sum_dev || ntiles < 1 || max_rows < 1)
    return fail(GPA_ERR_ARG, "gpa_tile_sums_dev: bad argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t need = (size_t)ntiles * tile_sums_bands(max_rows);
  if (need > p->tsum_cap) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->d_tsum_part) { (void)hipFree(p->d_tsum_part); p->ws_bytes -= p->tsum_cap * sizeof(double); }
    p->d_tsum_part = nullptr;
    p->tsum_cap = 0;
    TRY(dmalloc(p, (void**)&p->d_tsum_part, need * sizeof(double)));
    p->tsum_cap = need;
  }
  HIP_TRY(launch_tile_sums(p->dtype, wins, win_stride, win_pitch, rects_dev, ntiles, max_rows, p->d_tsum_part, sum_dev, p->stream));
  return GPA_OK;
}

int gpa_tile_set_mean_dev(gpa_plan* p, const double* sum_dev, double scale) {
  if (!p || !sum_dev) return fail(GPA_ERR_ARG, "gpa_tile_set_mean_dev: null argument");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(launch_set_mean(p->dtype, sum_dev, scale, p->d_tile_mean, p->stream));
  p->tile_mean = std::numeric_limits<double>::quiet_NaN();   // (a later host-valued mean is staged again)
  return GPA_OK;
}

int gpa_stitch_tiles_dev(gpa_plan* p, const void* tiles, size_t slot_stride, size_t field_stride, size_t tile_pitch,
                         const int* table_dev, int ntiles, int t0, int t1, int nf, void* const* dst, const size_t* dst_pitch,
                         const int* dst_rows, const int* dst_cols) {
  if (!p || !tiles || !table_dev || !dst || !dst_pitch || !dst_rows || !dst_cols)
    return fail(GPA_ERR_ARG, "gpa_stitch_tiles_dev: null argument");
  if (nf < 1 || nf > 6 || ntiles < 1 || t0 < 1 || t1 < 1) return fail(GPA_ERR_ARG, "gpa_stitch_tiles_dev: need 1 <= nf <= 6, tiles >= 1");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(launch_stitch(p->dtype, tiles, slot_stride, field_stride, tile_pitch, table_dev, ntiles, t0, t1, nf, dst, dst_pitch,
                        dst_rows, dst_cols, p->stream));
  return GPA_OK;
}

// stream-to-stream ordering without a host synchronisation
int plan_event(gpa_plan* p) {
  if (!p->ev_x) HIP_TRY(hipEventCreateWithFlags(&p->ev_x, hipEventDisableTiming));
  return GPA_OK;
}
int gpa_plan_wait_stream(gpa_plan* p, void* stream) {
  if (!p) return fail(GPA_ERR_ARG, "gpa_plan_wait_stream: null plan");
  HIP_TRY(hipSetDevice(p->device));
  TRY(plan_event(p));
  HIP_TRY(hipEventRecord(p->ev_x, (hipStream_t)stream));
  HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_x, 0));
  return GPA_OK;
}
int gpa_stream_wait_plan(gpa_plan* p, void* stream) {
  if (!p) return fail(GPA_ERR_ARG, "gpa_stream_wait_plan: null plan");
  HIP_TRY(hipSetDevice(p->device));
  TRY(plan_event(p));
  HIP_TRY(hipEventRecord(p->ev_x, p->stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->ev_x, 0));
  return GPA_OK;
}

