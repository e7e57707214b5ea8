// C ABI, a9 / f-2 / f-3 / f-4: periodic-component DFT, peak finding, Wiener deconvolution, Jacobian / properties, Huber plane fit.
#include "gpa_plan.h"

// the two axes of the plain 2-D DFT, built on first use (DFT_ENGINE=chirpz | chirpz2 forces an engine: tests)
int dft_axes(gpa_plan* p) {
  if (p->bx0.kind != DFT_NONE) return GPA_OK;
  int force = 0;
  if (opt_set(OPT_DFT_ENGINE)) {
    const char* v = opt(OPT_DFT_ENGINE).str;
    force = !strcmp(v, "chirpz2") ? DFT_BIG : !strcmp(v, "chirpz") ? DFT_BLUE : 0;
  }
  size_t b = 0;
  p->dftw.counted = &p->ws_bytes;
  hipError_t e = dft_axis_create(p->dtype, p->n0, p->stream, &p->bx0, &b, force);
  if (e == hipSuccess) e = dft_axis_create(p->dtype, p->n1, p->stream, &p->bx1, &b, force);
  if (e != hipSuccess) {
    dft_axis_destroy(&p->bx0);
    dft_axis_destroy(&p->bx1);
    return fail(GPA_ERR_HIP, std::string("DFT tables (axis lengths 1 ... 65536): ") + hipGetErrorString(e));
  }
  p->ws_bytes += b;
  return GPA_OK;
}

// periodic-component DFT of the device image -> p->d_lockin (plane 0), or (abs_out) |fftshift| of it as reals -> abs_out;
// u_hat stays in p->Tbuf
int per_dft_staged(gpa_plan* p, const void* d_image, void* abs_out) {
  TRY(dft_axes(p));
  if (!p->d_pertab) {
    TRY(dmalloc(p, &p->d_pertab, per_tables_bytes(p->dtype, p->n0, p->n1)));
    HIP_TRY(per_tables_fill(p->dtype, p->n0, p->n1, p->d_pertab, p->stream));
  }
  // border-difference vectors: d0 holds n1, d1 holds n0 complex values
  void* d0 = p->d_aux1;
  void* d1 = p->d_aux0;
  HIP_TRY(per_borders(p->dtype, d_image, p->n0, p->n1, d0, d1, p->stream));
  // |P^| is symmetric under (q, r) -> (-q, -r): for abs_out half of u_hat is enough (power-of-two rows; NO_DFT_HALF: tests)
  const size_t hp = abs_out && !opt_set(OPT_NO_DFT_HALF) ? dft2_half_pitch(p->bx1) : 0;
  if (hp) HIP_TRY(dft2_forward_real_half(p->dtype, p->bx0, p->bx1, d_image, p->Tbuf, &p->dftw, p->stream));
  else HIP_TRY(dft2_forward_real(p->dtype, p->bx0, p->bx1, d_image, p->Tbuf, &p->dftw, p->stream));
  HIP_TRY(dft_rows_inplace(p->dtype, p->bx1, 1, d0, &p->dftw, p->stream));
  HIP_TRY(dft_rows_inplace(p->dtype, p->bx0, 1, d1, &p->dftw, p->stream));
  HIP_TRY(per_combine(p->dtype, p->Tbuf, d0, d1, p->n0, p->n1, p->d_pertab, abs_out != nullptr,
                      abs_out ? abs_out : p->d_lockin, p->stream, hp));
  return GPA_OK;
}

int gpa_per_dft_dev(gpa_plan* p, const void* d_image, void* d_out) {
  if (!p || !d_image || !d_out) return fail(GPA_ERR_ARG, "gpa_per_dft_dev: null argument");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  TRY(per_dft_staged(p, d_image, nullptr));
  HIP_TRY(hipMemcpyAsync(d_out, p->d_lockin, (size_t)p->n0 * p->n1 * p->csz, hipMemcpyDeviceToDevice, p->stream));
  if (p->profiling) { HIP_TRY(hipStreamSynchronize(p->stream)); collect_kernel_profile(p); }
  return GPA_OK;
}

int gpa_per_dft(gpa_plan* p, const void* image, void* out) {
  if (!p || !image || !out) return fail(GPA_ERR_ARG, "gpa_per_dft: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(per_dft_staged(p, p->d_image, nullptr));
  HIP_TRY(hipMemcpyAsync(out, p->d_lockin, npx * p->csz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// moisan2011.per in full: inverse_dft == 0 -> (p_hat, s_hat) complex (s_out may be null), != 0 -> (p, s) real
int gpa_per(gpa_plan* p, const void* image, int inverse_dft, void* p_out, void* s_out) {
  if (!p || !image || !p_out || (inverse_dft && !s_out)) return fail(GPA_ERR_ARG, "gpa_per: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(per_dft_staged(p, p->d_image, nullptr));   // p_hat in d_lockin, u_hat still in Tbuf
  if (!inverse_dft) {
    HIP_TRY(hipMemcpyAsync(p_out, p->d_lockin, npx * p->csz, hipMemcpyDeviceToHost, p->stream));
    if (s_out) {
      HIP_TRY(per_smooth_hat(p->dtype, p->Tbuf, p->d_lockin, npx, p->stream));
      HIP_TRY(hipMemcpyAsync(s_out, p->Tbuf, npx * p->csz, hipMemcpyDeviceToHost, p->stream));
    }
  } else {
    HIP_TRY(per_components(p->dtype, p->bx0, p->bx1, p->d_lockin, p->d_image, p->d_wnorm, p->d_dudx, &p->dftw, p->stream));
    HIP_TRY(hipMemcpyAsync(p_out, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(s_out, p->d_dudx, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// ---- f-3 -------------------------------------------------------------------------
// scipy.ndimage._filters._gaussian_kernel1d (order 0): exp(-x^2 / 2 sigma^2) / sum, radius int(4 sigma + 0.5)
int gaussian_weights(double sigma, std::vector<double>& w) {
  const int R = (int)(4.0 * sigma + 0.5);
  w.resize(2 * (size_t)R + 1);
  double sum = 0.0;
  for (int k = -R; k <= R; ++k) { w[k + R] = exp(-0.5 / (sigma * sigma) * (double)k * k); sum += w[k + R]; }
  for (double& v : w) v /= sum;
  return R;
}

// device tables of the FFT form of one axis' filter: transfer function + twiddles of the segment length, two sigmas kept per axis
static int gauss_tables(gpa_plan* p, int axis, double sigma, int lg, const std::vector<double>& w, const void** H, const void** tw) {
  gpa_plan::GaussTab* slot = nullptr;
  for (auto& g : p->gft[axis])
    if (g.sigma == sigma && g.lg == lg) slot = &g;
  if (!slot) {
    slot = p->gft[axis][0].stamp <= p->gft[axis][1].stamp ? &p->gft[axis][0] : &p->gft[axis][1];
    HIP_TRY(hipStreamSynchronize(p->stream));   // an earlier call may still read the tables being replaced
    slot->sigma = -1.0;
    const size_t L = (size_t)1 << lg;
    if (slot->cap_lg < lg) {
      if (slot->H) { (void)hipFree(slot->H); p->ws_bytes -= ((size_t)1 << slot->cap_lg) * p->rsz; }
      if (slot->tw) { (void)hipFree(slot->tw); p->ws_bytes -= ((size_t)1 << slot->cap_lg) * p->csz; }
      slot->H = slot->tw = nullptr;
      slot->cap_lg = 0;
      TRY(dmalloc(p, &slot->H, L * p->rsz));
      TRY(dmalloc(p, &slot->tw, L * p->csz));
      slot->cap_lg = lg;
    }
    TRY(upload_real_table(p, slot->H, gaussfft_table(lg, w)));
    TRY(upload_twiddles(p, slot->tw, (int)L));
    slot->sigma = sigma;
    slot->lg = lg;
  }
  slot->stamp = ++p->gft_clock;
  *H = slot->H;
  *tw = slot->tw;
  return GPA_OK;
}

// out = [minuend -] scipy.ndimage.gaussian_filter(in, sigma) (axis 0 first, then axis 1; `tmp` holds the intermediate).
// Kernels of radius >= GAUSS_FFT_MINR (default 12) run as overlap-save FFT convolutions (gpa_gaussfft.hip), shorter ones as
// direct sums in double (gauss1d_kernel); the filter weights of the direct form live in d_scratch[1024 ...].
int gaussian_filter_dev(gpa_plan* p, const void* in, void* tmp, void* out, double sigma, const void* minuend) {
  std::vector<double> w;
  const int R = gaussian_weights(sigma, w), n0 = p->n0, n1 = p->n1;
  hipStream_t st = p->stream;
  const int minr = opt_set(OPT_GAUSS_FFT_MINR) ? (int)opt(OPT_GAUSS_FFT_MINR).num : 12;
  int lg0 = 0, lg1 = 0;
  if (R >= minr) {
    lg0 = gaussfft_choose_lg(p->dtype, n0, R, true);
    lg1 = gaussfft_choose_lg(p->dtype, n1, R, false);
  }
  if (lg0 && lg1) {
    const void *H0, *H1, *tw0, *tw1;
    TRY(gauss_tables(p, 0, sigma, lg0, w, &H0, &tw0));
    TRY(gauss_tables(p, 1, sigma, lg1, w, &H1, &tw1));
    HIP_TRY(launch_gaussfft(p->dtype, lg0, 0, in, tmp, n0, n1, R, H0, tw0, nullptr, st));
    HIP_TRY(launch_gaussfft(p->dtype, lg1, 1, tmp, out, n0, n1, R, H1, tw1, minuend, st));
    return GPA_OK;
  }
  if (2 * R + 1 > 3072) return fail(GPA_ERR_ARG, "gaussian filter: sigma too large for this image (radius > 1535 and no FFT length fits)");
  double* d_w = p->d_scratch + 1024;
  HIP_TRY(hipStreamSynchronize(st));   // the weight slot may still be read by an earlier filter
  HIP_TRY(hipMemcpyAsync(d_w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, st));
  if (gauss2d_small_ok(R) && !opt_set(OPT_NO_GAUSS2D)) {
    HIP_TRY(launch_gauss2d_small(p->dtype, in, out, n0, n1, d_w, R, minuend, st));
  } else {
    HIP_TRY(launch_gauss1d(p->dtype, in, tmp, n0, n1, 0, d_w, R, nullptr, st));
    HIP_TRY(launch_gauss1d(p->dtype, tmp, out, n0, n1, 1, d_w, R, minuend, st));
  }
  HIP_TRY(hipStreamSynchronize(st));   // w is a pageable host vector
  return GPA_OK;
}

// threshold + 3x3 maxima of the smoothed spectrum in p->d_peaksmooth -> host lists (synchronises the stream: the count decides the copy)
static int peaks_collect(gpa_plan* p, double threshold_rel, int max_out, int32_t* coords, void* values, int* count_out) {
  const int n0 = p->n0, n1 = p->n1;
  const size_t npx = (size_t)n0 * n1;
  hipStream_t st = p->stream;
  // candidates land in d_kidx (max_peaks * npx ints, two per candidate) and d_dudy (2 (n0 - 1) n1 reals)
  const size_t cap = std::min((size_t)p->max_peaks * npx / 2, 2 * (size_t)(n0 - 1) * n1);
  if ((size_t)max_out > cap) max_out = (int)cap;
  double* d_thr = p->d_peakws + 2 * PEAK_PARTS;
  int* d_count = reinterpret_cast<int*>(p->d_peakws + 2 * PEAK_PARTS + 8);
  void* d_vals = p->d_dudy;
  HIP_TRY(launch_localmax(p->dtype, p->d_peaksmooth, n0, n1, threshold_rel, p->d_peakws, d_thr, max_out, d_count, p->d_kidx, d_vals, st));
  int count = 0;
  HIP_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int stored = count < max_out ? count : max_out;
  if (stored > 0) {
    HIP_TRY(hipMemcpyAsync(coords, p->d_kidx, (size_t)stored * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(values, d_vals, (size_t)stored * p->rsz, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  *count_out = count;
  return GPA_OK;
}

// image: device pointer (not modified); coords / values / count_out: host; d_smooth_out: device, may be null
int gpa_find_peaks_dev(gpa_plan* p, const void* d_image, double sigma, double dog_sigma, double threshold_rel, int max_out,
                       int32_t* coords, void* values, int* count_out, void* d_smooth_out) {
  if (!p || !d_image || !coords || !values || !count_out) return fail(GPA_ERR_ARG, "gpa_find_peaks: null argument");
  if (!(sigma > 0.0) || max_out < 1) return fail(GPA_ERR_ARG, "gpa_find_peaks: need sigma > 0, max_out >= 1");
  if (p->n0 < 3 || p->n1 < 3) return fail(GPA_ERR_STATE, "gpa_find_peaks: image too small");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  const int n0 = p->n0, n1 = p->n1;
  const size_t npx = (size_t)n0 * n1;
  hipStream_t st = p->stream;
  p->peaks_smooth_valid = false;
  if (!p->d_peakws) TRY(dmalloc(p, (void**)&p->d_peakws, (2 * PEAK_PARTS + 16) * sizeof(double)));
  void* fftim = p->d_wnorm;
  void* tmp = p->d_dudx;
  if (!p->d_peaksmooth) TRY(dmalloc(p, &p->d_peaksmooth, npx * p->rsz));   // its own buffer: stays valid for gpa_find_peaks_again
  void* smooth = p->d_peaksmooth;
  TRY(per_dft_staged(p, d_image, fftim));                                    // |fftshift(p_hat)|, DC = 0
  TRY(gaussian_filter_dev(p, fftim, tmp, smooth, sigma, nullptr));
  if (dog_sigma > 0.0) TRY(gaussian_filter_dev(p, fftim, tmp, smooth, dog_sigma, smooth));
  p->peaks_smooth_valid = true;
  TRY(peaks_collect(p, threshold_rel, max_out, coords, values, count_out));
  if (d_smooth_out) {
    HIP_TRY(hipMemcpyAsync(d_smooth_out, smooth, npx * p->rsz, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  if (p->profiling) collect_kernel_profile(p);
  return GPA_OK;
}

// peak_local_max of the smoothed spectrum that the last gpa_find_peaks[_dev] call left in the plan, at another threshold: what
// the parameter relaxation of extract_primary_ks (geometric_phase_analysis.py:447-467) re-evaluates while sigma stays the same
int gpa_find_peaks_again(gpa_plan* p, double threshold_rel, int max_out, int32_t* coords, void* values, int* count_out) {
  if (!p || !coords || !values || !count_out) return fail(GPA_ERR_ARG, "gpa_find_peaks_again: null argument");
  if (max_out < 1) return fail(GPA_ERR_ARG, "gpa_find_peaks_again: max_out must be >= 1");
  if (!p->peaks_smooth_valid) return fail(GPA_ERR_STATE, "gpa_find_peaks_again: no smoothed spectrum in the plan (call gpa_find_peaks first)");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  TRY(peaks_collect(p, threshold_rel, max_out, coords, values, count_out));
  if (p->profiling) collect_kernel_profile(p);
  return GPA_OK;
}

int gpa_find_peaks(gpa_plan* p, const void* image, double sigma, double dog_sigma, double threshold_rel, int max_out,
                   int32_t* coords, void* values, int* count_out, void* smooth_out) {
  if (!p || !image) return fail(GPA_ERR_ARG, "gpa_find_peaks: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(gpa_find_peaks_dev(p, p->d_image, sigma, dog_sigma, threshold_rel, max_out, coords, values, count_out, nullptr));
  if (smooth_out) {
    HIP_TRY(hipMemcpyAsync(smooth_out, p->d_peaksmooth, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  return GPA_OK;
}

// ---- f-4 (gaussian_deconvolve) -----------------------------------------------------
// d_data: m0 x m1 reals on the device (m = plan shape - 4 dr), d_out likewise
int gpa_gaussian_deconvolve_dev(gpa_plan* p, const void* d_data, int dr, double sigma, double balance, void* d_out) {
  if (!p || !d_data || !d_out) return fail(GPA_ERR_ARG, "gpa_gaussian_deconvolve: null argument");
  if (dr < 0 || !(sigma > 0.0) || !(balance >= 0.0)) return fail(GPA_ERR_ARG, "gpa_gaussian_deconvolve: need dr >= 0, sigma > 0, balance >= 0");
  const int pad = 2 * dr, n0 = p->n0, n1 = p->n1, m0 = n0 - 2 * pad, m1 = n1 - 2 * pad;
  if (m0 < 2 || m1 < 2 || pad >= m0 || pad >= m1)
    return fail(GPA_ERR_STATE, "gpa_gaussian_deconvolve: the plan must have the padded shape (m + 4 dr), with 2 dr < m");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  TRY(dft_axes(p));
  hipStream_t st = p->stream;
  // k-space Gaussian factors (doubles)
  std::vector<double> gx = gaussian_kspace(n0, sigma), gy = gaussian_kspace(n1, sigma);
  double* d_gx = reinterpret_cast<double*>(p->d_aux0);
  double* d_gy = reinterpret_cast<double*>(p->d_aux1);
  HIP_TRY(hipMemcpyAsync(d_gx, gx.data(), (size_t)n0 * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_gy, gy.data(), (size_t)n1 * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(launch_deconv_pack(p->dtype, d_data, m0, m1, pad, p->Tbuf, st));
  HIP_TRY(dft2_inplace(p->dtype, p->bx0, p->bx1, p->Tbuf, &p->dftw, st));
  HIP_TRY(launch_deconv_filter(p->dtype, p->Tbuf, n0, n1, d_gx, d_gy, balance, st));
  HIP_TRY(dft2_inplace(p->dtype, p->bx0, p->bx1, p->Tbuf, &p->dftw, st));
  HIP_TRY(launch_deconv_unpack(p->dtype, p->Tbuf, m0, m1, pad, d_out, st));
  HIP_TRY(hipStreamSynchronize(st));   // gx / gy are pageable host vectors
  if (p->profiling) collect_kernel_profile(p);
  return GPA_OK;
}

int gpa_gaussian_deconvolve(gpa_plan* p, const void* data, int dr, double sigma, double balance, void* out) {
  if (!p || !data || !out) return fail(GPA_ERR_ARG, "gpa_gaussian_deconvolve: null argument");
  const int pad = 2 * dr, m0 = p->n0 - 2 * pad, m1 = p->n1 - 2 * pad;
  if (dr < 0 || m0 < 2 || m1 < 2)
    return fail(GPA_ERR_STATE, "gpa_gaussian_deconvolve: the plan must have the padded shape (m + 4 dr), with 2 dr < m");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipMemcpyAsync(p->d_image, data, (size_t)m0 * m1 * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(gpa_gaussian_deconvolve_dev(p, p->d_image, dr, sigma, balance, p->d_wnorm));
  HIP_TRY(hipMemcpyAsync(out, p->d_wnorm, (size_t)m0 * m1 * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// ---- f-2 -------------------------------------------------------------------------
int gpa_phasegradient2J_dev(gpa_plan* p, const double* kvecs, int P, const void* grads, const void* weights,
                            double nmperpixel, const double* dks, void* J) {
  if (!p || !kvecs || !grads || !weights || !J) return fail(GPA_ERR_ARG, "gpa_phasegradient2J: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_phasegradient2J: need 2 <= P <= 8 (and P <= max_batch)");
  if (!(nmperpixel > 0.0)) return fail(GPA_ERR_ARG, "gpa_phasegradient2J: nmperpixel must be positive");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  double kiso[16];
  for (int i = 0; i < 2 * P; ++i) kiso[i] = kvecs[i] + (dks ? dks[i] : 0.0);
  TRY(stage_kmat(p, kiso, P));
  HIP_TRY(launch_jacobian(p->dtype, grads, weights, p->d_kmat, P, (size_t)p->n0 * p->n1, nmperpixel, dks, J, p->stream));
  if (p->profiling) { HIP_TRY(hipStreamSynchronize(p->stream)); collect_kernel_profile(p); }
  return GPA_OK;
}

// weights of the lock-ins as the reference's callers form them, np.abs(g['lockin']) (P x n0 x n1 complex -> P x n0 x n1 real)
int gpa_lockin_weights_dev(gpa_plan* p, const void* lockins, int P, void* weights) {
  if (!p || !lockins || !weights) return fail(GPA_ERR_ARG, "gpa_lockin_weights: null argument");
  if (P < 1) return fail(GPA_ERR_ARG, "gpa_lockin_weights: P must be >= 1");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  HIP_TRY(launch_cabs(p->dtype, lockins, (size_t)P * p->n0 * p->n1, weights, p->stream));
  if (p->profiling) { HIP_TRY(hipStreamSynchronize(p->stream)); collect_kernel_profile(p); }
  return GPA_OK;
}

int gpa_phasegradient2J(gpa_plan* p, const double* kvecs, int P, const void* grads, const void* weights,
                        double nmperpixel, const double* dks, void* J) {
  if (!p || !kvecs || !grads || !weights || !J) return fail(GPA_ERR_ARG, "gpa_phasegradient2J: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_phasegradient2J: need 2 <= P <= 8 (and P <= max_batch)");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  TRY(ensure_tbuf(p, (P + 1) / 2));
  void* d_J = nullptr;
  HIP_TRY(hipMalloc(&d_J, 4 * npx * p->rsz));
  int rc = GPA_OK;
  hipError_t e = hipMemcpyAsync(p->d_lockin, grads, (size_t)P * npx * 2 * p->rsz, hipMemcpyHostToDevice, p->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(p->Tbuf, weights, (size_t)P * npx * p->rsz, hipMemcpyHostToDevice, p->stream);
  if (e == hipSuccess) rc = gpa_phasegradient2J_dev(p, kvecs, P, p->d_lockin, p->Tbuf, nmperpixel, dks, d_J);
  if (e == hipSuccess && rc == GPA_OK) e = hipMemcpyAsync(J, d_J, 4 * npx * p->rsz, hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  hipFree(d_J);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_phasegradient2J: ") + hipGetErrorString(e));
  return rc;
}

int gpa_props_from_jac_dev(int device, int dtype, size_t npx, const void* jac, int add_identity, double refangle,
                           double refscale, int diff, void* props, void* stream) {
  if (!jac || !props) return fail(GPA_ERR_ARG, "gpa_props_from_jac: null argument");
  if (dtype != GPA_F32 && dtype != GPA_F64) return fail(GPA_ERR_ARG, "gpa_props_from_jac: dtype must be GPA_F32 or GPA_F64");
  if (npx == 0) return GPA_OK;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(launch_props(dtype, jac, npx, add_identity, refangle, refscale, diff, props, (hipStream_t)stream));
  return GPA_OK;
}

int gpa_props_from_jac(int device, int dtype, size_t npx, const void* jac, int add_identity, double refangle,
                       double refscale, int diff, void* props) {
  if (!jac || !props) return fail(GPA_ERR_ARG, "gpa_props_from_jac: null argument");
  if (dtype != GPA_F32 && dtype != GPA_F64) return fail(GPA_ERR_ARG, "gpa_props_from_jac: dtype must be GPA_F32 or GPA_F64");
  if (npx == 0) return GPA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(GPA_ERR_NODEV, "gpa_props_from_jac: no HIP device");
  HIP_TRY(hipSetDevice(device));
  const size_t bytes = 4 * npx * (dtype == GPA_F32 ? 4 : 8);
  void *d_j = nullptr, *d_p = nullptr;
  HIP_TRY(hipMalloc(&d_j, bytes));
  hipError_t e = hipMalloc(&d_p, bytes);
  int rc = GPA_OK;
  if (e == hipSuccess) e = hipMemcpy(d_j, jac, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) rc = gpa_props_from_jac_dev(device, dtype, npx, d_j, add_identity, refangle, refscale, diff, d_p, nullptr);
  if (e == hipSuccess && rc == GPA_OK) e = hipMemcpy(props, d_p, bytes, hipMemcpyDeviceToHost);
  hipFree(d_j);
  hipFree(d_p);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_props_from_jac: ") + hipGetErrorString(e));
  return rc;
}

// ---- f-4 -------------------------------------------------------------------------
bool solve3(const double* m /*uu uv u vv v 1*/, const double* b, double* x) {
  const double A[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  if (!(fabs(det) > 0.0)) return false;
  for (int c = 0; c < 3; ++c) {
    double M[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i][j] = j == c ? b[i] : A[i][j];
    x[c] = (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
            M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) / det;
  }
  return true;
}

int gpa_fit_plane_dev(gpa_plan* p, const void* image, int max_iter, double tol, double* coef, int* iters_out) {
  if (!p || !image || !coef) return fail(GPA_ERR_ARG, "gpa_fit_plane: null argument");
  if (max_iter < 1 || !(tol >= 0.0)) return fail(GPA_ERR_ARG, "gpa_fit_plane: need max_iter >= 1, tol >= 0");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);
  const int n0 = p->n0, n1 = p->n1;
  // centred, unit-scaled coordinates keep the normal matrix well conditioned
  const double cx = 0.5 * (n0 - 1), cy = 0.5 * (n1 - 1), sx = 0.5 * n0, sy = 0.5 * n1;
  double c[3] = {0.0, 0.0, 0.0};   // start at the zero plane like the reference (x0 = [0, 0, 0])
  double sums[10];
  int it = 0;
  for (; it < max_iter; ++it) {
    HIP_TRY(launch_huber_moments(p->dtype, image, n0, n1, c, cx, cy, sx, sy, p->d_scratch, p->stream));
    HIP_TRY(hipMemcpyAsync(sums, p->d_scratch + huber_sums_offset(), sizeof(sums), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    double nc[3];
    if (!solve3(sums, sums + 6, nc)) return fail(GPA_ERR_STATE, "gpa_fit_plane: singular normal equations");
    // change of the fitted plane over the image, in units of the data
    const double step = fabs(nc[0] - c[0]) + fabs(nc[1] - c[1]) + fabs(nc[2] - c[2]);
    c[0] = nc[0]; c[1] = nc[1]; c[2] = nc[2];
    if (step <= tol) { ++it; break; }
  }
  // back to pixel indices: a0 x + a1 y + a2
  coef[0] = c[0] / sx;
  coef[1] = c[1] / sy;
  coef[2] = c[2] - c[0] * cx / sx - c[1] * cy / sy;
  if (p->profiling) collect_kernel_profile(p);
  if (iters_out) *iters_out = it;
  return GPA_OK;
}

int gpa_fit_plane(gpa_plan* p, const void* image, int max_iter, double tol, double* coef, int* iters_out) {
  if (!p || !image || !coef) return fail(GPA_ERR_ARG, "gpa_fit_plane: null argument");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipMemcpyAsync(p->d_image, image, (size_t)p->n0 * p->n1 * p->rsz, hipMemcpyHostToDevice, p->stream));
  return gpa_fit_plane_dev(p, p->d_image, max_iter, tol, coef, iters_out);
}

