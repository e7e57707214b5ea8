// C ABI, a1-a4: lock-ins and the reference-vector sweep (geometric_phase_analysis.py:20-89, :615-889; cuGPA.py:11-202).
#include "gpa_plan.h"

// ---- a1/a2 -------------------------------------------------------------------
int gpa_lockin_batch_dev(gpa_plan* p, const void* image, const double* kvecs, int B, double sigma,
                         void* out) {
  if (!p || !image || !kvecs || !out) return fail(GPA_ERR_ARG, "gpa_lockin_batch: null argument");
  if (B < 1 || B > p->max_batch) return fail(GPA_ERR_STATE, "gpa_lockin_batch: B exceeds the plan's max_batch");
  HIP_TRY(hipSetDevice(p->device));
  TRY(ensure_filters(p, sigma));
  int Bx = 0;
  TRY(stage_kvectors(p, kvecs, kvecs, B, &Bx));
  TRY(ensure_tbuf(p, Bx));
  TRY(run_passA(p, image, nullptr, p->Tbuf, Bx));
  HIP_TRY(launch_passB(p->dtype, p->ax1, p->n0, p->Tbuf, p->Hy, p->tw1, p->tb, B, 1, false, out, nullptr,
                       p->stream));
  return GPA_OK;
}

int gpa_lockin_batch(gpa_plan* p, const void* image, const double* kvecs, int B, double sigma, void* out) {
  if (!p || !image || !kvecs || !out) return fail(GPA_ERR_ARG, "gpa_lockin_batch: null argument");
  if (B < 1 || B > p->max_batch) return fail(GPA_ERR_STATE, "gpa_lockin_batch: B exceeds the plan's max_batch");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  // pass B reads Tbuf, so the B lock-ins land in the plan's grown-on-demand scratch
  TRY(ensure_sf(p, (size_t)B * npx * p->csz));
  TRY(gpa_lockin_batch_dev(p, p->d_image, kvecs, B, sigma, p->d_sf));
  HIP_TRY(hipMemcpyAsync(out, p->d_sf, (size_t)B * npx * p->csz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// pass A over the staged x-planes (per-plane forward transforms, the image tile kept in registers)
int run_passA(gpa_plan* p, const void* image, const void* mean, void* Tbuf, int Bx, int nimg) {
  HIP_TRY(launch_passA(p->dtype, p->ax0, p->n1, image, mean, p->tb, p->Hx, p->tw0, Tbuf, Bx, p->stream, nimg));
  return GPA_OK;
}

// pass B with selection.  A small image has few rows to spread over the 256 CUs and runs its K candidates one after
// the other in each workgroup (512^2, K = 16: 82 us, a quarter of the image's time): there the candidates are split over
// up to 4 workgroups per row and merged (launch_passB_split) -- same winners, same values.
// raw: the caller's consumer is reconstruct_setup (which takes the compensation's phase step): where the shared kernel
// runs it then skips its second visit of the winner rows; p->lk_raw says whether it did
int passB_select(gpa_plan* p, int P, int K, void* lockin, int32_t* kidx, bool raw) {
  p->lk_raw = false;
  const int rows_wg = (p->n0 + 7) / 8 * P;           // workgroups of the unsplit launch (at least: NF <= 8 rows each)
  int ksplit = 1;
  // (only while the unsplit launch has fewer workgroups than the chip has CUs: at 1024^2, 384 workgroups, the split
  //  measured slower -- 153 -> 188 us -- because the merge pass and the partial slabs cost more than they save)
  if (p->ax1.lg <= 10 && K >= 4 && !p->no_ksplit && rows_wg <= 256)
    while (ksplit < 4 && ksplit * 2 <= K && rows_wg * ksplit < 1024) ksplit *= 2;
  if (ksplit == 1) {
    TRY(shared_prepare(p, P, K));
    if (p->sh_use) {
      p->lk_raw = raw && !opt_set(OPT_NO_RAW) && p->sh_one_kref;
      HIP_TRY(launch_passB_shared(p->dtype, p->ax1s, p->n0, p->Tbuf, p->ax1s.L == p->ax1.L ? p->tw1 : p->tw1s, p->tb,
                                  p->sh, p->sh_E, p->sh_Epad, P, K, lockin, kidx, p->stream, 1, 0, p->sh_elems, p->sh_nbl,
                                  p->lk_raw));
    } else
      HIP_TRY(launch_passB(p->dtype, p->ax1, p->n0, p->Tbuf, p->Hy, p->tw1, p->tb, P, K, true, lockin, kidx, p->stream));
    return GPA_OK;
  }
  const size_t npx = (size_t)p->n0 * p->n1, cnt = (size_t)ksplit * P * npx;
  TRY(ensure_sf(p, cnt * (p->csz + sizeof(int32_t))));
  void* part = p->d_sf;
  int32_t* pidx = reinterpret_cast<int32_t*>((char*)p->d_sf + cnt * p->csz);
  HIP_TRY(launch_passB_split(p->dtype, p->ax1, p->n0, p->Tbuf, p->Hy, p->tw1, p->tb, P, K, ksplit, part, pidx, lockin, kidx,
                             p->stream));
  return GPA_OK;
}

// per-kernel event pairs of a profiled call, summed by name in order of first appearance -> p->kprof_table
void collect_kernel_profile(gpa_plan* p) {
  std::vector<std::string> names;
  std::vector<int> calls;
  std::vector<double> total;
  for (int i = 0; p->kprof && i < p->kprof->n; ++i) {
    const KernelProfiler::Rec& r = p->kprof->rec[i];
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    size_t j = 0;
    while (j < names.size() && names[j] != r.name) ++j;
    if (j == names.size()) { names.push_back(r.name); calls.push_back(0); total.push_back(0.0); }
    ++calls[j];
    total[j] += ms;
  }
  p->kprof_table.clear();
  char line[160];
  for (size_t j = 0; j < names.size(); ++j) {
    snprintf(line, sizeof(line), "%s %d %.6f\n", names[j].c_str(), calls[j], total[j]);
    p->kprof_table += line;
  }
}

// ---- a3 ----------------------------------------------------------------------
int sweep_peaks_dev(gpa_plan* p, const void* image, const void* mean, const double* krefs, int P,
                           const double* klists, int K, double sigma, void* lockin, int32_t* kidx, bool raw) {
  const int B = P * K;
  if (B > p->max_batch) return fail(GPA_ERR_STATE, "sweep: P*K exceeds the plan's max_batch");
  TRY(ensure_filters(p, sigma));
  std::vector<double> kr((size_t)B * 2);
  for (int pp = 0; pp < P; ++pp)
    for (int k = 0; k < K; ++k) {
      kr[2 * ((size_t)pp * K + k)] = krefs[2 * pp];
      kr[2 * ((size_t)pp * K + k) + 1] = krefs[2 * pp + 1];
    }
  int Bx = 0;
  TRY(stage_kvectors(p, klists, kr.data(), B, &Bx));
  TRY(ensure_tbuf(p, Bx));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[1], p->stream));
  TRY(run_passA(p, image, mean, p->Tbuf, Bx));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[2], p->stream));
  TRY(passB_select(p, P, K, lockin, kidx, raw));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[3], p->stream));
  return GPA_OK;
}

// one peak: stage tables, pass A, then pass B in the requested selection mode
int sweep_one_peak(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                          int mode, void* lockin, int32_t* kidx, const uint8_t* d_gate, void* d_psi) {
  if (K > p->max_batch) return fail(GPA_ERR_STATE, "gpa_sweep: K exceeds the plan's max_batch");
  TRY(ensure_filters(p, sigma));
  std::vector<double> kr((size_t)K * 2);
  for (int k = 0; k < K; ++k) { kr[2 * k] = kref[0]; kr[2 * k + 1] = kref[1]; }
  int Bx = 0;
  TRY(stage_kvectors(p, klist, kr.data(), K, &Bx));
  TRY(ensure_tbuf(p, Bx));
  TRY(run_passA(p, image, nullptr, p->Tbuf, Bx));
  HIP_TRY(launch_passB_ext(p->dtype, p->ax1, p->n0, p->Tbuf, p->Hy, p->tw1, p->tb, K, mode, lockin, kidx, d_gate, d_psi,
                           p->stream));
  return GPA_OK;
}

int gpa_sweep_grad_dev(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                       int grad_mode, void* lockin, int32_t* kidx, void* grad) {
  if (!p || !image || !kref || !klist || !lockin || !grad) return fail(GPA_ERR_ARG, "gpa_sweep_grad: null argument");
  if (K < 1) return fail(GPA_ERR_ARG, "gpa_sweep_grad: K must be >= 1");
  if (grad_mode < 0 || grad_mode > 2) return fail(GPA_ERR_ARG, "gpa_sweep_grad: grad_mode must be 0, 1 or 2");
  HIP_TRY(hipSetDevice(p->device));
  const bool top = g_kprof == nullptr;   // (called directly, not from sweep_host: this call owns the profile)
  ProfInstallIf prof(p, top);
  // a4: the winner is selected in registers by pass B as in the plain sweep; what the gradient stencil needs from
  // the OTHER candidates is only the phase of the winner's candidate at the four neighbours, so pass B also writes
  // one real per pixel and candidate (K reals instead of the K complex lock-ins of the first build)
  const size_t npx = (size_t)p->n0 * p->n1;
  TRY(ensure_sf(p, (size_t)K * npx * p->rsz));
  int32_t* ki = kidx ? kidx : p->d_kidx;
  // rows the shared-forward pass B takes (2048- / 4096- / f32 8192-point classes, lists with runs on their x-planes): one
  // forward transform per x-plane row instead of one per candidate, the phases written by that kernel (round 6; NO_SHARED or
  // NO_SHARED_PHASES keep the per-candidate kernel: the same gradient up to rounding, tests/test_gpu_shared_passb.py)
  bool shared = false;
  if (K <= p->max_batch && p->use_shared && !opt_set(OPT_NO_SHARED_PHASES)) {
    TRY(ensure_filters(p, sigma));
    std::vector<double> kr((size_t)K * 2);
    for (int k = 0; k < K; ++k) { kr[2 * k] = kref[0]; kr[2 * k + 1] = kref[1]; }
    int Bx = 0;
    TRY(stage_kvectors(p, klist, kr.data(), K, &Bx));
    TRY(shared_prepare(p, 1, K));
    if (p->sh_use && p->sh_one_kref) {
      TRY(ensure_tbuf(p, Bx));
      TRY(run_passA(p, image, nullptr, p->Tbuf, Bx));
      const hipError_t e = launch_passB_shared_phases(p->dtype, p->ax1s, p->n0, p->Tbuf, p->ax1s.L == p->ax1.L ? p->tw1 : p->tw1s,
                                                      p->tb, p->sh, p->sh_E, p->sh_Epad, 1, K, lockin, ki, p->d_sf, p->stream, 0,
                                                      p->sh_elems, p->sh_nbl);
      if (e == hipSuccess) {
        shared = true;
        HIP_TRY(launch_phasegrad(p->dtype, p->d_sf, K, ki, p->n0, p->n1, p->d_kl, p->d_kr, grad_mode, grad, p->stream, p->d_ystep));
      } else if (e != hipErrorInvalidValue) {
        return fail(GPA_ERR_HIP, std::string("shared pass B (phases): ") + hipGetErrorString(e));
      }
    }
  }
  if (!shared) {
    TRY(sweep_one_peak(p, image, kref, klist, K, sigma, 3, lockin, ki, nullptr, p->d_sf));
    HIP_TRY(launch_phasegrad(p->dtype, p->d_sf, K, ki, p->n0, p->n1, p->d_kl, p->d_kr, grad_mode, grad, p->stream));
  }
  if (top && p->profiling) { HIP_TRY(hipStreamSynchronize(p->stream)); collect_kernel_profile(p); }
  return GPA_OK;
}

int gpa_sweep_dev(gpa_plan* p, const void* image, const double* kref, const double* klist, int K,
                  double sigma, void* lockin, int32_t* kidx, void* grad) {
  if (!p || !image || !kref || !klist || !lockin) return fail(GPA_ERR_ARG, "gpa_sweep: null argument");
  if (K < 1) return fail(GPA_ERR_ARG, "gpa_sweep: K must be >= 1");
  HIP_TRY(hipSetDevice(p->device));
  if (!grad) return sweep_peaks_dev(p, image, nullptr, kref, 1, klist, K, sigma, lockin, kidx);
  return gpa_sweep_grad_dev(p, image, kref, klist, K, sigma, 0, lockin, kidx, grad);
}

int sweep_host(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                      int grad_mode, const uint8_t* gate, void* lockin, int32_t* kidx, void* grad) {
  if (!p || !image || !kref || !klist || !lockin) return fail(GPA_ERR_ARG, "gpa_sweep: null argument");
  if (K < 1) return fail(GPA_ERR_ARG, "gpa_sweep: K must be >= 1");
  HIP_TRY(hipSetDevice(p->device));
  ProfInstall prof(p);   // (gpa_set_profiling: which kernels this sweep ran, through gpa_last_kernel_profile)
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  if (grad && !p->d_grad) TRY(dmalloc(p, &p->d_grad, 2 * npx * p->rsz));
  if (gate) {
    TRY(ensure_sf(p, (size_t)K * K));
    HIP_TRY(hipMemcpyAsync(p->d_sf, gate, (size_t)K * K, hipMemcpyHostToDevice, p->stream));
    TRY(sweep_one_peak(p, p->d_image, kref, klist, K, sigma, 2, p->d_lockin, p->d_kidx, (const uint8_t*)p->d_sf, nullptr));
  } else if (grad) {
    TRY(gpa_sweep_grad_dev(p, p->d_image, kref, klist, K, sigma, grad_mode, p->d_lockin, p->d_kidx, p->d_grad));
  } else {
    TRY(sweep_peaks_dev(p, p->d_image, nullptr, kref, 1, klist, K, sigma, p->d_lockin, p->d_kidx));
  }
  if (grad) HIP_TRY(hipMemcpyAsync(grad, p->d_grad, 2 * npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(lockin, p->d_lockin, npx * p->csz, hipMemcpyDeviceToHost, p->stream));
  if (kidx) HIP_TRY(hipMemcpyAsync(kidx, p->d_kidx, npx * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->profiling) collect_kernel_profile(p);
  return GPA_OK;
}

int gpa_sweep(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
              void* lockin, int32_t* kidx, void* grad) {
  return sweep_host(p, image, kref, klist, K, sigma, 0, nullptr, lockin, kidx, grad);
}

int gpa_sweep_grad(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                   int grad_mode, void* lockin, int32_t* kidx, void* grad) {
  if (!grad) return fail(GPA_ERR_ARG, "gpa_sweep_grad: null argument");
  if (grad_mode < 0 || grad_mode > 2) return fail(GPA_ERR_ARG, "gpa_sweep_grad: grad_mode must be 0, 1 or 2");
  return sweep_host(p, image, kref, klist, K, sigma, grad_mode, nullptr, lockin, kidx, grad);
}

int gpa_sweep_gated(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                    const uint8_t* gate, void* lockin, int32_t* kidx) {
  if (!gate) return fail(GPA_ERR_ARG, "gpa_sweep_gated: null argument");
  return sweep_host(p, image, kref, klist, K, sigma, 0, gate, lockin, kidx, nullptr);
}

