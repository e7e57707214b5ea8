// C ABI of libgpa_hip.so (see include/gpa_hip.h): plans, host-built filter
// tables, and the drivers that chain the kernels on the plan's stream.
#include "gpa_plan.h"

static thread_local std::string g_err;
int gpa_fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
// ---- run-time options -------------------------------------------------------------------------
namespace gpa {
static const char* const kOptNames[OPT_COUNT] = {
    "PBS_FULLBAND", "SERIAL_UNWRAP", "NO_WORKER", "NO_KSPLIT", "NO_COMPACT", "NO_SHARED",
    "NO_PAIR", "PBS_E8", "TRI_SMALL", "TRI_Q", "NO_MR", "MR_FORCE_BLUESTEIN", "NO_ROWPQ", "COLSOLVE", "NO_LAT",
    "F32_EPS_FLOOR", "COLSTREAM_CHUNK", "NO_ROWHALF", "PAIR_MAXSIDE", "ROWHALF_MINLG", "NO_PQDCT",
    "NO_REORDER", "NO_RAW", "NO_TILEFUSE", "NO_ROWPERS", "NO_LFTILE", "LF_ALL_ROUNDS", "DFT_ENGINE", "GAUSS_FFT_MINR", "NO_GAUSS2D", "NO_DFT_HALF", "F32_STALL", "PBS_LDS_PAD", "NO_SHARED_PHASES", "PA_STAG", "PA_STAG_TICKS", "PA_ROT"};
static OptVal g_opts[OPT_COUNT];
static std::once_flag g_opts_once;
static void opt_assign(OptVal& o, const char* value) {
  o.set = value != nullptr;
  o.num = value ? atof(value) : 0.0;
  memset(o.str, 0, sizeof(o.str));
  if (value) strncpy(o.str, value, sizeof(o.str) - 1);
}
static void opts_from_env() {
  for (int k = 0; k < OPT_COUNT; ++k) opt_assign(g_opts[k], getenv((std::string("GPA_") + kOptNames[k]).c_str()));
}
const OptVal& opt(OptKey k) {
  std::call_once(g_opts_once, opts_from_env);
  return g_opts[k];
}
}  // namespace gpa

Axis make_axis(int n) {
  Axis a;
  a.n = n;
  int lg = 0;
  while ((1 << lg) < n) ++lg;
  if ((1 << lg) == n) {
    a.padded = false;
  } else {
    lg = 0;
    while ((1 << lg) < 2 * n - 1) ++lg;
    a.padded = true;
  }
  if (lg < 6) {   // shortest supported transform is 64: run tiny axes in padded mode
    lg = 6;
    a.padded = ((1 << lg) != n);
  }
  a.lg = lg;
  a.L = 1 << lg;
  a.extL = a.padded ? n - 1 : 0;
  a.extR = 0;
  return a;
}

// the axis geometry for a kernel whose taps vanish beyond E samples: the compact extension if it allows a shorter
// transform than the full one
Axis compact_axis(const Axis& full, int E) {
  if (!full.padded || E >= (full.n - 1) / 2) return full;
  int lg = 6;
  while ((1 << lg) < full.n + 2 * E) ++lg;
  if (lg >= full.lg) return full;
  Axis a = full;
  a.lg = lg;
  a.L = 1 << lg;
  a.extL = a.extR = E;
  return a;
}

int upload_real_table(gpa_plan* p, void* dst, const std::vector<double>& v) {
  if (p->dtype == GPA_F32) HIP_TRY(upload_as<float>(dst, v, p->stream));
  else HIP_TRY(upload_as<double>(dst, v, p->stream));
  return GPA_OK;
}

int dmalloc(gpa_plan* p, void** ptr, size_t bytes) {
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(ptr, bytes);
  if (e != hipSuccess)
    return fail(GPA_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  p->ws_bytes += bytes;
  return GPA_OK;
}
int plan_build(gpa_plan* p) {
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&p->ev0));
  HIP_TRY(hipEventCreate(&p->ev1));
  for (auto& e : p->stage_ev) HIP_TRY(hipEventCreate(&e));
  const size_t npx = (size_t)p->n0 * p->n1;
  const int B = p->max_batch;
  for (int ax = 0; ax < 2; ++ax) {
    const Axis& a = ax == 0 ? p->ax0 : p->ax1;
    std::vector<double> t((size_t)2 * a.L);
    for (int k = 0; k < a.L; ++k) {
      t[2 * k] = cos(-2.0 * M_PI * k / a.L);
      t[2 * k + 1] = sin(-2.0 * M_PI * k / a.L);
    }
    void** dst = ax == 0 ? &p->tw0 : &p->tw1;
    TRY(dmalloc(p, dst, (size_t)a.L * p->csz));
    TRY(upload_real_table(p, *dst, t));
  }
  TRY(dmalloc(p, &p->Hx, (size_t)p->ax0.L * p->csz));
  TRY(dmalloc(p, &p->Hy, (size_t)p->ax1.L * p->csz));
  // pass A writes one plane per distinct wx, not per candidate (a 6x6 grid x 3 peaks needs 18 planes, not 108):
  // start with what the non-sweep users need (<= 8 real planes) and grow in ensure_tbuf()
  p->tbuf_planes = B < 4 ? B : 4;
  TRY(dmalloc(p, &p->Tbuf, (size_t)p->tbuf_planes * npx * p->csz));
  // (carrier base tables: one entry per thread of a transform, L / 16)
  TRY(dmalloc(p, &p->tb.cxb, (size_t)B * std::max(p->ax0.L / 16, 256) * p->csz));
  TRY(dmalloc(p, &p->tb.sx, (size_t)B * 16 * p->csz));
  TRY(dmalloc(p, &p->tb.wxw, (size_t)B * p->csz));
  TRY(dmalloc(p, &p->tb.wxr, (size_t)B * p->csz));
  TRY(dmalloc(p, &p->tb.cyb, (size_t)B * std::max(p->ax1.L / 16, 256) * p->csz));
  TRY(dmalloc(p, &p->tb.sy, (size_t)B * 16 * p->csz));
  TRY(dmalloc(p, &p->tb.wyw, (size_t)B * p->csz));
  TRY(dmalloc(p, &p->tb.wyr, (size_t)B * p->csz));
  TRY(dmalloc(p, (void**)&p->tb.planeof, (size_t)B * sizeof(int)));
  TRY(dmalloc(p, (void**)&p->d_pw, (size_t)B * sizeof(double)));
  TRY(dmalloc(p, &p->tb.dx, (size_t)B * p->n0 * p->csz));
  TRY(dmalloc(p, &p->tb.dy, (size_t)B * p->n1 * p->csz));
  TRY(dmalloc(p, (void**)&p->d_kl, (size_t)B * 2 * sizeof(double)));
  TRY(dmalloc(p, (void**)&p->d_kr, (size_t)B * 2 * sizeof(double)));
  HIP_TRY(hipHostMalloc((void**)&p->h_k, ((size_t)B * 6 + 32) * sizeof(double)));
  HIP_TRY(hipHostMalloc((void**)&p->h_iters, 8 * sizeof(int)));
  TRY(dmalloc(p, &p->d_image, npx * p->rsz));
  TRY(dmalloc(p, &p->d_mean, 16));
  TRY(dmalloc(p, &p->d_tile_mean, 16));
  TRY(dmalloc(p, (void**)&p->d_scratch, 16384 * sizeof(double)));
  TRY(dmalloc(p, &p->d_aux0, (size_t)p->n0 * 2 * sizeof(double)));
  TRY(dmalloc(p, &p->d_aux1, (size_t)p->n1 * 2 * sizeof(double)));
  p->max_peaks = B < 8 ? B : 8;
  TRY(dmalloc(p, &p->d_lockin, (size_t)p->max_peaks * npx * p->csz));
  TRY(dmalloc(p, (void**)&p->d_kidx, (size_t)p->max_peaks * npx * sizeof(int32_t)));
  TRY(dmalloc(p, &p->d_dudx, 2 * npx * p->rsz));
  TRY(dmalloc(p, &p->d_dudy, 2 * npx * p->rsz));
  TRY(dmalloc(p, &p->d_wnorm, npx * p->rsz));
  TRY(dmalloc(p, &p->d_u, 2 * npx * p->rsz));
  TRY(dmalloc(p, (void**)&p->d_kmat, (size_t)p->max_peaks * 2 * sizeof(double)));
  if (!p->spectral_only) {
    size_t before = 0;
    hipError_t e = unwrap_workspace_create(p->dtype, p->n0, p->n1, p->stream, &p->uw, &before);
    if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap workspace: ") + hipGetErrorString(e));
    p->ws_bytes += before;
  }
  return GPA_OK;
}


int upload_twiddles(gpa_plan* p, void* dst, int L) {
  std::vector<double> t((size_t)2 * L);
  for (int k = 0; k < L; ++k) {
    t[2 * k] = cos(-2.0 * M_PI * k / L);
    t[2 * k + 1] = sin(-2.0 * M_PI * k / L);
  }
  return upload_real_table(p, dst, t);
}

// ---------------------------------------------------------------------------
// exported functions
// ---------------------------------------------------------------------------
extern "C" {

int gpa_version(void) { return 100; }

int gpa_set_option(const char* name, const char* value) {
  if (!name) return fail(GPA_ERR_ARG, "gpa_set_option: null name");
  if (strncmp(name, "GPA_", 4) == 0) name += 4;
  (void)opt(OPT_NO_LAT);   // make sure the environment has been read first
  for (int k = 0; k < OPT_COUNT; ++k)
    if (strcmp(name, kOptNames[k]) == 0) {
      opt_assign(g_opts[k], value);
      return GPA_OK;
    }
  return fail(GPA_ERR_ARG, std::string("gpa_set_option: unknown option ") + name);
}
const char* gpa_last_error(void) { return g_err.c_str(); }

int gpa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

gpa_plan* gpa_plan_create(int device, int n0, int n1, int max_batch, int dtype) {
  if (n0 < 4 || n1 < 4 || max_batch < 1 || (dtype != GPA_F32 && dtype != GPA_F64)) {
    fail(GPA_ERR_ARG, "gpa_plan_create: need n0,n1 >= 4, max_batch >= 1, dtype in {GPA_F32, GPA_F64}");
    return nullptr;
  }
  int ndev = gpa_device_count();
  if (ndev <= 0 || device < 0 || device >= ndev) {
    fail(GPA_ERR_NODEV, "gpa_plan_create: no such GPU (device " + std::to_string(device) + " of " +
                            std::to_string(ndev) + ")");
    return nullptr;
  }
  gpa_plan* p = new gpa_plan();
  p->device = device;
  p->dtype = dtype;
  p->n0 = n0;
  p->n1 = n1;
  p->max_batch = max_batch;
  p->rsz = dtype == GPA_F32 ? 4 : 8;
  p->csz = 2 * p->rsz;
  p->ax0 = p->ax0_full = make_axis(n0);
  p->ax1 = p->ax1_full = make_axis(n1);
  p->serial_unwrap = opt_set(OPT_SERIAL_UNWRAP);
  p->use_worker = !opt_set(OPT_NO_WORKER);
  p->no_ksplit = opt_set(OPT_NO_KSPLIT);
  p->no_compact = opt_set(OPT_NO_COMPACT);
  p->use_shared = !opt_set(OPT_NO_SHARED);
  const int maxlg = dtype == GPA_F32 ? 14 : 13;
  if (n0 > 65536 || n1 > 65536) {
    fail(GPA_ERR_ARG, "gpa_plan_create: axis longer than 65536");
    delete p;
    return nullptr;
  }
  // beyond an LDS-resident transform (f32: 16384 pow2 / 8192 other; f64: 8192 pow2 / 4096 other) the plan has no sweep / unwrap
  p->spectral_only = p->ax0.lg > maxlg || p->ax1.lg > maxlg;
  if (plan_build(p) != GPA_OK) {
    std::string keep = g_err;
    gpa_plan_destroy(p);
    g_err = keep;
    return nullptr;
  }
  return p;
}

void gpa_plan_destroy(gpa_plan* p) {
  if (!p) return;
  hipSetDevice(p->device);
  delete p->worker;
  p->worker = nullptr;
  if (p->stream) hipStreamSynchronize(p->stream);
  void* bufs[] = {p->tw0, p->tw1, p->Hx, p->Hy, p->Tbuf, p->tb.cxb, p->tb.sx, p->tb.wxw, p->tb.wxr, p->tb.cyb, p->tb.sy, p->tb.wyw, p->tb.wyr, p->tb.planeof, p->d_pw,
                  p->tb.dx, p->tb.dy, p->d_kl, p->d_kr, p->d_image, p->d_mean, p->d_tile_mean, p->d_scratch,
                  p->d_lockin, p->d_kidx, p->d_dudx, p->d_dudy, p->d_wnorm, p->d_u, p->d_kmat, p->d_sf, p->d_grad, p->d_aux0, p->d_aux1,
                  p->sh.Gb, p->sh.psi, p->sh.dyc, p->sh.gtab, p->sh.desc, p->sh.order, p->d_taps, p->tw1s,
                  p->sh.pre, p->sh.rot16, p->d_wys, p->d_shifts, p->d_ystep};
  for (void* b : bufs)
    if (b) hipFree(b);
  unwrap_workspace_destroy(&p->uw);
  unwrap_workspace_destroy(&p->uw2);
  if (p->uwb_images) unwrap_workspace_destroy(&p->uwb);
  if (p->have_uwp) unwrap_workspace_destroy(&p->uwp);
  if (p->d_wnorm_b) (void)hipFree(p->d_wnorm_b);
  if (p->h_iters_b) (void)hipHostFree(p->h_iters_b);
  for (void* b : {p->bT, p->bL, p->bMean, (void*)p->bScratch})
    if (b) (void)hipFree(b);
  if (p->stream2) { hipStreamSynchronize(p->stream2); hipStreamDestroy(p->stream2); }
  if (p->ev_fork) hipEventDestroy(p->ev_fork);
  if (p->ev_join) hipEventDestroy(p->ev_join);
  if (p->ev_x) hipEventDestroy(p->ev_x);
  if (p->d_tsum_part) (void)hipFree(p->d_tsum_part);
  warp_ws_free(&p->warp);
  dft_axis_destroy(&p->bx0);
  dft_axis_destroy(&p->bx1);
  dft_work_free(&p->dftw);
  for (auto& ax : p->gft)
    for (auto& g : ax) {
      if (g.H) (void)hipFree(g.H);
      if (g.tw) (void)hipFree(g.tw);
    }
  if (p->d_pertab) (void)hipFree(p->d_pertab);
  if (p->d_peakws) (void)hipFree(p->d_peakws);
  if (p->d_peaksmooth) (void)hipFree(p->d_peaksmooth);
  if (p->h_k) hipHostFree(p->h_k);
  if (p->h_iters) hipHostFree(p->h_iters);
  if (p->kprof) {
    for (int i = 0; i < p->kprof->npool; ++i) hipEventDestroy(p->kprof->pool[i]);
    delete p->kprof;
  }
  if (p->copy_stream) { hipStreamSynchronize(p->copy_stream); hipStreamDestroy(p->copy_stream); }
  if (p->ev_dl_ready) hipEventDestroy(p->ev_dl_ready);
  for (auto e : p->ev_dl_done)
    if (e) hipEventDestroy(e);
  if (p->ev0) hipEventDestroy(p->ev0);
  if (p->ev1) hipEventDestroy(p->ev1);
  for (auto e : p->stage_ev)
    if (e) hipEventDestroy(e);
  if (p->stream) hipStreamDestroy(p->stream);
  delete p;
}

int gpa_plan_sync(gpa_plan* p) {
  if (!p) return fail(GPA_ERR_ARG, "null plan");
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->stream2) HIP_TRY(hipStreamSynchronize(p->stream2));
  if (p->copy_stream) HIP_TRY(hipStreamSynchronize(p->copy_stream));
  return GPA_OK;
}
size_t gpa_plan_workspace_bytes(const gpa_plan* p) { return p ? p->ws_bytes : 0; }
void* gpa_plan_stream(const gpa_plan* p) { return p ? (void*)p->stream : nullptr; }
int gpa_plan_fft_len(const gpa_plan* p, int axis) {
  if (!p) return 0;
  return axis == 0 ? p->ax0.L : p->ax1.L;
}

// ---- timing --------------------------------------------------------------------
int gpa_timer_start(gpa_plan* p) {
  if (!p) return fail(GPA_ERR_ARG, "null plan");
  HIP_TRY(hipEventRecord(p->ev0, p->stream));
  return GPA_OK;
}
int gpa_timer_stop(gpa_plan* p, float* ms_out) {
  if (!p || !ms_out) return fail(GPA_ERR_ARG, "null argument");
  HIP_TRY(hipEventRecord(p->ev1, p->stream));
  HIP_TRY(hipEventSynchronize(p->ev1));
  HIP_TRY(hipEventElapsedTime(ms_out, p->ev0, p->ev1));
  return GPA_OK;
}
int gpa_set_profiling(gpa_plan* p, int on) {
  if (!p) return fail(GPA_ERR_ARG, "null plan");
  p->profiling = on != 0;
  return GPA_OK;
}
int gpa_last_kernel_profile(gpa_plan* p, char* out, size_t cap) {
  if (!p || !out || cap == 0) return fail(GPA_ERR_ARG, "null argument");
  if (p->kprof_table.size() + 1 > cap) return fail(GPA_ERR_STATE, "gpa_last_kernel_profile: buffer too small");
  memcpy(out, p->kprof_table.c_str(), p->kprof_table.size() + 1);
  return GPA_OK;
}

// ---- downloads overlapped with later work --------------------------------------------------
int gpa_download_async(gpa_plan* p, void* host_dst, const void* dev_src, size_t bytes, int slot) {
  if (!p || !host_dst || !dev_src) return fail(GPA_ERR_ARG, "gpa_download_async: null argument");
  if (slot < 0 || slot >= 4) return fail(GPA_ERR_ARG, "gpa_download_async: slot must be 0..3");
  HIP_TRY(hipSetDevice(p->device));
  if (!p->copy_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_dl_ready, hipEventDisableTiming));
    for (auto& e : p->ev_dl_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // after everything enqueued on the plan so far (the driver leaves its second stream joined into the first)
  HIP_TRY(hipEventRecord(p->ev_dl_ready, p->stream));
  HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_dl_ready, 0));
  HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, p->copy_stream));
  HIP_TRY(hipEventRecord(p->ev_dl_done[slot], p->copy_stream));
  return GPA_OK;
}

int gpa_download_wait(gpa_plan* p, int slot) {
  if (!p) return fail(GPA_ERR_ARG, "null plan");
  if (slot < 0 || slot >= 4) return fail(GPA_ERR_ARG, "gpa_download_wait: slot must be 0..3");
  if (!p->copy_stream) return GPA_OK;   // nothing was ever enqueued
  HIP_TRY(hipEventSynchronize(p->ev_dl_done[slot]));
  return GPA_OK;
}

int gpa_last_stage_ms(gpa_plan* p, float* ms5) {
  if (!p || !ms5) return fail(GPA_ERR_ARG, "null argument");
  for (int i = 0; i < 5; ++i) ms5[i] = p->stage_ms[i];
  return GPA_OK;
}

}  // extern "C"
