// Internal header of the C-ABI translation units (gpa_api*.hip): the plan object, error / try macros and the helpers the
// entry-point files share.  Not part of the public interface (include/gpa_hip.h is).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <complex>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <limits>
#include <vector>

#include "../../include/gpa_hip.h"
#include "gpa_internal.h"
#include "gpa_passb_shared.h"
#include "gpa_unwrap.h"
#include "gpa_dft.h"
#include "gpa_gaussfft.h"

using namespace gpa;

int gpa_fail(int code, const std::string& msg);   // sets gpa_last_error() of the calling thread, returns code
static inline int fail(int code, const std::string& msg) { return gpa_fail(code, msg); }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(GPA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
  } while (0)

#define NEED_UNWRAP(p, what)                                                                                     \
  do {                                                                                                             \
    if (!(p)->uw.impl)                                                                                             \
      return fail(GPA_ERR_STATE, std::string(what) + ": this plan's image is too large for the sweep / unwrap kernels " \
                                 "(an axis above 16384 pow2 / 8192 other in f32, 8192 / 4096 in f64); it serves per, "  \
                                 "find_peaks, gaussian_deconvolve and the per-pixel entry points only");                 \
  } while (0)

#define TRY(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != GPA_OK) return _r; \
  } while (0)

// One helper thread per plan: the fused driver enqueues the second displacement component's ~45 launches from it
// while the calling thread enqueues the first component's.  A 4096^2 image does not care (the GPU is the limit), but
// a call costs ~0.4 ms of host time for its ~110 launches, which IS the limit below ~1024^2 (tools/enqueue_cost.py).
struct EnqueueWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, quit = false;
  explicit EnqueueWorker(int device) {
    th = std::thread([this, device] {
      (void)hipSetDevice(device);
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        j();
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return done; });
  }
  ~EnqueueWorker() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

struct gpa_plan {
  int device = 0, dtype = 0, n0 = 0, n1 = 0, max_batch = 0;
  // an axis beyond what one workgroup transforms (f32: 16384 pow2 / 8192 other; f64: 8192 pow2 / 4096 other): no sweep and no
  // unwrap workspace -- the plan serves the plain-DFT rows (per, find_peaks, gaussian_deconvolve: any axis up to 65536) and
  // the per-pixel kernels (reconstruct, Lawler-Fujita, Jacobian / properties, plane fit)
  bool spectral_only = false;
  Axis ax0{}, ax1{};              // the geometry in use (depends on sigma for non-power-of-two axes)
  Axis ax0_full{}, ax1_full{};    // the plan's largest geometry (L >= 2n - 1): what the tables are sized for
  hipStream_t stream = nullptr;
  size_t rsz = 4, csz = 8;        // bytes per real / complex element
  size_t ws_bytes = 0;
  // device buffers
  void* tw0 = nullptr;            // twiddle tables exp(-2 pi i t / L)
  void* tw1 = nullptr;
  void* Hx = nullptr;             // filter tables for the current sigma
  void* Hy = nullptr;
  double sigma_cached = -1.0;
  // shared-forward pass B (gpa_passb_shared.h): per-sigma taps, per-(sigma, k-list) candidate tables
  PassBSharedTables sh{};
  double* d_taps = nullptr;       // g(0 .. sh_etab) of the y axis' circular filter, doubles
  int sh_etab = 0, sh_E = 0, sh_Epad = 0;
  int sh_elems = 16;              // elements per thread of its row transform (8 for 4096-point rows, see passB_shared_elems)
  int sh_nbl = 16;                // live spectral registers of the staged candidates (band rotation, passB_shared_nbl)
  double sh_sigma = 0.0;          // the sigma the taps belong to (band cut-off)
  double* d_wys = nullptr;        // [max_batch] candidate frequencies wy + rotation
  int* d_shifts = nullptr;        // [max_peaks] band rotation of every peak, in blocks of L / 16 bins
  Axis ax1s{};                    // its geometry of the y axis: periodic as ax1, or zero-padded to L >= n1 + E
  void* tw1s = nullptr;           // twiddles of ax1s.L when that differs from ax1.L
  int tw1s_L = 0;
  bool sh_ok = false;             // this sigma / axis can run it
  bool use_shared = true;         // GPA_NO_SHARED=1 keeps the per-candidate forward transforms
  int sh_epoch = 0, sh_built_epoch = -1, sh_built_K = 0, sh_built_B = 0;   // tables follow sigma and the staged k-list
  bool sh_built_reorder = true;   // ... and the NO_REORDER option they were built under
  double* d_ystep = nullptr;      // [max_peaks] 2 pi frac(ky_p + band rotation_p): phase step along y of the compensation phasor
  bool sh_one_kref = true;        // every candidate of a peak shares its reference vector (what d_ystep assumes)
  bool lk_raw = false;            // the last passB_select left the lock-ins raw (fused driver): the consumer applies d_ystep
  bool sh_built_ok = false;       // the tables of that key are complete and worth using
  bool sh_use = false;            // ... and the staged candidates form runs of >= 2 on an x-plane
  size_t sh_gb_bytes = 0, sh_psi_bytes = 0;
  std::vector<int> staged_planeof;
  void* Tbuf = nullptr;           // [tbuf_planes][n0][n1] complex: one plane per DISTINCT wx (x-plane), grown on demand
  int tbuf_planes = 0;
  SweepTables tb{};
  double* d_kl = nullptr;         // [max_batch][2]
  double* d_kr = nullptr;
  double* d_pw = nullptr;         // [max_batch] distinct wx values (x-planes)
  int last_planes = 0;
  std::vector<double> staged_kl, staged_kr, staged_kmat;   // what the device tables currently hold
  int* h_iters = nullptr;         // pinned: iteration counts of the last (possibly asynchronous) driver call
  int iters_stride = 1;           // 1: two-stream driver (h_iters[0], [1]); 4: paired workspace (flag words of 2 problems)
  int iters_off = 0;              // paired / batched: the word of a problem's flags that holds its count
  // images of up to 1024^2: both components of u in ONE set of launches (blockIdx.z) on one stream -- measured 8 %
  // (512^2) to 14 % (256^2) faster than two streams, whose kernels are too small to overlap; from 2048^2 on the two
  // streams win by 4 % (profiles/r02_image_stacks.txt, 'stack of 1')
  UnwrapWorkspace uwp{};
  bool have_uwp = false, use_pair = false;
  double* h_k = nullptr;          // pinned staging, 4 * max_batch doubles
  void* d_image = nullptr;        // staging for host-pointer entry points
  void* d_mean = nullptr;
  void* d_tile_mean = nullptr;    // whole-image mean of the tile path (gpa_tile_gradients_dev)
  double* d_tsum_part = nullptr;  // partial sums of gpa_tile_sums_dev, grown on demand
  size_t tsum_cap = 0;
  hipEvent_t ev_x = nullptr;      // stream-to-stream ordering (gpa_plan_wait_stream / gpa_stream_wait_plan)
  double tile_mean = std::numeric_limits<double>::quiet_NaN();
  void* d_sf = nullptr;           // [K][n0][n1] complex, grown on demand (a4 gradient path)
  size_t sf_bytes = 0;
  void* d_grad = nullptr;         // n0 x n1 x 2 staging for the host-pointer a4 call
  double* d_scratch = nullptr;    // 16384 doubles (mean partials, Gaussian weights at + 1024, Huber partials: 10 x 1024 + 10)
  void* d_aux0 = nullptr;         // n0 / n1 complex doubles: border-difference spectra (a9), Gaussian factors (f-4);
  void* d_aux1 = nullptr;         // NOT the sweep's compensation tables, which stay valid across those calls
  void* d_lockin = nullptr;       // [P<=max_peaks][n0][n1] complex (staging / fused driver)
  int32_t* d_kidx = nullptr;
  int max_peaks = 0;
  // reconstruct + unwrap workspace
  void* d_dudx = nullptr;         // 2 x n0 x (n1-1)
  void* d_dudy = nullptr;         // 2 x (n0-1) x n1
  void* d_wnorm = nullptr;        // n0 x n1
  void* d_u = nullptr;            // 2 x n0 x n1
  double* d_kmat = nullptr;       // [max_peaks][2]
  UnwrapWorkspace uw{};
  // batched driver (gpa_extract_displacement_field_batch_dev): one workspace for the 2 x images solves of a call
  UnwrapWorkspace uwb{};
  int uwb_images = 0;
  void* d_wnorm_b = nullptr;      // images x n0 x n1
  int* h_iters_b = nullptr;       // pinned: 4 ints per problem
  // sweep of a chunk of images in one set of launches: x-planes, lock-ins, means, mean scratch per image
  void *bT = nullptr, *bL = nullptr, *bMean = nullptr;
  double* bScratch = nullptr;
  size_t bT_bytes = 0, bL_bytes = 0;
  int b_chunk = 0;
  UnwrapWorkspace uw2{};          // second workspace + stream: the two components of u unwrap concurrently
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  DftAxis bx0{}, bx1{};           // the two axes of the plain 2-D DFT (a9, f-3, f-4: gpa_dft.h), built on first use
  DftWork dftw{};                 // scratch of its through-HBM engine, grown on demand
  void* d_pertab = nullptr;       // a9: per-axis factors 1 - e^(2 pi i j / n) and sin^2(pi j / n) (per_tables_create)
  // f-3: tables of the FFT form of gaussian_filter (gpa_gaussfft.h), two sigmas per axis (difference of Gaussians)
  struct GaussTab { double sigma = -1.0; int lg = 0; void* H = nullptr; void* tw = nullptr; int cap_lg = 0; unsigned stamp = 0; };
  GaussTab gft[2][2];
  unsigned gft_clock = 0;
  double* d_peakws = nullptr;     // f-3: min / max partials (2 x 2048), threshold, candidate counter
  void* d_peaksmooth = nullptr;   // f-3: the smoothed spectrum of the last gpa_find_peaks call, n0 x n1 reals (gpa_find_peaks_again)
  bool peaks_smooth_valid = false;
  WarpWs warp{};                  // scratch + taps of the Lawler-Fujita kernels (gpa_warp.hip), grown on first use
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool profiling = false;
  hipEvent_t stage_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float stage_ms[5] = {0, 0, 0, 0, 0};
  EnqueueWorker* worker = nullptr;   // second enqueueing thread of the fused driver
  bool use_worker = true, no_ksplit = false, no_compact = false;
  bool serial_unwrap = false;
  KernelProfiler* kprof = nullptr;   // per-kernel event pairs of the last profiled driver call
  std::string kprof_table;           // "name calls total_ms" lines of that call
  // downloads overlapped with the next call (gpa_download_async)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_dl_ready = nullptr, ev_dl_done[4] = {nullptr, nullptr, nullptr, nullptr};
};

template <class T>
static hipError_t upload_as(void* dst, const std::vector<double>& v, hipStream_t s) {
  std::vector<T> tmp(v.begin(), v.end());
  hipError_t e = hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

// ---- helpers shared by the entry-point files (defined in the file named) ----
void host_fft_pow2(std::vector<std::complex<double>>& a, bool inverse);
std::vector<double> gaussian_kspace(int n, double sigma);
std::vector<double> spatial_kernel(int n, const std::vector<double>& g);
int kernel_support(const std::vector<double>& h, double tol);
std::vector<double> spatial_taps(int n, const std::vector<double>& g, int mmax);
void build_filter_table(const Axis& ax, const std::vector<double>& g, const std::vector<double>& hsp,
                               std::vector<double>& out);
int ensure_filters(gpa_plan* p, double sigma);
int shared_prepare(gpa_plan* p, int P, int K);
int stage_kvectors(gpa_plan* p, const double* kl, const double* kr_per_b, int B, int* planes_out);
int ensure_tbuf(gpa_plan* p, int planes);
int ensure_sf(gpa_plan* p, size_t bytes);
int stage_kmat(gpa_plan* p, const double* kvecs, int P);
int run_passA(gpa_plan* p, const void* image, const void* mean, void* Tbuf, int Bx, int nimg = 1);
int passB_select(gpa_plan* p, int P, int K, void* lockin, int32_t* kidx, bool raw = false);
void collect_kernel_profile(gpa_plan* p);
int sweep_peaks_dev(gpa_plan* p, const void* image, const void* mean, const double* krefs, int P,
                           const double* klists, int K, double sigma, void* lockin, int32_t* kidx, bool raw = false);
int sweep_one_peak(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                          int mode, void* lockin, int32_t* kidx, const uint8_t* d_gate, void* d_psi);
int sweep_host(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
                      int grad_mode, const uint8_t* gate, void* lockin, int32_t* kidx, void* grad);
int extract_stage(gpa_plan* p, const double* kvecs, int P, const double* klists, int K, double sigma, int* Bx);
int extract_launch(gpa_plan* p, const void* image, int P, int K, int Bx, int mask_border, int kmax, void* u,
                          void* lk, int32_t* kidx, bool want_lockins);
int extract_enqueue(gpa_plan* p, const void* image, const double* kvecs, int P, const double* klists, int K,
                           double sigma, int mask_border, int kmax, void* u, void* lockins, int32_t* kidx);
int tile_gradients_impl(gpa_plan* p, const void* image, size_t image_pitch, int r0, int c0, bool mean_on_device,
                               double mean, const double* kvecs, int P, const double* klists, int K, double sigma,
                               int mask_border, int i0, int j0, int t0, int t1, void* dx, size_t dx_pitch, size_t dx_plane,
                               void* dy, size_t dy_pitch, size_t dy_plane, void* wn, size_t wn_pitch, size_t wn_plane);
int plan_event(gpa_plan* p);
int dft_axes(gpa_plan* p);
int per_dft_staged(gpa_plan* p, const void* d_image, void* abs_out);
int gaussian_filter_dev(gpa_plan* p, const void* in, void* tmp, void* out, double sigma, const void* minuend);
int gaussian_weights(double sigma, std::vector<double>& w);
bool solve3(const double* m /*uu uv u vv v 1*/, const double* b, double* x);
Axis make_axis(int n);
Axis compact_axis(const Axis& full, int E);
int upload_real_table(gpa_plan* p, void* dst, const std::vector<double>& v);
int dmalloc(gpa_plan* p, void** ptr, size_t bytes);
// an unwrap that did not start: hipErrorNotSupported is the workspace's "this shape has no kernels" (gpa_unwrap_tables.hip)
inline int unwrap_fail(hipError_t e) {
  if (e == hipErrorNotSupported)
    return fail(GPA_ERR_STATE, "unwrap: no kernels for this shape -- an image with an axis that is not a power of two needs both axes <= 8192 "
                               "(f32) / 4096 (f64); powers of two go to 16384 / 8192 per axis (INTEGRATION.md, size limits)");
  return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(e));
}
int plan_build(gpa_plan* p);
int upload_twiddles(gpa_plan* p, void* dst, int L);

// installs the plan's profiler on the calling thread for the lifetime of the object (while gpa_set_profiling is on)
// the same, only when `own` (an entry point that may also be called from inside another profiled one)
struct ProfInstallIf {
  bool own;
  ProfInstallIf(gpa_plan* p, bool own_) : own(own_ && p->profiling) {
    if (!own) return;
    if (!p->kprof) p->kprof = new KernelProfiler();
    p->kprof->n = 0;
    g_kprof = p->kprof;
  }
  ~ProfInstallIf() { if (own) g_kprof = nullptr; }
};
struct ProfInstall {
  explicit ProfInstall(gpa_plan* p) {
    if (!p->profiling) return;
    if (!p->kprof) p->kprof = new KernelProfiler();
    p->kprof->n = 0;
    g_kprof = p->kprof;
  }
  ~ProfInstall() { g_kprof = nullptr; }
};

