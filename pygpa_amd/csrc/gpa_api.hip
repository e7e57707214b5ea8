// C ABI of libgpa_hip.so (see include/gpa_hip.h): plans, host-built filter
// tables, and the drivers that chain the kernels on the plan's stream.
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

using namespace gpa;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
// ---- run-time options -------------------------------------------------------------------------
namespace gpa {
static const char* const kOptNames[OPT_COUNT] = {
    "PBS_FULLBAND", "USE_GRAPH", "SERIAL_UNWRAP", "NO_WORKER", "NO_KSPLIT", "NO_COMPACT", "NO_SHARED", "SHARED_A",
    "NO_PAIR", "PBS_E8", "TRI_SMALL", "TRI_Q", "NO_MR", "MR_FORCE_BLUESTEIN", "NO_ROWPQ", "COLSOLVE", "NO_LAT",
    "F32_EPS_FLOOR", "COLSTREAM_CHUNK", "NO_ROWHALF", "PAIR_MAXSIDE", "ROWHALF_MINLG", "NO_PQDCT", "NATIVE",
    "NATIVE_RATIO", "NATIVE_SHARED", "NO_REORDER", "NO_RAW", "NO_TILEFUSE"};
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

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(GPA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
  } while (0)

// ---------------------------------------------------------------------------
// host-side table construction (double precision)
// ---------------------------------------------------------------------------
static void host_fft_pow2(std::vector<std::complex<double>>& a, bool inverse) {
  const size_t n = a.size();
  int lg = 0;
  while ((size_t(1) << lg) < n) ++lg;
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
    if (r > i) std::swap(a[i], a[r]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = (inverse ? 2 : -2) * M_PI / (double)len;
    for (size_t s = 0; s < n; s += len)
      for (size_t j = 0; j < len / 2; ++j) {
        std::complex<double> w(cos(ang * (double)j), sin(ang * (double)j));
        auto u = a[s + j], v = a[s + j + len / 2] * w;
        a[s + j] = u + v;
        a[s + j + len / 2] = u - v;
      }
  }
}

// 1-D factor of scipy.ndimage.fourier_gaussian (called at
// geometric_phase_analysis.py:44/:75/:87, cuGPA.py:57): exp(-2 pi^2 sigma^2 f^2),
// f = fftfreq(n), flushed to 0 where the exponent exceeds 50 (SciPy does that per axis).
static std::vector<double> gaussian_kspace(int n, double sigma) {
  std::vector<double> g(n);
  for (int k = 0; k < n; ++k) {
    const int kk = k < (n + 1) / 2 ? k : k - n;   // fftfreq ordering
    const double f = (double)kk / (double)n;
    const double e = 2.0 * M_PI * M_PI * sigma * sigma * f * f;
    g[k] = e > 50.0 ? 0.0 : exp(-e);
  }
  return g;
}

// spatial kernel h[m] = (1/n) sum_k g[k] cos(2 pi k m / n), m = 0 .. n-1 (g is even): the taps of the circular filter
static std::vector<double> spatial_kernel(int n, const std::vector<double>& g) {
  // accumulated in long double (64-bit mantissa) so that the small taps are those of g as given, not summation noise
  std::vector<double> h((size_t)n);
  std::vector<long double> cs((size_t)n);
  const long double w = 2.0L * 3.14159265358979323846264338327950288L / (long double)n;
  for (int j = 0; j < n; ++j) cs[j] = cosl(w * (long double)j);
  for (int m = 0; m < n; ++m) {
    long double acc = 0;
    for (int k = 0; k < n; ++k)
      if (g[k] != 0.0) acc += (long double)g[k] * cs[(size_t)((long long)k * m % n)];
    h[m] = (double)(acc / (long double)n);
  }
  return h;
}

// smallest E such that the taps at circular distance > E from lag 0 sum (in magnitude) to less than tol times the
// sum of all taps: dropping them changes a filtered value by at most tol * max|input| * sum|h|, a guaranteed bound.
// The taps of a Gaussian filter fall to the rounding floor of its k-space samples (~1e-17 of the central tap each,
// the transform of the rounding errors of g) within ~9 sigma; beyond that they are noise the reference's own FFT
// does not resolve either.
static int kernel_support(const std::vector<double>& h, double tol) {
  const int n = (int)h.size();
  double total = 0;
  for (double v : h) total += fabs(v);
  double tail = 0;
  for (int m = n / 2; m >= 1; --m) {
    tail += fabs(h[m]) + (n - m != m ? fabs(h[n - m]) : 0.0);
    if (tail > tol * total) return m;
  }
  return 0;
}

// taps h[0 .. mmax] only (the shared-forward pass B needs the first few sigma of them, not all n): same sums as
// spatial_kernel
static std::vector<double> spatial_taps(int n, const std::vector<double>& g, int mmax) {
  std::vector<double> h((size_t)mmax + 1);
  std::vector<long double> cs((size_t)n);
  const long double w = 2.0L * 3.14159265358979323846264338327950288L / (long double)n;
  for (int j = 0; j < n; ++j) cs[j] = cosl(w * (long double)j);
  for (int m = 0; m <= mmax; ++m) {
    long double acc = 0;
    for (int k = 0; k < n; ++k)
      if (g[k] != 0.0) acc += (long double)g[k] * cs[(size_t)((long long)k * m % n)];
    h[m] = (double)(acc / (long double)n);
  }
  return h;
}

// Filter table of one axis in the spectral register layout [reg][thread]:
// periodic mode -> real g[k]/L; padded mode -> complex DFT_L(h~)/L with h~ the spatial kernel laid out at the lags
// the axis' extension covers (full: lags 0 .. n-1; compact: lags -extL .. extR around slot 0).
static void build_filter_table(const Axis& ax, const std::vector<double>& g, const std::vector<double>& hsp,
                               std::vector<double>& out) {
  const int L = ax.L, tpf = L / 16;
  if (!ax.padded) {
    out.assign((size_t)L, 0.0);
    for (int i = 0; i < 16; ++i)
      for (int t = 0; t < tpf; ++t) out[(size_t)i * tpf + t] = g[spec_index_rt(ax.lg, t, i)] / (double)L;
    return;
  }
  const int n = ax.n;
  std::vector<std::complex<double>> h((size_t)L, 0.0);
  if (ax.extR == 0) {
    for (int m = 0; m < n; ++m) h[m] = hsp[m];
  } else {
    for (int m = 0; m <= ax.extL; ++m) h[m] = hsp[m];                    // lags 0 .. E: samples to the left
    for (int m = 1; m <= ax.extR; ++m) h[(size_t)L - m] = hsp[n - m];     // lags -1 .. -E: samples to the right
  }
  host_fft_pow2(h, false);
  out.assign((size_t)2 * L, 0.0);
  for (int i = 0; i < 16; ++i)
    for (int t = 0; t < tpf; ++t) {
      auto v = h[spec_index_rt(ax.lg, t, i)] / (double)L;
      out[2 * ((size_t)i * tpf + t)] = v.real();
      out[2 * ((size_t)i * tpf + t) + 1] = v.imag();
    }
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
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

struct GraphKey {
  const void* image; void* u; void* lk; int32_t* kidx;
  int P, K, Bx, mask_border, kmax, epoch;
  int want_lockins, pad_;   // (compensated lock-ins asked for: the captured pass B / set-up launches differ)
};
struct GraphEntry {
  GraphKey key;
  hipGraph_t graph;
  hipGraphExec_t exec;
  bool failed;
};

struct gpa_plan {
  int device = 0, dtype = 0, n0 = 0, n1 = 0, max_batch = 0;
  Axis ax0{}, ax1{};              // the geometry in use (depends on sigma for non-power-of-two axes)
  Axis ax0_full{}, ax1_full{};    // the plan's largest geometry (L >= 2n - 1): what the tables are sized for
  // native mode of an axis (gpa_sweep_mr.hip): the length-n twiddles (uploaded once) and the filter table of the cached sigma
  void *natW0 = nullptr, *natW1 = nullptr, *natH0 = nullptr, *natH1 = nullptr;
  double natkey_cached = -2.0;    // the NO_NATIVE / NATIVE_RATIO options the cached geometry was chosen under
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
  bool lk_raw = false;            // the last passB_select left the lock-ins raw (fused driver): the consumer applies d_ystep
  bool sh_built_ok = false;       // the tables of that key are complete and worth using
  bool sh_use = false;            // ... and the staged candidates form runs of >= 2 on an x-plane
  size_t sh_gb_bytes = 0, sh_psi_bytes = 0;
  std::vector<int> staged_planeof;
  // shared-forward pass A: the same for the x axis (tables per x-plane)
  bool shA_ok = false;
  int shA_etab = 0, shA_E = 0, shA_Epad = 0;
  Axis ax0s{};
  void* tw0s = nullptr;
  int tw0s_L = 0;
  double* d_taps0 = nullptr;
  void *shA_gtab = nullptr, *shA_Gx = nullptr, *shA_psi = nullptr, *shA_sx = nullptr;
  size_t shA_gx_bytes = 0, shA_psi_bytes = 0;
  int shA_built_epoch = -1, shA_built_Bx = 0;
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
  double* d_tsum_part = nullptr;  // partial sums + ticket of gpa_tile_sums_dev, grown on demand
  size_t tsum_cap = 0;
  unsigned* d_ticket = nullptr;
  hipEvent_t ev_x = nullptr;      // stream-to-stream ordering (gpa_plan_wait_stream / gpa_stream_wait_plan)
  double tile_mean = std::numeric_limits<double>::quiet_NaN();
  void* d_sf = nullptr;           // [K][n0][n1] complex, grown on demand (a4 gradient path)
  size_t sf_bytes = 0;
  void* d_grad = nullptr;         // n0 x n1 x 2 staging for the host-pointer a4 call
  double* d_scratch = nullptr;    // 4096 doubles
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
  BlueAxis bx0{}, bx1{};          // Bluestein tables for gpa_per_dft, built on first use
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool profiling = false;
  hipEvent_t stage_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float stage_ms[5] = {0, 0, 0, 0, 0};
  EnqueueWorker* worker = nullptr;   // second enqueueing thread of the fused driver
  bool use_worker = true, no_ksplit = false, no_compact = false;
  std::vector<GraphEntry> graphs;    // captured fused-driver calls (extract_enqueue)
  bool use_graphs = true, serial_unwrap = false;
  int tbuf_epoch = 0;                // bumped when a buffer baked into the graphs is reallocated
  KernelProfiler* kprof = nullptr;   // per-kernel event pairs of the last profiled driver call
  std::string kprof_table;           // "name calls total_ms" lines of that call
  // downloads overlapped with the next call (gpa_download_async)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_dl_ready = nullptr, ev_dl_done[4] = {nullptr, nullptr, nullptr, nullptr};
};

static Axis make_axis(int n) {
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
  a.native = false;
  a.pl = MrPlan{};
  a.natW = a.natH = nullptr;
  return a;
}

// the axis geometry for a kernel whose taps vanish beyond E samples: the compact extension if it allows a shorter
// transform than the full one
static Axis compact_axis(const Axis& full, int E) {
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

template <class T>
static hipError_t upload_as(void* dst, const std::vector<double>& v, hipStream_t s) {
  std::vector<T> tmp(v.begin(), v.end());
  hipError_t e = hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

static int upload_real_table(gpa_plan* p, void* dst, const std::vector<double>& v) {
  if (p->dtype == GPA_F32) HIP_TRY(upload_as<float>(dst, v, p->stream));
  else HIP_TRY(upload_as<double>(dst, v, p->stream));
  return GPA_OK;
}

static int dmalloc(gpa_plan* p, void** ptr, size_t bytes) {
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(ptr, bytes);
  if (e != hipSuccess)
    return fail(GPA_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  p->ws_bytes += bytes;
  return GPA_OK;
}
#define TRY(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != GPA_OK) return _r; \
  } while (0)

static int plan_build(gpa_plan* p) {
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
  // (carrier base tables: one entry per thread of a transform -- L / 16, or up to 256 for an axis in native mode)
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
  TRY(dmalloc(p, (void**)&p->d_scratch, 4096 * sizeof(double)));
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
  {
    size_t before = 0;
    hipError_t e = unwrap_workspace_create(p->dtype, p->n0, p->n1, p->stream, &p->uw, &before);
    if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap workspace: ") + hipGetErrorString(e));
    p->ws_bytes += before;
  }
  return GPA_OK;
}

static void drop_graphs(gpa_plan* p);

static int upload_twiddles(gpa_plan* p, void* dst, int L) {
  std::vector<double> t((size_t)2 * L);
  for (int k = 0; k < L; ++k) {
    t[2 * k] = cos(-2.0 * M_PI * k / L);
    t[2 * k + 1] = sin(-2.0 * M_PI * k / L);
  }
  return upload_real_table(p, dst, t);
}

static int ensure_filters(gpa_plan* p, double sigma) {
  if (!(sigma > 0)) return fail(GPA_ERR_ARG, "sigma must be positive");
  // NATIVE=1 (opt-in, measured slower: Axis::native): an axis that is not a power of two long runs at its own length on
  // the mixed-radix engine when the padded transform would be at least NATIVE_RATIO (default 1.5) times as long
  const double natkey = !opt_set(OPT_NATIVE) ? -1.0 : (opt_set(OPT_NATIVE_RATIO) ? opt(OPT_NATIVE_RATIO).num : 1.5);
  if (sigma == p->sigma_cached && natkey == p->natkey_cached) return GPA_OK;
  HIP_TRY(hipStreamSynchronize(p->stream));   // the tables may still be read by an earlier asynchronous call
  p->sigma_cached = -1.0;   // a failure below must not leave half-switched tables behind a matching sigma
  for (int axis = 0; axis < 2; ++axis) {
    const Axis& full = axis == 0 ? p->ax0_full : p->ax1_full;
    Axis& cur = axis == 0 ? p->ax0 : p->ax1;
    std::vector<double> g = gaussian_kspace(full.n, sigma), hsp, table;
    Axis want = full;
    if (full.padded) {
      hsp = spatial_kernel(full.n, g);
      if (!p->no_compact) want = compact_axis(full, kernel_support(hsp, p->dtype == 0 ? 1e-9 : 1e-14));
    }
    {
      MrPlan pl{};
      want.native = natkey > 0 && full.padded && full.n >= 48 && mr_make_plan(full.n, &pl) && pl.T <= 256 &&
                    (double)want.L >= natkey * (double)full.n;
      if (want.native) {
        want.pl = pl;
        void** Wd = axis == 0 ? &p->natW0 : &p->natW1;
        void** Hd = axis == 0 ? &p->natH0 : &p->natH1;
        if (!*Wd) {
          TRY(dmalloc(p, Wd, (size_t)mr_lds_elems(full.n) * p->csz));
          TRY(dmalloc(p, Hd, (size_t)full.n * p->rsz));
          std::vector<double> t((size_t)2 * mr_lds_elems(full.n), 0.0);   // entry k at mr_pad(k), see mr_store()
          for (int k = 0; k < full.n; ++k) {
            t[2 * (size_t)mr_pad(k)] = cos(-2.0 * M_PI * k / full.n);
            t[2 * (size_t)mr_pad(k) + 1] = sin(-2.0 * M_PI * k / full.n);
          }
          TRY(upload_real_table(p, *Wd, t));
        }
        std::vector<double> hn((size_t)full.n);
        for (int k = 0; k < full.n; ++k) hn[k] = g[k] / (double)full.n;
        TRY(upload_real_table(p, *Hd, hn));
        want.natW = *Wd;
        want.natH = *Hd;
      }
    }
    if (want.lg != cur.lg || want.extR != cur.extR || want.extL != cur.extL || want.native != cur.native) {
      // another transform length for this sigma: twiddles of that length, and the carrier tables (laid out per
      // L / 16 threads) have to be staged again
      TRY(upload_twiddles(p, axis == 0 ? p->tw0 : p->tw1, want.L));
      p->staged_kl.clear();
      p->staged_kr.clear();
      drop_graphs(p);
      cur = want;
    }
    build_filter_table(cur, g, hsp, table);
    TRY(upload_real_table(p, axis == 0 ? p->Hx : p->Hy, table));
    if (axis == 0) {
      // shared-forward pass A: taps, support and (for lengths that are not powers of two) the zero-padded geometry
      // of the x axis, exactly as for the y axis below
      p->shA_ok = false;
      const int n = cur.n;
      int mmax = (int)ceil(10.0 * sigma) + 16;
      if (p->use_shared && mmax < n / 2 && mmax <= 1024) {
        std::vector<double> taps = spatial_taps(n, g, mmax);
        double total = fabs(taps[0]), tail = 0;
        for (int m = 1; m <= mmax; ++m) total += 2 * fabs(taps[m]);
        int E = 1;
        const double tol = p->dtype == 0 ? 1e-9 : 1e-14;
        for (int m = mmax; m >= 1; --m) {
          tail += 2 * fabs(taps[m]);
          if (tail > tol * total) { E = m; break; }
        }
        Axis sa = cur;
        if (cur.padded) {
          sa.lg = 6;
          while ((1 << sa.lg) < n + E) ++sa.lg;
          sa.L = 1 << sa.lg;
          sa.extL = sa.extR = 0;
        }
        if (passA_shared_supports(p->dtype, sa, E)) {
          const int Epad = (E + 15) & ~15;
          if (sa.L != cur.L && p->tw0s_L != sa.L) {
            if (!p->tw0s) TRY(dmalloc(p, &p->tw0s, (size_t)4096 * p->csz));
            TRY(upload_twiddles(p, p->tw0s, sa.L));
            p->tw0s_L = sa.L;
          }
          p->ax0s = sa;
          if (!p->d_taps0) TRY(dmalloc(p, (void**)&p->d_taps0, 1025 * sizeof(double)));
          if (!p->shA_gtab) TRY(dmalloc(p, &p->shA_gtab, (2 * 256 + 16) * p->rsz));
          HIP_TRY(hipMemcpyAsync(p->d_taps0, taps.data(), ((size_t)mmax + 1) * sizeof(double), hipMemcpyHostToDevice, p->stream));
          HIP_TRY(hipStreamSynchronize(p->stream));
          std::vector<double> gt((size_t)2 * Epad + 16, 0.0);
          for (int m = 1; m <= E; ++m) gt[m] = taps[m];
          TRY(upload_real_table(p, p->shA_gtab, gt));
          p->shA_etab = mmax;
          p->shA_E = E;
          p->shA_Epad = Epad;
          p->shA_ok = true;
        }
      }
    }
    if (axis == 1) {
      // shared-forward pass B: the taps of this axis' filter out to where they are rounding noise, the support E
      // beyond which they are dropped from the end fix (the same criterion as the compact extension above)
      p->sh_ok = false;
      ++p->sh_epoch;
      const int n = cur.n;
      int mmax = (int)ceil(10.0 * sigma) + 16;
      if (p->use_shared && mmax < n / 2 && mmax <= 1024) {
        std::vector<double> taps = spatial_taps(n, g, mmax);
        double total = fabs(taps[0]), tail = 0;
        for (int m = 1; m <= mmax; ++m) total += 2 * fabs(taps[m]);
        int E = 1;
        const double tol = p->dtype == 0 ? 1e-9 : 1e-14;
        for (int m = mmax; m >= 1; --m) {
          tail += 2 * fabs(taps[m]);
          if (tail > tol * total) { E = m; break; }
        }
        // a row that is not a power of two long is zero-padded to the next power of two >= n + E: the shared kernel
        // needs no periodic extension (its end fix supplies every wrapped pair), only room for the filter's reach
        Axis sa = cur;
        if (cur.padded) {
          sa.lg = 6;
          while ((1 << sa.lg) < n + E) ++sa.lg;
          sa.L = 1 << sa.lg;
          sa.extL = sa.extR = 0;
        }
        if (passB_shared_supports(p->dtype, sa, E)) {
          const int Epad = (E + 15) & ~15;
          if (sa.L != cur.L && p->tw1s_L != sa.L) {
            if (!p->tw1s) TRY(dmalloc(p, &p->tw1s, (size_t)8192 * p->csz));   // (the longest row transform of the shared kernel)
            TRY(upload_twiddles(p, p->tw1s, sa.L));
            p->tw1s_L = sa.L;
          }
          p->ax1s = sa;
          if (!p->d_taps) TRY(dmalloc(p, (void**)&p->d_taps, 1025 * sizeof(double)));
          if (!p->sh.gtab) TRY(dmalloc(p, &p->sh.gtab, (2 * 256 + 16) * p->rsz));
          HIP_TRY(hipMemcpyAsync(p->d_taps, taps.data(), ((size_t)mmax + 1) * sizeof(double), hipMemcpyHostToDevice, p->stream));
          HIP_TRY(hipStreamSynchronize(p->stream));
          std::vector<double> gt((size_t)2 * Epad + 16, 0.0);
          for (int m = 1; m <= E; ++m) gt[m] = taps[m];
          TRY(upload_real_table(p, p->sh.gtab, gt));
          p->sh_etab = mmax;
          p->sh_E = E;
          p->sh_Epad = Epad;
          p->sh_elems = passB_shared_elems(p->dtype, sa);
          p->sh_sigma = sigma;
          p->sh_ok = true;
        }
      }
    }
  }
  p->sigma_cached = sigma;
  p->natkey_cached = natkey;
  return GPA_OK;
}

// candidate tables of the shared-forward pass B for the staged k-list (P peaks of K candidates): rebuilt when sigma
// or the list changed.  Leaves p->sh_use = whether pass B should take that kernel for this (P, K).
static int shared_prepare(gpa_plan* p, int P, int K) {
  p->sh_use = false;
  const int B = P * K;
  if (!p->sh_ok || !p->use_shared || K < 2 || (int)p->staged_planeof.size() < B) return GPA_OK;
  const bool reorder = !opt_set(OPT_NO_REORDER);
  if (p->sh_built_epoch == p->sh_epoch && p->sh_built_K == K && p->sh_built_B == B && p->sh_built_reorder == reorder) {
    p->sh_use = p->sh_built_ok;
    return GPA_OK;
  }
  p->sh_built_reorder = reorder;
  // (the cache key is committed only when the tables are complete: a failed allocation below must not leave a key
  //  that sends the next call to the kernel with freed tables -- ADVICE r03)
  p->sh_built_epoch = -1;
  p->sh_built_ok = false;
  // Visiting order of every peak's candidates.  The selection rule is "strictly larger |sf| replaces, in LIST order"
  // (geometric_phase_analysis.py:679-684) = the first maximum of the list wins.  The kernel stores a winner the moment
  // it wins, so the order in which it visits the candidates sets how often a pixel is rewritten: in list order the
  // amplitude climbs towards the grid's centre (4.3 stores per pixel at configs[2]); visiting the candidates nearest
  // the reference vector first, most later candidates win nowhere in a wavefront and their stores are skipped.  The
  // candidates of one x-plane stay together (they share the forward transform; a list that interleaves the planes gains
  // its runs here), planes ordered by their nearest candidate, candidates within a plane by distance, ties by list
  // position (a stable order: duplicates of a k-vector keep the list's order, so "first maximum" still holds for
  // them; the reported kidx is the original list position).  Two DIFFERENT candidates whose amplitudes agree bit for
  // bit at a pixel may now resolve the other way -- the amplitude ties the tests already allow for.
  std::vector<int> order((size_t)B);
  for (int pp = 0; pp < P; ++pp) {
    std::vector<double> d2((size_t)K), pmin;
    std::vector<int> pfirst;
    int nplanes = 0;
    for (int k = 0; k < K; ++k) nplanes = std::max(nplanes, p->staged_planeof[pp * K + k] + 1);
    pmin.assign((size_t)nplanes, 1e300);
    pfirst.assign((size_t)nplanes, K);
    for (int k = 0; k < K; ++k) {
      const size_t b = (size_t)pp * K + k;
      const double ex = p->staged_kl[2 * b] - p->staged_kr[2 * b], ey = p->staged_kl[2 * b + 1] - p->staged_kr[2 * b + 1];
      d2[k] = ex * ex + ey * ey;
      const int pl = p->staged_planeof[b];
      pmin[pl] = std::min(pmin[pl], d2[k]);
      pfirst[pl] = std::min(pfirst[pl], k);
    }
    std::vector<int> idx((size_t)K);
    for (int k = 0; k < K; ++k) idx[k] = k;
    if (reorder)
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b2) {
        const int pa = p->staged_planeof[pp * K + a], pb = p->staged_planeof[pp * K + b2];
        if (pa != pb) return pmin[pa] != pmin[pb] ? pmin[pa] < pmin[pb] : pfirst[pa] < pfirst[pb];
        return d2[a] < d2[b2];
      });
    for (int k = 0; k < K; ++k) order[(size_t)pp * K + k] = pp * K + idx[k];
  }
  // runs of candidates on one x-plane, in visiting order; chunks of <= NC candidates per matrix pass
  const int NC = p->dtype == 0 ? 4 : 2;
  std::vector<int> desc((size_t)B, 0);
  int runs = 0;
  for (int pp = 0; pp < P; ++pp) {
    int k = 0, chunk = 0;
    while (k < K) {
      int e = k + 1;
      while (e < K && p->staged_planeof[order[pp * K + e]] == p->staged_planeof[order[pp * K + k]]) ++e;
      ++runs;
      for (int j = k; j < e; ++j) {
        const int r = j - k, slot = r % NC;
        int d = slot << 2;
        if (r == 0) d |= 1;
        if (slot == 0) { d |= 2 | (std::min(NC, e - j) << 4); ++chunk; }
        d |= (chunk & 1) << 7;   // parity of the chunk (double-buffered staging of its phasors)
        desc[(size_t)pp * K + j] = d;
      }
      k = e;
    }
  }
  if (2 * runs > B) {   // fewer than two candidates per forward transform on average: nothing to share
    p->sh_built_epoch = p->sh_epoch;
    p->sh_built_K = K;
    p->sh_built_B = B;
    return GPA_OK;
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  // Band rotation.  The shifted Gaussian of a candidate is negligible (below 1e-9 of its peak in f32, 1e-17 in f64)
  // outside |f + wy| < fc; over the candidates of a peak the live band is (-wy_max - fc, -wy_min + fc).  Rotating the
  // row by exp(-2 pi i s y / 16) and the candidates to wy + s / 16 moves that band down by s blocks of L / 16 bins:
  // s = the block the band starts in, so that it occupies blocks 0 .. need-1 -- the spectral registers the kernel keeps.
  const int EEs = p->sh_elems;
  const double fc = sqrt(log(p->dtype == 0 ? 1e9 : 1e17) / (2.0 * M_PI * M_PI * p->sh_sigma * p->sh_sigma));
  std::vector<int> shifts((size_t)P, 0);
  std::vector<double> wys((size_t)B);
  int need = 1;
  for (int pp = 0; pp < P; ++pp) {
    double wmin = p->staged_kl[2 * ((size_t)pp * K) + 1], wmax = wmin;
    for (int k = 1; k < K; ++k) {
      const double w = p->staged_kl[2 * ((size_t)pp * K + k) + 1];
      wmin = std::min(wmin, w);
      wmax = std::max(wmax, w);
    }
    const double lo = -wmax - fc, width = (wmax - wmin) + 2 * fc;
    const double flo = (lo - floor(lo)) * EEs;             // start of the band in blocks, in [0, 16)
    const int sft = (int)floor(flo) % EEs;
    const int blocks = width >= 1.0 ? EEs : (int)ceil((flo - floor(flo)) + width * EEs + 1e-9);
    shifts[pp] = blocks >= EEs ? 0 : sft;
    need = std::max(need, std::min(blocks, EEs));
    for (int k = 0; k < K; ++k)
      wys[(size_t)pp * K + k] = p->staged_kl[2 * (size_t)order[(size_t)pp * K + k] + 1] + (double)shifts[pp] / EEs;
  }
  p->sh_nbl = opt_set(OPT_PBS_FULLBAND) ? EEs : passB_shared_nbl(p->dtype, need);
  if (p->sh_nbl >= EEs) {   // nothing to gain: no rotation
    for (int pp = 0; pp < P; ++pp) shifts[pp] = 0;
    for (int bq = 0; bq < B; ++bq) wys[bq] = p->staged_kl[2 * (size_t)order[bq] + 1];
    p->sh_nbl = EEs;
  }
  const size_t gb = (size_t)B * p->ax1s.L * p->rsz, ps = (size_t)B * p->sh_Epad * p->csz;
  if (gb > p->sh_gb_bytes) {
    if (p->sh.Gb) { (void)hipFree(p->sh.Gb); p->ws_bytes -= p->sh_gb_bytes; p->sh.Gb = nullptr; p->sh_gb_bytes = 0; }
    TRY(dmalloc(p, &p->sh.Gb, gb));
    p->sh_gb_bytes = gb;
  }
  if (ps > p->sh_psi_bytes) {
    if (p->sh.psi) { (void)hipFree(p->sh.psi); (void)hipFree(p->sh.pre); p->ws_bytes -= 2 * p->sh_psi_bytes; p->sh.psi = p->sh.pre = nullptr; p->sh_psi_bytes = 0; }
    TRY(dmalloc(p, &p->sh.psi, ps));
    TRY(dmalloc(p, &p->sh.pre, ps));
    p->sh_psi_bytes = ps;
  }
  if (!p->sh.rot16) TRY(dmalloc(p, &p->sh.rot16, (size_t)p->max_peaks * 16 * p->csz));
  if (!p->d_wys) TRY(dmalloc(p, (void**)&p->d_wys, (size_t)p->max_batch * sizeof(double)));
  if (!p->d_shifts) TRY(dmalloc(p, (void**)&p->d_shifts, (size_t)p->max_peaks * sizeof(int)));
  HIP_TRY(hipMemcpyAsync(p->d_wys, wys.data(), (size_t)B * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_shifts, shifts.data(), (size_t)P * sizeof(int), hipMemcpyHostToDevice, p->stream));
  {
    // raw mode of the kernel: what the winners lack is dyc[p][y] = exp(2 pi i (ky_p + shift_p / EE) y), i.e. this
    // phase step per column (reduced to (-pi, pi] in double)
    std::vector<double> ys((size_t)P);
    for (int pp = 0; pp < P; ++pp) {
      const double c = p->staged_kr[2 * ((size_t)pp * K) + 1] + (double)shifts[pp] / EEs;
      ys[pp] = 2.0 * M_PI * (c - rint(c));
    }
    if (!p->d_ystep) TRY(dmalloc(p, (void**)&p->d_ystep, (size_t)p->max_peaks * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(p->d_ystep, ys.data(), (size_t)P * sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));   // `ys` is a local
  }
  if (!p->sh.dyc) TRY(dmalloc(p, &p->sh.dyc, (size_t)p->max_peaks * p->n1 * p->csz));
  if (!p->sh.desc) TRY(dmalloc(p, (void**)&p->sh.desc, (size_t)p->max_batch * sizeof(int)));
  HIP_TRY(hipMemcpyAsync(p->sh.desc, desc.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, p->stream));
  if (!p->sh.order) TRY(dmalloc(p, (void**)&p->sh.order, (size_t)p->max_batch * sizeof(int)));
  HIP_TRY(hipMemcpyAsync(p->sh.order, order.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(launch_shared_tables(p->dtype, p->ax1s, p->d_wys, p->d_kr, p->d_shifts, p->d_taps, p->sh_etab, p->sh_E, p->sh_Epad, B, K,
                               p->sh_nbl, p->sh, p->stream, p->sh_elems));
  HIP_TRY(hipStreamSynchronize(p->stream));   // `desc`, `order` are locals
  p->sh_built_epoch = p->sh_epoch;
  p->sh_built_K = K;
  p->sh_built_B = B;
  p->sh_built_ok = true;
  p->sh_use = true;
  return GPA_OK;
}

// copy the (candidate, reference) k-vector lists to the device, map the candidates onto
// x-planes (one per distinct wx, see SweepTables) and build the carrier tables.
// Returns the number of x-planes in *planes_out.
static int stage_kvectors(gpa_plan* p, const double* kl, const double* kr_per_b, int B, int* planes_out) {
  // same candidates as the previous call (a sequence of images analysed with one k-list):
  // the carrier tables on the device are still valid, nothing to copy and nothing to wait for
  if ((int)p->staged_kl.size() == 2 * B && memcmp(p->staged_kl.data(), kl, (size_t)B * 2 * sizeof(double)) == 0 &&
      memcmp(p->staged_kr.data(), kr_per_b, (size_t)B * 2 * sizeof(double)) == 0) {
    *planes_out = p->last_planes;
    return GPA_OK;
  }
  // the pinned staging buffer may still feed copies of an earlier asynchronous call
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->staged_kl.assign(kl, kl + 2 * (size_t)B);
  p->staged_kr.assign(kr_per_b, kr_per_b + 2 * (size_t)B);
  double* h_kl = p->h_k;
  double* h_kr = p->h_k + 2 * (size_t)B;
  double* h_pw = p->h_k + 4 * (size_t)B;
  int* h_po = reinterpret_cast<int*>(p->h_k + 5 * (size_t)B);
  memcpy(h_kl, kl, (size_t)B * 2 * sizeof(double));
  memcpy(h_kr, kr_per_b, (size_t)B * 2 * sizeof(double));
  int Bx = 0;
  for (int b = 0; b < B; ++b) {
    int found = -1;
    for (int q = 0; q < Bx; ++q)
      if (memcmp(&h_pw[q], &kl[2 * b], sizeof(double)) == 0) { found = q; break; }
    if (found < 0) { h_pw[Bx] = kl[2 * b]; found = Bx++; }
    h_po[b] = found;
  }
  p->staged_planeof.assign(h_po, h_po + B);
  ++p->sh_epoch;
  HIP_TRY(hipMemcpyAsync(p->d_kl, h_kl, (size_t)B * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_kr, h_kr, (size_t)B * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_pw, h_pw, (size_t)Bx * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->tb.planeof, h_po, (size_t)B * sizeof(int), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(launch_tables(p->dtype, p->ax0, p->ax1, p->d_kl, p->d_kr, B, p->d_pw, Bx, p->tb, p->stream));
  // h_k is reused by the next call: wait for the copies
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->last_planes = Bx;
  *planes_out = Bx;
  return GPA_OK;
}

// room for `planes` x-planes in Tbuf
static int ensure_tbuf(gpa_plan* p, int planes) {
  if (planes <= p->tbuf_planes) return GPA_OK;
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipFree(p->Tbuf));
  p->ws_bytes -= (size_t)p->tbuf_planes * npx * p->csz;
  p->Tbuf = nullptr;
  p->tbuf_planes = 0;
  TRY(dmalloc(p, &p->Tbuf, (size_t)planes * npx * p->csz));
  p->tbuf_planes = planes;
  ++p->tbuf_epoch;   // captured graphs hold the old pointer
  return GPA_OK;
}

// scratch of at least `bytes` in p->d_sf (per-candidate phases of the a4 path, gate table of wfr4, batched lock-ins)
static int ensure_sf(gpa_plan* p, size_t bytes) {
  if (p->sf_bytes >= bytes) return GPA_OK;
  if (p->d_sf) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipFree(p->d_sf));
    p->ws_bytes -= p->sf_bytes;
    p->d_sf = nullptr;
    p->sf_bytes = 0;
  }
  TRY(dmalloc(p, &p->d_sf, bytes));
  p->sf_bytes = bytes;
  return GPA_OK;
}

// 2 pi kvecs for the per-pixel solves, re-staged only when the peaks change
static int stage_kmat(gpa_plan* p, const double* kvecs, int P) {
  std::vector<double> km((size_t)2 * P);
  for (int i = 0; i < 2 * P; ++i) km[i] = 2.0 * M_PI * kvecs[i];
  if (km == p->staged_kmat) return GPA_OK;
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->staged_kmat = km;
  double* h = p->h_k + 6 * (size_t)p->max_batch;
  memcpy(h, km.data(), km.size() * sizeof(double));
  HIP_TRY(hipMemcpyAsync(p->d_kmat, h, km.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  return GPA_OK;
}

static int run_passA(gpa_plan* p, const void* image, const void* mean, void* Tbuf, int Bx, int nimg = 1);

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
  p->use_graphs = opt_set(OPT_USE_GRAPH);
  p->serial_unwrap = opt_set(OPT_SERIAL_UNWRAP);
  p->use_worker = !opt_set(OPT_NO_WORKER);
  p->no_ksplit = opt_set(OPT_NO_KSPLIT);
  p->no_compact = opt_set(OPT_NO_COMPACT);
  p->use_shared = !opt_set(OPT_NO_SHARED);
  const int maxlg = dtype == GPA_F32 ? 14 : 13;
  if (p->ax0.lg > maxlg || p->ax1.lg > maxlg) {
    fail(GPA_ERR_ARG, "gpa_plan_create: axis too long for an LDS-resident transform "
                      "(f32: 16384 pow2 / 8192 other; f64: 8192 pow2 / 4096 other)");
    delete p;
    return nullptr;
  }
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
  for (auto& g : p->graphs) {
    if (g.exec) hipGraphExecDestroy(g.exec);
    if (g.graph) hipGraphDestroy(g.graph);
  }
  p->graphs.clear();
  void* bufs[] = {p->tw0, p->tw1, p->Hx, p->Hy, p->Tbuf, p->tb.cxb, p->tb.sx, p->tb.wxw, p->tb.wxr, p->tb.cyb, p->tb.sy, p->tb.wyw, p->tb.wyr, p->tb.planeof, p->d_pw,
                  p->tb.dx, p->tb.dy, p->d_kl, p->d_kr, p->d_image, p->d_mean, p->d_tile_mean, p->d_scratch,
                  p->d_lockin, p->d_kidx, p->d_dudx, p->d_dudy, p->d_wnorm, p->d_u, p->d_kmat, p->d_sf, p->d_grad, p->d_aux0, p->d_aux1,
                  p->sh.Gb, p->sh.psi, p->sh.dyc, p->sh.gtab, p->sh.desc, p->sh.order, p->d_taps, p->tw1s,
                  p->tw0s, p->d_taps0, p->shA_gtab, p->shA_Gx, p->shA_psi, p->shA_sx, p->sh.pre, p->sh.rot16, p->d_wys, p->d_shifts, p->d_ystep};
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
  if (p->d_ticket) (void)hipFree(p->d_ticket);
  blue_axis_destroy(&p->bx0);
  blue_axis_destroy(&p->bx1);
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
int gpa_plan_axis_native(const gpa_plan* p, int axis) {
  if (!p) return 0;
  return (axis == 0 ? p->ax0.native : p->ax1.native) ? 1 : 0;
}

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

// per-plane tables of the shared-forward pass A for the staged x-planes: rebuilt when sigma or the k-list changed
static int sharedA_prepare(gpa_plan* p, int Bx, bool* use) {
  *use = false;
  // Opt-in (GPA_SHARED_A=1).  Measured at 4096^2, 3 x 4 planes, f32 (profiles/r03_passA_shared.txt): 1.0 - 1.2 ms against
  // the per-plane kernel's 0.815 ms although its transforms alone take 0.43 ms against 0.51 ms: pass A is bound by
  // the drain of its 32-byte-segment stores (~0.6 - 0.75 ms for 1.6 GB), which the per-plane kernel hides behind the
  // forward transform of the NEXT plane (it needs nothing from memory), while the shared kernel's next plane starts
  // with table loads that queue behind those stores.  Kept for the record and for the tests that pin its parity.
  if (!p->shA_ok || !p->use_shared || Bx < 2 || !opt_set(OPT_SHARED_A)) return GPA_OK;
  if (p->shA_built_epoch == p->sh_epoch && p->shA_built_Bx == Bx) { *use = true; return GPA_OK; }
  p->shA_built_epoch = -1;   // committed again only when the tables are complete
  HIP_TRY(hipStreamSynchronize(p->stream));
  const size_t gx = (size_t)Bx * p->ax0s.L * p->rsz, ps = (size_t)Bx * p->shA_Epad * p->csz;
  if (gx > p->shA_gx_bytes) {
    if (p->shA_Gx) { (void)hipFree(p->shA_Gx); p->ws_bytes -= p->shA_gx_bytes; p->shA_Gx = nullptr; p->shA_gx_bytes = 0; }
    TRY(dmalloc(p, &p->shA_Gx, gx));
    p->shA_gx_bytes = gx;
  }
  if (ps > p->shA_psi_bytes) {
    if (p->shA_psi) { (void)hipFree(p->shA_psi); p->ws_bytes -= p->shA_psi_bytes; p->shA_psi = nullptr; p->shA_psi_bytes = 0; }
    TRY(dmalloc(p, &p->shA_psi, ps));
    p->shA_psi_bytes = ps;
  }
  if (!p->shA_sx) TRY(dmalloc(p, &p->shA_sx, (size_t)p->max_batch * 16 * p->csz));
  HIP_TRY(launch_sharedA_tables(p->dtype, p->ax0s, p->d_pw, p->d_taps0, p->shA_etab, p->shA_E, p->shA_Epad, Bx, p->shA_Gx,
                                p->shA_psi, p->shA_sx, p->stream));
  p->shA_built_epoch = p->sh_epoch;
  p->shA_built_Bx = Bx;
  *use = true;
  return GPA_OK;
}

// pass A over the staged x-planes: one forward transform per column for all planes where the axis allows it
// (gpa_passb_shared.h), the per-plane forward transforms otherwise
static int run_passA(gpa_plan* p, const void* image, const void* mean, void* Tbuf, int Bx, int nimg) {
  bool shared = false;
  TRY(sharedA_prepare(p, Bx, &shared));
  if (shared) {
    SweepTables tb = p->tb;
    tb.sx = p->shA_sx;     // stride factors of the kernel's own transform length
    HIP_TRY(launch_passA_shared(p->dtype, p->ax0s, p->n1, image, mean, tb, p->ax0.L / 16, p->shA_Gx, p->shA_psi, p->shA_gtab,
                                p->ax0s.L == p->ax0.L ? p->tw0 : p->tw0s, p->shA_E, p->shA_Epad, Tbuf, Bx, p->stream, nimg));
  } else {
    HIP_TRY(launch_passA(p->dtype, p->ax0, p->n1, image, mean, p->tb, p->Hx, p->tw0, Tbuf, Bx, p->stream, nimg));
  }
  return GPA_OK;
}

// pass B with selection.  A small image has few rows to spread over the 256 CUs and runs its K candidates one after
// the other in each workgroup (512^2, K = 16: 82 us, a quarter of the image's time): there the candidates are split over
// up to 4 workgroups per row and merged (launch_passB_split) -- same winners, same values.
// raw: the caller's consumer is reconstruct_setup (which takes the compensation's phase step): where the shared kernel
// runs it then skips its second visit of the winner rows; p->lk_raw says whether it did
static int passB_select(gpa_plan* p, int P, int K, void* lockin, int32_t* kidx, bool raw = false) {
  p->lk_raw = false;
  const int rows_wg = (p->n0 + 7) / 8 * P;           // workgroups of the unsplit launch (at least: NF <= 8 rows each)
  int ksplit = 1;
  // (only while the unsplit launch has fewer workgroups than the chip has CUs: at 1024^2, 384 workgroups, the split
  //  measured slower -- 153 -> 188 us -- because the merge pass and the partial slabs cost more than they save)
  if (p->ax1.lg <= 10 && K >= 4 && !p->no_ksplit && rows_wg <= 256)
    while (ksplit < 4 && ksplit * 2 <= K && rows_wg * ksplit < 1024) ksplit *= 2;
  if (ksplit == 1) {
    TRY(shared_prepare(p, P, K));
    // (a row in native mode: the per-candidate kernel at length n rather than the shared-forward kernel on the
    //  zero-padded power of two, unless NATIVE_SHARED asks for the latter)
    if (p->sh_use && p->ax1.native && !opt_set(OPT_NATIVE_SHARED)) p->sh_use = false;
    if (p->sh_use) {
      p->lk_raw = raw && !opt_set(OPT_NO_RAW);
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
static void collect_kernel_profile(gpa_plan* p) {
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
// installs the plan's profiler on the calling thread for the lifetime of the object (while gpa_set_profiling is on)
struct ProfInstall {
  explicit ProfInstall(gpa_plan* p) {
    if (!p->profiling) return;
    if (!p->kprof) p->kprof = new KernelProfiler();
    p->kprof->n = 0;
    g_kprof = p->kprof;
  }
  ~ProfInstall() { g_kprof = nullptr; }
};

// ---- a3 ----------------------------------------------------------------------
static int sweep_peaks_dev(gpa_plan* p, const void* image, const void* mean, const double* krefs, int P,
                           const double* klists, int K, double sigma, void* lockin, int32_t* kidx, bool raw = false) {
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
static int sweep_one_peak(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
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
  // a4: the winner is selected in registers by pass B as in the plain sweep; what the gradient stencil needs from
  // the OTHER candidates is only the phase of the winner's candidate at the four neighbours, so pass B also writes
  // one real per pixel and candidate (K reals instead of the K complex lock-ins of the first build)
  const size_t npx = (size_t)p->n0 * p->n1;
  TRY(ensure_sf(p, (size_t)K * npx * p->rsz));
  int32_t* ki = kidx ? kidx : p->d_kidx;
  TRY(sweep_one_peak(p, image, kref, klist, K, sigma, 3, lockin, ki, nullptr, p->d_sf));
  HIP_TRY(launch_phasegrad(p->dtype, p->d_sf, K, ki, p->n0, p->n1, p->d_kl, p->d_kr, grad_mode, grad, p->stream));
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

static int sweep_host(gpa_plan* p, const void* image, const double* kref, const double* klist, int K, double sigma,
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
  HIP_TRY(hipSetDevice(p->device));
  int iters = 0;
  ProfInstall prof(p);   // (gpa_set_profiling: per-kernel times of this solve through gpa_last_kernel_profile)
  hipError_t e = unwrap_run(&p->uw, dx, dy, weight, false, kmax, eps, compat != 0, phi, &iters, p->stream);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(e));
  if (p->profiling) collect_kernel_profile(p);
  if (iters_out) *iters_out = iters;
  return GPA_OK;
}

int gpa_unwrap_prediff_enqueue_dev(gpa_plan* p, const void* dx, const void* dy, const void* weight, int kmax, double eps,
                                   int compat, void* phi) {
  if (!p || !dx || !dy || !phi) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff_enqueue: null argument");
  if (kmax < 1) return fail(GPA_ERR_ARG, "gpa_unwrap_prediff_enqueue: kmax must be >= 1");
  HIP_TRY(hipSetDevice(p->device));
  hipError_t e = unwrap_enqueue(&p->uw, dx, dy, weight, false, kmax, eps, compat != 0, phi, p->stream);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(e));
  return GPA_OK;
}

int gpa_unwrap_finish(gpa_plan* p, int* iters_out) {
  if (!p) return fail(GPA_ERR_ARG, "gpa_unwrap_finish: null plan");
  HIP_TRY(hipSetDevice(p->device));
  int iters = 0;
  hipError_t e = unwrap_finish(&p->uw, &iters, p->stream);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(e));
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
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, psi, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  if (weight) HIP_TRY(hipMemcpyAsync(p->d_wnorm, weight, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  int iters = 0;
  hipError_t e = unwrap_run(&p->uw, p->d_image, nullptr, weight ? p->d_wnorm : nullptr, true, kmax, eps,
                            compat != 0, p->d_u, &iters, p->stream);
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(e));
  if (iters_out) *iters_out = iters;
  HIP_TRY(hipMemcpyAsync(phi, p->d_u, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// ---- fused driver --------------------------------------------------------------
// host-side preparation: filter / carrier / k-matrix tables (re-staged only when they change; these upload
// synchronously), the x-plane buffer, and the second unwrap workspace + stream
static int extract_stage(gpa_plan* p, const double* kvecs, int P, const double* klists, int K, double sigma, int* Bx) {
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
static int extract_launch(gpa_plan* p, const void* image, int P, int K, int Bx, int mask_border, int kmax, void* u,
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
    if (ep != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(ep));
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
  // (not while profiling -- the per-kernel event pairs belong to this thread -- or capturing a graph)
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
  if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("unwrap: ") + hipGetErrorString(e));
  HIP_TRY(hipEventRecord(p->ev_join, p->stream2));
  HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_join, 0));
  if (p->profiling) HIP_TRY(hipEventRecord(p->stage_ev[5], p->stream));
  return GPA_OK;
}

static void drop_graphs(gpa_plan* p) {
  if (p->graphs.empty()) return;
  (void)hipStreamSynchronize(p->stream);   // an executable graph may still be running
  for (auto& g : p->graphs) {
    if (g.exec) hipGraphExecDestroy(g.exec);
    if (g.graph) hipGraphDestroy(g.graph);
  }
  p->graphs.clear();
}

// enqueue the whole driver on the plan's streams without any host synchronisation.
// A call is ~110 kernel launches.  They depend only on (pointers, P, K, x-planes, border, kmax) -- the tables
// the kernels read are restaged in place by extract_stage -- so with GPA_USE_GRAPH=1 the second call with one
// key captures them into a hipGraph (both streams) and later calls replay it.  Measured on MI355X / ROCm 7.2
// (profiles/r02_graph_vs_eager.txt) the replay is NOT faster than eager launches at any size (512^2: 0.65 vs
// 0.60 ms; 4096^2: equal) and it serialises with the copy stream of gpa_download_async (9.4 vs 6.75 ms with
// the D2H of u in the step), so eager launching is the default.
static int extract_enqueue(gpa_plan* p, const void* image, const double* kvecs, int P, const double* klists, int K,
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
  if (p->profiling || !p->use_graphs) return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  const GraphKey key = {image, u, lk, kidx, P, K, Bx, mask_border, kmax, p->tbuf_epoch, lockins != nullptr ? 1 : 0, 0};
  GraphEntry* ent = nullptr;
  for (auto& g : p->graphs)
    if (memcmp(&g.key, &key, sizeof(GraphKey)) == 0) ent = &g;
  if (ent && ent->exec) {
    HIP_TRY(hipGraphLaunch(ent->exec, p->stream));
    return GPA_OK;
  }
  if (!ent) {
    // first call with this key: run eagerly (lazy allocations and function attributes happen here)
    if (p->graphs.size() >= 8) drop_graphs(p);
    GraphEntry g{};
    g.key = key;
    p->graphs.push_back(g);
    return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  }
  if (ent->failed) return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  // second call: capture
  hipError_t ce = hipStreamBeginCapture(p->stream, hipStreamCaptureModeRelaxed);
  if (ce != hipSuccess) {
    (void)hipGetLastError();
    ent->failed = true;
    return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  }
  const int rc = extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  hipGraph_t graph = nullptr;
  ce = hipStreamEndCapture(p->stream, &graph);
  if (rc != GPA_OK || ce != hipSuccess || !graph) {
    (void)hipGetLastError();
    if (graph) hipGraphDestroy(graph);
    ent->failed = true;
    if (rc != GPA_OK) return rc;
    return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  }
  hipGraphExec_t exec = nullptr;
  ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (ce != hipSuccess || !exec) {
    (void)hipGetLastError();
    hipGraphDestroy(graph);
    ent->failed = true;
    return extract_launch(p, image, P, K, Bx, mask_border, kmax, u, lk, kidx, lockins != nullptr);
  }
  ent->graph = graph;
  ent->exec = exec;
  HIP_TRY(hipGraphLaunch(exec, p->stream));
  return GPA_OK;
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
  if (p->sh_use && p->ax1.native && !opt_set(OPT_NATIVE_SHARED)) p->sh_use = false;   // (as in passB_select)
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
static int tile_gradients_impl(gpa_plan* p, const void* image, size_t image_pitch, int r0, int c0, bool mean_on_device,
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
  if (need > p->tsum_cap || !p->d_ticket) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->d_tsum_part) (void)hipFree(p->d_tsum_part);
    p->d_tsum_part = nullptr;
    HIP_TRY(hipMalloc((void**)&p->d_tsum_part, need * sizeof(double)));
    p->tsum_cap = need;
    if (!p->d_ticket) {
      HIP_TRY(hipMalloc((void**)&p->d_ticket, 16));
      HIP_TRY(hipMemsetAsync(p->d_ticket, 0, 16, p->stream));
    }
  }
  HIP_TRY(launch_tile_sums(p->dtype, wins, win_stride, win_pitch, rects_dev, ntiles, max_rows, p->d_tsum_part, p->d_ticket,
                           sum_dev, p->stream));
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
static int plan_event(gpa_plan* p) {
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

static int invert_u_host(gpa_plan* p, const void* u, int iters, int edge, int shift, void* out, int mode, bool overlap) {
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

// periodic-component DFT of the image in p->d_image -> p->d_lockin (plane 0)
static int per_dft_staged(gpa_plan* p) {
  if (!p->bx0.tw) {
    size_t b = 0;
    hipError_t e = blue_axis_create(p->dtype, p->n0, p->stream, &p->bx0, &b);
    if (e == hipSuccess) e = blue_axis_create(p->dtype, p->n1, p->stream, &p->bx1, &b);
    if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_per_dft tables: ") + hipGetErrorString(e));
    p->ws_bytes += b;
  }
  // border-difference vectors: d0 holds n1, d1 holds n0 complex values
  void* d0 = p->d_aux1;
  void* d1 = p->d_aux0;
  HIP_TRY(per_pack(p->dtype, p->d_image, p->n0, p->n1, p->Tbuf, d0, d1, p->stream));
  HIP_TRY(dft2_inplace(p->dtype, p->bx0, p->bx1, p->Tbuf, p->stream));
  HIP_TRY(dft_rows_inplace(p->dtype, p->bx1, 1, d0, p->stream));
  HIP_TRY(dft_rows_inplace(p->dtype, p->bx0, 1, d1, p->stream));
  HIP_TRY(per_combine(p->dtype, p->Tbuf, d0, d1, p->n0, p->n1, p->d_lockin, p->stream));
  return GPA_OK;
}

int gpa_per_dft(gpa_plan* p, const void* image, void* out) {
  if (!p || !image || !out) return fail(GPA_ERR_ARG, "gpa_per_dft: null argument");
  HIP_TRY(hipSetDevice(p->device));
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, p->stream));
  TRY(per_dft_staged(p));
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
  TRY(per_dft_staged(p));   // p_hat in d_lockin, u_hat still in Tbuf
  if (!inverse_dft) {
    HIP_TRY(hipMemcpyAsync(p_out, p->d_lockin, npx * p->csz, hipMemcpyDeviceToHost, p->stream));
    if (s_out) {
      HIP_TRY(per_smooth_hat(p->dtype, p->Tbuf, p->d_lockin, npx, p->stream));
      HIP_TRY(hipMemcpyAsync(s_out, p->Tbuf, npx * p->csz, hipMemcpyDeviceToHost, p->stream));
    }
  } else {
    HIP_TRY(per_components(p->dtype, p->bx0, p->bx1, p->d_lockin, p->d_image, p->d_wnorm, p->d_dudx, p->stream));
    HIP_TRY(hipMemcpyAsync(p_out, p->d_wnorm, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(s_out, p->d_dudx, npx * p->rsz, hipMemcpyDeviceToHost, p->stream));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GPA_OK;
}

// ---- f-3 -------------------------------------------------------------------------
// scipy.ndimage._filters._gaussian_kernel1d (order 0): exp(-x^2 / 2 sigma^2) / sum, radius int(4 sigma + 0.5)
static int gaussian_weights(double sigma, std::vector<double>& w) {
  const int R = (int)(4.0 * sigma + 0.5);
  w.resize(2 * (size_t)R + 1);
  double sum = 0.0;
  for (int k = -R; k <= R; ++k) { w[k + R] = exp(-0.5 / (sigma * sigma) * (double)k * k); sum += w[k + R]; }
  for (double& v : w) v /= sum;
  return R;
}

int gpa_find_peaks(gpa_plan* p, const void* image, double sigma, double dog_sigma, double threshold_rel, int max_out,
                   int32_t* coords, void* values, int* count_out, void* smooth_out) {
  if (!p || !image || !coords || !values || !count_out) return fail(GPA_ERR_ARG, "gpa_find_peaks: null argument");
  if (!(sigma > 0.0) || max_out < 1) return fail(GPA_ERR_ARG, "gpa_find_peaks: need sigma > 0, max_out >= 1");
  if (p->n0 < 3 || p->n1 < 3) return fail(GPA_ERR_STATE, "gpa_find_peaks: image too small");
  HIP_TRY(hipSetDevice(p->device));
  const int n0 = p->n0, n1 = p->n1;
  const size_t npx = (size_t)n0 * n1;
  // candidates land in d_kidx (max_peaks * npx ints, two per candidate) and d_dudx (2 npx reals)
  const size_t cap = std::min((size_t)p->max_peaks * npx / 2, 2 * npx);
  if ((size_t)max_out > cap) max_out = (int)cap;
  std::vector<double> w1, w2;
  const int R1 = gaussian_weights(sigma, w1);
  const int R2 = dog_sigma > 0.0 ? gaussian_weights(dog_sigma, w2) : 0;
  // d_scratch (4096 doubles): [0, 1024) min/max partials + threshold, [1024, 4096) filter weights
  if (2 * R1 + 1 > 3072 || 2 * R2 + 1 > 3072) return fail(GPA_ERR_ARG, "gpa_find_peaks: sigma too large (radius > 1535)");
  hipStream_t st = p->stream;
  double* d_w = p->d_scratch + 1024;
  double* d_thr = p->d_scratch + 600;
  int* d_count = reinterpret_cast<int*>(p->d_scratch + 610);
  HIP_TRY(hipMemcpyAsync(p->d_image, image, npx * p->rsz, hipMemcpyHostToDevice, st));
  TRY(per_dft_staged(p));                                                    // p_hat in d_lockin
  void* fftim = p->d_image;                                                  // the staged image is consumed
  void* tmp = p->d_wnorm;
  void* smooth = p->d_u;
  HIP_TRY(launch_absshift(p->dtype, p->d_lockin, n0, n1, fftim, st));
  HIP_TRY(hipMemcpyAsync(d_w, w1.data(), w1.size() * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(launch_gauss1d(p->dtype, fftim, tmp, n0, n1, 0, d_w, R1, nullptr, st));
  HIP_TRY(launch_gauss1d(p->dtype, tmp, smooth, n0, n1, 1, d_w, R1, nullptr, st));
  if (dog_sigma > 0.0) {
    HIP_TRY(hipStreamSynchronize(st));   // w1 (pageable) and the weight slot are reused
    HIP_TRY(hipMemcpyAsync(d_w, w2.data(), w2.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(launch_gauss1d(p->dtype, fftim, tmp, n0, n1, 0, d_w, R2, nullptr, st));
    HIP_TRY(launch_gauss1d(p->dtype, tmp, smooth, n0, n1, 1, d_w, R2, smooth, st));
  }
  void* d_vals = p->d_dudx;                                                  // 2 npx reals >= max_out values
  HIP_TRY(launch_localmax(p->dtype, smooth, n0, n1, threshold_rel, p->d_scratch, d_thr, max_out, d_count, p->d_kidx,
                          d_vals, st));
  int count = 0;
  HIP_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int stored = count < max_out ? count : max_out;
  if (stored > 0) {
    HIP_TRY(hipMemcpyAsync(coords, p->d_kidx, (size_t)stored * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(values, d_vals, (size_t)stored * p->rsz, hipMemcpyDeviceToHost, st));
  }
  if (smooth_out) HIP_TRY(hipMemcpyAsync(smooth_out, smooth, npx * p->rsz, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *count_out = count;
  return GPA_OK;
}

// ---- f-4 (gaussian_deconvolve) -----------------------------------------------------
int gpa_gaussian_deconvolve(gpa_plan* p, const void* data, int dr, double sigma, double balance, void* out) {
  if (!p || !data || !out) return fail(GPA_ERR_ARG, "gpa_gaussian_deconvolve: null argument");
  if (dr < 0 || !(sigma > 0.0) || !(balance >= 0.0)) return fail(GPA_ERR_ARG, "gpa_gaussian_deconvolve: need dr >= 0, sigma > 0, balance >= 0");
  const int pad = 2 * dr, n0 = p->n0, n1 = p->n1, m0 = n0 - 2 * pad, m1 = n1 - 2 * pad;
  if (m0 < 2 || m1 < 2 || pad >= m0 || pad >= m1)
    return fail(GPA_ERR_STATE, "gpa_gaussian_deconvolve: the plan must have the padded shape (m + 4 dr), with 2 dr < m");
  HIP_TRY(hipSetDevice(p->device));
  if (!p->bx0.tw) {
    size_t b = 0;
    hipError_t e = blue_axis_create(p->dtype, p->n0, p->stream, &p->bx0, &b);
    if (e == hipSuccess) e = blue_axis_create(p->dtype, p->n1, p->stream, &p->bx1, &b);
    if (e != hipSuccess) return fail(GPA_ERR_HIP, std::string("gpa_gaussian_deconvolve tables: ") + hipGetErrorString(e));
    p->ws_bytes += b;
  }
  hipStream_t st = p->stream;
  // k-space Gaussian factors (doubles)
  std::vector<double> gx = gaussian_kspace(n0, sigma), gy = gaussian_kspace(n1, sigma);
  double* d_gx = reinterpret_cast<double*>(p->d_aux0);
  double* d_gy = reinterpret_cast<double*>(p->d_aux1);
  HIP_TRY(hipMemcpyAsync(d_gx, gx.data(), (size_t)n0 * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_gy, gy.data(), (size_t)n1 * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(p->d_image, data, (size_t)m0 * m1 * p->rsz, hipMemcpyHostToDevice, st));
  HIP_TRY(launch_deconv_pack(p->dtype, p->d_image, m0, m1, pad, p->Tbuf, st));
  HIP_TRY(dft2_inplace(p->dtype, p->bx0, p->bx1, p->Tbuf, st));
  HIP_TRY(launch_deconv_filter(p->dtype, p->Tbuf, n0, n1, d_gx, d_gy, balance, st));
  HIP_TRY(dft2_inplace(p->dtype, p->bx0, p->bx1, p->Tbuf, st));
  HIP_TRY(launch_deconv_unpack(p->dtype, p->Tbuf, m0, m1, pad, p->d_wnorm, st));
  HIP_TRY(hipMemcpyAsync(out, p->d_wnorm, (size_t)m0 * m1 * p->rsz, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));   // gx / gy are pageable host vectors
  return GPA_OK;
}

// ---- f-2 -------------------------------------------------------------------------
int gpa_phasegradient2J_dev(gpa_plan* p, const double* kvecs, int P, const void* grads, const void* weights,
                            double nmperpixel, const double* dks, void* J) {
  if (!p || !kvecs || !grads || !weights || !J) return fail(GPA_ERR_ARG, "gpa_phasegradient2J: null argument");
  if (P < 2 || P > p->max_peaks) return fail(GPA_ERR_STATE, "gpa_phasegradient2J: need 2 <= P <= 8 (and P <= max_batch)");
  if (!(nmperpixel > 0.0)) return fail(GPA_ERR_ARG, "gpa_phasegradient2J: nmperpixel must be positive");
  HIP_TRY(hipSetDevice(p->device));
  double kiso[16];
  for (int i = 0; i < 2 * P; ++i) kiso[i] = kvecs[i] + (dks ? dks[i] : 0.0);
  TRY(stage_kmat(p, kiso, P));
  HIP_TRY(launch_jacobian(p->dtype, grads, weights, p->d_kmat, P, (size_t)p->n0 * p->n1, nmperpixel, dks, J, p->stream));
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
static bool solve3(const double* m /*uu uv u vv v 1*/, const double* b, double* x) {
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
  const int n0 = p->n0, n1 = p->n1;
  // centred, unit-scaled coordinates keep the normal matrix well conditioned
  const double cx = 0.5 * (n0 - 1), cy = 0.5 * (n1 - 1), sx = 0.5 * n0, sy = 0.5 * n1;
  double c[3] = {0.0, 0.0, 0.0};   // start at the zero plane like the reference (x0 = [0, 0, 0])
  double sums[10];
  int it = 0;
  for (; it < max_iter; ++it) {
    HIP_TRY(launch_huber_moments(p->dtype, image, n0, n1, c, cx, cy, sx, sy, p->d_scratch, p->stream));
    HIP_TRY(hipMemcpyAsync(sums, p->d_scratch + 2560, sizeof(sums), hipMemcpyDeviceToHost, p->stream));
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
  if (iters_out) *iters_out = it;
  return GPA_OK;
}

int gpa_fit_plane(gpa_plan* p, const void* image, int max_iter, double tol, double* coef, int* iters_out) {
  if (!p || !image || !coef) return fail(GPA_ERR_ARG, "gpa_fit_plane: null argument");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipMemcpyAsync(p->d_image, image, (size_t)p->n0 * p->n1 * p->rsz, hipMemcpyHostToDevice, p->stream));
  return gpa_fit_plane_dev(p, p->d_image, max_iter, tol, coef, iters_out);
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
