// Internal interfaces between the translation units of libgpa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpa_fft.h"
#include "gpa_mrfft.h"

namespace gpa {

// ---- per-kernel timing (bench.py's per-kernel rooflines) ----------------------------------------
// While a KernelProfiler is installed on the calling thread (gpa_set_profiling), every launch site of
// the fused driver brackets its kernel with a pair of HIP events on the stream it launches on; the
// driver sums the pairs by kernel name after synchronising.  No profiler installed = two pointer tests.
struct KernelProfiler {
  struct Rec { const char* name; hipEvent_t a, b; };
  static constexpr int MAXREC = 4096;
  Rec rec[MAXREC];
  int n = 0;
  hipEvent_t pool[2 * MAXREC];
  int npool = 0;
};
extern thread_local KernelProfiler* g_kprof;
struct ProfScope {
  hipStream_t s;
  KernelProfiler::Rec* r = nullptr;
  ProfScope(const char* name, hipStream_t stream) : s(stream) {
    KernelProfiler* k = g_kprof;
    if (!k || k->n >= KernelProfiler::MAXREC) return;
    for (int j = 0; j < 2; ++j)
      if (k->npool < 2 * (k->n + 1)) { if (hipEventCreate(&k->pool[k->npool]) != hipSuccess) return; ++k->npool; }
    r = &k->rec[k->n];
    r->name = name;
    r->a = k->pool[2 * k->n];
    r->b = k->pool[2 * k->n + 1];
    ++k->n;
    (void)hipEventRecord(r->a, s);
  }
  ~ProfScope() { if (r) (void)hipEventRecord(r->b, s); }
};
// XCD-aware tile order of the column kernels.  Workgroups are dealt to the 8 XCDs round robin (block b -> XCD b % 8),
// and every XCD has its own L2: neighbouring column tiles share 128-byte lines of the rows they touch, so they must
// meet in ONE L2 -- otherwise every XCD fetches (and partially writes) the line for itself.  XCD x therefore gets a
// contiguous run of tiles: start(x) = x * (grid / 8) + min(x, grid % 8).  (Counters at 3000^2, where the grid of 375
// tiles is not a multiple of 8 and the old power-of-two-only remap did nothing: the transform-free column kernel
// moved 181 MB per launch for 72 MB of algorithmic traffic.)
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_tile(int b, int grid) {
  const int base = grid >> 3, rem = grid & 7, x = b & 7, idx = b >> 3;
  return x * base + (x < rem ? x : rem) + idx;
}
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel instantiation and device instead of once per
// launch: the fused driver makes ~110 launches per image and a small image is bound by the host's launch rate.
// `done` is a static of the calling template instantiation (one bit per device; a benign race repeats the call).
inline hipError_t set_dynamic_lds_once(const void* kern, int bytes, unsigned& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned bit = 1u << (dev & 31);
  if (done & bit) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done |= bit;
  return e;
}

// compute units of the current device (one slot per device index)
inline int device_cus() {
  static int n[32] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& slot = n[dev & 31];
  if (!slot) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    slot = cus;
  }
  return slot;
}

// ---- run-time options (gpa_set_option, include/gpa_hip.h) ---------------------------------------
// Diagnostic / test switches of the library.  The table is filled ONCE from the environment (GPA_<NAME>) when the
// library first looks at it and changed afterwards only through gpa_set_option(): no getenv() on any call path.
// A switch is "set" when it has a value at all (GPA_NO_LAT=0 is set, as it always was); num = atof(value).
enum OptKey {
  OPT_PBS_FULLBAND, OPT_SERIAL_UNWRAP, OPT_NO_WORKER, OPT_NO_KSPLIT, OPT_NO_COMPACT, OPT_NO_SHARED,
  OPT_NO_PAIR, OPT_PBS_E8, OPT_TRI_SMALL, OPT_TRI_Q, OPT_NO_MR, OPT_MR_FORCE_BLUESTEIN, OPT_NO_ROWPQ,
  OPT_COLSOLVE, OPT_NO_LAT, OPT_F32_EPS_FLOOR, OPT_COLSTREAM_CHUNK, OPT_NO_ROWHALF, OPT_PAIR_MAXSIDE, OPT_ROWHALF_MINLG, OPT_NO_PQDCT,
  OPT_NO_REORDER, OPT_NO_RAW, OPT_NO_TILEFUSE, OPT_NO_ROWPERS, OPT_NO_LFTILE, OPT_LF_ALL_ROUNDS, OPT_DFT_ENGINE, OPT_GAUSS_FFT_MINR, OPT_NO_GAUSS2D, OPT_NO_DFT_HALF, OPT_F32_STALL, OPT_PBS_LDS_PAD, OPT_NO_SHARED_PHASES, OPT_PA_STAG, OPT_PA_STAG_TICKS, OPT_PA_ROT, OPT_COUNT
};
struct OptVal {
  bool set;
  double num;
  char str[24];
};
const OptVal& opt(OptKey k);
inline bool opt_set(OptKey k) { return opt(k).set; }

#define GPA_PROF_CAT2(a, b) a##b
#define GPA_PROF_CAT(a, b) GPA_PROF_CAT2(a, b)
#define GPA_PROF(name, stream) ::gpa::ProfScope GPA_PROF_CAT(_gpa_prof_, __LINE__)(name, stream)

// One image axis: n samples, transformed with a power-of-two FFT of length L.
// n == L  : periodic mode, the k-space Gaussian is applied bin by bin.
// n <  L  : padded mode (L >= 2n-1): the circular convolution of length n is
//           evaluated as a linear convolution of the periodically extended input
//           with the full length-n spatial kernel -- the same numbers as
//           IDFT_n(G * DFT_n(a)), without ever needing a length-n FFT.
//           The extension covers extL samples to the left (FFT slots [L - extL, L)) and extR to the right
//           (slots [n, n + extR)) of the n samples in slots [0, n).  Full form: extL = n - 1, extR = 0, the
//           kernel's n taps at lags 0 .. n-1, L >= 2n - 1.  Compact form (chosen per sigma when it gives a shorter
//           transform): the kernel of a Gaussian filter is negligible beyond E samples (the host measures E on
//           the actual taps: everything dropped sums to < 1e-14 (f64) / 1e-9 (f32) of sum|h|), so lags -E .. E
//           suffice, extL = extR = E and L >= n + 2E.
// (Round 4's opt-in native-length mode of the sweep -- pass A / per-candidate pass B on the mixed-radix engine at the axis' own
//  length -- measured slower at every size (profiles/r04_native_sweep_rejected.txt) and was removed in round 5; git history.)
struct Axis {
  int n;
  int lg;      // log2(L)
  int L;
  bool padded;
  int extL, extR;
};
// threads a transform of the axis is spread over = entries per candidate of the carrier base table
inline int axis_tpf(const Axis& a) { return a.L / 16; }

// source sample of FFT slot m (-1: zero padding)
GPA_HD int axis_src(int m, int n, int L, bool padded, int extL, int extR) {
  if (!padded) return m;
  if (m < n) return m;
  if (m < n + extR) return m - n;
  if (m >= L - extL) return m - L + n;
  return -1;
}

// Device tables of one batch of B lock-ins, element type cpx<T> of the plan dtype.
// Carriers are factored as  exp(2 pi i w (t + (L/16) i)) = base[t] * stride[i]  (times a
// wrap factor for the periodically extended slots of padded mode), so a thread needs one
// table load plus 16 wave-uniform scalars per lock-in.
//
// x-planes: pass A's output depends on the candidate only through wx (cy[y] is a
// per-column scalar that commutes with the x-axis filter).  The B candidates are
// therefore mapped onto Bx <= B "x-planes", one per DISTINCT wx; pass A computes and
// stores one complex plane per x-plane and pass B multiplies cy in on load.  A 4x4
// candidate grid needs 4 planes per peak instead of 16, the reference's 7x7 grid 7
// instead of 49.
struct SweepTables {
  void* cxb;   // [Bx][L0/16] exp(2 pi i wx t): carrier along x at the thread's base row
  void* sx;    // [Bx][16]    exp(2 pi i wx (L0/16) i): per-register stride factor
  void* wxw;   // [Bx]        exp(-2 pi i wx (L0 - n0)): wrap factor of the left extension (padded mode)
  void* wxr;   // [Bx]        exp(-2 pi i wx n0): wrap factor of the right extension (compact padded mode)
  void* cyb;   // [B][L1/16]  the same four for the y carrier of every candidate
  void* sy;    // [B][16]
  void* wyw;   // [B]
  void* wyr;   // [B]
  void* dx;    // [B][n0]     exp(-2 pi i (wx - kx) x)   compensation to the peak centre
  void* dy;    // [B][n1]     exp(-2 pi i (wy - ky) y)
  int* planeof;   // [B] x-plane of each candidate
};

// ---- sweep (gpa_sweep.hip) --------------------------------------------------
// kl: device [B][2] doubles (wx, wy); kr: device [B][2] doubles (kx, ky) of the
// peak each candidate belongs to.
// pw: device [Bx] doubles, the distinct wx values
hipError_t launch_tables(int dtype, const Axis& a0, const Axis& a1, const double* kl,
                         const double* kr, int B, const double* pw, int Bx, const SweepTables& tb,
                         hipStream_t s);
// mean_out: device scalar of the plan dtype; scratch: >= 1024 doubles
hipError_t launch_mean(int dtype, const void* image, size_t count, double* scratch,
                       void* mean_out, hipStream_t s, int nimg = 1);
// x-axis pass over the Bx x-planes: Tbuf[plane][x][y] = Cx( (image - mean) * cx_plane )[x][y]
hipError_t launch_passA(int dtype, const Axis& a0, int n1, const void* image, const void* mean,
                        const SweepTables& tb, const void* Hx, const void* tw0, void* Tbuf,
                        int Bx, hipStream_t s, int nimg = 1);
// y-axis pass.  select = true: per peak p (grid.y = P) loop over its K candidates
// keeping the strictly-largest |sf|, write compensated lock-in (+ kidx).
// select = false: write all B lock-ins (P = B, K = 1).
hipError_t launch_passB(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy,
                        const void* tw1, const SweepTables& tb, int P, int K, bool select,
                        void* out, int32_t* kidx, hipStream_t s, int nimg = 1, int Bx = 0);
// one peak, K candidates, the less travelled selection modes (gpa_sweep_ext.hip): mode 2 = gated selection of
// wfr4 (gate: device K x K bytes, gate[j * K + k] != 0 where candidate k may replace the kept candidate j),
// mode 3 = plain selection that also writes psi[k][x][y] = -angle(sf_k) of every candidate
hipError_t launch_passB_ext(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy, const void* tw1,
                            const SweepTables& tb, int K, int mode, void* out, int32_t* kidx, const uint8_t* gate,
                            void* psi, hipStream_t s);
// small images (transform length <= 1024): best-of-K with the K candidates of a row split over ksplit workgroups and a
// merge pass; part / pidx: ksplit * P * n0 * n1 complex / int32 of scratch.  Same winners and values as launch_passB.
hipError_t launch_passB_split(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* Hy, const void* tw1,
                              const SweepTables& tb, int P, int K, int ksplit, void* part, int32_t* pidx, void* out,
                              int32_t* kidx, hipStream_t s);
// a4: phase gradient of the winner from the per-candidate phases (mode 0 np.gradient, 1 forward differences with
// NaN at the end, 2 the same with swapped components)
// ystep (device, one double) != null: psi holds the phases of the shared pass B (compensated along x, lacking the phasor
// exp(i ystep y)): no 2 pi (w - k) is added, ystep is subtracted from the differences along y
hipError_t launch_phasegrad(int dtype, const void* psi, int K, const int32_t* kidx, int n0, int n1, const double* kl,
                            const double* kr, int mode, void* grad, hipStream_t s, const double* ystep = nullptr);
int passA_cols(int dtype, int lg);
// frequency bin held by (thread, register) after the forward transform of length 2^lg
int spec_index_rt(int lg, int tid, int reg, int elems = 16);   // elems: elements per thread of the transform (16 or 8)

// ---- reconstruct (gpa_reconstruct.hip) -------------------------------------
hipError_t launch_reconstruct(int dtype, const void* lockin, const double* kmat /*dev P*2, 2 pi k*/,
                              int P, int n0, int n1, int border, void* dudx, void* dudy,
                              void* wnorm, hipStream_t s, const double* ystep = nullptr /* as in launch_reconstruct_setup */);

// the same, storing only the interior rectangle (origin i0, j0, size t0 x t1) of a halo window straight into the tile
// blocks of the pipeline: dx[c] / dy[c] = du_c/dx, du_c/dy (the difference fields are clipped to the columns / rows that
// exist), wn[0] and, if not null, wn[1] the weight; pitches in elements (gpa_tile_gradients_*)
hipError_t launch_reconstruct_tile(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1, int border,
                                   int i0, int j0, int t0, int t1, void* const dx[2], size_t dx_pitch, void* const dy[2],
                                   size_t dy_pitch, void* const wn[2], size_t wn_pitch, hipStream_t s,
                                   const double* ystep = nullptr);
// fused a5 + a6 + unwrap setup for both components (fused driver): wnorm, r0 of u_x / u_y and
// *nparts partial sums of ||r0||^2 each
hipError_t launch_reconstruct_setup(int dtype, const void* lockin, const double* kmat, int P, int n0, int n1,
                                    int border, void* wnorm, void* r0, void* r1, double* part0, double* part1,
                                    int* nparts, hipStream_t s, int nimg = 1, size_t rstride = 0, size_t pstride = 0,
                                    const double* ystep = nullptr /* device [P]: the lock-ins lack exp(i ystep_p y) */);

// pre_diff=True: grads (P x n0 x n1 x 2) are phase gradients along axis 1 ([..., 0]) and axis 0 ([..., 1]):
// wrap, per-pixel weighted least squares for both, crop to the difference grids
hipError_t launch_prediff(int dtype, const void* grads, const void* w, const double* kmat, int P, int n0, int n1,
                          void* dudx, void* dudy, void* wnorm, hipStream_t s);

hipError_t launch_wlstsq(int dtype, const void* b, const void* w, const double* kmat, int P, size_t npx,
                         void* out, hipStream_t s);

// f-2: grads (P x npx x 2) + weights (P x npx) -> J (npx x 2 x 2); Jac (+I) -> props (4 x npx)
hipError_t launch_jacobian(int dtype, const void* grads, const void* w, const double* kmat, int P, size_t npx,
                           double nmperpixel, const double* dks /*host P x 2 or null*/, void* J, hipStream_t s);
hipError_t launch_cabs(int dtype, const void* z, size_t count, void* out, hipStream_t s);
hipError_t launch_props(int dtype, const void* jac, size_t npx, int add_identity, double refangle, double refscale,
                        int diff, void* out, hipStream_t s);

// f-4: one IRLS pass of the Huber plane fit; ten sums at scratch + huber_sums_offset() (scratch: 10 more doubles than that)
int huber_sums_offset();
hipError_t launch_huber_moments(int dtype, const void* img, int n0, int n1, const double* coef, double cx, double cy,
                                double sx, double sy, double* scratch, hipStream_t s);

// ---- f-3 peak candidates (gpa_peaks.hip) --------------------------------------------------------
constexpr int PEAK_PARTS = 1024;   // workgroups of the min / max reduction (2 doubles each)
bool gauss2d_small_ok(int R);
hipError_t launch_gauss2d_small(int dtype, const void* in, void* out, int n0, int n1, const double* w, int R,
                                const void* minuend, hipStream_t s);
hipError_t launch_gauss1d(int dtype, const void* in, void* out, int n0, int n1, int axis, const double* w, int R,
                          const void* minuend, hipStream_t s);
hipError_t launch_localmax(int dtype, const void* smooth, int n0, int n1, double rel, double* part, double* thr,
                           int max_out, int* count, int32_t* coords, void* vals, hipStream_t s);

// f-4 gaussian_deconvolve pieces (gpa_peaks.hip); Z: (m0 + 2 pad) x (m1 + 2 pad) complex
hipError_t launch_deconv_pack(int dtype, const void* data, int m0, int m1, int pad, void* Z, hipStream_t s);
hipError_t launch_deconv_filter(int dtype, void* Z, int n0, int n1, const double* gx, const double* gy, double balance,
                                hipStream_t s);
hipError_t launch_deconv_unpack(int dtype, const void* Z, int m0, int m1, int pad, void* out, hipStream_t s);

// ---- tile pipeline (gpa_tiles.hip): all on stream s, nothing synchronises ------------------------------
// sum over the interior rectangles rects_dev[t] = (o0, o1, z0, z1) of ntiles windows (win_stride / pitch in elements);
// part: >= ntiles * tile_sums_bands(max_rows) doubles, ticket: one zeroed unsigned, out: one double
hipError_t launch_tile_sums(int dtype, const void* wins, size_t win_stride, size_t pitch, const int* rects_dev, int ntiles,
                            int max_rows, double* part, double* out, hipStream_t s);
int tile_sums_bands(int max_rows);
hipError_t launch_set_mean(int dtype, const double* sum, double scale, void* mean_out, hipStream_t s);
// nf <= 6 rectangular copies (rows x cols elements, pitches in elements) in one launch
hipError_t launch_copy_fields(int dtype, const void* const* src, void* const* dst, const size_t* src_pitch,
                              const size_t* dst_pitch, const int* rows, const int* cols, int nf, hipStream_t s);
// table_dev[t] = (slot, r0, c0, z0, z1); tiles[slot][f] (t0 x t1, pitch tile_pitch) -> dst[f] at (r0, c0), clipped to dst_rows / dst_cols
hipError_t launch_stitch(int dtype, const void* tiles, size_t slot_stride, size_t field_stride, size_t tile_pitch,
                         const int* table_dev, int ntiles, int t0, int t1, int nf, void* const* dst, const size_t* dst_pitch,
                         const int* dst_rows, const int* dst_cols, hipStream_t s);

// ---- f-1 Lawler-Fujita (gpa_warp.hip); both synchronise the stream before returning -----------
// mode: 0 = scipy 'nearest' (the reference's default), 1 = 'constant'; nan_last (mode 1 only): the last round samples
// with cval = NaN as invert_u_overlap does (geometric_phase_analysis.py:296-299), invert_u never does (:255-258)
// scratch + prefilter taps of the warp kernels, kept by the plan: no allocation and no host synchronisation per call
struct WarpWs {
  void* buf = nullptr;
  size_t cap = 0;
  void* taps = nullptr;
  int taps_dtype = -1;
  size_t* counted = nullptr;   // where the owner accounts the scratch (gpa_plan::ws_bytes), may be null
};
void warp_ws_free(WarpWs* ws);
// scratch of one undistort_image call on an n0 x n1 grid (inversion + resampling): reserved in ONE allocation up front
hipError_t warp_reserve_undistort(int dtype, int n0, int n1, WarpWs* ws, hipStream_t s);
// rects = nrect x {r0, c0, h, w}: the windows of the output grid that are computed (nrect = 0: all of it) -- the tiles a
// rank owns; the prefilter of the whole field runs once per call
hipError_t warp_invert_u(int dtype, const void* d_u, int n0, int n1, double scale, int iters, int edge, int shift,
                         void* d_out, hipStream_t s, int mode, int nan_last, WarpWs* ws, const int* rects = nullptr, int nrect = 0);
hipError_t warp_image(int dtype, const void* d_img, const void* d_uinv, int n0, int n1, void* d_out, hipStream_t s, WarpWs* ws,
                      const int* rects = nullptr, int nrect = 0);

}  // namespace gpa
