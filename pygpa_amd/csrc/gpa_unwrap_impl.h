// Shared between the translation units of the weighted unwrap (a7): the workspace (Impl), the scalar / flag / partial-sum
// conventions of the fused iteration, the block reductions every kernel uses, and the launch entry points each
// translation unit offers to the PCG driver (gpa_unwrap.hip):
//   gpa_unwrap_rows.hip     row kernels of the power-of-two fused iteration (rowdct_fused, rowidct_p, rowidct_pq);
//   gpa_unwrap_rowhalf.hip  the same for rows of 8192 points and more (half-length transforms); gpa_unwrap_pqdct.hip the
//                           stencil + forward transform in one launch; gpa_unwrap_rowpers.hip the persistent rowidct_p
//   gpa_unwrap_cols.hip     column solves (DCT kernel, transform-free recursion, streamed recursion)
//   gpa_unwrap_stencil.hip  set-up, stencil (pq), phi flush, the scalar / elementwise kernels of the plain scheme
//   gpa_unwrap_generic.hip  sizes that are not powers of two: mixed-radix fused kernels (gpa_unwrap_mr.h), Bluestein kernels
//   gpa_unwrap_tables.hip   workspace creation: twiddles, eigenvalue tables, chirps, the recursion's per-column constants
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "gpa_dct.h"
#include "gpa_mrfft.h"
#include "gpa_internal.h"
#include "gpa_unwrap.h"

namespace gpa {

constexpr int MAXPART = 65536;   // one partial sum per image row / per grid-stride block
constexpr int RING_MAX = 10;     // search directions kept so that phi is updated once per RING_MAX iterations
constexpr int SC_ALPHA = 16;     // scal[SC_ALPHA + j % ring] = alpha of iteration j
// Batched solves: blockIdx.z = problem.  The image-sized arrays of problem pb sit pb * pimg elements behind those of
// problem 0, its scalars / flags / partial sums SCAL_N / FLAGS_N / PART_N entries behind; problems 2i and 2i + 1 (the
// two displacement components of image i) share the weight of image i.  A launch with gridDim.z = 1 is the single
// solve it always was.
constexpr int SCAL_N = SC_ALPHA + RING_MAX + 6;
constexpr int FLAGS_N = 4;
constexpr size_t PART_N = (size_t)3 * MAXPART;

struct Impl {
  int dtype, n0, n1, lg0, lg1;
  int nprob;                 // problems solved per launch (blockIdx.z): 1, or 2 x images of a batched driver call
  int cap;                   // problems the buffers hold (nprob <= cap: unwrap_set_active)
  int iters_slot;            // flags[iters_slot] = iterations performed (3: fused iteration, 0: plain scheme)
  bool lat_ok;               // latency-tuned kernel variants allowed (GPA_NO_LAT unset), read once per solve
  bool supported;
  size_t rsz;
  void *r, *p, *p2, *q, *z;   // p / p2: double-buffered search direction (= ring[0], ring[1])
  int prepared_parts;        // number of partial norms of a prepared r0 (unwrap_enqueue_prepared)
  double stall_limit_dev;    // what scal[SC_STALL_LIMIT] of every problem holds (0: never written)
  void* ring[10];            // search directions of the last RING iterations (fused path), grown on demand
  int nring;
  void *tw0, *tw1;           // FFT twiddles per axis
  void* tw1h;                // twiddles of length n1 / 2: rows of 8192 points and more run half-length transforms
  void *wk1;                 // w_k along axis 1, natural order
  void *wk0s;                // w_k along axis 0, spectral layout
  void *ha0[2], *ham0[2];    // 1 - cos term of axis-0 bins (spectral layout); [compat]
  void *hb1[2];              // 1 - cos term of axis-1 bins (natural); [compat]
  int col_mode;              // COLSOLVE of the current solve: 0 default, 1 tri, 2 fft, 3 stream (read once per solve)
  void* tritab;              // TriCol per column (square images): transform-free column solve
  int triQ, triS, triR;      // its launch geometry, fixed when the table is built (the table depends on it)
  // streamed column solve (gpa_unwrap_colstream.hip; square images): per-column constants, chunk sums, chunk carries
  void* strtab;              // StreamCol per column
  double* strlam;            // lam per column
  void *stragg, *strcar;     // [cap][S][n1] double2: (a, b) of every chunk / (p entering it, z entering it from below)
  int strC, strS;            // rows per chunk, chunks per column
  double* scal;              // 8 doubles
  int* flags;                // [0] = iteration count, [1] = done
  double* part;              // 3 * MAXPART partial sums
  // generic sizes (any n0, n1 >= 2): DCTs through Bluestein DFTs of length n on FFTs of length Lb
  bool generic;
  int lgb0, lgb1;            // log2 of the Bluestein FFT lengths
  void *btw0, *btw1;         // twiddles of those FFTs
  void *chirp0, *chirp1;     // c_m = exp(i pi m^2 / n)
  void *bspec0, *bspec1;     // FFT_L(b)/L, spectral layout
  void *gwk0, *gwk1;         // w_k = exp(-i pi k / (2n)), natural order
  void *gha0[2], *gham0[2];  // 1 - cos term of axis-0 bins k and n0-k, natural order; [compat]
  // generic sizes whose axis lengths factor into 2, 3, 5, 7, 11, 13 (and rows of a multiple of 4 pixels): the fused
  // 4-kernel iteration on the mixed-radix FFT (gpa_unwrap_mr.h) instead of the Bluestein kernels
  bool mr_ok;
  MrDft mr0, mr1;            // per axis: direct transform of a smooth length, or chirp-z on a smooth L >= 2n - 1
  void *mrW0, *mrW1;         // twiddles of the transform that is run, w_L^i at mr_pad(i)
  void *mrB0, *mrB1;         // chirp-z only: FFT_L(b) / L in natural order (the chirp itself is chirp0 / chirp1)
};

struct TriCol {
  double lam, lamR, lamN, inv, zn;   // lam^ROWS, lam^N, 1 / (1 - lam^(2N)), -lam / (1 - lam); column 0: 1, 1, 1, 0, 0
};

// ---- launch entry points across the translation units (all enqueue on s, none synchronises) ----
// rows, power-of-two fused iteration (gpa_unwrap_rows.hip)
hipError_t pow2_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm,
                             int it, int* nnorm, int init, hipStream_t s);
hipError_t pow2_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                          hipStream_t s);
hipError_t pow2_rowidct_pq(const Impl* w, const void* pin, void* pout, const void* weight, const double* part_rho,
                           int nrho, double* part_pq, int* npq_out, int it, hipStream_t s);
// stencil + row transform in one launch (rows of 2048 / 4096 points): D = DCT_rows(A^T W^2 A p) into w->q, partial <p, q>
bool pow2_pqdct_offered(const Impl* w);
hipError_t pow2_pqdct(const Impl* w, const void* p, const void* weight, double* part_pq, int* npq, hipStream_t s);
// long rows: one row per half-length transform (gpa_unwrap_rowhalf.hip)
bool rowhalf_offered(const Impl* w);
hipError_t rowhalf_rowdct(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm, int it,
                          int* nnorm, int init, hipStream_t s);
hipError_t rowhalf_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it, hipStream_t s);
// persistent, software-pipelined rowidct_p for 4096-point f32 rows (gpa_unwrap_rowpers.hip; NO_ROWPERS: the one-pair-per-
// workgroup kernel of gpa_unwrap_rows.hip)
// gpa_unwrap_rowhalfpers.hip: the persistent, LDS-DMA-pipelined forms of the half-length kernels (f32, 8192 / 16384 points)
bool rowhalfpers_offered(const Impl* w);
hipError_t rowhalfpers_rowdct(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm, int it,
                              int* nnorm, int init, hipStream_t s);
hipError_t rowhalfpers_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it, hipStream_t s);
bool pow2_rowpers_offered(const Impl* w);
hipError_t pow2_rowidct_p_pers(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                               hipStream_t s);
// columns (gpa_unwrap_cols.hip): every size
hipError_t dispatch_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                             double eps, double* part_rho, int* nrho, const void* zin);
// streamed column solve (gpa_unwrap_colstream.hip)
int colstream_chunk(int n0, int n1);   // rows per chunk it would use for this shape, 0 = not offered
hipError_t build_streamtab(Impl* w, hipStream_t s, size_t* bytes);
hipError_t dispatch_colstream(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                              double eps, double* part_rho, int* nrho, const void* zin);
// the same after R (w->r) -= alpha dq, alpha from part_pq (the stencil-fused iteration); writes the partial ||r||^2
hipError_t dispatch_colstream_update(const Impl* w, int compat, hipStream_t s, int it, double eps, int ring, const void* dq,
                                     const double* part_pq, int npq, double* part_norm_out, int* nnorm_out, double* part_rho,
                                     int* nrho);
bool colstream_is_default(const Impl* w);   // the streamed solve is what dispatch_colsolve would run for this workspace / mode
// sizes that are not powers of two (gpa_unwrap_generic.hip)
hipError_t mr_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm,
                           int it, int* nnorm, int init, hipStream_t s);
hipError_t mr_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                        hipStream_t s);
hipError_t mr_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it, double eps,
                       double* part_rho, int* nrho, const void* zin);
hipError_t dispatch_g_rowdct(const Impl* w, hipStream_t s);
hipError_t dispatch_g_rowidct(const Impl* w, int* nparts, hipStream_t s);
hipError_t dispatch_g_colsolve(const Impl* w, int compat, hipStream_t s);
// set-up, stencil, flush, plain scheme (gpa_unwrap_stencil.hip)
hipError_t launch_unwrap_setup(const Impl* w, const void* a, const void* b, const void* weight, bool from_psi, void* phi,
                               hipStream_t s);
hipError_t launch_scal_init(const Impl* w, int nparts, hipStream_t s);
// need_q = false: only the partial sums of <p, q> are wanted (the last iteration of a solve): q is not written
hipError_t launch_pq(const Impl* w, const void* p, const void* weight, int it, double* part_pq, hipStream_t s, bool need_q = true);
int pq_partials(const Impl* w);   // partial sums launch_pq writes
// ring / final_it / init as phi_flush_kernel; commits (flags[2] = flags[0]) unless final_it
hipError_t launch_phi_flush(const Impl* w, int ring, void* phi, bool phi_unwritten, int final_it, const double* part_pq,
                            int npq, hipStream_t s);
hipError_t launch_plain_tail(const Impl* w, const void* weight, void* phi, int nrow, int kmax, double eps, hipStream_t s);
// tables (gpa_unwrap_tables.hip)
hipError_t build_tritab(Impl* w, hipStream_t s, size_t* bytes);

#if defined(__HIPCC__)
namespace {

template <class T> struct C2 { static constexpr T pi = T(3.14159265358979323846), two_pi = T(6.28318530717958647692); };

template <class T>
__device__ __forceinline__ T wrap_pi(T x) {
  const T t = x + C2<T>::pi;
  return t - C2<T>::two_pi * floor(t / C2<T>::two_pi) - C2<T>::pi;
}

// block-wide sum of a double (deterministic: fixed shuffle tree per wavefront, then a fixed
// order over the wavefronts); the result is returned to every thread.  sh: >= 17 doubles.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}
__device__ __forceinline__ double block_sum(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();   // sh may still be read from a previous call
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

// the loads of reduce_partials() alone: a kernel requests them with all its other inputs and reduces
// (block_sum) after its single wait
__device__ __forceinline__ double load_partials(const double* __restrict__ part, int n) {
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
  return acc;
}
// sum of n partial sums written by an EARLIER kernel, computed identically (same order) by
// every workgroup that needs it: a consumer-side reduction that costs no launch
__device__ __forceinline__ double reduce_partials(const double* __restrict__ part, int n, double* sh) {
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
  return block_sum(acc, sh);
}

// Start of a solve on prepared residuals, folded into the first row kernel of the fused iteration (it used to be
// a one-block kernel of its own): ||r0||^2 from the producer's partial sums, evaluated by every workgroup in the
// same order; block 0 files it and resets the flags, which no other workgroup of that launch reads.
// Returns false when r0 == 0 everywhere: nothing to do (phase_unwrap.py:326).
__device__ __forceinline__ bool solve_init(const double* __restrict__ part0, int nparts, double* scal, int* flags,
                                           double* sh) {
  const double tot = reduce_partials(part0, nparts, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[5] = tot;   // ||r0||^2
    scal[6] = tot;
    scal[7] = tot;   // smallest ||r||^2 seen
    scal[10] = tot;
    scal[11] = tot;
    scal[SC_ALPHA + RING_MAX] = 0.0;       // SC_STALL (defined below): iterations since the last new minimum
    scal[SC_ALPHA + RING_MAX + 1] = 0.0;
    scal[1] = 0.0;
    flags[0] = 0;
    flags[2] = 0;
    flags[3] = 0;
    flags[1] = tot == 0.0 ? 1 : 0;
  }
  return tot != 0.0;
}

// Breakdown guard of the stopping test (not in the reference, which iterates in f64 only).
//  (1) NaN, or a residual 100 x above the smallest one seen: CG has lost conjugacy and would iterate into garbage.
//  (2) f32 only, and only once the residual is at the f32 rounding floor (||r|| < 1e-5 ||r0||): PCG_STALL consecutive
//      iterations without a new smallest residual.  An f32 iteration cannot improve further there, and pushed on it DRIFTS:
//      rounding feeds the near-null low modes of the weighted Laplacian, which the residual does not see (measured: 2 % of
//      |phi| at kmax = 100 on the 63 x 65 golden case, where the f64 reference converges after 15 iterations).
// While the residual keeps falling -- the whole of a kmax = 10 solve of the benchmark image -- neither acts and the
// reference's test (phase_unwrap.py:348) decides alone.  stall_prev / *stall_new: iterations since the last new minimum
// (scal[SC_STALL + parity], double-buffered like the minimum itself).
constexpr int SC_STALL = SC_ALPHA + RING_MAX;   // two slots
// scal[SC_STALL_LIMIT]: the PCG_STALL of guard (2), written by the host (set_stall_limit_kernel) when a solve is enqueued:
// 2 by default, the library option F32_STALL=<n> sets another count, F32_STALL=0 switches guard (2) off (the reference's loop,
// phase_unwrap.py:326-349, has no such stop).  solve_init() leaves the slot alone.
constexpr int SC_STALL_LIMIT = SC_STALL + 2;
constexpr double PCG_STALL = 2.0;
__device__ __forceinline__ bool pcg_breakdown(double tot, double best, double norm0, bool f32, double stall_prev,
                                              double* stall_new, double stall_limit) {
  const double st = tot < best ? 0.0 : stall_prev + 1.0;
  *stall_new = st;
  if (!(tot == tot) || tot > 1e4 * best) return true;
  return f32 && best < 1e-10 * norm0 && st >= stall_limit;
}

// Scalars of the fused power-of-two path (single writer = block 0 of the named kernel;
// values a kernel both reads and replaces are double-buffered by iteration parity):
//   scal[5] = ||r0||^2                    (scal_init_kernel)
//   scal[8 + (it & 1)]  = rho of iteration it            (rowidct_p_kernel)
//   scal[10 + (it & 1)] = smallest ||r||^2 up to it      (colsolve_kernel)
//   scal[16 + (j % ring)] = alpha of iteration j         (rowdct_fused_kernel; the last one: final phi_flush_kernel)
//   flags[0] = completed updates k, flags[1] = done      (colsolve_kernel / final kernel)
//   flags[2] = updates already applied to phi            (phi_commit_kernel)
//   flags[3] = iterations performed, for the host        (the final phi_flush_kernel)

template <class T>
struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };
// V consecutive pixels of a row: V = 4 where rows are whole 16-byte (f32) vectors, V = 1 for any other row length
template <class T, int V>
struct alignas(V * sizeof(T)) VecN { T v[V]; };

}  // namespace
#endif

// Elements per thread of the fused kernels' transforms: 8 for axes up to 2048, 16 for longer ones.  A 512-point
// kernel puts ONE wavefront on a SIMD and is bound by that wavefront's instruction stream (2000-3200 instructions at
// 16 elements per thread; profiles/r02_gridbarrier_microbench.txt): half the elements per thread on twice the
// threads cut it to 1000-1900 (512^2: 635 -> 755 Mpix/s, 256^2: 189 -> 221, 2048^2: +2 %, f64 2048^2: +3 % and no
// spills).  At 4096 points the fourth pass and its LDS exchange cost more than the shorter stream saves
// (f32 2567 -> 2322 Mpix/s, f64 1045 -> 997): long axes keep 16.
#ifndef GPA_UNWRAP_E8_MAXLG
#define GPA_UNWRAP_E8_MAXLG 11
#endif
#ifndef GPA_UNWRAP_E8_MAXLG_F64
#define GPA_UNWRAP_E8_MAXLG_F64 11
#endif
constexpr int unwrap_elems(int lg, size_t real_size) {
  return lg <= (real_size == 8 ? GPA_UNWRAP_E8_MAXLG_F64 : GPA_UNWRAP_E8_MAXLG) ? 8 : 16;
}

// One image per call and axes up to 1024: the fused kernels are bound by their chains of dependent memory round
// trips, not by bandwidth or occupancy, and run as latency-tuned instantiations (every input requested before the
// first wait: ~30 more registers).  Stacks of frames and larger images fill the GPU and keep the lean ones
// (measured: 64 frames of 512^2 2596 -> 2475 Mpix/s and 2048^2 2565 -> 2493 with the latency-tuned kernels).
// The two kinds evaluate the same formulas; the compiler contracts multiply-adds differently in places, so results
// agree to rounding, not to the bit (GPA_NO_LAT=1 runs the lean kernels everywhere: tests use it to compare a stack
// with single calls exactly).
#ifndef GPA_UNWRAP_LAT_MAXLG
#define GPA_UNWRAP_LAT_MAXLG 10
#endif
#ifndef GPA_ROWHALF_MINLG
#define GPA_ROWHALF_MINLG 13   // rows from 2^13 points on: one row per half-length transform (gpa_unwrap_rows.hip)
#endif
#ifndef GPA_COLSTREAM_MIN
#define GPA_COLSTREAM_MIN 2048   // square images from this side on take the streamed column solve by default (2048^2: 26 -> 19 us per iteration, 3000^2: 58 -> 43; 1024^2: slower)
#endif
#define GPA_ROWPQ_MAXLG 9   // rows up to 512 pixels: row kernel and stencil in one launch (rowidct_pq_kernel)
inline bool unwrap_latency_tuned(const Impl* w, int lg) {
  return w->lat_ok && w->nprob <= 2 && lg <= GPA_UNWRAP_LAT_MAXLG;
}

// rows per thread: the f32 tile of a thread (ROWS x 4 columns) has to leave room for the double-precision recursions
// within the 128 VGPRs that 1024 threads per workgroup allow: 8 rows (32 registers); f64: 16 rows x 2 columns (64)
template <class T> struct TriRows { static constexpr int value = sizeof(T) == 4 ? 8 : 16; };
// Short columns take half as many rows per thread on twice the threads: with ~125 one-wavefront workgroups on 256
// CUs the kernel is bound by the instruction stream of a wavefront (4089 instructions at 8 rows x 4 columns, a
// quarter of them f64), not by anything the chip shares.  GPA_TRI_SMALL = largest n0 that does (diagnostic).
inline int tri_rows(size_t real_size, int n0) {
  const int base = real_size == 4 ? 8 : 16;
  if (real_size == 4 && n0 > 8192) return 2 * base;   // (1024 threads hold at most 1024 chunks)
  const int small = opt_set(OPT_TRI_SMALL) ? (int)opt(OPT_TRI_SMALL).num : 640;
  return n0 <= small ? base / 2 : base;
}

// threads side by side along a row (Q), chunks per workgroup (S, padded to whole wavefronts when there are several)
// and the rows of padding that geometry implies -- shared by the launcher and by the table builder
template <class T>
void tri_geometry(int n0, int n1, bool ragged, int R, int* Q_out, int* S_out, int* pad_out) {
  constexpr int VEC = 16 / sizeof(T);
  int S = (n0 + R - 1) / R;
  int Q = 4;
  // (workgroups wanted at least: 256 on the power-of-two path as tuned in round 2; the ragged sizes measured faster
  //  with wider workgroups down to ~100 of them -- 1000^2: Q = 2 18.6 us against 20.5 at Q = 1, 1500^2: 23.2 / 27.2)
  const int min_wgs = ragged ? 100 : 256;
  while (Q > 1 && (S * Q > 1024 || n1 / (Q * VEC) < min_wgs)) Q /= 2;
  if (opt_set(OPT_TRI_Q)) {   // diagnostic: force the number of column groups per workgroup
    const int q = (int)opt(OPT_TRI_Q).num;
    if ((q == 1 || q == 2 || q == 4) && S * q <= 1024) Q = q;
  }
  if (S * Q > 64) {
    const int cpw = 64 / Q;
    S = (S + cpw - 1) / cpw * cpw;
    while (Q > 1 && S * Q > 1024) { Q /= 2; S = ((n0 + R - 1) / R + 64 / Q - 1) / (64 / Q) * (64 / Q); }
  }
  *Q_out = Q;
  *S_out = S;
  *pad_out = S * R - n0;
}

}  // namespace gpa
