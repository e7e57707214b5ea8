// a7: DCT-Laplacian weighted least-squares phase unwrap (Ghiglia & Romero PCG),
// following phase_unwrap_prediff / phase_unwrap (phase_unwrap.py:282-350, :141-208).
//
//   r0 = A^T W^2 wrap(b);  repeat:  z = P^-1 r (DCT Poisson solve);  rho = <r,z>;
//   p = z + (rho/rho_prev) p;  q = A^T W^2 A p;  alpha = rho/<p,q>;
//   phi += alpha p;  r -= alpha q;  stop at kmax or ||r|| < eps ||r0||.
//
// The Poisson solve is three kernels: DCT-II along rows, a fused column kernel
// (DCT-II -> divide by the Laplacian eigenvalues -> DCT-III, never leaving LDS),
// DCT-III along rows.  Every scalar of the iteration (rho, alpha, beta, norms, the
// iteration counter and the stop flag) lives on the device: the host enqueues kmax
// iterations back to back and reads the iteration count once at the end; kernels
// of iterations after convergence return immediately.
//
// Power-of-two images take the fused path (run_pcg): four kernels per iteration and no vector is moved
// that does not have to be --
//   rowdct_fused : R -= alpha DCT_rows(q)   the residual is kept as its row spectrum R; ||r||^2 by Parseval
//   colsolve     : R -> Z                   column DCT-II / eigenvalue divide / DCT-III; stop test;
//                                           rho = <r,z> by Parseval from the spectra in registers
//   rowidct_p    : Z -> p = z + beta p_prev row DCT-III straight into the new search direction
//   pq           : q = A^T W^2 A p          one sliding-window stencil pass, partial <p,q>
// and phi += alpha p is applied for up to 10 iterations at once by phi_flush_kernel from the kept search
// directions.  11 array passes per iteration (+ 1.2 for the flush) instead of the 19 of the plain scheme.
// Other sizes (Bluestein DCTs) and rows that are not a multiple of 4 pixels take the plain scheme below it.
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

namespace {

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
  void* ring[10];            // search directions of the last RING iterations (fused path), grown on demand
  int nring;
  void *tw0, *tw1;           // FFT twiddles per axis
  void *wk1;                 // w_k along axis 1, natural order
  void *wk0s;                // w_k along axis 0, spectral layout
  void *ha0[2], *ham0[2];    // 1 - cos term of axis-0 bins (spectral layout); [compat]
  void *hb1[2];              // 1 - cos term of axis-1 bins (natural); [compat]
  int col_mode;              // GPA_COLSOLVE of the current solve: 0 default, 1 tri, 2 fft (read once per solve)
  void* tritab;              // TriCol per column (square images): transform-free column solve
  int triQ, triS, triR;      // its launch geometry, fixed when the table is built (the table depends on it)
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
    scal[1] = 0.0;
    flags[0] = 0;
    flags[2] = 0;
    flags[3] = 0;
    flags[1] = tot == 0.0 ? 1 : 0;
  }
  return tot != 0.0;
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

// ---------------------------------------------------------------------------
// setup: r0 = div( WW * wrap(grad) ), phi = 0, partial ||r0||^2
// ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T edge_x(const T* a, const T* w, bool from_psi, int n1, int x, int y) {
  // weighted wrapped difference across the edge (x,y)-(x,y+1); 0 outside
  if (y < 0 || y >= n1 - 1) return T(0);
  T d = from_psi ? a[(size_t)x * n1 + y + 1] - a[(size_t)x * n1 + y] : a[(size_t)x * (n1 - 1) + y];
  d = wrap_pi(d);
  if (w) {
    const T w0 = w[(size_t)x * n1 + y], w1 = w[(size_t)x * n1 + y + 1];
    const T a0 = w0 * w0, a1 = w1 * w1;
    d *= a0 < a1 ? a0 : a1;
  }
  return d;
}
template <class T>
__device__ __forceinline__ T edge_y(const T* a, const T* b, const T* w, bool from_psi, int n0, int n1, int x, int y) {
  if (x < 0 || x >= n0 - 1) return T(0);
  T d = from_psi ? a[(size_t)(x + 1) * n1 + y] - a[(size_t)x * n1 + y] : b[(size_t)x * n1 + y];
  d = wrap_pi(d);
  if (w) {
    const T w0 = w[(size_t)x * n1 + y], w1 = w[(size_t)(x + 1) * n1 + y];
    const T a0 = w0 * w0, a1 = w1 * w1;
    d *= a0 < a1 ? a0 : a1;
  }
  return d;
}

constexpr int SETUP_ROWS = 16;   // rows per workgroup band of setup_kernel

// One thread per column, sliding down a band of SETUP_ROWS rows: every weighted edge value
// is computed once (right edge and down edge of the thread's own pixel); the left edge
// comes from the neighbouring lane, the upper edge from the previous row's registers.
template <class T>
__global__ __launch_bounds__(256) void setup_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                   const T* __restrict__ w, int from_psi, int n0, int n1,
                                                   T* __restrict__ r, T* __restrict__ phi, double* part) {
  __shared__ double sh[256];
  const int y = blockIdx.x * 256 + threadIdx.x;
  const int x0 = blockIdx.y * SETUP_ROWS;
  const int x1 = x0 + SETUP_ROWS < n0 ? x0 + SETUP_ROWS : n0;
  const int lane = threadIdx.x & 63;
  double sq = 0;
  const bool act = y < n1;
  const int yc = act ? y : n1 - 1;
  auto ww = [&](int x, int yy) { const T t = w ? w[(size_t)x * n1 + yy] : T(1); return t * t; };
  T fy_up = T(0), wc = T(0);
  if (act) {
    wc = ww(x0, yc);
    if (x0 > 0) fy_up = edge_y(a, b, w, from_psi, n0, n1, x0 - 1, yc);
  }
  for (int x = x0; x < x1; ++x) {
    // own right edge (x,y)-(x,y+1) and own down edge (x,y)-(x+1,y)
    T fx = T(0), fy = T(0), wd = T(0);
    if (act) {
      if (yc + 1 < n1) {
        T d = from_psi ? a[(size_t)x * n1 + yc + 1] - a[(size_t)x * n1 + yc] : a[(size_t)x * (n1 - 1) + yc];
        d = wrap_pi(d);
        if (w) { const T wr = ww(x, yc + 1); d *= wr < wc ? wr : wc; }
        fx = d;
      }
      if (x + 1 < n0) {
        T d = from_psi ? a[(size_t)(x + 1) * n1 + yc] - a[(size_t)x * n1 + yc] : b[(size_t)x * n1 + yc];
        d = wrap_pi(d);
        wd = ww(x + 1, yc);
        if (w) d *= wd < wc ? wd : wc;
        fy = d;
      }
    }
    T fx_left = __shfl_up(fx, 1);
    if (lane == 0) fx_left = (act && yc > 0) ? edge_x(a, w, from_psi, n1, x, yc - 1) : T(0);
    if (act) {
      const T v = fx - fx_left + fy - fy_up;
      r[(size_t)x * n1 + yc] = v;
      phi[(size_t)x * n1 + yc] = T(0);
      sq += (double)v * (double)v;
    }
    fy_up = fy;
    wc = wd;
  }
  const double tot = block_sum(sq, sh);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// scalar kernels (one block each) -------------------------------------------
__global__ void scal_init_kernel(const double* part, int nparts, double* scal, int* flags) {
  part += blockIdx.z * PART_N;
  scal += blockIdx.z * SCAL_N;
  flags += blockIdx.z * FLAGS_N;
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  const double tot = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    scal[5] = tot;   // ||r0||^2
    scal[6] = tot;
    scal[7] = tot;   // smallest ||r||^2 seen
    scal[10] = tot;
    scal[11] = tot;
    scal[1] = 0.0;
    flags[0] = 0;
    flags[2] = 0;
    flags[3] = 0;
    flags[1] = tot == 0.0 ? 1 : 0;   // r == 0 everywhere: nothing to do (phase_unwrap.py:326)
  }
}
__global__ void scal_rho_kernel(const double* part, int nparts, double* scal, const int* flags) {
  if (flags[1]) return;
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  const double tot = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    scal[0] = tot;                                       // rho = <r, z>
    scal[4] = flags[0] == 0 ? 0.0 : tot / scal[1];       // beta (phase_unwrap.py:332-336)
  }
}
__global__ void scal_alpha_kernel(const double* part, int nparts, double* scal, const int* flags) {
  if (flags[1]) return;
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  const double tot = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    scal[2] = tot;                 // <p, Qp>
    scal[3] = scal[0] / tot;       // alpha (phase_unwrap.py:343)
    scal[1] = scal[0];             // rho_prev
  }
}
__global__ void scal_stop_kernel(const double* part, int nparts, double* scal, int* flags, int kmax, double eps) {
  if (flags[1]) return;
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  const double tot = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    scal[6] = tot;
    const int k = flags[0] + 1;
    flags[0] = k;
    if (k >= kmax || sqrt(tot) < eps * sqrt(scal[5]) || tot == 0.0) flags[1] = 1;   // phase_unwrap.py:348
    // breakdown guard (not in the reference, which iterates in f64 only): once the
    // residual has bottomed out at the working precision CG loses conjugacy and the
    // residual grows again; stop instead of iterating into garbage.
    if (!(tot == tot) || tot > 1e4 * scal[7]) flags[1] = 1;
    if (tot < scal[7]) scal[7] = tot;
  }
}

// elementwise / stencil kernels ------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void pupdate_kernel(const T* __restrict__ z, T* __restrict__ p, size_t count,
                                                     const double* scal, const int* flags) {
  if (flags[1]) return;
  const T beta = (T)scal[4];
  const bool first = flags[0] == 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    p[i] = first ? z[i] : z[i] + beta * p[i];
}

template <class T>
__global__ __launch_bounds__(256) void applyq_kernel(const T* __restrict__ p, const T* __restrict__ w, int n0,
                                                    int n1, T* __restrict__ q, double* part, const int* flags) {
  if (flags[1]) return;
  __shared__ double sh[256];
  const int x = blockIdx.x;
  double pq = 0;
  for (int y = threadIdx.x; y < n1; y += 256) {
    const size_t o = (size_t)x * n1 + y;
    const T pc = p[o];
    T wc = T(1);
    if (w) { wc = w[o]; wc *= wc; }
    T acc = T(0);
    // q = sum over the 4 edges of WW_edge * (p_neighbour - p_centre)   (phase_unwrap.py:118-132)
    if (y + 1 < n1) { T wn = T(1); if (w) { wn = w[o + 1]; wn *= wn; } acc += (wn < wc ? wn : wc) * (p[o + 1] - pc); }
    if (y > 0)      { T wn = T(1); if (w) { wn = w[o - 1]; wn *= wn; } acc += (wn < wc ? wn : wc) * (p[o - 1] - pc); }
    if (x + 1 < n0) { T wn = T(1); if (w) { wn = w[o + n1]; wn *= wn; } acc += (wn < wc ? wn : wc) * (p[o + n1] - pc); }
    if (x > 0)      { T wn = T(1); if (w) { wn = w[o - n1]; wn *= wn; } acc += (wn < wc ? wn : wc) * (p[o - n1] - pc); }
    q[o] = acc;
    pq += (double)pc * (double)acc;
  }
  const double tot = block_sum(pq, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// fused  p <- z + beta p  and  q = A^T W^2 A p  (phase_unwrap.py:332-342, :118-132).
// p is double-buffered (pin -> pout) so a row's neighbours can be recomputed from z and
// the OLD p while other workgroups are already writing the new one.  One workgroup per
// image row, 4 pixels per thread (16-byte accesses); partial <p, q> per row.
template <class T>
struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };
// V consecutive pixels of a row: V = 4 where rows are whole 16-byte (f32) vectors, V = 1 for any other row length
template <class T, int V>
struct alignas(V * sizeof(T)) VecN { T v[V]; };

#ifndef GPA_PQ_ROWS
#define GPA_PQ_ROWS 16
#endif
constexpr int PQ_ROWS = GPA_PQ_ROWS;   // rows per workgroup band of pq_kernel (large images; fewer for small ones)

// PGIVEN: `z` already holds the search direction p (written by rowidct_p_kernel): no combination with
// pin, no copy to pout, no beta
template <class T, bool PGIVEN = false, int V = 4>
__global__ __launch_bounds__(256) void pq_kernel(const T* __restrict__ z, const T* __restrict__ pin,
                                                T* __restrict__ pout, const T* __restrict__ w, int n0, int n1,
                                                T* __restrict__ q, double* part, double* scal,
                                                const int* flags, const double* part_rho, int nrho, int it,
                                                int band, size_t pimg = 0) {
  {
    const size_t pb = blockIdx.z;
    z += pb * pimg;
    if (pin) pin += pb * pimg;
    if (pout) pout += pb * pimg;
    if (w) w += (pb >> 1) * pimg;   // the two components of an image share its weight
    q += pb * pimg;
    part += pb * PART_N;
    scal += pb * SCAL_N;
    flags += pb * FLAGS_N;
    if (part_rho) part_rho += pb * PART_N;
  }
  if (flags[1]) return;
  __shared__ double sh[256];
  bool first;
  T beta;
  if constexpr (PGIVEN) {
    first = true;
    beta = T(0);
  } else if (it >= 0) {
    // rho = <r, z> from the producer's partial sums, beta = rho / rho_previous
    const double rho = reduce_partials(part_rho, nrho, sh);
    first = it == 0;                                 // first iteration: p = z (pin is uninitialised)
    beta = first ? T(0) : (T)(rho / scal[8 + ((it - 1) & 1)]);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  } else {
    first = flags[0] == 0;
    beta = first ? T(0) : (T)scal[4];
  }
  auto comb = [&](T zv, T pv) { return first ? zv : zv + beta * pv; };
  // a workgroup owns a band of PQ_ROWS rows x 1024 columns and slides down it with the
  // previous / current / next row in registers: every row of z, p, w is read once
  // (plus a 2-row halo per band) instead of three times by three different workgroups.
  const int y0 = (blockIdx.x * 256 + threadIdx.x) * V;
  const int x0 = blockIdx.y * band;
  const int x1 = x0 + band < n0 ? x0 + band : n0;
  double pq = 0;
  if (y0 < n1) {
    auto load_p = [&](int x, VecN<T, V>& out) {
      const size_t o = (size_t)x * n1 + y0;
      const VecN<T, V> a = *reinterpret_cast<const VecN<T, V>*>(z + o);
      if (first) { out = a; return; }
      const VecN<T, V> b = *reinterpret_cast<const VecN<T, V>*>(pin + o);
#pragma unroll
      for (int j = 0; j < V; ++j) out.v[j] = a.v[j] + beta * b.v[j];
    };
    auto load_w = [&](int x, VecN<T, V>& out) {
      if (!w) {
#pragma unroll
        for (int j = 0; j < V; ++j) out.v[j] = T(1);
        return;
      }
      out = *reinterpret_cast<const VecN<T, V>*>(w + (size_t)x * n1 + y0);
#pragma unroll
      for (int j = 0; j < V; ++j) out.v[j] *= out.v[j];
    };
    const bool hasl = y0 > 0, hasr = y0 + V < n1;
    VecN<T, V> pu, pc, pd, wu, wc, wd;
#pragma unroll
    for (int j = 0; j < V; ++j) pu.v[j] = wu.v[j] = T(0);
    if (x0 > 0) { load_p(x0 - 1, pu); load_w(x0 - 1, wu); }
    load_p(x0, pc);
    load_w(x0, wc);
    for (int x = x0; x < x1; ++x) {
      const bool up = x > 0, dn = x + 1 < n0;
      if (dn) { load_p(x + 1, pd); load_w(x + 1, wd); }
      const size_t o = (size_t)x * n1 + y0;
      // left / right neighbours come from the adjacent lanes' registers; only the two
      // lanes at the ends of a wavefront have to go to memory
      const int lane = threadIdx.x & 63;
      T pl = __shfl_up(pc.v[V - 1], 1), pr = __shfl_down(pc.v[0], 1);
      T wl = __shfl_up(wc.v[V - 1], 1), wr = __shfl_down(wc.v[0], 1);
      if (lane == 0 && hasl) {
        if constexpr (PGIVEN) pl = z[o - 1]; else pl = comb(z[o - 1], pin[o - 1]);
        wl = T(1);
        if (w) { wl = w[o - 1]; wl *= wl; }
      }
      if (lane == 63 && hasr) {
        if constexpr (PGIVEN) pr = z[o + V]; else pr = comb(z[o + V], pin[o + V]);
        wr = T(1);
        if (w) { wr = w[o + V]; wr *= wr; }
      }
      VecN<T, V> qv;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const T c = pc.v[j], wj = wc.v[j];
        T acc = T(0);
        // q = sum over the 4 edges of min(w^2, w_nb^2) * (p_nb - p)   (phase_unwrap.py:118-132)
        if (j < V - 1) { const T wn = wc.v[j + 1]; acc += (wn < wj ? wn : wj) * (pc.v[j + 1] - c); }
        else if (hasr) acc += (wr < wj ? wr : wj) * (pr - c);
        if (j > 0) { const T wn = wc.v[j - 1]; acc += (wn < wj ? wn : wj) * (pc.v[j - 1] - c); }
        else if (hasl) acc += (wl < wj ? wl : wj) * (pl - c);
        if (dn) { const T wn = wd.v[j]; acc += (wn < wj ? wn : wj) * (pd.v[j] - c); }
        if (up) { const T wn = wu.v[j]; acc += (wn < wj ? wn : wj) * (pu.v[j] - c); }
        qv.v[j] = acc;
        pq += (double)c * (double)acc;
      }
      if constexpr (!PGIVEN) *reinterpret_cast<VecN<T, V>*>(pout + o) = pc;
      *reinterpret_cast<VecN<T, V>*>(q + o) = qv;
      pu = pc; wu = wc;
      pc = pd; wc = wd;
    }
  }
  const double tot = block_sum(pq, sh);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// The same stencil for bands of a few rows (images up to 2048^2, where pq_kernel's band is 4 rows): every row the
// band touches -- its BAND rows, one above, one below, and the left / right neighbour pixels of the thread's
// vector -- is requested before anything waits, so the kernel pays ONE memory round trip instead of one per row
// of the sliding window (at 512^2 the kernel is nothing but its latency chain: 5.2 us).  Same arithmetic in the
// same order as pq_kernel<T, true, V>: results and partial sums are bit-identical.
template <class T, int V, int BAND>
__global__ __launch_bounds__(256) void pq_small_kernel(const T* __restrict__ p, const T* __restrict__ w, int n0, int n1,
                                                      T* __restrict__ q, double* part, const int* flags, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    p += pb * pimg;
    if (w) w += (pb >> 1) * pimg;   // the two components of an image share its weight
    q += pb * pimg;
    part += pb * PART_N;
    flags += pb * FLAGS_N;
  }
  __shared__ double sh[256];
  const int stop = flags[1];
  const int y0 = (blockIdx.x * 256 + threadIdx.x) * V;
  const int x0 = blockIdx.y * BAND;
  const bool act = y0 < n1;
  const int yc = act ? y0 : 0;
  const bool hasl = act && y0 > 0, hasr = act && y0 + V < n1;   // (idle threads read their clamped addresses only)
  VecN<T, V> pr[BAND + 2], wr[BAND + 2];
  T pl[BAND], prr[BAND], wl[BAND], wrr[BAND];
#pragma unroll
  for (int r = 0; r < BAND + 2; ++r) {
    int x = x0 - 1 + r;
    x = x < 0 ? 0 : (x > n0 - 1 ? n0 - 1 : x);   // rows outside the image are loaded from a clamped address and not used
    pr[r] = *reinterpret_cast<const VecN<T, V>*>(p + (size_t)x * n1 + yc);
  }
#pragma unroll
  for (int r = 0; r < BAND; ++r) {
    int x = x0 + r;
    x = x > n0 - 1 ? n0 - 1 : x;
    pl[r] = p[(size_t)x * n1 + (hasl ? yc - 1 : yc)];
    prr[r] = p[(size_t)x * n1 + (hasr ? yc + V : yc)];
  }
  if (w) {
#pragma unroll
    for (int r = 0; r < BAND + 2; ++r) {
      int x = x0 - 1 + r;
      x = x < 0 ? 0 : (x > n0 - 1 ? n0 - 1 : x);
      wr[r] = *reinterpret_cast<const VecN<T, V>*>(w + (size_t)x * n1 + yc);
    }
#pragma unroll
    for (int r = 0; r < BAND; ++r) {
      int x = x0 + r;
      x = x > n0 - 1 ? n0 - 1 : x;
      wl[r] = w[(size_t)x * n1 + (hasl ? yc - 1 : yc)];
      wrr[r] = w[(size_t)x * n1 + (hasr ? yc + V : yc)];
    }
  } else {
#pragma unroll
    for (int r = 0; r < BAND + 2; ++r)
#pragma unroll
      for (int j = 0; j < V; ++j) wr[r].v[j] = T(1);
#pragma unroll
    for (int r = 0; r < BAND; ++r) wl[r] = wrr[r] = T(1);
  }
  if (stop) return;
#pragma unroll
  for (int r = 0; r < BAND + 2; ++r)
#pragma unroll
    for (int j = 0; j < V; ++j) wr[r].v[j] *= wr[r].v[j];
#pragma unroll
  for (int r = 0; r < BAND; ++r) { wl[r] *= wl[r]; wrr[r] *= wrr[r]; }
  double pq = 0;
#pragma unroll
  for (int r = 0; r < BAND; ++r) {
    const int x = x0 + r;
    if (!act || x >= n0) continue;
    const bool up = x > 0, dn = x + 1 < n0;
    const VecN<T, V>&pc = pr[r + 1], &wc = wr[r + 1];
    VecN<T, V> qv;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const T c = pc.v[j], wj = wc.v[j];
      T acc = T(0);
      // q = sum over the 4 edges of min(w^2, w_nb^2) * (p_nb - p)   (phase_unwrap.py:118-132)
      if (j < V - 1) { const T wn = wc.v[j + 1]; acc += (wn < wj ? wn : wj) * (pc.v[j + 1] - c); }
      else if (hasr) acc += (wrr[r] < wj ? wrr[r] : wj) * (prr[r] - c);
      if (j > 0) { const T wn = wc.v[j - 1]; acc += (wn < wj ? wn : wj) * (pc.v[j - 1] - c); }
      else if (hasl) acc += (wl[r] < wj ? wl[r] : wj) * (pl[r] - c);
      if (dn) { const T wn = wr[r + 2].v[j]; acc += (wn < wj ? wn : wj) * (pr[r + 2].v[j] - c); }
      if (up) { const T wn = wr[r].v[j]; acc += (wn < wj ? wn : wj) * (pr[r].v[j] - c); }
      qv.v[j] = acc;
      pq += (double)c * (double)acc;
    }
    *reinterpret_cast<VecN<T, V>*>(q + (size_t)x * n1 + y0) = qv;
  }
  const double tot = block_sum(pq, sh);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

template <class T>
__global__ __launch_bounds__(256) void update_kernel(const T* __restrict__ p, const T* __restrict__ q,
                                                    T* __restrict__ phi, T* __restrict__ r, size_t count,
                                                    const double* scal, double* part, const int* flags) {
  if (flags[1]) return;
  __shared__ double sh[256];
  const T alpha = (T)scal[3];
  double sq = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    phi[i] += alpha * p[i];
    const T rv = r[i] - alpha * q[i];
    r[i] = rv;
    sq += (double)rv * (double)rv;
  }
  const double tot = block_sum(sq, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// DCT kernels
// ---------------------------------------------------------------------------
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

#ifndef GPA_ROW_TWLDS
#define GPA_ROW_TWLDS 1   // 16-element three-pass row transforms: pass-1 base twiddles from a small LDS table (12 VGPRs less in f32)
#endif
template <class T, int LG, bool LAT = false>
struct RowGeom {
  using F = WgFFT<T, LG, unwrap_elems(LG, sizeof(T))>;
  static constexpr bool TWLDS = GPA_ROW_TWLDS && F::E == 16 && F::P == 3;
  using TW = typename std::conditional<TWLDS, typename F::TwiddlesP1Lds, typename F::Twiddles>::type;
  static constexpr int T1N = TWLDS ? F::P1_SETS * 6 : 1;
  using D = WgDCT<T, LG, unwrap_elems(LG, sizeof(T))>;
  // threads per workgroup: 256; the latency-tuned kernels of ONE image with rows up to 512 pixels take 128 (twice the
  // workgroups on a GPU that such an image leaves mostly empty: 512^2 893 -> 935 Mpix/s; stacks prefer 256)
#ifndef GPA_ROW_THREADS_LAT
#define GPA_ROW_THREADS_LAT 128
#endif
#ifndef GPA_ROW_THREADS
#define GPA_ROW_THREADS 256
#endif
  // (rows up to 256 pixels: one wavefront per workgroup, 256^2 280 -> 290 Mpix/s; at 512 that loses 8 %)
  static constexpr int WGT = (LAT && LG <= 8) ? 64 : (LAT && LG == 9) ? GPA_ROW_THREADS_LAT : GPA_ROW_THREADS;
  static constexpr int NF = F::TPF >= WGT ? 1 : WGT / F::TPF;   // row PAIRS per workgroup
  static constexpr int RS = F::LDS_ELEMS + (NF > 1 ? (F::TPF < 32 ? F::TPF : 0) : 0);
  static constexpr int THREADS = NF * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NF * RS * sizeof(cpx<T>);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024;
};
template <class T, int LG, bool LAT = false>
struct ColGeom {
  using F = WgFFT<T, LG, unwrap_elems(LG, sizeof(T))>;
  using D = WgDCT<T, LG, unwrap_elems(LG, sizeof(T))>;
  static constexpr int cols() {
    // as many column pairs as LDS and 1024 threads allow (wide tiles = long row segments) ...
    int c = 16;
    while (c > 1 && (c * F::TPF > 1024 || (size_t)c * (F::LDS_ELEMS + 32) * sizeof(cpx<T>) > 160 * 1024)) c /= 2;
    // ... but a (square) image of side N has only N/2 pairs: keep >= 512 workgroups in flight
    // on the 256 CUs, small images are cache resident and do not care about segment length
#ifndef GPA_COL_WANT
#define GPA_COL_WANT 16
#endif
#ifndef GPA_COL_WANT11
#define GPA_COL_WANT11 2
#endif
#ifndef GPA_COL_WANT10
#define GPA_COL_WANT10 4   // 1024-point columns: 4 pairs (32-byte row segments): single image 1600 -> 1700 Mpix/s; 2 -> 1660, 8 -> 1630
#endif
#ifndef GPA_COL_WANT9
#define GPA_COL_WANT9 4   // 512-point columns in stacks (lean kernels): 64 frames 2580 -> 2830 Mpix/s; a single image (LAT) keeps 1 (879 against 862)
#endif
    const int want = LG >= 12 ? GPA_COL_WANT : (LG == 11 ? GPA_COL_WANT11 : (LG == 10 ? GPA_COL_WANT10 : (LG == 9 && !LAT ? GPA_COL_WANT9 : 1)));
    return c < want ? c : want;
  }
  static constexpr int CC = cols();   // packed column PAIRS (complex transforms) per workgroup
  // as in pass A of the sweep: two transforms per f32 thread (adjacent pairs = 4 real
  // columns = one 16-byte access per row), the CT pairs that sit side by side in the
  // thread index interleaved element by element in LDS
#ifndef GPA_COL_NT
#define GPA_COL_NT 2
#endif
  static constexpr int NT = (sizeof(T) == 4 && CC >= 2) ? GPA_COL_NT : 1;
  static constexpr int CT = CC / NT;
  static constexpr int REGION = CT * F::LDS_ELEMS;
  static constexpr int THREADS = CT * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NT * REGION * sizeof(cpx<T>);
  static constexpr bool FITS = (size_t)(F::LDS_ELEMS + 32) * sizeof(cpx<T>) <= 160 * 1024;
};

// fused path: apply the pending update of the previous iteration (alpha from the pq
// kernel's partial sums), then DCT-II along axis 1 of the new residual
//   r -= alpha q;  phi += alpha p;  partial ||r||^2;  Z = DCT(r)
#ifndef GPA_DCTF_WAVES
#define GPA_DCTF_WAVES 1
#endif
#ifndef GPA_EARLY16
#define GPA_EARLY16 0   // experiment: the every-input-first kernel variants also for 16-element transforms
#endif
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64 row kernels: 2 waves/SIMD (256 VGPRs) beat 1 wave with AGPR spill-over
#endif
// LAT: the latency-tuned variant (one image per call, axes up to 1024) -- same arithmetic, same results
template <class T, int LG, bool LAT = false>
__global__ __launch_bounds__((RowGeom<T, LG, LAT>::THREADS), (sizeof(T) == 8 ? GPA_F64_WAVES : GPA_DCTF_WAVES)) void rowdct_fused_kernel(
    T* __restrict__ r, const T* __restrict__ q, int n0, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ wk, int* flags, const double* part_pq, int npq, double* part_norm,
    double* scal, int it, int ring, int init, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    r += pb * pimg;
    q += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_pq += pb * PART_N;
    part_norm += pb * PART_N;
  }
  // The fused iteration keeps the residual as its row spectrum R = DCT-II_rows(r) (the only consumers of r
  // are this transform, ||r|| and <r,z>, and the last two follow from the spectra by Parseval):
  //   it == 0: r (spatial, from the set-up) -> R, in place;
  //   it  > 0: R -= alpha DCT-II_rows(q)    (linearity; phase_unwrap.py:345), partial ||r||^2 from R.
  // So the update reads q and R and writes R: three arrays instead of r, q in and r, Z out.
  using G = RowGeom<T, LG, LAT>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = F::TPF, N = F::L, E = F::E;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowGeom<T, LG, LAT>::THREADS];
  // init (first iteration of a solve on prepared residuals): part_pq / npq are the producer's partial norms of r0
  // EARLY (short transforms): every input of an update -- flags, q, the kept spectrum, w_k, partial sums, rho -- is
  // requested before anything waits, so the kernel pays one memory round trip instead of five in a row
  constexpr bool EARLY = LAT && (E == 8 || GPA_EARLY16);
  const bool early = EARLY && it > 0;
  int stop = 0;
  if (init) { if (!solve_init(part_pq, npq, scal, flags, sh)) return; }
  else if (early) stop = flags[1];
  else if (flags[1]) return;
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const bool valid = 2 * pr + 1 < n0;
  const size_t oa = (size_t)(valid ? 2 * pr : 0) * N, ob = oa + N;
  // (register twiddles by default: the LDS table of rowidct_p_kernel made this kernel's allocation worse, 156 -> 160 VGPRs)
#ifndef GPA_DCTF_TWLDS
#define GPA_DCTF_TWLDS 0
#endif
  constexpr bool TWL = GPA_DCTF_TWLDS && G::TWLDS;
  typename std::conditional<TWL, typename F::TwiddlesP1Lds, typename F::Twiddles>::type tw;
  __shared__ cpx<T> t1s[TWL ? G::T1N : 1];
  if constexpr (TWL) {
    F::fill_pass1_table(t1s, twtab, threadIdx.x, G::THREADS);
    __syncthreads();
    F::load_twiddles(tw, twtab, tid, t1s);
  } else {
    F::load_twiddles(tw, twtab, tid);
  }
  cpx<T> x[E];
  cpx<T> rk[E];
  cpx<T> wkv[EARLY ? E : 1];
  T alpha = T(0);
  if (early) {
    // up to 1024 points the even/odd-permuted DCT input is fetched directly (stride-2 accesses: these sizes are
    // latency-, not bandwidth-bound, and the detour through LDS costs two barriers); 2048 points: 16-byte loads
    constexpr bool DIRECTQ = LG <= 10;
    constexpr int NQ = N / (4 * TPF);   // 16-byte vectors of q per thread and row
    Vec4<T> qa[NQ], qb[NQ];
    if constexpr (DIRECTQ) {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int src = makhoul_src(tid + TPF * i, N);
        x[i] = {q[oa + src], q[ob + src]};
      }
    } else {
#pragma unroll
      for (int v = 0; v < NQ; ++v) {
        const int c0 = 4 * (tid + TPF * v);
        qa[v] = *reinterpret_cast<const Vec4<T>*>(q + oa + c0);
        qb[v] = *reinterpret_cast<const Vec4<T>*>(q + ob + c0);
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      rk[i] = {r[oa + tid + TPF * i], r[ob + tid + TPF * i]};
      wkv[EARLY ? i : 0] = wk[tid + TPF * i];
    }
    const double pq_part = load_partials(part_pq, npq);
    const double rho = scal[8 + ((it - 1) & 1)];
    if (stop) return;
    const double pq = block_sum(pq_part, sh);
    const double alpha_d = rho / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
    if constexpr (!DIRECTQ) {
#pragma unroll
      for (int v = 0; v < NQ; ++v) {
        const int c0 = 4 * (tid + TPF * v);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[F::pad(c0 + j)] = {qa[v].v[j], qb[v].v[j]};
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < E; ++i) x[i] = lds[F::pad(makhoul_src(tid + TPF * i, N))];
      __syncthreads();
    }
  } else if (it > 0) {
    const double pq = reduce_partials(part_pq, npq, sh);
    const double alpha_d = scal[8 + ((it - 1) & 1)] / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    // phi += alpha p is not applied here: alpha is filed for phi_flush_kernel, which adds the kept
    // search directions of up to `ring` iterations in one pass over phi
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
    // q comes in with coalesced 16-byte accesses and is parked in LDS, so that the even/odd-permuted
    // DCT input does not have to be fetched with stride-2 accesses
    for (int c0 = 4 * tid; c0 < N; c0 += 4 * TPF) {
      const Vec4<T> qa = *reinterpret_cast<const Vec4<T>*>(q + oa + c0), qb = *reinterpret_cast<const Vec4<T>*>(q + ob + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[F::pad(c0 + j)] = {qa.v[j], qb.v[j]};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = lds[F::pad(makhoul_src(tid + TPF * i, N))];
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int src = makhoul_src(tid + TPF * i, N);
      x[i] = {r[oa + src], r[ob + src]};
    }
    __syncthreads();   // in place: every sample of the two rows is in registers before any bin is written
  }
#ifndef GPA_DCTF_LATE_RK
#define GPA_DCTF_LATE_RK 0
#endif
  // the kept spectrum is requested before the transform so that its latency hides behind it
  // (GPA_DCTF_LATE_RK: after it instead -- 32 registers less across the transform, one more wave per SIMD)
  if (it > 0 && !early && !GPA_DCTF_LATE_RK) {
#pragma unroll
    for (int i = 0; i < E; ++i) rk[i] = {r[oa + tid + TPF * i], r[ob + tid + TPF * i]};
  }
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::fwd_scatter(x, lds, tid);
  __syncthreads();
  if constexpr (EARLY) { if (early) D::fwd_gather(x, lds, tid, wkv); else D::fwd_gather(x, lds, tid, wk); }
  else D::fwd_gather(x, lds, tid, wk);
  if (it > 0 && GPA_DCTF_LATE_RK) {
#pragma unroll
    for (int i = 0; i < E; ++i) rk[i] = {r[oa + tid + TPF * i], r[ob + tid + TPF * i]};
  }
  double sq = 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    T ra = x[i].x, rb = x[i].y;
    if (it > 0) {
      ra = rk[i].x - alpha * ra;
      rb = rk[i].y - alpha * rb;
      // sum_n r^2 = (1 / 2N) sum_k c_k R_k^2, c_0 = 1/2 (SciPy's unnormalised DCT-II)
      const double t = (double)ra * (double)ra + (double)rb * (double)rb;
      sq += k == 0 ? 0.5 * t : t;
    }
    if (valid) {
      r[oa + k] = ra;
      r[ob + k] = rb;
    }
  }
  if (it > 0) {
    if (!valid) sq = 0;
    const double tot = block_sum(sq, sh);
    if (threadIdx.x == 0) part_norm[blockIdx.x] = tot / (2.0 * N);
  }
}

// phi += sum_j alpha_j p_j over the updates j in [flags[2], flags[0]) that the iteration has completed
// but phi has not seen yet, in iteration order (the same additions the reference makes one per
// iteration, phase_unwrap.py:344, without writing phi back in between).  Runs whether or not the
// iteration has stopped; phi_commit_kernel then records what was applied.
template <class T> struct RingPtrs { const T* p[RING_MAX]; };
template <class T, int V = 4>
__global__ __launch_bounds__(256) void phi_flush_kernel(RingPtrs<T> ringp, int ring, T* __restrict__ phi, size_t count4,
                                                       const double* __restrict__ scal, int* __restrict__ flags,
                                                       int init, size_t pimg, int final_it, const double* part_pq,
                                                       int npq) {
  const size_t pb = blockIdx.z;
  phi += pb * pimg;
  scal += pb * SCAL_N;
  flags += pb * FLAGS_N;
  // init: phi has not been written yet (prepared start) -- this flush starts from 0 instead of reading it
  const int a = flags[2];
  int b = flags[0];
  // final_it = kmax: the flush that ends the solve.  If the iteration has not stopped by itself, the step length
  // of its last update (no further row kernel computes it) is evaluated here, by every workgroup, from the stencil
  // kernel's partial sums -- and nothing the other workgroups read is written: the iteration count goes to
  // flags[3].  (This used to take two more one-block kernels and a commit.)
  int jlast = -1;
  double alpha_last = 0.0;
  if (final_it > 0) {
    __shared__ double sh[256];
    if (!flags[1]) {
      const double pq = reduce_partials(part_pq + pb * PART_N, npq, sh);
      alpha_last = scal[8 + ((final_it - 1) & 1)] / pq;   // phase_unwrap.py:343
      jlast = final_it - 1;
      b = final_it;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[3] = b;
  }
  if (a >= b && !init) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (size_t)gridDim.x * 256) {
    VecN<T, V> f;
#pragma unroll
    for (int c = 0; c < V; ++c) f.v[c] = T(0);
    if (!init) f = reinterpret_cast<const VecN<T, V>*>(phi)[i];
    for (int j = a; j < b; ++j) {
      const T alpha = j == jlast ? (T)alpha_last : (T)scal[SC_ALPHA + j % ring];
      const VecN<T, V> pv = reinterpret_cast<const VecN<T, V>*>(ringp.p[j % ring] + pb * pimg)[i];
#pragma unroll
      for (int c = 0; c < V; ++c) f.v[c] += alpha * pv.v[c];
    }
    reinterpret_cast<VecN<T, V>*>(phi)[i] = f;
  }
}
__global__ void phi_commit_kernel(int* flags) {
  flags += blockIdx.z * FLAGS_N;
  flags[2] = flags[0];
}

// columns: Z -> DCT-II along axis 0, divide by eigenvalues, DCT-III along axis 0 (in place)
template <class T, int LG, bool RHO, bool LAT = false>
__global__ __launch_bounds__((ColGeom<T, LG, LAT>::THREADS)) void colsolve_kernel(T* __restrict__ Z, int n1,
                                                                          const cpx<T>* __restrict__ twtab,
                                                                          const cpx<T>* __restrict__ wspec,
                                                                          const T* __restrict__ ha,
                                                                          const T* __restrict__ ham,
                                                                          const T* __restrict__ hb,
                                                                          int* flags, const double* part_norm,
                                                                          int nnorm, int it, double eps,
                                                                          double* scal, double* part_rho,
                                                                          const T* __restrict__ Zin, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    if (Zin) Zin += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_norm += pb * PART_N;
    part_rho += pb * PART_N;
  }
  using G = ColGeom<T, LG, LAT>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int E = F::E;
  // EARLY (short transforms): flags, tile, partial sums and scalars are requested together, the early exit and the
  // stopping test come after that single round trip (see rowdct_fused_kernel)
  constexpr bool EARLY = LAT && E == 8;
  const int stopped = flags[1];
  if (!EARLY && stopped) return;
  const T* Zsrc = Zin ? Zin : Z;   // fused path: reads the kept row spectrum of r, writes the solve to Z
  constexpr int TPF = F::TPF, N = F::L, CT = G::CT, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = threadIdx.x % CT, t = threadIdx.x / CT;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + c;
  // XCD-aware tile order (see passA_kernel): neighbouring column tiles meet in one L2
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int y0 = (tile * G::CC + c * NT) * 2;       // first real column of this thread
  const bool valid = y0 + 2 * NT - 1 < n1;          // n1 is a power of two >= 64: tiles are never ragged
  const int yy = valid ? y0 : 0;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, t);
  cpx<T> x[NT][E];
  struct alignas(NT * sizeof(cpx<T>)) Vec { cpx<T> v[NT]; };
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int row = makhoul_src(t + TPF * i, N);
    const Vec q = *reinterpret_cast<const Vec*>(Zsrc + (size_t)row * n1 + yy);
#pragma unroll
    for (int n = 0; n < NT; ++n) x[n][i] = q.v[n];
  }
  __shared__ double shn[ColGeom<T, LG, LAT>::THREADS];
  // (short transforms: the solve's tables too)
  typename D::SolveTables stb;
  T hbv[NT][2];
  if constexpr (EARLY) {
    D::load_solve_tables(stb, t, wspec, ha, ham);
#pragma unroll
    for (int n = 0; n < NT; ++n) { hbv[n][0] = hb[yy + 2 * n]; hbv[n][1] = hb[yy + 2 * n + 1]; }
  }
  double norm_part = 0, best = 0, norm0 = 0;
  if (it > 0) {
    norm_part = load_partials(part_norm, nnorm);
    best = scal[10 + ((it - 1) & 1)];
    norm0 = scal[5];
  }
  if (EARLY && stopped) return;
  if (it > 0) {
    // (placed after the tile loads have been issued so its latency hides behind them)
    // fused path: the update of iteration it-1 was applied by this iteration's row kernel;
    // every workgroup evaluates the reference's stopping test (phase_unwrap.py:348) on it
    const double tot = block_sum(norm_part, shn);
    const bool stop = sqrt(tot) < eps * sqrt(norm0) || tot == 0.0 || !(tot == tot) || tot > 1e4 * best;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[0] = it;                                   // updates completed
      scal[6] = tot;
      scal[10 + (it & 1)] = tot < best ? tot : best;
      if (stop) flags[1] = 1;
    }
    if (stop) return;
  }
  F::template forward_multi<NT, CT>(x, lds, G::REGION, t, tw);
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NT; ++n) D::template solve_scatter<CT>(x[n], lds + n * G::REGION, t);
  __syncthreads();
  // fused path: rho = <r, z> of the whole image from the spectra in registers (Parseval), so that the
  // row kernel that follows need not read r again
  double rho = 0.0;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if constexpr (EARLY)
      D::template solve_combine<CT>(x[n], lds + n * G::REGION, t, stb, hbv[n][0], hbv[n][1], yy + 2 * n == 0, false,
                                    T(1) / T(N), RHO ? &rho : nullptr);
    else
      D::template solve_combine<CT>(x[n], lds + n * G::REGION, t, wspec, ha, ham, hb[yy + 2 * n], hb[yy + 2 * n + 1],
                                    yy + 2 * n == 0, false, T(1) / T(N), RHO ? &rho : nullptr);
  }
  __syncthreads();
  // parked in (static) LDS; reduced after the stores, where no transform data is live any more
  if constexpr (RHO) shn[threadIdx.x] = valid ? rho : 0.0;
  // the inverse exchanges through the same LDS addresses as the forward transform: recomputed from an
  // opaque copy of t instead of being kept alive (or spilled) across the solve
  int ti = t;
  asm volatile("" : "+v"(ti));
  F::template inverse_multi<NT, CT>(x, lds, G::REGION, ti, tw);
  if (valid) {
    // the store addresses equal the load addresses; recomputed from an opaque copy of t so that the
    // compiler does not keep 16 64-bit addresses alive (or spilled) across the transforms
    int ts = t;
    asm volatile("" : "+v"(ts));
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int row = makhoul_src(ts + TPF * i, N);
      Vec q;
#pragma unroll
      for (int n = 0; n < NT; ++n) q.v[n] = x[n][i];
      *reinterpret_cast<Vec*>(Z + (size_t)row * n1 + y0) = q;
    }
  }
  if constexpr (RHO) {
    __syncthreads();
    if (threadIdx.x < 64) {
      double a = 0.0;
      for (int i = threadIdx.x; i < G::THREADS; i += 64) a += shn[i];
      a = wave_sum(a);
      if (threadIdx.x == 0) part_rho[blockIdx.x] = a / (double)n1;
    }
  }
}

// ---------------------------------------------------------------------------
// columns without a transform (square images): for row frequency j the column solve
//     z = C^-1 diag(1 / (lambda_k + mu_j)) C r ,  lambda_k = 2 cos(pi k / N) - 2,  mu_j = 2 cos(pi j / N) - 2
// (C = DCT-II along the column) is the solution of the tridiagonal system (T + mu_j) z = r with T the second
// difference matrix with reflecting ends -- the DCT-II basis diagonalises exactly that matrix.  Its Green's
// function is the two-sided exponential -lam^(|n|+1) / (1 - lam^2), lam = (1 + h) - sqrt(h (2 + h)), h = 1 - cos(pi j / N),
// on the half-sample symmetric extension of r, i.e. the cascade of a causal and an anticausal first-order recursion
//     p_n = r_n + lam p_(n-1)        with p_(-1) = (A + lam^N B) / (1 - lam^(2N)),  A = sum lam^m r_m,  B = sum lam^m r_(N-1-m)
//     z_n = -lam p_n + lam z_(n+1)   with z_N = -lam / (1 - lam) p_(N-1)
// -- 5 multiply-adds per sample instead of two 4096-point FFTs, so the kernel is a pure stream.  A thread owns
// ROWS consecutive rows of VEC adjacent columns (16-byte accesses; Q threads side by side cover Q * VEC columns
// = 64 bytes of a row at 4096^2 f32); the recursions run in double (f32 data: error 1e-8, below an f32 FFT's),
// chunk carries are combined by a scan through LDS.  Column j = 0 (mu = 0, the row means) is the singular one:
// the reference divides its DC bin by 1 (phase_unwrap.py:110-114), i.e. z = T^+ (r - mean) + mean, which is the same
// recursion with lam = 1 on r - mean followed by the removal of the mean of z.
// rho = <r, z> follows from z alone: z'(T + mu) z = -sum (z_(n+1) - z_n)^2 + mu sum z_n^2, no cancellation.
// ---------------------------------------------------------------------------
struct TriCol {
  double lam, lamR, lamN, inv, zn;   // lam^ROWS, lam^N, 1 / (1 - lam^(2N)), -lam / (1 - lam); column 0: 1, 1, 1, 0, 0
};
// rows per thread: the f32 tile of a thread (ROWS x 4 columns) has to leave room for the double-precision recursions
// within the 128 VGPRs that 1024 threads per workgroup allow: 8 rows (32 registers); f64: 16 rows x 2 columns (64)
template <class T> struct TriRows { static constexpr int value = sizeof(T) == 4 ? 8 : 16; };
// Short columns take half as many rows per thread on twice the threads: with ~125 one-wavefront workgroups on 256
// CUs the kernel is bound by the instruction stream of a wavefront (4089 instructions at 8 rows x 4 columns, a
// quarter of them f64), not by anything the chip shares.  GPA_TRI_SMALL = largest n0 that does (diagnostic).
inline int tri_rows(size_t real_size, int n0) {
  const int base = real_size == 4 ? 8 : 16;
  if (real_size == 4 && n0 > 8192) return 2 * base;   // (1024 threads hold at most 1024 chunks)
  static const int small = getenv("GPA_TRI_SMALL") ? atoi(getenv("GPA_TRI_SMALL")) : 640;
  return n0 <= small ? base / 2 : base;
}

// scan x_s = v_s + m x_(s-1) over the S chunks of every column (REVERSE: from the last chunk down).  Threads are
// laid out chunk-major with Q threads side by side, so a wavefront holds 64 / Q consecutive chunks of its Q column
// groups: the scan runs inside the wavefront with lane shuffles (log2(64 / Q) steps), the wavefronts' totals are
// chained through LDS (one barrier).  excl = x of the previous chunk in scan order (0 for the first), total = x of
// the last one.  lds: >= 16 * Q * VEC doubles.
template <int NV, int VEC, int Q, bool REVERSE>
__device__ __forceinline__ void chunk_scan(const double (&v)[NV], const double (&m)[NV], double (&excl)[NV],
                                           double (&total)[NV], double* lds, int col) {
  constexpr int CPW = 64 / Q;                       // chunks per wavefront
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6, q = lane % Q;
  const int active = blockDim.x < 64 ? blockDim.x / Q : CPW;   // chunks in this wavefront (tiny images: fewer)
  const int sl = REVERSE ? active - 1 - lane / Q : lane / Q;   // position of this chunk in scan order inside the wavefront
  const int wo = REVERSE ? nw - 1 - wave : wave;               // position of the wavefront in scan order
  // ---- level 1: inside the wavefront, by lane shuffles; pw collects m^(sl + 1) from the squared multipliers
  double cur[NV], mp[NV], pw[NV];
#pragma unroll
  for (int a = 0; a < NV; ++a) { cur[a] = v[a]; mp[a] = m[a]; pw[a] = 1.0; }
#pragma unroll
  for (int off = 1; off < CPW; off <<= 1) {
#pragma unroll
    for (int a = 0; a < NV; ++a) {
      const double t = REVERSE ? __shfl_down(cur[a], off * Q) : __shfl_up(cur[a], off * Q);
      if (sl >= off) cur[a] += mp[a] * t;
      if ((sl + 1) & off) pw[a] *= mp[a];
      mp[a] *= mp[a];
    }
  }
#pragma unroll
  for (int a = 0; a < NV; ++a)
    if (sl + 1 == CPW) pw[a] = mp[a];
  // mp = m^CPW: one wavefront's worth of chunks.  (active < CPW only when there is a single wavefront.)
  __syncthreads();   // lds may still be read from an earlier scan
  if (sl == active - 1) {
#pragma unroll
    for (int a = 0; a < NV; ++a) lds[(size_t)wo * (Q * VEC) + col + a] = cur[a];
  }
  __syncthreads();
  // ---- level 2: the (at most 16) wavefront totals, scanned by every wavefront for itself: the lane group that
  // holds chunk g (mod 16) of the wavefront takes total g, four shuffle steps chain them, two shuffles fetch the
  // value in front of this wavefront and the grand total
  const int g = (lane / Q) & 15;
#pragma unroll
  for (int a = 0; a < NV; ++a) {
    double t16 = g < nw ? lds[(size_t)g * (Q * VEC) + col + a] : 0.0;
    double mq = mp[a];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const double up = __shfl_up(t16, off * Q);
      if (g >= off) t16 += mq * up;
      mq *= mq;
    }
    const int gq = q;                                       // lane of group 0 that works on this thread's columns
    const double acc = __shfl(t16, (wo > 0 ? wo - 1 : 0) * Q + gq);
    total[a] = __shfl(t16, (nw - 1) * Q + gq);
    const double before = wo > 0 ? acc : 0.0;               // x at the end of the previous wavefront
    const double incl = cur[a] + pw[a] * before;
    const double prev = REVERSE ? __shfl_down(incl, Q) : __shfl_up(incl, Q);
    excl[a] = sl > 0 ? prev : before;
  }
}

// RAGGED (image sizes that are not powers of two): n0 need not be a multiple of R nor the chunk count of 64 / Q --
// the last real chunk sL holds nv < R rows, chunks behind it none, and the workgroup is padded with such empty
// chunks to whole wavefronts.  The causal recursion simply runs on over the zero rows; its grand total then carries
// lam^pad too much (pad = rows of padding), which the host folds into the table: tab.lamN = lam^(N - pad).  The
// anticausal recursion starts at the last real row.  A column group beyond n1 (n1 not a multiple of Q * VEC) computes
// on zeros and stores nothing.
template <class T, int VEC, int Q, int R, bool RAGGED>
__global__ __launch_bounds__(1024) void colsolve_tri_kernel(const T* __restrict__ Zin, T* __restrict__ Z, int n0, int n1,
                                                           const TriCol* __restrict__ tab, const T* __restrict__ hb,
                                                           int* flags, const double* part_norm, int nnorm, int it,
                                                           double eps, double* scal, double* part_rho, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Zin += pb * pimg;
    Z += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_norm += pb * PART_N;
    part_rho += pb * PART_N;
  }
  if (flags[1]) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds = reinterpret_cast<double*>(smem);
  __shared__ double shn[1024];
  const int S = blockDim.x / Q;
  const int q = threadIdx.x % Q, s = threadIdx.x / Q;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int row0 = s * R, col = q * VEC;
  const bool cv = !RAGGED || (tile * Q + q) * VEC < n1;   // this thread's columns exist
  const int y0 = cv ? (tile * Q + q) * VEC : 0;
  // last real chunk and the rows this thread's chunk holds
  const int sL = RAGGED ? (n0 - 1) / R : S - 1;
  const int nv = !RAGGED ? R : (s < sL ? R : (s == sL ? n0 - sL * R : 0));
  struct alignas(VEC * sizeof(T)) Vec { T v[VEC]; };
  Vec x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (!RAGGED || (k < nv && cv)) {
      x[k] = *reinterpret_cast<const Vec*>(Zin + (size_t)(row0 + k) * n1 + y0);
    } else {
#pragma unroll
      for (int a = 0; a < VEC; ++a) x[k].v[a] = T(0);
    }
  }
  if (it > 0) {
    // the reference's stopping test (phase_unwrap.py:348) on the update the row kernel has just applied
    const double tot = reduce_partials(part_norm, nnorm, shn);
    const double best = scal[10 + ((it - 1) & 1)];
    const bool stop = sqrt(tot) < eps * sqrt(scal[5]) || tot == 0.0 || !(tot == tot) || tot > 1e4 * best;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[0] = it;
      scal[6] = tot;
      scal[10 + (it & 1)] = tot < best ? tot : best;
      if (stop) flags[1] = 1;
    }
    if (stop) return;
  }
  // (per-column constants are re-read from the table where they are needed instead of being kept in registers:
  //  1024 threads leave 128 VGPRs per lane, 64 of which hold the tile)
  const bool c0 = y0 == 0 && cv;   // this thread's first column is column 0, the singular one
  // A sample is read by three recursions.  Each re-read goes through an opaque copy (reread()): hipcc otherwise
  // shares the f32 -> f64 conversion between the passes and keeps all 64 converted samples of the thread alive
  // from one pass to the next -- 128 registers more, i.e. spills at the 128 this launch geometry allows.
  auto reread = [](T v) {
    if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(v));
    return (double)v;
  };
  // ---- pass 1: chunk aggregates of the zero-initialised causal sum (b) and of A's weighted sum (aw)
  double carry[VEC];         // becomes: the true p just above this chunk
  double shift0 = 0.0;       // mean of column 0
  // (two columns at a time: four columns' scan inputs, outputs and multiplier powers at once do not fit)
#pragma unroll
  for (int h = 0; h < VEC; h += 2) {
    double b[2], aw[2], lamR[2];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
      const int a = h + a2;
      const double lam = tab[y0 + a].lam;
      lamR[a2] = tab[y0 + a].lamR;
      double bb = 0.0;
#pragma unroll
      for (int k = 0; k < R; ++k) bb = (double)x[k].v[a] + lam * bb;
      // A's share of this chunk in the data's own precision: it only enters through p_(-1) (f32: relative error
      // 1e-7 in a boundary term)
      const T lamT = (T)lam;
      T ww = T(0);
#pragma unroll
      for (int k = R - 1; k >= 0; --k) ww = x[k].v[a] + lamT * ww;
      b[a2] = bb;
      aw[a2] = (double)ww;
    }
    double A[2], dummy[2], cP[2], B[2];
    chunk_scan<2, VEC, Q, true>(aw, lamR, dummy, A, lds, col + h);
    chunk_scan<2, VEC, Q, false>(b, lamR, cP, B, lds, col + h);
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
      const int a = h + a2;
      // lamR^s by squaring (s < 1024)
      double pw = 1.0, base = lamR[a2];
      for (int bit = s; bit; bit >>= 1) { if (bit & 1) pw *= base; base *= base; }
      carry[a] = cP[a2] + pw * (A[a2] + tab[y0 + a].lamN * B[a2]) * tab[y0 + a].inv;
    }
    if (h == 0 && c0) {
      shift0 = B[0] / (double)n0;
      carry[0] -= (double)row0 * shift0;
    }
  }
  // ---- pass 2: causal recursion in place from the true carry; aggregate of the anticausal one
  double e[VEC];
  T plast[VEC];   // RAGGED: p of the last real row (chunk sL only)
#pragma unroll
  for (int a = 0; a < VEC; ++a) plast[a] = T(0);
#pragma unroll
  for (int a = 0; a < VEC; ++a) {
    const double lam = tab[y0 + a].lam, sh = (a == 0 && c0) ? shift0 : 0.0;
    double p = carry[a];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      p = (reread(x[k].v[a]) - sh) + lam * p;
      x[k].v[a] = (T)p;
    }
    // z_N = zn p_(N-1) enters the last chunk's aggregate (the stored, rounded p: the same value pass 3 starts from)
    double ee;
    if constexpr (!RAGGED) {
      ee = s == S - 1 ? tab[y0 + a].zn * (double)x[R - 1].v[a] : 0.0;
#pragma unroll
      for (int k = R - 1; k >= 0; --k) ee = lam * (ee - (double)x[k].v[a]);
    } else {
      ee = 0.0;
#pragma unroll
      for (int k = R - 1; k >= 0; --k) {
        if (k < nv) {
          if (s == sL && k == nv - 1) { plast[a] = x[k].v[a]; ee = tab[y0 + a].zn * (double)x[k].v[a]; }
          ee = lam * (ee - (double)x[k].v[a]);
        }
      }
    }
    e[a] = ee;
  }
#pragma unroll
  for (int h = 0; h < VEC; h += 2) {
    double lamR[2], Ztot[2], e2[2] = {e[h], e[h + 1]}, cz[2];
    lamR[0] = tab[y0 + h].lamR;
    lamR[1] = tab[y0 + h + 1].lamR;
    chunk_scan<2, VEC, Q, true>(e2, lamR, cz, Ztot, lds, col + h);   // z just below this chunk
    carry[h] = cz[0];
    carry[h + 1] = cz[1];
  }
  // ---- pass 3: anticausal recursion in place, rho from the quadratic form
  double rho = 0.0, zsum0 = 0.0;
#pragma unroll
  for (int a = 0; a < VEC; ++a) {
    const double lam = tab[y0 + a].lam;
    double z = s == sL ? tab[y0 + a].zn * (double)(RAGGED ? plast[a] : x[R - 1].v[a]) : carry[a];
    double dsq = 0.0, zsq = 0.0;
#pragma unroll
    for (int k = R - 1; k >= 0; --k) {
      if (!RAGGED || k < nv) {
        const double zn = lam * (z - reread(x[k].v[a]));
        if (!(s == sL && k == nv - 1)) dsq += (z - zn) * (z - zn);   // no difference across the reflecting end
        zsq += zn * zn;
        z = zn;
        x[k].v[a] = (T)zn;
      }
    }
    double r = -dsq - 2.0 * (double)hb[y0 + a] * zsq;
    if (a == 0 && c0) {
#pragma unroll
      for (int k = 0; k < R; ++k)
        if (!RAGGED || k < nv) zsum0 += (double)x[k].v[0];
      if (s == 0) r += (double)n0 * shift0 * shift0;
      r *= 0.5;                                   // c_0 = 1/2 of SciPy's DCT-II normalisation along the rows
    }
    rho += (!RAGGED || cv) ? r : 0.0;
  }
  if (tile == 0) {
    // column 0: remove the mean of z, add the mean of r (its DC bin is divided by 1)
    const double zs = block_sum(c0 ? zsum0 : 0.0, shn);
    if (c0) {
      const T fix = (T)(shift0 - zs / (double)n0);
#pragma unroll
      for (int k = 0; k < R; ++k) x[k].v[0] += fix;
    }
  }
  {
    // the store addresses equal the load addresses: recomputed from an opaque copy of the row so that the compiler
    // does not keep 16 64-bit addresses alive across the three passes
    int rs = row0;
    asm volatile("" : "+v"(rs));
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (!RAGGED || (k < nv && cv)) *reinterpret_cast<Vec*>(Z + (size_t)(rs + k) * n1 + y0) = x[k];
  }
  if (part_rho) {
    const double tot = block_sum(rho, shn);
    if (threadIdx.x == 0) part_rho[blockIdx.x] = tot / (2.0 * (double)n1);
  }
}

// fused path: rows Z -> z = DCT-III along axis 1, and straight on to the new search direction
// p = z + beta p_prev (phase_unwrap.py:336-340) -- z itself never goes to HBM.  beta = rho / rho_prev with
// rho from the column kernel's Parseval partial sums.
// f32, 4096-point rows: 4 waves per SIMD (<= 128 VGPRs; the unconstrained allocation takes 130 and runs at 3):
// 49 -> 43 us.  Other lengths would spill under that cap (2048: 13 -> 16 us) and keep the default.
template <class T, int LG, bool LAT = false>
#ifndef GPA_IDCTP_COND
#define GPA_IDCTP_COND (sizeof(T) == 4 && (LG == 12 || LG == 13))   // (8192 points: 28 B of scratch buy a second workgroup per CU, -12 %)
#endif
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64 row kernels: 2 waves/SIMD (256 VGPRs) beat 1 wave with AGPR spill-over
#endif
__global__ __launch_bounds__((RowGeom<T, LG, LAT>::THREADS), ((GPA_IDCTP_COND && !LAT) ? 4 : (sizeof(T) == 8 ? GPA_F64_WAVES : 1))) void rowidct_p_kernel(
    const T* __restrict__ Z, const T* __restrict__ pin, T* __restrict__ pout, int n0,
    const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ wk, const int* flags, const double* part_rho,
    int nrho, double* scal, int it, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    pin += pb * pimg;
    pout += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_rho += pb * PART_N;
  }
  // Every input of the kernel is requested before anything waits (one memory round trip for the flags, the
  // spectrum, the previous search direction, the tables and the partial sums together -- a 512-point kernel is
  // little more than its chain of dependent round trips), then the early exit, then the arithmetic.
  const int stop = flags[1];
  using G = RowGeom<T, LG, LAT>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = F::TPF, N = F::L, E = F::E;
  // (short transforms only: the long ones are bandwidth-bound, hide latency behind other workgroups and have no
  //  registers to spare for 2 E more values)
  constexpr bool EARLY = LAT && (E == 8 || GPA_EARLY16);
  if (!EARLY && stop) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[RowGeom<T, LG, LAT>::THREADS];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const bool valid = 2 * pr + 1 < n0;
  const size_t oa = (size_t)(valid ? 2 * pr : 0) * N, ob = oa + N;
  typename G::TW tw;
  __shared__ cpx<T> t1s[G::T1N];
  if constexpr (G::TWLDS) {
    F::fill_pass1_table(t1s, twtab, threadIdx.x, G::THREADS);
    __syncthreads();
    F::load_twiddles(tw, twtab, tid, t1s);
  } else {
    F::load_twiddles(tw, twtab, tid);
  }
  cpx<T> x[E], xm[E], wkv[EARLY ? E : 1], pv[EARLY ? E : 1];
  const bool first = it == 0;                        // first iteration: p = z (pin is uninitialised)
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    x[i] = {Z[oa + k], Z[ob + k]};
    xm[i] = k == 0 ? cpx<T>{T(0), T(0)} : cpx<T>{Z[oa + N - k], Z[ob + N - k]};
    if constexpr (EARLY) {
      wkv[i] = wk[k];
      pv[i] = first ? cpx<T>{T(0), T(0)} : cpx<T>{pin[oa + k], pin[ob + k]};
    }
  }
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  if constexpr (EARLY) D::inv_prepare(x, xm, wkv); else D::inv_prepare(x, xm, tid, wk);
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::inv_scatter(x, lds, tid, T(1) / T(N));
  __syncthreads();
  D::inv_gather(x, lds, tid);
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int c = tid + TPF * i;
    T pa = x[i].x, pb = x[i].y;
    if (!first) {
      if constexpr (EARLY) {
        pa += beta * pv[i].x;
        pb += beta * pv[i].y;
      } else {
        pa += beta * pin[oa + c];
        pb += beta * pin[ob + c];
      }
    }
    pout[oa + c] = pa;
    pout[ob + c] = pb;
  }
}

// One image, rows of at most 512 pixels: rowidct_p_kernel and the stencil kernel in ONE launch.  At these sizes a
// launch costs more than the work of either (an empty kernel: 3.6 us; the stencil kernel: 4.0), so the row kernel
// also transforms the row pair above and the one below its own NF pairs, keeps all 2 NF + 4 rows of the new search
// direction in LDS and applies q = A^T W^2 A p to its own rows there.  (NF + 2) / NF of the transforms instead of one
// more launch per iteration; the GPU is far from full at these sizes.  Same formulas as the two kernels.
// (1024-pixel rows: two own pairs per workgroup, i.e. twice the transforms -- measured slower, 1612 -> 1530 Mpix/s.)
#define GPA_ROWPQ_MAXLG 9
template <class T, int LG>
struct RowPqGeom {
  using G = RowGeom<T, LG, true>;   // (this kernel only serves one small image: the latency-tuned geometry)
  static constexpr int NFH = G::NF + 2;                       // transform groups: own pairs + one halo pair each side
  static constexpr int THREADS = NFH * G::F::TPF;
  static constexpr size_t FFT_BYTES = (size_t)NFH * G::RS * sizeof(cpx<T>);
  static constexpr int PROWS = 2 * NFH;                       // rows of p kept for the stencil
  static constexpr int PPITCH = G::F::L + 4;                  // (a pad of 4 keeps 16-byte row alignment)
  static constexpr size_t LDS_BYTES = FFT_BYTES + (size_t)PROWS * PPITCH * sizeof(T);
};
template <class T, int LG>
__global__ __launch_bounds__((RowPqGeom<T, LG>::THREADS)) void rowidct_pq_kernel(
    const T* __restrict__ Z, const T* __restrict__ pin, T* __restrict__ pout, const T* __restrict__ wgt,
    T* __restrict__ q, int n0, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ wk, const int* flags,
    const double* part_rho, int nrho, double* part_pq, double* scal, int it, size_t pimg) {
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    pin += pb * pimg;
    pout += pb * pimg;
    q += pb * pimg;
    if (wgt) wgt += (pb >> 1) * pimg;   // the two components of an image share its weight
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_rho += pb * PART_N;
    part_pq += pb * PART_N;
  }
  using G = RowGeom<T, LG, true>;
  using H = RowPqGeom<T, LG>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = F::TPF, N = F::L, E = F::E, NF = G::NF;
  static_assert(E == 8, "latency-tuned kernels use the 8-element transforms");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[H::THREADS];
  const int stop = flags[1];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  T* prow = reinterpret_cast<T*>(smem + H::FFT_BYTES);
  const int npairs = n0 / 2;
  const int pr = (int)blockIdx.x * NF + f - 1;        // group 0 / NF + 1: the halo pairs
  const bool valid = pr >= 0 && pr < npairs;
  const bool own = valid && f >= 1 && f <= NF;
  const size_t oa = (size_t)(valid ? 2 * pr : 0) * N, ob = oa + N;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[E], xm[E], wkv[E], pv[E];
  const bool first = it == 0;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = tid + TPF * i;
    x[i] = {Z[oa + k], Z[ob + k]};
    xm[i] = k == 0 ? cpx<T>{T(0), T(0)} : cpx<T>{Z[oa + N - k], Z[ob + N - k]};
    wkv[i] = wk[k];
    pv[i] = first ? cpx<T>{T(0), T(0)} : cpx<T>{pin[oa + k], pin[ob + k]};
  }
  // f32: the stencil's weights are requested here too, with everything else (f64 has no registers to spare for them)
  const int xbase = 2 * (int)blockIdx.x * NF;                // first own image row; LDS row of image row x: x - xbase + 2
  constexpr int VPR = N / 4;                                  // 4-pixel items per row
  constexpr int NITEM = (2 * NF * VPR + H::THREADS - 1) / H::THREADS;
  constexpr bool WPRE = sizeof(T) == 4;
  Vec4<T> wcv[WPRE ? NITEM : 1], wuv[WPRE ? NITEM : 1], wdv[WPRE ? NITEM : 1];
  T wlv[WPRE ? NITEM : 1], wrv[WPRE ? NITEM : 1];
  if constexpr (WPRE) {
#pragma unroll
    for (int t = 0; t < NITEM; ++t) {
      const int item = threadIdx.x + t * H::THREADS;
      const int rl = item / VPR, c0 = (item % VPR) * 4;
      int xg = xbase + rl;
      const bool act = item < 2 * NF * VPR && xg < n0;
      xg = act ? xg : 0;
      const bool up = xg > 0, dn = xg + 1 < n0, hasl = c0 > 0, hasr = c0 + 4 < N;
      if (wgt) {
        const T* wp = wgt + (size_t)xg * N + c0;
        wcv[t] = *reinterpret_cast<const Vec4<T>*>(wp);
        wuv[t] = *reinterpret_cast<const Vec4<T>*>(up ? wp - N : wp);
        wdv[t] = *reinterpret_cast<const Vec4<T>*>(dn ? wp + N : wp);
        wlv[t] = wp[hasl ? -1 : 0];
        wrv[t] = wp[hasr ? 4 : 0];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) wcv[t].v[j] = wuv[t].v[j] = wdv[t].v[j] = T(1);
        wlv[t] = wrv[t] = T(1);
      }
    }
  }
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  D::inv_prepare(x, xm, wkv);
  F::forward(x, lds, tid, tw);
  __syncthreads();
  D::inv_scatter(x, lds, tid, T(1) / T(N));
  __syncthreads();
  D::inv_gather(x, lds, tid);
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int c = tid + TPF * i;
    T pa = x[i].x, pb = x[i].y;
    if (!first) {
      pa += beta * pv[i].x;
      pb += beta * pv[i].y;
    }
    if (own) {
      pout[oa + c] = pa;
      pout[ob + c] = pb;
    }
    prow[(2 * f) * H::PPITCH + c] = pa;
    prow[(2 * f + 1) * H::PPITCH + c] = pb;
  }
  __syncthreads();
  // ---- q = A^T W^2 A p on the 2 NF own rows, 4 pixels per item (as pq_kernel: min of the squared weights per edge)
  double pq = 0;
#pragma unroll
  for (int t = 0; t < NITEM; ++t) {
    const int item = threadIdx.x + t * H::THREADS;
    if (item >= 2 * NF * VPR) continue;
    const int rl = item / VPR, c0 = (item % VPR) * 4;
    const int xg = xbase + rl;
    if (xg >= n0) continue;
    const bool up = xg > 0, dn = xg + 1 < n0, hasl = c0 > 0, hasr = c0 + 4 < N;
    const T* pc = prow + (rl + 2) * H::PPITCH + c0;
    const Vec4<T> vc = *reinterpret_cast<const Vec4<T>*>(pc);
    const Vec4<T> vu = *reinterpret_cast<const Vec4<T>*>(pc - H::PPITCH), vd = *reinterpret_cast<const Vec4<T>*>(pc + H::PPITCH);
    const T pl = hasl ? pc[-1] : T(0), prr = hasr ? pc[4] : T(0);
    Vec4<T> wc, wu, wd;
    T wl = T(1), wr = T(1);
    if constexpr (WPRE) {
      wc = wcv[t]; wu = wuv[t]; wd = wdv[t]; wl = wlv[t]; wr = wrv[t];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) wc.v[j] = wu.v[j] = wd.v[j] = T(1);
      if (wgt) {
        const T* wp = wgt + (size_t)xg * N + c0;
        wc = *reinterpret_cast<const Vec4<T>*>(wp);
        wu = *reinterpret_cast<const Vec4<T>*>(up ? wp - N : wp);
        wd = *reinterpret_cast<const Vec4<T>*>(dn ? wp + N : wp);
        wl = hasl ? wp[-1] : T(1);
        wr = hasr ? wp[4] : T(1);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { wc.v[j] *= wc.v[j]; wu.v[j] *= wu.v[j]; wd.v[j] *= wd.v[j]; }
    wl *= wl;
    wr *= wr;
    Vec4<T> qv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T c = vc.v[j], wj = wc.v[j];
      T acc = T(0);
      if (j < 3) { const T wn = wc.v[j + 1]; acc += (wn < wj ? wn : wj) * (vc.v[j + 1] - c); }
      else if (hasr) acc += (wr < wj ? wr : wj) * (prr - c);
      if (j > 0) { const T wn = wc.v[j - 1]; acc += (wn < wj ? wn : wj) * (vc.v[j - 1] - c); }
      else if (hasl) acc += (wl < wj ? wl : wj) * (pl - c);
      if (dn) { const T wn = wd.v[j]; acc += (wn < wj ? wn : wj) * (vd.v[j] - c); }
      if (up) { const T wn = wu.v[j]; acc += (wn < wj ? wn : wj) * (vu.v[j] - c); }
      qv.v[j] = acc;
      pq += (double)c * (double)acc;
    }
    *reinterpret_cast<Vec4<T>*>(q + (size_t)xg * N + c0) = qv;
  }
  const double tot = block_sum(pq, sh);
  if (threadIdx.x == 0) part_pq[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// generic-size DCT kernels (Bluestein).  Same data flow as the power-of-two kernels,
// everything in the natural layout; 4 FFTs of length L >= 2n-1 per column instead of 2 of
// length n, so roughly 4-8x the arithmetic -- the price of accepting any image size.
// ---------------------------------------------------------------------------

// rows: r (n0 x n) -> Z = DCT-II along axis 1; two rows per complex transform
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_rowdct_kernel(
    const T* __restrict__ r, int n0, int n, T* __restrict__ Z, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec, const cpx<T>* __restrict__ wk,
    const int* flags) {
  if (flags[1]) return;
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const int xa = 2 * pr, xb = 2 * pr + 1;
  const bool va = xa < n0, vb = xb < n0;
  const T* ra = r + (size_t)(va ? xa : 0) * n;
  const T* rb = r + (size_t)(vb ? xb : 0) * n;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) {
      const int src = makhoul_src(slot, n);
      x[i] = {va ? ra[src] : T(0), vb ? rb[src] : T(0)};
    } else {
      x[i] = {T(0), T(0)};
    }
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) lds[F::pad(slot)] = x[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = tid + TPF * i;
    if (k < n) {
      const cpx<T> zm = lds[F::pad(k == 0 ? 0 : n - k)];
      const cpx<T> w = wk[k];
      const cpx<T> X = cmul(w, x[i]) + cmulc(zm, w);
      if (va) Z[(size_t)xa * n + k] = X.x;
      if (vb) Z[(size_t)xb * n + k] = X.y;
    }
  }
}

// rows: Z -> z = DCT-III along axis 1 (in place), partial <r, z>
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_rowidct_kernel(
    T* __restrict__ Z, const T* __restrict__ r, int n0, int n, const cpx<T>* __restrict__ twtab,
    const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec, const cpx<T>* __restrict__ wk,
    double* part, const int* flags) {
  if (flags[1]) return;
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[1024];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int pr = blockIdx.x * G::NF + f;
  const int xa = 2 * pr, xb = 2 * pr + 1;
  const bool va = xa < n0, vb = xb < n0;
  T* za = Z + (size_t)(va ? xa : 0) * n;
  T* zb = Z + (size_t)(vb ? xb : 0) * n;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
  const T inv_n = T(1) / T(n);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = tid + TPF * i;
    x[i] = {T(0), T(0)};
    if (k < n) {
      const cpx<T> X = {va ? za[k] : T(0), vb ? zb[k] : T(0)};
      const cpx<T> Xm = k == 0 ? cpx<T>{T(0), T(0)} : cpx<T>{va ? za[n - k] : T(0), vb ? zb[n - k] : T(0)};
      const cpx<T> d = {X.x + Xm.y, X.y - Xm.x};     // X_k - i X_{n-k}
      const cpx<T> v = cmulc(d, wk[k]);              // V_k = conj(w_k) (.) / 2
      x[i] = {T(0.5) * v.x, T(-0.5) * v.y};          // conj(V_k): IDFT = conj(DFT(conj .))
    }
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = tid + TPF * i;
    if (m < n) lds[F::pad(makhoul_src(m, n))] = {x[i].x * inv_n, -x[i].y * inv_n};
  }
  __syncthreads();
  double dot = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = tid + TPF * i;
    if (c < n) {
      const cpx<T> v = lds[F::pad(c)];
      if (va) { za[c] = v.x; dot += (double)r[(size_t)xa * n + c] * (double)v.x; }
      if (vb) { zb[c] = v.y; dot += (double)r[(size_t)xb * n + c] * (double)v.y; }
    }
  }
  const double tot = block_sum(dot, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// columns: DCT-II along axis 0 -> divide by eigenvalues -> DCT-III along axis 0, in place;
// two adjacent columns per complex transform, NF transforms per workgroup
template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void g_colsolve_kernel(
    T* __restrict__ Z, int n, int n1, const cpx<T>* __restrict__ twtab, const cpx<T>* __restrict__ chirp,
    const cpx<T>* __restrict__ bspec, const cpx<T>* __restrict__ wk, const T* __restrict__ ha,
    const T* __restrict__ ham, const T* __restrict__ hb, const int* flags) {
  if (flags[1]) return;
  using F = WgFFT<T, LG>;
  using B = WgBluestein<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF, NF = G::NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // column-pair index fastest in the thread index: neighbouring lanes read neighbouring columns
  const int f = threadIdx.x % NF, tid = threadIdx.x / NF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int ya = (blockIdx.x * NF + f) * 2, yb = ya + 1;
  const bool va = ya < n1, vb = yb < n1;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    x[i] = {T(0), T(0)};
    if (slot < n) {
      const size_t row = (size_t)makhoul_src(slot, n) * n1;
      x[i] = {va ? Z[row + ya] : T(0), vb ? Z[row + yb] : T(0)};
    }
  }
  B::dft(x, lds, tid, n, chirp, bspec, tw);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) lds[F::pad(slot)] = x[i];
  }
  __syncthreads();
  const T inv_n = T(1) / T(n);
  const T hba = va ? hb[ya] : T(1), hbb = vb ? hb[yb] : T(1);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = tid + TPF * i;
    if (k < n) {
      const cpx<T> zk = x[i], zm = lds[F::pad(k == 0 ? 0 : n - k)];
      const cpx<T> w = wk[k];
      const T h = ha[k], hm = ham[k];
      const cpx<T> qa = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};
      const cpx<T> qb = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};
      const cpx<T> ua = cmul(w, qa), ub = cmul(w, qb);
      T sa = inv_n / (T(-2) * (h + hba)), sb = inv_n / (T(-2) * (h + hbb));
      T sam = inv_n / (T(-2) * (hm + hba)), sbm = inv_n / (T(-2) * (hm + hbb));
      if (k == 0) {
        sam = T(0);
        sbm = T(0);
        if (ya == 0) sa = inv_n;
      }
      const cpx<T> pa = cmulc(cpx<T>{sa * ua.x, sam * ua.y}, w);
      const cpx<T> pb = cmulc(cpx<T>{sb * ub.x, sbm * ub.y}, w);
      // V'_k of the packed pair, conjugated for the conj-DFT-conj inverse
      x[i] = {pa.x - pb.y, -(pa.y + pb.x)};
    } else {
      x[i] = {T(0), T(0)};
    }
  }
  __syncthreads();
  B::dft(x, lds, tid, n, chirp, bspec, tw);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    if (slot < n) {
      const size_t row = (size_t)makhoul_src(slot, n) * n1;
      if (va) Z[row + ya] = x[i].x;
      if (vb) Z[row + yb] = -x[i].y;
    }
  }
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
static bool unwrap_latency_tuned(const Impl* w, int lg) {
  return w->lat_ok && w->nprob <= 2 && lg <= GPA_UNWRAP_LAT_MAXLG;
}

// mixed-radix fused kernels (gpa_unwrap_mr.h, included below)
template <class T>
hipError_t run_mr_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq,
                               double* part_norm, int it, int* nnorm, int init, hipStream_t s);
template <class T>
hipError_t run_mr_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                            hipStream_t s);
template <class T>
hipError_t run_mr_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                           double eps, double* part_rho, int* nrho, const void* zin);

template <class T, int LG>
hipError_t run_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm = nullptr, int nnorm = 0,
                        int it = 0, double eps = 0.0, double* part_rho = nullptr, int* nrho = nullptr,
                        const void* zin = nullptr) {
  if constexpr (!ColGeom<T, LG>::FITS) return hipErrorInvalidValue;
  else {
    if (!part_rho) return hipErrorInvalidValue;   // (the only caller is the fused iteration)
    const bool lat = unwrap_latency_tuned(w, LG);
    // (the latency-tuned instantiation has its own tile geometry: narrower column tiles for one 512^2 image)
    auto launch = [&](auto latc) -> hipError_t {
      constexpr bool LATC = decltype(latc)::value;
      using G = ColGeom<T, LG, LATC>;
      auto kern = colsolve_kernel<T, LG, true, LATC>;
      static unsigned lds_set = 0;   // one flag word per instantiation of this lambda
      hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
      if (e != hipSuccess) return e;
      const int npairs = w->n1 / 2, grid = (npairs + G::CC - 1) / G::CC;
      if (nrho) *nrho = grid;
      GPA_PROF("colsolve_kernel", s);
      kern<<<dim3(grid, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((T*)w->z, w->n1, (const cpx<T>*)w->tw0, (const cpx<T>*)w->wk0s,
                                                   (const T*)w->ha0[compat], (const T*)w->ham0[compat],
                                                   (const T*)w->hb1[compat], w->flags, part_norm, nnorm, it, eps, w->scal,
                                                   part_rho, (const T*)zin, (size_t)w->n0 * w->n1);
      return hipGetLastError();
    };
    if constexpr (LG <= GPA_UNWRAP_LAT_MAXLG) { if (lat) return launch(std::true_type{}); }
    return launch(std::false_type{});
  }
}
template <class T, int LG>
hipError_t run_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq,
                            double* part_norm, int it, int* nnorm, int init, hipStream_t s) {
  if constexpr (!RowGeom<T, LG>::FITS) return hipErrorInvalidValue;
  else {
    const bool lat = unwrap_latency_tuned(w, LG);
    auto launch = [&](auto latc) -> hipError_t {
      constexpr bool LATC = decltype(latc)::value;
      using G = RowGeom<T, LG, LATC>;
      auto kern = rowdct_fused_kernel<T, LG, LATC>;
      static unsigned lds_set = 0;   // one flag word per instantiation of this lambda
      hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
      if (e != hipSuccess) return e;
      const int npairs = w->n0 / 2, grid = (npairs + G::NF - 1) / G::NF;
      *nnorm = grid;
      GPA_PROF("rowdct_fused_kernel", s);
      kern<<<dim3(grid, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((T*)w->r, (const T*)q, w->n0, (const cpx<T>*)w->tw1,
                                                   (const cpx<T>*)w->wk1, w->flags, part_pq, npq, part_norm, w->scal, it,
                                                   ring, init, (size_t)w->n0 * w->n1);
      return hipGetLastError();
    };
    if constexpr (LG <= GPA_UNWRAP_LAT_MAXLG) { if (lat) return launch(std::true_type{}); }
    return launch(std::false_type{});
  }
}

template <class T, int LG>
hipError_t run_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                         hipStream_t s) {
  if constexpr (!RowGeom<T, LG>::FITS) return hipErrorInvalidValue;
  else {
    const bool lat = unwrap_latency_tuned(w, LG);
    auto launch = [&](auto latc) -> hipError_t {
      constexpr bool LATC = decltype(latc)::value;
      using G = RowGeom<T, LG, LATC>;
      auto kern = rowidct_p_kernel<T, LG, LATC>;
      static unsigned lds_set = 0;
      hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
      if (e != hipSuccess) return e;
      const int npairs = w->n0 / 2, grid = (npairs + G::NF - 1) / G::NF;
      GPA_PROF("rowidct_p_kernel", s);
      kern<<<dim3(grid, 1, w->nprob), G::THREADS, G::LDS_BYTES, s>>>((const T*)w->z, (const T*)pin, (T*)pout, w->n0, (const cpx<T>*)w->tw1,
                                                   (const cpx<T>*)w->wk1, w->flags, part_rho, nrho, w->scal, it, (size_t)w->n0 * w->n1);
      return hipGetLastError();
    };
    if constexpr (LG <= GPA_UNWRAP_LAT_MAXLG) { if (lat) return launch(std::true_type{}); }
    return launch(std::false_type{});
  }
}
// the row kernel and the stencil in one launch (one image, rows up to 512 pixels); *npq_out = partial sums of <p, q>
template <class T, int LG>
hipError_t run_rowidct_pq(const Impl* w, const void* pin, void* pout, const void* weight, const double* part_rho,
                          int nrho, double* part_pq, int* npq_out, int it, hipStream_t s) {
  if constexpr (LG > GPA_ROWPQ_MAXLG || unwrap_elems(LG, sizeof(T)) != 8) return hipErrorInvalidValue;
  else {
    using G = RowGeom<T, LG, true>;
    using H = RowPqGeom<T, LG>;
    auto kern = rowidct_pq_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)H::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = w->n0 / 2, grid = (npairs + G::NF - 1) / G::NF;
    if (grid > MAXPART) return hipErrorInvalidValue;
    *npq_out = grid;
    GPA_PROF("rowidct_pq_kernel", s);
    kern<<<dim3(grid, 1, w->nprob), H::THREADS, H::LDS_BYTES, s>>>((const T*)w->z, (const T*)pin, (T*)pout, (const T*)weight,
                                                 (T*)w->q, w->n0, (const cpx<T>*)w->tw1, (const cpx<T>*)w->wk1, w->flags,
                                                 part_rho, nrho, part_pq, w->scal, it, (size_t)w->n0 * w->n1);
    return hipGetLastError();
  }
}
hipError_t dispatch_rowidct_pq(const Impl* w, const void* pin, void* pout, const void* weight, const double* part_rho,
                               int nrho, double* part_pq, int* npq_out, int it, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowidct_pq<float, LG>(w, pin, pout, weight, part_rho, nrho, part_pq, npq_out, it, s) \
                                               : run_rowidct_pq<double, LG>(w, pin, pout, weight, part_rho, nrho, part_pq, npq_out, it, s);
  switch (w->lg1) { CASE(6) CASE(7) CASE(8) CASE(9) }
#undef CASE
  return hipErrorInvalidValue;
}
hipError_t dispatch_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                              hipStream_t s) {
  if (w->generic)
    return w->dtype == 0 ? run_mr_rowidct_p<float>(w, pin, pout, part_rho, nrho, it, s)
                         : run_mr_rowidct_p<double>(w, pin, pout, part_rho, nrho, it, s);
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowidct_p<float, LG>(w, pin, pout, part_rho, nrho, it, s) \
                                               : run_rowidct_p<double, LG>(w, pin, pout, part_rho, nrho, it, s);
  switch (w->lg1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
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
  if (const char* fq = getenv("GPA_TRI_Q")) {   // diagnostic: force the number of column groups per workgroup
    const int q = atoi(fq);
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

template <class T, int VEC, int Q, int R>
hipError_t run_colsolve_tri(const Impl* w, int S, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                            double eps, double* part_rho, int* nrho, const void* zin) {
  const bool ragged = w->generic;
  const int threads = S * Q, grid = (w->n1 + Q * VEC - 1) / (Q * VEC);
  const size_t lds = (size_t)16 * Q * VEC * sizeof(double);
  if (nrho) *nrho = grid;
  GPA_PROF("colsolve_kernel", s);
  if (ragged) {
    auto kern = colsolve_tri_kernel<T, VEC, Q, R, true>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)lds, lds_set);
    if (e != hipSuccess) return e;
    kern<<<dim3(grid, 1, w->nprob), threads, lds, s>>>((const T*)(zin ? zin : w->z), (T*)w->z, w->n0, w->n1,
                                    (const TriCol*)w->tritab, (const T*)w->hb1[compat], w->flags, part_norm, nnorm, it, eps,
                                    w->scal, part_rho, (size_t)w->n0 * w->n1);
  } else {
    auto kern = colsolve_tri_kernel<T, VEC, Q, R, false>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)lds, lds_set);
    if (e != hipSuccess) return e;
    kern<<<dim3(grid, 1, w->nprob), threads, lds, s>>>((const T*)(zin ? zin : w->z), (T*)w->z, w->n0, w->n1,
                                    (const TriCol*)w->tritab, (const T*)w->hb1[compat], w->flags, part_norm, nnorm, it, eps,
                                    w->scal, part_rho, (size_t)w->n0 * w->n1);
  }
  return hipGetLastError();
}

// square images: the transform-free column solve
template <class T>
hipError_t dispatch_colsolve_tri(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                                 double eps, double* part_rho, int* nrho, const void* zin) {
  constexpr int VEC = 16 / sizeof(T), RB = TriRows<T>::value;
  const int Q = w->triQ, S = w->triS;
#define GPA_TRI_CALL(QQ, RR) run_colsolve_tri<T, VEC, QQ, RR>(w, S, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
  if constexpr (sizeof(T) == 4) {
    // f32 columns of 16384 points: 16 rows per thread (1024 chunks), Q = 1 -- spills, and still ahead of a transform
    // kernel that is down to ONE column pair (8-byte row segments) per workgroup there
    if (w->triR == 2 * RB) return GPA_TRI_CALL(1, 2 * RB);
  }
  if (w->triR == RB) {
    switch (Q) {
      case 4: return GPA_TRI_CALL(4, RB);
      case 2: return GPA_TRI_CALL(2, RB);
      default: return GPA_TRI_CALL(1, RB);
    }
  }
  switch (Q) {
    case 4: return GPA_TRI_CALL(4, RB / 2);
    case 2: return GPA_TRI_CALL(2, RB / 2);
    default: return GPA_TRI_CALL(1, RB / 2);
  }
#undef GPA_TRI_CALL
}

hipError_t dispatch_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm = nullptr,
                             int nnorm = 0, int it = 0, double eps = 0.0, double* part_rho = nullptr,
                             int* nrho = nullptr, const void* zin = nullptr) {
  if (w->generic && part_rho) {
    // smooth sizes: the transform-free solve where it applies (square images; it is 2-3x faster than two mixed-radix
    // transforms per column pair), GPA_COLSOLVE=fft keeps the transforms
    if (w->tritab && w->col_mode != 2)
      return w->dtype == 0 ? dispatch_colsolve_tri<float>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
                           : dispatch_colsolve_tri<double>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
    return w->dtype == 0 ? run_mr_colsolve<float>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
                         : run_mr_colsolve<double>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
  }
  // Square images can solve the columns without a transform (colsolve_tri_kernel).  Measured at 4096^2 on MI355X
  // (profiles/r02_colsolve_tri.txt): f64 1.54 ms per step against 2.0 for the DCT kernel (whose f64 transforms
  // spill), f32 82 us per launch against 68 -- the f32 DCT kernel is the faster one.  So: f64 by default,
  // GPA_COLSOLVE=tri / fft forces one or the other (tests compare the two).
  // (f32 columns of 8192 points: the transform kernel is down to two column pairs -- 16-byte row segments -- per
  //  workgroup there and loses to the recursion: 453 against ~330 us per launch)
  const bool want_tri = w->col_mode ? w->col_mode == 1 : (w->dtype != 0 || w->lg0 >= 13);
  if (w->tritab && part_rho && want_tri && w->n0 / w->triR <= 1024)
    return w->dtype == 0 ? dispatch_colsolve_tri<float>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
                         : dispatch_colsolve_tri<double>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
#define CASE(LG) case LG: return w->dtype == 0 ? run_colsolve<float, LG>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin) \
                                               : run_colsolve<double, LG>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
  switch (w->lg0) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
// init: first iteration of a solve on prepared residuals -- part_pq / npq are then the producer's partial norms
hipError_t dispatch_rowdct_fused(const Impl* w, const void* q, int ring, const double* part_pq, int npq,
                                 double* part_norm, int it, int* nnorm, int init, hipStream_t s) {
  if (w->generic)
    return w->dtype == 0 ? run_mr_rowdct_fused<float>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s)
                         : run_mr_rowdct_fused<double>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
#define CASE(LG) case LG: return w->dtype == 0 ? run_rowdct_fused<float, LG>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s) \
                                               : run_rowdct_fused<double, LG>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  switch (w->lg1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}

template <class T, int LG>
hipError_t run_g_rowdct(const Impl* w, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_rowdct_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = (w->n0 + 1) / 2, grid = (npairs + G::NF - 1) / G::NF;
    GPA_PROF("g_rowdct_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>((const T*)w->r, w->n0, w->n1, (T*)w->z, (const cpx<T>*)w->btw1,
                                                 (const cpx<T>*)w->chirp1, (const cpx<T>*)w->bspec1,
                                                 (const cpx<T>*)w->gwk1, w->flags);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_g_rowidct(const Impl* w, int* nparts, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_rowidct_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = (w->n0 + 1) / 2, grid = (npairs + G::NF - 1) / G::NF;
    *nparts = grid;
    GPA_PROF("g_rowidct_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>((T*)w->z, (const T*)w->r, w->n0, w->n1, (const cpx<T>*)w->btw1,
                                                 (const cpx<T>*)w->chirp1, (const cpx<T>*)w->bspec1,
                                                 (const cpx<T>*)w->gwk1, w->part, w->flags);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_g_colsolve(const Impl* w, int compat, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = g_colsolve_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int npairs = (w->n1 + 1) / 2, grid = (npairs + G::NF - 1) / G::NF;
    GPA_PROF("g_colsolve_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>((T*)w->z, w->n0, w->n1, (const cpx<T>*)w->btw0,
                                                 (const cpx<T>*)w->chirp0, (const cpx<T>*)w->bspec0,
                                                 (const cpx<T>*)w->gwk0, (const T*)w->gha0[compat],
                                                 (const T*)w->gham0[compat], (const T*)w->hb1[compat], w->flags);
    return hipGetLastError();
  }
}
hipError_t dispatch_g_rowdct(const Impl* w, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_g_rowdct<float, LG>(w, s) : run_g_rowdct<double, LG>(w, s);
  switch (w->lgb1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
hipError_t dispatch_g_rowidct(const Impl* w, int* nparts, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_g_rowidct<float, LG>(w, nparts, s) : run_g_rowidct<double, LG>(w, nparts, s);
  switch (w->lgb1) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}
hipError_t dispatch_g_colsolve(const Impl* w, int compat, hipStream_t s) {
#define CASE(LG) case LG: return w->dtype == 0 ? run_g_colsolve<float, LG>(w, compat, s) : run_g_colsolve<double, LG>(w, compat, s);
  switch (w->lgb0) { GPA_FOR_LG(CASE) }
#undef CASE
  return hipErrorInvalidValue;
}

}  // namespace
}  // namespace gpa
#include "gpa_unwrap_mr.h"
namespace gpa {
namespace {

// host-side radix-2 FFT in double, for the Bluestein kernel spectra
void host_fft(std::vector<double>& re, std::vector<double>& im) {
  const size_t n = re.size();
  int lg = 0;
  while ((size_t(1) << lg) < n) ++lg;
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
    if (r > i) { std::swap(re[i], re[r]); std::swap(im[i], im[r]); }
  }
  for (size_t len = 2; len <= n; len <<= 1)
    for (size_t s0 = 0; s0 < n; s0 += len)
      for (size_t j = 0; j < len / 2; ++j) {
        const double ang = -2.0 * M_PI * (double)j / (double)len, wr = cos(ang), wi = sin(ang);
        const size_t a = s0 + j, b = s0 + j + len / 2;
        const double vr = re[b] * wr - im[b] * wi, vi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - vr; im[b] = im[a] - vi;
        re[a] += vr; im[a] += vi;
      }
}

template <class T>
hipError_t upload_vec(void** dst, const std::vector<double>& v, size_t* bytes, hipStream_t s) {
  std::vector<T> tmp(v.begin(), v.end());
  hipError_t e = hipMalloc(dst, tmp.size() * sizeof(T) + 16);
  if (e != hipSuccess) return e;
  *bytes += tmp.size() * sizeof(T);
  e = hipMemcpyAsync(*dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}
hipError_t upload(int dtype, void** dst, const std::vector<double>& v, size_t* bytes, hipStream_t s) {
  return dtype == 0 ? upload_vec<float>(dst, v, bytes, s) : upload_vec<double>(dst, v, bytes, s);
}

int ilog2_exact(int n) {
  int lg = 0;
  while ((1 << lg) < n) ++lg;
  return (1 << lg) == n ? lg : -1;
}

}  // namespace

// transform-free column solve (colsolve_tri_kernel): per row frequency j the decay lam_j of the Green's function of
// (T + mu_j) and the constants of its boundary terms, in long double.  lamN carries the launch geometry's padding
// (lam^(N - pad), see the kernel); no table if the column does not fit one workgroup.
hipError_t build_tritab(Impl* w, hipStream_t s, size_t* bytes) {
  const int n0 = w->n0, n1 = w->n1;
  int Q, S, pad;
  const int R = tri_rows(w->rsz, n0);
  if (w->dtype == 0) tri_geometry<float>(n0, n1, w->generic, R, &Q, &S, &pad);
  else tri_geometry<double>(n0, n1, w->generic, R, &Q, &S, &pad);
  if (S * Q > 1024 || n1 % (w->dtype == 0 ? 4 : 2)) return hipSuccess;   // (16-byte column vectors)
  w->triQ = Q;
  w->triS = S;
  w->triR = R;
  std::vector<TriCol> tc((size_t)n1);
  for (int j = 0; j < n1; ++j) {
    if (j == 0) { tc[0] = {1.0, 1.0, 1.0, 0.0, 0.0}; continue; }
    const long double sj = sinl((long double)M_PI * j / (2.0L * n1)), h = 2 * sj * sj;
    const long double lam = (1 + h) - sqrtl(h * (2 + h));
    tc[j].lam = (double)lam;
    tc[j].lamR = (double)powl(lam, R);
    tc[j].lamN = (double)powl(lam, (long double)(n0 - pad));
    tc[j].inv = (double)(1.0L / (1.0L - powl(lam, 2.0L * n0)));
    tc[j].zn = (double)(-lam / (1.0L - lam));
  }
  hipError_t e = hipMalloc(&w->tritab, tc.size() * sizeof(TriCol));
  if (e != hipSuccess) return e;
  *bytes += tc.size() * sizeof(TriCol);
  e = hipMemcpyAsync(w->tritab, tc.data(), tc.size() * sizeof(TriCol), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return e;
}

hipError_t unwrap_workspace_create(int dtype, int n0, int n1, hipStream_t s, UnwrapWorkspace* ws, size_t* bytes_out,
                                   int nprob) {
  Impl* w = new Impl();
  memset(w, 0, sizeof(Impl));
  ws->impl = w;
  ws->dtype = dtype;
  ws->n0 = n0;
  ws->n1 = n1;
  w->dtype = dtype;
  w->n0 = n0;
  w->n1 = n1;
  w->nprob = nprob < 1 ? 1 : nprob;
  w->cap = w->nprob;
  w->rsz = dtype == 0 ? 4 : 8;
  w->lg0 = ilog2_exact(n0);
  w->lg1 = ilog2_exact(n1);
  const int maxlg = dtype == 0 ? 14 : 13;
  const bool pow2ok = w->lg0 >= 6 && w->lg1 >= 6 && w->lg0 <= maxlg && w->lg1 <= maxlg;
  auto blue_lg = [](int n) { int lg = 6; while ((1 << lg) < 2 * n - 1) ++lg; return lg; };
  w->lgb0 = blue_lg(n0);
  w->lgb1 = blue_lg(n1);
  w->generic = !pow2ok;
  w->supported = pow2ok || (n0 >= 2 && n1 >= 2 && w->lgb0 <= maxlg && w->lgb1 <= maxlg);
  size_t bytes = 0;
  const size_t npx = (size_t)n0 * n1;
  hipError_t e;
  void** arrs[] = {&w->r, &w->p, &w->p2, &w->q, &w->z};
  for (void** a : arrs) {
    e = hipMalloc(a, npx * w->rsz * w->cap);
    if (e != hipSuccess) return e;
    bytes += npx * w->rsz * w->cap;
  }
  w->ring[0] = w->p;
  w->ring[1] = w->p2;
  w->nring = 2;
  e = hipMalloc((void**)&w->scal, (size_t)SCAL_N * w->cap * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&w->flags, (size_t)FLAGS_N * w->cap * sizeof(int));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&w->part, PART_N * w->cap * sizeof(double));
  if (e != hipSuccess) return e;
  if (w->supported && w->generic) {
    for (int ax = 0; ax < 2; ++ax) {
      const int n = ax == 0 ? n0 : n1, lgb = ax == 0 ? w->lgb0 : w->lgb1, L = 1 << lgb, tpf = L / 16;
      std::vector<double> t((size_t)2 * L), ch((size_t)2 * n), wkv((size_t)2 * n);
      for (int k = 0; k < L; ++k) { t[2 * k] = cos(-2.0 * M_PI * k / L); t[2 * k + 1] = sin(-2.0 * M_PI * k / L); }
      std::vector<double> bre((size_t)L, 0.0), bim((size_t)L, 0.0);
      for (int m = 0; m < n; ++m) {
        const long long mm = ((long long)m * m) % (2LL * n);     // c_m = exp(i pi m^2 / n), argument reduced exactly
        const double cr = cos(M_PI * (double)mm / n), ci = sin(M_PI * (double)mm / n);
        ch[2 * m] = cr; ch[2 * m + 1] = ci;
        bre[m] = cr; bim[m] = ci;
        if (m > 0) { bre[L - m] = cr; bim[L - m] = ci; }
        wkv[2 * m] = cos(-M_PI * m / (2.0 * n)); wkv[2 * m + 1] = sin(-M_PI * m / (2.0 * n));
      }
      host_fft(bre, bim);
      std::vector<double> bs((size_t)2 * L);
      for (int i = 0; i < 16; ++i)
        for (int tt = 0; tt < tpf; ++tt) {
          const int k = spec_index_rt(lgb, tt, i);
          bs[2 * ((size_t)i * tpf + tt)] = bre[k] / L;
          bs[2 * ((size_t)i * tpf + tt) + 1] = bim[k] / L;
        }
      if ((e = upload(dtype, ax == 0 ? &w->btw0 : &w->btw1, t, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, ax == 0 ? &w->chirp0 : &w->chirp1, ch, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, ax == 0 ? &w->bspec0 : &w->bspec1, bs, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, ax == 0 ? &w->gwk0 : &w->gwk1, wkv, &bytes, s)) != hipSuccess) return e;
    }
    // fused path on the mixed-radix engine: rows of whole 4-pixel vectors (pq_kernel) and, per axis, a transform that
    // fits LDS -- length n itself when it is smooth, chirp-z on the smallest smooth L >= 2n - 1 otherwise
    {
      const int max_elems = (int)((size_t)159 * 1024 / (2 * w->rsz));
      w->mr_ok = !getenv("GPA_NO_MR") && mr_make_dft(n0, max_elems, &w->mr0) &&
                 mr_make_dft(n1, max_elems, &w->mr1);
      if (getenv("GPA_MR_FORCE_BLUESTEIN") && w->mr_ok) {   // diagnostic / tests: chirp-z also for smooth lengths
        for (MrDft* d : {&w->mr0, &w->mr1}) {
          if (d->blue) continue;
          MrDft t = *d;
          t.blue = 1;
          bool found = false;
          for (int L = 2 * d->n - 1; mr_lds_elems(L) <= max_elems && !found; ++L) found = mr_make_plan(L, &t.pl);
          if (found) *d = t; else w->mr_ok = false;
        }
      }
    }
    if (w->mr_ok) {
      for (int ax = 0; ax < 2; ++ax) {
        const MrDft& d = ax == 0 ? w->mr0 : w->mr1;
        const int L = d.pl.n;
        std::vector<double> t((size_t)2 * mr_lds_elems(L), 0.0);   // entry k at mr_pad(k), see mr_store()
        for (int k = 0; k < L; ++k) {
          t[2 * (size_t)mr_pad(k)] = cos(-2.0 * M_PI * k / L);
          t[2 * (size_t)mr_pad(k) + 1] = sin(-2.0 * M_PI * k / L);
        }
        if ((e = upload(dtype, ax == 0 ? &w->mrW0 : &w->mrW1, t, &bytes, s)) != hipSuccess) return e;
        if (d.blue) {
          // FFT_L(b) / L, b[m] = b[L - m] = exp(i pi m^2 / n), by the engine's own passes in double on the host
          std::vector<cpx<double>> W((size_t)mr_lds_elems(L)), img((size_t)mr_lds_elems(L), cpx<double>{0.0, 0.0});
          std::vector<cpx<double>> regs((size_t)MR_REGS * d.pl.T);
          for (int k = 0; k < L; ++k) W[mr_pad(k)] = {t[2 * (size_t)mr_pad(k)], t[2 * (size_t)mr_pad(k) + 1]};
          for (int m = 0; m < d.n; ++m) {
            const long long mm = ((long long)m * m) % (2LL * d.n);
            const cpx<double> c = {cos(M_PI * (double)mm / d.n), sin(M_PI * (double)mm / d.n)};
            img[mr_pad(m)] = c;
            if (m > 0) img[mr_pad(L - m)] = c;
          }
          for (int p = 0; p < d.pl.np; ++p) {
#define GPA_HOST_PASS(R)                                                                                              \
  case R:                                                                                                             \
    for (int tt = 0; tt < d.pl.T; ++tt)                                                                               \
      mr_load<double, R>(&regs[(size_t)MR_REGS * tt], reinterpret_cast<const double*>(img.data()), L, tt, d.pl.T);    \
    for (int tt = 0; tt < d.pl.T; ++tt)                                                                               \
      mr_store<double, R>(&regs[(size_t)MR_REGS * tt], reinterpret_cast<double*>(img.data()), L, d.pl.stride[p],      \
                          d.pl.magic[p], tt, d.pl.T, reinterpret_cast<const double*>(W.data()));                      \
    break;
            switch (d.pl.radix[p]) {
              GPA_HOST_PASS(2) GPA_HOST_PASS(3) GPA_HOST_PASS(4) GPA_HOST_PASS(5) GPA_HOST_PASS(6) GPA_HOST_PASS(7)
              GPA_HOST_PASS(8) GPA_HOST_PASS(10) GPA_HOST_PASS(11) GPA_HOST_PASS(12) GPA_HOST_PASS(13) GPA_HOST_PASS(14)
              GPA_HOST_PASS(15) GPA_HOST_PASS(16)
            }
#undef GPA_HOST_PASS
          }
          std::vector<double> bs((size_t)2 * L);
          for (int k = 0; k < L; ++k) { bs[2 * k] = img[mr_pad(k)].x / L; bs[2 * k + 1] = img[mr_pad(k)].y / L; }
          if ((e = upload(dtype, ax == 0 ? &w->mrB0 : &w->mrB1, bs, &bytes, s)) != hipSuccess) return e;
        }
      }
    }
    for (int compat = 0; compat < 2; ++compat) {
      const double A0 = compat ? n1 : n0, A1 = compat ? n0 : n1;
      std::vector<double> a((size_t)n0), am((size_t)n0), b((size_t)n1);
      for (int k = 0; k < n0; ++k) {
        const double sk = sin(M_PI * k / (2.0 * A0)), sm = sin(M_PI * (n0 - k) / (2.0 * A0));
        a[k] = 2 * sk * sk;
        am[k] = 2 * sm * sm;
      }
      for (int j = 0; j < n1; ++j) { const double sj = sin(M_PI * j / (2.0 * A1)); b[j] = 2 * sj * sj; }
      if ((e = upload(dtype, &w->gha0[compat], a, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->gham0[compat], am, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->hb1[compat], b, &bytes, s)) != hipSuccess) return e;
    }
  }
  if (w->supported && !w->generic) {
    for (int ax = 0; ax < 2; ++ax) {
      const int n = ax == 0 ? n0 : n1;
      std::vector<double> t((size_t)2 * n);
      for (int k = 0; k < n; ++k) {
        t[2 * k] = cos(-2.0 * M_PI * k / n);
        t[2 * k + 1] = sin(-2.0 * M_PI * k / n);
      }
      e = upload(dtype, ax == 0 ? &w->tw0 : &w->tw1, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    {
      std::vector<double> t((size_t)2 * n1);
      for (int k = 0; k < n1; ++k) {
        t[2 * k] = cos(-M_PI * k / (2.0 * n1));
        t[2 * k + 1] = sin(-M_PI * k / (2.0 * n1));
      }
      e = upload(dtype, &w->wk1, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    // tables of the column kernel in the spectral layout [register][thread] of ITS transform (unwrap_elems)
    const int E0 = unwrap_elems(w->lg0, (size_t)w->rsz), tpf0 = n0 / E0;
    {
      std::vector<double> t((size_t)2 * n0);
      for (int i = 0; i < E0; ++i)
        for (int tt = 0; tt < tpf0; ++tt) {
          const int k = spec_index_rt(w->lg0, tt, i, E0);
          t[2 * ((size_t)i * tpf0 + tt)] = cos(-M_PI * k / (2.0 * n0));
          t[2 * ((size_t)i * tpf0 + tt) + 1] = sin(-M_PI * k / (2.0 * n0));
        }
      e = upload(dtype, &w->wk0s, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    // 1 - cos(pi i / A) = 2 sin^2(pi i / (2A)).  Reference (compat = 1): axis-0 bins use A = n1
    // and axis-1 bins use A = n0 (phase_unwrap.py:107-109); compat = 0: A = own axis length.
    for (int compat = 0; compat < 2; ++compat) {
      const double A0 = compat ? n1 : n0, A1 = compat ? n0 : n1;
      std::vector<double> a((size_t)n0), am((size_t)n0), b((size_t)n1);
      for (int i = 0; i < E0; ++i)
        for (int tt = 0; tt < tpf0; ++tt) {
          const int k = spec_index_rt(w->lg0, tt, i, E0);
          const double sk = sin(M_PI * k / (2.0 * A0)), sm = sin(M_PI * (n0 - k) / (2.0 * A0));
          a[(size_t)i * tpf0 + tt] = 2 * sk * sk;
          am[(size_t)i * tpf0 + tt] = 2 * sm * sm;
        }
      for (int j = 0; j < n1; ++j) {
        const double sj = sin(M_PI * j / (2.0 * A1));
        b[j] = 2 * sj * sj;
      }
      if ((e = upload(dtype, &w->ha0[compat], a, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->ham0[compat], am, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->hb1[compat], b, &bytes, s)) != hipSuccess) return e;
    }
    if (n0 == n1 && (e = build_tritab(w, s, &bytes)) != hipSuccess) return e;
  }
  if (w->supported && w->generic && w->mr_ok && n0 == n1 && (e = build_tritab(w, s, &bytes)) != hipSuccess) return e;
  if (bytes_out) *bytes_out = bytes;
  return hipSuccess;
}

bool unwrap_supports_batch(const UnwrapWorkspace* ws) {
  const Impl* w = (const Impl*)ws->impl;
  return w && w->supported && (!w->generic || w->mr_ok);
}

void unwrap_workspace_destroy(UnwrapWorkspace* ws) {
  Impl* w = (Impl*)ws->impl;
  if (!w) return;
  void* bufs[] = {w->r, w->p, w->p2, w->q, w->z, w->tw0, w->tw1, w->wk1, w->wk0s, w->ha0[0], w->ha0[1], w->ham0[0],
                  w->ham0[1], w->hb1[0], w->hb1[1], w->scal, w->flags, w->part, w->btw0, w->btw1, w->chirp0, w->chirp1,
                  w->bspec0, w->bspec1, w->gwk0, w->gwk1, w->gha0[0], w->gha0[1], w->gham0[0], w->gham0[1], w->tritab,
                  w->mrW0, w->mrW1, w->mrB0, w->mrB1};
  for (void* b : bufs)
    if (b) hipFree(b);
  for (int j = 2; j < w->nring; ++j)
    if (w->ring[j]) hipFree(w->ring[j]);
  delete w;
  ws->impl = nullptr;
}

// GPA_NO_ROWPQ=1 (diagnostic): keep the stencil a launch of its own for small single images
static bool getenv_rowpq_off() {
  static const bool v = getenv("GPA_NO_ROWPQ") != nullptr;
  return v;
}

template <class T>
static hipError_t run_pcg(Impl* w, const void* a, const void* b, const void* weight, bool from_psi, int kmax,
                          double eps, int compat, void* phi, hipStream_t s) {
  const int n0 = w->n0, n1 = w->n1;
  const size_t npx = (size_t)n0 * n1;
  // The reference's preconditioner table uses cos(pi I / M) + cos(pi J / N) with the axis lengths swapped
  // (phase_unwrap.py:107-109); `compat` reproduces that.  Once one side is at least twice the other that
  // table has a zero away from the DC bin (I = 2M) and the reference itself returns NaN: there is nothing
  // to reproduce, and the true Laplacian eigenvalues are used instead.
  if (compat && (n0 >= 2 * n1 || n1 >= 2 * n0)) compat = 0;
  {
    const char* mode = getenv("GPA_COLSOLVE");   // once per solve, not per launch (environment lookups cost host time)
    w->col_mode = !mode ? 0 : (mode[0] == 't' ? 1 : 2);
    w->lat_ok = getenv("GPA_NO_LAT") == nullptr;
  }
  const int g2 = n0, np2 = n0;   // stencil kernels: one workgroup per image row
  if (np2 > MAXPART) return hipErrorInvalidValue;
  // band height of the stencil kernel: 16 rows for large images, fewer when that would leave
  // less than ~2048 workgroups (small images are latency-, not bandwidth-bound)
  // rows of whole 4-pixel vectors take 16-byte accesses in the stencil / flush / mixed-radix row kernels, any other
  // row length (generic sizes only: power-of-two rows are always whole vectors) their one-pixel instantiations
  const int V = (n1 % 4) == 0 ? 4 : 1;
  const int pqcols = 256 * V;
  int band = PQ_ROWS;
  while (band > 4 && (size_t)((n1 + pqcols - 1) / pqcols) * ((n0 + band - 1) / band) < 2048) band /= 2;
  const dim3 gpq((n1 + pqcols - 1) / pqcols, (n0 + band - 1) / band);
  int npq = gpq.x * gpq.y;   // (the fused row + stencil kernel of small images reports its own count)
  if (npq > MAXPART) return hipErrorInvalidValue;
  const int gl = 2048;   // grid-stride elementwise kernels
  // the residual of an f32 iteration cannot fall below a few ulps of ||r0||
  double eps_floor = sizeof(T) == 4 ? 4e-6 : 0.0;
  if (const char* ef = getenv("GPA_F32_EPS_FLOOR")) { if (sizeof(T) == 4) eps_floor = atof(ef); }   // diagnostic
  if (eps < eps_floor) eps = eps_floor;
  hipError_t e;
  const dim3 gsu((n1 + 255) / 256, (n0 + SETUP_ROWS - 1) / SETUP_ROWS);
  const int nsu = gsu.x * gsu.y;
  if (nsu > MAXPART) return hipErrorInvalidValue;
  if (a) {
    setup_kernel<T><<<gsu, 256, 0, s>>>((const T*)a, (const T*)b, (const T*)weight, from_psi ? 1 : 0, n0, n1, (T*)w->r,
                                        (T*)phi, w->part);
    scal_init_kernel<<<1, 256, 0, s>>>(w->part, nsu, w->scal, w->flags);
  } else {
    // prepared: r0 and its w->prepared_parts partial norms were written by the producer of the gradients
    // (reconstruct_setup_kernel).  phi = 0: the fused path's first phi_flush_kernel starts from 0, the
    // other paths update phi in place and need it cleared
    const bool fused_path = !w->generic || w->mr_ok;
    if (!fused_path && (e = hipMemsetAsync(phi, 0, npx * w->rsz, s)) != hipSuccess) return e;
    // (fused path: the first row kernel starts the solve from the producer's partial norms, solve_init())
    if (!fused_path) {
      GPA_PROF("scalar_kernels", s);
      scal_init_kernel<<<dim3(1, 1, w->nprob), 256, 0, s>>>(w->part, w->prepared_parts, w->scal, w->flags);
    }
  }
  const bool vec4 = !w->generic || w->mr_ok;   // the fused 4-kernel iteration
  w->iters_slot = vec4 ? 3 : 0;
  if (w->nprob > 1 && (a || !vec4)) return hipErrorNotSupported;   // batched: prepared start on the fused path only
  if (vec4) {
    // fused power-of-two path: 4 kernels per iteration, no scalar kernels.  The phi / r update
    // of iteration it-1 rides in the row-DCT kernel of iteration it; the stopping test is
    // evaluated by the column kernel from that kernel's partial norms.
    double* part_rho = w->part;
    double* part_pq = w->part + MAXPART;
    double* part_norm = w->part + 2 * MAXPART;
    // the search directions of the last `ring` iterations stay in HBM (64 MiB each at 4096^2 f32 -- HBM is
    // plentiful), so phi += alpha p costs one pass over phi per `ring` iterations instead of one per iteration
    int ring = kmax < RING_MAX ? kmax : RING_MAX;
    if (ring < 2) ring = 2;
    while (w->nring < ring) {
      void* buf = nullptr;
      if (hipMalloc(&buf, npx * w->rsz * w->cap) != hipSuccess) { (void)hipGetLastError(); break; }
      w->ring[w->nring++] = buf;
    }
    if (w->nring < ring) ring = w->nring;   // out of memory: flush more often
    RingPtrs<T> rp;
    for (int j = 0; j < RING_MAX; ++j) rp.p[j] = (const T*)w->ring[j < ring ? j : 0];
    bool phi_unwritten = a == nullptr;   // prepared start: nobody has zeroed phi
    // final_it = kmax: the flush that ends the solve (takes the last step length from the stencil kernel's partial
    // sums and files the iteration count); otherwise a flush in mid-solve, committed by a one-thread kernel
    auto flush = [&](int final_it) {
      { GPA_PROF("phi_flush_kernel", s);
        if (V == 4)
          phi_flush_kernel<T, 4><<<dim3(gl, 1, w->nprob), 256, 0, s>>>(rp, ring, (T*)phi, npx / 4, w->scal, w->flags,
                                                                      phi_unwritten ? 1 : 0, npx, final_it, part_pq, npq);
        else
          phi_flush_kernel<T, 1><<<dim3(gl, 1, w->nprob), 256, 0, s>>>(rp, ring, (T*)phi, npx, w->scal, w->flags,
                                                                      phi_unwritten ? 1 : 0, npx, final_it, part_pq, npq); }
      if (!final_it) { GPA_PROF("scalar_kernels", s); phi_commit_kernel<<<dim3(1, 1, w->nprob), 1, 0, s>>>(w->flags); }
      phi_unwritten = false;
    };
    int nnorm = 0;
    // one image with rows of at most 512 pixels: row kernel and stencil in one launch (rowidct_pq_kernel)
    const bool rowpq = !w->generic && w->lat_ok && w->nprob <= 2 && w->lg1 <= GPA_ROWPQ_MAXLG && w->n0 >= 4 &&
                       !getenv_rowpq_off();
    for (int it = 0; it < kmax; ++it) {
      // (first iteration of a prepared start: the partial norms of r0 ride in the part_pq / npq arguments)
      const bool init = it == 0 && a == nullptr;
      if ((e = dispatch_rowdct_fused(w, w->q, ring, init ? w->part : part_pq, init ? w->prepared_parts : npq, part_norm, it,
                                     &nnorm, init ? 1 : 0, s)) != hipSuccess) return e;
      int nrow = 0;   // partial sums of rho = <r, z>: one per column workgroup (Parseval, solve_combine)
      if ((e = dispatch_colsolve(w, compat, s, part_norm, nnorm, it, eps, part_rho, &nrow, w->r)) != hipSuccess) return e;
      if (it > 0 && it % ring == 0) flush(0);   // slot it % ring still holds p of iteration it - ring
      const T* pin = (const T*)w->ring[(it + ring - 1) % ring];
      T* pout = (T*)w->ring[it % ring];
      if (rowpq) {
        if ((e = dispatch_rowidct_pq(w, pin, pout, weight, part_rho, nrow, part_pq, &npq, it, s)) != hipSuccess) return e;
        continue;
      }
      if ((e = dispatch_rowidct_p(w, pin, pout, part_rho, nrow, it, s)) != hipSuccess) return e;
      { GPA_PROF("pq_kernel", s);
        if (band == 4 && w->lat_ok && w->nprob <= 2 && npx <= ((size_t)1 << 20)) {
          if (V == 4)
            pq_small_kernel<T, 4, 4><<<dim3(gpq.x, gpq.y, w->nprob), 256, 0, s>>>(pout, (const T*)weight, n0, n1, (T*)w->q,
                                                                                 part_pq, w->flags, npx);
          else
            pq_small_kernel<T, 1, 4><<<dim3(gpq.x, gpq.y, w->nprob), 256, 0, s>>>(pout, (const T*)weight, n0, n1, (T*)w->q,
                                                                                 part_pq, w->flags, npx);
        } else if (V == 4)
          pq_kernel<T, true, 4><<<dim3(gpq.x, gpq.y, w->nprob), 256, 0, s>>>(pout, nullptr, nullptr, (const T*)weight, n0,
                                                                             n1, (T*)w->q, part_pq, w->scal, w->flags,
                                                                             nullptr, 0, it, band, npx);
        else
          pq_kernel<T, true, 1><<<dim3(gpq.x, gpq.y, w->nprob), 256, 0, s>>>(pout, nullptr, nullptr, (const T*)weight, n0,
                                                                             n1, (T*)w->q, part_pq, w->scal, w->flags,
                                                                             nullptr, 0, it, band, npx); }
    }
    flush(kmax);
    return hipGetLastError();
  }
  // sizes without a fused path (no mixed-radix plan fits LDS): the Bluestein kernels, one vector update per kernel
  for (int it = 0; it < kmax; ++it) {
    int nrow = 0;
    if ((e = dispatch_g_rowdct(w, s)) != hipSuccess) return e;
    if ((e = dispatch_g_colsolve(w, compat, s)) != hipSuccess) return e;
    if ((e = dispatch_g_rowidct(w, &nrow, s)) != hipSuccess) return e;
    { GPA_PROF("scalar_kernels", s); scal_rho_kernel<<<1, 256, 0, s>>>(w->part, nrow, w->scal, w->flags); }
    T* pcur = (T*)w->p;
    { GPA_PROF("pupdate_kernel", s);
      pupdate_kernel<T><<<gl, 256, 0, s>>>((const T*)w->z, pcur, npx, w->scal, w->flags); }
    { GPA_PROF("applyq_kernel", s);
      applyq_kernel<T><<<g2, 256, 0, s>>>((const T*)pcur, (const T*)weight, n0, n1, (T*)w->q, w->part + MAXPART,
                                          w->flags); }
    scal_alpha_kernel<<<1, 256, 0, s>>>(w->part + MAXPART, np2, w->scal, w->flags);
    { GPA_PROF("update_kernel", s);
      update_kernel<T><<<gl, 256, 0, s>>>((const T*)pcur, (const T*)w->q, (T*)phi, (T*)w->r, npx, w->scal,
                                          w->part + 2 * MAXPART, w->flags); }
    scal_stop_kernel<<<1, 256, 0, s>>>(w->part + 2 * MAXPART, gl, w->scal, w->flags, kmax, eps);
  }
  return hipGetLastError();
}

hipError_t unwrap_enqueue(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight, bool from_psi,
                          int kmax, double eps, bool axes_compat, void* phi, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  return w->dtype == 0 ? run_pcg<float>(w, a, b, weight, from_psi, kmax, eps, axes_compat ? 1 : 0, phi, s)
                       : run_pcg<double>(w, a, b, weight, from_psi, kmax, eps, axes_compat ? 1 : 0, phi, s);
}

void* unwrap_residual_buffer(UnwrapWorkspace* ws, int problem) {
  Impl* w = (Impl*)ws->impl;
  return w ? (char*)w->r + (size_t)problem * w->n0 * w->n1 * w->rsz : nullptr;
}
double* unwrap_partials_buffer(UnwrapWorkspace* ws, int problem) {
  Impl* w = (Impl*)ws->impl;
  return w ? w->part + (size_t)problem * PART_N : nullptr;
}

hipError_t unwrap_enqueue_prepared(UnwrapWorkspace* ws, const void* weight, int nparts, int kmax, double eps,
                                   bool axes_compat, void* phi, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  if (nparts < 1 || nparts > MAXPART) return hipErrorInvalidValue;
  w->prepared_parts = nparts;
  return w->dtype == 0 ? run_pcg<float>(w, nullptr, nullptr, weight, false, kmax, eps, axes_compat ? 1 : 0, phi, s)
                       : run_pcg<double>(w, nullptr, nullptr, weight, false, kmax, eps, axes_compat ? 1 : 0, phi, s);
}

hipError_t unwrap_finish(UnwrapWorkspace* ws, int* iters_out, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  int it = 0;
  hipError_t e = hipMemcpyAsync(&it, w->flags + w->iters_slot, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  if (iters_out) *iters_out = it;
  return hipSuccess;
}

hipError_t unwrap_fetch_iters(UnwrapWorkspace* ws, int* host_pinned, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  // one problem: its count alone; several: the flag words of all of them, the count of problem j at
  // host_pinned[FLAGS_N * j + unwrap_iters_slot()]
  if (w->nprob == 1) return hipMemcpyAsync(host_pinned, w->flags + w->iters_slot, sizeof(int), hipMemcpyDeviceToHost, s);
  return hipMemcpyAsync(host_pinned, w->flags, (size_t)FLAGS_N * w->nprob * sizeof(int), hipMemcpyDeviceToHost, s);
}
// a workspace created for `cap` problems solves the first n of them (n <= cap): a stack whose last chunk is ragged,
// or a shorter stack, reuses the buffers instead of paying a destroy / create (hipMalloc, table upload) per change
bool unwrap_set_active(UnwrapWorkspace* ws, int n) {
  Impl* w = (Impl*)ws->impl;
  if (!w || n < 1 || n > w->cap) return false;
  w->nprob = n;
  return true;
}
int unwrap_iters_slot(const UnwrapWorkspace* ws) {
  const Impl* w = (const Impl*)ws->impl;
  return w ? w->iters_slot : 0;
}

hipError_t unwrap_run(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight, bool from_psi, int kmax,
                      double eps, bool axes_compat, void* phi, int* iters_out, hipStream_t s) {
  hipError_t e = unwrap_enqueue(ws, a, b, weight, from_psi, kmax, eps, axes_compat, phi, s);
  if (e != hipSuccess) return e;
  return unwrap_finish(ws, iters_out, s);
}

// ---------------------------------------------------------------------------
// arbitrary-size 2-D DFT + a9 helpers (exported)
// ---------------------------------------------------------------------------
hipError_t blue_axis_create(int dtype, int n, hipStream_t s, BlueAxis* out, size_t* bytes) {
  int lgb = 6;
  while ((1 << lgb) < 2 * n - 1) ++lgb;
  if (lgb > (dtype == 0 ? 14 : 13)) return hipErrorInvalidValue;
  const int L = 1 << lgb, tpf = L / 16;
  out->n = n;
  out->lg = lgb;
  std::vector<double> t((size_t)2 * L), ch((size_t)2 * n), bre((size_t)L, 0.0), bim((size_t)L, 0.0);
  for (int k = 0; k < L; ++k) { t[2 * k] = cos(-2.0 * M_PI * k / L); t[2 * k + 1] = sin(-2.0 * M_PI * k / L); }
  for (int m = 0; m < n; ++m) {
    const long long mm = ((long long)m * m) % (2LL * n);
    const double cr = cos(M_PI * (double)mm / n), ci = sin(M_PI * (double)mm / n);
    ch[2 * m] = cr; ch[2 * m + 1] = ci;
    bre[m] = cr; bim[m] = ci;
    if (m > 0) { bre[L - m] = cr; bim[L - m] = ci; }
  }
  host_fft(bre, bim);
  std::vector<double> bs((size_t)2 * L);
  for (int i = 0; i < 16; ++i)
    for (int tt = 0; tt < tpf; ++tt) {
      const int k = spec_index_rt(lgb, tt, i);
      bs[2 * ((size_t)i * tpf + tt)] = bre[k] / L;
      bs[2 * ((size_t)i * tpf + tt) + 1] = bim[k] / L;
    }
  size_t b = 0;
  hipError_t e;
  if ((e = upload(dtype, &out->tw, t, &b, s)) != hipSuccess) return e;
  if ((e = upload(dtype, &out->chirp, ch, &b, s)) != hipSuccess) return e;
  if ((e = upload(dtype, &out->bspec, bs, &b, s)) != hipSuccess) return e;
  if (bytes) *bytes += b;
  return hipSuccess;
}

void blue_axis_destroy(BlueAxis* a) {
  if (a->tw) hipFree(a->tw);
  if (a->chirp) hipFree(a->chirp);
  if (a->bspec) hipFree(a->bspec);
  a->tw = a->chirp = a->bspec = nullptr;
}

}  // namespace gpa
