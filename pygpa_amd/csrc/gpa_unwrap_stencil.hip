// a7, the kernels of the weighted unwrap that are not transforms: set-up r0 = A^T W^2 wrap(b) (phase_unwrap.py:296-316),
// the stencil q = A^T W^2 A p (:118-132) with the partial <p, q>, the deferred phi += sum alpha_j p_j (:344), and the
// scalar / elementwise kernels of the plain one-update-per-kernel scheme that sizes without a fused path run (:326-349).
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

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
    scal[SC_STALL] = 0.0;
    scal[SC_STALL + 1] = 0.0;
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
__global__ void scal_stop_kernel(const double* part, int nparts, double* scal, int* flags, int kmax, double eps, int f32) {
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
    double stall;
    if (pcg_breakdown(tot, scal[7], scal[5], f32 != 0, scal[SC_STALL], &stall, scal[SC_STALL_LIMIT])) flags[1] = 1;
    scal[SC_STALL] = stall;
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
    if (q) q += pb * pimg;          // (q == nullptr: the last iteration of a solve needs <p, q> only)
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
      if (q) *reinterpret_cast<VecN<T, V>*>(q + o) = qv;
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
}  // namespace

namespace {
// grid of the stencil kernel: rows of whole 4-pixel vectors take 16-byte accesses (V = 4), any other row length the
// one-pixel instantiation; band height 16 rows for large images, fewer when that would leave less than ~2048
// workgroups (small images are latency-, not bandwidth-bound)
struct PqGrid { int V, band; dim3 g; };
PqGrid pq_grid(const Impl* w) {
  PqGrid r;
  r.V = (w->n1 % 4) == 0 ? 4 : 1;
  const int pqcols = 256 * r.V;
  r.band = PQ_ROWS;
  while (r.band > 4 && (size_t)((w->n1 + pqcols - 1) / pqcols) * ((w->n0 + r.band - 1) / r.band) < 2048) r.band /= 2;
  r.g = dim3((w->n1 + pqcols - 1) / pqcols, (w->n0 + r.band - 1) / r.band);
  return r;
}

template <class T>
hipError_t setup_t(const Impl* w, const void* a, const void* b, const void* weight, bool from_psi, void* phi, hipStream_t s) {
  const dim3 gsu((w->n1 + 255) / 256, (w->n0 + SETUP_ROWS - 1) / SETUP_ROWS);
  const int nsu = gsu.x * gsu.y;
  if (nsu > MAXPART) return hipErrorInvalidValue;
  setup_kernel<T><<<gsu, 256, 0, s>>>((const T*)a, (const T*)b, (const T*)weight, from_psi ? 1 : 0, w->n0, w->n1, (T*)w->r,
                                      (T*)phi, w->part);
  scal_init_kernel<<<1, 256, 0, s>>>(w->part, nsu, w->scal, w->flags);
  return hipGetLastError();
}

template <class T>
hipError_t pq_t(const Impl* w, const void* p, const void* weight, int it, double* part_pq, hipStream_t s, bool need_q) {
  const PqGrid g = pq_grid(w);
  const int n0 = w->n0, n1 = w->n1;
  const size_t npx = (size_t)n0 * n1;
  const dim3 grid(g.g.x, g.g.y, w->nprob);
  GPA_PROF("pq_kernel", s);
  if (g.band == 4 && w->lat_ok && w->nprob <= 2 && npx <= ((size_t)1 << 20)) {
    if (g.V == 4) pq_small_kernel<T, 4, 4><<<grid, 256, 0, s>>>((const T*)p, (const T*)weight, n0, n1, (T*)w->q, part_pq, w->flags, npx);
    else pq_small_kernel<T, 1, 4><<<grid, 256, 0, s>>>((const T*)p, (const T*)weight, n0, n1, (T*)w->q, part_pq, w->flags, npx);
  } else if (g.V == 4) {
    pq_kernel<T, true, 4><<<grid, 256, 0, s>>>((const T*)p, nullptr, nullptr, (const T*)weight, n0, n1, need_q ? (T*)w->q : nullptr,
                                               part_pq, w->scal, w->flags, nullptr, 0, it, g.band, npx);
  } else {
    pq_kernel<T, true, 1><<<grid, 256, 0, s>>>((const T*)p, nullptr, nullptr, (const T*)weight, n0, n1, need_q ? (T*)w->q : nullptr,
                                               part_pq, w->scal, w->flags, nullptr, 0, it, g.band, npx);
  }
  return hipGetLastError();
}

template <class T>
hipError_t flush_t(const Impl* w, int ring, void* phi, bool phi_unwritten, int final_it, const double* part_pq, int npq,
                   hipStream_t s) {
  const size_t npx = (size_t)w->n0 * w->n1;
  const int gl = 2048;   // grid-stride
  RingPtrs<T> rp;
  for (int j = 0; j < RING_MAX; ++j) rp.p[j] = (const T*)w->ring[j < ring ? j : 0];
  {
    GPA_PROF("phi_flush_kernel", s);
    if ((w->n1 % 4) == 0)
      phi_flush_kernel<T, 4><<<dim3(gl, 1, w->nprob), 256, 0, s>>>(rp, ring, (T*)phi, npx / 4, w->scal, w->flags,
                                                                  phi_unwritten ? 1 : 0, npx, final_it, part_pq, npq);
    else
      phi_flush_kernel<T, 1><<<dim3(gl, 1, w->nprob), 256, 0, s>>>(rp, ring, (T*)phi, npx, w->scal, w->flags,
                                                                  phi_unwritten ? 1 : 0, npx, final_it, part_pq, npq);
  }
  if (!final_it) {
    GPA_PROF("scalar_kernels", s);
    phi_commit_kernel<<<dim3(1, 1, w->nprob), 1, 0, s>>>(w->flags);
  }
  return hipGetLastError();
}

// the plain scheme's iteration after its three DCT kernels: rho, p, q, alpha, phi / r, stopping test -- one kernel each
template <class T>
hipError_t plain_tail_t(const Impl* w, const void* weight, void* phi, int nrow, int kmax, double eps, hipStream_t s) {
  const int n0 = w->n0, n1 = w->n1;
  const size_t npx = (size_t)n0 * n1;
  const int gl = 2048;
  { GPA_PROF("scalar_kernels", s); scal_rho_kernel<<<1, 256, 0, s>>>(w->part, nrow, w->scal, w->flags); }
  T* pcur = (T*)w->p;
  { GPA_PROF("pupdate_kernel", s);
    pupdate_kernel<T><<<gl, 256, 0, s>>>((const T*)w->z, pcur, npx, w->scal, w->flags); }
  { GPA_PROF("applyq_kernel", s);
    applyq_kernel<T><<<n0, 256, 0, s>>>((const T*)pcur, (const T*)weight, n0, n1, (T*)w->q, w->part + MAXPART, w->flags); }
  scal_alpha_kernel<<<1, 256, 0, s>>>(w->part + MAXPART, n0, w->scal, w->flags);
  { GPA_PROF("update_kernel", s);
    update_kernel<T><<<gl, 256, 0, s>>>((const T*)pcur, (const T*)w->q, (T*)phi, (T*)w->r, npx, w->scal,
                                        w->part + 2 * MAXPART, w->flags); }
  scal_stop_kernel<<<1, 256, 0, s>>>(w->part + 2 * MAXPART, gl, w->scal, w->flags, kmax, eps, w->dtype == 0 ? 1 : 0);
  return hipGetLastError();
}
}  // namespace

int pq_partials(const Impl* w) {
  const PqGrid g = pq_grid(w);
  return g.g.x * g.g.y;
}
hipError_t launch_unwrap_setup(const Impl* w, const void* a, const void* b, const void* weight, bool from_psi, void* phi,
                               hipStream_t s) {
  return w->dtype == 0 ? setup_t<float>(w, a, b, weight, from_psi, phi, s) : setup_t<double>(w, a, b, weight, from_psi, phi, s);
}
hipError_t launch_scal_init(const Impl* w, int nparts, hipStream_t s) {
  GPA_PROF("scalar_kernels", s);
  scal_init_kernel<<<dim3(1, 1, w->nprob), 256, 0, s>>>(w->part, nparts, w->scal, w->flags);
  return hipGetLastError();
}
hipError_t launch_pq(const Impl* w, const void* p, const void* weight, int it, double* part_pq, hipStream_t s, bool need_q) {
  return w->dtype == 0 ? pq_t<float>(w, p, weight, it, part_pq, s, need_q) : pq_t<double>(w, p, weight, it, part_pq, s, need_q);
}
hipError_t launch_phi_flush(const Impl* w, int ring, void* phi, bool phi_unwritten, int final_it, const double* part_pq,
                            int npq, hipStream_t s) {
  return w->dtype == 0 ? flush_t<float>(w, ring, phi, phi_unwritten, final_it, part_pq, npq, s)
                       : flush_t<double>(w, ring, phi, phi_unwritten, final_it, part_pq, npq, s);
}
hipError_t launch_plain_tail(const Impl* w, const void* weight, void* phi, int nrow, int kmax, double eps, hipStream_t s) {
  return w->dtype == 0 ? plain_tail_t<float>(w, weight, phi, nrow, kmax, eps, s)
                       : plain_tail_t<double>(w, weight, phi, nrow, kmax, eps, s);
}

}  // namespace gpa
