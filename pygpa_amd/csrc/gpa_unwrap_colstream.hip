// a7, column half of the preconditioner z = idctn(dctn(r) / eig) (phase_unwrap.py:95-115) as a STREAM: the
// transform-free recursion of colsolve_tri_kernel (gpa_unwrap_cols.hip) without its need to hold a whole column in one
// workgroup.
//
// For row frequency j the column solve is (T + mu_j) z = r with T the second-difference matrix with reflecting ends; with
// lam = (1 + h) - sqrt(h (2 + h)), h = 1 - cos(pi j / N), it is the cascade
//     p_n = r_n + lam p_(n-1)            p_(-1) = (A + lam^N B) / (1 - lam^(2N)),  A = sum lam^m r_m,  B = sum lam^m r_(N-1-m)
//     z_n = lam (z_(n+1) - p_n)          z_N    = -lam / (1 - lam) p_(N-1)
// A column of 4096 .. 16384 points does not fit the registers of a workgroup together with enough neighbours to make
// its row segments long (the resident kernels are down to 32 / 16 / 8-byte segments at 4096 / 8192 / 16384 points and
// run at 0.25 / 0.12 / 0.07 of the HBM rate).  Both recursions are linear, so a column is cut into chunks of C rows and
// the solve becomes three launches that never hold more than C rows of a column:
//   colstream_agg_kernel    per chunk and column the two zero-start sums  a = sum lam^k r_k,  b = sum lam^(len-1-k) r_k
//                           (reads R once, writes 16 bytes per C samples)
//   colstream_scan_kernel   per column, over its S chunks: A, B, p_(-1); the p that enters every chunk; z_N; the z that
//                           enters every chunk from below (the zero-start z-sum of a chunk follows from a, b in closed
//                           form, see the kernel).  Also the stopping test of the iteration (phase_unwrap.py:348) and
//                           the singular column j = 0 (lam = 1: the reference divides its DC bin by 1, :110-114).
//   colstream_apply_kernel  per chunk and column: both recursions from the true carries, rho = <r, z> from z alone as the
//                           quadratic form -sum (z_(n+1) - z_n)^2 + mu sum z_n^2 (reads R again, writes Z)
// Every access is a 256-thread workgroup reading C rows of 256 adjacent columns: 1 KiB (f32) contiguous per row, any
// number of workgroups in flight.  12 bytes per sample instead of 8, at the streaming rate.
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

constexpr int CS_COLS = 256;   // columns per workgroup = threads per workgroup

// per-column constants of the scan (doubles; built on the host in long double)
struct StreamCol {
  double lam, lamC, lamL, lamN, inv, zn, g, q2;   // lam^C, lam^len(last chunk), lam^N, 1 / (1 - lam^(2N)), -lam / (1 - lam),
                                                  // g = lam / (1 - lam^2), q2 = lam^2 / (1 - lam^2); column 0: lam = 1, rest 0
};

// exclusive prefix sum over the threads of a workgroup in thread order (fixed shuffle tree per wavefront, fixed order over
// the wavefronts: deterministic); *total = the sum over all threads.  sh: >= 17 doubles.
__device__ __forceinline__ double block_excl_scan(double v, double* sh, double* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  double inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  __syncthreads();   // sh may still be read from an earlier call
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  double front = 0.0, tot = 0.0;
  for (int i = 0; i < nw; ++i) {
    const double w = sh[i];
    if (i < wave) front += w;
    tot += w;
  }
  *total = tot;
  return front + inc - v;
}

// A sample is read by two recursions.  The second read goes through an opaque copy: hipcc otherwise shares the
// f32 -> f64 conversion between the passes and keeps all C converted samples of the thread alive (2 C registers more).
template <class T>
__device__ __forceinline__ double reread(T v) {
  if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(v));
  return (double)v;
}

// ---- launch 1: chunk sums -----------------------------------------------------------------------
// UPDATE (iterations >= 1 of the stencil-fused iteration, gpa_unwrap.hip): the kernel first applies the pending update of the
// row spectrum, R <- R - alpha D with D = DCT-II_rows(q) from pqdct_kernel and alpha = rho / <p, q> from that kernel's
// partial sums (phase_unwrap.py:343-345; by linearity the update of r is the update of its row spectrum), writes R back,
// and adds the partial ||r||^2 = (1 / 2N) sum_k c_k R_k^2 (Parseval, c_0 = 1/2) that the scan kernel's stopping test reads.
template <class T, int C, bool RAGGED, bool UPDATE>
__global__ __launch_bounds__(CS_COLS) void colstream_agg_kernel(T* __restrict__ R, const T* __restrict__ Dq, int n0, int n1,
                                                               const double* __restrict__ lamtab, double2* __restrict__ agg,
                                                               double2* __restrict__ agg0, int S, const int* flags,
                                                               const double* part_pq, int npq, double* scal, int it, int ring,
                                                               double* part_norm, size_t pimg, size_t pagg) {
  {
    const size_t pb = blockIdx.z;
    R += pb * pimg;
    if (UPDATE) Dq += pb * pimg;
    agg += pb * pagg;
    agg0 += pb * S;
    flags += pb * FLAGS_N;
    if (UPDATE) {
      part_pq += pb * PART_N;
      part_norm += pb * PART_N;
      scal += pb * SCAL_N;
    }
  }
  // (stopped in an EARLIER iteration?  this iteration's test is the scan kernel's.  The flag is requested with the data and
  //  looked at after the loads have been issued: one memory round trip, not two)
  const int stopped = flags[1];
  const int y = blockIdx.x * CS_COLS + threadIdx.x, s = blockIdx.y;
  const bool cv = !RAGGED || y < n1;
  const int yc = cv ? y : 0;
  const int row0 = s * C;
  const int len = !RAGGED ? C : (n0 - row0 < C ? n0 - row0 : C);
  // (a wave-uniform row pointer plus the lane's 32-bit column: the scalar-base form of the load, no 64-bit address
  //  per row in vector registers)
  const unsigned yo = (unsigned)yc * (unsigned)sizeof(T);   // byte offset, 32 bits: base (scalar) + zext(offset)
  T x[C];
  {
    const T* rp = R + (size_t)row0 * n1;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      x[k] = (!RAGGED || k < len) ? *reinterpret_cast<const T*>(reinterpret_cast<const char*>(rp) + yo) : T(0);
      rp += n1;
    }
  }
  const double lam = lamtab[yc];
  [[maybe_unused]] __shared__ double sh[32];
  if constexpr (UPDATE) {
    double pq_part = load_partials(part_pq, npq);
    const double rho = scal[8 + ((it - 1) & 1)];
    if (stopped) return;
    const double pq = block_sum(pq_part, sh);
    const double alpha_d = rho / pq;   // phase_unwrap.py:343
    const T alpha = (T)alpha_d;
    // (phi += alpha p is not applied here: alpha is filed for phi_flush_kernel)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
    const T* dp = Dq + (size_t)row0 * n1;
    T* wp = R + (size_t)row0 * n1;
    double sq = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < C; k0 += 16) {
      T d[16];
#pragma unroll
      for (int j = 0; j < 16; ++j)
        d[j] = (!RAGGED || k0 + j < len) ? *reinterpret_cast<const T*>(reinterpret_cast<const char*>(dp + (size_t)(k0 + j) * n1) + yo) : T(0);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = k0 + j;
        x[k] = x[k] - alpha * d[j];
        if ((!RAGGED || k < len) && cv) *reinterpret_cast<T*>(reinterpret_cast<char*>(wp + (size_t)k * n1) + yo) = x[k];
        sq += (double)x[k] * (double)x[k];
      }
    }
    if (y == 0) sq *= 0.5;
    const double tot = block_sum(cv ? sq : 0.0, sh);
    if (threadIdx.x == 0) part_norm[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot / (2.0 * (double)n1);
  } else {
    if (stopped) return;
  }
  double b = 0.0, a = 0.0;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if (!RAGGED || k < len) b = (double)x[k] + lam * b;
    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // (see colstream_apply_kernel)
  }
#pragma unroll
  for (int k = C - 1; k >= 0; --k) {
    if (!RAGGED || k < len) a = reread(x[k]) + lam * a;
    if ((k & 7) == 0) __builtin_amdgcn_sched_barrier(0);
  }
  if (cv) agg[(size_t)s * n1 + y] = make_double2(a, b);
  if (y == 0) {
    // column 0 (lam = 1, the row means) is solved on r - mean with the mean of z removed afterwards (scan kernel): that
    // needs, of the zero-start running sum p0_k of the chunk, c = sum_k p0_k and d = sum_k (k + 1) p0_k as well
    double p0 = 0.0, c = 0.0, d = 0.0;
#pragma unroll
    for (int k = 0; k < C; ++k)
      if (!RAGGED || k < len) {
        p0 += reread(x[k]);
        c += p0;
        d += (double)(k + 1) * p0;
      }
    agg0[s] = make_double2(c, d);
  }
}

// ---- launch 2: the scan over the chunks of every column ------------------------------------------
// Threads are laid out (column, segment): G segments of M consecutive chunks per column, G side by side in the thread
// index's high part so that a wavefront reads 64 adjacent columns.  Three linear recurrences over the chunks --
//   A (from the last chunk up), B and the chunk-end p (from the first chunk down), the chunk-start z (from the last up)
// -- each as: compose the segment's chunks (registers), exchange the G segment results through LDS, apply.
template <int G, int M>
__global__ __launch_bounds__(64 * G) void colstream_scan_kernel(int n0, int n1, int C, int S, const StreamCol* __restrict__ tab,
                                                               const double2* __restrict__ agg, const double2* __restrict__ agg0,
                                                               double* __restrict__ carP, double* __restrict__ carZ, double* __restrict__ col0,
                                                               int* flags, const double* part_norm, int nnorm, int it, double eps,
                                                               double* scal, double* part_rho, int rho_slot, size_t pimg,
                                                               size_t pagg, int f32) {
  {
    const size_t pb = blockIdx.z;
    agg += pb * pagg;
    agg0 += pb * S;
    carP += pb * pagg;
    carZ += pb * pagg;
    col0 += pb * 2;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_norm += pb * PART_N;
    part_rho += pb * PART_N;
  }
  // every input -- flag, partial norms, scalars, the column's constants and chunk sums -- is requested before the first
  // wait: the kernel is a chain of dependent round trips otherwise (65 workgroups at 4096^2: nothing to overlap them with)
  const int stopped = flags[1];
  __shared__ double shn[64 * G];
  __shared__ double xa[G][64], xb[G][64], xm[G][64];
  double norm_part = 0.0, best = 0.0, norm0 = 0.0;
  if (it > 0) {
    norm_part = load_partials(part_norm, nnorm);
    best = scal[10 + ((it - 1) & 1)];
    norm0 = scal[5];
  }
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int ncb = (n1 + 63) / 64;   // column blocks; the block after them solves column 0
  const bool colblock = (int)blockIdx.x < ncb;
  const int y = blockIdx.x * 64 + lane;
  const bool cv = colblock && y < n1 && y > 0;     // (column 0 is the extra workgroup's)
  const int yc = (colblock && y < n1) ? y : 0;
  const StreamCol tc = tab[yc];
  const int s0 = g * M;                // first chunk of this thread's segment
  double av[M], bv[M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int s = s0 + i;
    const double2 v = (s < S && colblock) ? agg[(size_t)s * n1 + yc] : make_double2(0.0, 0.0);
    av[i] = v.x;
    bv[i] = v.y;
  }
  if (stopped) return;
  if (it > 0) {
    // the reference's stopping test (phase_unwrap.py:348) on the update the row kernel has just applied, evaluated by
    // every workgroup of this launch; the other two launches of the solve read the flag
    const double tot = block_sum(norm_part, shn);
    double stall;
    const bool stop = sqrt(tot) < eps * sqrt(norm0) || tot == 0.0 || pcg_breakdown(tot, best, norm0, f32 != 0, scal[SC_STALL + ((it - 1) & 1)], &stall, scal[SC_STALL_LIMIT]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[0] = it;
      scal[6] = tot;
      scal[10 + (it & 1)] = tot < best ? tot : best;
      scal[SC_STALL + (it & 1)] = stall;
      if (stop) flags[1] = 1;
    }
    if (stop) return;
  }
  if (!colblock) {
    // ---- column 0 (lam = 1, the row means): z = T^+ (r - mean) + mean, i.e. p = cumsum(r - mean), z_n = -sum_(m >= n) p_m,
    // then the mean of z removed and the mean of r added (the reference divides its DC bin by 1, phase_unwrap.py:110-114).
    // From the chunk sums b = sum r, c, d of colstream_agg_kernel; one wavefront, lane i owns chunks [i NJ, (i + 1) NJ).
    if (threadIdx.x >= 64) return;
    constexpr int NJ = (G * M + 63) / 64;
    auto wave_incl = [&](double v) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off);
        if (lane >= off) v += t;
      }
      return v;
    };
    double bj[NJ], cj[NJ], dj[NJ], lj[NJ];
    double bsum = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int sc = lane * NJ + j;
      const bool v = sc < S;
      const double2 cd = v ? agg0[sc] : make_double2(0.0, 0.0);
      bj[j] = v ? agg[(size_t)sc * n1].y : 0.0;
      cj[j] = cd.x;
      dj[j] = cd.y;
      lj[j] = !v ? 0.0 : (double)(sc == S - 1 ? n0 - (S - 1) * C : C);
      bsum += bj[j];
    }
    const double shift0 = __shfl(wave_incl(bsum), 63) / (double)n0;
    // p entering every chunk
    double lp = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) lp += bj[j] - lj[j] * shift0;
    double Pin = wave_incl(lp) - lp;
    double pin[NJ], ps[NJ], ws = 0.0, lps = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int sc = lane * NJ + j;
      const double len = lj[j], tri = 0.5 * len * (len + 1.0);
      pin[j] = Pin;
      ps[j] = len * Pin + cj[j] - shift0 * tri;                                   // sum of p over the chunk
      ws += (double)sc * (double)C * ps[j] + dj[j] - shift0 * tri * (2.0 * len + 1.0) / 3.0 + Pin * tri;   // sum (n + 1) p_n
      lps += ps[j];
      Pin += bj[j] - len * shift0;
    }
    const double incl = wave_incl(lps), total = __shfl(incl, 63);
    double after = total - incl;                         // sum of p over the chunks of the lanes behind this one
    const double zs = -__shfl(wave_incl(ws), 63);        // sum_n z_n
    const double fix = shift0 - zs / (double)n0;
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      const int sc = lane * NJ + j;
      if (sc < S) {
        carP[(size_t)sc * n1] = pin[j];
        carZ[(size_t)sc * n1] = -after;                  // z entering the chunk from below
      }
      after += ps[j];
    }
    if (lane == 0) {
      col0[0] = shift0;
      col0[1] = fix;
      // (colsolve_tri_kernel: rho of column 0 = (-sum p^2 + N shift0^2) c_0, c_0 = 1/2; the sum is the apply kernel's)
      part_rho[rho_slot] = 0.5 * (double)n0 * shift0 * shift0 / (2.0 * (double)n1);
    }
    return;
  }
  auto mult = [&](int s) { return s == S - 1 ? tc.lamL : tc.lamC; };   // lam^len(s)
  // segment results: A-part (from the segment's last chunk up), B-part (down), total multiplier
  double Aseg = 0.0, Bseg = 0.0, Mseg = 1.0;
#pragma unroll
  for (int i = M - 1; i >= 0; --i)
    if (s0 + i < S) Aseg = av[i] + mult(s0 + i) * Aseg;
#pragma unroll
  for (int i = 0; i < M; ++i)
    if (s0 + i < S) { Bseg = bv[i] + mult(s0 + i) * Bseg; Mseg *= mult(s0 + i); }
  xa[g][lane] = Aseg;
  xb[g][lane] = Bseg;
  xm[g][lane] = Mseg;
  __syncthreads();
  double A = 0.0, B = 0.0;
#pragma unroll
  for (int j = G - 1; j >= 0; --j) A = xa[j][lane] + xm[j][lane] * A;
#pragma unroll
  for (int j = 0; j < G; ++j) B = xb[j][lane] + xm[j][lane] * B;
  const double pm1 = (A + tc.lamN * B) * tc.inv;       // p_(-1)
  // p at the end of the segments before this one, and at the end of the column
  double P = pm1, Plast = pm1;
#pragma unroll
  for (int j = 0; j < G; ++j) {
    Plast = xb[j][lane] + xm[j][lane] * Plast;
    if (j < g) P = Plast;
  }
  // forward over the segment's chunks: the p that enters each; and its zero-start z-sum
  //   e_s = -sum_k lam^(k+1) p_k,  p_k = p0_k + lam^(k+1) Pin  =>  e_s = -g (a_s - lam^(len+1) b_s) - q2 (1 - lam^(2 len)) Pin
  // (p0 the zero-start recursion of the chunk; the first term is that recursion's z-sum in closed form)
  // (the entering p goes to memory at once and e_s replaces a_s in its register: two arrays of M doubles live, not four)
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const int s = s0 + i;
    if (s < S) {
      if (cv) carP[(size_t)s * n1 + y] = P;
      const double ml = mult(s);
      av[i] = -tc.g * (av[i] - tc.lam * ml * bv[i]) - tc.q2 * (1.0 - ml * ml) * P;
      P = bv[i] + ml * P;
    } else {
      av[i] = 0.0;
    }
  }
  double Eseg = 0.0;
#pragma unroll
  for (int i = M - 1; i >= 0; --i)
    if (s0 + i < S) Eseg = av[i] + mult(s0 + i) * Eseg;
  __syncthreads();   // xa is read above by every thread
  xa[g][lane] = Eseg;
  __syncthreads();
  double Zc = tc.zn * Plast;                           // z_N
#pragma unroll
  for (int j = G - 1; j >= 0; --j)
    if (j > g) Zc = xa[j][lane] + xm[j][lane] * Zc;
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    const int s = s0 + i;
    if (s < S) {
      if (cv) carZ[(size_t)s * n1 + y] = Zc;
      Zc = av[i] + mult(s) * Zc;
    }
  }
}

// ---- launch 3: the recursions, chunk by chunk ---------------------------------------------------
template <class T, int C, bool RAGGED>
__global__ __launch_bounds__(CS_COLS) void colstream_apply_kernel(const T* __restrict__ R, T* __restrict__ Z, int n0, int n1, int S,
                                                                 const double* __restrict__ lamtab, const T* __restrict__ hb,
                                                                 const double* __restrict__ carP,
                                                                 const double* __restrict__ carZ, const double* __restrict__ col0,
                                                                 const int* flags, double* part_rho, size_t pimg, size_t pagg) {
  {
    const size_t pb = blockIdx.z;
    R += pb * pimg;
    Z += pb * pimg;
    carP += pb * pagg;
    carZ += pb * pagg;
    col0 += pb * 2;
    flags += pb * FLAGS_N;
    part_rho += pb * PART_N;
  }
  const int stopped = flags[1];   // (requested with the data, see colstream_agg_kernel)
  __shared__ double shn[CS_COLS];
  const int y = blockIdx.x * CS_COLS + threadIdx.x, s = blockIdx.y;
  const bool cv = !RAGGED || y < n1;
  const int yc = (!RAGGED || y < n1) ? y : 0;
  const int row0 = s * C;
  const int len = !RAGGED ? C : (n0 - row0 < C ? n0 - row0 : C);
  const T* rp = R + (size_t)row0 * n1;
  const unsigned yo = (unsigned)yc * (unsigned)sizeof(T);   // byte offset, 32 bits: base (scalar) + zext(offset)
  T x[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    x[k] = (!RAGGED || k < len) ? *reinterpret_cast<const T*>(reinterpret_cast<const char*>(rp) + yo) : T(0);
    rp += n1;
  }
  const double lam = lamtab[yc];
  const double cp = carP[(size_t)s * n1 + yc], cz = carZ[(size_t)s * n1 + yc];
  const double mu2 = 2.0 * (double)hb[yc];
  // column 0: r - mean goes in, z + (mean of r - mean of z) comes out (both from the scan kernel); any other column: 0, 0
  const double sh0 = y == 0 ? col0[0] : 0.0, fx0 = y == 0 ? col0[1] : 0.0;
  if (stopped) return;
  // causal recursion from the true carry, in place (stored in the data's precision, as colsolve_tri_kernel does)
  // (scheduling fences every 8 rows: left alone, hipcc hoists the f32 -> f64 conversions of all C samples to the top of
  //  the chain and keeps 2 C more registers alive)
  double p = cp;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if (!RAGGED || k < len) {
      p = ((double)x[k] - sh0) + lam * p;
      x[k] = (T)p;
    }
    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  // anticausal recursion in place; rho from the quadratic form (no difference across the reflecting end)
  double z = cz, dsq = 0.0, zsq = 0.0;
  const bool last = s == S - 1;
#pragma unroll
  for (int k = C - 1; k >= 0; --k) {
    if (!RAGGED || k < len) {
      const double zn = lam * (z - reread(x[k]));
      if (!(last && k == len - 1)) dsq += (z - zn) * (z - zn);
      zsq += zn * zn;
      z = zn;
      x[k] = (T)(zn + fx0);
    }
    if ((k & 7) == 0) {
      // (the two sums are pinned here: hipcc otherwise sinks their whole accumulation behind the stores and keeps every
      //  z of the chunk alive as a double until then)
      asm volatile("" : "+v"(dsq), "+v"(zsq));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (cv) {
    T* zp = Z + (size_t)row0 * n1;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      if (!RAGGED || k < len) *reinterpret_cast<T*>(reinterpret_cast<char*>(zp) + yo) = x[k];
      zp += n1;
    }
  }
  const double rr = (-dsq - mu2 * zsq) * (y == 0 ? 0.5 : 1.0);   // (c_0 = 1/2 of SciPy's DCT-II normalisation along the rows)
  const double tot = block_sum(cv ? rr : 0.0, shn);
  if (threadIdx.x == 0) part_rho[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot / (2.0 * (double)n1);
}

// upd != nullptr: the stencil-fused iteration -- the first launch applies R -= alpha D before it sums (see the kernel)
struct StreamUpdate {
  const void* dq;
  const double* part_pq;
  int npq, ring;
  double* part_norm_out;
  int* nnorm_out;
};
template <class T, int C>
hipError_t run_stream(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it, double eps,
                      double* part_rho, int* nrho, const void* zin, const StreamUpdate* upd = nullptr) {
  const int n0 = w->n0, n1 = w->n1, S = w->strS;
  const size_t pimg = (size_t)n0 * n1, pagg = (size_t)S * n1;
  const bool ragged = (n0 % C) != 0 || (n1 % CS_COLS) != 0;
  const dim3 grid((n1 + CS_COLS - 1) / CS_COLS, S, w->nprob);
  const int nparts = grid.x * grid.y;
  if (nparts + 1 > MAXPART) return hipErrorInvalidValue;
  T* R = (T*)(zin ? const_cast<void*>(zin) : w->z);
  // column 0's own chunk sums and its two scalars live behind the carries: [cap][S] double2, [cap][2] doubles
  double* carP = (double*)w->strcar;
  double* carZ = carP + w->cap * pagg;
  double2* agg0 = (double2*)(carZ + w->cap * pagg);
  double* col0 = (double*)(agg0 + (size_t)w->cap * S);
  {
    GPA_PROF("colstream_agg_kernel", s);
    if (upd) {
      if (nparts > MAXPART) return hipErrorInvalidValue;
      *upd->nnorm_out = nparts;
      part_norm = upd->part_norm_out;
      nnorm = nparts;
      if (ragged) colstream_agg_kernel<T, C, true, true><<<grid, CS_COLS, 0, s>>>(R, (const T*)upd->dq, n0, n1, w->strlam, (double2*)w->stragg, agg0, S, w->flags, upd->part_pq, upd->npq, w->scal, it, upd->ring, upd->part_norm_out, pimg, pagg);
      else colstream_agg_kernel<T, C, false, true><<<grid, CS_COLS, 0, s>>>(R, (const T*)upd->dq, n0, n1, w->strlam, (double2*)w->stragg, agg0, S, w->flags, upd->part_pq, upd->npq, w->scal, it, upd->ring, upd->part_norm_out, pimg, pagg);
    } else {
      if (ragged) colstream_agg_kernel<T, C, true, false><<<grid, CS_COLS, 0, s>>>(R, nullptr, n0, n1, w->strlam, (double2*)w->stragg, agg0, S, w->flags, nullptr, 0, nullptr, it, 0, nullptr, pimg, pagg);
      else colstream_agg_kernel<T, C, false, false><<<grid, CS_COLS, 0, s>>>(R, nullptr, n0, n1, w->strlam, (double2*)w->stragg, agg0, S, w->flags, nullptr, 0, nullptr, it, 0, nullptr, pimg, pagg);
    }
  }
  {
    GPA_PROF("colstream_scan_kernel", s);
    const dim3 gs((n1 + 63) / 64 + 1, 1, w->nprob);
    // segments per column x chunks per segment: G * M >= S
#define GPA_SCAN(GG, MM)                                                                                                   \
  colstream_scan_kernel<GG, MM><<<gs, 64 * GG, 0, s>>>(n0, n1, C, S, (const StreamCol*)w->strtab, (const double2*)w->stragg, \
                                                          agg0, carP, carZ, col0, w->flags, part_norm, nnorm, it, eps, w->scal, \
                                                          part_rho, nparts, pimg, pagg, w->dtype == 0 ? 1 : 0)
    if (S <= 16) GPA_SCAN(4, 4);
    else if (S <= 32) GPA_SCAN(8, 4);
    else if (S <= 64) GPA_SCAN(8, 8);
    else if (S <= 128) GPA_SCAN(16, 8);
    else if (S <= 256) GPA_SCAN(16, 16);
    else return hipErrorInvalidValue;
#undef GPA_SCAN
  }
  {
    GPA_PROF("colstream_apply_kernel", s);
    if (ragged)
      colstream_apply_kernel<T, C, true><<<grid, CS_COLS, 0, s>>>(R, (T*)w->z, n0, n1, S, w->strlam, (const T*)w->hb1[compat],
                                                                 carP, carZ, col0, w->flags, part_rho, pimg, pagg);
    else
      colstream_apply_kernel<T, C, false><<<grid, CS_COLS, 0, s>>>(R, (T*)w->z, n0, n1, S, w->strlam, (const T*)w->hb1[compat],
                                                                  carP, carZ, col0, w->flags, part_rho, pimg, pagg);
  }
  if (nrho) *nrho = nparts + 1;
  return hipGetLastError();
}

}  // namespace

// rows per chunk of the streamed column solve for a column of n0 points (0: not offered)
int colstream_chunk(int n0, int n1) {
  if (n0 != n1 || n0 < 64) return 0;
  int C = n0 >= 8192 ? 128 : (n0 >= 1024 ? 64 : 32);
  if (opt_set(OPT_COLSTREAM_CHUNK)) {
    const int c = (int)opt(OPT_COLSTREAM_CHUNK).num;
    if (c == 32 || c == 64 || c == 128) C = c;
  }
  while (C > 32 && (n0 + C - 1) / C < 4) C /= 2;
  if ((n0 + C - 1) / C > 256) return 0;
  return C;
}

// per-column constants (long double on the host) and the chunk-sum / carry buffers
hipError_t build_streamtab(Impl* w, hipStream_t s, size_t* bytes) {
  const int n0 = w->n0, n1 = w->n1;
  const int C = colstream_chunk(n0, n1);
  if (!C) return hipSuccess;
  const int S = (n0 + C - 1) / C, L = n0 - (S - 1) * C;
  std::vector<StreamCol> tc((size_t)n1);
  std::vector<double> lam((size_t)n1);
  for (int j = 0; j < n1; ++j) {
    if (j == 0) { tc[0] = {1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0}; lam[0] = 1.0; continue; }
    const long double sj = sinl((long double)M_PI * j / (2.0L * n1)), h = 2 * sj * sj;
    const long double l = (1 + h) - sqrtl(h * (2 + h));
    tc[j].lam = (double)l;
    tc[j].lamC = (double)powl(l, C);
    tc[j].lamL = (double)powl(l, L);
    tc[j].lamN = (double)powl(l, (long double)n0);
    tc[j].inv = (double)(1.0L / (1.0L - powl(l, 2.0L * n0)));
    tc[j].zn = (double)(-l / (1.0L - l));
    tc[j].g = (double)(l / (1.0L - l * l));
    tc[j].q2 = (double)(l * l / (1.0L - l * l));
    lam[j] = (double)l;
  }
  hipError_t e = hipMalloc(&w->strtab, tc.size() * sizeof(StreamCol));
  if (e == hipSuccess) e = hipMalloc((void**)&w->strlam, lam.size() * sizeof(double));
  const size_t ab = (size_t)S * n1 * 2 * sizeof(double) * w->cap;
  if (e == hipSuccess) e = hipMalloc(&w->stragg, ab);
  if (e == hipSuccess) e = hipMalloc(&w->strcar, ab + (size_t)w->cap * (S * 2 + 2) * sizeof(double));
  if (e != hipSuccess) return e;
  *bytes += tc.size() * sizeof(StreamCol) + lam.size() * sizeof(double) + 2 * ab;
  e = hipMemcpyAsync(w->strtab, tc.data(), tc.size() * sizeof(StreamCol), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(w->strlam, lam.data(), lam.size() * sizeof(double), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  w->strC = C;
  w->strS = S;
  return e;
}

// the stencil-fused iteration's column solve: R (w->r) -= alpha DCT_rows(q) (dq, from pqdct_kernel) first, then the solve;
// part_norm_out / nnorm_out: the partial ||r||^2 of the updated residual (read by this call's own scan kernel)
hipError_t dispatch_colstream_update(const Impl* w, int compat, hipStream_t s, int it, double eps, int ring, const void* dq,
                                     const double* part_pq, int npq, double* part_norm_out, int* nnorm_out, double* part_rho,
                                     int* nrho) {
  if (!w->strtab) return hipErrorInvalidValue;
  const StreamUpdate upd = {dq, part_pq, npq, ring, part_norm_out, nnorm_out};
#define GPA_STREAMU(CC)                                                                                                \
  return w->dtype == 0 ? run_stream<float, CC>(w, compat, s, nullptr, 0, it, eps, part_rho, nrho, w->r, &upd)          \
                       : run_stream<double, CC>(w, compat, s, nullptr, 0, it, eps, part_rho, nrho, w->r, &upd)
  switch (w->strC) {
    case 32: GPA_STREAMU(32);
    case 64: GPA_STREAMU(64);
    case 128: GPA_STREAMU(128);
  }
#undef GPA_STREAMU
  return hipErrorInvalidValue;
}

hipError_t dispatch_colstream(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it, double eps,
                              double* part_rho, int* nrho, const void* zin) {
  if (!w->strtab) return hipErrorInvalidValue;
#define GPA_STREAM(CC)                                                                                                 \
  return w->dtype == 0 ? run_stream<float, CC>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)           \
                       : run_stream<double, CC>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
  switch (w->strC) {
    case 32: GPA_STREAM(32);
    case 64: GPA_STREAM(64);
    case 128: GPA_STREAM(128);
  }
#undef GPA_STREAM
  return hipErrorInvalidValue;
}

}  // namespace gpa
