// a7, column half of the preconditioner z = idctn(dctn(r) / eig) (phase_unwrap.py:95-115) on the row spectra:
//   colsolve_kernel      DCT-II -> divide by the Laplacian eigenvalues -> DCT-III along the columns, never leaving LDS
//   colsolve_tri_kernel  square images: the same solve as a causal + anticausal first-order recursion (no transform)
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

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
    double stall;
    const bool stop = sqrt(tot) < eps * sqrt(norm0) || tot == 0.0 || pcg_breakdown(tot, best, norm0, sizeof(T) == 4, scal[SC_STALL + ((it - 1) & 1)], &stall, scal[SC_STALL_LIMIT]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[0] = it;                                   // updates completed
      scal[6] = tot;
      scal[10 + (it & 1)] = tot < best ? tot : best;
      scal[SC_STALL + (it & 1)] = stall;
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
    double stall;
    const bool stop = sqrt(tot) < eps * sqrt(scal[5]) || tot == 0.0 || pcg_breakdown(tot, best, scal[5], sizeof(T) == 4, scal[SC_STALL + ((it - 1) & 1)], &stall, scal[SC_STALL_LIMIT]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[0] = it;
      scal[6] = tot;
      scal[10 + (it & 1)] = tot < best ? tot : best;
      scal[SC_STALL + (it & 1)] = stall;
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
template <class T, int VEC, int Q, int R>
hipError_t run_colsolve_tri(const Impl* w, int S, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                            double eps, double* part_rho, int* nrho, const void* zin) {
  const bool ragged = w->generic;
  const int threads = S * Q, grid = (w->n1 + Q * VEC - 1) / (Q * VEC);
  const size_t lds = (size_t)16 * Q * VEC * sizeof(double);
  if (nrho) *nrho = grid;
  GPA_PROF("colsolve_tri_kernel", s);
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
}  // namespace

bool colstream_is_default(const Impl* w) {
  return w->strtab && (w->col_mode == 3 || (w->col_mode == 0 && w->n0 >= GPA_COLSTREAM_MIN));
}
hipError_t dispatch_colsolve(const Impl* w, int compat, hipStream_t s, const double* part_norm, int nnorm, int it,
                             double eps, double* part_rho, int* nrho, const void* zin) {
  // Square images from 4096 points a side: the streamed recursion (gpa_unwrap_colstream.hip) -- three launches that
  // read 1 KiB row pieces at the streaming rate instead of one that holds whole columns and is down to 32 / 16 / 8-byte
  // pieces at 4096 / 8192 / 16384 points.  COLSOLVE=stream forces it wherever it is offered, =tri / =fft the resident kernels.
  if (part_rho && colstream_is_default(w))
    return dispatch_colstream(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
  if (w->generic && part_rho) {
    // smooth sizes: the transform-free solve where it applies (square images; it is 2-3x faster than two mixed-radix
    // transforms per column pair), COLSOLVE=fft keeps the transforms
    if (w->tritab && w->col_mode != 2)
      return w->dtype == 0 ? dispatch_colsolve_tri<float>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin)
                           : dispatch_colsolve_tri<double>(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
    return mr_colsolve(w, compat, s, part_norm, nnorm, it, eps, part_rho, nrho, zin);
  }
  // Square images can solve the columns without a transform (colsolve_tri_kernel).  Measured at 4096^2 on MI355X
  // (profiles/r02_colsolve_tri.txt): f64 1.54 ms per step against 2.0 for the DCT kernel (whose f64 transforms
  // spill), f32 82 us per launch against 68 -- the f32 DCT kernel is the faster one.  So: f64 by default,
  // COLSOLVE=tri / fft forces one or the other (tests compare the two).
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

}  // namespace gpa
