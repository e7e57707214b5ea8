// a7, PERSISTENT row kernels for LONG f32 rows (8192 and 16384 points) of the fused PCG iteration (round 6;
// phase_unwrap.py:84-103, :326-349): the half-length transforms of gpa_unwrap_rowhalf.hip, software-pipelined like the
// 4096-point kernel of gpa_unwrap_rowpers.hip.
//
// One row per workgroup and launch slot leaves a 16384-point row kernel at 0.37-0.43 of the HBM rate: its 69.6 KB exchange
// buffer allows two workgroups per CU, each of which requests its row, waits, transforms (ten barriers), requests the
// previous search direction and the two phase tables, waits again, stores -- 34 us per row with at most two of those chains
// per CU to cover each other (1.09 / 0.95 ms per launch at 16384^2; without the table loads 0.91: profiles/r06_rowhalf_pers.txt).
// Here a workgroup is resident and walks a contiguous band of rows:
//   * the row of step j + 1 lands in the OTHER of two LDS buffers by LDS-DMA while row j is transformed (a buffer is first
//     the landing zone of a row, lane-linear as the DMA writes it, then the exchange buffer of its transform);
//   * the second operand of row j (the previous search direction / the kept spectrum) is requested into registers before the
//     transform and used after it; results leave as 16-byte stores nobody waits for (counted vmcnt);
//   * the two phase tables (w_k of the DCT, e^(-2 pi i k / N) of the real-to-complex split) live in registers for the
//     whole band instead of being read, 128 KB of them, per row.
// Barriers are raw s_barrier + lgkmcnt(0): __syncthreads() would drain the DMA in flight.
// Same arithmetic as rowidct_p_half_kernel / rowdct_half_kernel, same order of operations per value, both files compiled
// without contraction: bit-identical output (tests/test_gpu_unwrap_long.py holds them to that against NO_ROWPERS=1).
// (compiled with -ffp-contract=off: pygpa_amd/build.py, EXTRA_FLAGS, says why)
#include "gpa_unwrap_pers.h"

namespace gpa {
namespace {

// four reals at the alignment of ONE (global memory takes 16-byte accesses at 4-byte alignment): the partner bins N - k
template <class T> struct alignas(sizeof(T)) Vec4U {
  T v[4];
  __device__ __forceinline__ operator Vec4<T>() const { Vec4<T> r; r.v[0] = v[0]; r.v[1] = v[1]; r.v[2] = v[2]; r.v[3] = v[3]; return r; }
  __device__ __forceinline__ Vec4U& operator=(const Vec4<T>& o) { v[0] = o.v[0]; v[1] = o.v[1]; v[2] = o.v[2]; v[3] = o.v[3]; return *this; }
};

template <int LG>
struct HalfPersGeom {
  using T = float;
  static constexpr int E = 16;
  using F = WgFFT<T, LG - 1, E>;
  static constexpr int N = 1 << LG, HN = N / 2, TPF = F::TPF, P = F::P;
  static constexpr int BUF_BYTES = F::LDS_ELEMS * (int)sizeof(cpx<T>);   // exchange buffer >= the row it receives first
  static_assert(BUF_BYTES >= N * (int)sizeof(T), "a buffer holds a row");
  static constexpr int NV = N / (4 * TPF);                               // 16-byte vectors per thread and row
  static constexpr int NDMA = N * (int)sizeof(T) / (TPF * 16);           // DMA wave-instructions per wave and row (1 KB each)
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES;
  static constexpr int WGS_PER_CU = LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1;
  static constexpr int WAVES_PER_SIMD = WGS_PER_CU * TPF / 256;          // launch bound: 2 either way (256 VGPRs)
};

// one row -> buffer at LDS byte address dst, lane-linear (row: wave-uniform)
template <int LG>
__device__ __forceinline__ void dma_row(const float* __restrict__ row, unsigned dst, int wave, unsigned lane16) {
  using G = HalfPersGeom<LG>;
  constexpr int WAVES = G::TPF / 64;
  const unsigned long long base = (unsigned long long)row + (unsigned)wave * 1024u;
  const unsigned d0 = dst + (unsigned)wave * 1024u;
#pragma unroll
  for (int i = 0; i < G::NDMA; ++i) {
    const unsigned long long b = base + (unsigned long long)i * WAVES * 1024u;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    glds16s(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo), lane16,
            (unsigned)__builtin_amdgcn_readfirstlane((int)(d0 + i * WAVES * 1024)));
  }
}

// the pass-0 base twiddles load_twiddles() wrote have arrived (opaque uses: see gpa_unwrap_rowpers.hip)
template <class F>
__device__ __forceinline__ void settle_twiddles(typename F::TwiddlesLds& tw) {
  constexpr int r = 1 << F::bits(0), g = F::E / r;
#pragma unroll
  for (int q = 0; q < g; ++q)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if ((c + 1) < r) { settle(tw.lo[q][c].x); settle(tw.lo[q][c].y); }
      if (4 * (c + 1) < r) { settle(tw.hi[q][c].x); settle(tw.hi[q][c].y); }
    }
}
// ---------------------------------------------------------------------------
// row DCT-III of Z straight into p = z + beta p_prev (contract of rowidct_p_half_kernel)
// ---------------------------------------------------------------------------
template <int LG>
__global__ __launch_bounds__((HalfPersGeom<LG>::TPF), (HalfPersGeom<LG>::WAVES_PER_SIMD)) void rowidct_p_halfpers_kernel(
    const float* __restrict__ Z, const float* __restrict__ pin, float* __restrict__ pout, int n0,
    const cpx<float>* __restrict__ twh, const cpx<float>* __restrict__ twn, const cpx<float>* __restrict__ wk,
    const int* flags, const double* part_rho, int nrho, double* scal, int it, size_t pimg) {
  using T = float;
  using G = HalfPersGeom<LG>;
  using F = typename G::F;
  constexpr int TPF = G::TPF, N = G::N, HN = G::HN, E = G::E, NV = G::NV;
  {
    const size_t pb = blockIdx.z;
    Z += pb * pimg;
    pin += pb * pimg;
    pout += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_rho += pb * PART_N;
  }
  const int stop = flags[1];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[HalfPersGeom<LG>::TPF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const bool first = it == 0;
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  // the band of rows of this workgroup
  const int nwg = (int)gridDim.x;
  const int r0 = (int)((long long)blockIdx.x * n0 / nwg), r1 = (int)((long long)(blockIdx.x + 1) * n0 / nwg);
  if (r0 >= r1) return;
  const unsigned buf0 = lds_addr(smem);
  const unsigned lane16 = (unsigned)lane * 16u;
  dma_row<LG>(Z + (size_t)r0 * N, buf0, wave, lane16);
  // (base twiddles of the later passes from a small LDS table: the registers go to the two phase tables)
  __shared__ cpx<T> tws[F::LDS_TABLE_ELEMS];
  typename F::TwiddlesLds tw;
  F::fill_lds_tables(tws, twh, tid, TPF);
  F::load_twiddles(tw, twh, tid, tws);
  cpx<T> wkv[E], tnv[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    wkv[i] = wk[tid + TPF * i];
    tnv[i] = twn[tid + TPF * i];
  }
  // loop-invariant registers have ARRIVED before the loop (a wait inside it would be a vmcnt(0) that drains the DMA)
#pragma unroll
  for (int i = 0; i < E; ++i) { settle(wkv[i].x); settle(wkv[i].y); settle(tnv[i].x); settle(tnv[i].y); }
  settle_twiddles<F>(tw);
  __syncthreads();   // (tws)
  const T inv = T(1) / T(HN);
  for (int row = r0; row < r1; ++row) {
    // (index arithmetic from an opaque copy of the thread index: hipcc would otherwise hoist every LDS address and lane offset
    //  of the row out of the loop and keep them, dozens of registers, alive across it)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int cur = (row - r0) & 1;
    char* bcur = smem + cur * G::BUF_BYTES;
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(bcur);
    const T* st = reinterpret_cast<const T*>(bcur);
    const size_t o = (size_t)row * N;
    const T* prow = pin + o;
    T* qrow = pout + o;
    const unsigned t4 = 4u * (unsigned)tl;
    // this wave's pieces of the row have landed: everything older than the previous row's NV stores is done.  (Holds only
    // while the youngest vector-memory operations of an iteration are exactly those stores: tests/test_isa_invariants.py.)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");
    GPA_PBAR();   // ... and everybody else's; the other buffer's last reads (previous gather) are done
    if (row + 1 < r1) dma_row<LG>(Z + (size_t)(row + 1) * N, buf0 + (cur ^ 1) * G::BUF_BYTES, wave, lane16);
    // (the previous search direction in two halves: the second one is requested after the transform's first pass, whose
    //  sixteen-point butterflies are the register peak -- 16384-point rows run at the 256-register limit)
    Vec4<T> pv[NV];
    constexpr int NV0 = NV;
    if (!first) {
#pragma unroll
      for (int v = 0; v < NV0; ++v) pv[v] = *reinterpret_cast<const Vec4<T>*>(prow + (t4 + 4u * TPF * v));
    }
    // V_k = conj(w_k) (X_k - i X_(N-k)) / 2;  V_0 = X_0 / 2 and V_(N/2) = X_(N/2) / sqrt 2 are real
    cpx<T> x[E];
    T vh = T(0);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tl + TPF * i;
      const T zlo = st[k], zhi = st[k == 0 ? HN : N - k];
      if (k == 0) {
        x[i] = {T(0.5) * zlo, T(0)};
        vh = T(0.70710678118654752440) * zhi;
      } else {
        x[i] = cmulc(cpx<T>{T(0.5) * zlo, T(-0.5) * zhi}, wkv[i]);
      }
    }
    GPA_PBAR();   // every thread holds its inputs: the landing zone becomes the exchange buffer
#pragma unroll
    for (int i = 0; i < E; ++i) lds[F::pad(tl + TPF * i)] = x[i];
    GPA_PBAR();
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tl + TPF * i;
      cpx<T> Tk;
      if (k == 0) {
        Tk = {T(0.5) * (x[i].x + vh), T(0.5) * (x[i].x - vh)};
      } else {
        const cpx<T> vm = lds[F::pad(HN - k)];
        const cpx<T> ve = {T(0.5) * (x[i].x + vm.x), T(0.5) * (x[i].y - vm.y)};       // (V_k + conj V_m) / 2
        const cpx<T> d = {x[i].x - vm.x, x[i].y + vm.y};                              // V_k - conj V_m
        const cpx<T> vo = cscale(cmulc(d, tnv[i]), T(0.5));                           // conj(E_k) (.) / 2
        Tk = {ve.x - vo.y, ve.y + vo.x};                                              // Ve + i Vo
      }
      x[i] = {Tk.x, -Tk.y};                                                           // IFFT = conj(FFT(conj .))
    }
    GPA_PBAR();
    F::template fwd_phase<0>(x, lds, tl, tw);
    GPA_PBAR();
    if (!first) {
#pragma unroll
      for (int v = NV0; v < NV; ++v) pv[v] = *reinterpret_cast<const Vec4<T>*>(prow + (t4 + 4u * TPF * v));
    }
    F::template fwd_phase<1>(x, lds, tl, tw);
    GPA_PBAR();
    F::template fwd_phase<2>(x, lds, tl, tw);
    if constexpr (F::P > 3) { GPA_PBAR(); F::template fwd_phase<3>(x, lds, tl, tw); }
    GPA_PBAR();
#pragma unroll
    for (int i = 0; i < E; ++i) lds[F::pad(F::spec_index(tl, i))] = {x[i].x * inv, -x[i].y * inv};
    GPA_PBAR();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int j = tl + TPF * v;
      const cpx<T> a = lds[F::pad(j)], b = lds[F::pad(HN - 1 - j)];
      Vec4<T> out = {{a.x, b.y, a.y, b.x}};
      if (!first) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out.v[c] += beta * pv[v].v[c];
      }
      *reinterpret_cast<Vec4<T>*>(qrow + (t4 + 4u * TPF * v)) = out;
    }
  }
}

template <int LG>
hipError_t run_rowidct_p_halfpers(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it, hipStream_t s) {
  using G = HalfPersGeom<LG>;
  auto kern = rowidct_p_halfpers_kernel<LG>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
  if (e != hipSuccess) return e;
  const int cap = pers_grid(G::WGS_PER_CU);
  const int grid = w->n0 < cap ? w->n0 : cap;
  GPA_PROF("rowidct_p_kernel", s);
  kern<<<dim3(grid, 1, w->nprob), G::TPF, G::LDS_BYTES, s>>>((const float*)w->z, (const float*)pin, (float*)pout, w->n0,
                                                            (const cpx<float>*)w->tw1h, (const cpx<float>*)w->tw1,
                                                            (const cpx<float>*)w->wk1, w->flags, part_rho, nrho, w->scal, it,
                                                            (size_t)w->n0 * w->n1);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// forward row transform (contract of rowdct_half_kernel): it == 0: r (spatial) -> R in place; it > 0: R -= alpha
// DCT-II_rows(q) and one partial ||r||^2 per ROW (the consumer reduces n0 of them, as before)
// ---------------------------------------------------------------------------
template <int LG>
__global__ __launch_bounds__((HalfPersGeom<LG>::TPF), (HalfPersGeom<LG>::WAVES_PER_SIMD)) void rowdct_halfpers_kernel(
    float* __restrict__ r, const float* __restrict__ q, int n0, const cpx<float>* __restrict__ twh,
    const cpx<float>* __restrict__ twn, const cpx<float>* __restrict__ wk, int* flags, const double* part_pq, int npq,
    double* part_norm, double* scal, int it, int ring, int init, size_t pimg) {
  using T = float;
  using G = HalfPersGeom<LG>;
  using F = typename G::F;
  constexpr int TPF = G::TPF, N = G::N, HN = G::HN, E = G::E, NV = G::NV, NB = E / 4;
  {
    const size_t pb = blockIdx.z;
    r += pb * pimg;
    q += pb * pimg;
    flags += pb * FLAGS_N;
    scal += pb * SCAL_N;
    part_pq += pb * PART_N;
    part_norm += pb * PART_N;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double sh[HalfPersGeom<LG>::TPF];
  if (init) { if (!solve_init(part_pq, npq, scal, flags, sh)) return; }
  else if (flags[1]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T alpha = T(0);
  if (it > 0) {
    const double pq = reduce_partials(part_pq, npq, sh);
    const double alpha_d = scal[8 + ((it - 1) & 1)] / pq;   // phase_unwrap.py:343
    alpha = (T)alpha_d;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[SC_ALPHA + (it - 1) % ring] = alpha_d;
  }
  const int nwg = (int)gridDim.x;
  const int r0 = (int)((long long)blockIdx.x * n0 / nwg), r1 = (int)((long long)(blockIdx.x + 1) * n0 / nwg);
  if (r0 >= r1) return;
  const T* src = it > 0 ? q : r;
  const unsigned buf0 = lds_addr(smem);
  const unsigned lane16 = (unsigned)lane * 16u;
  __syncthreads();   // (sh is free again; nothing in flight yet)
  dma_row<LG>(src + (size_t)r0 * N, buf0, wave, lane16);
  __shared__ cpx<T> tws[F::LDS_TABLE_ELEMS];
  typename F::TwiddlesLds tw;
  F::fill_lds_tables(tws, twh, tid, TPF);
  F::load_twiddles(tw, twh, tid, tws);
  // the two phase tables of the post-processing, which owns its bins in blocks of four: k = 4 (tid + TPF v) + e
  cpx<T> tnv[E], wkv[E];
#pragma unroll
  for (int v = 0; v < NB; ++v) {
    const int k0 = 4 * (tid + TPF * v);
    struct alignas(16) C2 { cpx<T> a, b; };
    const C2 tn01 = *reinterpret_cast<const C2*>(twn + k0), tn23 = *reinterpret_cast<const C2*>(twn + k0 + 2);
    const C2 wk01 = *reinterpret_cast<const C2*>(wk + k0), wk23 = *reinterpret_cast<const C2*>(wk + k0 + 2);
    tnv[4 * v] = tn01.a; tnv[4 * v + 1] = tn01.b; tnv[4 * v + 2] = tn23.a; tnv[4 * v + 3] = tn23.b;
    wkv[4 * v] = wk01.a; wkv[4 * v + 1] = wk01.b; wkv[4 * v + 2] = wk23.a; wkv[4 * v + 3] = wk23.b;
  }
#pragma unroll
  for (int i = 0; i < E; ++i) { settle(wkv[i].x); settle(wkv[i].y); settle(tnv[i].x); settle(tnv[i].y); }
  settle_twiddles<F>(tw);
  __syncthreads();   // (tws)
  for (int row = r0; row < r1; ++row) {
    // (index arithmetic from an opaque copy of the thread index: hipcc would otherwise hoist every LDS address and lane offset
    //  of the row out of the loop and keep them, dozens of registers, alive across it)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int cur = (row - r0) & 1;
    char* bcur = smem + cur * G::BUF_BYTES;
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(bcur);
    const T* st = reinterpret_cast<const T*>(bcur);
    const size_t o = (size_t)row * N;
    // the row has landed: everything older than the previous row's stores (at least 2 NB per wave: the wave of thread 0
    // issues a few more for bin 0 / N/2 and the partial sum) is done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");
    GPA_PBAR();
    if (row + 1 < r1) dma_row<LG>(src + (size_t)(row + 1) * N, buf0 + (cur ^ 1) * G::BUF_BYTES, wave, lane16);
    // the kept spectrum (it > 0): requested here, used after the transform.  16384-point rows request it after the
    // transform's first pass instead (its sixteen-point butterflies are the register peak).
    Vec4<T> rlo[NB], rhi[NB];     // rhi[v].v[3 - e] = R[N - k0 - e]
    auto load_kept = [&]() {
#pragma unroll
      for (int v = 0; v < NB; ++v) {
        const int k0 = 4 * (tl + TPF * v);
        rlo[v] = *reinterpret_cast<const Vec4<T>*>(r + o + k0);
        if (k0 == 0) {
          rhi[v].v[3] = r[o + HN];
          rhi[v].v[2] = r[o + N - 1];
          rhi[v].v[1] = r[o + N - 2];
          rhi[v].v[0] = r[o + N - 3];
        } else {
          rhi[v] = *reinterpret_cast<const Vec4U<T>*>(r + o + N - k0 - 3);
        }
      }
    };
    if (it > 0) load_kept();
    // t[j] = (x[4j], x[4j+2]), t[N/2-1-j] = (x[4j+3], x[4j+1]): out of the landing zone with 16-byte reads ...
    Vec4<T> stage[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) stage[v] = *reinterpret_cast<const Vec4<T>*>(st + 4 * (tl + TPF * v));
    GPA_PBAR();   // ... every thread holds its samples: the landing zone becomes the exchange buffer
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int j = tl + TPF * v;
      lds[F::pad(j)] = {stage[v].v[0], stage[v].v[2]};
      lds[F::pad(HN - 1 - j)] = {stage[v].v[3], stage[v].v[1]};
    }
    GPA_PBAR();
    cpx<T> x[E];
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = lds[F::pad(tl + TPF * i)];
    GPA_PBAR();
    F::template fwd_phase<0>(x, lds, tl, tw);
    GPA_PBAR();
    F::template fwd_phase<1>(x, lds, tl, tw);
    GPA_PBAR();
    F::template fwd_phase<2>(x, lds, tl, tw);
    if constexpr (F::P > 3) { GPA_PBAR(); F::template fwd_phase<3>(x, lds, tl, tw); }
    GPA_PBAR();
#pragma unroll
    for (int i = 0; i < E; ++i) lds[F::pad(F::spec_index(tl, i))] = x[i];
    GPA_PBAR();
    double sq = 0;
#pragma unroll
    for (int v = 0; v < NB; ++v) {
      const int k0 = 4 * (tl + TPF * v);
      Vec4<T> olo, ohi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        const cpx<T> zk = lds[F::pad(k)], zm = lds[F::pad((HN - k) & (HN - 1))];
        T xlo, xhi;
        if (k == 0) {
          xlo = T(2) * (zk.x + zk.y);
          xhi = T(1.41421356237309504880) * (zk.x - zk.y);
        } else {
          const cpx<T> ve = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};   // (T_k + conj T_m) / 2
          const cpx<T> vo = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};  // -i (T_k - conj T_m) / 2
          const cpx<T> V = ve + cmul(tnv[4 * v + e], vo);
          const cpx<T> U = cmul(wkv[4 * v + e], V);
          xlo = T(2) * U.x;
          xhi = T(-2) * U.y;
        }
        if (it > 0) {
          xlo = rlo[v].v[e] - alpha * xlo;
          xhi = rhi[v].v[3 - e] - alpha * xhi;
          // sum_n r^2 = (1 / 2N) sum_k c_k R_k^2, c_0 = 1/2 (SciPy's unnormalised DCT-II)
          sq += (k == 0 ? 0.5 : 1.0) * (double)xlo * (double)xlo + (double)xhi * (double)xhi;
        }
        olo.v[e] = xlo;
        ohi.v[3 - e] = xhi;
      }
      *reinterpret_cast<Vec4<T>*>(r + o + k0) = olo;
      if (k0 == 0) {
        r[o + HN] = ohi.v[3];
        r[o + N - 1] = ohi.v[2];
        r[o + N - 2] = ohi.v[1];
        r[o + N - 3] = ohi.v[0];
      } else {
        *reinterpret_cast<Vec4U<T>*>(r + o + N - k0 - 3) = ohi;
      }
    }
    if (it > 0) {
      // the row's partial sum, in the order block_sum() takes it (wavefront trees, then the wavefronts in order) -- with a
      // barrier that leaves the DMA alone; sh is reused only after the next row's first barrier
      const double ws = wave_sum(sq);
      if (lane == 0) sh[wave] = ws;
      GPA_PBAR();
      if (tl == 0) {
        double t = 0;
        for (int i = 0; i < TPF / 64; ++i) t += sh[i];
        part_norm[row] = t / (2.0 * N);
      }
    }
  }
}

template <int LG>
hipError_t run_rowdct_halfpers(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm, int it,
                               int* nnorm, int init, hipStream_t s) {
  using G = HalfPersGeom<LG>;
  auto kern = rowdct_halfpers_kernel<LG>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
  if (e != hipSuccess) return e;
  if (w->n0 > MAXPART) return hipErrorInvalidValue;
  *nnorm = w->n0;
  const int cap = pers_grid(G::WGS_PER_CU);
  const int grid = w->n0 < cap ? w->n0 : cap;
  GPA_PROF("rowdct_fused_kernel", s);
  kern<<<dim3(grid, 1, w->nprob), G::TPF, G::LDS_BYTES, s>>>((float*)w->r, (const float*)q, w->n0, (const cpx<float>*)w->tw1h,
                                                            (const cpx<float>*)w->tw1, (const cpx<float>*)w->wk1, w->flags, part_pq,
                                                            npq, part_norm, w->scal, it, ring, init, (size_t)w->n0 * w->n1);
  return hipGetLastError();
}

}  // namespace

bool rowhalfpers_offered(const Impl* w) {
  return !w->generic && w->dtype == 0 && (w->lg1 == 13 || w->lg1 == 14) && w->tw1h != nullptr && w->n0 >= 64 && !opt_set(OPT_NO_ROWPERS);
}
hipError_t rowhalfpers_rowidct_p(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it, hipStream_t s) {
  if (w->lg1 == 13) return run_rowidct_p_halfpers<13>(w, pin, pout, part_rho, nrho, it, s);
  if (w->lg1 == 14) return run_rowidct_p_halfpers<14>(w, pin, pout, part_rho, nrho, it, s);
  return hipErrorInvalidValue;
}
hipError_t rowhalfpers_rowdct(const Impl* w, const void* q, int ring, const double* part_pq, int npq, double* part_norm, int it,
                              int* nnorm, int init, hipStream_t s) {
  if (w->lg1 == 13) return run_rowdct_halfpers<13>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  if (w->lg1 == 14) return run_rowdct_halfpers<14>(w, q, ring, part_pq, npq, part_norm, it, nnorm, init, s);
  return hipErrorInvalidValue;
}

}  // namespace gpa
