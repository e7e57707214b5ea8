// a7, PERSISTENT row kernel of the fused PCG iteration on 4096-point f32 rows (round 5; phase_unwrap.py:326-349, the
// preconditioner's row DCT-III :95-103 and the search-direction update :332-340): rowidct_p.  (The same treatment of the
// stencil + forward-transform kernel pqdct measured SLOWER -- 69 against 58 us -- and is not in the tree:
// profiles/r05_pqdct_persistent_rejected.txt.)
//
// The one-row-pair-per-workgroup kernels of gpa_unwrap_rows.hip run 2048 workgroups as two rounds of the chip: the four
// workgroups of a CU load, transform and store IN STEP, so the memory system idles while they compute and the SIMDs idle
// while they wait (measured round = load time + compute time: 44 us = 2 x (15.6 + 6.6), nothing overlapped).  Here a
// workgroup is resident (two per CU, <= 256 VGPRs, 72 KB of LDS) and walks a contiguous band of row pairs with an
// explicit software pipeline:
//   * the spectrum Z of pair j + 1 lands in the OTHER of two LDS buffers by LDS-DMA (global_load_lds_dwordx4: no VGPRs,
//     8 wave-instructions per pair) while pair j is transformed; a buffer is first the landing zone of a pair's two rows
//     (2 x 16 KB, lane-linear as the DMA writes it), then -- once every thread has its inputs in registers -- the
//     exchange buffer of that pair's transform;
//   * the previous search direction of pair j is requested into registers (16-byte loads) before the transform and used
//     after it; the new one leaves as 16-byte stores that the next iteration does not wait for (counted vmcnt).
// Barriers are raw s_barrier + lgkmcnt(0): __syncthreads() would drain the DMA in flight (vmcnt(0)).
// Same arithmetic as rowidct_p_kernel, same order of operations per value: bit-identical p.
#include "gpa_unwrap_pers.h"

namespace gpa {
namespace {

template <int LG>
struct PersGeom {
  using T = float;
  static constexpr int E = 16;
  using F = WgFFT<T, LG, E>;
  using D = WgDCT<T, LG, E>;
  static_assert(F::P == 3 && F::TPF == 256, "persistent row kernels: 4096-point rows, 256 threads");
  using TW = typename F::TwiddlesP1Lds;
  static constexpr int N = F::L, TPF = F::TPF;
  static constexpr int T1N = F::P1_SETS * 6;
  static constexpr int BUF_BYTES = F::LDS_ELEMS * (int)sizeof(cpx<T>);   // 34816: exchange buffer >= the two rows it receives first
  static_assert(BUF_BYTES >= 2 * N * (int)sizeof(T), "a buffer holds a row pair");
  static constexpr int NQ = N / (4 * TPF);           // 16-byte vectors per thread and row
  static constexpr int NDMA = 2 * N * (int)sizeof(T) / (TPF * 16);   // DMA wave-instructions per wave and pair (1 KB each)
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES;
};

// the two rows of pair `pr` -> buffer at LDS byte address dst: row a at [0, 4N), row b at [4N, 8N)
template <int LG>
__device__ __forceinline__ void dma_pair(const float* __restrict__ rows, unsigned dst, int wave, int lane) {
  using G = PersGeom<LG>;
  constexpr int WAVES = G::TPF / 64;
#pragma unroll
  for (int i = 0; i < G::NDMA; ++i) {
    const int piece = i * WAVES + wave;            // 1 KB pieces of the 2 x 4N contiguous bytes (the pair's rows are adjacent)
    glds16(reinterpret_cast<const char*>(rows) + (size_t)piece * 1024 + lane * 16,
           (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + piece * 1024)));
  }
}

template <int LG>
__global__ __launch_bounds__(256, 2) void rowidct_p_pers_kernel(
    const float* __restrict__ Z, const float* __restrict__ pin, float* __restrict__ pout, int n0,
    const cpx<float>* __restrict__ twtab, const cpx<float>* __restrict__ wk, const int* flags, const double* part_rho,
    int nrho, double* scal, int it, size_t pimg) {
  using T = float;
  using G = PersGeom<LG>;
  using F = typename G::F;
  using D = typename G::D;
  constexpr int TPF = G::TPF, N = G::N, E = G::E, NQ = G::NQ;
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
  __shared__ double sh[256];
  __shared__ cpx<T> t1s[G::T1N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double rho_part = load_partials(part_rho, nrho);
  const double rho_prev = scal[8 + ((it - 1) & 1)];
  if (stop) return;
  const bool first = it == 0;                        // first iteration: p = z (pin is uninitialised)
  const double rho = block_sum(rho_part, sh);
  const T beta = first ? T(0) : (T)(rho / rho_prev);
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[8 + (it & 1)] = rho;
  // the band of row pairs of this workgroup
  const int npairs = n0 / 2, nwg = (int)gridDim.x;
  const int p0 = (int)((long long)blockIdx.x * npairs / nwg), p1 = (int)((long long)(blockIdx.x + 1) * npairs / nwg);
  if (p0 >= p1) return;
  const unsigned buf0 = lds_addr(smem);
  dma_pair<LG>(Z + (size_t)2 * p0 * N, buf0, wave, lane);
  typename G::TW tw;
  F::fill_pass1_table(t1s, twtab, tid, TPF);
  cpx<T> wkv[E];
#pragma unroll
  for (int i = 0; i < E; ++i) wkv[i] = wk[tid + TPF * i];
  __syncthreads();   // (t1s)
  F::load_twiddles(tw, twtab, tid, t1s);
  // The loop-invariant tables must have ARRIVED before the loop: a wait for them inside the loop body (where the compiler
  // would otherwise put it, at their first use) is a vmcnt(0) that every iteration executes -- it would drain the next
  // pair's DMA in the middle of the transform.  An opaque use of each register forces the wait here.
#pragma unroll
  for (int i = 0; i < E; ++i) { settle(wkv[i].x); settle(wkv[i].y); }
#pragma unroll
  for (int q = 0; q < F::GMAX; ++q)
#pragma unroll
    for (int c = 0; c < 3; ++c) { settle(tw.lo[q][c].x); settle(tw.lo[q][c].y); settle(tw.hi[q][c].x); settle(tw.hi[q][c].y); }
  for (int pr = p0; pr < p1; ++pr) {
    const int cur = (pr - p0) & 1;
    char* bcur = smem + cur * G::BUF_BYTES;
    cpx<T>* lds = reinterpret_cast<cpx<T>*>(bcur);
    const T* st = reinterpret_cast<const T*>(bcur);
    const size_t oa = (size_t)2 * pr * N, ob = oa + N;
    // this wave's pieces of pair pr have landed (everything older than the previous pair's NQ * 2 stores is done).  The count
    // holds only while the youngest vector-memory operations of an iteration are exactly those stores -- no split stores, no
    // scratch accesses in the loop: tests/test_isa_invariants.py checks the compiled loop for precisely that.
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ) : "memory");
    GPA_PBAR();   // ... and everybody else's; the other buffer's last reads (previous gather) are done
    if (pr + 1 < p1) dma_pair<LG>(Z + (size_t)2 * (pr + 1) * N, buf0 + (cur ^ 1) * G::BUF_BYTES, wave, lane);
    Vec4<T> pva[NQ], pvb[NQ];
    if (!first) {
#pragma unroll
      for (int v = 0; v < NQ; ++v) {
        const int c0 = 4 * (tid + TPF * v);
        pva[v] = *reinterpret_cast<const Vec4<T>*>(pin + oa + c0);
        pvb[v] = *reinterpret_cast<const Vec4<T>*>(pin + ob + c0);
      }
    }
    cpx<T> x[E], xm[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tid + TPF * i;
      const int km = (N - k) & (N - 1);
      x[i] = {st[k], st[N + k]};
      xm[i] = {st[km], st[N + km]};
      if (k == 0) xm[i] = cpx<T>{T(0), T(0)};
    }
    D::inv_prepare(x, xm, wkv);
    GPA_PBAR();   // every thread holds its inputs: the landing zone becomes the exchange buffer
    F::template fwd_phase<0>(x, lds, tid, tw);
    GPA_PBAR();
    F::template fwd_phase<1>(x, lds, tid, tw);
    GPA_PBAR();
    F::template fwd_phase<2>(x, lds, tid, tw);
    GPA_PBAR();
    D::inv_scatter(x, lds, tid, T(1) / T(N));
    GPA_PBAR();
#pragma unroll
    for (int v = 0; v < NQ; ++v) {
      const int c0 = 4 * (tid + TPF * v);
      Vec4<T> oa4, ob4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const cpx<T> z = lds[F::pad(c0 + j)];
        T pa = z.x, pb = z.y;
        if (!first) {
          pa += beta * pva[v].v[j];
          pb += beta * pvb[v].v[j];
        }
        oa4.v[j] = pa;
        ob4.v[j] = pb;
      }
      *reinterpret_cast<Vec4<T>*>(pout + oa + c0) = oa4;
      *reinterpret_cast<Vec4<T>*>(pout + ob + c0) = ob4;
    }
  }
}

// two resident workgroups per CU of the CURRENT device (one slot per device index, like set_dynamic_lds_once: ADVICE r05)
int pers_workgroups() {
  static int n[32] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& slot = n[dev & 31];
  if (!slot) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    slot = 2 * cus;
  }
  return slot;
}

}  // namespace

bool pow2_rowpers_offered(const Impl* w) {
  return !w->generic && w->dtype == 0 && w->lg1 == 12 && (w->n0 % 2) == 0 && w->n0 >= 64 && !opt_set(OPT_NO_ROWPERS);
}

hipError_t pow2_rowidct_p_pers(const Impl* w, const void* pin, void* pout, const double* part_rho, int nrho, int it,
                               hipStream_t s) {
  using G = PersGeom<12>;
  auto kern = rowidct_p_pers_kernel<12>;
  static unsigned lds_set = 0;
  hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
  if (e != hipSuccess) return e;
  const int npairs = w->n0 / 2;
  const int grid = npairs < pers_workgroups() ? npairs : pers_workgroups();
  GPA_PROF("rowidct_p_kernel", s);
  kern<<<dim3(grid, 1, w->nprob), G::TPF, G::LDS_BYTES, s>>>((const float*)w->z, (const float*)pin, (float*)pout, w->n0,
                                                            (const cpx<float>*)w->tw1, (const cpx<float>*)w->wk1, w->flags, part_rho,
                                                            nrho, w->scal, it, (size_t)w->n0 * w->n1);
  return hipGetLastError();
}

}  // namespace gpa
