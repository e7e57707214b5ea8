// Pass B of the lock-in sweep (y-axis filter on rows + per-pixel selection), shared by the translation
// units that instantiate its modes: gpa_sweep.hip (all lock-ins / plain best-of-K, the headline path) and
// gpa_sweep_ext.hip (gated selection of wfr4, selection + per-candidate phases for the a4 gradient).
#pragma once
#include "gpa_internal.h"

namespace gpa {

enum { PB_ALL = 0, PB_SELECT = 1, PB_GATED = 2, PB_PHASES = 3, PB_PART = 4 };

template <class T, int LG>
struct PassBGeom {
  using F = WgFFT<T, LG>;
  // rows per workgroup: up to 1024-point rows ONE wavefront per workgroup (a row of 64 threads, or two of 32: every
  // exchange stays inside the wavefront and small images get more, smaller workgroups: 1024^2 131 -> 117 us);
  // longer rows fill 256 threads
#ifndef GPA_PB_THREADS_SMALL
#define GPA_PB_THREADS_SMALL 64
#endif
  static constexpr int WGT = LG <= 10 ? GPA_PB_THREADS_SMALL : 256;
  static constexpr int NF = F::TPF >= WGT ? 1 : WGT / F::TPF;   // rows per workgroup
  static constexpr int RS = F::LDS_ELEMS + (NF > 1 ? (F::TPF < 32 ? F::TPF : 0) : 0);
  static constexpr int THREADS = NF * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NF * RS * sizeof(cpx<T>);
};

template <bool PADDED, class T>
struct HType { using type = T; };
template <class T>
struct HType<true, T> { using type = cpx<T>; };

template <class T> __device__ __forceinline__ cpx<T> hmul(cpx<T> a, T h) { return {a.x * h, a.y * h}; }
template <class T> __device__ __forceinline__ cpx<T> hmul(cpx<T> a, cpx<T> h) { return cmul(a, h); }

// entry j of the filter table behind an opaque pointer (asm volatile hides it from LICM so that the table is re-read
// per candidate instead of being pinned in 16+ registers).  The opaque pointer also loses its address space -- hipcc
// emitted FLAT loads for the table -- so it is read through a global-address-space scalar pointer.
template <bool PADDED, class T>
__device__ __forceinline__ typename HType<PADDED, T>::type hload(const typename HType<PADDED, T>::type* Hb, int j) {
  using gscalar = const __attribute__((address_space(1))) T;
  gscalar* g = (gscalar*)reinterpret_cast<const T*>(Hb);
  if constexpr (PADDED) return cpx<T>{g[2 * j], g[2 * j + 1]};
  else return g[j];
}

// ---------------------------------------------------------------------------
// pass B: y-axis filter on rows, best-of-K select
// ---------------------------------------------------------------------------
template <class T, int LG, bool PADDED, int MODE>
#ifndef GPA_PASSB_PADDED_WAVES
#define GPA_PASSB_PADDED_WAVES 3   // f32, padded axis: the selects of the extension slots push hipcc to 200 VGPRs (2 waves); capped at 168
#endif
#ifndef GPA_F64_WAVES
#define GPA_F64_WAVES 2   // f64: cap at 256 VGPRs (2 waves/SIMD) instead of 299 at 1 wave: pass B 7.5 -> 5.9 ms
#endif
__global__ __launch_bounds__((PassBGeom<T, LG>::THREADS), (sizeof(T) == 8 ? GPA_F64_WAVES : (PADDED ? GPA_PASSB_PADDED_WAVES : 1))) void passB_kernel(
    const cpx<T>* __restrict__ Tin, int n0, int n1,
    const typename HType<PADDED, T>::type* __restrict__ H, const cpx<T>* __restrict__ twtab,
    const int* __restrict__ planeof, const cpx<T>* __restrict__ cyb, const cpx<T>* __restrict__ sy,
    const cpx<T>* __restrict__ wyw, const cpx<T>* __restrict__ wyr, int extL, int extR,
    const cpx<T>* __restrict__ dx, const cpx<T>* __restrict__ dy, int K,
    cpx<T>* __restrict__ out, int32_t* __restrict__ kidx, const uint8_t* __restrict__ gate, T* __restrict__ psi,
    int P, int Bx) {
  constexpr bool SELECT = MODE != PB_ALL;
  using F = WgFFT<T, LG>;
  using G = PassBGeom<T, LG>;
  constexpr int TPF = F::TPF, L = F::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int row = blockIdx.x * G::NF + f;
  const bool valid = row < n0;
  // image stacks: blockIdx.y = image * P + peak.  `p` indexes the outputs (one plane per image and peak), `pt` the
  // candidate tables (the same for every image); image img reads its own Bx x-planes
  const int p = blockIdx.y, pt = p % P, img = p / P;

  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);

  cpx<T> best[16];
  int bidx[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { best[i] = {T(0), T(0)}; bidx[i] = -1; }

  // PB_PART (small images): workgroup z of gridDim.z takes the candidates [z Kc, (z + 1) Kc) and leaves its RAW winner
  // (uncompensated value + index) in slab z of out / kidx; merge_parts_kernel picks among the slabs in candidate
  // order with the same strict '>' and applies the compensation -- the K candidates of a row then run on gridDim.z
  // workgroups instead of one after the other in a single one
  int k0 = 0, nk = SELECT ? K : 1;
  if constexpr (MODE == PB_PART) {
    const int kc = (K + (int)gridDim.z - 1) / (int)gridDim.z;
    k0 = (int)blockIdx.z * kc;
    nk = k0 + kc < K ? k0 + kc : K;
  }
  // padded axes: which of this thread's slots lie in the right / left periodic extension or in the zero gap (the
  // same for every candidate).  Registers whose 'TPF' slots all lie inside the image (a wave-uniform test) take the
  // periodic code path; the others pick their source sample and wrap factor by selects, not branches -- per-element
  // branches around the loads cost 83 exec-mask branches per candidate and serialised the 16 loads of a thread.
  unsigned rmask = 0, wmask = 0, zmask = 0;
  if constexpr (PADDED) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int slot = tid + TPF * i;
      if (slot >= n1) {
        if (slot < n1 + extR) rmask |= 1u << i;
        else if (slot >= L - extL) wmask |= 1u << i;
        else zmask |= 1u << i;
      }
    }
  }
  for (int k = k0; k < nk; ++k) {
    const int b = SELECT ? pt * K + k : pt;
    F::refresh(tw);
    // the x-plane of this candidate (shared by every candidate with the same wx: re-reads hit L2)
    const cpx<T>* src = Tin + (((size_t)img * Bx + planeof[b]) * n0 + (valid ? row : 0)) * n1;
    const cpx<T> cbase = cyb[(size_t)b * TPF + tid];
    cpx<T> x[16];
    cpx<T> fr = {T(1), T(0)}, fw = {T(1), T(0)};
    if constexpr (PADDED) { fr = wyr[b]; fw = wyw[b]; }
    // (the source offsets of the extension slots do not depend on the candidate: computed from an opaque copy of
    //  the thread index so that hipcc does not keep them -- 64-bit each -- in registers across the whole loop:
    //  217 VGPRs / 2 waves per SIMD otherwise, against 165 / 3 of the periodic kernel)
    int tl = tid;
    if constexpr (PADDED) asm volatile("" : "+v"(tl));
    if constexpr (PADDED) {
      // all sixteen loads first, their addresses chosen by selects, then straight-line arithmetic: with per-register
      // branches the kernel had 98 branches, 55 vmcnt waits and 44 scratch accesses in its body against 3 / 19 / 0 of
      // the periodic kernel, and needed 1.95x its wave cycles for 1.12x its instructions
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int slot = tl + TPF * i;
        const bool r = (rmask >> i) & 1, w = (wmask >> i) & 1, z = (zmask >> i) & 1;
        const int ys = z ? 0 : slot - (r ? n1 : (w ? L - n1 : 0));
        x[i] = src[ys];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        cpx<T> ph = cmul(cbase, sy[b * 16 + i]);   // exp(2 pi i wy y) at y = tid + TPF*i
        // wrap factor of the extension this slot lies in (1 inside the image, 0 in the zero gap): selects, no branches
        const bool r = (rmask >> i) & 1, w = (wmask >> i) & 1, z = (zmask >> i) & 1;
        const cpx<T> f = {z ? T(0) : (r ? fr.x : (w ? fw.x : T(1))), z ? T(0) : (r ? fr.y : (w ? fw.y : T(0)))};
        x[i] = cmul(x[i], cmul(ph, f));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cpx<T> ph = cmul(cbase, sy[b * 16 + i]);   // exp(2 pi i wy y) at y = tid + TPF*i
        x[i] = cmul(src[tid + TPF * i], ph);             // rows past the image reuse row 0; their results are dropped
      }
    }
    F::forward(x, lds, tid, tw);
    {
      const typename HType<PADDED, T>::type* Hb = H;
      asm volatile("" : "+s"(Hb));   // re-read the filter table per candidate instead of pinning 16+ VGPRs
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = hmul(x[i], hload<PADDED, T>(Hb, i * TPF + tid));
    }
    F::inverse(x, lds, tid, tw);
    if constexpr (SELECT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // |sf|^2 of the kept candidate is recomputed rather than kept in a register
        const T a = x[i].x * x[i].x + x[i].y * x[i].y;
        const T ab = best[i].x * best[i].x + best[i].y * best[i].y;
        if constexpr (MODE == PB_GATED) {
          // wfr4 (geometric_phase_analysis.py:857-858): also within 2 sqrt(2) dk of the kept k-vector, which
          // is klist[0] until something is accepted; gate[j * K + k] is that test, made by the host in double
          const int j = bidx[i] < 0 ? 0 : bidx[i];
          if (a > ab && gate[(size_t)j * K + k]) { best[i] = x[i]; bidx[i] = k; }
        } else {
          if (a > ab) { best[i] = x[i]; bidx[i] = k; }
        }
      }
      if constexpr (MODE == PB_PHASES) {
        // a4: -angle(sf) of EVERY candidate goes out (one real per pixel and candidate; the gradient
        // kernel then reads the winner's phase at the pixel's four neighbours)
        if (valid) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int yy = tid + TPF * i;
            if (!PADDED || yy < n1) psi[((size_t)(p * K + k) * n0 + row) * n1 + yy] = -atan2(x[i].y, x[i].x);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) best[i] = x[i];
    }
  }
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int yy = tid + TPF * i;
    if (!PADDED || yy < n1) {
      const size_t o = ((size_t)p * n0 + row) * n1 + yy;
      if constexpr (MODE == PB_PART) {
        const size_t slab = (size_t)gridDim.y * n0 * n1;     // P * n0 * n1 values per slab
        out[(size_t)blockIdx.z * slab + o] = best[i];
        kidx[(size_t)blockIdx.z * slab + o] = bidx[i];
      } else if constexpr (SELECT) {
        cpx<T> v = {T(0), T(0)};
        if (bidx[i] >= 0) {
          const size_t bb = (size_t)pt * K + bidx[i];
          v = cmul(best[i], cmul(dx[bb * n0 + row], dy[bb * n1 + yy]));
        }
        out[o] = v;
        if (kidx) kidx[o] = bidx[i];
      } else {
        out[o] = best[i];
      }
    }
  }
}


template <class T, int LG, bool PADDED, int MODE>
static hipError_t run_passB(const Axis& a1, int n0, const void* Tbuf, const void* Hy,
                            const void* tw1, const SweepTables& tb, int P, int K, void* out,
                            int32_t* kidx, const uint8_t* gate, void* psi, hipStream_t s, int ksplit = 1, int nimg = 1,
                            int Bx = 0) {
  using G = PassBGeom<T, LG>;
  if constexpr (G::LDS_BYTES > 160 * 1024) {
    return hipErrorInvalidValue;
  } else {
    auto kern = passB_kernel<T, LG, PADDED, MODE>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    dim3 grid((n0 + G::NF - 1) / G::NF, P * nimg, ksplit);
    GPA_PROF("passB_kernel", s);
    kern<<<grid, G::THREADS, G::LDS_BYTES, s>>>(
        (const cpx<T>*)Tbuf, n0, a1.n, (const typename HType<PADDED, T>::type*)Hy,
        (const cpx<T>*)tw1, tb.planeof, (const cpx<T>*)tb.cyb, (const cpx<T>*)tb.sy, (const cpx<T>*)tb.wyw,
        (const cpx<T>*)tb.wyr, a1.extL, a1.extR,
        (const cpx<T>*)tb.dx, (const cpx<T>*)tb.dy, K, (cpx<T>*)out, kidx, gate, (T*)psi, P, Bx);
    return hipGetLastError();
  }
}

#define GPA_FOR_LG(X) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)

}  // namespace gpa
