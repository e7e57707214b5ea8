// Pass B of the best-of-K sweep with the forward transform SHARED by the candidates of an x-plane.
//
// The reference computes, for every candidate b of a peak (geometric_phase_analysis.py:72-75, :679-684),
//     sf_b(y) = sum_y' g((y - y') mod n) T(y') c_b(y'),   c_b(y) = exp(2 pi i wy_b y),
// T = the x-plane of the candidate (pass A), g = the taps of the circular Gaussian filter.  Pulling the carrier
// through the sum,
//     sf_b(y) = c_b(y) [ (g_b (*) T)(y) + fix_b(y) ],      g_b(m) = g(m) exp(-2 pi i wy_b m)   (signed lag m),
// the transform of g_b is REAL (the Gaussian shifted by wy_b, tabulated in double per candidate) and T no longer
// carries anything of the candidate: the candidates that share an x-plane share ONE forward transform of its row
// (4 forward + 16 inverse transforms per row and peak for a 4 x 4 grid instead of 16 + 16, and one read of the row
// instead of four).  c_b(y) is a unit phasor: it does not change |sf| and merges with the compensation
// exp(-2 pi i (wy_b - ky) y) of :683 into the candidate-independent exp(2 pi i ky y), applied once to the winner.
//
// fix_b is where the identity fails: the carrier is not n-periodic, so the pairs (y, y') that the circular filter
// joins AROUND the row end carry exp(-/+ 2 pi i wy_b n) relative to what g_b (*) T gives them.  With taps that vanish
// beyond E samples only the outputs within E of either end are touched,
//     fix_b(n - a)  = (phi_b - 1)       sum_{j=0}^{E-a} g(a + j) exp(+2 pi i wy_b (a + j)) T(j)           a = 1 .. E
//     fix_b(a - 1)  = (conj phi_b - 1)  sum_{j=0}^{E-a} g(a + j) exp(-2 pi i wy_b (a + j)) T(n - 1 - j),  phi_b = exp(-2 pi i wy_b n),
// i.e. one small dense REAL Hankel matrix G[a][j] = g(a + j) (the same for every candidate) times the end strips of
// the row pre-multiplied by the candidate's phasor -- a dense contraction, which is what the matrix cores are for:
// it runs as v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 (exact f32 / f64 FMA chains) on the otherwise idle
// matrix pipe beside the VALU-bound transforms, 16 columns = 2 ends x NC candidates x (re, im) at a time.
// On a zero-padded (non-power-of-two) row the same formulas hold with (phi_b - 0): g_b (*) T then contains no
// wrapped pair at all and the transform length only has to be >= n + E.
//
// Selection ("strictly larger |sf| replaces", in list order = the first maximum of the list wins) keeps only |best|^2
// in registers: a winner is stored to `out` the moment it wins and the row is revisited once at the end for the
// compensation phasor, which frees the registers that the shared spectrum needs.  The candidates of a peak are
// VISITED nearest-to-the-reference-vector first (host: shared_prepare, PassBSharedTables::order; kidx reports list
// positions): the maximum does not depend on the order, later candidates then seldom win, and a register none of whose
// 64 lanes wins issues no store.
#include <stdlib.h>

#include "gpa_internal.h"
#include "gpa_passb_shared.h"

namespace gpa {

template <class T> struct MfmaVec;
template <> struct MfmaVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct MfmaVec<double> { typedef double type __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ MfmaVec<float>::type mfma16(float a, float b, MfmaVec<float>::type c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ MfmaVec<double>::type mfma16(double a, double b, MfmaVec<double>::type c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// row of the 16 x 16 result held in element r of lane-quarter kq (f64 uses a different map than every other dtype)
template <class T> __device__ __forceinline__ int mfma_row(int kq, int r) {
  if constexpr (sizeof(T) == 8) return kq + 4 * r;
  else return 4 * kq + r;
}

template <class T> struct upair { T u, v; };

#ifndef GPA_PBS_NOFIX
#define GPA_PBS_NOFIX 0     // diagnosis only: skip the end fix (wrong results at the row ends)
#endif
#ifndef GPA_PBS_EXECSTORE
#define GPA_PBS_EXECSTORE 0  // experiment: winners stored under the EXEC mask instead of through dropped offsets
#endif
#ifndef GPA_PBS_NOSTORE
#define GPA_PBS_NOSTORE 0   // diagnosis only: skip the winner stores
#endif
#ifndef GPA_PBS_WAVESKIP
#define GPA_PBS_WAVESKIP 1   // a register with no winning lane in the wavefront issues no store
#endif
#ifndef GPA_PBS_NTLOAD
#define GPA_PBS_NTLOAD 1    // the x-plane row (read once) as a non-temporal load, so that it does not evict the winners' rows from L2 (-2.5 %, L2 hit rate 0.69 -> 0.80)
#endif

// the shifted Gaussian of one candidate: sixteen reals per thread in the spectral register layout
template <class T, int TPF, int EE>
__device__ __forceinline__ void load_gb(T (&h)[EE], const T* gbrow, int tid) {
  using gscalar = const __attribute__((address_space(1))) T;
  gscalar* gb = (gscalar*)gbrow;
#pragma unroll
  for (int i = 0; i < EE; ++i) h[i] = gb[i * TPF + tid];
}
template <class T> __device__ __forceinline__ cpx<T> load_cpx(__amdgpu_buffer_rsrc_t r, int voff, int soff, int aux = 16);
// a value that is read once: non-temporal (streams through L2)
__device__ __forceinline__ cpx<float> load_once(const cpx<float>* p) {
  typedef float v2f_t __attribute__((ext_vector_type(2)));
  const v2f_t d = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(p));
  return {d.x, d.y};
}
__device__ __forceinline__ cpx<double> load_once(const cpx<double>* p) {
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  const v2d_t d = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
  return {d.x, d.y};
}

typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_cpx(cpx<double> v, __amdgpu_buffer_rsrc_t r, int voff, int soff);
__device__ __forceinline__ void store_cpx(cpx<float> v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  v2i_t d = {__float_as_int(v.x), __float_as_int(v.y)};
  __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 0);
}
// aux 16 = sc1: served by L2, not by whatever L1 holds (the read-back of the winners); 2 = nt (read once); 0 = an ordinary load
template <> __device__ __forceinline__ cpx<float> load_cpx<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  v2i_t d = aux == 16 ? __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 16) : aux == 2 ? __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2) : __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return {__int_as_float(d.x), __int_as_float(d.y)};
}
template <> __device__ __forceinline__ cpx<double> load_cpx<double>(__amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  v4i_t d = aux == 16 ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16) : aux == 2 ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  const long long a = ((long long)(unsigned)d.x) | ((long long)d.y << 32), b = ((long long)(unsigned)d.z) | ((long long)d.w << 32);
  return {__longlong_as_double(a), __longlong_as_double(b)};
}
__device__ __forceinline__ void store_pair(cpx<float> a, cpx<float> b, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  v4i_t d = {__float_as_int(a.x), __float_as_int(a.y), __float_as_int(b.x), __float_as_int(b.y)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);
}
__device__ __forceinline__ void store_pair(cpx<double> a, cpx<double> b, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  store_cpx(a, r, voff, soff);
  store_cpx(b, r, voff + 16, soff);
}
__device__ __forceinline__ void store_cpx(cpx<double> v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const long long a = __double_as_longlong(v.x), b = __double_as_longlong(v.y);
  v4i_t d = {(int)a, (int)(a >> 32), (int)b, (int)(b >> 32)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);
}

// EE = elements per thread of the row transform: 16 (256 threads per 4096-point row, three passes) or 8 (512 threads,
// four passes: half the registers per thread -- the shared spectrum, the working copy and |best|^2 are per-element
// arrays -- for one more exchange per transform)
template <class T, int LG, int EE = 16>
struct PassBSGeom {
  using F = WgFFT<T, LG, EE>;
  static_assert(F::P == 3 || F::P == 4, "three- or four-pass transforms");
  static constexpr int TPF = F::TPF;
  static_assert(TPF >= 64, "a row needs whole wavefronts");
  // (2048-point rows, 128 threads each: ONE row per workgroup -- two rows per workgroup cost a third of the occupancy
  //  to LDS: 514 -> 428 us at 2048^2, 3 x 8)
#ifndef GPA_PBS_MINTHREADS
#define GPA_PBS_MINTHREADS 128
#endif
  static constexpr int NF = TPF >= GPA_PBS_MINTHREADS ? 1 : GPA_PBS_MINTHREADS / TPF;   // rows per workgroup
  static constexpr int THREADS = NF * TPF;
  static constexpr int NW = TPF / 64;                      // wavefronts per row
  static constexpr int NC = PassBSharedNC<T>::value;       // candidates per matrix pass
  static constexpr int LOGNC = NC == 4 ? 2 : 1;
  // f32: the candidate phasors the end fix needs (pre-factors of the strips, post-factors of the results) are staged
  // in LDS once per chunk of NC candidates, so that neither the matrix pass nor the fix-up waits on global loads
  // (which queue behind the winner stores in vmcnt).  f64 has no LDS left for that at 4096 points.
  static constexpr bool STAGE = sizeof(T) == 4;
#ifndef GPA_PBS_TWL
#define GPA_PBS_TWL 1
#endif
  static constexpr bool TWL = GPA_PBS_TWL && F::P == 3;    // pass-1 twiddles from an LDS table (three-pass transforms)
  static constexpr int T1 = TWL ? F::P1_SETS * 6 : 0;      // that table (complex), one per workgroup
  // end strips: f32 keeps the four (end, re/im) forms the matrix pass multiplies directly, f64 (no LDS to spare at
  // 4096 points) the two complex strips and forms them per lane with selects
  static constexpr int SV = sizeof(T) == 4 ? 4 : 2;
  // per-row LDS in complex elements: transform image | end strips [end][re/im variant][Epad] | results [NC][end][Epad]
  // | staged pre-factors [NC][Epad] | staged post-factors [2][NC][Epad] (double-buffered by chunk parity)
  // (rows of these small tables are ES = Epad + 4 elements apart: with a stride of Epad * 8 bytes, a multiple of the
  //  256-byte bank period, the 4 - 8 rows a matrix-pass instruction touches would all sit on the same banks)
  __host__ __device__ static size_t row_elems(int Epad) {
    const size_t ES = (size_t)Epad + 4;
    return (size_t)F::LDS_ELEMS + SV * ES + 2 * NC * ES + (STAGE ? 3 * NC * ES : 0);
  }
  __host__ __device__ static size_t lds_bytes(int Epad) {
    return (NF * row_elems(Epad) + T1) * sizeof(cpx<T>) + (2 * (size_t)Epad + 16) * sizeof(T);
  }
};

#ifndef GPA_PBS_F32_WAVES
#define GPA_PBS_F32_WAVES 3
#endif
#ifndef GPA_PBS_PAD_WAVES
#define GPA_PBS_PAD_WAVES 3   // f32, zero-padded rows
#endif
#ifndef GPA_PBS_F64_WAVES
#define GPA_PBS_F64_WAVES 2
#endif

#ifndef GPA_PBS_E8_WAVES
#define GPA_PBS_E8_WAVES 4    // eight elements per thread, f32: 128 VGPRs
#endif
#ifndef GPA_PBS_E8_F64_WAVES
#define GPA_PBS_E8_F64_WAVES 2
#endif
// NBL = live spectral registers.  The Gaussian keeps a band of bins; in the spectral register layout register i of
// every thread holds bin kappa(thread) + (L / EE) i, i.e. block i of L / EE consecutive bins.  The host rotates each
// peak's band to block 0 (an input phasor exp(-2 pi i s y / EE) of period EE and candidate frequencies wy + s / EE: the
// same products, the spectrum shifted by s blocks), so only registers 0 .. NBL-1 of the shared spectrum and of the
// candidate's Gaussian are live: fewer registers (the next candidate's Gaussian is requested before this
// candidate's stores), and the zeros of the other registers prune the first inverse pass at compile time.
#ifndef GPA_PBS_L13_WAVES
#define GPA_PBS_L13_WAVES 3   // 8192-point rows: one workgroup of 8 wavefronts per CU whatever the register budget
#endif
// PSI (a4, round 6): -angle of EVERY candidate's row goes out as one real per pixel (psi_out[list position][row][y]) for the
// phase-gradient stencil of wfr2_grad_opt (geometric_phase_analysis.py:763-813).  The values in registers are the lock-ins
// WITHOUT the candidate-independent phasor exp(2 pi i (ky + s / 16) y) -- compensation along x included --, so the stencil
// subtracts that phasor's phase step per column and needs no 2 pi (w - k) (launch_phasegrad, compensated form).
template <class T, int LG, bool PADDED, int EE, int NBL, bool PSI = false>
__global__ __launch_bounds__((PassBSGeom<T, LG, EE>::THREADS),
                             (EE == 8 ? (sizeof(T) == 8 ? GPA_PBS_E8_F64_WAVES : GPA_PBS_E8_WAVES)
                                      : (sizeof(T) == 8 ? GPA_PBS_F64_WAVES : (LG >= 13 ? GPA_PBS_L13_WAVES : (PADDED ? GPA_PBS_PAD_WAVES : GPA_PBS_F32_WAVES))))) void passB_shared_kernel(
    const cpx<T>* __restrict__ Tin, int n0, int n1, const T* __restrict__ Gb, const cpx<T>* __restrict__ twtab,
    const int* __restrict__ planeof, const int* __restrict__ order, const int* __restrict__ desc, const cpx<T>* __restrict__ pre_g,
    const cpx<T>* __restrict__ psi, const T* __restrict__ gtab, const cpx<T>* __restrict__ dx,
    const cpx<T>* __restrict__ dyc, const cpx<T>* __restrict__ rot16, int K, int E, int Epad, cpx<T>* out,
    int32_t* kidx, int P, int Bx, int raw, T* __restrict__ psi_out) {
  using F = WgFFT<T, LG, EE>;
  using G = PassBSGeom<T, LG, EE>;
  using V4 = typename MfmaVec<T>::type;
  constexpr int TPF = F::TPF, L = F::L, NC = G::NC;
  constexpr bool STAGE = G::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (a row owns whole wavefronts: its index is wave-uniform, which keeps every row base address in SGPRs)
  const int tid = threadIdx.x % TPF, f = __builtin_amdgcn_readfirstlane(threadIdx.x / TPF);
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * G::row_elems(Epad);
  upair<T>* strip = reinterpret_cast<upair<T>*>(lds + F::LDS_ELEMS);
  const int ES = Epad + 4;                       // row stride of the small tables
  cpx<T>* fixb = lds + F::LDS_ELEMS + G::SV * ES;
  cpx<T>* pre_l = fixb + 2 * NC * ES;            // [NC][ES]       (STAGE)
  cpx<T>* psi_l = pre_l + NC * ES;               // [2][NC][ES]    (STAGE)
  cpx<T>* t1 = reinterpret_cast<cpx<T>*>(smem) + (size_t)G::NF * G::row_elems(Epad);
  T* glds = reinterpret_cast<T*>(t1 + G::T1);
  const int row = blockIdx.x * G::NF + f;
  const bool valid = row < n0;
  const int rr = valid ? row : 0;
  const int p = blockIdx.y, pt = p % P, img = p / P;
  const int lane = threadIdx.x & 63, wrow = tid >> 6;

  // Hankel taps g(1 .. E) (zero beyond) and the pass-1 twiddles: first read after the barriers of the first transform
  for (int i = threadIdx.x; i < 2 * Epad + 16; i += G::THREADS) glds[i] = gtab[i];
  // (pass-1 twiddles: from the LDS table where registers are short -- f64, zero-padded rows -- and in registers for
  //  periodic f32 rows, which have had 20 registers to spare at 3 waves per SIMD since the winner stores became rare:
  //  1.227 -> 1.203 ms; the table's LDS stays allocated either way, the geometry does not know PADDED)
#ifndef GPA_PBS_TWREG_F32
#define GPA_PBS_TWREG_F32 1
#endif
  constexpr bool TWLK = G::TWL && !(GPA_PBS_TWREG_F32 && sizeof(T) == 4 && !PADDED);
  typename std::conditional<TWLK, typename F::TwiddlesP1Lds, typename F::Twiddles>::type tw;
  if constexpr (TWLK) {
    F::fill_pass1_table(t1, twtab, threadIdx.x, G::THREADS);
    __syncthreads();
    F::load_twiddles(tw, twtab, tid, t1);
  } else {
    F::load_twiddles(tw, twtab, tid);
  }

  T ab[EE];
#pragma unroll
  for (int i = 0; i < EE; ++i) ab[i] = T(0);
  const size_t obase = ((size_t)p * n0 + rr) * n1;
  // the winner's row as a buffer: a lane that does not win (or lies beyond the image) stores to an out-of-range
  // offset, which the hardware drops -- sixteen predicated stores per candidate without a branch
  const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc((void*)(out + obase), 0, n1 * (int)sizeof(cpx<T>), 0x00020000);
  const __amdgpu_buffer_rsrc_t krow = __builtin_amdgcn_make_buffer_rsrc((void*)(kidx ? kidx + obase : nullptr), 0, kidx ? n1 * 4 : 0, 0x00020000);
  constexpr int OOB = (int)0x80000000;
  // registers that hold the last E samples of the row (zero-padded rows: wherever n1 puts them)
  const int iA = PADDED ? (n1 - E) / TPF : EE - 1, iB = PADDED ? (n1 - 1) / TPF : EE - 1;

  cpx<T> X[NBL];
  T h[NBL];
  // the rotation phasor of this peak at the thread's columns (period EE in y: one value per thread) and at the
  // column of its tail sample
  const cpx<T> rot = rot16[pt * EE + (tid & (EE - 1))];
  const cpx<T> rot_tail = rot16[pt * EE + ((n1 - 1 - (tid < E ? tid : 0)) & (EE - 1))];
  // (f64: no prefetch since the stores became rare -- visiting order + wave-level skip -- so that loads issued behind them
  //  no longer wait for sixteen acknowledgements; the 16 registers it frees take the kernel from 68 to 12 bytes of
  //  scratch: pass B 3.06 -> 2.89 ms.  f32 measured the same either way and keeps it.)
#ifndef GPA_PBS_PREFETCH_F64
#define GPA_PBS_PREFETCH_F64 0
#endif
#ifndef GPA_PBS_PREFETCH_F32
#define GPA_PBS_PREFETCH_F32 1
#endif
  constexpr bool PREFETCH = NBL < EE && (sizeof(T) == 8 ? GPA_PBS_PREFETCH_F64 : GPA_PBS_PREFETCH_F32);
  for (int k = 0; k < K; ++k) {
    const int b = pt * K + k;      // position in visiting order: the candidate tables of this file
    const int ob = order[b];       // its position in the staged list: x-plane, compensation along x, reported index
    const int d = desc[b];
    if (d & 1) {
      // ---- a new x-plane: read its row once, take the end strips, forward transform -------------------------
      const cpx<T>* src = Tin + (((size_t)img * Bx + planeof[ob]) * n0 + rr) * n1;
      const cpx<T> cs0 = dx[(size_t)ob * n0 + rr];   // exp(-2 pi i (wx - kx) x): the same for every candidate of the plane
      const cpx<T> cs = cmul(cs0, rot);              // ... times the band rotation at this thread's columns
      cpx<T> tail = {T(0), T(0)};
      if (tid < E) tail = src[n1 - 1 - tid];
      cpx<T> XX[EE];
      if constexpr (PADDED) {
        // (zero-padded rows: the slots beyond the row read as zero through the range check of a buffer descriptor)
        const __amdgpu_buffer_rsrc_t srow = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n1 * (int)sizeof(cpx<T>), 0x00020000);
#pragma unroll
        for (int i = 0; i < EE; ++i)
          XX[i] = cmul(load_cpx<T>(srow, (tid + TPF * i) * (int)sizeof(cpx<T>), 0, GPA_PBS_NTLOAD ? 2 : 0), cs);
      } else {
#pragma unroll
        for (int i = 0; i < EE; ++i) XX[i] = cmul(GPA_PBS_NTLOAD ? load_once(src + tid + TPF * i) : src[tid + TPF * i], cs);
      }
      if (tid < Epad) {
        // strips in the two forms the matrix pass reads (it forms ONE real of t * p or t * conj(p) per lane as
        // u p.x + v p.y): end 0 = T(j), j < E (feeds the outputs at the row's end), end 1 = T(n - 1 - j)
        const bool in = tid < E;
        const cpx<T> s0 = in ? XX[0] : cpx<T>{T(0), T(0)};
        const cpx<T> s1 = in ? cmul(tail, cmul(cs0, rot_tail)) : cpx<T>{T(0), T(0)};
        if constexpr (G::SV == 4) {
          strip[0 * ES + tid] = {s0.x, -s0.y};   // end 0, real part of t p
          strip[1 * ES + tid] = {s0.y, s0.x};    // end 0, imaginary part
          strip[2 * ES + tid] = {s1.x, s1.y};    // end 1, real part of t conj(p)
          strip[3 * ES + tid] = {s1.y, -s1.x};   // end 1, imaginary part
        } else {
          strip[0 * ES + tid] = {s0.x, s0.y};
          strip[1 * ES + tid] = {s1.x, s1.y};
        }
      }
      F::forward(XX, lds, tid, tw);
      // only the band's blocks are kept (the transform's other outputs are dead code)
#pragma unroll
      for (int i = 0; i < NBL; ++i) X[i] = XX[i];
    }
    // ---- candidate b: shifted Gaussian, inverse transform (the matrix pass of a new chunk rides between its barriers)
    const int par = (d >> 7) & 1;   // parity of the chunk: which copy of the staged post-factors this candidate reads
    cpx<T> y[EE];
    {
      if (!PREFETCH || k == 0) load_gb<T, TPF, NBL>(h, Gb + (size_t)b * (NBL * TPF), tid);
      if constexpr (STAGE) {
        if (d & 2) {
          // phasors of this chunk's candidates into LDS: exp(2 pi i wy j) (strip pre-factors) and the post-factors
          const int nc = (d >> 4) & 7;
          for (int e = tid; e < NC * Epad; e += TPF) {
            const int c = e / Epad, j = e - c * Epad, cc = c < nc ? c : nc - 1;
            pre_l[c * ES + j] = pre_g[(size_t)(b + cc) * Epad + j];
            psi_l[(par * NC + c) * ES + j] = psi[(size_t)(b + cc) * Epad + j];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < EE; ++i) {
        if (i < NBL) y[i] = {X[i < NBL ? i : 0].x * h[i < NBL ? i : 0], X[i < NBL ? i : 0].y * h[i < NBL ? i : 0]};
        else y[i] = {T(0), T(0)};
      }
    }
    cpx<T> psL = {T(0), T(0)}, psR = {T(0), T(0)};
    const int a0R = PADDED ? 0 : TPF - 1 - tid;
    if constexpr (!STAGE) {
      // post-factors of the end fix, requested now, used after the transform
      psL = psi[(size_t)b * Epad + (tid < E ? tid : 0)];
      psR = psi[(size_t)b * Epad + (a0R < E ? a0R : 0)];
    }
    F::template inv_phase<0>(y, lds, tid, tw);
    __syncthreads();
    if ((d & 2) && !GPA_PBS_NOFIX) {
      // end-fix contraction for the candidates b .. b + nc - 1: D[a][n] = sum_j g(a + 1 + j) B[j][n], column
      // n = (end, candidate, re/im).  Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] of each k-step.
      const int nc = (d >> 4) & 7;
      const int n = lane & 15, kq = lane >> 4;
      const int c = (n >> 1) & (NC - 1), end = (n >> (1 + G::LOGNC)) & 1, reim = n & 1;
      const cpx<T>* pre = STAGE ? pre_l + c * ES : pre_g + (size_t)(b + (c < nc ? c : nc - 1)) * Epad;
      const upair<T>* sp = strip + (G::SV == 4 ? end * 2 + reim : end) * ES;
      // (two-strip form: u, v and the sign of v picked per lane)
      const bool swp = reim != 0;
      const T sgn = (end == 0) == swp ? T(1) : T(-1);
      const int Mt = (E + 15) >> 4;
      for (int mt = wrow; mt < Mt; mt += G::NW) {
        V4 acc0 = {T(0), T(0), T(0), T(0)}, acc1 = {T(0), T(0), T(0), T(0)};
        // k-steps whose taps g(16 mt + 1 + 4 s + ...) are not all beyond E
        int ns = ((E - 16 * mt - 1) >> 2) + 1;
        if (ns > (Epad >> 2)) ns = Epad >> 2;
        const T* ga = glds + 16 * mt + (lane & 15) + kq + 1;
        // four k-steps at a time, their twelve operand loads requested together: the contraction sits between two
        // barriers of the transform, on the critical path of the whole workgroup
        for (int s = 0; s < ns; s += 4) {
          upair<T> q[4];
          cpx<T> pp[4];
          T av[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            int j = 4 * (s + u) + kq;
            j = j < Epad ? j : Epad - 1;
            q[u] = sp[j];
            if constexpr (G::SV == 2) q[u] = {swp ? q[u].v : q[u].u, sgn * (swp ? q[u].u : q[u].v)};
            pp[u] = pre[j];
            av[u] = ga[4 * (s + u)];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            // (k-steps past ns multiply taps that are zero by zero-padded strips: harmless, and only whole batches run)
            const T bv = q[u].u * pp[u].x + q[u].v * pp[u].y;
            if (u & 1) acc1 = mfma16(av[u], bv, acc1);
            else acc0 = mfma16(av[u], bv, acc0);
          }
        }
        if (n < 4 * NC) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a0 = 16 * mt + mfma_row<T>(kq, r);
            reinterpret_cast<T*>(fixb + (c * 2 + end) * ES + a0)[reim] = acc0[r] + acc1[r];
          }
        }
      }
    }
    F::template inv_phase<1>(y, lds, tid, tw);
    __syncthreads();
    F::template inv_phase<2>(y, lds, tid, tw);
    if constexpr (F::P == 4) {
      __syncthreads();
      F::template inv_phase<3>(y, lds, tid, tw);
    }
    // ---- the outputs within E of either end get their wrapped pairs ------------------------------------------
#if !GPA_PBS_NOFIX
    {
      const int c = (d >> 2) & 3;
      const cpx<T>* pl = psi_l + (par * NC + c) * ES;
      if (tid < E) {
        const cpx<T> fx = fixb[(c * 2 + 1) * ES + tid];
        if constexpr (STAGE) psL = pl[tid];
        const cpx<T> t = cmulc(fx, psL);
        y[0].x += t.x;
        y[0].y += t.y;
      }
      if constexpr (!PADDED) {
        if (a0R < E) {
          const cpx<T> fx = fixb[(c * 2 + 0) * ES + a0R];
          if constexpr (STAGE) psR = pl[a0R];
          const cpx<T> t = cmul(fx, psR);
          y[EE - 1].x += t.x;
          y[EE - 1].y += t.y;
        }
      } else {
        // zero-padded rows: the last E samples sit in register iA and, where they straddle a register, iB = iA + 1
        // (wave-uniform): the two possible corrections are formed without touching y, then added to whichever
        // register holds them under scalar branches
        cpx<T> fA = {T(0), T(0)}, fB = {T(0), T(0)};
        {
          const int aA = n1 - 1 - (tid + TPF * iA), aB = aA - TPF;
          const bool inA = aA >= 0 && aA < E, inB = aB >= 0 && aB < E && iB != iA;
          const int qa = inA ? aA : 0, qb = inB ? aB : 0;
          const cpx<T> xa = fixb[(c * 2 + 0) * ES + qa], xb = fixb[(c * 2 + 0) * ES + qb];
          const cpx<T> pa = STAGE ? pl[qa] : psi[(size_t)b * Epad + qa], pb = STAGE ? pl[qb] : psi[(size_t)b * Epad + qb];
          const cpx<T> ta = cmul(xa, pa), tb2 = cmul(xb, pb);
          fA = {inA ? ta.x : T(0), inA ? ta.y : T(0)};
          fB = {inB ? tb2.x : T(0), inB ? tb2.y : T(0)};
        }
#pragma unroll
        for (int i = 0; i < EE; ++i) {
          if (i == iA) { y[i].x += fA.x; y[i].y += fA.y; }
          if (i == iB && iB != iA) { y[i].x += fB.x; y[i].y += fB.y; }
        }
      }
    }
#endif
    // the next candidate's Gaussian (NBL values) is requested BEFORE this candidate's stores: vmcnt counts loads and
    // stores in order, so loads issued behind the stores would wait for the stores' acknowledgements as well
    if constexpr (PREFETCH) {
      if (k + 1 < K) load_gb<T, TPF, NBL>(h, Gb + (size_t)(b + 1) * (NBL * TPF), tid);
    }
    const int kout = ob - pt * K;
    if constexpr (PSI) {
      if (valid) {
        T* prow = psi_out + ((size_t)(p * K + kout) * n0 + rr) * n1;
#pragma unroll
        for (int i = 0; i < EE; ++i) {
          const int yy = tid + TPF * i;
          if (!PADDED || yy < n1) prow[yy] = -atan2(y[i].y, y[i].x);
        }
      }
    }
    // ---- strict '>' in visiting order; the winner goes to memory at once ------------------------------------
    // (a register none of whose 64 lanes wins issues no store at all: with the likeliest winners visited first that is
    //  most registers of most candidates -- GPA_PBS_WAVESKIP=0 restores the sixteen dropped-offset stores per candidate)
#pragma unroll
    for (int i = 0; i < EE; ++i) {
      const int yy = tid + TPF * i;
      const T a = y[i].x * y[i].x + y[i].y * y[i].y;
      const bool win = a > ab[i];
      ab[i] = win ? a : ab[i];
#if GPA_PBS_WAVESKIP
      if (__builtin_amdgcn_ballot_w64(win) == 0) continue;
#endif
      // (byte offset; OOB = 0x80000000 stays out of range after the arithmetic shift below, never multiply it)
      // (zero-padded rows: the whole column offset goes into the range-checked voffset, so that the slots beyond
      //  the row are dropped by the descriptor's num_records; soffset is not range checked)
      const int woff = (win && valid) ? (PADDED ? yy : tid) * (int)sizeof(cpx<T>) : OOB;
      constexpr int SOFF = PADDED ? 0 : 1;
#if GPA_PBS_EXECSTORE
      if (woff != OOB) store_cpx(y[i], orow, woff, SOFF * i * TPF * (int)sizeof(cpx<T>));
      if (kidx && woff != OOB) __builtin_amdgcn_raw_buffer_store_b32(kout, krow, woff >> (sizeof(cpx<T>) == 8 ? 1 : 2), SOFF * i * TPF * 4, 0);
#elif !GPA_PBS_NOSTORE
      store_cpx(y[i], orow, woff, SOFF * i * TPF * (int)sizeof(cpx<T>));
      constexpr int SH = sizeof(cpx<T>) == 8 ? 1 : 2;   // complex byte offset -> int32 byte offset
      if (kidx) __builtin_amdgcn_raw_buffer_store_b32(kout, krow, woff >> SH, SOFF * i * TPF * 4, 0);
#endif
    }
  }
  if (!valid) return;
  if (raw) {
    // ---- raw mode (the fused driver): the winners stay as stored; the consumer accounts for the missing phasor
    // exp(2 pi i ky y) as a constant step of the phase along y (reconstruct_setup_kernel).  Only a pixel that nothing
    // ever won (an all-zero row) still has to be written: 0, winner -1.
#pragma unroll
    for (int i = 0; i < EE; ++i) {
      const int yy = tid + TPF * i;
      const bool never = !(ab[i] > T(0)) && (!PADDED || yy < n1);
      if (__builtin_amdgcn_ballot_w64(never) == 0) continue;
      if (never) {
        out[obase + yy] = cpx<T>{T(0), T(0)};
        if (kidx) kidx[obase + yy] = -1;
      }
    }
    return;
  }
  // ---- compensation to the peak centre: exp(2 pi i ky y), the same for every candidate ---------------------------
  // The winners were stored through the buffer descriptor by this very lane; the stores must have reached L2 before
  // the row is read back (the compiler sees no alias between the descriptor and a plain pointer, and a load may
  // overtake a store in flight), and the read goes to L2 (sc1), not to whatever L1 holds.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const cpx<T>* dyp = dyc + (size_t)pt * n1;
  cpx<T> w[EE];
#pragma unroll
  for (int i = 0; i < EE; ++i) {
    const int yy = tid + TPF * i;
    w[i] = PADDED ? load_cpx<T>(orow, yy * (int)sizeof(cpx<T>), 0) : load_cpx<T>(orow, tid * (int)sizeof(cpx<T>), i * TPF * (int)sizeof(cpx<T>));
  }
#pragma unroll
  for (int i = 0; i < EE; ++i) {
    const int yy = tid + TPF * i;
    if (!PADDED || yy < n1) {
      cpx<T> v = {T(0), T(0)};
      if (ab[i] > T(0)) v = cmul(w[i], dyp[yy]);
      else if (kidx) kidx[obase + yy] = -1;
      out[obase + yy] = v;
    }
  }
}

template <class T>
__device__ __forceinline__ cpx<T> unit_phasor_s(double cycles) {
  double fr = cycles - rint(cycles);
  double s, c;
  sincospi(2.0 * fr, &s, &c);
  return {(T)c, (T)s};
}

// ---------------------------------------------------------------------------------------------------------------
// tables: the shifted Gaussian of every candidate in the spectral register layout, the post-factors of the end fix
// and the compensation phasor of every peak, all evaluated in double
// ---------------------------------------------------------------------------------------------------------------

template <class T, int LG, int EE>
__global__ __launch_bounds__(256) void shared_tables_kernel(const double* __restrict__ wys, const double* __restrict__ kr,
                                                           const int* __restrict__ shifts, const double* __restrict__ taps,
                                                           int Etab, int n1, int E, int Epad, int wrapped, int K, int nbl,
                                                           T* __restrict__ Gb, cpx<T>* __restrict__ psi,
                                                           cpx<T>* __restrict__ dyc, cpx<T>* __restrict__ pre,
                                                           cpx<T>* __restrict__ rot16) {
  using F = WgFFT<T, LG, EE>;
  constexpr int L = F::L, TPF = F::TPF;
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const double wy = wys[b];     // the candidate's frequency plus its peak's band rotation s / EE
  if (idx < nbl * TPF) {
    // G_b[k] = sum_m g(m) exp(-2 pi i (wy + k / L) m) over the signed lags: real, g even; live registers only
    const int i = idx / TPF, t = idx % TPF;
    const int kbin = F::spec_index(t, i);
    const double fq = wy + (double)kbin / (double)L;
    double acc = 0;
    for (int m = Etab; m >= 1; --m) {
      const double ph = fq * (double)m;
      acc += taps[m] * cospi(2.0 * (ph - rint(ph)));
    }
    Gb[(size_t)b * (nbl * TPF) + idx] = (T)((taps[0] + 2.0 * acc) / (double)L);
  }
  if (idx < Epad) {
    // (phi_b - [row is periodic]) exp(2 pi i wy (a0 + 1)),  phi_b = exp(-2 pi i wy n);  pre-factor exp(2 pi i wy j)
    double ps, pc, qs, qc;
    const double c0 = -wy * (double)n1, c1 = wy * (double)(idx + 1);
    sincospi(2.0 * (c0 - rint(c0)), &ps, &pc);
    sincospi(2.0 * (c1 - rint(c1)), &qs, &qc);
    pc -= wrapped ? 1.0 : 0.0;
    cpx<T> v = {(T)(pc * qc - ps * qs), (T)(pc * qs + ps * qc)};
    if (idx >= E) v = {T(0), T(0)};
    psi[(size_t)b * Epad + idx] = v;
    pre[(size_t)b * Epad + idx] = unit_phasor_s<T>(wy * (double)idx);
  }
  if (b % K == 0) {
    const int p = b / K;
    const double sh = (double)shifts[p] / (double)EE;
    if (idx < n1) dyc[(size_t)p * n1 + idx] = unit_phasor_s<T>((kr[2 * b + 1] + sh) * (double)idx);
    if (idx < EE) rot16[p * EE + idx] = unit_phasor_s<T>(-sh * (double)idx);
  }
}

template <class T, int LG, int EE>
static hipError_t run_shared_tables(const Axis& a1, const double* wys, const double* kr, const int* shifts, const double* taps,
                                    int Etab, int E, int Epad, int B, int K, int nbl, const PassBSharedTables& st, hipStream_t s) {
  int len = a1.L > a1.n ? a1.L : a1.n;
  if (len < Epad) len = Epad;
  dim3 grid((len + 255) / 256, B);
  shared_tables_kernel<T, LG, EE><<<grid, 256, 0, s>>>(wys, kr, shifts, taps, Etab, a1.n, E, Epad, a1.padded ? 0 : 1, K, nbl,
                                                       (T*)st.Gb, (cpx<T>*)st.psi, (cpx<T>*)st.dyc, (cpx<T>*)st.pre,
                                                       (cpx<T>*)st.rot16);
  return hipGetLastError();
}

hipError_t launch_shared_tables(int dtype, const Axis& a1, const double* wys, const double* kr, const int* shifts,
                                const double* taps, int Etab, int E, int Epad, int B, int K, int nbl, const PassBSharedTables& st,
                                hipStream_t s, int elems) {
#ifdef GPA_PBS_BUILD_E8      // the eight-element instantiations: measured slower (profiles/r03_passB_variants.txt), not built by default
#define CASE_T8(LG) CASE_T(LG, 8)
#else
#define CASE_T8(LG)
#endif
#define CASE_T(LG, EE)                                                                                                  \
  if (a1.lg == LG && elems == EE)                                                                                       \
    return dtype == 0 ? run_shared_tables<float, LG, EE>(a1, wys, kr, shifts, taps, Etab, E, Epad, B, K, nbl, st, s)    \
                      : run_shared_tables<double, LG, EE>(a1, wys, kr, shifts, taps, Etab, E, Epad, B, K, nbl, st, s);
  CASE_T(11, 16) CASE_T(12, 16) CASE_T(13, 16) CASE_T8(12)
#undef CASE_T
  return hipErrorInvalidValue;
}

template <class T, int LG, bool PADDED, int EE, int NBL, bool PSI = false>
static hipError_t run_passB_shared(const Axis& a1, int n0, const void* Tbuf, const void* tw1, const SweepTables& tb,
                                   const PassBSharedTables& st, int E, int Epad, int P, int K, void* out, int32_t* kidx,
                                   hipStream_t s, int nimg, int Bx, bool raw, void* psi_out = nullptr) {
  using G = PassBSGeom<T, LG, EE>;
  size_t lds = G::lds_bytes(Epad);
  if (lds > 160 * 1024 || E > G::TPF || Epad > G::TPF) return hipErrorInvalidValue;
  // (experiment switch PBS_LDS_PAD=<bytes>: extra dynamic LDS per workgroup, i.e. fewer rows per CU, so that kernels of other
  //  streams -- the unwrap of the previous image, bench.py --inflight 2 -- can share the CUs with this one)
  if (opt_set(OPT_PBS_LDS_PAD)) lds = std::min<size_t>(160 * 1024, lds + (size_t)opt(OPT_PBS_LDS_PAD).num);
  auto kern = passB_shared_kernel<T, LG, PADDED, EE, NBL, PSI>;
  // (the dynamic LDS size depends on Epad: raise the limit whenever a larger one comes along)
  static int lds_set[32] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds_set[dev & 31] < (int)lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set[dev & 31] = (int)lds;
  }
  dim3 grid((n0 + G::NF - 1) / G::NF, P * nimg);
  GPA_PROF(PSI ? "passB_shared_phases_kernel" : "passB_shared_kernel", s);
  kern<<<grid, G::THREADS, lds, s>>>((const cpx<T>*)Tbuf, n0, a1.n, (const T*)st.Gb, (const cpx<T>*)tw1, tb.planeof, st.order, st.desc,
                                     (const cpx<T>*)st.pre, (const cpx<T>*)st.psi, (const T*)st.gtab, (const cpx<T>*)tb.dx,
                                     (const cpx<T>*)st.dyc, (const cpx<T>*)st.rot16, K, E, Epad, (cpx<T>*)out, kidx, P, Bx, raw ? 1 : 0, (T*)psi_out);
  return hipGetLastError();
}

// elements per thread the kernel runs this axis with
int passB_shared_elems(int dtype, const Axis& a1) {
  (void)dtype;
#ifdef GPA_PBS_BUILD_E8
  const bool e8 = opt_set(OPT_PBS_E8);   // experiment switch
  return (a1.lg == 12 && e8) ? 8 : 16;
#else
  (void)a1;
  return 16;
#endif
}

bool passB_shared_supports(int dtype, const Axis& a1, int E) {
  // 1024-point rows (one wavefront per row: the whole matrix pass and both end fixes on it) measured slower than
  // the per-candidate kernel (1024^2, 3 x 16: 189 against 131 us): from 2048 points on
  // (8192-point rows: f32 only -- four passes, one workgroup of 512 threads per CU)
  if (a1.lg < 11 || a1.lg > (dtype == 0 ? 13 : 12)) return false;
  const int tpf = a1.L / 16;
  const int Epad = (E + 15) & ~15;
  if (E < 1 || Epad > tpf || 2 * E > a1.n) return false;
  if (a1.padded && a1.n + E > a1.L) return false;
  const size_t csz = dtype == 0 ? 8 : 16;
  const int nf = tpf >= GPA_PBS_MINTHREADS ? 1 : GPA_PBS_MINTHREADS / tpf, nc = dtype == 0 ? 4 : 2;
  const size_t ES = (size_t)Epad + 4;
  const size_t stage = dtype == 0 ? 3 * nc * ES : 0;
  const size_t lds = (nf * ((size_t)(a1.L + a1.L / 16) + (dtype == 0 ? 4 : 2) * ES + 2 * nc * ES + stage) + 16 * 6) * csz +
                     (2 * (size_t)Epad + 16) * (csz / 2);
  return lds <= 160 * 1024;
}

// the live-register counts that are built: the smallest one that holds `need` blocks
int passB_shared_nbl(int dtype, int need) {
  if (dtype == 0 && need <= 6) return 6;
  if (need <= 8) return 8;
  return 16;
}

// the a4 form: the same sweep, and -angle of every candidate's lock-in (without the candidate-independent phasor along y) into
// psi_out [P][K (list positions)][n0][n1] reals.  Instantiated for the row classes the pipeline uses (2048 / 4096 / f32 8192).
hipError_t launch_passB_shared_phases(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* tw1, const SweepTables& tb,
                                      const PassBSharedTables& st, int E, int Epad, int P, int K, void* out, int32_t* kidx,
                                      void* psi_out, hipStream_t s, int Bx, int elems, int nbl) {
  if (!psi_out || !kidx) return hipErrorInvalidValue;
#define CALL_P(T, LG, EE, NBL) \
  (a1.padded ? run_passB_shared<T, LG, true, EE, NBL, true>(a1, n0, Tbuf, tw1, tb, st, E, Epad, P, K, out, kidx, s, 1, Bx, false, psi_out) \
             : run_passB_shared<T, LG, false, EE, NBL, true>(a1, n0, Tbuf, tw1, tb, st, E, Epad, P, K, out, kidx, s, 1, Bx, false, psi_out))
#define CASE_P(LG, EE, NBL) \
  if (a1.lg == LG && elems == EE && nbl == NBL) return dtype == 0 ? CALL_P(float, LG, EE, NBL) : CALL_P(double, LG, EE, NBL);
#define CASE_PF32(LG, EE, NBL) \
  if (a1.lg == LG && elems == EE && nbl == NBL && dtype == 0) return CALL_P(float, LG, EE, NBL);
  CASE_PF32(11, 16, 6) CASE_PF32(12, 16, 6) CASE_PF32(13, 16, 6) CASE_PF32(13, 16, 8) CASE_PF32(13, 16, 16)
  CASE_P(11, 16, 8) CASE_P(12, 16, 8) CASE_P(11, 16, 16) CASE_P(12, 16, 16)
#undef CASE_PF32
#undef CASE_P
#undef CALL_P
  return hipErrorInvalidValue;
}

hipError_t launch_passB_shared(int dtype, const Axis& a1, int n0, const void* Tbuf, const void* tw1, const SweepTables& tb,
                               const PassBSharedTables& st, int E, int Epad, int P, int K, void* out, int32_t* kidx,
                               hipStream_t s, int nimg, int Bx, int elems, int nbl, bool raw) {
#define CALL_S(T, LG, EE, NBL) \
  (a1.padded ? run_passB_shared<T, LG, true, EE, NBL>(a1, n0, Tbuf, tw1, tb, st, E, Epad, P, K, out, kidx, s, nimg, Bx, raw) \
             : run_passB_shared<T, LG, false, EE, NBL>(a1, n0, Tbuf, tw1, tb, st, E, Epad, P, K, out, kidx, s, nimg, Bx, raw))
#define CASE_S(LG, EE, NBL) \
  if (a1.lg == LG && elems == EE && nbl == NBL) return dtype == 0 ? CALL_S(float, LG, EE, NBL) : CALL_S(double, LG, EE, NBL);
#define CASE_F32(LG, EE, NBL) \
  if (a1.lg == LG && elems == EE && nbl == NBL && dtype == 0) return CALL_S(float, LG, EE, NBL);
  CASE_F32(11, 16, 6) CASE_F32(12, 16, 6) CASE_F32(13, 16, 6) CASE_F32(13, 16, 8) CASE_F32(13, 16, 16)
  CASE_S(11, 16, 8) CASE_S(12, 16, 8) CASE_S(11, 16, 16) CASE_S(12, 16, 16)
#ifdef GPA_PBS_BUILD_E8
  CASE_S(12, 8, 8)
#endif
#undef CASE_F32
#undef CASE_S
#undef CALL_S
  return hipErrorInvalidValue;
}

}  // namespace gpa
