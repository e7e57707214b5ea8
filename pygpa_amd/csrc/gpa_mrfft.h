// Workgroup FFT of any length n = 2^a 3^b 5^c 7^d 11^e 13^f, the transform resident in LDS.
//
// The power-of-two engine (gpa_fft.h) keeps a transform in registers, 16 elements per thread, which ties the
// length to 16 * threads and the radices to powers of two.  Image sizes that are not powers of two (500, 1000,
// 1500, 3000 pixels; tile + 2 halo windows) used to be transformed through Bluestein's chirp-z on that engine:
// two FFTs of length >= 2n - 1 per DFT of length n.  This engine transforms length n itself:
//
//   * Stockham autosort, decimation in frequency: pass p (radix R, s = product of the earlier radices) takes
//     butterfly b = q + s pp (q < s, pp < n / (s R)) from elements b + j n/R (j < R: lane-consecutive, no bank
//     conflicts), and writes X_k w_(n/s)^(pp k) to q + s (R pp + k).  Natural order in, natural order out,
//     no digit reversal.
//   * one LDS buffer: every thread loads its butterflies (<= 16 values) into registers, barrier, butterflies and
//     twiddles in registers, stores, barrier.  T = threads per transform is chosen by the host so that
//     ceil(n / (R T)) * R <= 16 for every radix (n = 3000: T = 256).
//   * radices 16 / 8 / 4 / 2 come from gpa_fft.h's register DFTs, odd primes from one generic butterfly that
//     pairs a_j with a_(R-j) (half the multiplications of the plain O(R^2) sum), 6 / 10 / 12 / 14 / 15 from those by
//     Good-Thomas index maps (no inner twiddles).  The host picks the factorisation with the fewest passes.
//   * twiddles w_n^i from one table of n entries in global memory (L1 / L2 resident: 24 KB at n = 3000 f32; a copy
//     in LDS measured no faster and costs occupancy -- tools/ubench/mrfft_bench.hip); pp k s < n needs no modulo.
//     b / s is a multiply-high by ceil(2^32 / s), exact for b s < 2^32 (n <= 2^14 here).
//
// The per-thread steps are GPA_HD: tests/host/mrfft_emulator.cpp runs them thread by thread on the CPU.
#pragma once
#include <stdint.h>

#include "gpa_fft.h"

namespace gpa {

constexpr int MR_MAXPASS = 12;
constexpr int MR_REGS = 16;

struct MrPlan {
  int n;                        // transform length
  int np;                       // passes
  int T;                        // threads per transform (multiple of 64)
  int radix[MR_MAXPASS];
  int stride[MR_MAXPASS];       // s_p = product of the radices of the passes before p
  unsigned magic[MR_MAXPASS];   // ceil(2^32 / s_p)
};

// LDS image: one element of padding per 32 (stride-R stores of the first passes would otherwise pile onto a few banks)
GPA_HD int mr_pad(int i) { return i + (i >> 5); }
GPA_HD int mr_lds_elems(int n) { return mr_pad(n - 1) + 1; }

GPA_HD unsigned mr_mulhi(unsigned a, unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

// (cos, sin)(2 pi i / R) of the odd radices
template <int R> struct MrTrig;
template <> struct MrTrig<3> {
  static GPA_HD double c(int i) { constexpr double t[3] = {1.0, -0.5, -0.5}; return t[i]; }
  static GPA_HD double s(int i) { constexpr double t[3] = {0.0, 0.866025403784438646764, -0.866025403784438646764}; return t[i]; }
};
template <> struct MrTrig<5> {
  static GPA_HD double c(int i) { constexpr double t[5] = {1.0, 0.309016994374947424102, -0.809016994374947424102, -0.809016994374947424102, 0.309016994374947424102}; return t[i]; }
  static GPA_HD double s(int i) { constexpr double t[5] = {0.0, 0.951056516295153572116, 0.587785252292473129169, -0.587785252292473129169, -0.951056516295153572116}; return t[i]; }
};
template <> struct MrTrig<7> {
  static GPA_HD double c(int i) { constexpr double t[7] = {1.0, 0.623489801858733530525, -0.222520933956314404289, -0.900968867902419126236, -0.900968867902419126236, -0.222520933956314404289, 0.623489801858733530525}; return t[i]; }
  static GPA_HD double s(int i) { constexpr double t[7] = {0.0, 0.781831482468029808708, 0.974927912181823607018, 0.433883739117558120476, -0.433883739117558120476, -0.974927912181823607018, -0.781831482468029808708}; return t[i]; }
};
template <> struct MrTrig<11> {
  static GPA_HD double c(int i) { constexpr double t[11] = {1.0, 0.841253532831181168862, 0.415415013001886425529, -0.142314838273285140444, -0.654860733945285064057, -0.95949297361449738989, -0.95949297361449738989, -0.654860733945285064057, -0.142314838273285140444, 0.415415013001886425529, 0.841253532831181168862}; return t[i]; }
  static GPA_HD double s(int i) { constexpr double t[11] = {0.0, 0.540640817455597582108, 0.909631995354518371412, 0.989821441880932732376, 0.755749574354258283774, 0.281732556841429697711, -0.281732556841429697711, -0.755749574354258283774, -0.989821441880932732376, -0.909631995354518371412, -0.540640817455597582108}; return t[i]; }
};
template <> struct MrTrig<13> {
  static GPA_HD double c(int i) { constexpr double t[13] = {1.0, 0.8854560256532098959, 0.568064746731155802512, 0.120536680255323053349, -0.35460488704253562597, -0.748510748171101098635, -0.970941817426052027157, -0.970941817426052027157, -0.748510748171101098635, -0.35460488704253562597, 0.120536680255323053349, 0.568064746731155802512, 0.8854560256532098959}; return t[i]; }
  static GPA_HD double s(int i) { constexpr double t[13] = {0.0, 0.464723172043768545656, 0.82298386589365639458, 0.992708874098053992801, 0.93501624268541482344, 0.663122658240795202377, 0.239315664287557767149, -0.239315664287557767149, -0.663122658240795202377, -0.93501624268541482344, -0.992708874098053992801, -0.82298386589365639458, -0.464723172043768545656}; return t[i]; }
};

template <class T, int R> GPA_HD void mr_bfly(cpx<T>* a);

constexpr int mr_inv_mod(int a, int m) {
  for (int x = 1; x < m; ++x)
    if ((a * x) % m == 1) return x;
  return 0;
}

// R = R1 R2 with coprime factors: Good-Thomas index maps turn the DFT into an R1 x R2 two-dimensional one WITHOUT
// twiddles between the two stages: input j = (R2 j1 + R1 j2) mod R, output k with k = k1 (mod R1), k = k2 (mod R2).
// All indices are compile-time constants after unrolling: the maps cost no instructions.
template <class T, int R1, int R2>
GPA_HD void mr_bfly_pfa(cpx<T>* a) {
  constexpr int R = R1 * R2, T1 = mr_inv_mod(R1 % R2, R2), T2 = mr_inv_mod(R2 % R1, R1);
  cpx<T> y[R2][R1];
#pragma unroll
  for (int j2 = 0; j2 < R2; ++j2) {
#pragma unroll
    for (int j1 = 0; j1 < R1; ++j1) y[j2][j1] = a[(R2 * j1 + R1 * j2) % R];
    mr_bfly<T, R1>(y[j2]);
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    cpx<T> z[R2];
#pragma unroll
    for (int j2 = 0; j2 < R2; ++j2) z[j2] = y[j2][k1];
    mr_bfly<T, R2>(z);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) a[(R2 * T2 * k1 + R1 * T1 * k2) % R] = z[k2];
  }
}

// forward DFT of R values in place, natural order.  Odd prime R: with P_j = a_j + a_(R-j), M_j = a_j - a_(R-j),
//   X_k, X_(R-k) = a_0 + sum_j cos(2 pi j k / R) P_j  -/+  i sum_j sin(2 pi j k / R) M_j
template <class T, int R>
GPA_HD void mr_bfly(cpx<T>* a) {
  if constexpr (R == 2 || R == 4 || R == 8 || R == 16) {
    dft_regs<R, false>(a);
  } else if constexpr (R == 6) {
    mr_bfly_pfa<T, 3, 2>(a);
  } else if constexpr (R == 10) {
    mr_bfly_pfa<T, 5, 2>(a);
  } else if constexpr (R == 12) {
    mr_bfly_pfa<T, 3, 4>(a);
  } else if constexpr (R == 14) {
    mr_bfly_pfa<T, 7, 2>(a);
  } else if constexpr (R == 15) {
    mr_bfly_pfa<T, 3, 5>(a);
  } else {
    constexpr int H = (R - 1) / 2;
    cpx<T> P[H], M[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) {
      P[j - 1] = a[j] + a[R - j];
      M[j - 1] = a[j] - a[R - j];
    }
    const cpx<T> a0 = a[0];
    cpx<T> x0 = a0;
#pragma unroll
    for (int j = 0; j < H; ++j) x0 = x0 + P[j];
    a[0] = x0;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
      cpx<T> re = a0, im = {T(0), T(0)};
#pragma unroll
      for (int j = 1; j <= H; ++j) {
        const T c = (T)MrTrig<R>::c((j * k) % R), s = (T)MrTrig<R>::s((j * k) % R);
        re.x += c * P[j - 1].x;
        re.y += c * P[j - 1].y;
        im.x += s * M[j - 1].x;
        im.y += s * M[j - 1].y;
      }
      a[k] = {re.x + im.y, re.y - im.x};
      a[R - k] = {re.x - im.y, re.y + im.x};
    }
  }
}

// The LDS image and the twiddle table are addressed through pointers to their SCALARS (element i = scalars 2i, 2i + 1)
// so that mr_pass can hand over address-space-qualified pointers: behind a real function call hipcc no longer knows
// that `lds` is LDS and `W` global memory and emitted FLAT loads / stores for both (flat_load_dwordx2 where the
// inlined code has ds_read_b64).
template <class T, class P> GPA_HD cpx<T> mr_get(P p, int i) { return {p[2 * i], p[2 * i + 1]}; }
template <class T, class P> GPA_HD void mr_put(P p, int i, cpx<T> v) { p[2 * i] = v.x; p[2 * i + 1] = v.y; }

// pass, first half: this thread's butterflies from LDS into registers
template <class T, int R, class LP>
GPA_HD void mr_load(cpx<T>* x, LP lds, int n, int tid, int Tn) {
  constexpr int NB = MR_REGS / R;
  const int nb = n / R;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int b = tid + u * Tn;
    if (b < nb) {
#pragma unroll
      for (int j = 0; j < R; ++j) x[u * R + j] = mr_get<T>(lds, mr_pad(b + j * nb));
    }
  }
}

// pass, second half (after a barrier): butterflies, twiddles, autosort store.  W: w_n^i = exp(-2 pi i / n * i), i < n,
// (entry i at mr_pad(i): the table may equally be a padded LDS copy)
template <class T, int R, class LP, class WP>
GPA_HD void mr_store(cpx<T>* x, LP lds, int n, int s, unsigned magic, int tid, int Tn, WP W) {
  constexpr int NB = MR_REGS / R;
  const int nb = n / R;
  const bool last = s * R == n;   // pp == 0 throughout: no twiddles
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int b = tid + u * Tn;
    if (b < nb) {
      mr_bfly<T, R>(x + u * R);
      const int pp = s == 1 ? b : (int)mr_mulhi((unsigned)b, magic), q = b - pp * s;
      const int base = q + s * R * pp, ws = pp * s;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        cpx<T> v = x[u * R + k];
        if (k > 0 && !last) v = cmul(v, mr_get<T>(W, mr_pad(ws * k)));
        mr_put<T>(lds, mr_pad(base + s * k), v);
      }
    }
  }
}

#if defined(__HIPCC__)
// one pass as a real function call: inlined into a switch inside the pass loop, the nine radix bodies share one
// register allocation problem and hipcc ends up with > 200 VGPRs (or the pass registers in scratch) where each body
// alone needs 60-90; as callees they keep their own budgets and the kernel pays one s_swappc per pass
// (CAP = the calling kernel's launch bound: one callee per bound, so that a kernel limited to 256 threads does not
// inherit the 128-register budget of the 1024-thread variant)
template <class T, int R, int CAP>
__device__ __attribute__((noinline)) void mr_pass(cpx<T>* lds, int n, int s, unsigned mg, int tid, int Tn,
                                                  const cpx<T>* W) {
  using lds_scalar = __attribute__((address_space(3))) T;
  using glb_scalar = const __attribute__((address_space(1))) T;
  lds_scalar* l = (lds_scalar*)reinterpret_cast<T*>(lds);        // (the caller's LDS image)
  glb_scalar* w = (glb_scalar*)reinterpret_cast<const T*>(W);    // (the twiddle table in global memory)
  cpx<T> x[MR_REGS];
  mr_load<T, R>(x, l, n, tid, Tn);
  __syncthreads();
  mr_store<T, R>(x, l, n, s, mg, tid, Tn, w);
}

// the whole transform; the data must be in LDS and a barrier passed before the call, ends with a barrier.
// Every thread of the workgroup must call (the launch geometry is blockDim = transforms per workgroup * pl.T).
template <int CAP, class T>
__device__ __forceinline__ void mr_run(cpx<T>* lds, const MrPlan& pl, const cpx<T>* W, int tid) {
  for (int p = 0; p < pl.np; ++p) {
    const int s = pl.stride[p];
    const unsigned mg = pl.magic[p];
#define GPA_MR_CASE(R) case R: mr_pass<T, R, CAP>(lds, pl.n, s, mg, tid, pl.T, W); break;
    switch (pl.radix[p]) {
      GPA_MR_CASE(16) GPA_MR_CASE(8) GPA_MR_CASE(4) GPA_MR_CASE(2) GPA_MR_CASE(3) GPA_MR_CASE(5) GPA_MR_CASE(7)
      GPA_MR_CASE(11) GPA_MR_CASE(13) GPA_MR_CASE(6) GPA_MR_CASE(10) GPA_MR_CASE(12) GPA_MR_CASE(14) GPA_MR_CASE(15)
    }
#undef GPA_MR_CASE
    __syncthreads();
  }
}
#endif

// A DFT of length n as the kernels see it: the transform above when n is smooth, otherwise Bluestein's chirp-z on a
// smooth length L >= 2n - 1 (NOT a power of two: n = 1392 runs L = 2800 = 2^4 5^2 7, not 4096):
//   X[k] = conj(c_k) sum_m (x[m] conj(c_m)) c_(k-m),  c_m = exp(i pi m^2 / n)
// = chirp multiply -> FFT_L -> multiply by bspec = FFT_L(b) / L (b[m] = b[L-m] = c_m) -> inverse FFT_L -> chirp multiply,
// all on the LDS image.  Tables (double on the host): chirp[m], m < n; bspec[k], k < L, natural order.
struct MrDft {
  MrPlan pl;    // the FFT that is run: length n, or L for chirp-z
  int n;        // length of the DFT
  int blue;     // 1: chirp-z
};

#if defined(__HIPCC__)
// in: x[m] at lds[mr_pad(m)], m < n (slots n .. L-1 may hold anything); out: X[k] at lds[mr_pad(k)], k < n.
// A barrier must have been passed before the call; ends with a barrier.  Every thread of the workgroup calls.
template <int CAP, class T>
__device__ __forceinline__ void mr_dft(cpx<T>* lds, const MrDft& d, const cpx<T>* W, const cpx<T>* __restrict__ chirp,
                                       const cpx<T>* __restrict__ bspec, int tid) {
  if (!d.blue) {
    mr_run<CAP>(lds, d.pl, W, tid);
    return;
  }
  const int n = d.n, L = d.pl.n, Tn = d.pl.T;
  for (int m = tid; m < L; m += Tn) lds[mr_pad(m)] = m < n ? cmulc(lds[mr_pad(m)], chirp[m]) : cpx<T>{T(0), T(0)};
  __syncthreads();
  mr_run<CAP>(lds, d.pl, W, tid);
  for (int k = tid; k < L; k += Tn) {
    const cpx<T> v = cmul(lds[mr_pad(k)], bspec[k]);
    lds[mr_pad(k)] = {v.x, -v.y};   // conjugated: the inverse transform is conj(FFT(conj .)), 1/L is in bspec
  }
  __syncthreads();
  mr_run<CAP>(lds, d.pl, W, tid);
  for (int k = tid; k < n; k += Tn) {
    const cpx<T> v = lds[mr_pad(k)];
    lds[mr_pad(k)] = cmulc(cpx<T>{v.x, -v.y}, chirp[k]);
  }
  __syncthreads();
}
#endif

// host: fewest passes with radices <= 16 (composite ones included: 3000 = 15 10 10 2 instead of 5 5 5 3 8; every pass
// costs one trip of the whole transform through LDS whatever its radix)
inline int mr_min_passes(int n, int* out) {
  static const int allowed[] = {16, 15, 14, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2};
  if (n == 1) return 0;
  int best = 1 << 20, best_min = 0, pick[64], sub[64];
  for (int r : allowed) {
    if (n % r) continue;
    const int c = mr_min_passes(n / r, sub);
    if (c < 0 || c + 1 > 60) continue;
    int mn = r;
    for (int i = 0; i < c; ++i) mn = sub[i] < mn ? sub[i] : mn;
    // among equally short factorisations the most balanced one (largest smallest radix): the pass of the smallest
    // radix has the most butterflies and sets the thread count
    if (c + 1 > best || (c + 1 == best && mn <= best_min)) continue;
    best = c + 1;
    best_min = mn;
    pick[0] = r;
    for (int i = 0; i < c; ++i) pick[i + 1] = sub[i];
  }
  if (best == 1 << 20) return -1;
  for (int i = 0; i < best; ++i) out[i] = pick[i];
  return best;
}

// host: factorise n and size the thread group; false if n has a prime factor > 13 (those lengths stay on Bluestein)
inline bool mr_make_plan(int n, MrPlan* pl) {
  if (n < 2) return false;
  {
    int m = n;
    for (int r : {2, 3, 5, 7, 11, 13})
      while (m % r == 0) m /= r;
    if (m != 1) return false;
  }
  int radix[64];
  const int np = mr_min_passes(n, radix);
  if (np < 1 || np > MR_MAXPASS) return false;
  // odd radices first: the first passes store with a stride of R elements, which is free of bank conflicts for odd R
  for (int i = 0; i < np; ++i)
    for (int j = i + 1; j < np; ++j) {
      const bool oi = radix[i] & 1, oj = radix[j] & 1;
      if ((!oi && oj) || (oi == oj && radix[j] > radix[i])) { const int t = radix[i]; radix[i] = radix[j]; radix[j] = t; }
    }
  pl->n = n;
  pl->np = np;
  // threads per transform: one butterfly per thread and pass where 256 threads suffice (short transforms are
  // latency-bound: fewer rounds per pass), otherwise as few as the 16 registers of a thread allow
  int need = 1, want = 1, s = 1;
  for (int p = 0; p < np; ++p) {
    const int R = radix[p], nbmax = MR_REGS / R, nb = n / R;
    if ((nb + nbmax - 1) / nbmax > need) need = (nb + nbmax - 1) / nbmax;
    if (nb > want) want = nb;
    pl->radix[p] = R;
    pl->stride[p] = s;
    pl->magic[p] = s == 1 ? 0u : (unsigned)(((1ull << 32) + (uint64_t)s - 1) / (uint64_t)s);   // (s == 1: b / 1 = b)
    s *= R;
  }
  need = (need + 63) / 64 * 64;
  want = (want + 63) / 64 * 64;
  pl->T = want <= 256 ? want : need;
  if (pl->T < need) pl->T = need;
  return pl->T <= 1024 && (uint64_t)n * (uint64_t)n < (1ull << 32);
}

// host: the DFT plan of an arbitrary length: direct when n is smooth, otherwise chirp-z on the smallest smooth
// L >= 2n - 1 that has a plan; false if nothing fits max_lds_elems complex elements of LDS
inline bool mr_make_dft(int n, int max_lds_elems, MrDft* d) {
  d->n = n;
  d->blue = 0;
  if (mr_make_plan(n, &d->pl)) return mr_lds_elems(n) <= max_lds_elems;
  d->blue = 1;
  for (int L = 2 * n - 1; mr_lds_elems(L) <= max_lds_elems; ++L)
    if (mr_make_plan(L, &d->pl)) return true;
  return false;
}

}  // namespace gpa
