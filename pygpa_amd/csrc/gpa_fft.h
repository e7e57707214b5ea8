// Workgroup-level power-of-two FFT for gfx950: radix-2^b butterflies in registers,
// padded in-place exchanges through LDS.  Written for 64-wide wavefronts: every
// thread owns E = 16 (or 8) complex elements, a transform of length L uses L/E threads,
// and a pass of radix r makes each thread do E/r butterflies.
//
// Layouts
//   natural  : thread t, register i  <->  element  t + (L/16) * i          (coalesced in t)
//   spectral : what the last forward pass leaves in registers; spec_index(t, i)
//              gives the frequency bin.  Filter tables are pre-permuted on the
//              host into this layout, so forward -> pointwise multiply -> inverse
//              never pays for a digit-reversal.
//
// The forward transform is decimation-in-frequency, the inverse is the exact
// mirror (decimation-in-time with conjugated twiddles), so both start and end in
// the natural layout and use the same LDS addresses.  In-place LDS address of
// element (kappa, m) before pass p is kappa * L_p + m; pad(a) = a + (a >> 4)
// keeps every 32-lane group on distinct banks for all strides that occur.
//
// All per-thread steps are GPA_HD so that tests/host_fft_emulator.cpp can run the
// very same index arithmetic on the CPU, thread by thread, phase by phase.
#pragma once
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPA_HD __host__ __device__ __forceinline__
#else
#define GPA_HD inline
#endif

namespace gpa {

template <class T>
struct cpx {
  T x, y;
};

template <class T> GPA_HD cpx<T> operator+(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <class T> GPA_HD cpx<T> operator-(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <class T> GPA_HD cpx<T> cmul(cpx<T> a, cpx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
template <class T> GPA_HD cpx<T> cmulc(cpx<T> a, cpx<T> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
template <bool CONJ, class T> GPA_HD cpx<T> cmul_maybe_conj(cpx<T> a, cpx<T> b) {
  if constexpr (CONJ) return cmulc(a, b);
  else return cmul(a, b);
}
template <class T> GPA_HD cpx<T> cscale(cpx<T> a, T s) { return {a.x * s, a.y * s}; }

// ---------------------------------------------------------------------------
// small DFTs on register arrays, natural order in -> natural order out.
// INV = false: w = exp(-2 pi i / R); INV = true: conjugate.  Unnormalised.
// ---------------------------------------------------------------------------
template <bool INV, class T>
GPA_HD void dft2(cpx<T>& a, cpx<T>& b) {
  cpx<T> t = a;
  a = t + b;
  b = t - b;
}

template <bool INV, class T>
GPA_HD void dft4(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2, cpx<T>& a3) {
  cpx<T> s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
  a0 = s02 + s13;
  a2 = s02 - s13;
  // forward: X1 = d02 - i d13, X3 = d02 + i d13
  cpx<T> m = {d02.x + d13.y, d02.y - d13.x};
  cpx<T> p = {d02.x - d13.y, d02.y + d13.x};
  if constexpr (INV) { a1 = p; a3 = m; }
  else { a1 = m; a3 = p; }
}

// multiply by w16^M (forward) or its conjugate (inverse); M in [0,16)
template <int M, bool INV, class T>
GPA_HD cpx<T> mul_w16(cpx<T> a) {
  constexpr T c1 = T(0.92387953251128673848), s1 = T(0.38268343236508978178), rh = T(0.70710678118654752440);
  constexpr int m = M & 15;
  if constexpr (m == 0) return a;
  else if constexpr (m == 8) return {-a.x, -a.y};
  else if constexpr (m == 4) { if constexpr (INV) return {-a.y, a.x}; else return {a.y, -a.x}; }
  else if constexpr (m == 12) { if constexpr (INV) return {a.y, -a.x}; else return {-a.y, a.x}; }
  else {
    // w = (cr, -ci) forward, (cr, +ci) inverse with (cr, ci) = (cos, sin)(2 pi m / 16)
    constexpr T cr = (m == 1 || m == 15) ? c1 : (m == 2 || m == 14) ? rh : (m == 3 || m == 13) ? s1
                   : (m == 5 || m == 11) ? -s1 : (m == 6 || m == 10) ? -rh : -c1;  // m == 7, 9
    constexpr T ci = (m == 1 || m == 7) ? s1 : (m == 2 || m == 6) ? rh : (m == 3 || m == 5) ? c1
                   : (m == 9 || m == 15) ? -s1 : (m == 10 || m == 14) ? -rh : -c1;  // m == 11, 13
    constexpr T wi = INV ? ci : -ci;
    return {a.x * cr - a.y * wi, a.x * wi + a.y * cr};
  }
}

template <int R, bool INV, class T>
GPA_HD void dft_regs(cpx<T>* v) {
  if constexpr (R == 2) {
    dft2<INV>(v[0], v[1]);
  } else if constexpr (R == 4) {
    dft4<INV>(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 8) {
    // j = 4 j1 + j0, k = k0 + 2 k1
    dft2<INV>(v[0], v[4]);
    dft2<INV>(v[1], v[5]);
    dft2<INV>(v[2], v[6]);
    dft2<INV>(v[3], v[7]);
    v[5] = mul_w16<2, INV>(v[5]);
    v[6] = mul_w16<4, INV>(v[6]);
    v[7] = mul_w16<6, INV>(v[7]);
    dft4<INV>(v[0], v[1], v[2], v[3]);
    dft4<INV>(v[4], v[5], v[6], v[7]);
    // v[4 k0 + k1] holds X[k0 + 2 k1]
    cpx<T> t[8];
#pragma unroll
    for (int k0 = 0; k0 < 2; ++k0)
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) t[k0 + 2 * k1] = v[4 * k0 + k1];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = t[i];
  } else {
    static_assert(R == 16, "radix must be 2, 4, 8 or 16");
    // j = 4 j1 + j0, k = k0 + 4 k1
    dft4<INV>(v[0], v[4], v[8], v[12]);
    dft4<INV>(v[1], v[5], v[9], v[13]);
    dft4<INV>(v[2], v[6], v[10], v[14]);
    dft4<INV>(v[3], v[7], v[11], v[15]);
    // y[j0][k0] sits at v[j0 + 4 k0]; twiddle w16^(j0 k0)
    v[5] = mul_w16<1, INV>(v[5]);
    v[6] = mul_w16<2, INV>(v[6]);
    v[7] = mul_w16<3, INV>(v[7]);
    v[9] = mul_w16<2, INV>(v[9]);
    v[10] = mul_w16<4, INV>(v[10]);
    v[11] = mul_w16<6, INV>(v[11]);
    v[13] = mul_w16<3, INV>(v[13]);
    v[14] = mul_w16<6, INV>(v[14]);
    v[15] = mul_w16<9, INV>(v[15]);
    dft4<INV>(v[0], v[1], v[2], v[3]);
    dft4<INV>(v[4], v[5], v[6], v[7]);
    dft4<INV>(v[8], v[9], v[10], v[11]);
    dft4<INV>(v[12], v[13], v[14], v[15]);
    // v[4 k0 + k1] holds X[k0 + 4 k1]: transpose the 4x4
    cpx<T> t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
  }
}

// ---------------------------------------------------------------------------
// Workgroup FFT of length L = 2^LOG2L, 64 <= L <= 16384.
// ---------------------------------------------------------------------------
// EE = elements per thread: 16 (the sweep kernels and long transforms: fewest LDS exchanges), or 8 -- half the
// instructions per wavefront on twice the threads, which is what a SHORT transform wants: a 512-point kernel
// runs one wavefront per SIMD and is bound by that wavefront's instruction issue, not by the chip.
template <class T, int LOG2L, int EE = 16>
struct WgFFT {
  static_assert(LOG2L >= 6 && LOG2L <= 14, "supported lengths: 64 .. 16384");
  static_assert(EE == 16 || EE == 8, "elements per thread: 16 or 8");
  static constexpr int L = 1 << LOG2L;
  static constexpr int E = EE;
  static constexpr int LGE = EE == 16 ? 4 : 3;
  static constexpr int TPF = L / E;           // threads per transform
  static constexpr int P = (LOG2L + LGE - 1) / LGE;   // passes (radix <= E)
  static_assert(P <= 4, "at most four passes");
  static constexpr int LDS_ELEMS = L + L / 16;

  // bits of pass p (balanced split of LOG2L into P digits, larger digits first)
  static constexpr int bits(int p) { return LOG2L / P + (p < LOG2L % P ? 1 : 0); }
  // log2 of the sub-transform length entering pass p
  static constexpr int lg_len(int p) {
    int s = LOG2L;
    for (int q = 0; q < p; ++q) s -= bits(q);
    return s;
  }
  GPA_HD static int pad(int a) { return a + (a >> 4); }

  // in-place LDS address of butterfly element e of group G in pass p, split as
  // addr_base(G) + addr_offs(e) with a compile-time offset so that the 16 accesses of an
  // exchange share one address register and use the ds_* immediate offset field.
  // (kappa L_p + m0 + e S) padded: the low nibble of kappa L_p + m0 never carries into
  // the e S term because m0 < S and S | 16 or 16 | S.)
  template <int p>
  GPA_HD static int addr_base(int G) {
    constexpr int lgLp = lg_len(p), lgS = lgLp - bits(p);
    const int kappa = G >> lgS, m0 = G & ((1 << lgS) - 1);
    return pad((kappa << lgLp) + m0);
  }
  template <int p>
  static constexpr int addr_offs(int e) {
    constexpr int lgS = lg_len(p) - bits(p);
    return (e << lgS) + ((e << lgS) >> 4);
  }
  template <int p>
  GPA_HD static int addr(int G, int e) { return addr_base<p>(G) + addr_offs<p>(e); }

  // frequency bin held by thread tid, register i after the forward transform
  GPA_HD static int spec_index(int tid, int i) {
    constexpr int pl = P - 1;
    constexpr int g = E >> bits(pl);
    const int q = i % g, kl = i / g;
    int kappa = tid + TPF * q;   // digits (k_1 .. k_{P-1}), k_1 most significant
    int k = 0, mult = L >> bits(pl);
    // peel digits from least significant (k_{P-1}) to most (k_1)
    for (int p = P - 2; p >= 0; --p) {
      const int r = 1 << bits(p);
      mult >>= bits(p);
      k += (kappa & (r - 1)) * mult;
      kappa >>= bits(p);
    }
    return k + kl * (L >> bits(pl));
  }

  // Base twiddles kept in registers for the whole kernel (loop invariant across
  // the k-vector sweep): for twiddled pass p and butterfly group q,
  // lo = w^1, w^2, w^3 and hi = w^4, w^8, w^12 with w = exp(-2 pi i m0 / L_p).
  static constexpr int GMAX = 2;   // radix >= E / 2 in every twiddled pass
  struct Twiddles {
    cpx<T> lo[P > 1 ? P - 1 : 1][GMAX][3];
    cpx<T> hi[P > 1 ? P - 1 : 1][GMAX][3];
  };

  template <int p>
  GPA_HD static void load_twiddles_pass(Twiddles& tw, const cpx<T>* __restrict__ table, int tid) {
    constexpr int b = bits(p), r = 1 << b, g = E / r, lgLp = lg_len(p), lgS = lgLp - b;
    static_assert(g <= GMAX, "twiddled passes must have radix >= E / 2");
#pragma unroll
    for (int q = 0; q < g; ++q) {
      const int G = tid + TPF * q;
      const int m0 = G & ((1 << lgS) - 1);
      const int u = m0 << (LOG2L - lgLp);   // exponent unit in the length-L table
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if ((c + 1) < r) tw.lo[p][q][c] = table[u * (c + 1)];
        if (4 * (c + 1) < r) tw.hi[p][q][c] = table[u * 4 * (c + 1)];
      }
    }
  }
  GPA_HD static void load_twiddles(Twiddles& tw, const cpx<T>* __restrict__ table, int tid) {
    if constexpr (P > 1) load_twiddles_pass<0>(tw, table, tid);
    if constexpr (P > 2) load_twiddles_pass<1>(tw, table, tid);
    if constexpr (P > 3) load_twiddles_pass<2>(tw, table, tid);
  }

  // The same six base twiddles per pass, fetched from the table (L1 / L2 resident) where a butterfly needs them
  // instead of living in registers for the whole kernel: 12 complex values per twiddled pass are 48 VGPRs in
  // f64, which is what pushes f64 pass A over the 256-register limit (68 B of spills: 3.05 -> 2.2 ms without).
  // Measured worse in f64 pass B (5.9 -> 6.9 ms: the loads sit in its K loop) and in the f64 row DCT kernels, which
  // keep the register-resident form.  KTw = the per-precision choice of pass A.
  struct TwiddlesMem {
    const cpx<T>* table;
    int tid;
  };
  GPA_HD static void load_twiddles(TwiddlesMem& tw, const cpx<T>* __restrict__ table, int tid) {
    tw.table = table;
    tw.tid = tid;
  }
  using KTw = typename std::conditional<sizeof(T) == 8, TwiddlesMem, Twiddles>::type;
  // call once per iteration of a loop around transforms: keeps the compiler from hoisting the (loop invariant)
  // table loads of TwiddlesMem out of the loop and back into registers
  GPA_HD static void refresh(Twiddles&) {}
  GPA_HD static void refresh(TwiddlesMem& tw) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(tw.table));
#else
    (void)tw;
#endif
  }

  // v[k] *= w^k (forward) or conj(w^k) (inverse), k = 1 .. r-1
  template <int p, int r, bool INV>
  GPA_HD static void twiddle(cpx<T>* v, const Twiddles& tw, int q) {
#pragma unroll
    for (int k = 1; k < r; ++k) {
      const int a = k >> 2, b = k & 3;
      cpx<T> t = v[k];
      if (a > 0) t = cmul_maybe_conj<INV>(t, tw.hi[p][q][a - 1]);
      if (b > 0) t = cmul_maybe_conj<INV>(t, tw.lo[p][q][b - 1]);
      v[k] = t;
    }
  }
  template <int p, int r, bool INV>
  GPA_HD static void twiddle(cpx<T>* v, const TwiddlesMem& tw, int q) {
    constexpr int lgLp = lg_len(p), lgS = lgLp - bits(p);
    const int G = tw.tid + TPF * q;
    const int m0 = G & ((1 << lgS) - 1);
    const int u = m0 << (LOG2L - lgLp);   // exponent unit in the length-L table
    cpx<T> lo[3], hi[3];
#if defined(__HIP_DEVICE_COMPILE__)
    // (the table pointer went through refresh()'s opaque asm and lost its address space: read it as global memory,
    //  not through FLAT instructions)
    const __attribute__((address_space(1))) T* g = (const __attribute__((address_space(1))) T*)reinterpret_cast<const T*>(tw.table);
#define GPA_TW_AT(j) cpx<T>{g[2 * (j)], g[2 * (j) + 1]}
#else
#define GPA_TW_AT(j) tw.table[(j)]
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if ((c + 1) < r) lo[c] = GPA_TW_AT(u * (c + 1));
      if (4 * (c + 1) < r) hi[c] = GPA_TW_AT(u * 4 * (c + 1));
    }
#undef GPA_TW_AT
#pragma unroll
    for (int k = 1; k < r; ++k) {
      const int a = k >> 2, b = k & 3;
      cpx<T> t = v[k];
      if (a > 0) t = cmul_maybe_conj<INV>(t, hi[a - 1]);
      if (b > 0) t = cmul_maybe_conj<INV>(t, lo[b - 1]);
      v[k] = t;
    }
  }

  // Three-pass transforms with the base twiddles of pass 0 in registers and those of pass 1 in LDS: pass 1's
  // exponents depend on the thread only through m0 = G mod S_1 (16 distinct sets at 4096 points, 8 below), so ONE
  // small table serves the workgroup and the thread keeps 12 registers (f32; 24 in f64) fewer across the k-loop.
  // t1: [S_1][6] complex in LDS, filled by fill_pass1_table (every thread of the workgroup calls it, then a barrier
  // before the first transform).
  struct TwiddlesP1Lds {
    cpx<T> lo[GMAX][3], hi[GMAX][3];
    const cpx<T>* t1;
    int tid;
  };
  static constexpr int P1_SETS = P == 3 ? (1 << (lg_len(1) - bits(1))) : 1;
  GPA_HD static void fill_pass1_table(cpx<T>* t1, const cpx<T>* __restrict__ table, int thread, int nthreads) {
    static_assert(P == 3, "TwiddlesP1Lds: three-pass transforms only");
    constexpr int lgLp = lg_len(1);
    for (int e = thread; e < P1_SETS * 6; e += nthreads) {
      const int m0 = e / 6, c = e % 6;
      const int u = m0 << (LOG2L - lgLp);
      // (a radix-8 pass uses w^1..w^4 only: the other exponents would reach past the table)
      constexpr int r1 = 1 << bits(1);
      const int ex = c < 3 ? c + 1 : 4 * (c - 2);
      t1[e] = ex < r1 ? table[u * ex] : cpx<T>{T(1), T(0)};
    }
  }
  GPA_HD static void load_twiddles(TwiddlesP1Lds& tw, const cpx<T>* __restrict__ table, int tid, const cpx<T>* t1) {
    static_assert(P == 3, "TwiddlesP1Lds: three-pass transforms only");
    constexpr int b = bits(0), r = 1 << b, g = E / r, lgS = LOG2L - b;
#pragma unroll
    for (int q = 0; q < g; ++q) {
      const int G = tid + TPF * q;
      const int u = G & ((1 << lgS) - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if ((c + 1) < r) tw.lo[q][c] = table[u * (c + 1)];
        if (4 * (c + 1) < r) tw.hi[q][c] = table[u * 4 * (c + 1)];
      }
    }
    tw.t1 = t1;
    tw.tid = tid;
  }
  template <int p, int r, bool INV>
  GPA_HD static void twiddle(cpx<T>* v, const TwiddlesP1Lds& tw, int q) {
    cpx<T> lo[3], hi[3];
    if constexpr (p == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { lo[c] = tw.lo[q][c]; hi[c] = tw.hi[q][c]; }
    } else {
      constexpr int lgS = lg_len(p) - bits(p);
      const int m0 = (tw.tid + TPF * q) & ((1 << lgS) - 1);
      const cpx<T>* t = tw.t1 + m0 * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if ((c + 1) < r) lo[c] = t[c];
        if (4 * (c + 1) < r) hi[c] = t[3 + c];
      }
    }
#pragma unroll
    for (int k = 1; k < r; ++k) {
      const int a = k >> 2, b = k & 3;
      cpx<T> t = v[k];
      if (a > 0) t = cmul_maybe_conj<INV>(t, hi[a - 1]);
      if (b > 0) t = cmul_maybe_conj<INV>(t, lo[b - 1]);
      v[k] = t;
    }
  }
  GPA_HD static void refresh(TwiddlesP1Lds&) {}

  // The general form of TwiddlesP1Lds for three- and four-pass transforms: pass 0 in registers, EVERY later twiddled pass from
  // a small LDS table (pass p: [S_p][6] complex, S_p = distinct m0 of the pass; 8192 points: 64 + 8 sets = 3.4 KB).  What a
  // kernel at its register limit wants: the persistent 16384-point row kernels keep two phase tables in registers instead.
  struct TwiddlesLds {
    cpx<T> lo[GMAX][3], hi[GMAX][3];
    const cpx<T>* t;   // tables of passes 1 .. P-2, one after the other
    int tid;
  };
  static constexpr int pass_sets(int p) { return 1 << (lg_len(p) - bits(p)); }
  static constexpr int lds_table_offset(int p) {   // in complex elements, p >= 1
    int o = 0;
    for (int q = 1; q < p; ++q) o += pass_sets(q) * 6;
    return o;
  }
  static constexpr int LDS_TABLE_ELEMS = P > 2 ? lds_table_offset(P - 1) : 1;
  GPA_HD static void fill_lds_tables(cpx<T>* t, const cpx<T>* __restrict__ table, int thread, int nthreads) {
    static_assert(P >= 3, "TwiddlesLds: three- and four-pass transforms");
    fill_one_table<1>(t, table, thread, nthreads);
    if constexpr (P > 3) fill_one_table<2>(t, table, thread, nthreads);
  }
  template <int p>
  GPA_HD static void fill_one_table(cpx<T>* t, const cpx<T>* __restrict__ table, int thread, int nthreads) {
    constexpr int lgLp = lg_len(p), r = 1 << bits(p);
    cpx<T>* tp = t + lds_table_offset(p);
    for (int e = thread; e < pass_sets(p) * 6; e += nthreads) {
      const int m0 = e / 6, c = e % 6;
      const int u = m0 << (LOG2L - lgLp);
      const int ex = c < 3 ? c + 1 : 4 * (c - 2);
      tp[e] = ex < r ? table[u * ex] : cpx<T>{T(1), T(0)};
    }
  }
  GPA_HD static void load_twiddles(TwiddlesLds& tw, const cpx<T>* __restrict__ table, int tid, const cpx<T>* t) {
    constexpr int b = bits(0), r = 1 << b, g = E / r, lgS = LOG2L - b;
#pragma unroll
    for (int q = 0; q < g; ++q) {
      const int G = tid + TPF * q;
      const int u = G & ((1 << lgS) - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if ((c + 1) < r) tw.lo[q][c] = table[u * (c + 1)];
        if (4 * (c + 1) < r) tw.hi[q][c] = table[u * 4 * (c + 1)];
      }
    }
    tw.t = t;
    tw.tid = tid;
  }
  template <int p, int r, bool INV>
  GPA_HD static void twiddle(cpx<T>* v, const TwiddlesLds& tw, int q) {
    cpx<T> lo[3], hi[3];
    if constexpr (p == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { lo[c] = tw.lo[q][c]; hi[c] = tw.hi[q][c]; }
    } else {
      constexpr int lgS = lg_len(p) - bits(p);
      const int m0 = (tw.tid + TPF * q) & ((1 << lgS) - 1);
      const cpx<T>* t = tw.t + lds_table_offset(p) + m0 * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if ((c + 1) < r) lo[c] = t[c];
        if (4 * (c + 1) < r) hi[c] = t[3 + c];
      }
    }
#pragma unroll
    for (int k = 1; k < r; ++k) {
      const int a = k >> 2, b = k & 3;
      cpx<T> t = v[k];
      if (a > 0) t = cmul_maybe_conj<INV>(t, hi[a - 1]);
      if (b > 0) t = cmul_maybe_conj<INV>(t, lo[b - 1]);
      v[k] = t;
    }
  }
  GPA_HD static void refresh(TwiddlesLds&) {}

  // one pass of butterflies on the thread's 16 registers (q and j loops are
  // fully unrolled, so every register index is static)
  template <int p, bool INV, class TW>
  GPA_HD static void butterflies(cpx<T> (&x)[E], const TW& tw) {
    constexpr int r = 1 << bits(p), g = E / r;
    constexpr bool TWIDDLED = p < P - 1;
#pragma unroll
    for (int q = 0; q < g; ++q) {
      cpx<T> v[r];
#pragma unroll
      for (int j = 0; j < r; ++j) v[j] = x[q + g * j];
      if constexpr (INV && TWIDDLED) twiddle<p, r, true>(v, tw, q);
      dft_regs<r, INV>(v);
      if constexpr (!INV && TWIDDLED) twiddle<p, r, false>(v, tw, q);
#pragma unroll
      for (int j = 0; j < r; ++j) x[q + g * j] = v[j];
    }
  }

  // LS = element stride of this transform's LDS image: LS transforms of neighbouring
  // lanes are interleaved element by element (address = LS * padded index + lane slot),
  // which keeps every lane group on distinct banks when LS columns sit side by side
  // in the thread index.
  template <int p, int LS = 1>
  GPA_HD static void lds_write(const cpx<T> (&x)[E], cpx<T>* lds, int tid) {
    constexpr int r = 1 << bits(p), g = E / r;
#pragma unroll
    for (int q = 0; q < g; ++q) {
      cpx<T>* base = lds + LS * addr_base<p>(tid + TPF * q);
#pragma unroll
      for (int e = 0; e < r; ++e) base[LS * addr_offs<p>(e)] = x[q + g * e];
    }
  }
  template <int p, int LS = 1>
  GPA_HD static void lds_read(cpx<T> (&x)[E], const cpx<T>* lds, int tid) {
    constexpr int r = 1 << bits(p), g = E / r;
#pragma unroll
    for (int q = 0; q < g; ++q) {
      const cpx<T>* base = lds + LS * addr_base<p>(tid + TPF * q);
#pragma unroll
      for (int e = 0; e < r; ++e) x[q + g * e] = base[LS * addr_offs<p>(e)];
    }
  }

  // ---- phases (the code between two workgroup barriers) -------------------
  // forward: phase 0 .. P-1, barrier after every phase but the last
  template <int ph, int LS = 1, class TW = Twiddles>
  GPA_HD static void fwd_phase(cpx<T> (&x)[E], cpx<T>* lds, int tid, const TW& tw) {
    if constexpr (ph > 0) lds_read<ph, LS>(x, lds, tid);
    butterflies<ph, false>(x, tw);
    if constexpr (ph < P - 1) lds_write<ph, LS>(x, lds, tid);
  }
  // inverse: phase 0 handles pass P-1, phase P-1 handles pass 0
  template <int ph, int LS = 1, class TW = Twiddles>
  GPA_HD static void inv_phase(cpx<T> (&x)[E], cpx<T>* lds, int tid, const TW& tw) {
    constexpr int p = P - 1 - ph;
    if constexpr (ph > 0) lds_read<p, LS>(x, lds, tid);
    butterflies<p, true>(x, tw);
    if constexpr (p > 0) lds_write<p, LS>(x, lds, tid);
  }

#if defined(__HIPCC__)
  // whole transforms with workgroup barriers (every thread of the workgroup must call).
  // NT independent transforms per thread share each barrier: x[n] uses lds + n * lds_stride.
  template <int NT, int LS = 1, class TW = Twiddles>
  __device__ __forceinline__ static void forward_multi(cpx<T> (&x)[NT][E], cpx<T>* lds, int lds_stride, int tid,
                                                       const TW& tw) {
#pragma unroll
    for (int n = 0; n < NT; ++n) { fwd_phase<0, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    if constexpr (P > 1) {
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) { fwd_phase<1, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    }
    if constexpr (P > 2) {
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) { fwd_phase<2, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    }
    if constexpr (P > 3) {
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) { fwd_phase<3, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    }
  }
  template <int NT, int LS = 1, class TW = Twiddles>
  __device__ __forceinline__ static void inverse_multi(cpx<T> (&x)[NT][E], cpx<T>* lds, int lds_stride, int tid,
                                                       const TW& tw) {
#pragma unroll
    for (int n = 0; n < NT; ++n) { inv_phase<0, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    if constexpr (P > 1) {
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) { inv_phase<1, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    }
    if constexpr (P > 2) {
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) { inv_phase<2, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    }
    if constexpr (P > 3) {
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) { inv_phase<3, LS>(x[n], lds + n * lds_stride, tid, tw); if (NT > 1) __builtin_amdgcn_sched_barrier(0); }
    }
  }
  template <class TW>
  __device__ __forceinline__ static void forward(cpx<T> (&x)[E], cpx<T>* lds, int tid, const TW& tw) {
    forward_multi<1>(*reinterpret_cast<cpx<T>(*)[1][E]>(&x), lds, 0, tid, tw);
  }
  template <class TW>
  __device__ __forceinline__ static void inverse(cpx<T> (&x)[E], cpx<T>* lds, int tid, const TW& tw) {
    inverse_multi<1>(*reinterpret_cast<cpx<T>(*)[1][E]>(&x), lds, 0, tid, tw);
  }
#endif
};

}  // namespace gpa
