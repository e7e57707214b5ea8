// DCT-II / DCT-III building blocks on top of WgFFT (power-of-two lengths).
//
// Convention = SciPy's dctn/idctn defaults used by the reference
// (phase_unwrap.py:84-103): forward X[k] = 2 sum_n x[n] cos(pi k (2n+1) / (2N)),
// idct = exact inverse.  Makhoul's algorithm maps an N-point DCT-II onto an N-point
// complex FFT of the even/odd-permuted input v:
//     v[m] = x[2m] (m < N/2),  v[m] = x[2(N-1-m)+1] (m >= N/2)
//     X[k] = w_k V[k] + conj(w_k) V[N-k],   w_k = exp(-i pi k / (2N))
// The relation is complex-linear, so TWO real sequences ride through one complex
// transform as real and imaginary parts.
//
// Three fused forms are provided; each is a list of phases separated by workgroup
// barriers, written GPA_HD so tests/host/dct_emulator.cpp can run them on the CPU.
#pragma once
#include "gpa_fft.h"

namespace gpa {

// 1/x: the f32 device path uses the hardware reciprocal (1 ulp, one instruction) instead of
// the ~10-instruction IEEE division; it only scales the CG preconditioner, whose exactness
// does not enter the fixed point of the iteration.
template <class T> GPA_HD T fast_recip(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof(T) == 4) return __builtin_amdgcn_rcpf(x);
#endif
  return T(1) / x;
}

// slot of the permuted sequence -> index in the original sequence
// (valid for odd N too: the first ceil(N/2) slots take the even samples)
GPA_HD int makhoul_src(int m, int N) { return m < (N + 1) / 2 ? 2 * m : 2 * (N - 1 - m) + 1; }

template <class T, int LG, int EE = 16>
struct WgDCT {
  using F = WgFFT<T, LG, EE>;
  static constexpr int N = F::L, TPF = F::TPF, E = EE;

  // ---- forward DCT-II of a packed pair -------------------------------------
  // in : x = permuted input in the natural layout (slot m = tid + TPF*i)
  // out: x[i] = Xa[k] + i Xb[k] at k = tid + TPF*i
  // sequence: F::forward; barrier; fwd_scatter; barrier; fwd_gather
  // LS: element stride of the LDS image (see WgFFT::lds_write)
  template <int LS = 1>
  GPA_HD static void fwd_scatter(const cpx<T> (&x)[E], cpx<T>* lds, int tid) {
#pragma unroll
    for (int i = 0; i < E; ++i) lds[LS * F::pad(F::spec_index(tid, i))] = x[i];
  }
  // wk: table w_k in natural order
  GPA_HD static void fwd_gather(cpx<T> (&x)[E], const cpx<T>* lds, int tid, const cpx<T>* __restrict__ wk) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tid + TPF * i;
      const cpx<T> zk = lds[F::pad(k)], zm = lds[F::pad((N - k) & (N - 1))];
      const cpx<T> w = wk[k];
      x[i] = cmul(w, zk) + cmulc(zm, w);
    }
  }

  // the same with w_k of the thread's bins already in registers (requested before the transform)
  GPA_HD static void fwd_gather(cpx<T> (&x)[E], const cpx<T>* lds, int tid, const cpx<T> (&wkv)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tid + TPF * i;
      const cpx<T> zk = lds[F::pad(k)], zm = lds[F::pad((N - k) & (N - 1))];
      x[i] = cmul(wkv[i], zk) + cmulc(zm, wkv[i]);
    }
  }

  // ---- fused DCT-II -> per-bin scale -> DCT-III of a packed pair -------------
  // After F::forward the thread holds Z[k], k = spec_index(tid, i).
  // sequence: F::forward; barrier; solve_scatter; barrier; solve_combine; F::inverse
  // Tables in the spectral layout [i][tid]: wspec = w_k, ha = 1 - cos(.) of bin k,
  // ham = the same of bin N-k.  hb_a / hb_b: 1 - cos(.) of the two packed sequences'
  // own (other-axis) bins; inv_n = 1/N.  The divisor of bin (k, other) is
  // 2 (cos + cos - 2) = -2 (ha + hb), replaced by 1 where k == 0 and first_a /
  // first_b says the other-axis bin is 0 too (phase_unwrap.py:106-115).  The tables
  // hold 1 - cos = 2 sin^2(half angle), evaluated in double on the host, because
  // cos + cos - 2 cancels catastrophically in f32 near the DC corner.
  template <int LS = 1>
  GPA_HD static void solve_scatter(const cpx<T> (&x)[E], cpx<T>* lds, int tid) { fwd_scatter<LS>(x, lds, tid); }
  // the thread's table entries in registers: requested before the forward transform by kernels that are bound by
  // their chain of memory round trips (short transforms), see solve_combine below
  struct SolveTables { cpx<T> w[E]; T h[E], hm[E]; };
  GPA_HD static void load_solve_tables(SolveTables& t, int tid, const cpx<T>* __restrict__ wspec,
                                       const T* __restrict__ ha, const T* __restrict__ ham) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      t.w[i] = wspec[i * TPF + tid];
      t.h[i] = ha[i * TPF + tid];
      t.hm[i] = ham[i * TPF + tid];
    }
  }
  template <int LS = 1>
  GPA_HD static void solve_combine(cpx<T> (&x)[E], const cpx<T>* lds, int tid, const SolveTables& tb, T hb_a, T hb_b,
                                   bool first_a, bool first_b, T inv_n, double* rho = nullptr) {
    solve_combine_impl<LS, true>(x, lds, tid, &tb, nullptr, nullptr, nullptr, hb_a, hb_b, first_a, first_b, inv_n, rho);
  }
  template <int LS = 1>
  GPA_HD static void solve_combine(cpx<T> (&x)[E], const cpx<T>* lds, int tid,
                                   const cpx<T>* __restrict__ wspec, const T* __restrict__ ha,
                                   const T* __restrict__ ham, T hb_a, T hb_b, bool first_a,
                                   bool first_b, T inv_n, double* rho = nullptr) {
    solve_combine_impl<LS, false>(x, lds, tid, nullptr, wspec, ha, ham, hb_a, hb_b, first_a, first_b, inv_n, rho);
  }
  template <int LS, bool REGS>
  GPA_HD static void solve_combine_impl(cpx<T> (&x)[E], const cpx<T>* lds, int tid, const SolveTables* tb,
                                        const cpx<T>* __restrict__ wspec, const T* __restrict__ ha,
                                        const T* __restrict__ ham, T hb_a, T hb_b, bool first_a,
                                        bool first_b, T inv_n, double* rho) {
    // rho (optional): this thread's share of <input, output> of the solve, summed over both packed sequences.
    // The forward / inverse transforms are unnormalised with 1/N folded into the scale, so by the DFT's
    // Parseval relation  sum_n (a za + b zb) = Re sum_k Z_in[k] conj(Z_out[k])  on the PACKED spectra -- two
    // FMAs per bin on values that are in registers anyway (the cross terms between the two sequences cancel
    // between bins k and N-k).  In SciPy's DCT normalisation <r,z> = sum_j c_j / (2 N_other) * (that sum of
    // column j), c_0 = 1/2: the caller applies 1 / N_other, the factor 1/2 is applied here, and the one
    // column with c_j = 1/2 (first_a) is corrected by half of its own share, which for that single column is
    // evaluated from the unpacked spectrum: sum_k c_k X_k^2 / lambda_k with X_k = 2 Re(w_k V_k).
    T packed = T(0), corr = T(0);
    if (rho && first_a) {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int k = F::spec_index(tid, i);
        const cpx<T> zk = x[i], zm = lds[LS * F::pad((N - k) & (N - 1))];
        cpx<T> w;
        if constexpr (REGS) w = tb->w[i]; else w = wspec[i * TPF + tid];
        const cpx<T> va = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};
        const T uax = w.x * va.x - w.y * va.y;
        T h0;
        if constexpr (REGS) h0 = tb->h[i]; else h0 = ha[i * TPF + tid];
        T sa = T(-0.5) * inv_n * fast_recip(h0 + hb_a);
        if (k == 0) sa = inv_n;
        corr += (k == 0 ? T(0.5) : T(1)) * uax * uax * sa;
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = F::spec_index(tid, i);
      const cpx<T> zk = x[i], zm = lds[LS * F::pad((N - k) & (N - 1))];
      cpx<T> w;
      T h, hm;
      if constexpr (REGS) { w = tb->w[i]; h = tb->h[i]; hm = tb->hm[i]; }
      else { w = wspec[i * TPF + tid]; h = ha[i * TPF + tid]; hm = ham[i * TPF + tid]; }
      // split the packed transform: Va = (Zk + conj Zm)/2, Vb = (Zk - conj Zm)/(2i)
      const cpx<T> va = {T(0.5) * (zk.x + zm.x), T(0.5) * (zk.y - zm.y)};
      const cpx<T> vb = {T(0.5) * (zk.y + zm.y), T(-0.5) * (zk.x - zm.x)};
      const cpx<T> ua = cmul(w, va), ub = cmul(w, vb);
      const T cn = T(-0.5) * inv_n;   // 1 / (2 (cos + cos - 2)) / N = cn / (ha + hb)
      T sa = cn * fast_recip(h + hb_a), sb = cn * fast_recip(h + hb_b);
      T sam = cn * fast_recip(hm + hb_a), sbm = cn * fast_recip(hm + hb_b);
      if (k == 0) {
        sam = T(0);
        sbm = T(0);
        if (first_a) sa = inv_n;
        if (first_b) sb = inv_n;
      }
      const cpx<T> ya = cmulc(cpx<T>{sa * ua.x, sam * ua.y}, w);
      const cpx<T> yb = cmulc(cpx<T>{sb * ub.x, sbm * ub.y}, w);
      const cpx<T> xn = {ya.x - yb.y, ya.y + yb.x};
      x[i] = xn;
      if (rho) packed = fma(zk.x, xn.x, fma(zk.y, xn.y, packed));
    }
    if (rho) *rho += 0.5 * ((double)packed - (double)corr);
  }

  // ---- inverse (DCT-III) of a packed pair ------------------------------------
  // in : xk[i] = Xa[k] + i Xb[k], xm[i] = same at N-k (0 for k == 0), k = tid + TPF*i
  // out: x[i] = (a[c], b[c]) / N at c = tid + TPF*i  (original sample order)
  // sequence: inv_prepare; F::forward; barrier; inv_scatter; barrier; inv_gather
  GPA_HD static void inv_prepare(cpx<T> (&x)[E], const cpx<T> (&xm)[E], int tid,
                                 const cpx<T>* __restrict__ wk) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = tid + TPF * i;
      const cpx<T> d = {x[i].x + xm[i].y, x[i].y - xm[i].x};   // X_k - i X_{N-k}
      const cpx<T> v = cmulc(d, wk[k]);                        // * conj(w_k)
      x[i] = {T(0.5) * v.x, T(-0.5) * v.y};                    // conj(V_k): IFFT = conj(FFT(conj .))
    }
  }
  GPA_HD static void inv_prepare(cpx<T> (&x)[E], const cpx<T> (&xm)[E], const cpx<T> (&wkv)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const cpx<T> d = {x[i].x + xm[i].y, x[i].y - xm[i].x};   // X_k - i X_{N-k}
      const cpx<T> v = cmulc(d, wkv[i]);                       // * conj(w_k)
      x[i] = {T(0.5) * v.x, T(-0.5) * v.y};
    }
  }
  GPA_HD static void inv_scatter(const cpx<T> (&x)[E], cpx<T>* lds, int tid, T inv_n) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int m = F::spec_index(tid, i);
      lds[F::pad(makhoul_src(m, N))] = {x[i].x * inv_n, -x[i].y * inv_n};
    }
  }
  GPA_HD static void inv_gather(cpx<T> (&x)[E], const cpx<T>* lds, int tid) {
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = lds[F::pad(tid + TPF * i)];
  }
};


// ---------------------------------------------------------------------------
// DFT of ARBITRARY length n on the power-of-two workgroup FFT (Bluestein / chirp-z):
//   X[k] = conj(c_k) * sum_m (x[m] conj(c_m)) c_{k-m},   c_m = exp(i pi m^2 / n)
// i.e. chirp multiply -> circular convolution of length L >= 2n-1 -> chirp multiply.
// The convolution is forward FFT -> table multiply -> inverse FFT, exactly the shape of
// the lock-in filters, so it never leaves registers + LDS.  Tables (built in double on
// the host): chirp[m] = c_m (m < n), bspec = FFT_L(b)/L in the spectral layout with
// b[m] = b[L-m] = c_m.
// ---------------------------------------------------------------------------
template <class T, int LG>
struct WgBluestein {
  using F = WgFFT<T, LG>;
  static constexpr int L = F::L, TPF = F::TPF, E = 16;

#if defined(__HIPCC__)
  // in : x[i] = sample at slot tid + TPF*i (must be 0 for slots >= n), natural layout
  // out: x[i] = X[k] at k = slot (< n); garbage for slots >= n
  // every thread of the workgroup must call (contains barriers)
  template <class TW>
  __device__ __forceinline__ static void dft(cpx<T> (&x)[E], cpx<T>* lds, int tid, int n,
                                             const cpx<T>* __restrict__ chirp, const cpx<T>* __restrict__ bspec,
                                             const TW& tw) {
    cpx<T> ch[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int slot = tid + TPF * i;
      ch[i] = slot < n ? chirp[slot] : cpx<T>{T(0), T(0)};
      x[i] = cmulc(x[i], ch[i]);
    }
    F::forward(x, lds, tid, tw);
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = cmul(x[i], bspec[i * TPF + tid]);
    F::inverse(x, lds, tid, tw);
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = cmulc(x[i], ch[i]);
  }
#endif
};

// workgroup geometry of the natural-layout (Bluestein) kernels: the generic-size DCT kernels of gpa_unwrap.hip and the
// plain DFTs of gpa_dft2.hip
template <class T, int LG>
struct GenGeom {
  using F = WgFFT<T, LG>;
  static constexpr int NF = F::TPF >= 256 ? 1 : 256 / F::TPF;   // transforms per workgroup
  static constexpr int RS = F::LDS_ELEMS + (NF > 1 ? (F::TPF < 32 ? F::TPF : 0) : 0);
  static constexpr int THREADS = NF * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NF * RS * sizeof(cpx<T>);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024;
};

// every transform length of the 16-element engine: 64 ... 16384
#define GPA_FOR_LG(X) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)

}  // namespace gpa
