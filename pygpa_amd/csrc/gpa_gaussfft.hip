// f-3: scipy.ndimage.gaussian_filter along one axis as an FFT convolution (geometric_phase_analysis.py:431-433 smooths the
// spectrum with sigma and, for the difference of Gaussians, with sigma = 50: a 401-tap kernel).
//
// SciPy's filter is a correlation with the normalised, truncated kernel w[-R .. R] (R = int(4 sigma + 0.5)) on the signal
// extended by half-sample-symmetric reflection ('reflect', period 2n).  A direct sum costs 2R + 1 multiply-adds per sample
// (13.5 G for the two passes of a 4096^2 image at sigma = 50: 3.3 ms per pass in the round-5 kernel).  Here the axis is cut
// into segments of S = L - 2R outputs; a segment loads the L samples [seg S - R, seg S - R + L) of the REFLECTED signal,
// runs  forward FFT_L -> multiply by the kernel's transfer function -> inverse FFT_L  in registers + LDS (the shape of the
// lock-in filters) and keeps the S outputs that the circular wrap does not touch (overlap-save).  The kernel is real and even,
// so its transfer function H is real: two real lines packed as a + i b are filtered by ONE complex transform pair and come
// out as the real and imaginary part.  Rows: two adjacent rows per transform; columns: tiles of adjacent columns, a pair of
// columns per transform (ColGeom of gpa_dft2.hip: 4 real columns per thread in f32 = one 16-byte access per row).
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "gpa_dct.h"
#include "gpa_gaussfft.h"
#include "gpa_internal.h"

namespace gpa {
namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // scipy 'reflect': (d c b a | a b c d | d c b a), period 2n
  const int p = 2 * n;
  int m = i % p;
  if (m < 0) m += p;
  return m < n ? m : p - 1 - m;
}

template <class T, int LG>
__global__ __launch_bounds__((GenGeom<T, LG>::THREADS)) void gf_rows_kernel(const T* __restrict__ in, T* out, int n0,
                                                                          int n1, int R, int nseg, const T* __restrict__ H,
                                                                          const cpx<T>* __restrict__ twtab,
                                                                          const T* minuend) {
  using F = WgFFT<T, LG>;
  using G = GenGeom<T, LG>;
  constexpr int TPF = F::TPF, L = F::L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x % TPF, f = threadIdx.x / TPF;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + f * G::RS;
  const int g = blockIdx.x * G::NF + f;
  const int pair = g / nseg, seg = g - pair * nseg;
  const int ra = 2 * pair, rb = ra + 1;
  const bool va = ra < n0, vb = rb < n0;
  const int S = L - 2 * R, base = seg * S - R;
  const T* pa = in + (size_t)(va ? ra : 0) * n1;
  const T* pb = in + (size_t)(vb ? rb : 0) * n1;
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, tid);
  cpx<T> x[16];
  // interior segments need no reflection: one test per thread instead of 16 remainders
  const bool inside = base >= 0 && base + L <= n1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i;
    const int j = inside ? base + slot : reflect_idx(base + slot, n1);
    x[i] = {va ? pa[j] : T(0), vb ? pb[j] : T(0)};
  }
  F::forward(x, lds, tid, tw);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const T h = H[i * TPF + tid];
    x[i] = {x[i].x * h, x[i].y * h};
  }
  F::inverse(x, lds, tid, tw);
  if (!va) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = tid + TPF * i, n = base + slot;
    if (slot >= R && slot < L - R && n < n1) {
      const size_t oa = (size_t)ra * n1 + n, ob = (size_t)rb * n1 + n;
      out[oa] = minuend ? minuend[oa] - x[i].x : x[i].x;
      if (vb) out[ob] = minuend ? minuend[ob] - x[i].y : x[i].y;
    }
  }
}

// columns: C complex transforms = 2C adjacent real columns per workgroup
template <class T, int LG>
struct GfColGeom {
  using F = WgFFT<T, LG>;
  static constexpr int cols() {
    int c = 16;
    while (c > 1 && (c * F::TPF > 1024 || (size_t)c * F::LDS_ELEMS * sizeof(cpx<T>) > 140 * 1024)) c /= 2;
    return c;
  }
  static constexpr int C = cols();
  static constexpr int NT = (sizeof(T) == 4 && C >= 2) ? 2 : 1;   // complex transforms per thread
  static constexpr int CT = C / NT;
  static constexpr int REGION = CT * F::LDS_ELEMS;
  static constexpr int THREADS = CT * F::TPF;
  static constexpr size_t LDS_BYTES = (size_t)NT * REGION * sizeof(cpx<T>);
  static constexpr bool FITS = (size_t)F::LDS_ELEMS * sizeof(cpx<T>) <= 140 * 1024;
  static constexpr int RC = 2 * C;   // real columns per workgroup
};

template <class T, int LG>
__global__ __launch_bounds__((GfColGeom<T, LG>::THREADS)) void gf_cols_kernel(const T* __restrict__ in, T* __restrict__ out, int n0,
                                                                            int n1, int R, const T* __restrict__ H,
                                                                            const cpx<T>* __restrict__ twtab) {
  using F = WgFFT<T, LG>;
  using G = GfColGeom<T, LG>;
  constexpr int CT = G::CT, NT = G::NT, TPF = F::TPF, L = F::L, W = 2 * NT;   // W real columns per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = threadIdx.x % CT, t = threadIdx.x / CT;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + c;
  const int tile = xcd_tile(blockIdx.x, gridDim.x), seg = blockIdx.y;
  const int y0 = tile * G::RC + c * W;
  const int S = L - 2 * R, base = seg * S - R;
  struct alignas(W * sizeof(T)) Vec { T v[W]; };
  const bool vec = y0 + W <= n1 && (n1 % W) == 0 && (reinterpret_cast<size_t>(in) & (W * sizeof(T) - 1)) == 0 &&
                   (reinterpret_cast<size_t>(out) & (W * sizeof(T) - 1)) == 0;
  cpx<T> x[NT][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = reflect_idx(base + t + TPF * i, n0);
    Vec v;
    if (vec) v = *reinterpret_cast<const Vec*>(in + (size_t)row * n1 + y0);
    else {
#pragma unroll
      for (int k = 0; k < W; ++k) v.v[k] = y0 + k < n1 ? in[(size_t)row * n1 + y0 + k] : T(0);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) x[n][i] = {v.v[2 * n], v.v[2 * n + 1]};
  }
  typename F::Twiddles tw;
  F::load_twiddles(tw, twtab, t);
  F::template forward_multi<NT, CT>(x, lds, G::REGION, t, tw);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const T h = H[i * TPF + t];
#pragma unroll
    for (int n = 0; n < NT; ++n) x[n][i] = {x[n][i].x * h, x[n][i].y * h};
  }
  F::template inverse_multi<NT, CT>(x, lds, G::REGION, t, tw);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int slot = t + TPF * i, row = base + slot;
    if (slot >= R && slot < L - R && row < n0) {
      Vec v;
#pragma unroll
      for (int n = 0; n < NT; ++n) { v.v[2 * n] = x[n][i].x; v.v[2 * n + 1] = x[n][i].y; }
      if (vec) *reinterpret_cast<Vec*>(out + (size_t)row * n1 + y0) = v;
      else {
#pragma unroll
        for (int k = 0; k < W; ++k)
          if (y0 + k < n1) out[(size_t)row * n1 + y0 + k] = v.v[k];
      }
    }
  }
}

template <class T, int LG>
hipError_t run_rows(const T* in, T* out, int n0, int n1, int R, const void* H, const void* tw, const T* minuend, hipStream_t s) {
  using G = GenGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = gf_rows_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int S = (1 << LG) - 2 * R, nseg = (n1 + S - 1) / S;
    const size_t ntr = (size_t)((n0 + 1) / 2) * nseg;
    GPA_PROF("gauss_fft_rows_kernel", s);
    kern<<<(unsigned)((ntr + G::NF - 1) / G::NF), G::THREADS, G::LDS_BYTES, s>>>(in, out, n0, n1, R, nseg, (const T*)H,
                                                                                 (const cpx<T>*)tw, minuend);
    return hipGetLastError();
  }
}
template <class T, int LG>
hipError_t run_cols(const T* in, T* out, int n0, int n1, int R, const void* H, const void* tw, hipStream_t s) {
  using G = GfColGeom<T, LG>;
  if constexpr (!G::FITS) return hipErrorInvalidValue;
  else {
    auto kern = gf_cols_kernel<T, LG>;
    static unsigned lds_set = 0;
    hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, lds_set);
    if (e != hipSuccess) return e;
    const int S = (1 << LG) - 2 * R, nseg = (n0 + S - 1) / S;
    GPA_PROF("gauss_fft_cols_kernel", s);
    kern<<<dim3((n1 + G::RC - 1) / G::RC, nseg), G::THREADS, G::LDS_BYTES, s>>>(in, out, n0, n1, R, (const T*)H, (const cpx<T>*)tw);
    return hipGetLastError();
  }
}

int real_cols_per_tile(int dtype, int lg) {
#define CASE(LG) case LG: return dtype == 0 ? GfColGeom<float, LG>::RC : GfColGeom<double, LG>::RC;
  switch (lg) { GPA_FOR_LG(CASE) }
#undef CASE
  return 1;
}

}  // namespace

// transform length of the overlap-save segments: the cheapest of the lengths with L >= 4R (cost = segments x L x passes,
// columns: x 64 bytes / row piece where a tile's row piece is narrower than that)
int gaussfft_choose_lg(int dtype, int n, int R, bool cols) {
  const int maxlg = dtype == 0 ? 14 : 13;
  int lgmin = 6;
  while ((1 << lgmin) < 4 * R) ++lgmin;
  if (lgmin > maxlg) return 0;
  double best = 0;
  int best_lg = 0;
  for (int lg = lgmin; lg <= maxlg; ++lg) {
    const int L = 1 << lg, S = L - 2 * R, nseg = (n + S - 1) / S, passes = (lg + 3) / 4;
    double cost = (double)nseg * L * passes;
    if (cols) {
      const int piece = real_cols_per_tile(dtype, lg) * (dtype == 0 ? 4 : 8);
      if (piece < 64) cost *= 64.0 / piece;
    }
    if (!best_lg || cost < best) { best = cost; best_lg = lg; }
  }
  return best_lg;
}

// H[k] = (w_0 + 2 sum_j w_j cos(2 pi k j / L)) / L in the register engine's spectral layout [register][thread], doubles
std::vector<double> gaussfft_table(int lg, const std::vector<double>& w /* 2R + 1 taps */) {
  const int L = 1 << lg, R = (int)(w.size() / 2), tpf = L / 16;
  std::vector<long double> cs((size_t)L);
  const long double u = 2.0L * 3.14159265358979323846264338327950288L / (long double)L;
  for (int j = 0; j < L; ++j) cs[j] = cosl(u * (long double)j);
  std::vector<double> nat((size_t)L), out((size_t)L);
  for (int k = 0; k < L; ++k) {
    long double acc = 0;
    for (int j = R; j >= 1; --j) acc += 2.0L * (long double)w[R + j] * cs[(size_t)(((long long)k * j) % L)];
    acc += (long double)w[R];
    nat[k] = (double)(acc / (long double)L);
  }
  for (int i = 0; i < 16; ++i)
    for (int t = 0; t < tpf; ++t) out[(size_t)i * tpf + t] = nat[spec_index_rt(lg, t, i)];
  return out;
}

hipError_t launch_gaussfft(int dtype, int lg, int axis, const void* in, void* out, int n0, int n1, int R, const void* H,
                           const void* tw, const void* minuend, hipStream_t s) {
  if (axis == 1) {
#define CASE(LG) case LG: return dtype == 0 ? run_rows<float, LG>((const float*)in, (float*)out, n0, n1, R, H, tw, (const float*)minuend, s) \
                                            : run_rows<double, LG>((const double*)in, (double*)out, n0, n1, R, H, tw, (const double*)minuend, s);
    switch (lg) { GPA_FOR_LG(CASE) }
#undef CASE
  } else {
    if (minuend) return hipErrorInvalidValue;
#define CASE(LG) case LG: return dtype == 0 ? run_cols<float, LG>((const float*)in, (float*)out, n0, n1, R, H, tw, s) \
                                            : run_cols<double, LG>((const double*)in, (double*)out, n0, n1, R, H, tw, s);
    switch (lg) { GPA_FOR_LG(CASE) }
#undef CASE
  }
  return hipErrorInvalidValue;
}

}  // namespace gpa
