// f-3: Bragg-peak candidates of an image, the device part of extract_primary_ks
// (geometric_phase_analysis.py:397-505):
//   fftim  = | fftshift( DFT of the periodic component ) |               (:427-429, a9 kernels)
//   smooth = gaussian_filter(fftim, sigma) [- gaussian_filter(fftim, 50)]  (:431-433)
//   peaks  = skimage.feature.peak_local_max(smooth, threshold_rel)         (:437)
// The Gaussian filters are scipy.ndimage.gaussian_filter's: separable correlation with the
// normalised kernel exp(-x^2 / 2 sigma^2), radius int(4 sigma + 0.5), half-sample-symmetric
// ("reflect") boundary, axis 0 first, accumulated in double in scipy's order (outer taps first).
// peak_local_max with its defaults is a 3x3 maximum test above
// max(min, threshold_rel * max) away from the one-pixel border.  All of it is streaming /
// cache-resident stencil work; the candidate list that leaves the device is a few entries long.
#include "gpa_internal.h"

namespace gpa {
namespace {

__device__ __forceinline__ int reflect_index(int i, int n) {
  // scipy 'reflect': (d c b a | a b c d | d c b a), period 2n
  const int p = 2 * n;
  int m = i % p;
  if (m < 0) m += p;
  return m < n ? m : p - 1 - m;
}

// out = [minuend -] sum_k w[k] in[reflect(c + k - R)] along `axis`
template <class T>
__global__ __launch_bounds__(256) void gauss1d_kernel(const T* __restrict__ in, T* __restrict__ out, int n0, int n1,
                                                     int axis, const double* __restrict__ w, int R,
                                                     const T* __restrict__ minuend) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (size_t)n0 * n1) return;
  const int i = (int)(o / n1), j = (int)(o - (size_t)i * n1);
  const int n = axis == 0 ? n0 : n1, c = axis == 0 ? i : j;
  const size_t stride = axis == 0 ? (size_t)n1 : 1, base = axis == 0 ? (size_t)j : (size_t)i * n1;
  double acc = (double)in[base + (size_t)c * stride] * w[R];
  if (c - R >= 0 && c + R < n) {
    for (int k = R; k >= 1; --k)
      acc += ((double)in[base + (size_t)(c - k) * stride] + (double)in[base + (size_t)(c + k) * stride]) * w[R - k];
  } else {
    for (int k = R; k >= 1; --k)
      acc += ((double)in[base + (size_t)reflect_index(c - k, n) * stride] +
              (double)in[base + (size_t)reflect_index(c + k, n) * stride]) * w[R - k];
  }
  out[o] = minuend ? (T)((double)minuend[o] - acc) : (T)acc;
}

// both axes of a SHORT kernel (radius <= GAUSS2D_RMAX) in one pass over the image: a 32 x 64 output tile, its halo loaded
// once into LDS through the reflection of both axes, axis 0 first (the intermediate is rounded to the array's type exactly
// where scipy stores it), then axis 1 -- the same sums in the same order as two launches of gauss1d_kernel, bit for bit,
// with one read and one write of the image instead of two each
constexpr int GAUSS2D_RMAX = 11, G2_TH = 32, G2_TW = 64;
template <class T>
__global__ __launch_bounds__(256) void gauss2d_small_kernel(const T* __restrict__ in, T* out, int n0, int n1,
                                                           const double* __restrict__ w, int R, const T* minuend) {
  extern __shared__ __attribute__((aligned(16))) char g2_smem[];
  const int PW = G2_TW + 2 * R, PH = G2_TH + 2 * R;
  T* tin = reinterpret_cast<T*>(g2_smem);          // [PH][PW]
  T* mid = tin + PH * PW;                          // [G2_TH][PW]
  const int i0 = blockIdx.y * G2_TH, j0 = blockIdx.x * G2_TW;
  // threads as 4 rows x 64 columns: no division by a run-time width anywhere, the row's reflection once per wavefront
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  int jc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) jc[q] = reflect_index(j0 - R + tx + 64 * q, n1);
  for (int a = ty; a < PH; a += 4) {
    const T* row = in + (size_t)reflect_index(i0 - R + a, n0) * n1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (tx + 64 * q < PW) tin[a * PW + tx + 64 * q] = row[jc[q]];
  }
  __syncthreads();
  for (int a = ty; a < G2_TH; a += 4) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int b2 = tx + 64 * q;
      if (b2 >= PW) continue;
      const T* col = tin + (a + R) * PW + b2;
      double acc = (double)col[0] * w[R];
      for (int k = R; k >= 1; --k) acc += ((double)col[-k * PW] + (double)col[k * PW]) * w[R - k];
      mid[a * PW + b2] = (T)acc;
    }
  }
  __syncthreads();
  const int j = j0 + tx;
  for (int a = ty; a < G2_TH; a += 4) {
    const int i = i0 + a;
    if (i >= n0 || j >= n1) continue;
    const T* row = mid + a * PW + tx + R;
    double acc = (double)row[0] * w[R];
    for (int k = R; k >= 1; --k) acc += ((double)row[-k] + (double)row[k]) * w[R - k];
    const size_t o = (size_t)i * n1 + j;
    out[o] = minuend ? (T)((double)minuend[o] - acc) : (T)acc;
  }
}

// grid-stride min / max with 16-byte loads and two independent chains per thread (64 MB in ~15 us; the round-5 kernel ran 256
// workgroups of scalar loads: 72 us)
template <class T>
__global__ __launch_bounds__(256) void minmax_partial_kernel(const T* __restrict__ a, size_t count,
                                                            double* __restrict__ part) {
  __shared__ double smin[256], smax[256];
  constexpr int W = 16 / sizeof(T);
  struct alignas(16) Vec { T v[W]; };
  T lo = a[0], hi = a[0];
  const size_t nv = (reinterpret_cast<size_t>(a) & 15) == 0 ? count / W : 0, step = (size_t)gridDim.x * 256;
  const Vec* av = reinterpret_cast<const Vec*>(a);
  T lo2 = lo, hi2 = hi;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + step < nv; i += 2 * step) {
    const Vec u = av[i], w = av[i + step];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      lo = u.v[k] < lo ? u.v[k] : lo; hi = u.v[k] > hi ? u.v[k] : hi;
      lo2 = w.v[k] < lo2 ? w.v[k] : lo2; hi2 = w.v[k] > hi2 ? w.v[k] : hi2;
    }
  }
  if (i < nv) {
    const Vec u = av[i];
#pragma unroll
    for (int k = 0; k < W; ++k) { lo = u.v[k] < lo ? u.v[k] : lo; hi = u.v[k] > hi ? u.v[k] : hi; }
  }
  for (size_t j = nv * W + (size_t)blockIdx.x * 256 + threadIdx.x; j < count; j += step) {
    lo = a[j] < lo ? a[j] : lo; hi = a[j] > hi ? a[j] : hi;
  }
  smin[threadIdx.x] = (double)(lo2 < lo ? lo2 : lo);
  smax[threadIdx.x] = (double)(hi2 > hi ? hi2 : hi);
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + s]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}

// thr[0] = max(min, rel * max), thr[1] = min, thr[2] = max; also clears the peak counter
__global__ __launch_bounds__(256) void threshold_kernel(const double* __restrict__ part, int nparts, double rel,
                                                       double* __restrict__ thr, int* __restrict__ count) {
  __shared__ double smin[256], smax[256];
  double lo = 1e300, hi = -1e300;
  for (int i = threadIdx.x; i < nparts; i += 256) { lo = fmin(lo, part[2 * i]); hi = fmax(hi, part[2 * i + 1]); }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + s]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    thr[0] = fmax(smin[0], rel * smax[0]);
    thr[1] = smin[0];
    thr[2] = smax[0];
    *count = 0;
  }
}

template <class T>
__global__ __launch_bounds__(256) void localmax_kernel(const T* __restrict__ s, int n0, int n1,
                                                      const double* __restrict__ thr, int max_out,
                                                      int* __restrict__ count, int32_t* __restrict__ coords,
                                                      T* __restrict__ vals) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (size_t)n0 * n1) return;
  const int i = (int)(o / n1), j = (int)(o - (size_t)i * n1);
  if (i < 1 || j < 1 || i >= n0 - 1 || j >= n1 - 1) return;   // exclude_border = min_distance = 1
  const T v = s[o];
  if (!((double)v > thr[0])) return;
  bool is_max = true;
#pragma unroll
  for (int di = -1; di <= 1; ++di)
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj) is_max = is_max && !(s[o + (ptrdiff_t)di * n1 + dj] > v);
  if (!is_max) return;
  const int slot = atomicAdd(count, 1);
  if (slot < max_out) {
    coords[2 * slot] = i;
    coords[2 * slot + 1] = j;
    vals[slot] = v;
  }
}

// ---- f-4: gaussian_deconvolve (geometric_phase_analysis.py:892-904) -----------------------------
// np.pad(mode='reflect') (whole-sample mirror, edge not repeated) of the m0 x m1 field into the
// n0 x n1 = (m0 + 2 pad) x (m1 + 2 pad) complex work array
template <class T>
__global__ __launch_bounds__(256) void deconv_pack_kernel(const T* __restrict__ data, int m0, int m1, int pad,
                                                         cpx<T>* __restrict__ Z) {
  const int n0 = m0 + 2 * pad, n1 = m1 + 2 * pad;
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (size_t)n0 * n1) return;
  int i = (int)(o / n1) - pad, j = (int)(o - (size_t)(o / n1) * n1) - pad;
  i = i < 0 ? -i : (i >= m0 ? 2 * (m0 - 1) - i : i);
  j = j < 0 ? -j : (j >= m1 ? 2 * (m1 - 1) - j : j);
  Z[o] = {data[(size_t)i * m1 + j], T(0)};
}

// Z <- conj(Z * W), W = G / (G^2 + balance L^2): the Wiener-Hunt filter of skimage.restoration.wiener
// for the Gaussian transfer function G = gx[kx] gy[ky] and the Laplacian regulariser
// L = 4 - 2 cos(2 pi kx / n0) - 2 cos(2 pi ky / n1); conjugated so that a second forward DFT inverts
template <class T>
__global__ __launch_bounds__(256) void deconv_filter_kernel(cpx<T>* __restrict__ Z, int n0, int n1,
                                                           const double* __restrict__ gx,
                                                           const double* __restrict__ gy, double balance) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (size_t)n0 * n1) return;
  const int kx = (int)(o / n1), ky = (int)(o - (size_t)kx * n1);
  const double G = gx[kx] * gy[ky];
  const double L = 4.0 - 2.0 * cospi(2.0 * kx / (double)n0) - 2.0 * cospi(2.0 * ky / (double)n1);
  const double den = G * G + balance * L * L;
  const double W = den > 0.0 ? G / den : 0.0;
  const cpx<T> z = Z[o];
  Z[o] = {(T)(z.x * W), (T)(-z.y * W)};
}

template <class T>
__global__ __launch_bounds__(256) void deconv_unpack_kernel(const cpx<T>* __restrict__ Z, int m0, int m1, int pad,
                                                           double scale, T* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (size_t)m0 * m1) return;
  const int i = (int)(o / m1), j = (int)(o - (size_t)i * m1);
  out[o] = (T)(Z[(size_t)(i + pad) * (m1 + 2 * pad) + (j + pad)].x * scale);
}

}  // namespace

hipError_t launch_deconv_pack(int dtype, const void* data, int m0, int m1, int pad, void* Z, hipStream_t s) {
  const unsigned grid = (unsigned)(((size_t)(m0 + 2 * pad) * (m1 + 2 * pad) + 255) / 256);
  if (dtype == 0)
    deconv_pack_kernel<float><<<grid, 256, 0, s>>>((const float*)data, m0, m1, pad, (cpx<float>*)Z);
  else
    deconv_pack_kernel<double><<<grid, 256, 0, s>>>((const double*)data, m0, m1, pad, (cpx<double>*)Z);
  return hipGetLastError();
}

hipError_t launch_deconv_filter(int dtype, void* Z, int n0, int n1, const double* gx, const double* gy, double balance,
                                hipStream_t s) {
  const unsigned grid = (unsigned)(((size_t)n0 * n1 + 255) / 256);
  if (dtype == 0)
    deconv_filter_kernel<float><<<grid, 256, 0, s>>>((cpx<float>*)Z, n0, n1, gx, gy, balance);
  else
    deconv_filter_kernel<double><<<grid, 256, 0, s>>>((cpx<double>*)Z, n0, n1, gx, gy, balance);
  return hipGetLastError();
}

hipError_t launch_deconv_unpack(int dtype, const void* Z, int m0, int m1, int pad, void* out, hipStream_t s) {
  const unsigned grid = (unsigned)(((size_t)m0 * m1 + 255) / 256);
  const double scale = 1.0 / ((double)(m0 + 2 * pad) * (double)(m1 + 2 * pad));
  if (dtype == 0)
    deconv_unpack_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)Z, m0, m1, pad, scale, (float*)out);
  else
    deconv_unpack_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)Z, m0, m1, pad, scale, (double*)out);
  return hipGetLastError();
}

// gaussian_filter with a kernel of radius R <= GAUSS2D_RMAX, both axes in one launch
bool gauss2d_small_ok(int R) { return R <= GAUSS2D_RMAX; }
hipError_t launch_gauss2d_small(int dtype, const void* in, void* out, int n0, int n1, const double* w, int R,
                                const void* minuend, hipStream_t s) {
  const dim3 grid((n1 + G2_TW - 1) / G2_TW, (n0 + G2_TH - 1) / G2_TH);
  const size_t elems = (size_t)(G2_TH + 2 * R) * (G2_TW + 2 * R) + (size_t)G2_TH * (G2_TW + 2 * R);
  GPA_PROF("gauss2d_small_kernel", s);
  if (dtype == 0)
    gauss2d_small_kernel<float><<<grid, 256, elems * sizeof(float), s>>>((const float*)in, (float*)out, n0, n1, w, R, (const float*)minuend);
  else
    gauss2d_small_kernel<double><<<grid, 256, elems * sizeof(double), s>>>((const double*)in, (double*)out, n0, n1, w, R,
                                                                          (const double*)minuend);
  return hipGetLastError();
}

hipError_t launch_gauss1d(int dtype, const void* in, void* out, int n0, int n1, int axis, const double* w, int R,
                          const void* minuend, hipStream_t s) {
  const unsigned grid = (unsigned)(((size_t)n0 * n1 + 255) / 256);
  GPA_PROF("gauss1d_kernel", s);
  if (dtype == 0)
    gauss1d_kernel<float><<<grid, 256, 0, s>>>((const float*)in, (float*)out, n0, n1, axis, w, R, (const float*)minuend);
  else
    gauss1d_kernel<double><<<grid, 256, 0, s>>>((const double*)in, (double*)out, n0, n1, axis, w, R,
                                                (const double*)minuend);
  return hipGetLastError();
}

// part: >= 2 * PEAK_PARTS doubles; thr: 3 doubles; count: 1 int (cleared)
hipError_t launch_localmax(int dtype, const void* smooth, int n0, int n1, double rel, double* part, double* thr,
                           int max_out, int* count, int32_t* coords, void* vals, hipStream_t s) {
  const size_t npx = (size_t)n0 * n1;
  const unsigned grid = (unsigned)((npx + 255) / 256);
  const size_t want = npx / 8192;
  const int nparts = want > PEAK_PARTS ? PEAK_PARTS : (want < 32 ? 32 : (int)want);
  {
    GPA_PROF("minmax_kernels", s);
    if (dtype == 0)
      minmax_partial_kernel<float><<<nparts, 256, 0, s>>>((const float*)smooth, npx, part);
    else
      minmax_partial_kernel<double><<<nparts, 256, 0, s>>>((const double*)smooth, npx, part);
    threshold_kernel<<<1, 256, 0, s>>>(part, nparts, rel, thr, count);
  }
  GPA_PROF("localmax_kernel", s);
  if (dtype == 0)
    localmax_kernel<float><<<grid, 256, 0, s>>>((const float*)smooth, n0, n1, thr, max_out, count, coords, (float*)vals);
  else
    localmax_kernel<double><<<grid, 256, 0, s>>>((const double*)smooth, n0, n1, thr, max_out, count, coords,
                                                 (double*)vals);
  return hipGetLastError();
}

}  // namespace gpa
