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

template <class T>
__global__ __launch_bounds__(256) void absshift_kernel(const cpx<T>* __restrict__ phat, int n0, int n1,
                                                      T* __restrict__ out) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (size_t)n0 * n1) return;
  const int i = (int)(o / n1), j = (int)(o - (size_t)i * n1);
  // DC of (image - mean) is zero; the reference holds its rounding residue there
  const cpx<T> v = phat[o];
  const T a = o == 0 ? T(0) : (T)hypot((double)v.x, (double)v.y);
  const int si = (i + n0 / 2) % n0, sj = (j + n1 / 2) % n1;   // np.fft.fftshift
  out[(size_t)si * n1 + sj] = a;
}

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

template <class T>
__global__ __launch_bounds__(256) void minmax_partial_kernel(const T* __restrict__ a, size_t count,
                                                            double* __restrict__ part) {
  __shared__ double smin[256], smax[256];
  double lo = 1e300, hi = -1e300;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const double v = (double)a[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
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

hipError_t launch_absshift(int dtype, const void* phat, int n0, int n1, void* out, hipStream_t s) {
  const unsigned grid = (unsigned)(((size_t)n0 * n1 + 255) / 256);
  if (dtype == 0)
    absshift_kernel<float><<<grid, 256, 0, s>>>((const cpx<float>*)phat, n0, n1, (float*)out);
  else
    absshift_kernel<double><<<grid, 256, 0, s>>>((const cpx<double>*)phat, n0, n1, (double*)out);
  return hipGetLastError();
}

hipError_t launch_gauss1d(int dtype, const void* in, void* out, int n0, int n1, int axis, const double* w, int R,
                          const void* minuend, hipStream_t s) {
  const unsigned grid = (unsigned)(((size_t)n0 * n1 + 255) / 256);
  if (dtype == 0)
    gauss1d_kernel<float><<<grid, 256, 0, s>>>((const float*)in, (float*)out, n0, n1, axis, w, R, (const float*)minuend);
  else
    gauss1d_kernel<double><<<grid, 256, 0, s>>>((const double*)in, (double*)out, n0, n1, axis, w, R,
                                                (const double*)minuend);
  return hipGetLastError();
}

// part: >= 512 doubles; thr: 3 doubles; count: 1 int (cleared)
hipError_t launch_localmax(int dtype, const void* smooth, int n0, int n1, double rel, double* part, double* thr,
                           int max_out, int* count, int32_t* coords, void* vals, hipStream_t s) {
  const size_t npx = (size_t)n0 * n1;
  const unsigned grid = (unsigned)((npx + 255) / 256);
  if (dtype == 0)
    minmax_partial_kernel<float><<<256, 256, 0, s>>>((const float*)smooth, npx, part);
  else
    minmax_partial_kernel<double><<<256, 256, 0, s>>>((const double*)smooth, npx, part);
  threshold_kernel<<<1, 256, 0, s>>>(part, 256, rel, thr, count);
  if (dtype == 0)
    localmax_kernel<float><<<grid, 256, 0, s>>>((const float*)smooth, n0, n1, thr, max_out, count, coords, (float*)vals);
  else
    localmax_kernel<double><<<grid, 256, 0, s>>>((const double*)smooth, n0, n1, thr, max_out, count, coords,
                                                 (double*)vals);
  return hipGetLastError();
}

}  // namespace gpa
