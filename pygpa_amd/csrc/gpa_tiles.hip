// Data path of the tile pipeline (SURVEY.md 8(e) option 2; pygpa_amd/distributed.py) that is not the sweep or the
// unwrap: the image mean over tile interiors, the hand-over of a window's interior fields, the stitching of gathered
// tiles into the full-size fields.  The reference has no tiling (its only batching is the dask k-batch,
// geometric_phase_analysis.py:705-719); the semantics are this build's.  Every result stays on the device and every
// launch is on a caller-visible stream: the pipeline runs an image without a host synchronisation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpa_internal.h"

namespace gpa {
namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}
// sum over the workgroup, returned to thread 0 (fixed order: deterministic)
__device__ __forceinline__ double block_sum0(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum_d(v);
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

// ---- sum over the interior rectangles of a rank's windows ----------------------------------------
// rects[t] = (o0, o1, z0, z1) of window t; one workgroup per (band of TS_ROWS rows, tile) writes its partial sum to
// part[tile][band]; a second one-workgroup launch adds the partials in index order.  (Two launches, not a
// last-workgroup-done ticket: an agent-scope fence per workgroup -- L2 write-back and invalidate across the 8 XCDs -- made
// the ticket version take 1.0 ms for 320 MB at 16384^2; profiles/r04_tile_kernels.txt.)
constexpr int TS_ROWS = 16;
template <class T>
__global__ __launch_bounds__(256) void tile_sums_kernel(const T* __restrict__ wins, size_t win_stride, size_t pitch,
                                                       const int* __restrict__ rects, int nbands, double* part) {
  __shared__ double sh[8];
  const int t = blockIdx.y, band = blockIdx.x;
  const int o0 = rects[4 * t], o1 = rects[4 * t + 1], z0 = rects[4 * t + 2], z1 = rects[4 * t + 3];
  const T* w = wins + (size_t)t * win_stride;
  double acc = 0.0;
  const int ra = band * TS_ROWS;
  // (the TS_ROWS loads of a column are independent: all requested before the first is added)
  for (int c = threadIdx.x; c < z1; c += 256) {
    T v[TS_ROWS];
#pragma unroll
    for (int j = 0; j < TS_ROWS; ++j) v[j] = ra + j < z0 ? w[(size_t)(o0 + ra + j) * pitch + o1 + c] : T(0);
#pragma unroll
    for (int j = 0; j < TS_ROWS; ++j) acc += (double)v[j];
  }
  const double tot = block_sum0(acc, sh);
  if (threadIdx.x == 0) part[(size_t)t * nbands + band] = tot;
}
__global__ __launch_bounds__(256) void tile_sums_final_kernel(const double* __restrict__ part, int n, double* out) {
  __shared__ double sh[8];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];   // (fixed assignment of partials to threads)
  const double s = block_sum0(a, sh);
  if (threadIdx.x == 0) out[0] = s;
}

// mean of the whole image in the plan's element type from the (all-reduced) sum
template <class T>
__global__ void set_mean_kernel(const double* sum, double scale, T* mean_out) {
  mean_out[0] = (T)(sum[0] * scale);
}

// ---- interiors of a window's fields -> their places in the rank's tile buffer --------------------
struct FieldCopy {
  const void* src;
  void* dst;
  size_t src_pitch, dst_pitch;
  int rows, cols;
};
struct FieldCopies { FieldCopy f[6]; };
constexpr int CP_ROWS = 32;  // rows per workgroup of the two copy kernels (8 at a time in flight per thread)
template <class T>
__global__ __launch_bounds__(256) void copy_fields_kernel(FieldCopies fc) {
  const FieldCopy f = fc.f[blockIdx.z];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= f.cols) return;
  for (int j0 = 0; j0 < CP_ROWS; j0 += 8) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = blockIdx.y * CP_ROWS + j0 + j;
      v[j] = r < f.rows ? ((const T*)f.src)[(size_t)r * f.src_pitch + c] : T(0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = blockIdx.y * CP_ROWS + j0 + j;
      if (r < f.rows) ((T*)f.dst)[(size_t)r * f.dst_pitch + c] = v[j];
    }
  }
}

// ---- gathered tiles -> full-size fields ------------------------------------------------------------
// table[t] = (slot, r0, c0, z0, z1): tile t sits in slot `slot` of the gathered buffer and covers rows r0 .. r0 + z0,
// columns c0 .. c0 + z1 of the image; field f of a tile is clipped to the destination's rows / columns (the difference
// fields are one column / row short of the image).
struct StitchDst {
  void* dst[6];
  size_t pitch[6];
  int rows[6], cols[6];
};
template <class T>
__global__ __launch_bounds__(256) void stitch_kernel(const T* __restrict__ tiles, size_t slot_stride, size_t field_stride,
                                                    size_t tile_pitch, const int* __restrict__ table, int nf, StitchDst d) {
  const int t = blockIdx.z / nf, f = blockIdx.z % nf;
  const int slot = table[5 * t], r0 = table[5 * t + 1], c0 = table[5 * t + 2], z0 = table[5 * t + 3], z1 = table[5 * t + 4];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= z1 || c0 + c >= d.cols[f]) return;
  const T* src = tiles + (size_t)slot * slot_stride + (size_t)f * field_stride;
  for (int j0 = 0; j0 < CP_ROWS; j0 += 8) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = blockIdx.y * CP_ROWS + j0 + j;
      v[j] = (r < z0 && r0 + r < d.rows[f]) ? src[(size_t)r * tile_pitch + c] : T(0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = blockIdx.y * CP_ROWS + j0 + j;
      if (r < z0 && r0 + r < d.rows[f]) ((T*)d.dst[f])[(size_t)(r0 + r) * d.pitch[f] + c0 + c] = v[j];
    }
  }
}

}  // namespace

hipError_t launch_tile_sums(int dtype, const void* wins, size_t win_stride, size_t pitch, const int* rects_dev, int ntiles,
                            int max_rows, double* part, double* out, hipStream_t s) {
  const int nbands = (max_rows + TS_ROWS - 1) / TS_ROWS;
  GPA_PROF("tile_sums_kernel", s);
  if (dtype == 0)
    tile_sums_kernel<float><<<dim3(nbands, ntiles), 256, 0, s>>>((const float*)wins, win_stride, pitch, rects_dev, nbands, part);
  else
    tile_sums_kernel<double><<<dim3(nbands, ntiles), 256, 0, s>>>((const double*)wins, win_stride, pitch, rects_dev, nbands, part);
  tile_sums_final_kernel<<<1, 256, 0, s>>>(part, nbands * ntiles, out);
  return hipGetLastError();
}
int tile_sums_bands(int max_rows) { return (max_rows + TS_ROWS - 1) / TS_ROWS; }

hipError_t launch_set_mean(int dtype, const double* sum, double scale, void* mean_out, hipStream_t s) {
  if (dtype == 0) set_mean_kernel<float><<<1, 1, 0, s>>>(sum, scale, (float*)mean_out);
  else set_mean_kernel<double><<<1, 1, 0, s>>>(sum, scale, (double*)mean_out);
  return hipGetLastError();
}

hipError_t launch_copy_fields(int dtype, const void* const* src, void* const* dst, const size_t* src_pitch,
                              const size_t* dst_pitch, const int* rows, const int* cols, int nf, hipStream_t s) {
  if (nf < 1 || nf > 6) return hipErrorInvalidValue;
  FieldCopies fc;
  int mr = 0, mc = 0;
  for (int i = 0; i < nf; ++i) {
    fc.f[i] = {src[i], dst[i], src_pitch[i], dst_pitch[i], rows[i], cols[i]};
    mr = rows[i] > mr ? rows[i] : mr;
    mc = cols[i] > mc ? cols[i] : mc;
  }
  if (mr < 1 || mc < 1) return hipSuccess;
  GPA_PROF("tile_copy_fields_kernel", s);
  const dim3 grid((mc + 255) / 256, (mr + CP_ROWS - 1) / CP_ROWS, nf);
  if (dtype == 0) copy_fields_kernel<float><<<grid, 256, 0, s>>>(fc);
  else copy_fields_kernel<double><<<grid, 256, 0, s>>>(fc);
  return hipGetLastError();
}

hipError_t launch_stitch(int dtype, const void* tiles, size_t slot_stride, size_t field_stride, size_t tile_pitch,
                         const int* table_dev, int ntiles, int t0, int t1, int nf, void* const* dst, const size_t* dst_pitch,
                         const int* dst_rows, const int* dst_cols, hipStream_t s) {
  if (nf < 1 || nf > 6 || ntiles < 1) return hipErrorInvalidValue;
  StitchDst d;
  for (int i = 0; i < nf; ++i) {
    d.dst[i] = dst[i];
    d.pitch[i] = dst_pitch[i];
    d.rows[i] = dst_rows[i];
    d.cols[i] = dst_cols[i];
  }
  GPA_PROF("tile_stitch_kernel", s);
  const dim3 grid((t1 + 255) / 256, (t0 + CP_ROWS - 1) / CP_ROWS, ntiles * nf);
  if (dtype == 0)
    stitch_kernel<float><<<grid, 256, 0, s>>>((const float*)tiles, slot_stride, field_stride, tile_pitch, table_dev, nf, d);
  else
    stitch_kernel<double><<<grid, 256, 0, s>>>((const double*)tiles, slot_stride, field_stride, tile_pitch, table_dev, nf, d);
  return hipGetLastError();
}

}  // namespace gpa
