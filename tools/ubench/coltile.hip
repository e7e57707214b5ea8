// microbenchmark: what the column-tile access pattern of passA / colsolve can stream on gfx950.
// An N x N float matrix is processed in column tiles of SEG bytes per row (the kernels' row segments):
// a workgroup of THREADS threads owns one tile [N rows][SEG bytes]; lane groups of SEG/16 adjacent lanes
// cover one row segment with 16-byte accesses, the other lanes take other rows.  Modes:
//   0 copy    : load the tile, store it to the second matrix (colsolve's traffic, no compute, no barrier)
//   1 store   : store only (passA's dominant traffic)
//   2 load    : load only (sum -> one value per thread so the loads are not dead)
//   3 copy with a load -> barrier -> "compute" (busy VALU loop of W instructions per element) -> barrier -> store
//     phase structure, i.e. what a workgroup that cannot overlap its own phases looks like
// Occupancy is set by the dynamic LDS request (LDS bytes per workgroup), like the real kernels.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/coltile.hip -o /tmp/coltile && /tmp/coltile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct alignas(16) V4 { float v[4]; };

template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ in, float* __restrict__ out, int n, int seg_bytes,
                                          int work, float* sink) {
  extern __shared__ char smem[];
  const int lanes_per_row = seg_bytes / 16;                 // lanes that share one row segment
  const int rows_per_pass = blockDim.x / lanes_per_row;     // rows covered by one access of the workgroup
  const int lr = threadIdx.x % lanes_per_row, r0 = threadIdx.x / lanes_per_row;
  // XCD-aware tile order as in the real kernels
  int tile = blockIdx.x;
  if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const size_t col0 = (size_t)tile * (seg_bytes / 4) + lr * 4;
  const int per_thread = n / rows_per_pass;                  // accesses per thread (16 for the real kernels at 512 thr, 32 B)
  V4 acc = {0, 0, 0, 0};
  if (MODE == 1) {
    V4 v = {1.f * threadIdx.x, 2.f, 3.f, 4.f};
    for (int i = 0; i < per_thread; ++i) {
      const size_t row = r0 + (size_t)rows_per_pass * i;
      *reinterpret_cast<V4*>(out + row * n + col0) = v;
      v.v[0] += 1.f;
    }
    return;
  }
  if (MODE == 2) {
    for (int i = 0; i < per_thread; ++i) {
      const size_t row = r0 + (size_t)rows_per_pass * i;
      const V4 v = *reinterpret_cast<const V4*>(in + row * n + col0);
      for (int c = 0; c < 4; ++c) acc.v[c] += v.v[c];
    }
    if (acc.v[0] + acc.v[1] + acc.v[2] + acc.v[3] == 12345.678f) sink[0] = 1.f;
    return;
  }
  // copy: 16 accesses per thread are kept in registers (like the real kernels), tiles with more rows loop
  for (int base = 0; base < per_thread; base += 16) {
    V4 x[16];
    const int cnt = per_thread - base < 16 ? per_thread - base : 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const size_t row = r0 + (size_t)rows_per_pass * (base + i);
      if (i < cnt) x[i] = *reinterpret_cast<const V4*>(in + row * n + col0);
    }
    if (MODE == 3) {
      __syncthreads();
      for (int w = 0; w < work; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          for (int c = 0; c < 4; ++c) x[i].v[c] = __builtin_fmaf(x[i].v[c], 1.0001f, 0.5f);
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const size_t row = r0 + (size_t)rows_per_pass * (base + i);
      if (i < cnt) *reinterpret_cast<V4*>(out + row * n + col0) = x[i];
    }
  }
}

template <int MODE>
float run(const float* in, float* out, float* sink, int n, int seg, int threads, int lds, int work, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tiles = n * 4 / seg;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<tiles, threads, lds>>>(in, out, n, seg, work, sink);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<MODE><<<tiles, threads, lds>>>(in, out, n, seg, work, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int n = 4096;
  float *in, *out, *sink;
  hipMalloc(&in, (size_t)n * n * 4); hipMalloc(&out, (size_t)n * n * 4); hipMalloc(&sink, 16);
  hipMemset(in, 0, (size_t)n * n * 4);
  const double mb = (double)n * n * 4 / 1e6;
  printf("column tiles of a %d^2 float matrix (%.0f MB per pass); us per launch and GB/s of moved bytes\n", n, mb);
  printf("%-8s %-8s %-8s %-8s | %-18s %-18s %-18s\n", "seg B", "threads", "LDS KiB", "WG/CU", "copy", "store", "load");
  struct Cfg { int seg, threads, lds; };
  std::vector<Cfg> cfgs = {{16, 256, 70}, {16, 512, 140}, {32, 256, 70}, {32, 512, 140}, {32, 512, 70}, {32, 1024, 140},
                           {64, 512, 140}, {64, 1024, 140}, {64, 512, 70}, {128, 1024, 140}, {128, 512, 70}, {256, 1024, 140},
                           {32, 256, 32}, {64, 256, 32}, {128, 256, 32}, {1024, 256, 32}};
  for (auto c : cfgs) {
    const int lds = c.lds * 1024;
    const float t0 = run<0>(in, out, sink, n, c.seg, c.threads, lds, 0, 20);
    const float t1 = run<1>(in, out, sink, n, c.seg, c.threads, lds, 0, 20);
    const float t2 = run<2>(in, out, sink, n, c.seg, c.threads, lds, 0, 20);
    printf("%-8d %-8d %-8d %-8d | %7.1f us %6.0f   %7.1f us %6.0f   %7.1f us %6.0f\n", c.seg, c.threads, c.lds, 160 / c.lds,
           t0 * 1e3, 2 * mb / t0, t1 * 1e3, mb / t1, t2 * 1e3, mb / t2);
  }
  printf("\nphase-serial copy (load -> barrier -> W fma per element -> barrier -> store), 32 B segments, 512 threads:\n");
  for (int lds : {140, 70})
    for (int work : {0, 8, 16, 32, 64}) {
      const float t = run<3>(in, out, sink, n, 32, 512, lds * 1024, work, 20);
      printf("  LDS %3d KiB (%d WG/CU)  W = %2d: %7.1f us  %6.0f GB/s\n", lds, 160 / lds, work, t * 1e3, 2 * mb / t);
    }
  return 0;
}
