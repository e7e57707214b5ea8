// microbenchmark: scalar v_fma_f32 vs packed v_pk_fma_f32 / v_pk_add_f32 throughput on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  v2f p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  const v2f va = {a, a}, vb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // 8 independent scalar FMAs
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    } else if (MODE == 1) {   // 8 independent packed FMAs (16 flop-pairs)
      p0 = __builtin_elementwise_fma(p0, va, vb); p1 = __builtin_elementwise_fma(p1, va, vb);
      p2 = __builtin_elementwise_fma(p2, va, vb); p3 = __builtin_elementwise_fma(p3, va, vb);
      p4 = __builtin_elementwise_fma(p4, va, vb); p5 = __builtin_elementwise_fma(p5, va, vb);
      p6 = __builtin_elementwise_fma(p6, va, vb); p7 = __builtin_elementwise_fma(p7, va, vb);
    } else {   // 8 packed adds
      p0 += va; p1 += vb; p2 += va; p3 += vb; p4 += va; p5 += vb; p6 += va; p7 += vb;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                        p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
void run(const char* name, float* d, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 20000;   // 256-thread blocks = 1 wave per SIMD each
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 100, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = 8.0 * iters;   // per wave
  // cycles per wave-instruction per SIMD at 2.4 GHz nominal, waves_per_simd waves sharing the SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / (instr * waves_per_simd);
  printf("%-12s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<2>("v_pk_add_f32", d, w); }
  return 0;
}
