// Timing harness for pqdct_kernel (stencil + forward row transform of the fused PCG iteration) at 4096-point f32 rows,
// outside the solver: random finite inputs, HIP events, and an occupancy sweep (dynamic LDS padded so that 4, 3, 2 and 1
// workgroups fit a CU).  Results are not checked here -- tests/test_gpu_unwrap_long.py does that through the C-ABI.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize [-DGPA_PQ_PLAIN] \
//         tools/micro/pqdct_micro.hip -Lpygpa_amd -lgpa_hip -Wl,-rpath,'$ORIGIN/../../../pygpa_amd' -o tools/micro/bin/pqdct_micro
//   (tools/micro/bin/ is git-ignored and travels to the GPU box)      gpurun -- tools/micro/bin/pqdct_micro [rows]
#include "../../pygpa_amd/csrc/gpa_unwrap_pqdct.hip"
#include <cstdio>
#include <random>
#include <vector>
using namespace gpa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int n0 = argc > 1 ? atoi(argv[1]) : 4096, N = 4096, reps = 20;
  const size_t px = (size_t)n0 * N;
  std::vector<float> h(px);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(0.1f, 1.0f);
  for (auto& v : h) v = U(rng);
  float *p, *w, *D;
  cpx<float>*tw, *wk;
  int* flags;
  double* part;
  CK(hipMalloc(&p, px * 4)); CK(hipMalloc(&w, px * 4)); CK(hipMalloc(&D, px * 4));
  CK(hipMalloc(&tw, 8 * N * 8)); CK(hipMalloc(&wk, N * 8)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&part, PART_N * 8));
  CK(hipMemcpy(p, h.data(), px * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), px * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(tw, h.data(), 8 * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(wk, h.data(), N * 8, hipMemcpyHostToDevice));
  CK(hipMemset(flags, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  using G = RowGeom<float, 12>;
  auto k = pqdct_kernel<float, 12>;
  const int lds_opts[4] = {(int)G::LDS_BYTES, 50 * 1024, 70 * 1024, 150 * 1024};   // 4, 3, 2, 1 workgroups per CU
  for (int which = 0; which < 4; ++which) {
    const int lb = lds_opts[which];
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    float best = 1e9, sum = 0;
    for (int r = 0; r < reps + 3; ++r) {
      CK(hipEventRecord(e0, 0));
      k<<<dim3(n0 / 2, 1, 1), G::THREADS, lb, 0>>>(p, w, D, n0, tw, wk, flags, part, px);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    printf("pqdct_kernel<float,12> rows %d  %d workgroups/CU  avg %.1f us  min %.1f us\n", n0, 4 - which, 1e3f * sum / reps, 1e3f * best);
  }
  return 0;
}
