// microbenchmark: a chain of dependent small kernels against ONE persistent kernel with grid-wide barriers --
// the question behind "a persistent PCG for small images" (DESIGN 8 (2)).
// A phase = every workgroup reads a 16 KB chunk that ANOTHER workgroup (of another XCD) wrote in the previous
// phase, adds one, writes its own chunk: a 512^2 f32 PCG kernel in miniature (1 MiB per array, one dependent
// memory round trip, nothing to compute).
//   chain      : P launches of phase_kernel on one stream
//   persistent : one launch, P phases separated by a grid barrier (monotonic counter, device-scope release /
//                acquire; the spin is bounded -- a workgroup that waits more than ~50 ms sets an error flag and
//                every workgroup leaves, so a scheduling surprise cannot hang the GPU)
// The result is checked (every element = P), so a barrier that lets stale data through is seen.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/gridbarrier.hip -o /tmp/gridbarrier && /tmp/gridbarrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct alignas(16) V4 { float v[4]; };

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void phase(const float* __restrict__ in, float* __restrict__ out, int wg, int nwg, int chunk4) {
  // read the chunk of the workgroup 3 places on (another XCD under round-robin dispatch), write our own
  const int src = (wg + 3) % nwg;
  const V4* s = reinterpret_cast<const V4*>(in) + (size_t)src * chunk4;
  V4* d = reinterpret_cast<V4*>(out) + (size_t)wg * chunk4;
  for (int i = threadIdx.x; i < chunk4; i += blockDim.x) {
    V4 v = s[i];
    for (int c = 0; c < 4; ++c) v.v[c] += 1.f;
    d[i] = v;
  }
}

__global__ __launch_bounds__(256) void phase_kernel(const float* in, float* out, int chunk4) {
  phase(in, out, blockIdx.x, gridDim.x, chunk4);
}

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // our stores are visible before the count
    int good = 1;
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 2000000 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        good = 0;
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    ok = good;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok != 0;
}

__global__ __launch_bounds__(256) void persistent_kernel(float* a, float* b, int chunk4, int phases, unsigned* counter,
                                                          int* err) {
  float* in = a;
  float* out = b;
  for (int p = 0; p < phases; ++p) {
    phase(in, out, blockIdx.x, gridDim.x, chunk4);
    if (!grid_barrier(counter, (unsigned)(p + 1) * gridDim.x, err)) return;
    float* t = in; in = out; out = t;
  }
}

// barrier only (no memory phase): the floor
__global__ __launch_bounds__(256) void barrier_only_kernel(int phases, unsigned* counter, int* err) {
  for (int p = 0; p < phases; ++p)
    if (!grid_barrier(counter, (unsigned)(p + 1) * gridDim.x, err)) return;
}
__global__ void empty_kernel(int* x) { if (x && threadIdx.x == 1 << 20) *x = 1; }

int main() {
  const int P = 200;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned* counter;
  int* err;
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&err, 4));
  printf("%-10s %-8s %-14s %-16s %-16s %-14s\n", "workgroups", "KiB/wg", "chain us/phase", "persist us/phase", "barrier-only us", "empty launch us");
  for (int nwg : {64, 128, 256, 512}) {
    for (int kib : {4, 16}) {
      const int chunk4 = kib * 1024 / 16;
      const size_t n = (size_t)nwg * chunk4 * 4;
      float *a, *b;
      CK(hipMalloc(&a, n * 4));
      CK(hipMalloc(&b, n * 4));
      float tchain = 0, tpers = 0, tbar = 0, tempty = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(a, 0, n * 4, s));
        CK(hipEventRecord(e0, s));
        for (int p = 0; p < P; ++p) {
          if (p & 1) phase_kernel<<<nwg, 256, 0, s>>>(b, a, chunk4);
          else phase_kernel<<<nwg, 256, 0, s>>>(a, b, chunk4);
        }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&tchain, e0, e1));
      }
      std::vector<float> h(n);
      CK(hipMemcpy(h.data(), a, n * 4, hipMemcpyDeviceToHost));   // P even: result in a
      bool good = true;
      for (size_t i = 0; i < n; ++i) good &= h[i] == (float)P;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(a, 0, n * 4, s));
        CK(hipMemsetAsync(counter, 0, 4, s));
        CK(hipMemsetAsync(err, 0, 4, s));
        CK(hipEventRecord(e0, s));
        persistent_kernel<<<nwg, 256, 0, s>>>(a, b, chunk4, P, counter, err);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&tpers, e0, e1));
      }
      int herr = 0;
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h.data(), a, n * 4, hipMemcpyDeviceToHost));
      bool goodp = herr == 0;
      for (size_t i = 0; i < n; ++i) goodp &= h[i] == (float)P;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, s));
        CK(hipMemsetAsync(err, 0, 4, s));
        CK(hipEventRecord(e0, s));
        barrier_only_kernel<<<nwg, 256, 0, s>>>(P, counter, err);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&tbar, e0, e1));
      }
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int p = 0; p < P; ++p) empty_kernel<<<nwg, 256, 0, s>>>(nullptr);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&tempty, e0, e1));
      }
      printf("%-10d %-8d %-14.2f %-16.2f %-16.2f %-14.2f %s%s\n", nwg, kib, tchain * 1e3 / P, tpers * 1e3 / P, tbar * 1e3 / P,
             tempty * 1e3 / P, good ? "" : " CHAIN-WRONG", goodp ? "" : (herr ? " PERSISTENT-TIMEOUT" : " PERSISTENT-WRONG"));
      CK(hipFree(a));
      CK(hipFree(b));
    }
  }
  return 0;
}
