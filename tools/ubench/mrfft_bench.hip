// microbenchmark: where the time of an LDS-resident mixed-radix transform kernel goes (gpa_mrfft.h).
// One workgroup transforms NF row pairs of length n: global -> LDS -> passes -> global, like the unwrap kernels.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I pygpa_amd/csrc tools/ubench/mrfft_bench.hip -o /tmp/mrfft_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <cmath>
#include "gpa_mrfft.h"
using namespace gpa;

// mode: 0 full, 1 no twiddles, 2 no butterflies and no twiddles (LDS traffic + barriers only)
template <class T, int R, int MODE>
__device__ __attribute__((noinline)) void pass(cpx<T>* lds, int n, int s, unsigned mg, int tid, int Tn, const cpx<T>* W) {
  cpx<T> x[MR_REGS];
  mr_load<T, R>(x, reinterpret_cast<const T*>(lds), n, tid, Tn);
  __syncthreads();
  constexpr int NB = MR_REGS / R;
  const int nb = n / R;
  const bool last = s * R == n;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int b = tid + u * Tn;
    if (b < nb) {
      if (MODE < 2) mr_bfly<T, R>(x + u * R);
      const int pp = s == 1 ? b : (int)mr_mulhi((unsigned)b, mg), q = b - pp * s;
      const int base = q + s * R * pp, ws = pp * s;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        cpx<T> v = x[u * R + k];
        if (MODE == 0 && k > 0 && !last) v = cmul(v, W[mr_pad(ws * k)]);
        lds[mr_pad(base + s * k)] = v;
      }
    }
  }
}

template <class T, int MODE>
__global__ __launch_bounds__(1024) void k(const T* __restrict__ in, T* __restrict__ out, int rows, const MrPlan pl, int rs,
                                          const cpx<T>* __restrict__ W, int npass, int wlds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = pl.n, Tn = pl.T;
  const int tid = threadIdx.x % Tn, f = threadIdx.x / Tn, nf = blockDim.x / Tn;
  cpx<T>* lds = reinterpret_cast<cpx<T>*>(smem) + (size_t)f * rs;
  cpx<T>* Wl = reinterpret_cast<cpx<T>*>(smem) + (size_t)nf * rs;
  if (wlds) for (int i = threadIdx.x; i < n; i += blockDim.x) Wl[mr_pad(i)] = W[i];
  const cpx<T>* Wt = wlds ? Wl : W;   // (global table: unpadded indices differ, only the timing matters here)
  const int pr = blockIdx.x * nf + f;
  const bool v = 2 * pr + 1 < rows;
  const size_t oa = (size_t)(v ? 2 * pr : 0) * n, ob = oa + n;
  for (int c = tid; c < n; c += Tn) lds[mr_pad((c & 1) ? n - 1 - (c >> 1) : (c >> 1))] = {in[oa + c], in[ob + c]};
  __syncthreads();
  for (int p = 0; p < npass && p < pl.np; ++p) {
    const int s = pl.stride[p];
    const unsigned mg = pl.magic[p];
#define C(R) case R: pass<T, R, MODE>(lds, n, s, mg, tid, Tn, Wt); break;
    switch (pl.radix[p]) { C(16) C(8) C(4) C(2) C(3) C(5) C(7) C(11) C(13) C(6) C(10) C(12) C(14) C(15) }
#undef C
    __syncthreads();
  }
  if (!v) return;
  for (int c = tid; c < n; c += Tn) {
    const cpx<T> z = lds[mr_pad(c)];
    out[oa + c] = z.x;
    out[ob + c] = z.y;
  }
}

template <class T, int MODE>
float run(const T* in, T* out, int rows, const MrPlan& pl, const cpx<T>* W, int nf, int npass, int wlds) {
  const int rs = mr_lds_elems(pl.n);
  const size_t lds = (size_t)(nf + 1) * rs * sizeof(cpx<T>);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  const int grid = (rows / 2 + nf - 1) / nf;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<T, MODE><<<grid, nf * pl.T, lds>>>(in, out, rows, pl, rs, W, npass, wlds);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<T, MODE><<<grid, nf * pl.T, lds>>>(in, out, rows, pl, rs, W, npass, wlds);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) return -1;
  return ms / 20 * 1e3f;
}

int main(int argc, char** argv) {
  for (int n : {500, 1000, 3000, 4096}) {
    MrPlan pl;
    if (!mr_make_plan(n, &pl)) return 1;
    const int rows = n;
    float *in, *out; cpx<float>* W;
    hipMalloc(&in, (size_t)rows * n * 4); hipMalloc(&out, (size_t)rows * n * 4); hipMalloc(&W, (size_t)(n + n / 32 + 8) * 8);
    hipMemset(in, 0, (size_t)rows * n * 4);
    std::vector<cpx<float>> w(n);
    for (int i = 0; i < n; ++i) w[i] = {(float)cos(-2 * M_PI * i / n), (float)sin(-2 * M_PI * i / n)};
    hipMemcpy(W, w.data(), n * 8, hipMemcpyHostToDevice);
    printf("n = %d  (T = %d, passes:", n, pl.T);
    for (int p = 0; p < pl.np; ++p) printf(" %d", pl.radix[p]);
    printf(")  %d x %d f32, us per launch\n", rows, n);
    for (int nf : {1, 2, 4}) {
      if (nf * pl.T > 1024) continue;
      printf("  NF %d:", nf);
      for (int np = 0; np <= pl.np; ++np) printf("  %dp %.1f", np, run<float, 0>(in, out, rows, pl, W, nf, np, 1));
      printf("  | all passes: W in global %.1f, no twiddles %.1f, LDS traffic only %.1f\n", run<float, 0>(in, out, rows, pl, W, nf, 99, 0),
             run<float, 1>(in, out, rows, pl, W, nf, 99, 1), run<float, 2>(in, out, rows, pl, W, nf, 99, 1));
    }
    hipFree(in); hipFree(out); hipFree(W);
  }
  return 0;
}
