// Host emulator for pygpa_amd/csrc/gpa_fft.h: runs the per-thread phases of the
// workgroup FFT sequentially on the CPU (one loop iteration per GPU thread, one
// loop nest per barrier-delimited phase) and checks them against a naive DFT.
// Build: g++ -O2 -std=c++17 -I pygpa_amd/csrc tests/host/fft_emulator.cpp -o /tmp/fft_emu
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gpa_fft.h"

using namespace gpa;

template <class T, int LG, int ph, int EE>
void run_fwd_phase(std::vector<cpx<T>>& regs, std::vector<cpx<T>>& lds,
                   std::vector<typename WgFFT<T, LG, EE>::Twiddles>& tw) {
  using F = WgFFT<T, LG, EE>;
  for (int t = 0; t < F::TPF; ++t) {
    cpx<T>(&x)[EE] = *reinterpret_cast<cpx<T>(*)[EE]>(&regs[EE * t]);
    F::template fwd_phase<ph>(x, lds.data(), t, tw[t]);
  }
}
template <class T, int LG, int ph, int EE>
void run_inv_phase(std::vector<cpx<T>>& regs, std::vector<cpx<T>>& lds,
                   std::vector<typename WgFFT<T, LG, EE>::Twiddles>& tw) {
  using F = WgFFT<T, LG, EE>;
  for (int t = 0; t < F::TPF; ++t) {
    cpx<T>(&x)[EE] = *reinterpret_cast<cpx<T>(*)[EE]>(&regs[EE * t]);
    F::template inv_phase<ph>(x, lds.data(), t, tw[t]);
  }
}

template <class T, int LG, int EE = 16>
double test_one() {
  using F = WgFFT<T, LG, EE>;
  constexpr int E = EE;
  const int L = F::L;
  std::vector<std::complex<double>> in(L), ref(L);
  srand(LG * 7 + sizeof(T));
  for (auto& v : in) v = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
  // reference: radix-2 recursive-free O(L log L) in double via naive split (use O(L^2) for small L)
  {
    std::vector<std::complex<double>> w(L);
    for (int t = 0; t < L; ++t) w[t] = std::polar(1.0, -2 * M_PI * t / L);
    if (L <= 2048) {
      for (int k = 0; k < L; ++k) {
        std::complex<double> s = 0;
        for (int n = 0; n < L; ++n) s += in[n] * w[(long long)n * k % L];
        ref[k] = s;
      }
    } else {
      // iterative radix-2 DIT on a bit-reversed copy
      ref = in;
      int lg = LG;
      for (int i = 0; i < L; ++i) {
        int r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
        if (r > i) std::swap(ref[i], ref[r]);
      }
      for (int len = 2; len <= L; len <<= 1)
        for (int s = 0; s < L; s += len)
          for (int j = 0; j < len / 2; ++j) {
            auto u = ref[s + j], v = ref[s + j + len / 2] * w[j * (L / len)];
            ref[s + j] = u + v;
            ref[s + j + len / 2] = u - v;
          }
    }
  }
  std::vector<cpx<T>> table(L);
  for (int t = 0; t < L; ++t) table[t] = {(T)std::cos(-2 * M_PI * t / L), (T)std::sin(-2 * M_PI * t / L)};
  std::vector<typename F::Twiddles> tw(F::TPF);
  for (int t = 0; t < F::TPF; ++t) F::load_twiddles(tw[t], table.data(), t);
  std::vector<cpx<T>> regs(E * F::TPF), lds(F::LDS_ELEMS, cpx<T>{(T)1e30, (T)1e30});
  for (int t = 0; t < F::TPF; ++t)
    for (int i = 0; i < E; ++i) {
      auto v = in[t + F::TPF * i];
      regs[E * t + i] = {(T)v.real(), (T)v.imag()};
    }
  run_fwd_phase<T, LG, 0, EE>(regs, lds, tw);
  if constexpr (F::P > 1) run_fwd_phase<T, LG, 1, EE>(regs, lds, tw);
  if constexpr (F::P > 2) run_fwd_phase<T, LG, 2, EE>(regs, lds, tw);
  if constexpr (F::P > 3) run_fwd_phase<T, LG, 3, EE>(regs, lds, tw);
  double err = 0, nrm = 0;
  std::vector<char> seen(L, 0);
  for (int t = 0; t < F::TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int k = F::spec_index(t, i);
      if (k < 0 || k >= L || seen[k]) { printf("bad spec_index L=%d t=%d i=%d k=%d\n", L, t, i, k); exit(1); }
      seen[k] = 1;
      auto v = regs[E * t + i];
      err = std::max(err, std::abs(std::complex<double>(v.x, v.y) - ref[k]));
      nrm = std::max(nrm, std::abs(ref[k]));
    }
  double e1 = err / nrm;
  run_inv_phase<T, LG, 0, EE>(regs, lds, tw);
  if constexpr (F::P > 1) run_inv_phase<T, LG, 1, EE>(regs, lds, tw);
  if constexpr (F::P > 2) run_inv_phase<T, LG, 2, EE>(regs, lds, tw);
  if constexpr (F::P > 3) run_inv_phase<T, LG, 3, EE>(regs, lds, tw);
  double e2 = 0;
  for (int t = 0; t < F::TPF; ++t)
    for (int i = 0; i < E; ++i) {
      auto v = regs[E * t + i];
      e2 = std::max(e2, std::abs(std::complex<double>(v.x, v.y) / (double)L - in[t + F::TPF * i]));
    }
  printf("L=%5d E=%2d %s P=%d fwd_rel_err=%.2e roundtrip_err=%.2e\n", L, E, sizeof(T) == 4 ? "f32" : "f64", F::P, e1, e2);
  double tol = sizeof(T) == 4 ? 2e-6 : 4e-15;
  if (e1 > tol || e2 > tol) { printf("FAIL\n"); exit(1); }
  return e1;
}

template <class T>
void all() {
  test_one<T, 6>(); test_one<T, 7>(); test_one<T, 8>(); test_one<T, 9>(); test_one<T, 10>();
  test_one<T, 11>(); test_one<T, 12>(); test_one<T, 13>(); test_one<T, 14>();
  // eight elements per thread (the short transforms of the unwrap kernels)
  test_one<T, 6, 8>(); test_one<T, 7, 8>(); test_one<T, 8, 8>(); test_one<T, 9, 8>(); test_one<T, 10, 8>();
  test_one<T, 11, 8>(); test_one<T, 12, 8>();
}

int main() {
  all<float>();
  all<double>();
  printf("OK\n");
  return 0;
}
