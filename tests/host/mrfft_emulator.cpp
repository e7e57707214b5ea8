// Host emulator for pygpa_amd/csrc/gpa_mrfft.h: the mixed-radix LDS-resident workgroup FFT, run thread by
// thread on the CPU (one loop iteration per GPU thread, one loop nest per barrier-delimited half pass) and
// checked against a naive DFT in double.
// Build: g++ -O2 -std=c++17 -I pygpa_amd/csrc tests/host/mrfft_emulator.cpp -o /tmp/mrfft_emu
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gpa_mrfft.h"

using namespace gpa;

template <class T, int R>
void half_pass_load(std::vector<cpx<T>>& regs, const std::vector<cpx<T>>& lds, const MrPlan& pl) {
  for (int t = 0; t < pl.T; ++t) mr_load<T, R>(&regs[(size_t)MR_REGS * t], reinterpret_cast<const T*>(lds.data()), pl.n, t, pl.T);
}
template <class T, int R>
void half_pass_store(std::vector<cpx<T>>& regs, std::vector<cpx<T>>& lds, const MrPlan& pl, int p,
                     const std::vector<cpx<T>>& W) {
  for (int t = 0; t < pl.T; ++t)
    mr_store<T, R>(&regs[(size_t)MR_REGS * t], reinterpret_cast<T*>(lds.data()), pl.n, pl.stride[p], pl.magic[p], t, pl.T,
                   reinterpret_cast<const T*>(W.data()));
}

template <class T>
double test_one(int n, bool expect_plan) {
  MrPlan pl;
  const bool ok = mr_make_plan(n, &pl);
  if (ok != expect_plan) {
    printf("n = %d: plan %s\n", n, ok ? "made but not expected" : "missing");
    return 1.0;
  }
  if (!ok) return 0.0;
  // the invariants the kernels rely on
  int prod = 1;
  for (int p = 0; p < pl.np; ++p) {
    if (pl.stride[p] != prod) return 1.0;
    prod *= pl.radix[p];
    const int nb = n / pl.radix[p], nbmax = MR_REGS / pl.radix[p];
    if ((nb + pl.T - 1) / pl.T > nbmax) { printf("n = %d: %d butterflies of radix %d on %d threads\n", n, nb, pl.radix[p], pl.T); return 1.0; }
    for (int b = 0; b < n && pl.stride[p] > 1; ++b)
      if ((int)mr_mulhi((unsigned)b, pl.magic[p]) != b / pl.stride[p]) { printf("n = %d: magic division fails\n", n); return 1.0; }
  }
  if (prod != n || pl.T % 64 || (n + pl.T - 1) / pl.T > 16) return 1.0;
  std::vector<std::complex<double>> in(n), ref(n), w(n);
  srand(n * 3 + (int)sizeof(T));
  for (auto& v : in) v = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
  for (int t = 0; t < n; ++t) w[t] = std::polar(1.0, -2 * M_PI * t / n);
  for (int k = 0; k < n; ++k) {
    std::complex<double> s = 0;
    for (int m = 0; m < n; ++m) s += in[m] * w[(long long)m * k % n];
    ref[k] = s;
  }
  std::vector<cpx<T>> W(mr_lds_elems(n)), lds(mr_lds_elems(n)), regs((size_t)MR_REGS * pl.T);
  for (int t = 0; t < n; ++t) W[mr_pad(t)] = {(T)w[t].real(), (T)w[t].imag()};   // padded like the LDS copy
  for (int m = 0; m < n; ++m) lds[mr_pad(m)] = {(T)in[m].real(), (T)in[m].imag()};
  for (int p = 0; p < pl.np; ++p) {
    switch (pl.radix[p]) {
#define CASE(R) case R: half_pass_load<T, R>(regs, lds, pl); half_pass_store<T, R>(regs, lds, pl, p, W); break;
      CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
#undef CASE
      default: return 1.0;
    }
  }
  double err = 0, mag = 0;
  for (int k = 0; k < n; ++k) {
    const cpx<T> v = lds[mr_pad(k)];
    err = std::max(err, std::abs(std::complex<double>(v.x, v.y) - ref[k]));
    mag = std::max(mag, std::abs(ref[k]));
  }
  return err / mag;
}

// chirp-z (mr_dft with blue = 1) restated with the same per-thread passes: x conj(c) -> FFT_L -> * FFT_L(b) / L ->
// conj -> FFT_L -> conj * conj(c), the table FFT_L(b) / L built by the engine itself in double (as the host does at
// plan creation), against the naive DFT of length n
template <class T>
void run_passes(std::vector<cpx<T>>& lds, const MrPlan& pl, const std::vector<cpx<T>>& W) {
  std::vector<cpx<T>> regs((size_t)MR_REGS * pl.T);
  for (int p = 0; p < pl.np; ++p) {
    switch (pl.radix[p]) {
#define CASE(R) case R: half_pass_load<T, R>(regs, lds, pl); half_pass_store<T, R>(regs, lds, pl, p, W); break;
      CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
#undef CASE
    }
  }
}

template <class T>
double test_chirpz(int n) {
  MrDft d;
  if (!mr_make_dft(n, 1 << 20, &d) || !d.blue) { printf("n = %d: expected a chirp-z plan\n", n); return 1.0; }
  const int L = d.pl.n;
  if (L < 2 * n - 1) return 1.0;
  std::vector<cpx<double>> Wd(mr_lds_elems(L)), b(mr_lds_elems(L), cpx<double>{0.0, 0.0});
  std::vector<cpx<T>> W(mr_lds_elems(L));
  for (int k = 0; k < L; ++k) {
    Wd[mr_pad(k)] = {cos(-2 * M_PI * k / L), sin(-2 * M_PI * k / L)};
    W[mr_pad(k)] = {(T)Wd[mr_pad(k)].x, (T)Wd[mr_pad(k)].y};
  }
  std::vector<std::complex<double>> chirp(n);
  for (int m = 0; m < n; ++m) {
    const long long mm = ((long long)m * m) % (2LL * n);
    chirp[m] = std::polar(1.0, M_PI * (double)mm / n);
    b[mr_pad(m)] = {chirp[m].real(), chirp[m].imag()};
    if (m > 0) b[mr_pad(L - m)] = b[mr_pad(m)];
  }
  run_passes<double>(b, d.pl, Wd);   // FFT_L(b)
  std::vector<std::complex<double>> in(n), ref(n);
  srand(n + 11);
  for (auto& v : in) v = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
  for (int k = 0; k < n; ++k) {
    std::complex<double> s = 0;
    for (int m = 0; m < n; ++m) s += in[m] * std::polar(1.0, -2 * M_PI * (double)((long long)m * k % n) / n);
    ref[k] = s;
  }
  std::vector<cpx<T>> lds(mr_lds_elems(L), cpx<T>{T(0), T(0)});
  for (int m = 0; m < n; ++m) {
    const std::complex<double> v = in[m] * std::conj(chirp[m]);
    lds[mr_pad(m)] = {(T)v.real(), (T)v.imag()};
  }
  run_passes<T>(lds, d.pl, W);
  for (int k = 0; k < L; ++k) {
    const cpx<T> bs = {(T)(b[mr_pad(k)].x / L), (T)(b[mr_pad(k)].y / L)};
    const cpx<T> v = cmul(lds[mr_pad(k)], bs);
    lds[mr_pad(k)] = {v.x, -v.y};
  }
  run_passes<T>(lds, d.pl, W);
  double err = 0, mag = 0;
  for (int k = 0; k < n; ++k) {
    const std::complex<double> v = std::conj(std::complex<double>(lds[mr_pad(k)].x, lds[mr_pad(k)].y)) * std::conj(chirp[k]);
    err = std::max(err, std::abs(v - ref[k]));
    mag = std::max(mag, std::abs(ref[k]));
  }
  return err / mag;
}

int main() {
  const int sizes[] = {2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 15, 16, 28, 30, 32, 48, 60, 63, 64, 65, 80, 84, 96, 100, 130, 200, 240, 256, 500,
                       512, 1000, 1040, 1500, 2000, 2112, 2160, 3000, 3003, 4096};
  int bad = 0;
  for (int n : sizes) {
    const double e32 = test_one<float>(n, true), e64 = test_one<double>(n, true);
    MrPlan pl;
    mr_make_plan(n, &pl);
    printf("n = %5d  T %4d  passes", n, pl.T);
    for (int p = 0; p < pl.np; ++p) printf(" %d", pl.radix[p]);
    printf("  rel err f32 %.2e  f64 %.2e\n", e32, e64);
    if (!(e32 < 3e-6) || !(e64 < 2e-14)) ++bad;   // (the naive reference sum itself carries ~sqrt(n) eps)
  }
  // lengths with a prime factor > 13 have no plan (they stay on Bluestein)
  for (int n : {17, 34, 1003, 2047})
    if (test_one<float>(n, false) != 0.0) ++bad;
  // sides with a prime factor > 13: chirp-z on the smallest smooth L >= 2n - 1
  for (int n : {17, 29, 68, 97, 116, 251, 1006, 1392}) {
    const double e32 = test_chirpz<float>(n), e64 = test_chirpz<double>(n);
    MrDft d;
    mr_make_dft(n, 1 << 20, &d);
    printf("chirp-z n = %5d  L = %5d  rel err f32 %.2e  f64 %.2e\n", n, d.pl.n, e32, e64);
    if (!(e32 < 2e-5) || !(e64 < 1e-13)) ++bad;
  }
  printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
