// Host emulator for pygpa_amd/csrc/gpa_dct.h (see fft_emulator.cpp).
// Build: g++ -O2 -std=c++17 -I pygpa_amd/csrc tests/host/dct_emulator.cpp -o /tmp/dct_emu
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gpa_dct.h"

using namespace gpa;

template <class T, int LG, int EE>
struct Emu {
  using F = WgFFT<T, LG, EE>;
  using D = WgDCT<T, LG, EE>;
  static constexpr int N = F::L, TPF = F::TPF, E = EE;
  std::vector<cpx<T>> regs, lds, table, wk;
  std::vector<typename F::Twiddles> tw;
  Emu() : regs(E * TPF), lds(F::LDS_ELEMS), table(N), wk(N), tw(TPF) {
    for (int t = 0; t < N; ++t) {
      table[t] = {(T)std::cos(-2 * M_PI * t / N), (T)std::sin(-2 * M_PI * t / N)};
      wk[t] = {(T)std::cos(-M_PI * t / (2.0 * N)), (T)std::sin(-M_PI * t / (2.0 * N))};
    }
    for (int t = 0; t < TPF; ++t) F::load_twiddles(tw[t], table.data(), t);
  }
  cpx<T> (&R(int t))[E] { return *reinterpret_cast<cpx<T>(*)[E]>(&regs[E * t]); }
  void forward() {
    for (int t = 0; t < TPF; ++t) F::template fwd_phase<0>(R(t), lds.data(), t, tw[t]);
    if constexpr (F::P > 1) for (int t = 0; t < TPF; ++t) F::template fwd_phase<1>(R(t), lds.data(), t, tw[t]);
    if constexpr (F::P > 2) for (int t = 0; t < TPF; ++t) F::template fwd_phase<2>(R(t), lds.data(), t, tw[t]);
    if constexpr (F::P > 3) for (int t = 0; t < TPF; ++t) F::template fwd_phase<3>(R(t), lds.data(), t, tw[t]);
  }
  void inverse() {
    for (int t = 0; t < TPF; ++t) F::template inv_phase<0>(R(t), lds.data(), t, tw[t]);
    if constexpr (F::P > 1) for (int t = 0; t < TPF; ++t) F::template inv_phase<1>(R(t), lds.data(), t, tw[t]);
    if constexpr (F::P > 2) for (int t = 0; t < TPF; ++t) F::template inv_phase<2>(R(t), lds.data(), t, tw[t]);
    if constexpr (F::P > 3) for (int t = 0; t < TPF; ++t) F::template inv_phase<3>(R(t), lds.data(), t, tw[t]);
  }
};

static void naive_dct2(const std::vector<double>& x, std::vector<double>& X) {
  const int N = (int)x.size();
  X.assign(N, 0);
  for (int k = 0; k < N; ++k) {
    double s = 0;
    for (int n = 0; n < N; ++n) s += x[n] * std::cos(M_PI * k * (2 * n + 1) / (2.0 * N));
    X[k] = 2 * s;
  }
}
static void naive_idct2(const std::vector<double>& X, std::vector<double>& x) {
  const int N = (int)X.size();
  x.assign(N, 0);
  for (int n = 0; n < N; ++n) {
    double s = X[0] / 2;
    for (int k = 1; k < N; ++k) s += X[k] * std::cos(M_PI * k * (2 * n + 1) / (2.0 * N));
    x[n] = s / N;
  }
}

template <class T, int LG, int EE = 16>
void test_one() {
  using F = WgFFT<T, LG, EE>;
  using D = WgDCT<T, LG, EE>;
  constexpr int N = F::L, TPF = F::TPF, E = EE;
  Emu<T, LG, EE> e;
  std::vector<double> a(N), b(N), Xa, Xb;
  srand(LG);
  for (int i = 0; i < N; ++i) { a[i] = rand() / (double)RAND_MAX - 0.5; b[i] = rand() / (double)RAND_MAX - 0.3; }
  naive_dct2(a, Xa);
  naive_dct2(b, Xb);
  // ---- forward
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int src = makhoul_src(t + TPF * i, N);
      e.R(t)[i] = {(T)a[src], (T)b[src]};
    }
  e.forward();
  for (int t = 0; t < TPF; ++t) D::fwd_scatter(e.R(t), e.lds.data(), t);
  for (int t = 0; t < TPF; ++t) D::fwd_gather(e.R(t), e.lds.data(), t, e.wk.data());
  double err = 0, nrm = 0;
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int k = t + TPF * i;
      err = std::max(err, std::abs(e.R(t)[i].x - Xa[k]));
      err = std::max(err, std::abs(e.R(t)[i].y - Xb[k]));
      nrm = std::max(nrm, std::abs(Xa[k]));
    }
  double e_fwd = err / nrm;
  // ---- inverse (registers now hold X natural; build the mirror)
  std::vector<cpx<T>> xm(E * TPF);
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int k = t + TPF * i;
      xm[E * t + i] = k == 0 ? cpx<T>{0, 0} : cpx<T>{(T)Xa[N - k], (T)Xb[N - k]};
      e.R(t)[i] = {(T)Xa[k], (T)Xb[k]};
    }
  for (int t = 0; t < TPF; ++t)
    D::inv_prepare(e.R(t), *reinterpret_cast<cpx<T>(*)[E]>(&xm[E * t]), t, e.wk.data());
  e.forward();
  for (int t = 0; t < TPF; ++t) D::inv_scatter(e.R(t), e.lds.data(), t, (T)(1.0 / N));
  for (int t = 0; t < TPF; ++t) D::inv_gather(e.R(t), e.lds.data(), t);
  double e_inv = 0;
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int c = t + TPF * i;
      e_inv = std::max(e_inv, std::abs(e.R(t)[i].x - a[c]));
      e_inv = std::max(e_inv, std::abs(e.R(t)[i].y - b[c]));
    }
  // ---- fused solve: idct(dct(x) / (2 (ca[k] + cb - 2)))
  const double A = N * 1.37;   // deliberately "wrong" denominator like the reference's swapped axes
  std::vector<T> ca(N), cam(N);
  std::vector<cpx<T>> wspec(N);
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int k = F::spec_index(t, i);
      ca[i * TPF + t] = (T)(2 * std::pow(std::sin(M_PI * k / (2 * A)), 2));
      cam[i * TPF + t] = (T)(2 * std::pow(std::sin(M_PI * (N - k) / (2 * A)), 2));
      wspec[i * TPF + t] = e.wk[k];
    }
  const double cba = 1.0, cbb = std::cos(M_PI * 5 / 77.0);   // sequence a sits at the other axis' bin 0
  std::vector<double> Ya(N), Yb(N), ya, yb;
  for (int k = 0; k < N; ++k) {
    const double hk = 2 * std::pow(std::sin(M_PI * k / (2 * A)), 2);   // 1 - cos, without cancellation
    double sa = -2 * (hk + (1 - cba)), sb = -2 * (hk + (1 - cbb));
    if (k == 0) sa = 1;
    Ya[k] = Xa[k] / sa;
    Yb[k] = Xb[k] / sb;
  }
  naive_idct2(Ya, ya);
  naive_idct2(Yb, yb);
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int src = makhoul_src(t + TPF * i, N);
      e.R(t)[i] = {(T)a[src], (T)b[src]};
    }
  e.forward();
  for (int t = 0; t < TPF; ++t) D::solve_scatter(e.R(t), e.lds.data(), t);
  for (int t = 0; t < TPF; ++t)
    D::solve_combine(e.R(t), e.lds.data(), t, wspec.data(), ca.data(), cam.data(), (T)(1.0 - cba), (T)(1.0 - cbb), true, false,
                     (T)(1.0 / N));
  e.inverse();
  double e_sol = 0, n_sol = 0;
  for (int t = 0; t < TPF; ++t)
    for (int i = 0; i < E; ++i) {
      int dst = makhoul_src(t + TPF * i, N);
      e_sol = std::max(e_sol, std::abs(e.R(t)[i].x - ya[dst]));
      e_sol = std::max(e_sol, std::abs(e.R(t)[i].y - yb[dst]));
      n_sol = std::max(n_sol, std::abs(ya[dst]));
    }
  e_sol /= n_sol;
  printf("N=%5d E=%2d %s dct_fwd=%.2e idct=%.2e fused_solve=%.2e\n", N, E, sizeof(T) == 4 ? "f32" : "f64", e_fwd, e_inv, e_sol);
  double tol = sizeof(T) == 4 ? 5e-6 : 1e-11;  // the naive O(N^2) reference itself loses digits
  if (e_fwd > tol || e_inv > tol || e_sol > 20 * tol) { printf("FAIL\n"); exit(1); }
}

int main() {
  test_one<double, 6>(); test_one<double, 7>(); test_one<double, 9>(); test_one<double, 10>(); test_one<double, 12>();
  test_one<float, 6>(); test_one<float, 8>(); test_one<float, 11>(); test_one<float, 13>();
  // eight elements per thread
  test_one<double, 6, 8>(); test_one<double, 8, 8>(); test_one<double, 9, 8>(); test_one<double, 10, 8>(); test_one<double, 11, 8>();
  test_one<float, 7, 8>(); test_one<float, 9, 8>(); test_one<float, 10, 8>(); test_one<float, 12, 8>();
  printf("OK\n");
}
