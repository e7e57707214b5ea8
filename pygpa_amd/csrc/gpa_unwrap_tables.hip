// a7, workspace of the weighted unwrap: buffers, twiddles, the Laplacian eigenvalue tables (phase_unwrap.py:106-115, with
// the reference's swapped-axis quirk as table [1]), Bluestein chirps, mixed-radix plans, and the per-column constants of
// the transform-free column solve.  Host code only.
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

// host-side radix-2 FFT in double, for the Bluestein kernel spectra
void host_fft(std::vector<double>& re, std::vector<double>& im) {
  const size_t n = re.size();
  int lg = 0;
  while ((size_t(1) << lg) < n) ++lg;
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
    if (r > i) { std::swap(re[i], re[r]); std::swap(im[i], im[r]); }
  }
  for (size_t len = 2; len <= n; len <<= 1)
    for (size_t s0 = 0; s0 < n; s0 += len)
      for (size_t j = 0; j < len / 2; ++j) {
        const double ang = -2.0 * M_PI * (double)j / (double)len, wr = cos(ang), wi = sin(ang);
        const size_t a = s0 + j, b = s0 + j + len / 2;
        const double vr = re[b] * wr - im[b] * wi, vi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - vr; im[b] = im[a] - vi;
        re[a] += vr; im[a] += vi;
      }
}

template <class T>
hipError_t upload_vec(void** dst, const std::vector<double>& v, size_t* bytes, hipStream_t s) {
  std::vector<T> tmp(v.begin(), v.end());
  hipError_t e = hipMalloc(dst, tmp.size() * sizeof(T) + 16);
  if (e != hipSuccess) return e;
  *bytes += tmp.size() * sizeof(T);
  e = hipMemcpyAsync(*dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}
hipError_t upload(int dtype, void** dst, const std::vector<double>& v, size_t* bytes, hipStream_t s) {
  return dtype == 0 ? upload_vec<float>(dst, v, bytes, s) : upload_vec<double>(dst, v, bytes, s);
}

int ilog2_exact(int n) {
  int lg = 0;
  while ((1 << lg) < n) ++lg;
  return (1 << lg) == n ? lg : -1;
}

}  // namespace

// transform-free column solve (colsolve_tri_kernel): per row frequency j the decay lam_j of the Green's function of
// (T + mu_j) and the constants of its boundary terms, in long double.  lamN carries the launch geometry's padding
// (lam^(N - pad), see the kernel); no table if the column does not fit one workgroup.
hipError_t build_tritab(Impl* w, hipStream_t s, size_t* bytes) {
  const int n0 = w->n0, n1 = w->n1;
  int Q, S, pad;
  const int R = tri_rows(w->rsz, n0);
  if (w->dtype == 0) tri_geometry<float>(n0, n1, w->generic, R, &Q, &S, &pad);
  else tri_geometry<double>(n0, n1, w->generic, R, &Q, &S, &pad);
  if (S * Q > 1024 || n1 % (w->dtype == 0 ? 4 : 2)) return hipSuccess;   // (16-byte column vectors)
  w->triQ = Q;
  w->triS = S;
  w->triR = R;
  std::vector<TriCol> tc((size_t)n1);
  for (int j = 0; j < n1; ++j) {
    if (j == 0) { tc[0] = {1.0, 1.0, 1.0, 0.0, 0.0}; continue; }
    const long double sj = sinl((long double)M_PI * j / (2.0L * n1)), h = 2 * sj * sj;
    const long double lam = (1 + h) - sqrtl(h * (2 + h));
    tc[j].lam = (double)lam;
    tc[j].lamR = (double)powl(lam, R);
    tc[j].lamN = (double)powl(lam, (long double)(n0 - pad));
    tc[j].inv = (double)(1.0L / (1.0L - powl(lam, 2.0L * n0)));
    tc[j].zn = (double)(-lam / (1.0L - lam));
  }
  hipError_t e = hipMalloc(&w->tritab, tc.size() * sizeof(TriCol));
  if (e != hipSuccess) return e;
  *bytes += tc.size() * sizeof(TriCol);
  e = hipMemcpyAsync(w->tritab, tc.data(), tc.size() * sizeof(TriCol), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return e;
}

hipError_t unwrap_workspace_create(int dtype, int n0, int n1, hipStream_t s, UnwrapWorkspace* ws, size_t* bytes_out,
                                   int nprob) {
  Impl* w = new Impl();
  memset(w, 0, sizeof(Impl));
  ws->impl = w;
  ws->dtype = dtype;
  ws->n0 = n0;
  ws->n1 = n1;
  w->dtype = dtype;
  w->n0 = n0;
  w->n1 = n1;
  w->nprob = nprob < 1 ? 1 : nprob;
  w->cap = w->nprob;
  w->rsz = dtype == 0 ? 4 : 8;
  w->lg0 = ilog2_exact(n0);
  w->lg1 = ilog2_exact(n1);
  const int maxlg = dtype == 0 ? 14 : 13;
  const bool pow2ok = w->lg0 >= 6 && w->lg1 >= 6 && w->lg0 <= maxlg && w->lg1 <= maxlg;
  auto blue_lg = [](int n) { int lg = 6; while ((1 << lg) < 2 * n - 1) ++lg; return lg; };
  w->lgb0 = blue_lg(n0);
  w->lgb1 = blue_lg(n1);
  w->generic = !pow2ok;
  w->supported = pow2ok || (n0 >= 2 && n1 >= 2 && w->lgb0 <= maxlg && w->lgb1 <= maxlg);
  size_t bytes = 0;
  const size_t npx = (size_t)n0 * n1;
  hipError_t e;
  void** arrs[] = {&w->r, &w->p, &w->p2, &w->q, &w->z};
  for (void** a : arrs) {
    e = hipMalloc(a, npx * w->rsz * w->cap);
    if (e != hipSuccess) return e;
    bytes += npx * w->rsz * w->cap;
  }
  w->ring[0] = w->p;
  w->ring[1] = w->p2;
  w->nring = 2;
  e = hipMalloc((void**)&w->scal, (size_t)SCAL_N * w->cap * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&w->flags, (size_t)FLAGS_N * w->cap * sizeof(int));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&w->part, PART_N * w->cap * sizeof(double));
  if (e != hipSuccess) return e;
  if (w->supported && w->generic) {
    for (int ax = 0; ax < 2; ++ax) {
      const int n = ax == 0 ? n0 : n1, lgb = ax == 0 ? w->lgb0 : w->lgb1, L = 1 << lgb, tpf = L / 16;
      std::vector<double> t((size_t)2 * L), ch((size_t)2 * n), wkv((size_t)2 * n);
      for (int k = 0; k < L; ++k) { t[2 * k] = cos(-2.0 * M_PI * k / L); t[2 * k + 1] = sin(-2.0 * M_PI * k / L); }
      std::vector<double> bre((size_t)L, 0.0), bim((size_t)L, 0.0);
      for (int m = 0; m < n; ++m) {
        const long long mm = ((long long)m * m) % (2LL * n);     // c_m = exp(i pi m^2 / n), argument reduced exactly
        const double cr = cos(M_PI * (double)mm / n), ci = sin(M_PI * (double)mm / n);
        ch[2 * m] = cr; ch[2 * m + 1] = ci;
        bre[m] = cr; bim[m] = ci;
        if (m > 0) { bre[L - m] = cr; bim[L - m] = ci; }
        wkv[2 * m] = cos(-M_PI * m / (2.0 * n)); wkv[2 * m + 1] = sin(-M_PI * m / (2.0 * n));
      }
      host_fft(bre, bim);
      std::vector<double> bs((size_t)2 * L);
      for (int i = 0; i < 16; ++i)
        for (int tt = 0; tt < tpf; ++tt) {
          const int k = spec_index_rt(lgb, tt, i);
          bs[2 * ((size_t)i * tpf + tt)] = bre[k] / L;
          bs[2 * ((size_t)i * tpf + tt) + 1] = bim[k] / L;
        }
      if ((e = upload(dtype, ax == 0 ? &w->btw0 : &w->btw1, t, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, ax == 0 ? &w->chirp0 : &w->chirp1, ch, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, ax == 0 ? &w->bspec0 : &w->bspec1, bs, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, ax == 0 ? &w->gwk0 : &w->gwk1, wkv, &bytes, s)) != hipSuccess) return e;
    }
    // fused path on the mixed-radix engine: rows of whole 4-pixel vectors (pq_kernel) and, per axis, a transform that
    // fits LDS -- length n itself when it is smooth, chirp-z on the smallest smooth L >= 2n - 1 otherwise
    {
      const int max_elems = (int)((size_t)159 * 1024 / (2 * w->rsz));
      w->mr_ok = !opt_set(OPT_NO_MR) && mr_make_dft(n0, max_elems, &w->mr0) &&
                 mr_make_dft(n1, max_elems, &w->mr1);
      if (opt_set(OPT_MR_FORCE_BLUESTEIN) && w->mr_ok) {   // diagnostic / tests: chirp-z also for smooth lengths
        for (MrDft* d : {&w->mr0, &w->mr1}) {
          if (d->blue) continue;
          MrDft t = *d;
          t.blue = 1;
          bool found = false;
          for (int L = 2 * d->n - 1; mr_lds_elems(L) <= max_elems && !found; ++L) found = mr_make_plan(L, &t.pl);
          if (found) *d = t; else w->mr_ok = false;
        }
      }
    }
    if (w->mr_ok) {
      for (int ax = 0; ax < 2; ++ax) {
        const MrDft& d = ax == 0 ? w->mr0 : w->mr1;
        const int L = d.pl.n;
        std::vector<double> t((size_t)2 * mr_lds_elems(L), 0.0);   // entry k at mr_pad(k), see mr_store()
        for (int k = 0; k < L; ++k) {
          t[2 * (size_t)mr_pad(k)] = cos(-2.0 * M_PI * k / L);
          t[2 * (size_t)mr_pad(k) + 1] = sin(-2.0 * M_PI * k / L);
        }
        if ((e = upload(dtype, ax == 0 ? &w->mrW0 : &w->mrW1, t, &bytes, s)) != hipSuccess) return e;
        if (d.blue) {
          // FFT_L(b) / L, b[m] = b[L - m] = exp(i pi m^2 / n), by the engine's own passes in double on the host
          std::vector<cpx<double>> W((size_t)mr_lds_elems(L)), img((size_t)mr_lds_elems(L), cpx<double>{0.0, 0.0});
          std::vector<cpx<double>> regs((size_t)MR_REGS * d.pl.T);
          for (int k = 0; k < L; ++k) W[mr_pad(k)] = {t[2 * (size_t)mr_pad(k)], t[2 * (size_t)mr_pad(k) + 1]};
          for (int m = 0; m < d.n; ++m) {
            const long long mm = ((long long)m * m) % (2LL * d.n);
            const cpx<double> c = {cos(M_PI * (double)mm / d.n), sin(M_PI * (double)mm / d.n)};
            img[mr_pad(m)] = c;
            if (m > 0) img[mr_pad(L - m)] = c;
          }
          for (int p = 0; p < d.pl.np; ++p) {
#define GPA_HOST_PASS(R)                                                                                              \
  case R:                                                                                                             \
    for (int tt = 0; tt < d.pl.T; ++tt)                                                                               \
      mr_load<double, R>(&regs[(size_t)MR_REGS * tt], reinterpret_cast<const double*>(img.data()), L, tt, d.pl.T);    \
    for (int tt = 0; tt < d.pl.T; ++tt)                                                                               \
      mr_store<double, R>(&regs[(size_t)MR_REGS * tt], reinterpret_cast<double*>(img.data()), L, d.pl.stride[p],      \
                          d.pl.magic[p], tt, d.pl.T, reinterpret_cast<const double*>(W.data()));                      \
    break;
            switch (d.pl.radix[p]) {
              GPA_HOST_PASS(2) GPA_HOST_PASS(3) GPA_HOST_PASS(4) GPA_HOST_PASS(5) GPA_HOST_PASS(6) GPA_HOST_PASS(7)
              GPA_HOST_PASS(8) GPA_HOST_PASS(10) GPA_HOST_PASS(11) GPA_HOST_PASS(12) GPA_HOST_PASS(13) GPA_HOST_PASS(14)
              GPA_HOST_PASS(15) GPA_HOST_PASS(16)
            }
#undef GPA_HOST_PASS
          }
          std::vector<double> bs((size_t)2 * L);
          for (int k = 0; k < L; ++k) { bs[2 * k] = img[mr_pad(k)].x / L; bs[2 * k + 1] = img[mr_pad(k)].y / L; }
          if ((e = upload(dtype, ax == 0 ? &w->mrB0 : &w->mrB1, bs, &bytes, s)) != hipSuccess) return e;
        }
      }
    }
    for (int compat = 0; compat < 2; ++compat) {
      const double A0 = compat ? n1 : n0, A1 = compat ? n0 : n1;
      std::vector<double> a((size_t)n0), am((size_t)n0), b((size_t)n1);
      for (int k = 0; k < n0; ++k) {
        const double sk = sin(M_PI * k / (2.0 * A0)), sm = sin(M_PI * (n0 - k) / (2.0 * A0));
        a[k] = 2 * sk * sk;
        am[k] = 2 * sm * sm;
      }
      for (int j = 0; j < n1; ++j) { const double sj = sin(M_PI * j / (2.0 * A1)); b[j] = 2 * sj * sj; }
      if ((e = upload(dtype, &w->gha0[compat], a, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->gham0[compat], am, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->hb1[compat], b, &bytes, s)) != hipSuccess) return e;
    }
  }
  if (w->supported && !w->generic) {
    for (int ax = 0; ax < 2; ++ax) {
      const int n = ax == 0 ? n0 : n1;
      std::vector<double> t((size_t)2 * n);
      for (int k = 0; k < n; ++k) {
        t[2 * k] = cos(-2.0 * M_PI * k / n);
        t[2 * k + 1] = sin(-2.0 * M_PI * k / n);
      }
      e = upload(dtype, ax == 0 ? &w->tw0 : &w->tw1, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    if (w->lg1 >= 12) {   // (rows of 4096 points can be switched to the half-length kernels for measurements: ROWHALF_MINLG)
      const int h = n1 / 2;
      std::vector<double> t((size_t)2 * h);
      for (int k = 0; k < h; ++k) {
        t[2 * k] = cos(-2.0 * M_PI * k / h);
        t[2 * k + 1] = sin(-2.0 * M_PI * k / h);
      }
      e = upload(dtype, &w->tw1h, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    {
      std::vector<double> t((size_t)2 * n1);
      for (int k = 0; k < n1; ++k) {
        t[2 * k] = cos(-M_PI * k / (2.0 * n1));
        t[2 * k + 1] = sin(-M_PI * k / (2.0 * n1));
      }
      e = upload(dtype, &w->wk1, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    // tables of the column kernel in the spectral layout [register][thread] of ITS transform (unwrap_elems)
    const int E0 = unwrap_elems(w->lg0, (size_t)w->rsz), tpf0 = n0 / E0;
    {
      std::vector<double> t((size_t)2 * n0);
      for (int i = 0; i < E0; ++i)
        for (int tt = 0; tt < tpf0; ++tt) {
          const int k = spec_index_rt(w->lg0, tt, i, E0);
          t[2 * ((size_t)i * tpf0 + tt)] = cos(-M_PI * k / (2.0 * n0));
          t[2 * ((size_t)i * tpf0 + tt) + 1] = sin(-M_PI * k / (2.0 * n0));
        }
      e = upload(dtype, &w->wk0s, t, &bytes, s);
      if (e != hipSuccess) return e;
    }
    // 1 - cos(pi i / A) = 2 sin^2(pi i / (2A)).  Reference (compat = 1): axis-0 bins use A = n1
    // and axis-1 bins use A = n0 (phase_unwrap.py:107-109); compat = 0: A = own axis length.
    for (int compat = 0; compat < 2; ++compat) {
      const double A0 = compat ? n1 : n0, A1 = compat ? n0 : n1;
      std::vector<double> a((size_t)n0), am((size_t)n0), b((size_t)n1);
      for (int i = 0; i < E0; ++i)
        for (int tt = 0; tt < tpf0; ++tt) {
          const int k = spec_index_rt(w->lg0, tt, i, E0);
          const double sk = sin(M_PI * k / (2.0 * A0)), sm = sin(M_PI * (n0 - k) / (2.0 * A0));
          a[(size_t)i * tpf0 + tt] = 2 * sk * sk;
          am[(size_t)i * tpf0 + tt] = 2 * sm * sm;
        }
      for (int j = 0; j < n1; ++j) {
        const double sj = sin(M_PI * j / (2.0 * A1));
        b[j] = 2 * sj * sj;
      }
      if ((e = upload(dtype, &w->ha0[compat], a, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->ham0[compat], am, &bytes, s)) != hipSuccess) return e;
      if ((e = upload(dtype, &w->hb1[compat], b, &bytes, s)) != hipSuccess) return e;
    }
    if (n0 == n1 && (e = build_tritab(w, s, &bytes)) != hipSuccess) return e;
    if (n0 == n1 && (e = build_streamtab(w, s, &bytes)) != hipSuccess) return e;
  }
  if (w->supported && w->generic && w->mr_ok && n0 == n1 && (e = build_tritab(w, s, &bytes)) != hipSuccess) return e;
  if (w->supported && w->generic && w->mr_ok && n0 == n1 && (n1 % 4) == 0 && (e = build_streamtab(w, s, &bytes)) != hipSuccess) return e;
  if (bytes_out) *bytes_out = bytes;
  return hipSuccess;
}

bool unwrap_supports_batch(const UnwrapWorkspace* ws) {
  const Impl* w = (const Impl*)ws->impl;
  return w && w->supported && (!w->generic || w->mr_ok);
}

void unwrap_workspace_destroy(UnwrapWorkspace* ws) {
  Impl* w = (Impl*)ws->impl;
  if (!w) return;
  void* bufs[] = {w->r, w->p, w->p2, w->q, w->z, w->tw0, w->tw1, w->wk1, w->wk0s, w->ha0[0], w->ha0[1], w->ham0[0],
                  w->ham0[1], w->hb1[0], w->hb1[1], w->scal, w->flags, w->part, w->btw0, w->btw1, w->chirp0, w->chirp1,
                  w->bspec0, w->bspec1, w->gwk0, w->gwk1, w->gha0[0], w->gha0[1], w->gham0[0], w->gham0[1], w->tritab,
                  w->mrW0, w->mrW1, w->mrB0, w->mrB1, w->strtab, w->strlam, w->stragg, w->strcar, w->tw1h};
  for (void* b : bufs)
    if (b) hipFree(b);
  for (int j = 2; j < w->nring; ++j)
    if (w->ring[j]) hipFree(w->ring[j]);
  delete w;
  ws->impl = nullptr;
}
}  // namespace gpa
