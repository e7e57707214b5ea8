// Host-side tables of a plan (double precision): the circular Gaussian filters of a sigma in the plan's geometry, the
// compact extension of non-power-of-two axes, the shared-forward pass B tables of a staged k-list, the carrier / compensation
// tables of the staged candidates, and the buffers that grow with them.  Split off gpa_api.hip in round 5.
#include "gpa_plan.h"

// ---------------------------------------------------------------------------
// host-side table construction (double precision)
// ---------------------------------------------------------------------------
void host_fft_pow2(std::vector<std::complex<double>>& a, bool inverse) {
  const size_t n = a.size();
  int lg = 0;
  while ((size_t(1) << lg) < n) ++lg;
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
    if (r > i) std::swap(a[i], a[r]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = (inverse ? 2 : -2) * M_PI / (double)len;
    for (size_t s = 0; s < n; s += len)
      for (size_t j = 0; j < len / 2; ++j) {
        std::complex<double> w(cos(ang * (double)j), sin(ang * (double)j));
        auto u = a[s + j], v = a[s + j + len / 2] * w;
        a[s + j] = u + v;
        a[s + j + len / 2] = u - v;
      }
  }
}

// 1-D factor of scipy.ndimage.fourier_gaussian (called at
// geometric_phase_analysis.py:44/:75/:87, cuGPA.py:57): exp(-2 pi^2 sigma^2 f^2),
// f = fftfreq(n), flushed to 0 where the exponent exceeds 50 (SciPy does that per axis).
std::vector<double> gaussian_kspace(int n, double sigma) {
  std::vector<double> g(n);
  for (int k = 0; k < n; ++k) {
    const int kk = k < (n + 1) / 2 ? k : k - n;   // fftfreq ordering
    const double f = (double)kk / (double)n;
    const double e = 2.0 * M_PI * M_PI * sigma * sigma * f * f;
    g[k] = e > 50.0 ? 0.0 : exp(-e);
  }
  return g;
}

// spatial kernel h[m] = (1/n) sum_k g[k] cos(2 pi k m / n), m = 0 .. n-1 (g is even): the taps of the circular filter
std::vector<double> spatial_kernel(int n, const std::vector<double>& g) {
  // accumulated in long double (64-bit mantissa) so that the small taps are those of g as given, not summation noise
  std::vector<double> h((size_t)n);
  std::vector<long double> cs((size_t)n);
  const long double w = 2.0L * 3.14159265358979323846264338327950288L / (long double)n;
  for (int j = 0; j < n; ++j) cs[j] = cosl(w * (long double)j);
  for (int m = 0; m < n; ++m) {
    long double acc = 0;
    for (int k = 0; k < n; ++k)
      if (g[k] != 0.0) acc += (long double)g[k] * cs[(size_t)((long long)k * m % n)];
    h[m] = (double)(acc / (long double)n);
  }
  return h;
}

// smallest E such that the taps at circular distance > E from lag 0 sum (in magnitude) to less than tol times the
// sum of all taps: dropping them changes a filtered value by at most tol * max|input| * sum|h|, a guaranteed bound.
// The taps of a Gaussian filter fall to the rounding floor of its k-space samples (~1e-17 of the central tap each,
// the transform of the rounding errors of g) within ~9 sigma; beyond that they are noise the reference's own FFT
// does not resolve either.
int kernel_support(const std::vector<double>& h, double tol) {
  const int n = (int)h.size();
  double total = 0;
  for (double v : h) total += fabs(v);
  double tail = 0;
  for (int m = n / 2; m >= 1; --m) {
    tail += fabs(h[m]) + (n - m != m ? fabs(h[n - m]) : 0.0);
    if (tail > tol * total) return m;
  }
  return 0;
}

// taps h[0 .. mmax] only (the shared-forward pass B needs the first few sigma of them, not all n): same sums as
// spatial_kernel
std::vector<double> spatial_taps(int n, const std::vector<double>& g, int mmax) {
  std::vector<double> h((size_t)mmax + 1);
  std::vector<long double> cs((size_t)n);
  const long double w = 2.0L * 3.14159265358979323846264338327950288L / (long double)n;
  for (int j = 0; j < n; ++j) cs[j] = cosl(w * (long double)j);
  for (int m = 0; m <= mmax; ++m) {
    long double acc = 0;
    for (int k = 0; k < n; ++k)
      if (g[k] != 0.0) acc += (long double)g[k] * cs[(size_t)((long long)k * m % n)];
    h[m] = (double)(acc / (long double)n);
  }
  return h;
}

// Filter table of one axis in the spectral register layout [reg][thread]:
// periodic mode -> real g[k]/L; padded mode -> complex DFT_L(h~)/L with h~ the spatial kernel laid out at the lags
// the axis' extension covers (full: lags 0 .. n-1; compact: lags -extL .. extR around slot 0).
void build_filter_table(const Axis& ax, const std::vector<double>& g, const std::vector<double>& hsp,
                               std::vector<double>& out) {
  const int L = ax.L, tpf = L / 16;
  if (!ax.padded) {
    out.assign((size_t)L, 0.0);
    for (int i = 0; i < 16; ++i)
      for (int t = 0; t < tpf; ++t) out[(size_t)i * tpf + t] = g[spec_index_rt(ax.lg, t, i)] / (double)L;
    return;
  }
  const int n = ax.n;
  std::vector<std::complex<double>> h((size_t)L, 0.0);
  if (ax.extR == 0) {
    for (int m = 0; m < n; ++m) h[m] = hsp[m];
  } else {
    for (int m = 0; m <= ax.extL; ++m) h[m] = hsp[m];                    // lags 0 .. E: samples to the left
    for (int m = 1; m <= ax.extR; ++m) h[(size_t)L - m] = hsp[n - m];     // lags -1 .. -E: samples to the right
  }
  host_fft_pow2(h, false);
  out.assign((size_t)2 * L, 0.0);
  for (int i = 0; i < 16; ++i)
    for (int t = 0; t < tpf; ++t) {
      auto v = h[spec_index_rt(ax.lg, t, i)] / (double)L;
      out[2 * ((size_t)i * tpf + t)] = v.real();
      out[2 * ((size_t)i * tpf + t) + 1] = v.imag();
    }
}

int ensure_filters(gpa_plan* p, double sigma) {
  if (!(sigma > 0)) return fail(GPA_ERR_ARG, "sigma must be positive");
  NEED_UNWRAP(p, "lock-in sweep");
  if (sigma == p->sigma_cached) return GPA_OK;
  HIP_TRY(hipStreamSynchronize(p->stream));   // the tables may still be read by an earlier asynchronous call
  p->sigma_cached = -1.0;   // a failure below must not leave half-switched tables behind a matching sigma
  for (int axis = 0; axis < 2; ++axis) {
    const Axis& full = axis == 0 ? p->ax0_full : p->ax1_full;
    Axis& cur = axis == 0 ? p->ax0 : p->ax1;
    std::vector<double> g = gaussian_kspace(full.n, sigma), hsp, table;
    Axis want = full;
    if (full.padded) {
      hsp = spatial_kernel(full.n, g);
      if (!p->no_compact) want = compact_axis(full, kernel_support(hsp, p->dtype == 0 ? 1e-9 : 1e-14));
    }
    if (want.lg != cur.lg || want.extR != cur.extR || want.extL != cur.extL) {
      // another transform length for this sigma: twiddles of that length, and the carrier tables (laid out per
      // L / 16 threads) have to be staged again
      TRY(upload_twiddles(p, axis == 0 ? p->tw0 : p->tw1, want.L));
      p->staged_kl.clear();
      p->staged_kr.clear();
      cur = want;
    }
    build_filter_table(cur, g, hsp, table);
    TRY(upload_real_table(p, axis == 0 ? p->Hx : p->Hy, table));
    if (axis == 1) {
      // shared-forward pass B: the taps of this axis' filter out to where they are rounding noise, the support E
      // beyond which they are dropped from the end fix (the same criterion as the compact extension above)
      p->sh_ok = false;
      ++p->sh_epoch;
      const int n = cur.n;
      int mmax = (int)ceil(10.0 * sigma) + 16;
      if (p->use_shared && mmax < n / 2 && mmax <= 1024) {
        std::vector<double> taps = spatial_taps(n, g, mmax);
        double total = fabs(taps[0]), tail = 0;
        for (int m = 1; m <= mmax; ++m) total += 2 * fabs(taps[m]);
        int E = 1;
        const double tol = p->dtype == 0 ? 1e-9 : 1e-14;
        for (int m = mmax; m >= 1; --m) {
          tail += 2 * fabs(taps[m]);
          if (tail > tol * total) { E = m; break; }
        }
        // a row that is not a power of two long is zero-padded to the next power of two >= n + E: the shared kernel
        // needs no periodic extension (its end fix supplies every wrapped pair), only room for the filter's reach
        Axis sa = cur;
        if (cur.padded) {
          sa.lg = 6;
          while ((1 << sa.lg) < n + E) ++sa.lg;
          sa.L = 1 << sa.lg;
          sa.extL = sa.extR = 0;
        }
        if (passB_shared_supports(p->dtype, sa, E)) {
          const int Epad = (E + 15) & ~15;
          if (sa.L != cur.L && p->tw1s_L != sa.L) {
            if (!p->tw1s) TRY(dmalloc(p, &p->tw1s, (size_t)8192 * p->csz));   // (the longest row transform of the shared kernel)
            TRY(upload_twiddles(p, p->tw1s, sa.L));
            p->tw1s_L = sa.L;
          }
          p->ax1s = sa;
          if (!p->d_taps) TRY(dmalloc(p, (void**)&p->d_taps, 1025 * sizeof(double)));
          if (!p->sh.gtab) TRY(dmalloc(p, &p->sh.gtab, (2 * 256 + 16) * p->rsz));
          HIP_TRY(hipMemcpyAsync(p->d_taps, taps.data(), ((size_t)mmax + 1) * sizeof(double), hipMemcpyHostToDevice, p->stream));
          HIP_TRY(hipStreamSynchronize(p->stream));
          std::vector<double> gt((size_t)2 * Epad + 16, 0.0);
          for (int m = 1; m <= E; ++m) gt[m] = taps[m];
          TRY(upload_real_table(p, p->sh.gtab, gt));
          p->sh_etab = mmax;
          p->sh_E = E;
          p->sh_Epad = Epad;
          p->sh_elems = passB_shared_elems(p->dtype, sa);
          p->sh_sigma = sigma;
          p->sh_ok = true;
        }
      }
    }
  }
  p->sigma_cached = sigma;
  return GPA_OK;
}

// candidate tables of the shared-forward pass B for the staged k-list (P peaks of K candidates): rebuilt when sigma
// or the list changed.  Leaves p->sh_use = whether pass B should take that kernel for this (P, K).
int shared_prepare(gpa_plan* p, int P, int K) {
  p->sh_use = false;
  const int B = P * K;
  if (!p->sh_ok || !p->use_shared || K < 2 || (int)p->staged_planeof.size() < B) return GPA_OK;
  const bool reorder = !opt_set(OPT_NO_REORDER);
  if (p->sh_built_epoch == p->sh_epoch && p->sh_built_K == K && p->sh_built_B == B && p->sh_built_reorder == reorder) {
    p->sh_use = p->sh_built_ok;
    return GPA_OK;
  }
  p->sh_built_reorder = reorder;
  // (the cache key is committed only when the tables are complete: a failed allocation below must not leave a key
  //  that sends the next call to the kernel with freed tables -- ADVICE r03)
  p->sh_built_epoch = -1;
  p->sh_built_ok = false;
  // Visiting order of every peak's candidates.  The selection rule is "strictly larger |sf| replaces, in LIST order"
  // (geometric_phase_analysis.py:679-684) = the first maximum of the list wins.  The kernel stores a winner the moment
  // it wins, so the order in which it visits the candidates sets how often a pixel is rewritten: in list order the
  // amplitude climbs towards the grid's centre (4.3 stores per pixel at configs[2]); visiting the candidates nearest
  // the reference vector first, most later candidates win nowhere in a wavefront and their stores are skipped.  The
  // candidates of one x-plane stay together (they share the forward transform; a list that interleaves the planes gains
  // its runs here), planes ordered by their nearest candidate, candidates within a plane by distance, ties by list
  // position (a stable order: duplicates of a k-vector keep the list's order, so "first maximum" still holds for
  // them; the reported kidx is the original list position).  Two DIFFERENT candidates whose amplitudes agree bit for
  // bit at a pixel may now resolve the other way -- the amplitude ties the tests already allow for.
  // One kind of tie between different candidates is structural, not accidental: k-vectors that differ by WHOLE cycles per pixel
  // sample the same carrier on the integer grid, and where the products k x are exact (dyadic k) the device's lock-ins are
  // equal bit for bit at every pixel.  A peak whose list holds such a pair keeps the LIST order, so that the earlier of
  // the two wins everywhere as the reference's rule says (round 6; tests/test_gpu_shared_passb.py::test_aliased_candidates_*).
  std::vector<int> order((size_t)B);
  for (int pp = 0; pp < P; ++pp) {
    bool aliased = false;
    for (int a = 0; a < K && !aliased; ++a)
      for (int b2 = a + 1; b2 < K && !aliased; ++b2) {
        const double *ka = &p->staged_kl[2 * ((size_t)pp * K + a)], *kb = &p->staged_kl[2 * ((size_t)pp * K + b2)];
        const bool same = ka[0] == kb[0] && ka[1] == kb[1];
        aliased = !same && ka[0] - floor(ka[0]) == kb[0] - floor(kb[0]) && ka[1] - floor(ka[1]) == kb[1] - floor(kb[1]);
      }
    std::vector<double> d2((size_t)K), pmin;
    std::vector<int> pfirst;
    int nplanes = 0;
    for (int k = 0; k < K; ++k) nplanes = std::max(nplanes, p->staged_planeof[pp * K + k] + 1);
    pmin.assign((size_t)nplanes, 1e300);
    pfirst.assign((size_t)nplanes, K);
    for (int k = 0; k < K; ++k) {
      const size_t b = (size_t)pp * K + k;
      const double ex = p->staged_kl[2 * b] - p->staged_kr[2 * b], ey = p->staged_kl[2 * b + 1] - p->staged_kr[2 * b + 1];
      d2[k] = ex * ex + ey * ey;
      const int pl = p->staged_planeof[b];
      pmin[pl] = std::min(pmin[pl], d2[k]);
      pfirst[pl] = std::min(pfirst[pl], k);
    }
    std::vector<int> idx((size_t)K);
    for (int k = 0; k < K; ++k) idx[k] = k;
    if (reorder && !aliased)
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b2) {
        const int pa = p->staged_planeof[pp * K + a], pb = p->staged_planeof[pp * K + b2];
        if (pa != pb) return pmin[pa] != pmin[pb] ? pmin[pa] < pmin[pb] : pfirst[pa] < pfirst[pb];
        return d2[a] < d2[b2];
      });
    for (int k = 0; k < K; ++k) order[(size_t)pp * K + k] = pp * K + idx[k];
  }
  // runs of candidates on one x-plane, in visiting order; chunks of <= NC candidates per matrix pass
  const int NC = p->dtype == 0 ? 4 : 2;
  std::vector<int> desc((size_t)B, 0);
  int runs = 0;
  for (int pp = 0; pp < P; ++pp) {
    int k = 0, chunk = 0;
    while (k < K) {
      int e = k + 1;
      while (e < K && p->staged_planeof[order[pp * K + e]] == p->staged_planeof[order[pp * K + k]]) ++e;
      ++runs;
      for (int j = k; j < e; ++j) {
        const int r = j - k, slot = r % NC;
        int d = slot << 2;
        if (r == 0) d |= 1;
        if (slot == 0) { d |= 2 | (std::min(NC, e - j) << 4); ++chunk; }
        d |= (chunk & 1) << 7;   // parity of the chunk (double-buffered staging of its phasors)
        desc[(size_t)pp * K + j] = d;
      }
      k = e;
    }
  }
  if (2 * runs > B) {   // fewer than two candidates per forward transform on average: nothing to share
    p->sh_built_epoch = p->sh_epoch;
    p->sh_built_K = K;
    p->sh_built_B = B;
    return GPA_OK;
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  // Band rotation.  The shifted Gaussian of a candidate is negligible (below 1e-9 of its peak in f32, 1e-17 in f64)
  // outside |f + wy| < fc; over the candidates of a peak the live band is (-wy_max - fc, -wy_min + fc).  Rotating the
  // row by exp(-2 pi i s y / 16) and the candidates to wy + s / 16 moves that band down by s blocks of L / 16 bins:
  // s = the block the band starts in, so that it occupies blocks 0 .. need-1 -- the spectral registers the kernel keeps.
  const int EEs = p->sh_elems;
  const double fc = sqrt(log(p->dtype == 0 ? 1e9 : 1e17) / (2.0 * M_PI * M_PI * p->sh_sigma * p->sh_sigma));
  std::vector<int> shifts((size_t)P, 0);
  std::vector<double> wys((size_t)B);
  int need = 1;
  for (int pp = 0; pp < P; ++pp) {
    double wmin = p->staged_kl[2 * ((size_t)pp * K) + 1], wmax = wmin;
    for (int k = 1; k < K; ++k) {
      const double w = p->staged_kl[2 * ((size_t)pp * K + k) + 1];
      wmin = std::min(wmin, w);
      wmax = std::max(wmax, w);
    }
    const double lo = -wmax - fc, width = (wmax - wmin) + 2 * fc;
    const double flo = (lo - floor(lo)) * EEs;             // start of the band in blocks, in [0, 16)
    const int sft = (int)floor(flo) % EEs;
    const int blocks = width >= 1.0 ? EEs : (int)ceil((flo - floor(flo)) + width * EEs + 1e-9);
    shifts[pp] = blocks >= EEs ? 0 : sft;
    need = std::max(need, std::min(blocks, EEs));
    for (int k = 0; k < K; ++k)
      wys[(size_t)pp * K + k] = p->staged_kl[2 * (size_t)order[(size_t)pp * K + k] + 1] + (double)shifts[pp] / EEs;
  }
  p->sh_nbl = opt_set(OPT_PBS_FULLBAND) ? EEs : passB_shared_nbl(p->dtype, need);
  if (p->sh_nbl >= EEs) {   // nothing to gain: no rotation
    for (int pp = 0; pp < P; ++pp) shifts[pp] = 0;
    for (int bq = 0; bq < B; ++bq) wys[bq] = p->staged_kl[2 * (size_t)order[bq] + 1];
    p->sh_nbl = EEs;
  }
  const size_t gb = (size_t)B * p->ax1s.L * p->rsz, ps = (size_t)B * p->sh_Epad * p->csz;
  if (gb > p->sh_gb_bytes) {
    if (p->sh.Gb) { (void)hipFree(p->sh.Gb); p->ws_bytes -= p->sh_gb_bytes; p->sh.Gb = nullptr; p->sh_gb_bytes = 0; }
    TRY(dmalloc(p, &p->sh.Gb, gb));
    p->sh_gb_bytes = gb;
  }
  if (ps > p->sh_psi_bytes) {
    if (p->sh.psi) { (void)hipFree(p->sh.psi); (void)hipFree(p->sh.pre); p->ws_bytes -= 2 * p->sh_psi_bytes; p->sh.psi = p->sh.pre = nullptr; p->sh_psi_bytes = 0; }
    TRY(dmalloc(p, &p->sh.psi, ps));
    TRY(dmalloc(p, &p->sh.pre, ps));
    p->sh_psi_bytes = ps;
  }
  if (!p->sh.rot16) TRY(dmalloc(p, &p->sh.rot16, (size_t)p->max_peaks * 16 * p->csz));
  if (!p->d_wys) TRY(dmalloc(p, (void**)&p->d_wys, (size_t)p->max_batch * sizeof(double)));
  if (!p->d_shifts) TRY(dmalloc(p, (void**)&p->d_shifts, (size_t)p->max_peaks * sizeof(int)));
  HIP_TRY(hipMemcpyAsync(p->d_wys, wys.data(), (size_t)B * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_shifts, shifts.data(), (size_t)P * sizeof(int), hipMemcpyHostToDevice, p->stream));
  {
    // raw mode of the kernel: what the winners lack is dyc[p][y] = exp(2 pi i (ky_p + shift_p / EE) y), i.e. this
    // phase step per column (reduced to (-pi, pi] in double)
    std::vector<double> ys((size_t)P);
    p->sh_one_kref = true;
    for (int pp = 0; pp < P; ++pp) {
      const double c = p->staged_kr[2 * ((size_t)pp * K) + 1] + (double)shifts[pp] / EEs;
      ys[pp] = 2.0 * M_PI * (c - rint(c));
      //  the step is ONE number per peak: every candidate of a peak must share the reference vector, as
      // stage_kvectors' callers stage them today; a list that does not keeps the compensating pass B (passB_select)
      for (int k = 1; k < K; ++k)
        if (p->staged_kr[2 * ((size_t)pp * K + k) + 1] != p->staged_kr[2 * ((size_t)pp * K) + 1]) p->sh_one_kref = false;
    }
    if (!p->d_ystep) TRY(dmalloc(p, (void**)&p->d_ystep, (size_t)p->max_peaks * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(p->d_ystep, ys.data(), (size_t)P * sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));   // `ys` is a local
  }
  if (!p->sh.dyc) TRY(dmalloc(p, &p->sh.dyc, (size_t)p->max_peaks * p->n1 * p->csz));
  if (!p->sh.desc) TRY(dmalloc(p, (void**)&p->sh.desc, (size_t)p->max_batch * sizeof(int)));
  HIP_TRY(hipMemcpyAsync(p->sh.desc, desc.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, p->stream));
  if (!p->sh.order) TRY(dmalloc(p, (void**)&p->sh.order, (size_t)p->max_batch * sizeof(int)));
  HIP_TRY(hipMemcpyAsync(p->sh.order, order.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(launch_shared_tables(p->dtype, p->ax1s, p->d_wys, p->d_kr, p->d_shifts, p->d_taps, p->sh_etab, p->sh_E, p->sh_Epad, B, K,
                               p->sh_nbl, p->sh, p->stream, p->sh_elems));
  HIP_TRY(hipStreamSynchronize(p->stream));   // `desc`, `order` are locals
  p->sh_built_epoch = p->sh_epoch;
  p->sh_built_K = K;
  p->sh_built_B = B;
  p->sh_built_ok = true;
  p->sh_use = true;
  return GPA_OK;
}

// copy the (candidate, reference) k-vector lists to the device, map the candidates onto
// x-planes (one per distinct wx, see SweepTables) and build the carrier tables.
// Returns the number of x-planes in *planes_out.
int stage_kvectors(gpa_plan* p, const double* kl, const double* kr_per_b, int B, int* planes_out) {
  // same candidates as the previous call (a sequence of images analysed with one k-list):
  // the carrier tables on the device are still valid, nothing to copy and nothing to wait for
  if ((int)p->staged_kl.size() == 2 * B && memcmp(p->staged_kl.data(), kl, (size_t)B * 2 * sizeof(double)) == 0 &&
      memcmp(p->staged_kr.data(), kr_per_b, (size_t)B * 2 * sizeof(double)) == 0) {
    *planes_out = p->last_planes;
    return GPA_OK;
  }
  // the pinned staging buffer may still feed copies of an earlier asynchronous call
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->staged_kl.assign(kl, kl + 2 * (size_t)B);
  p->staged_kr.assign(kr_per_b, kr_per_b + 2 * (size_t)B);
  double* h_kl = p->h_k;
  double* h_kr = p->h_k + 2 * (size_t)B;
  double* h_pw = p->h_k + 4 * (size_t)B;
  int* h_po = reinterpret_cast<int*>(p->h_k + 5 * (size_t)B);
  memcpy(h_kl, kl, (size_t)B * 2 * sizeof(double));
  memcpy(h_kr, kr_per_b, (size_t)B * 2 * sizeof(double));
  int Bx = 0;
  for (int b = 0; b < B; ++b) {
    int found = -1;
    for (int q = 0; q < Bx; ++q)
      if (memcmp(&h_pw[q], &kl[2 * b], sizeof(double)) == 0) { found = q; break; }
    if (found < 0) { h_pw[Bx] = kl[2 * b]; found = Bx++; }
    h_po[b] = found;
  }
  p->staged_planeof.assign(h_po, h_po + B);
  ++p->sh_epoch;
  HIP_TRY(hipMemcpyAsync(p->d_kl, h_kl, (size_t)B * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_kr, h_kr, (size_t)B * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->d_pw, h_pw, (size_t)Bx * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->tb.planeof, h_po, (size_t)B * sizeof(int), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(launch_tables(p->dtype, p->ax0, p->ax1, p->d_kl, p->d_kr, B, p->d_pw, Bx, p->tb, p->stream));
  // h_k is reused by the next call: wait for the copies
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->last_planes = Bx;
  *planes_out = Bx;
  return GPA_OK;
}

// room for `planes` x-planes in Tbuf
int ensure_tbuf(gpa_plan* p, int planes) {
  if (planes <= p->tbuf_planes) return GPA_OK;
  const size_t npx = (size_t)p->n0 * p->n1;
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipFree(p->Tbuf));
  p->ws_bytes -= (size_t)p->tbuf_planes * npx * p->csz;
  p->Tbuf = nullptr;
  p->tbuf_planes = 0;
  TRY(dmalloc(p, &p->Tbuf, (size_t)planes * npx * p->csz));
  p->tbuf_planes = planes;
  return GPA_OK;
}

// scratch of at least `bytes` in p->d_sf (per-candidate phases of the a4 path, gate table of wfr4, batched lock-ins)
int ensure_sf(gpa_plan* p, size_t bytes) {
  if (p->sf_bytes >= bytes) return GPA_OK;
  if (p->d_sf) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipFree(p->d_sf));
    p->ws_bytes -= p->sf_bytes;
    p->d_sf = nullptr;
    p->sf_bytes = 0;
  }
  TRY(dmalloc(p, &p->d_sf, bytes));
  p->sf_bytes = bytes;
  return GPA_OK;
}

// 2 pi kvecs for the per-pixel solves, re-staged only when the peaks change
int stage_kmat(gpa_plan* p, const double* kvecs, int P) {
  std::vector<double> km((size_t)2 * P);
  for (int i = 0; i < 2 * P; ++i) km[i] = 2.0 * M_PI * kvecs[i];
  if (km == p->staged_kmat) return GPA_OK;
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->staged_kmat = km;
  double* h = p->h_k + 6 * (size_t)p->max_batch;
  memcpy(h, km.data(), km.size() * sizeof(double));
  HIP_TRY(hipMemcpyAsync(p->d_kmat, h, km.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  return GPA_OK;
}

