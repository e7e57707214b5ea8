// a7: DCT-Laplacian weighted least-squares phase unwrap (Ghiglia & Romero PCG),
// following phase_unwrap_prediff / phase_unwrap (phase_unwrap.py:282-350, :141-208).
//
//   r0 = A^T W^2 wrap(b);  repeat:  z = P^-1 r (DCT Poisson solve);  rho = <r,z>;
//   p = z + (rho/rho_prev) p;  q = A^T W^2 A p;  alpha = rho/<p,q>;
//   phi += alpha p;  r -= alpha q;  stop at kmax or ||r|| < eps ||r0||.
//
// This file is the PCG driver: it enqueues kmax iterations back to back on the caller's stream and reads the
// iteration count once at the end; every scalar of the iteration (rho, alpha, beta, norms, the iteration counter
// and the stop flag) lives on the device, and kernels of iterations after convergence return immediately.  The
// kernels live in the translation units named in gpa_unwrap_impl.h.
//
// Every image size the workspace supports takes the fused iteration (run_pcg): four kernels and no vector moved
// that does not have to be --
//   rowdct_fused : R -= alpha DCT_rows(q)   the residual is kept as its row spectrum R; ||r||^2 by Parseval
//   colsolve     : R -> Z                   column DCT-II / eigenvalue divide / DCT-III (or the transform-free
//                                           recursion); stop test; rho = <r,z> by Parseval
//   rowidct_p    : Z -> p = z + beta p_prev row DCT-III straight into the new search direction
//   pq           : q = A^T W^2 A p          one sliding-window stencil pass, partial <p,q>
// and phi += alpha p is applied for up to 10 iterations at once by phi_flush_kernel from the kept search
// directions.  11 array passes per iteration (+ 1.2 for the flush) instead of the 19 of the plain scheme, which
// survives for sizes no fused path covers (Bluestein DCTs, one vector update per kernel).
#include "gpa_unwrap_impl.h"

namespace gpa {

static __global__ void set_stall_limit_kernel(double* scal, int nprob, double limit) {
  for (int pb = threadIdx.x; pb < nprob; pb += blockDim.x) scal[(size_t)pb * SCAL_N + SC_STALL_LIMIT] = limit;
}

hipError_t run_pcg(Impl* w, const void* a, const void* b, const void* weight, bool from_psi, int kmax, double eps,
                          int compat, void* phi, hipStream_t s) {
  const int n0 = w->n0, n1 = w->n1;
  const size_t npx = (size_t)n0 * n1;
  // The reference's preconditioner table uses cos(pi I / M) + cos(pi J / N) with the axis lengths swapped
  // (phase_unwrap.py:107-109); `compat` reproduces that.  Once one side is at least twice the other that
  // table has a zero away from the DC bin (I = 2M) and the reference itself returns NaN: there is nothing
  // to reproduce, and the true Laplacian eigenvalues are used instead.
  if (compat && (n0 >= 2 * n1 || n1 >= 2 * n0)) compat = 0;
  {
    const OptVal& mode = opt(OPT_COLSOLVE);   // once per solve: 0 default, 1 tri, 2 fft, 3 stream
    w->col_mode = !mode.set ? 0 : (mode.str[0] == 't' ? 1 : (mode.str[0] == 's' ? 3 : 2));
    w->lat_ok = !opt_set(OPT_NO_LAT);
  }
  {
    // the f32 stagnation guard's count (pcg_breakdown (2)): 2 by default, F32_STALL=<n>, F32_STALL=0 = off
    const OptVal& st = opt(OPT_F32_STALL);
    const double limit = !st.set ? PCG_STALL : (st.num > 0.0 ? st.num : 1e300);
    if (w->stall_limit_dev != limit) {
      set_stall_limit_kernel<<<1, 64, 0, s>>>(w->scal, w->cap, limit);
      hipError_t e0 = hipGetLastError();
      if (e0 != hipSuccess) return e0;
      w->stall_limit_dev = limit;
    }
  }
  if (n0 > MAXPART) return hipErrorInvalidValue;   // (one partial sum per image row in the plain scheme)
  int npq = pq_partials(w);   // (the fused row + stencil kernel of small images reports its own count)
  if (npq > MAXPART) return hipErrorInvalidValue;
  // The default is the reference's stopping test alone (phase_unwrap.py:348), in both precisions.  The residual of an
  // f32 iteration cannot fall below a few ulps of ||r0||, so an f32 solve never meets eps = 1e-9 and runs its kmax
  // iterations; F32_EPS_FLOOR=<eps> (opt-in, e.g. 4e-6) lets f32 solves stop at that relative residual instead.
  // (The breakdown guard of the stop test -- NaN, or a residual 100 x above the best seen -- is always on.)
  if (w->dtype == 0 && opt_set(OPT_F32_EPS_FLOOR)) {
    const double eps_floor = opt(OPT_F32_EPS_FLOOR).num;
    if (eps < eps_floor) eps = eps_floor;
  }
  hipError_t e;
  const bool fused_path = !w->generic || w->mr_ok;   // the fused 4-kernel iteration
  if (a) {
    if ((e = launch_unwrap_setup(w, a, b, weight, from_psi, phi, s)) != hipSuccess) return e;
  } else {
    // prepared: r0 and its w->prepared_parts partial norms were written by the producer of the gradients
    // (reconstruct_setup_kernel).  phi = 0: the fused path's first phi_flush_kernel starts from 0, the
    // other paths update phi in place and need it cleared
    if (!fused_path && (e = hipMemsetAsync(phi, 0, npx * w->rsz, s)) != hipSuccess) return e;
    // (fused path: the first row kernel starts the solve from the producer's partial norms, solve_init())
    if (!fused_path && (e = launch_scal_init(w, w->prepared_parts, s)) != hipSuccess) return e;
  }
  w->iters_slot = fused_path ? 3 : 0;
  if (w->nprob > 1 && (a || !fused_path)) return hipErrorNotSupported;   // batched: prepared start on the fused path only
  if (fused_path) {
    // 4 kernels per iteration, no scalar kernels.  The phi / r update of iteration it-1 rides in the row-DCT
    // kernel of iteration it; the stopping test is evaluated by the column kernel from that kernel's partial norms.
    double* part_rho = w->part;
    double* part_pq = w->part + MAXPART;
    double* part_norm = w->part + 2 * MAXPART;
    // the search directions of the last `ring` iterations stay in HBM (64 MiB each at 4096^2 f32 -- HBM is
    // plentiful), so phi += alpha p costs one pass over phi per `ring` iterations instead of one per iteration
    int ring = kmax < RING_MAX ? kmax : RING_MAX;
    if (ring < 2) ring = 2;
    while (w->nring < ring) {
      void* buf = nullptr;
      if (hipMalloc(&buf, npx * w->rsz * w->cap) != hipSuccess) { (void)hipGetLastError(); break; }
      w->ring[w->nring++] = buf;
    }
    if (w->nring < ring) ring = w->nring;   // out of memory: flush more often
    bool phi_unwritten = a == nullptr;   // prepared start: nobody has zeroed phi
    int nnorm = 0;
    // one image with rows of at most 512 pixels: row kernel and stencil in one launch (rowidct_pq_kernel)
    const bool rowpq = !w->generic && w->lat_ok && w->nprob <= 2 && w->lg1 <= GPA_ROWPQ_MAXLG && w->n0 >= 4 && !opt_set(OPT_NO_ROWPQ);
    // rows of 2048 / 4096 points with the streamed column solve: stencil and row transform in one launch (pqdct_kernel),
    // the residual update applied by the column solve's first launch -- five launches and 44 bytes per pixel per
    // iteration instead of six and 48 (NO_PQDCT keeps the separate kernels)
    const bool fuse_pq = !rowpq && pow2_pqdct_offered(w) && colstream_is_default(w) && !opt_set(OPT_NO_PQDCT);
    for (int it = 0; it < kmax; ++it) {
      // (first iteration of a prepared start: the partial norms of r0 ride in the part_pq / npq arguments)
      const bool init = it == 0 && a == nullptr;
      const double* pin_part = init ? w->part : part_pq;
      const int pin_n = init ? w->prepared_parts : npq;
      int nrow = 0;   // partial sums of rho = <r, z>: one per column workgroup (Parseval, solve_combine)
      if (fuse_pq && it > 0) {
        // w->q holds D = DCT_rows(q) of the previous iteration: R -= alpha D inside the column solve
        if ((e = dispatch_colstream_update(w, compat, s, it, eps, ring, w->q, part_pq, npq, part_norm, &nnorm, part_rho, &nrow)) != hipSuccess) return e;
      } else {
        e = w->generic ? mr_rowdct_fused(w, w->q, ring, pin_part, pin_n, part_norm, it, &nnorm, init ? 1 : 0, s)
                       : pow2_rowdct_fused(w, w->q, ring, pin_part, pin_n, part_norm, it, &nnorm, init ? 1 : 0, s);
        if (e != hipSuccess) return e;
        if ((e = dispatch_colsolve(w, compat, s, part_norm, nnorm, it, eps, part_rho, &nrow, w->r)) != hipSuccess) return e;
      }
      if (it > 0 && it % ring == 0) {   // slot it % ring still holds p of iteration it - ring: a flush in mid-solve
        if ((e = launch_phi_flush(w, ring, phi, phi_unwritten, 0, part_pq, npq, s)) != hipSuccess) return e;
        phi_unwritten = false;
      }
      const void* pin = w->ring[(it + ring - 1) % ring];
      void* pout = w->ring[it % ring];
      if (rowpq) {
        if ((e = pow2_rowidct_pq(w, pin, pout, weight, part_rho, nrow, part_pq, &npq, it, s)) != hipSuccess) return e;
        continue;
      }
      e = w->generic ? mr_rowidct_p(w, pin, pout, part_rho, nrow, it, s) : pow2_rowidct_p(w, pin, pout, part_rho, nrow, it, s);
      if (e != hipSuccess) return e;
      if (fuse_pq && it + 1 < kmax) {
        if ((e = pow2_pqdct(w, pout, weight, part_pq, &npq, s)) != hipSuccess) return e;
      } else {
        // (the last iteration's q is only needed for its <p, q>: the plain stencil)
        npq = pq_partials(w);
        if ((e = launch_pq(w, pout, weight, it, part_pq, s, it + 1 < kmax)) != hipSuccess) return e;
      }
    }
    // the flush that ends the solve takes the last step length from the stencil kernel's partial sums and files the
    // iteration count
    return launch_phi_flush(w, ring, phi, phi_unwritten, kmax, part_pq, npq, s);
  }
  // sizes without a fused path (no mixed-radix plan fits LDS): the Bluestein kernels, one vector update per kernel
  for (int it = 0; it < kmax; ++it) {
    int nrow = 0;
    if ((e = dispatch_g_rowdct(w, s)) != hipSuccess) return e;
    if ((e = dispatch_g_colsolve(w, compat, s)) != hipSuccess) return e;
    if ((e = dispatch_g_rowidct(w, &nrow, s)) != hipSuccess) return e;
    if ((e = launch_plain_tail(w, weight, phi, nrow, kmax, eps, s)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

hipError_t unwrap_enqueue(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight, bool from_psi,
                          int kmax, double eps, bool axes_compat, void* phi, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  return run_pcg(w, a, b, weight, from_psi, kmax, eps, axes_compat ? 1 : 0, phi, s);
}

void* unwrap_residual_buffer(UnwrapWorkspace* ws, int problem) {
  Impl* w = (Impl*)ws->impl;
  return w ? (char*)w->r + (size_t)problem * w->n0 * w->n1 * w->rsz : nullptr;
}
double* unwrap_partials_buffer(UnwrapWorkspace* ws, int problem) {
  Impl* w = (Impl*)ws->impl;
  return w ? w->part + (size_t)problem * PART_N : nullptr;
}

hipError_t unwrap_enqueue_prepared(UnwrapWorkspace* ws, const void* weight, int nparts, int kmax, double eps,
                                   bool axes_compat, void* phi, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  if (nparts < 1 || nparts > MAXPART) return hipErrorInvalidValue;
  w->prepared_parts = nparts;
  return run_pcg(w, nullptr, nullptr, weight, false, kmax, eps, axes_compat ? 1 : 0, phi, s);
}


hipError_t unwrap_finish(UnwrapWorkspace* ws, int* iters_out, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  int it = 0;
  hipError_t e = hipMemcpyAsync(&it, w->flags + w->iters_slot, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  if (iters_out) *iters_out = it;
  return hipSuccess;
}

hipError_t unwrap_fetch_iters(UnwrapWorkspace* ws, int* host_pinned, hipStream_t s) {
  Impl* w = (Impl*)ws->impl;
  if (!w || !w->supported) return hipErrorNotSupported;
  // one problem: its count alone; several: the flag words of all of them, the count of problem j at
  // host_pinned[FLAGS_N * j + unwrap_iters_slot()]
  if (w->nprob == 1) return hipMemcpyAsync(host_pinned, w->flags + w->iters_slot, sizeof(int), hipMemcpyDeviceToHost, s);
  return hipMemcpyAsync(host_pinned, w->flags, (size_t)FLAGS_N * w->nprob * sizeof(int), hipMemcpyDeviceToHost, s);
}
// a workspace created for `cap` problems solves the first n of them (n <= cap): a stack whose last chunk is ragged,
// or a shorter stack, reuses the buffers instead of paying a destroy / create (hipMalloc, table upload) per change
bool unwrap_set_active(UnwrapWorkspace* ws, int n) {
  Impl* w = (Impl*)ws->impl;
  if (!w || n < 1 || n > w->cap) return false;
  w->nprob = n;
  return true;
}
int unwrap_iters_slot(const UnwrapWorkspace* ws) {
  const Impl* w = (const Impl*)ws->impl;
  return w ? w->iters_slot : 0;
}

hipError_t unwrap_run(UnwrapWorkspace* ws, const void* a, const void* b, const void* weight, bool from_psi, int kmax,
                      double eps, bool axes_compat, void* phi, int* iters_out, hipStream_t s) {
  hipError_t e = unwrap_enqueue(ws, a, b, weight, from_psi, kmax, eps, axes_compat, phi, s);
  if (e != hipSuccess) return e;
  return unwrap_finish(ws, iters_out, s);
}

}  // namespace gpa
