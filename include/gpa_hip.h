/* gpa_hip.h -- C ABI of libgpa_hip.so, the MI355X (gfx950) implementation of the
 * pyGPA displacement-field hot path.
 *
 * The reference (TAdeJong/pyGPA) is pure Python and has no FFI layer; its GPU
 * path is the CuPy module pyGPA/cuGPA.py.  Each entry point below names the
 * reference function(s) whose arithmetic it replaces (file:line in the reference
 * checkout); the Python host mirror in pygpa_amd/ binds them with ctypes (see
 * INTEGRATION.md for the stub a pyGPA maintainer would add).
 *
 * Conventions
 *   - return 0 on success, < 0 on error; gpa_last_error() gives the message of
 *     the last failing call on the calling thread.
 *   - images are C-contiguous (n0, n1): axis 0 <-> "x" <-> kvec[0], axis 1 <->
 *     "y" <-> kvec[1] (geometric_phase_analysis.py:72-73).
 *   - `dtype` of a plan fixes the element type of EVERY real/complex buffer
 *     passed to it: GPA_F32 -> float / interleaved float2, GPA_F64 -> double /
 *     interleaved double2.  k-vectors and sigma are always double.
 *   - functions without suffix take HOST pointers (copies in/out are part of the
 *     call); `_dev` variants take DEVICE pointers valid on the plan's device
 *     (e.g. torch.Tensor.data_ptr()) and are asynchronous on the plan's stream
 *     until gpa_plan_sync().
 *   - the caller owns every buffer; nothing passed in is modified unless
 *     documented as an output.  A plan is bound to one device and one stream and
 *     is not re-entrant: serialise calls per plan.
 */
#ifndef GPA_HIP_H
#define GPA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: these declarations are its whole dynamic symbol table (tests/test_abi.py) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define GPA_F32 0
#define GPA_F64 1

#define GPA_OK 0
#define GPA_ERR_ARG (-1)      /* bad argument (shape, NULL, unsupported size)   */
#define GPA_ERR_HIP (-2)      /* a HIP runtime call failed                       */
#define GPA_ERR_NODEV (-3)    /* no usable GPU                                   */
#define GPA_ERR_STATE (-4)    /* plan too small for the request (max_batch, ...) */

typedef struct gpa_plan gpa_plan;

int gpa_version(void);
const char* gpa_last_error(void);
int gpa_device_count(void);
/* Diagnostic / test switches of the library (no counterpart in the reference).  `name` is one of
 * the switches documented in INTEGRATION.md ("NO_LAT", "COLSOLVE", "F32_EPS_FLOOR", ..., with or
 * without the GPA_ prefix), `value` its text value or NULL to clear it.  The switches start from
 * the environment variables GPA_<NAME> as they were when the library was first used; after that
 * only this call changes them (the library never reads the environment on a call path).
 * Switches read at plan creation (NO_SHARED, NO_COMPACT, NO_KSPLIT, NO_MR, MR_FORCE_BLUESTEIN,
 * SERIAL_UNWRAP, NO_WORKER, DFT_ENGINE) apply to plans created afterwards.  Process-wide.        */
int gpa_set_option(const char* name, const char* value);

/* One plan = one device + one stream + workspace for images of shape (n0, n1)
 * and up to max_batch simultaneous lock-ins (peaks x k-vectors).              */
gpa_plan* gpa_plan_create(int device, int n0, int n1, int max_batch, int dtype);
void gpa_plan_destroy(gpa_plan* plan);
int gpa_plan_sync(gpa_plan* plan);
size_t gpa_plan_workspace_bytes(const gpa_plan* plan);
/* the hipStream_t of the plan, as an opaque pointer */
void* gpa_plan_stream(const gpa_plan* plan);
/* FFT lengths used along each axis (== n for powers of two, else the next power
 * of two >= 2n-1; the circular Gaussian convolution is evaluated exactly either way) */
int gpa_plan_fft_len(const gpa_plan* plan, int axis);

/* a1/a2 -- batched spatial lock-in
 *   out[b] = ifft2( fft2(image * exp(2 pi i (x kx_b + y ky_b))) * G_sigma )
 * replaces GPA (geometric_phase_analysis.py:20-45), optGPA (:48-76), vecGPA
 * (:79-89) and cuGPA.cuGPA (cuGPA.py:11-38).  kvecs: B x 2 doubles, out: B x n0 x n1
 * complex.  B <= max_batch.                                                    */
int gpa_lockin_batch(gpa_plan* plan, const void* image, const double* kvecs, int B,
                     double sigma, void* out);
int gpa_lockin_batch_dev(gpa_plan* plan, const void* image, const double* kvecs, int B,
                         double sigma, void* out);

/* a3 -- reference-vector sweep over an explicit, host-built k-list (K x 2,
 * candidate order = list order).  Per pixel the candidate with the strictly
 * largest |lock-in| wins (first maximum wins ties); the stored value is
 * re-referenced to kref: sf * exp(-2 pi i ((wx-kx) x + (wy-ky) y)).
 * replaces optwfr2 (geometric_phase_analysis.py:669-686), wfr2 (:615-644), wfr3
 * (:647-666), wfr2_only_lockin (:689-702) and cuGPA.wfr2_only_lockin
 * (cuGPA.py:136-158).
 *   lockin : n0 x n1 complex (output)
 *   kidx   : n0 x n1 int32 index of the winner in klist, -1 where nothing won
 *            (nullable)
 *   grad   : n0 x n1 x 2 real, a4 -- gradient of the winner's -angle(sf) by
 *            np.gradient stencils + 2 pi (w - kref), wrapped as wrapToPi(2g)/2
 *            (nullable); replaces wfr2_grad_opt (:763-813) and
 *            cuGPA.wfr2_grad_opt / wfr2_grad_single / wfr2_only_grad
 *            (cuGPA.py:41-133, :161-202).                                     */
int gpa_sweep(gpa_plan* plan, const void* image, const double* kref, const double* klist,
              int K, double sigma, void* lockin, int32_t* kidx, void* grad);
int gpa_sweep_dev(gpa_plan* plan, const void* image, const double* kref, const double* klist,
                  int K, double sigma, void* lockin, int32_t* kidx, void* grad);

/* a4 with the other gradient stencils of the reference, and a3's gated form:
 * gpa_sweep_grad: gpa_sweep with grad != NULL and a selectable stencil.  grad_mode 0 = np.gradient (= gpa_sweep);
 *   1 = forward differences along axis 0, axis 1 with NaN at the last index -- grad='diff' of cuGPA.wfr2_grad_opt /
 *   wfr2_grad_single / wfr2_only_grad (cuGPA.py:58-62); 2 = the same with the two components exchanged -- grad='diff'
 *   of wfr2_grad (geometric_phase_analysis.py:738-742).  wfr2_grad (:722-760) differentiates the compensated
 *   lock-in and wraps per candidate, which is the same number up to rounding.
 * gpa_sweep_gated: wfr4 (geometric_phase_analysis.py:839-862): candidate k replaces the kept candidate j (klist[0]
 *   until something is accepted) only where |sf_k| is strictly larger AND gate[j * K + k] != 0; the caller builds
 *   gate (K x K bytes, host) as || klist[j] - klist[k] || < 2 sqrt(2) dk in double.  kidx = -1 where nothing was
 *   accepted (the reference reports klist[0] as 'w' there).                                                   */
int gpa_sweep_grad(gpa_plan* plan, const void* image, const double* kref, const double* klist, int K,
                   double sigma, int grad_mode, void* lockin, int32_t* kidx, void* grad);
int gpa_sweep_grad_dev(gpa_plan* plan, const void* image, const double* kref, const double* klist, int K,
                       double sigma, int grad_mode, void* lockin, int32_t* kidx, void* grad);
int gpa_sweep_gated(gpa_plan* plan, const void* image, const double* kref, const double* klist, int K,
                    double sigma, const uint8_t* gate, void* lockin, int32_t* kidx);

/* a5+a6 -- phases/weights glue and per-pixel weighted least squares.
 *   phases = angle(lockin), weights = |lockin| * (mask + 1e-6), mask = 1 on
 *   [mask_border:-mask_border]^2 (extract_displacement_field,
 *   geometric_phase_analysis.py:922-926); dbdx = wrapToPi(diff along axis 1),
 *   dbdy = wrapToPi(diff along axis 0) (:234-235); per pixel solve
 *   min || w (2 pi kvecs x - b) || (myweighed_lstsq :97-113).
 * lockin: P x n0 x n1 complex; kvecs: P x 2; dudx: 2 x n0 x (n1-1); dudy:
 * 2 x (n0-1) x n1; wnorm: n0 x n1 = || weights ||_2 over peaks (:240), nullable. */
int gpa_reconstruct_grad(gpa_plan* plan, const void* lockin, const double* kvecs, int P,
                         int mask_border, void* dudx, void* dudy, void* wnorm);
int gpa_reconstruct_grad_dev(gpa_plan* plan, const void* lockin, const double* kvecs, int P,
                             int mask_border, void* dudx, void* dudy, void* wnorm);

/* a6 with pre_diff=True (reconstruct_u_inv_from_phases, geometric_phase_analysis.py:228-237): grads
 * (P x n0 x n1 x 2, host) already hold the phase gradients along axis 1 ([..., 0]) and axis 0 ([..., 1]) -- e.g.
 * the `grad` outputs of gpa_sweep; they are wrapped, solved per pixel against 2 pi kvecs with `weights`
 * (P x n0 x n1) and cropped to the difference grids: dudx 2 x n0 x (n1-1), dudy 2 x (n0-1) x n1, wnorm as above. */
int gpa_reconstruct_prediff(gpa_plan* plan, const void* grads, const void* weights, const double* kvecs, int P,
                            void* dudx, void* dudy, void* wnorm);

/* a8 helper -- per-pixel weighted least squares on given right-hand sides:
 * minimise || w (2 pi kvecs x - b) || per pixel; the weighted branch of reconstruct_u_inv
 * (geometric_phase_analysis.py:188 -> myweighed_lstsq :97-113).  b, weights: P x n0 x n1
 * (host), out: 2 x n0 x n1.                                                    */
int gpa_weighted_lstsq(gpa_plan* plan, const void* b, const void* weights, const double* kvecs, int P,
                       void* out);

/* a7 -- DCT-Laplacian weighted least-squares phase unwrap (Ghiglia-Romero PCG)
 * from pre-differenced gradients: replaces phase_unwrap_prediff
 * (phase_unwrap.py:282-350) with helpers :95-132.
 *   dx: n0 x (n1-1), dy: (n0-1) x n1, weight: n0 x n1 or NULL (unweighted),
 *   phi: n0 x n1 output.  Stops after kmax iterations or when
 *   ||r|| < eps ||r0|| (eps = 1e-9 in the reference).  poisson_axes_compat != 0
 *   replicates the reference's swapped-axis eigenvalues (phase_unwrap.py:107-109;
 *   identical for square images; where one side is at least twice the other that
 *   table is singular and the reference returns NaN -- the true eigenvalues are used
 *   there).  The f32 build floors eps at 4e-6 (what an f32 recurrence can reach).
 *   *iters_out receives the iteration count.                                    */
int gpa_unwrap_prediff(gpa_plan* plan, const void* dx, const void* dy, const void* weight,
                       int kmax, double eps, int poisson_axes_compat, void* phi, int* iters_out);
int gpa_unwrap_prediff_dev(gpa_plan* plan, const void* dx, const void* dy, const void* weight,
                           int kmax, double eps, int poisson_axes_compat, void* phi,
                           int* iters_out);
/* the two halves of gpa_unwrap_prediff_dev for callers that overlap a global unwrap with other work on the same GPU
 * (the image-pipelined multi-GPU schedule, pygpa_amd/distributed.py): `_enqueue_dev` puts the whole solve on the
 * plan's stream and returns at once; gpa_unwrap_finish waits for it and hands back the iteration count.  One solve
 * in flight per plan.                                                                                        */
int gpa_unwrap_prediff_enqueue_dev(gpa_plan* plan, const void* dx, const void* dy, const void* weight,
                                   int kmax, double eps, int poisson_axes_compat, void* phi);
int gpa_unwrap_finish(gpa_plan* plan, int* iters_out);
/* same from a wrapped phase image psi (phase_unwrap.py:141-208) */
int gpa_unwrap(gpa_plan* plan, const void* psi, const void* weight, int kmax, double eps,
               int poisson_axes_compat, void* phi, int* iters_out);

/* fused driver: extract_displacement_field (geometric_phase_analysis.py:907-932,
 * deconvolve=False) with every intermediate kept in HBM.
 *   image : n0 x n1 (its mean is subtracted on the device, :919)
 *   kvecs : P x 2 peak centres; klists: P x K x 2 host-built candidate lists
 *   sigma : Gaussian width; mask_border = 2*sigma in the reference (:924)
 *   kmax  : PCG iterations of the weighted unwrap (10 in the reference, :241)
 *   u     : 2 x n0 x n1 output;  lockins (P x n0 x n1 complex) and kidx
 *           (P x n0 x n1 int32) optional outputs;  iters_out[2] optional.
 * P*K <= max_batch.                                                           */
int gpa_extract_displacement_field(gpa_plan* plan, const void* image, const double* kvecs, int P,
                                   const double* klists, int K, double sigma, int mask_border,
                                   int kmax, void* u, void* lockins, int32_t* kidx,
                                   int* iters_out);
int gpa_extract_displacement_field_dev(gpa_plan* plan, const void* image, const double* kvecs,
                                       int P, const double* klists, int K, double sigma,
                                       int mask_border, int kmax, void* u, void* lockins,
                                       int32_t* kidx, int* iters_out);

/* Asynchronous form of the fused driver (device pointers): enqueues everything on the plan's
 * streams and returns without waiting, so that several plans (one per image in flight) overlap
 * on the GPU -- the VALU-bound sweep of one image runs beside the bandwidth-bound unwrap of
 * another.  Calls on ONE plan still execute in order.  gpa_plan_sync() waits for completion,
 * gpa_last_iters() (which synchronises) returns the two PCG iteration counts of the last call. */
int gpa_extract_displacement_field_async(gpa_plan* plan, const void* image, const double* kvecs, int P,
                                         const double* klists, int K, double sigma, int mask_border,
                                         int kmax, void* u, void* lockins, int32_t* kidx);
int gpa_last_iters(gpa_plan* plan, int* iters2);

/* A STACK of images of the plan's shape in one call (device pointers; no counterpart in the reference,
 * whose extract_displacement_field (geometric_phase_analysis.py:907-932) takes one image): images
 * B x n0 x n1, u B x 2 x n0 x n1.  The sweep and the least squares run image after image; the 2 B
 * weighted unwraps (phase_unwrap.py:282-350) share one set of kernel launches, which is what bounds a
 * small image.  Results equal B separate gpa_extract_displacement_field_dev calls bit for bit.
 * iters_out: 2 B iteration counts, or NULL to return without synchronising (gpa_plan_sync).         */
int gpa_extract_displacement_field_batch_dev(gpa_plan* plan, const void* images, int B,
                                             const double* kvecs, int P, const double* klists, int K,
                                             double sigma, int mask_border, int kmax, void* u,
                                             int* iters_out);
/* waits for the plan's stream and returns the 2 B iteration counts of the last batch call (B <= its stack size) */
int gpa_last_batch_iters(gpa_plan* plan, int B, int* iters_out);
/* 1 if the batch call above covers the plan's image shape (every shape whose unwrap runs the fused
 * iteration), 0 if the caller has to loop over gpa_extract_displacement_field_dev instead            */
int gpa_supports_batch(gpa_plan* plan);

/* tile stage of the multi-GPU path: sweep + phases/weights + per-pixel least squares of
 * extract_displacement_field (:919-926, :234-237) WITHOUT the unwrap; the gradient tiles of all
 * ranks are stitched and unwrapped once globally.  The image is used
 * as given (the caller subtracts the mean of the WHOLE image from its tiles).  Host pointers;
 * dudx: 2 x n0 x (n1-1), dudy: 2 x (n0-1) x n1, wnorm: n0 x n1.                  */
int gpa_extract_gradients(gpa_plan* plan, const void* image, const double* kvecs, int P,
                          const double* klists, int K, double sigma, int mask_border, void* dudx,
                          void* dudy, void* wnorm);

/* device-resident tile stage (BASELINE configs 4-5): the window [r0, r0+n0) x [c0, c0+n1) of a
 * larger image that lives on the device (row pitch image_pitch, in elements) is analysed like
 * gpa_extract_gradients with `mean` (of the whole image, e.g. from gpa_mean_dev) subtracted, and
 * the interior rectangle [i0, i0+t0) x [j0, j0+t1) (window coordinates) of dudx / dudy / wnorm is
 * written to dx / dy / wn: 2 planes each for dx, dy (plane stride *_plane, row pitch *_pitch,
 * elements), clipped to the one-short last column / row of the window's difference fields.  The
 * destinations may be compact per-tile buffers (to be all-gathered) or the stitched fields of the
 * whole image.  Asynchronous on the plan's stream (gpa_plan_sync).                            */
int gpa_mean_dev(gpa_plan* plan, const void* data, size_t count, double* mean_out);
int gpa_tile_gradients_dev(gpa_plan* plan, const void* image, size_t image_pitch, int r0, int c0,
                           double mean, const double* kvecs, int P, const double* klists, int K,
                           double sigma, int mask_border, int i0, int j0, int t0, int t1, void* dx,
                           size_t dx_pitch, size_t dx_plane, void* dy, size_t dy_pitch,
                           size_t dy_plane, void* wn, size_t wn_pitch);

/* The tile pipeline's own data path (SURVEY.md 8(b) "gpa_extract_sharded", 8(e) option 2; no counterpart in the
 * reference, whose only batching is the dask k-batch, geometric_phase_analysis.py:705-719).  Everything below is
 * asynchronous on the plan's stream and leaves its result on the device: an image goes through
 * pygpa_amd.distributed.TiledPipeline without a host synchronisation between its stages.
 *
 * gpa_tile_sums_dev: sum over the interior rectangles of `ntiles` windows that lie win_stride elements apart (row
 *   pitch win_pitch): rects_dev = ntiles x (o0, o1, z0, z1) ints ON THE DEVICE, max_rows >= every z0, the sum as ONE
 *   double at sum_dev (fixed summation order: reruns are bit-identical).  The ranks all-reduce these sums.
 * gpa_tile_set_mean_dev: the plan's tile mean = sum_dev[0] * scale (scale = 1 / pixels of the whole image): what
 *   gpa_tile_gradients_meandev_dev subtracts (geometric_phase_analysis.py:919) instead of a host-side value.
 * gpa_tile_gradients_meandev_dev: gpa_tile_gradients_dev with that mean; a window whose pitch equals the plan's n1 is
 *   read in place; wn_plane != 0 writes a second copy of the weight wn_plane elements behind the first (each unwrap
 *   owner receives its component's two gradients AND the weight as one contiguous block).
 * gpa_stitch_tiles_dev: tiles[slot][f] (t0 x t1 elements, pitch tile_pitch; slots slot_stride and fields field_stride
 *   elements apart) -> dst[f] (host array of nf <= 6 device pointers) at (r0, c0) of table_dev = ntiles x (slot, r0, c0,
 *   z0, z1) ints on the device, clipped to dst_rows[f] x dst_cols[f] (the difference fields are one column / row
 *   short of the image).  One launch for all tiles and fields.
 * gpa_plan_wait_stream / gpa_stream_wait_plan: the plan's stream waits for the work enqueued so far on `stream` (a
 *   hipStream_t as an opaque pointer, NULL = the default stream; e.g. torch.cuda.current_stream().cuda_stream, on
 *   which the RCCL collectives are ordered) and the other way round -- events, no host synchronisation.          */
int gpa_tile_sums_dev(gpa_plan* plan, const void* wins, size_t win_stride, size_t win_pitch, const int* rects_dev,
                      int ntiles, int max_rows, double* sum_dev);
int gpa_tile_set_mean_dev(gpa_plan* plan, const double* sum_dev, double scale);
int gpa_tile_gradients_meandev_dev(gpa_plan* plan, const void* image, size_t image_pitch, int r0, int c0,
                                   const double* kvecs, int P, const double* klists, int K, double sigma,
                                   int mask_border, int i0, int j0, int t0, int t1, void* dx, size_t dx_pitch,
                                   size_t dx_plane, void* dy, size_t dy_pitch, size_t dy_plane, void* wn,
                                   size_t wn_pitch, size_t wn_plane);
int gpa_stitch_tiles_dev(gpa_plan* plan, const void* tiles, size_t slot_stride, size_t field_stride,
                         size_t tile_pitch, const int* table_dev, int ntiles, int t0, int t1, int nf,
                         void* const* dst, const size_t* dst_pitch, const int* dst_rows, const int* dst_cols);
int gpa_plan_wait_stream(gpa_plan* plan, void* stream);
int gpa_stream_wait_plan(gpa_plan* plan, void* stream);

/* f-1 -- Lawler-Fujita undistortion.
 * gpa_invert_u_overlap: fixed-point inverse of a displacement field, `iters` rounds of cubic-
 *   spline resampling with mode='nearest' on the grid extended by `edge` pixels; replaces
 *   invert_u_overlap (geometric_phase_analysis.py:262-300).  u: 2 x n0 x n1 (host),
 *   out: 2 x (n0+2 edge) x (n1+2 edge).
 * gpa_undistort_image: out = deformed resampled at r + invert_u_overlap(-u)(r) with
 *   scipy.ndimage.map_coordinates' defaults (order 3, mode='constant', cval=0); replaces
 *   undistort_image (:935-974).  deformed, out: n0 x n1.                          */
int gpa_invert_u_overlap(gpa_plan* plan, const void* u, int iters, int edge, void* out);
/* invert_u (geometric_phase_analysis.py:248-259), the variant without overlap: out is 2 x n0 x n1 and every
 * round after the first samples at r + u_it(r) - edge.                                              */
int gpa_invert_u(gpa_plan* plan, const void* u, int iters, int edge, void* out);
/* both with scipy.ndimage's boundary mode as an argument (the `mode=` keyword of the two reference functions):
 * mode 0 = 'nearest' (their default, what the two entry points above run), 1 = 'constant' (cval 0; the last round of
 * invert_u_overlap passes cval = nan, geometric_phase_analysis.py:297-299).  overlap != 0: invert_u_overlap.      */
int gpa_invert_u_mode(gpa_plan* plan, const void* u, int iters, int edge, int overlap, int mode, void* out);
int gpa_undistort_image(gpa_plan* plan, const void* deformed, const void* u, void* out);
/* The same on device pointers, enqueued on the plan's stream WITHOUT a host synchronisation (gpa_plan_sync, or
 * gpa_stream_wait_plan for another stream); scratch is kept by the plan.  gpa_invert_u_mode_dev inverts scale * u
 * (undistort_image passes -u: scale = -1).  rects = nrect x {r0, c0, h, w} (host ints; nrect = 0: the whole grid) restricts
 * the fixed-point rounds and the resampling to those windows of the output grid -- the tiles a rank owns once it holds the
 * stitched field; the rest of out_dev is left untouched, the spline prefilter of the whole field runs once per call.
 * gpa_undistort_image_dev: uinv_dev (2 x n0 x n1, nullable) receives u_inv.                                           */
int gpa_invert_u_mode_dev(gpa_plan* plan, const void* u_dev, double scale, int iters, int edge, int overlap, int mode,
                          const int* rects, int nrect, void* out_dev);
int gpa_undistort_image_dev(gpa_plan* plan, const void* deformed_dev, const void* u_dev, const int* rects, int nrect,
                            void* uinv_dev, void* out_dev);
/* undistort_image(deformed, scale * u) on device pointers: scale = -1 undistorts with the field exactly as
 * extract_displacement_field returns it (the reference's tests recover the true displacement as MINUS that field,
 * tests/test_geometric_phase_analysis.py:63, and undistort with the true one, :76).                                   */
int gpa_undistort_image_scaled_dev(gpa_plan* plan, const void* deformed_dev, const void* u_dev, double scale,
                                   const int* rects, int nrect, void* uinv_dev, void* out_dev);

/* f-2 -- phase gradient -> Jacobian -> lattice properties.
 * gpa_phasegradient2J: J[n,m,i,j] = (per-pixel weighted least squares of grads[:,n,m,j] against
 *   2 pi kvecs)_i / nmperpixel; replaces phasegradient2J (property_extract.py:69-101).
 *   dks == NULL is iso_ref=False; with dks (P x 2, = calc_diff_from_isotropic(kvecs)) the
 *   gradients are taken relative to the isotropic lattice: K = 2 pi (kvecs + dks) and
 *   g <- wrapToPi(g - 2 pi dk) (:87-92).  grads: P x n0 x n1 x 2 (the `grad` output of
 *   gpa_sweep per peak), weights: P x n0 x n1, J: n0 x n1 x 2 x 2.
 * gpa_props_from_jac: (angle [deg], aniangle [deg, mod 180], alpha, kappa) per pixel from the 2x2
 *   Jacobian (+ identity when add_identity != 0, i.e. phasegradient2Jac :104-111); replaces
 *   props_from_Jac (:137-178).  Needs no plan: jac is npx x 2 x 2 for any number of pixels,
 *   props 4 x npx.
 * The _dev variants take device pointers and run asynchronously (on the plan's stream, or on
 * the given hipStream_t for gpa_props_from_jac_dev).                                        */
int gpa_phasegradient2J(gpa_plan* plan, const double* kvecs, int P, const void* grads,
                        const void* weights, double nmperpixel, const double* dks, void* J);
int gpa_phasegradient2J_dev(gpa_plan* plan, const double* kvecs, int P, const void* grads,
                            const void* weights, double nmperpixel, const double* dks, void* J);
/* np.abs of P lock-ins (P x n0 x n1 complex -> P x n0 x n1 real), the `weights` the reference's callers hand to
 * phasegradient2J (property_extract.py:69); device pointers, on the plan's stream.                              */
int gpa_lockin_weights_dev(gpa_plan* plan, const void* lockins, int P, void* weights);
int gpa_props_from_jac(int device, int dtype, size_t npx, const void* jac, int add_identity,
                       double refangle, double refscale, int diff, void* props);
int gpa_props_from_jac_dev(int device, int dtype, size_t npx, const void* jac, int add_identity,
                           double refangle, double refscale, int diff, void* props,
                           void* stream);

/* f-4 -- robust plane fit a0*x + a1*y + a2 (x = row, y = column index) through an n0 x n1 map:
 * the minimiser of the Huber cost that fit_plane (mathtools.py:30-47, scipy least_squares with
 * loss='huber', f_scale 1, start at 0) approaches, by iteratively reweighted least squares with the
 * per-iteration sums reduced on the device.  Stops when the plane changes by <= tol (sum of the
 * three coefficient changes in centred unit coordinates) or after max_iter passes.             */
int gpa_fit_plane(gpa_plan* plan, const void* image, int max_iter, double tol, double* coef3,
                  int* iters_out);
int gpa_fit_plane_dev(gpa_plan* plan, const void* image, int max_iter, double tol, double* coef3,
                      int* iters_out);

/* gaussian_deconvolve (geometric_phase_analysis.py:892-904) of ONE field: reflect-pad by 2 dr,
 * Wiener-Hunt deconvolution (skimage.restoration.wiener, Laplacian regulariser, `balance`) with the
 * Gaussian the lock-ins were smoothed by, crop.  The plan has the PADDED shape
 * (m0 + 4 dr) x (m1 + 4 dr); data, out: m0 x m1 (host).                                       */
int gpa_gaussian_deconvolve(gpa_plan* plan, const void* data, int dr, double sigma, double balance,
                            void* out);
/* the same on device pointers (d_data, d_out: m0 x m1 reals), on the plan's stream; returns after the stream has drained */
int gpa_gaussian_deconvolve_dev(gpa_plan* plan, const void* d_data, int dr, double sigma, double balance,
                                void* d_out);

/* a9 -- DFT of the periodic component of Moisan's periodic+smooth decomposition
 * (third-party moisan2011.per(image, inverse_dft=False)[0], call site
 * geometric_phase_analysis.py:429).  out: n0 x n1 complex.                    */
int gpa_per_dft(gpa_plan* plan, const void* image, void* out);
/* the same on device pointers, enqueued on the plan's stream (no host synchronisation); d_image is not modified.
 * Image sizes: every axis length 1 ... 65536 in both precisions (gpa_dft.h: a power of two up to 16384 (f32) / 8192 (f64)
 * runs the register FFT at its own length, other lengths a chirp-z transform, in one workgroup while 2n - 1 <= 16384 / 8192
 * and as a two-level transform through HBM beyond).                                                                        */
int gpa_per_dft_dev(gpa_plan* plan, const void* d_image, void* d_out);

/* a9, the whole of moisan2011.per(image, inverse_dft) (the reference imports it at
 * geometric_phase_analysis.py:9 and calls it at :429 with inverse_dft=False):
 * inverse_dft == 0: p_out, s_out = DFTs of the periodic and the smooth component (n0 x n1 complex; s_out may be NULL);
 * inverse_dft != 0: p_out, s_out = the two components themselves (n0 x n1 real), p + s = image.      */
int gpa_per(gpa_plan* plan, const void* image, int inverse_dft, void* p_out, void* s_out);

/* f-3 -- Bragg-peak candidates, the array work of extract_primary_ks
 * (geometric_phase_analysis.py:427-437): smooth = gaussian_filter(|fftshift(per_dft(image - mean))|,
 * sigma) minus, when dog_sigma > 0, the same with dog_sigma (scipy.ndimage.gaussian_filter
 * semantics: truncate 4, mode 'reflect'); peaks = skimage.feature.peak_local_max(smooth,
 * threshold_rel=threshold_rel) with its defaults (3x3 maxima above max(min, rel*max), 1-pixel
 * border excluded).  image: n0 x n1 (host).  coords: up to max_out (row, column) pairs in the
 * fftshift-ed frame, values: their smooth values (plan dtype), in no particular order (sort by
 * value, then raster index, for skimage's order); *count_out = number found (> max_out means
 * the arrays hold only a subset: call again with more room).  smooth_out: n0 x n1 or NULL.   */
int gpa_find_peaks(gpa_plan* plan, const void* image, double sigma, double dog_sigma,
                   double threshold_rel, int max_out, int32_t* coords, void* values,
                   int* count_out, void* smooth_out);
/* the same with the image (not modified) and the optional smooth_out on the device; coords, values, count_out stay host
 * arrays (a handful of entries).  Synchronises the plan's stream (the count decides what is copied).                    */
int gpa_find_peaks_dev(gpa_plan* plan, const void* d_image, double sigma, double dog_sigma,
                       double threshold_rel, int max_out, int32_t* coords, void* values,
                       int* count_out, void* d_smooth_out);
/* peak_local_max at another threshold_rel of the smoothed spectrum that the last gpa_find_peaks / gpa_find_peaks_dev call
 * of this plan computed (the plan keeps it): the re-evaluations of extract_primary_ks' parameter relaxation
 * (geometric_phase_analysis.py:447-467) while sigma and DoG stay the same cost one threshold + maxima pass instead of the
 * transforms and filters.  GPA_ERR_STATE before the first gpa_find_peaks call.                                           */
int gpa_find_peaks_again(gpa_plan* plan, double threshold_rel, int max_out, int32_t* coords, void* values,
                         int* count_out);

/* timing hooks used by bench.py: elapsed milliseconds between two recorded
 * events on the plan's stream (HIP events, so it measures the stream the
 * kernels are launched on).                                                   */
int gpa_timer_start(gpa_plan* plan);
int gpa_timer_stop(gpa_plan* plan, float* ms_out);
/* per-stage device times (ms) of the last fused-driver call, measured with HIP
 * events when gpa_set_profiling(plan, 1) is on:
 *   [0] tables+mean  [1] sweep pass A (x-axis)  [2] sweep pass B (y-axis+select)
 *   [3] reconstruct  [4] unwrap (both components)                             */
int gpa_set_profiling(gpa_plan* plan, int on);
int gpa_last_stage_ms(gpa_plan* plan, float* ms5);
/* per-KERNEL device times of the last fused-driver call made with profiling on (every launch is
 * bracketed by HIP events on the stream it runs on; the two unwraps then run one after the other):
 * one text line "name launches total_ms" per kernel, NUL-terminated, into out[cap].              */
int gpa_last_kernel_profile(gpa_plan* plan, char* out, size_t cap);

/* Download of a result while the GPU goes on with the next call: copies `bytes` from device memory
 * (e.g. the u of gpa_extract_displacement_field_async) to PAGE-LOCKED host memory on the plan's copy
 * stream, ordered after everything enqueued on the plan so far.  `slot` (0..3) names the completion
 * event: gpa_download_wait(plan, slot) blocks the host until that copy has landed.  The caller must not
 * let a later call overwrite dev_src before the copy is done (alternate two result buffers and wait for
 * slot i before reusing buffer i).  This is how bench.py keeps the D2H of u inside the timed step
 *  without serialising it with the kernels.                                       */
int gpa_download_async(gpa_plan* plan, void* host_dst, const void* dev_src, size_t bytes, int slot);
int gpa_download_wait(gpa_plan* plan, int slot);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GPA_HIP_H */
