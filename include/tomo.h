/*
 * include/tomo.h -- C-ABI of libtomo_hip.so, the MI355X (gfx950) implementation of the
 * ray-driven projection hot path of pandekan/tomography_alignment.
 *
 * This is the drop-in boundary: plain pointers and sizes, no framework types.  Each entry point
 * cites the reference interface it replaces (path:line in the reference tree).  The reference's
 * native surface is two f2py modules (src/ray_wt_grad.f90, src/vox_wt_grad.f90) plus three
 * un-wired matrix-free Fortran routines (src/forward_projection.f90, src/back_projection.f90,
 * src/projection_gradient.f90); a maintainer binds this library with ctypes exactly where
 * utilities/ray_voxel_utilities.py:3 does `from src import ray_wt_grad`.
 *
 * Conventions (all from the reference):
 *   volume   float32, C order [x][y][z], linear index (ix*ny+iy)*nz+iz   src/ray_wt_grad.f90:38
 *   detector ray index r = ix*ndz + iz                                    utilities/geometry.py:94-100
 *   sinogram float32 [n_proj][ndx*ndz]                                    recon/sirt.py:60
 *   pose     7 doubles per projection: phi, alpha, beta, tx, ty, tz, cor_x
 *            T(x) = Rz(phi) Rx(alpha) (Ry(beta) x + t), source/detector x first shifted by cor_x
 *                                                                         utilities/ray_voxel_utilities.py:6-12,72-73
 *   gradient rows tx, ty, tz, phi, alpha, beta                            utilities/ray_voxel_utilities.py:39-49
 *
 * Ownership: `d_*` arguments are device pointers obtained from tomo_malloc (or any hipMalloc'ed
 * memory of the same device); `h_*` are caller-owned host buffers.  Nothing allocated by the
 * library is ever returned to the caller except through tomo_malloc.
 * Errors: every function returns 0 on success or a negative tomo_status; tomo_last_error() gives
 * text.  (The reference has no error channel at all.)  Out-of-volume samples contribute zero, as in
 * src/ray_wt_grad.f90:35-89 -- that is semantics, not an error.
 * Threading: one ctx per GPU per process; calls on a ctx are ordered on its HIP stream.  Compute
 * entry points are asynchronous w.r.t. the host unless they return host results.
 */
#ifndef TOMO_H_
#define TOMO_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__) || defined(__clang__)
#define TOMO_API __attribute__((visibility("default")))
#else
#define TOMO_API
#endif

#define TOMO_ABI_VERSION 1
#define TOMO_POSE_STRIDE 7

typedef struct tomo_ctx tomo_ctx;

typedef enum tomo_status {
    TOMO_OK = 0,
    TOMO_ERR_HIP = -1,        /* a HIP runtime call failed (text has hipGetErrorString) */
    TOMO_ERR_ARG = -2,        /* bad argument / shape */
    TOMO_ERR_STATE = -3,      /* geometry not set, comm not initialised, ... */
    TOMO_ERR_RCCL = -4,       /* an RCCL call failed */
    TOMO_ERR_UNSUPPORTED = -5
} tomo_status;

/* Mirror of the attributes of utilities/geometry.py:9-105 `Geometry` that the hot path reads. */
typedef struct tomo_geom {
    int32_t nx, ny, nz;      /* vox_shape                                   geometry.py:24 */
    int32_t ndx, ndz;        /* det_shape                                   geometry.py:28 */
    double vox_origin[3];    /* world coords of voxel (0,0,0) centre        geometry.py:87 */
    double vox_pitch[3];     /* voxel pitch (only the voxel-driven back-projector reads it; the ray
                                path, like the reference, takes index = world - origin) geometry.py:25 */
    double det_x0, det_z0;   /* world x,z of detector pixel (0,0) centre    geometry.py:92-93 */
    double det_dx, det_dz;   /* detector pixel pitch                        geometry.py:29 */
    double src_y, det_y;     /* y of the source / detector planes (-sy,+sy) geometry.py:95-96 */
    double step;             /* ray-march step                              geometry.py:46 */
} tomo_geom;

/* ---------------------------------------------------------------- context / memory */
TOMO_API int tomo_abi_version(void);
TOMO_API int tomo_device_count(int *n);
TOMO_API int tomo_ctx_create(int device, tomo_ctx **out);
TOMO_API int tomo_ctx_destroy(tomo_ctx *ctx);
TOMO_API const char *tomo_last_error(const tomo_ctx *ctx);          /* ctx may be NULL: last global error */
TOMO_API int tomo_device_name(tomo_ctx *ctx, char *buf, size_t n);
TOMO_API int tomo_malloc(tomo_ctx *ctx, size_t bytes, void **d_ptr);
TOMO_API int tomo_free(tomo_ctx *ctx, void *d_ptr);
TOMO_API int tomo_memcpy_h2d(tomo_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
TOMO_API int tomo_memcpy_d2h(tomo_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
TOMO_API int tomo_memcpy_d2d(tomo_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
TOMO_API int tomo_memset0(tomo_ctx *ctx, void *d_ptr, size_t bytes);
TOMO_API int tomo_sync(tomo_ctx *ctx);
/* Integer knobs; unknown keys are TOMO_ERR_ARG.
 *   "fwd_variant" 1 ray-driven (plain), 2 ray-driven (SGPR block base), 3 LDS tile kernels (default)
 *   "adj_variant" 1 global float atomics, 2 LDS tile kernels with fixed-point accumulation (default)
 *   "grad_variant" kernels behind tomo_proj_grad / tomo_cost_grad(_rows); all give the same sums:
 *                  1 plain (per-lane 64-bit addressing, corner pairs as dwordx2 gathers);
 *                  2 wave-uniform block base + 32-bit lane offsets, eight dword gathers, two samples in flight, pair-packed
 *                    lerps, cache-aware block order -- fastest for near-untilted poses;
 *                  3 as 2 with four gathers per sample: the upper-z corners come from the neighbouring lane (DPP wave
 *                    shift) -- nearly insensitive to tilt, fastest beyond |alpha| + |beta| ~ 1 deg;
 *                  4 (default) each pose of a call takes 2 or 3 by its tilt (two launches)
 *   "tile_flat"   1 (default): untilted projections (alpha = beta = 0, detector-z pitch 1) take the flat tile kernels
 *   "fwd_flat_ztiles" 2 (default): the flat forward kernel owns two z-adjacent tiles per work-group and builds each detector
 *                 row's sample table once for both; 1: one tile per work-group
 *   "fwd_flat_tab" 1 (default): the flat forward over 16 x 16 x 128-voxel blocks with the two 64-plane images interleaved per plane and
 *                 each row's weights in an LDS table (k_fwd_flat_tab), run over the blocks that hold a non-zero voxel only; 0: the round-2
 *                 kernel (entries broadcast with v_readlane)
 *   "fwd_flat_wide" 0 (default); 1: measurement variant of the flat forward with a 32 x 16 x 63 tile footprint and one image
 *                 per work-group (half the tile crossings, hence half the sinogram atomics; whole-volume calls only)
 *   "adj_flat_gather" 1 (default): untilted projections on a unit lattice (detector pitch = step = voxel) take the gather-form
 *                 adjoint (accumulators in registers, no atomics) instead of the LDS-atomic flat tile kernel
 *   "reuse_staged_volume" 1: the caller vouches that the volume passed to tomo_proj_grad / tomo_cost_grad / the ray-driven
 *                 forward has not changed since the previous such call with the same pointer, so its zero-padded staging
 *                 copy is reused (alignment loops evaluate hundreds of poses against one volume); default 0
 *   "reuse_sino_flags" 1: the caller vouches that the sinogram passed to tomo_adjoint / tomo_adjoint_xslab has not changed since the
 *                 previous such call with the same pointer and number of projections, so the scan that marks its non-empty detector-z
 *                 planes is not repeated (the x-slab calls of ONE back-projection pass: the solver sets it for slabs 2..n); default 0.
 *                 A call with another pointer, number of projections, detector shape, volume height or set of integer pose z offsets
 *                 finds the marks stale and scans again; so does the first back-projection after a forward call that took the tile
 *                 kernels (their block lists share the buffer) or after tomo_set_geometry
 *   "grad_v1_prec" 0..3 (diagnostic, tomo_proj_grad with grad_variant 1 only): bit 0 float64 sample positions, bit 1 float64 lerps and
 *                 sums -- shows which float32 step an error comes from (profiles/round4_grad_error_model.md); default 0
 *   "comm_test_poison_us" n (tests only): every asynchronous collective first doubles its buffer, idles n microseconds and halves it
 *                 again on the communication stream, so that a compute-stream kernel that did not wait for it is caught by a
 *                 one-rank run; default 0
 *   "comm_test_copy_eighths" k, "comm_test_copy_wgs" w (measurements only; tools/contention_probe.py): every asynchronous collective also copies
 *                 k/8 of the buffer it touches to a scratch buffer on the communication stream -- by hipMemcpyAsync (w = 0) or by a copy kernel
 *                 of w work-groups -- so that a one-rank run carries the HBM / CU load the collectives of an N-rank run would add; default 0
 *   "roctx" 0 / 1 (process-wide; also TOMO_ROCTX=1 in the environment): roctx ranges named after the entry point around tomo_forward(_xslab),
 *                 tomo_adjoint(_xslab), tomo_backproject_voxel, tomo_proj_grad, tomo_cost_grad(_rows) and the collectives, through
 *                 librocprofiler-sdk-roctx.so / libroctx64.so loaded on demand (rocprofv3 --marker-trace); TOMO_ERR_UNSUPPORTED if neither loads;
 *                 default 0 -- the reference has print timings only (recon/sirt.py:57,80-82) */
/* HIP's current device is per thread: a thread other than the context's creator calls this once before it uses the context
 * (entry points that need a geometry also do it themselves).  One context must still not be used by two threads at the same time. */
TOMO_API int tomo_ctx_make_current(tomo_ctx *ctx);
TOMO_API int tomo_set_option(tomo_ctx *ctx, const char *key, int value);
/* Restrict the context's compute stream to a subset of the CUs (hipExtStreamCreateWithCUMask: bit i of the mask = CU i in the HIP
 * runtime's numbering; n_words 32-bit words; n_words = 0 restores the unrestricted stream; an all-zero mask is refused).  Work queued on
 * the old stream is waited for.  AS MEASURED on this runtime (tools/overlap_probe.py, profiles/round5_overlap_probe.md): masks that are
 * LEADING RANGES of the CUs ("the first k") restrict the stream and two contexts with disjoint ranges ran their kernels side by side;
 * masks that keep some CUs of every group of 8 had no effect.  The masked stream is created with default flags, i.e. it is a BLOCKING
 * stream: unlike the context's normal compute stream (hipStreamNonBlocking) it synchronises implicitly with the NULL stream.  A
 * measurement aid (one-GPU overlap probes), not used by the solvers. */
TOMO_API int tomo_ctx_set_cu_mask(tomo_ctx *ctx, const uint32_t *mask, int n_words);

/* Geometry: replaces passing a `Geometry` object to utilities/ray_voxel_utilities.py:53,113.
 * tomo_check_geometry is the validation tomo_set_geometry applies, callable without a context or a GPU: TOMO_ERR_ARG for
 * non-positive shapes / step or det_y <= src_y, TOMO_ERR_UNSUPPORTED for a zero-padded volume of 2^31 voxels or more;
 * *flags (nullable) receives TOMO_GEOM_* bits: WIDE_ROWS = one padded x-row ((ny+4)*(nz+4) floats) spans 2^23 bytes or
 * more (a slab such as 16 x 2048 x 2048), or the detector-z pitch or the sample step exceeds one voxel -- geometries for which the ray-driven
 * kernels with 24-bit offset multiplies and fixed lane biases (fwd_variant 2, grad_variant 2-4) are replaced by their plain
 * 64-bit-indexing twins (variant 1): same results, slower. */
#define TOMO_GEOM_WIDE_ROWS 1
TOMO_API int tomo_check_geometry(const tomo_geom *g, int *flags);
TOMO_API int tomo_set_geometry(tomo_ctx *ctx, const tomo_geom *g);

/* ---------------------------------------------------------------- projectors
 * tomo_forward: proj[ip, r] = (A x)[ip*n_det + r] for the operator assembled by
 *   utilities/projection_operators.py:22-76 (ProjectionMatrix.projection_matrix -> _forward_ray ->
 *   utilities/ray_voxel_utilities.py:53-110 forward_sparse -> src/ray_wt_grad.f90:1-92
 *   trilinear_ray_sparse), applied as recon/sirt.py:59 `sparse.csr_matrix.dot(A, x)`;
 *   same semantics as the matrix-free src/forward_projection.f90:1-68 forward_project (which
 *   ignores cor_shift: pass cor_x = 0 to mimic it).
 *   d_vol [nx*ny*nz], d_proj [n_proj*ndx*ndz] (overwritten). */
TOMO_API int tomo_forward(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_vol, float *d_proj);

/* tomo_adjoint: vol = A^T y (accumulate != 0: vol += A^T y) -- the exact transpose of the weights
 *   of src/ray_wt_grad.f90:35-89, i.e. recon/sirt.py:61
 *   `sparse.csc_matrix.dot(sparse.csr_matrix.transpose(A), y)`. */
TOMO_API int tomo_adjoint(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol,
                 int accumulate);

/* x-slab form of tomo_adjoint for pipelining the all-reduce of recon/sirt_mpi.py:103 with the back-projection:
 *   the volume is x-major, so the tile columns [xt0, xt1) (width *tile_width voxels, *n_xtiles of them) finalise the
 *   CONTIGUOUS voxel range x in [tile_width*xt0 - 1, tile_width*xt1 - 1) (clipped to [0, nx); the last column runs to nx)
 *   once the columns below it are done.  Adds into d_vol (zero it first).  TOMO_ERR_UNSUPPORTED for poses the tile
 *   kernels decline -- use tomo_adjoint then. */
TOMO_API int tomo_adjoint_xslab_info(tomo_ctx *ctx, int *n_xtiles, int *tile_width);
TOMO_API int tomo_adjoint_xslab(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol, int xt0, int xt1);
/* tomo_forward_xslab: its forward counterpart -- the partial ray sums of the x tile columns [xt0, xt1) of the same tile grid,
 *   which read only the voxels x in [w*xt0 - 1, w*xt1] (w = tile_width): the next iteration's forward projection of a slab
 *   starts as soon as that slab of the volume is final (all-reduced and updated) while the other slabs' all-reduces are still
 *   on the links -- the overlap recon/sirt_mpi.py:92-110 (one blocking Allreduce per iteration) cannot have.  ADDS into d_proj
 *   (zero it before the first slab); TOMO_ERR_UNSUPPORTED for poses the tile kernels decline (nothing launched). */
TOMO_API int tomo_forward_xslab(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_vol, float *d_proj, int xt0, int xt1);

/* tomo_backproject_voxel: the voxel-driven bilinear back-projector src/back_projection.f90:1-34
 *   (voxel_rigid_transformation + voxel_back_bilinear, src/external_back_projection.f90:1-68):
 *   x' = Ry(beta)(Rx(alpha) Rz(phi) c + t), bilinear gather at (x'_x - origin_x, x'_z - origin_z).
 *   NOT the adjoint of tomo_forward (SURVEY 8a quirk vii).  cor_x of the pose is ignored.
 *   d_det [n_proj][ndx][ndz], d_vol overwritten. */
TOMO_API int tomo_backproject_voxel(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_det,
                           float *d_vol);

/* tomo_proj_grad: projection and its 6-DoF pose gradient for ONE projection --
 *   utilities/projection_operators.py:112-122 projection_gradient ->
 *   utilities/ray_voxel_utilities.py:113-170 forward_proj_grad -> src/ray_wt_grad.f90:95-223
 *   trilinear_ray_interp; float32 twin src/projection_gradient.f90:1-79 compute_gradient.
 *   d_proj [n_det], d_grad [6][n_det].  row_order 0: tx,ty,tz,phi,alpha,beta (Python API);
 *   row_order 1: tx,ty,tz,alpha,beta,phi (src/external_forward_projection.f90:56-69). */
TOMO_API int tomo_proj_grad(tomo_ctx *ctx, const double *h_pose, const float *d_vol, float *d_proj,
                   float *d_grad, int row_order);

/* tomo_cost_grad: fused alignment evaluation for n projections (batched form of
 *   utilities/alignment_functions.py:16-37 AlignmentUtilities.cost/.gradient followed by the
 *   reductions of :124 and :146): residual = b - proj, cost = 0.5*||residual||^2,
 *   grad6[k] = sum_r (-dproj/dp_k)[r] * residual[r]   (rows tx,ty,tz,phi,alpha,beta).
 *   d_b [n][n_det]; h_cost [n], h_grad6 [n][6] (host, double).  d_resid may be NULL or [n][n_det]. */
TOMO_API int tomo_cost_grad(tomo_ctx *ctx, const double *h_poses, int n, const float *d_vol, const float *d_b,
                   double *h_cost, double *h_grad6, float *d_resid);
/* Same, with the measured projections picked from a device-resident table instead of being packed per call: pose i is
 *   compared with row h_rows[i] of d_b_table [n_rows_total][n_det] (h_rows NULL = identity).  An alignment loop keeps all
 *   measured projections in HBM and evaluates a changing subset each round (examples/align_rigid.py:40-52 evaluates one
 *   projection at a time); outputs are indexed by i.  d_resid may be NULL or [n][n_det]. */
TOMO_API int tomo_cost_grad_rows(tomo_ctx *ctx, const double *h_poses, int n, const float *d_vol, const float *d_b_table,
                   const int32_t *h_rows, int n_rows_total, double *h_cost, double *h_grad6, float *d_resid);

/* tomo_triplets: COO triplets of one projection with the emission order and values of
 *   src/ray_wt_grad.f90:1-92 trilinear_ray_sparse (ray-major, sample, corner; float64 weights).
 *   Call with h_dat == NULL to get the count.  Intended for N <= 128. */
TOMO_API int tomo_triplets(tomo_ctx *ctx, const double *h_pose, int64_t capacity, int32_t *h_dat, int32_t *h_det,
                  double *h_wts, int64_t *n_inds);

/* tomo_vox_splat / tomo_vox_triplets: the voxel-driven bilinear splat of src/vox_wt_grad.f90 as driven by
 *   utilities/voxel_utilities.py:51-108 (forward_sparse, forward_proj_grad): x' = Ry(beta)(Rx(alpha) Rz(phi) c + t),
 *   u = x'_x - (origin_x - cor_x), v = x'_z - (origin_z - cor_z); 4 bilinear weights per voxel with per-pixel bounds
 *   tests (src/vox_wt_grad.f90:25-49,80-106).  Detector index is x-fastest: fx + ndx*fz (:83) -- unlike the ray path.
 *   tomo_vox_splat: d_img [ndz*ndx] (index z*ndx+x) = bilinear_vox_interp's det_img.ravel(); d_grad (nullable)
 *   [6][ndz*ndx], rows tx,ty,tz,phi,alpha,beta (utilities/voxel_utilities.py:23-48 derivative_rigid), float32.
 *   tomo_vox_triplets: 4 slots per voxel in emission order; h_det[slot] = -1 for an out-of-bounds corner. */
TOMO_API int tomo_vox_splat(tomo_ctx *ctx, const double *h_pose, const double *h_cor3, const float *d_vol, float *d_img, float *d_grad);
TOMO_API int tomo_vox_triplets(tomo_ctx *ctx, const double *h_pose, const double *h_cor3, int32_t *h_det4, float *h_wts4);

/* tomo_phantom_ellipsoids: synthetic test volume (sum of ellipsoid indicator values, clipped at 0) with the
 *   parametrisation of utilities/generate_phantom.py:81-179,194-209: table rows
 *   (A, a, b, c, x0, y0, z0, phi, theta, psi), coordinates linspace(-1,1,n) per axis.  bench/test input only. */
TOMO_API int tomo_phantom_ellipsoids(tomo_ctx *ctx, float *d_vol, int nx, int ny, int nz, const double *h_table, int n_rows);

/* ---------------------------------------------------------------- solver vector kernels
 * (device-resident forms of the numpy lines of recon/sirt.py:33-40,60-73 and recon/cgls.py:56-82) */
TOMO_API int tomo_vec_recip_guard(tomo_ctx *ctx, float *d_v, int64_t n, float thresh, int strict_zero); /* sirt.py:37-40 / sirt_mpi.py:69-72 */
TOMO_API int tomo_vec_fill(tomo_ctx *ctx, float *d_v, int64_t n, float value);
TOMO_API int tomo_vec_residual_scale(tomo_ctx *ctx, const float *d_b, const float *d_ax, const float *d_w,
                            float *d_out, int64_t n, double *h_sumsq); /* out = w*(b-ax); sumsq = ||b-ax||^2  sirt.py:60-61,69 */
TOMO_API int tomo_vec_update(tomo_ctx *ctx, float *d_rec, const float *d_bp, const float *d_v, int64_t n,
                    int positivity, const float *d_gt, double *h_sumsq_err); /* rec += v*bp; clamp; ||gt-rec||^2  sirt.py:63-67,73 */
/* the same update on one x slab of a pipelined iteration: no host synchronisation; ||gt-rec||^2 accumulates over the slabs on the
 * device (zeroed when `first`) and is read back once by tomo_vec_update_acc_fetch */
TOMO_API int tomo_vec_update_acc(tomo_ctx *ctx, float *d_rec, const float *d_bp, const float *d_v, int64_t n,
                        int positivity, const float *d_gt, int first);
TOMO_API int tomo_vec_update_acc_fetch(tomo_ctx *ctx, double *h_sumsq_err);
TOMO_API int tomo_vec_axpy(tomo_ctx *ctx, float *d_y, const float *d_x, float a, int64_t n);          /* y += a*x */
TOMO_API int tomo_vec_xpay(tomo_ctx *ctx, float *d_y, const float *d_x, float a, int64_t n);          /* y = x + a*y   cgls.py:78 */
TOMO_API int tomo_vec_sub(tomo_ctx *ctx, float *d_out, const float *d_a, const float *d_b, int64_t n); /* out = a-b */
TOMO_API int tomo_vec_mul(tomo_ctx *ctx, float *d_y, const float *d_x, int64_t n);                     /* y *= x */
TOMO_API int tomo_vec_dot(tomo_ctx *ctx, const float *d_a, const float *d_b, int64_t n, double *h_dot);
TOMO_API int tomo_vec_diff_sumsq(tomo_ctx *ctx, const float *d_a, const float *d_b, int64_t n, double *h_sumsq);
/* Several scalars of one solver iteration with ONE host synchronisation (round 5): TOMO_N_ACC double accumulators live on the device.
 * tomo_acc_zero clears slots [slot0, slot0 + n); tomo_vec_dot_acc ADDS sum(a*b) (diff 0) or sum((a-b)^2) (diff 1) of a segment to a
 * slot without touching the host -- so gamma = ||A^T r||^2 accumulates over the x slabs of a pipelined back-projection as their
 * reductions arrive (recon/cgls_mpi.py:98-99), and ||A p||^2 and ||b - A p||^2 (:72-76) come from one pass over two slots;
 * tomo_acc_fetch reads slots [slot0, slot0 + n) back -- with `allreduce` != 0 and a communicator, after ONE ncclAllReduce(sum) of
 * exactly those slots on the compute stream: the three host-synchronous scalar allreduces of recon/cgls_mpi.py:75-76,107 become one
 * small device-side collective per reduction point. */
#define TOMO_N_ACC 16
TOMO_API int tomo_acc_zero(tomo_ctx *ctx, int slot0, int n);
TOMO_API int tomo_vec_dot_acc(tomo_ctx *ctx, const float *d_a, const float *d_b, int64_t n, int diff, int slot);
TOMO_API int tomo_acc_fetch(tomo_ctx *ctx, int slot0, int n, int allreduce, double *h_out);

/* ---------------------------------------------------------------- regularised solvers' vector kernels
 * tomo_vec_soft_threshold: out = x - l where x > l, x + l where x < -l, else 0 -- recon/regularized.py:433-440
 *   soft_thresholding (the proximal step of run_lasso_ista :278, its line search :321, run_lasso_fista :375).  out may alias x.
 * tomo_tv_denoise_fista: utilities/tv_denoise.py:98-170 denoise_fista on a float32 volume [nx][ny][nz] resident in HBM (the TV
 *   proximal step of recon/regularized.py:93): argmin 0.5*||im - res||^2 + weight*TV(res) by FISTA on the dual, with the
 *   reference's helpers gradient (:34-59, forward differences, 0 at the last index), div (:20-31), _projector_on_dual (:67-75) and
 *   dual_gap (:78-95).  d_out receives `new` exactly as the reference returns it: the iterate of the LAST GAP CHECK (every
 *   check_gap_frequency iterations; the loop leaves as soon as the dual gap < eps), or a copy of im when niter = 0.
 *   *h_iters = iterations completed, *h_dual_gap = the last gap evaluated (both nullable).  Every axis >= 2.
 * tomo_tv_norm_3d: utilities/tv_denoise.py:62-64 tv_norm_3d = ||gradient(x)||_2 (used at recon/regularized.py:107). */
TOMO_API int tomo_vec_soft_threshold(tomo_ctx *ctx, float *d_out, const float *d_x, int64_t n, float lambda);
TOMO_API int tomo_tv_denoise_fista(tomo_ctx *ctx, const float *d_im, float *d_out, int nx, int ny, int nz, double weight, int niter, double eps,
                          int check_gap_frequency, int *h_iters, double *h_dual_gap);
TOMO_API int tomo_tv_norm_3d(tomo_ctx *ctx, const float *d_x, int nx, int ny, int nz, double *h_norm);
/* tomo_tv_denoise_fista keeps its workspace (7 volumes: the dual fields and two images) in the context between calls, grow-only;
 * tomo_release_workspace frees it (synchronises the stream first).  The next call that needs it allocates it again. */
TOMO_API int tomo_release_workspace(tomo_ctx *ctx);

/* ---------------------------------------------------------------- the assembled CSR matrix, built on the device (csrc/tomo_csr.hip)
 * Replaces utilities/projection_operators.py:54-76: per-projection triplets of src/ray_wt_grad.f90:1-92 (float64 weights cast to
 * `precision`, row = detector index + iproj * n_det), optional voxel mask (d_mask: device float[n_vox], 0 = masked out; NULL = none;
 * when EVERY entry is masked the reference keeps them all with weight 0, :63-65 -- so does this), COO -> CSR with duplicates summed,
 * explicit zeros kept, column indices sorted.  tomo_csr_assemble builds the matrix in device memory held by the context and returns its
 * number of stored entries; tomo_csr_fetch copies data (float32 or float64 by precision_bits) / indices (int32) / indptr (int64,
 * n_proj * n_det + 1) into the caller's host arrays and frees the device copy; with all three pointers NULL it only frees it (a
 * caller that finds the matrix larger than it is willing to download).  Small volumes only (the reference's matrix at
 * 128^3 x 64 is 4.5 GB); TOMO_ERR_UNSUPPORTED beyond int32 indices or 2^31 triplets. */
TOMO_API int tomo_csr_assemble(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_mask, int precision_bits, int64_t *h_nnz);
TOMO_API int tomo_csr_fetch(tomo_ctx *ctx, void *h_data, int32_t *h_indices, int64_t *h_indptr);

/* ---------------------------------------------------------------- array-level twins of the two f2py routines (csrc/tomo_f2py.hip)
 * For a binding one level BELOW the operator API: utilities/ray_voxel_utilities.py:103,164 call
 *   dat_inds, det_inds, wts, n_inds = ray_wt_grad.trilinear_ray_sparse(floor_points, w_floor, nx, ny, nz, n_rays, n_points)   src/ray_wt_grad.f90:1-92
 *   det_img, grad_det_img = ray_wt_grad.trilinear_ray_interp(floor_points, w_floor, nx, ny, nz, n_rays, n_points, recon, step, der)   :95-223
 * with sample tables built in numpy.  Same arguments here, as HOST arrays in the Fortran (column-major) layout f2py passes:
 * floor_points int32 (3, n_rays, n_points), w_floor float64 (3, n_rays, n_points), recon float64 [nx*ny*nz], step float64
 * (n_rays, n_points), der float64 (9, 3, n_rays); outputs det_img float64 [n_rays], grad_det_img float64 (6, n_rays) rows
 * tx, ty, tz, phi, alpha, beta; dat_inds / det_inds int32 and wts float64 of length 8 * n_rays * n_points, pre-filled with -999 as the
 * reference does (:15-17) and written in the reference's emission order, *h_n_inds = entries written.  float64 arithmetic in the
 * reference's order.  The compatibility surface, not the fast path (36 bytes of table per sample). */
TOMO_API int tomo_trilinear_ray_interp(tomo_ctx *ctx, const int32_t *h_floor_points, const double *h_w_floor, int nx, int ny, int nz, int n_rays,
                              int n_points, const double *h_recon, const double *h_step, const double *h_der, double *h_det_img,
                              double *h_grad_det_img);
TOMO_API int tomo_trilinear_ray_sparse(tomo_ctx *ctx, const int32_t *h_floor_points, const double *h_w_floor, int nx, int ny, int nz, int n_rays,
                              int n_points, int32_t *h_dat_inds, int32_t *h_det_inds, double *h_wts, int32_t *h_n_inds);

/* The other f2py module, src.vox_wt_grad (round 6), as utilities/voxel_utilities.py:69-75,98-104 calls it:
 *   det_img, grad_det_img = vox_wt_grad.bilinear_vox_interp(n_vox, floor_x, floor_z, alpha_x, alpha_z, rec, ndim_x, ndim_z, der_points)   src/vox_wt_grad.f90:1-55
 *   dat_inds, det_inds, wts, n_inds = vox_wt_grad.bilinear_sparse(n_vox, floor_x, floor_z, alpha_x, alpha_z, ndim_x, ndim_z)              src/vox_wt_grad.f90:58-112
 * HOST arrays as f2py passes them: floor_x, floor_z int32 [n_vox] (0-based floor pixel; the Fortran adds 1, :21-22), alpha_x, alpha_z, rec
 * float32 [n_vox], der_points float32 Fortran (6, 3, n_vox) (element (q, a, i) at q + 6 a + 18 i; columns 1 and 3 are read, :27-28).  Outputs:
 * det_img float32 Fortran (ndim_z, ndim_x), grad_det_img float32 Fortran (6, ndim_z, ndim_x), rows tx, ty, tz, phi, alpha, beta; dat_inds /
 * det_inds int32 and wts float32 of length 4 * n_vox, pre-filled with -999 (:73-75), emission order voxel-major then (fx,fz), (fx+1,fz),
 * (fx,fz+1), (fx+1,fz+1) with per-pixel bounds tests, det index x-fastest fx + ndim_x fz (:83).  Single precision in the reference's operation
 * ORDER, including the order of the additions into a pixel (voxel order): bit-identical to the f2py module (tests/golden/g13). */
TOMO_API int tomo_bilinear_vox_interp(tomo_ctx *ctx, int n_vox, const int32_t *h_floor_x, const int32_t *h_floor_z, const float *h_alpha_x,
                             const float *h_alpha_z, const float *h_rec, int ndim_x, int ndim_z, const float *h_der_points, float *h_det_img,
                             float *h_grad_det_img);
TOMO_API int tomo_bilinear_sparse(tomo_ctx *ctx, int n_vox, const int32_t *h_floor_x, const int32_t *h_floor_z, const float *h_alpha_x, const float *h_alpha_z,
                         int ndim_x, int ndim_z, int32_t *h_dat_inds, int32_t *h_det_inds, float *h_wts, int32_t *h_n_inds);

/* ---------------------------------------------------------------- multi-GPU (RCCL over xGMI)
 * Replaces mpi4py COMM_WORLD Allreduce(SUM) of recon/sirt_mpi.py:68,103 and recon/cgls_mpi.py:55,98
 * and the scalar allreduce of recon/sirt_mpi.py:110. */
#define TOMO_COMM_ID_BYTES 128
TOMO_API int tomo_comm_get_unique_id(void *h_id128);
TOMO_API int tomo_comm_init(tomo_ctx *ctx, const void *h_id128, int n_ranks, int rank);
TOMO_API int tomo_comm_destroy(tomo_ctx *ctx);
TOMO_API int tomo_allreduce_sum_f32(tomo_ctx *ctx, float *d_buf, int64_t n);     /* in place, on ctx stream */
/* asynchronous form: the all-reduce runs on the ctx's communication stream after everything queued so far on the compute
 * stream; tomo_comm_join makes the compute stream wait for every asynchronous all-reduce issued before it. */
TOMO_API int tomo_allreduce_sum_f32_async(tomo_ctx *ctx, float *d_buf, int64_t n);
TOMO_API int tomo_comm_join(tomo_ctx *ctx);
/* the compute stream waits for the OLDEST asynchronous all-reduce / reduce-scatter not yet waited for (issue order); no-op when none is pending */
TOMO_API int tomo_comm_wait_next(tomo_ctx *ctx);
/* Sharding the volume-sized vector work of an iteration (round 4): where recon/sirt_mpi.py:101-110 all-reduces the update and every
 * rank then applies it to its full replica, a rank may instead receive only the sum of ITS 1/P of a segment
 * (tomo_reduce_scatter_sum_f32_async: in place on d_buf[0 .. n_ranks * n_per_rank), rank r's sums land in d_buf + r * n_per_rank,
 * the other pieces are left undefined), update that piece of the reconstruction, and hand the pieces round
 * (tomo_allgather_f32_async: in place, rank r contributes d_buf + r * n_per_rank).  Same bytes on the links as the all-reduce (a ring
 * all-reduce IS these two phases), 1/P of the update / scaling / error-sum work per rank.  Both run on the communication stream like
 * tomo_allreduce_sum_f32_async; all-gathers have a wait queue of their own (tomo_comm_wait_next_gather), so a caller can wait for
 * "reduce-scatter of slab s + 1" before "all-gather of slab s" although they were issued the other way round. */
TOMO_API int tomo_reduce_scatter_sum_f32_async(tomo_ctx *ctx, float *d_buf, int64_t n_per_rank);
TOMO_API int tomo_allgather_f32_async(tomo_ctx *ctx, float *d_buf, int64_t n_per_rank);
TOMO_API int tomo_comm_wait_next_gather(tomo_ctx *ctx);
TOMO_API int tomo_allreduce_sum_f64_host(tomo_ctx *ctx, double *h_vals, int n);  /* small host scalars */
TOMO_API int tomo_allreduce_max_f64_host(tomo_ctx *ctx, double *h_vals, int n);

/* ---------------------------------------------------------------- measurement (HIP events on the ctx stream) */
TOMO_API int tomo_timer_start(tomo_ctx *ctx);
TOMO_API int tomo_timer_stop(tomo_ctx *ctx, float *h_ms);            /* synchronises */
TOMO_API int tomo_profile_enable(tomo_ctx *ctx, int on);              /* bracket every kernel launch with events */
TOMO_API int tomo_profile_reset(tomo_ctx *ctx);
TOMO_API int tomo_profile_get(tomo_ctx *ctx, const char *kernel, int64_t *n_launch, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* TOMO_H_ */
