// tomo_f2py.hip -- array-level twins of the reference's two f2py routines, for a binding ONE LEVEL BELOW the operator API:
//   tomo_trilinear_ray_interp   <-  src/ray_wt_grad.f90:95-223   trilinear_ray_interp(floor_points, w_floor, nx, ny, nz, n_rays, n_points,
//                                                                                      recon, step, der) -> det_img, grad_det_img
//   tomo_trilinear_ray_sparse   <-  src/ray_wt_grad.f90:1-92     trilinear_ray_sparse(floor_points, w_floor, nx, ny, nz, n_rays, n_points)
//                                                                                      -> dat_inds, det_inds, wts, n_inds
// The reference's utilities/ray_voxel_utilities.py:103,164 calls these with the (3, n_rays, n_points) sample tables it has built in
// numpy; `tomography_alignment_amd/src/ray_wt_grad.py` exposes them under the f2py module's name and signatures, so that file binds
// this library without an edit.  HOST arrays in the Fortran (column-major) layout f2py hands to the routines, float64 / int32 as the
// reference's; the arithmetic is the reference's float64 arithmetic in the reference's order (unfused: __dmul_rn / __dadd_rn), one ray
// per thread.  This is the compatibility surface, not the fast path: the tables are 36 bytes per sample (77 GB per projection at
// 1024^3), which is what the lattice kernels of tomo_project.hip exist to avoid.
#include "tomo_ctx.h"

namespace {

// element (a, r, p) of a Fortran array of shape (3, n_rays, n_points)
__device__ __forceinline__ size_t at3(int a, int r, int p, int n_rays) { return (size_t)a + 3u * ((size_t)r + (size_t)n_rays * p); }

struct Corner { int x, y, z; double wx, wy, wz; int sx, sy, sz; };      // s*: -1 for a floor factor, +1 for a ceil factor (:146-148 ... :216-218)

// the eight corners of sample (r, p) in the reference's order (fff, ffc, fcf, fcc, cff, cfc, ccf, ccc: :35-89 / :142-220), 0-based cells
__device__ __forceinline__ void corners_of(const int32_t *fp, const double *wf, int r, int p, int n_rays, Corner c[8])
{
    const int fx = fp[at3(0, r, p, n_rays)], fy = fp[at3(1, r, p, n_rays)], fz = fp[at3(2, r, p, n_rays)];
    const double wfx = wf[at3(0, r, p, n_rays)], wfy = wf[at3(1, r, p, n_rays)], wfz = wf[at3(2, r, p, n_rays)];
    const double wcx = __dsub_rn(1.0, wfx), wcy = __dsub_rn(1.0, wfy), wcz = __dsub_rn(1.0, wfz);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ax = k >> 2, ay = (k >> 1) & 1, az = k & 1;
        c[k].x = fx + ax; c[k].y = fy + ay; c[k].z = fz + az;
        c[k].wx = ax ? wcx : wfx; c[k].wy = ay ? wcy : wfy; c[k].wz = az ? wcz : wfz;
        c[k].sx = ax ? 1 : -1; c[k].sy = ay ? 1 : -1; c[k].sz = az ? 1 : -1;
    }
}

__global__ __launch_bounds__(64) void k_f2py_interp(const int32_t *__restrict__ fp, const double *__restrict__ wf, int nx, int ny, int nz, int n_rays,
                                                    int n_points, const double *__restrict__ recon, const double *__restrict__ step,
                                                    const double *__restrict__ der, double *__restrict__ det_img, double *__restrict__ grad)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rays) return;
    double gt[9][3];                                     // g_temp = der(:, :, r)                      :134
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) gt[k][a] = der[(size_t)k + 9u * a + 27u * (size_t)r];
    double img = 0.0, gr[6] = {0, 0, 0, 0, 0, 0};
    for (int p = 0; p < n_points; ++p) {
        const double st = step[(size_t)r + (size_t)n_rays * p];
        double g[6][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g[0][a] = gt[0][a]; g[1][a] = gt[1][a]; g[2][a] = gt[2][a];
            g[3][a] = __dadd_rn(gt[3][a], __dmul_rn(st, gt[6][a]));      // :139-141
            g[4][a] = __dadd_rn(gt[4][a], __dmul_rn(st, gt[7][a]));
            g[5][a] = __dadd_rn(gt[5][a], __dmul_rn(st, gt[8][a]));
        }
        Corner c[8];
        corners_of(fp, wf, r, p, n_rays, c);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (c[k].x < 0 || c[k].x >= nx || c[k].y < 0 || c[k].y >= ny || c[k].z < 0 || c[k].z >= nz) continue;     // per corner  :142
            const double v = recon[((size_t)c[k].x * ny + c[k].y) * nz + c[k].z];
            const double wt = __dmul_rn(__dmul_rn(c[k].wx, c[k].wy), c[k].wz);                                        // :144
            img = __dadd_rn(img, __dmul_rn(v, wt));                                                                 // :145
            // g1 = -+ wy wz rec g(:,1) etc. (:146-148): the products are formed left to right as the Fortran writes them
            const double a1 = __dmul_rn(__dmul_rn((c[k].sx < 0 ? -c[k].wy : c[k].wy), c[k].wz), v);
            const double a2 = __dmul_rn(__dmul_rn((c[k].sy < 0 ? -c[k].wx : c[k].wx), c[k].wz), v);
            const double a3 = __dmul_rn(__dmul_rn((c[k].sz < 0 ? -c[k].wx : c[k].wx), c[k].wy), v);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                gr[q] = __dadd_rn(gr[q], __dadd_rn(__dadd_rn(__dmul_rn(a1, g[q][0]), __dmul_rn(a2, g[q][1])), __dmul_rn(a3, g[q][2])));   // :149
        }
    }
    det_img[r] = img;
#pragma unroll
    for (int q = 0; q < 6; ++q) grad[(size_t)q + 6u * (size_t)r] = gr[q];
}

// pass 1: in-bounds corners per ray; pass 2 (after a prefix sum): the triplets in the reference's emission order (ray, point, corner)
template <bool FILL>
__global__ __launch_bounds__(64) void k_f2py_sparse(const int32_t *__restrict__ fp, const double *__restrict__ wf, int nx, int ny, int nz, int n_rays,
                                                    int n_points, int64_t *__restrict__ count, const int64_t *__restrict__ start,
                                                    int32_t *__restrict__ dat, int32_t *__restrict__ det, double *__restrict__ wts)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rays) return;
    int64_t n = FILL ? start[r] : 0;
    for (int p = 0; p < n_points; ++p) {
        Corner c[8];
        corners_of(fp, wf, r, p, n_rays, c);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (c[k].x < 0 || c[k].x >= nx || c[k].y < 0 || c[k].y >= ny || c[k].z < 0 || c[k].z >= nz) continue;
            if (FILL) {
                det[n] = r;                                                               // :37
                dat[n] = (c[k].x * ny + c[k].y) * nz + c[k].z;                            // :38
                wts[n] = __dmul_rn(__dmul_rn(c[k].wx, c[k].wy), c[k].wz);                 // :39
            }
            ++n;
        }
    }
    if (!FILL) count[r] = n;
}

struct DevBuf {      // scoped device allocation
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

#define F2PY_ALLOC(buf, bytes) TOMO_HIP(ctx, hipMalloc(&(buf).p, (bytes) ? (bytes) : 1))

extern "C" int tomo_trilinear_ray_interp(tomo_ctx *ctx, const int32_t *h_floor_points, const double *h_w_floor, int nx, int ny, int nz, int n_rays,
                                         int n_points, const double *h_recon, const double *h_step, const double *h_der, double *h_det_img,
                                         double *h_grad_det_img)
{
    if (!ctx || !h_floor_points || !h_w_floor || !h_recon || !h_step || !h_der || !h_det_img || !h_grad_det_img || nx <= 0 || ny <= 0 || nz <= 0 ||
        n_rays < 0 || n_points < 0)
        return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_trilinear_ray_interp: bad args");
    if (n_rays == 0) return TOMO_OK;
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ns = (size_t)n_rays * n_points, nv = (size_t)nx * ny * nz;
    DevBuf fp, wf, rec, st, der, img, gr;
    F2PY_ALLOC(fp, 3 * ns * sizeof(int32_t)); F2PY_ALLOC(wf, 3 * ns * sizeof(double)); F2PY_ALLOC(rec, nv * sizeof(double));
    F2PY_ALLOC(st, ns * sizeof(double)); F2PY_ALLOC(der, 27 * (size_t)n_rays * sizeof(double));
    F2PY_ALLOC(img, (size_t)n_rays * sizeof(double)); F2PY_ALLOC(gr, 6 * (size_t)n_rays * sizeof(double));
    TOMO_HIP(ctx, hipMemcpyAsync(fp.p, h_floor_points, 3 * ns * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(wf.p, h_w_floor, 3 * ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(rec.p, h_recon, nv * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(st.p, h_step, ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(der.p, h_der, 27 * (size_t)n_rays * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TOMO_LAUNCH(ctx, "k_f2py_interp", k_f2py_interp, dim3((n_rays + 63) / 64), dim3(64), 0, (const int32_t *)fp.p, (const double *)wf.p, nx, ny, nz, n_rays,
                n_points, (const double *)rec.p, (const double *)st.p, (const double *)der.p, (double *)img.p, (double *)gr.p);
    TOMO_HIP(ctx, hipMemcpyAsync(h_det_img, img.p, (size_t)n_rays * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(h_grad_det_img, gr.p, 6 * (size_t)n_rays * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_trilinear_ray_sparse(tomo_ctx *ctx, const int32_t *h_floor_points, const double *h_w_floor, int nx, int ny, int nz, int n_rays,
                                         int n_points, int32_t *h_dat_inds, int32_t *h_det_inds, double *h_wts, int32_t *h_n_inds)
{
    if (!ctx || !h_floor_points || !h_w_floor || !h_dat_inds || !h_det_inds || !h_wts || !h_n_inds || nx <= 0 || ny <= 0 || nz <= 0 || n_rays < 0 ||
        n_points < 0)
        return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_trilinear_ray_sparse: bad args");
    const size_t ns = (size_t)n_rays * n_points;
    if (8 * ns >= ((size_t)1 << 31)) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_trilinear_ray_sparse: 8 * n_rays * n_points must fit an int32 (as the reference's outputs do)");
    for (size_t i = 0; i < 8 * ns; ++i) { h_dat_inds[i] = -999; h_det_inds[i] = -999; h_wts[i] = -999.0; }      // :15-17
    *h_n_inds = 0;
    if (ns == 0) return TOMO_OK;
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf fp, wf, cnt, dat, det, wts;
    F2PY_ALLOC(fp, 3 * ns * sizeof(int32_t)); F2PY_ALLOC(wf, 3 * ns * sizeof(double)); F2PY_ALLOC(cnt, (size_t)n_rays * sizeof(int64_t));
    TOMO_HIP(ctx, hipMemcpyAsync(fp.p, h_floor_points, 3 * ns * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(wf.p, h_w_floor, 3 * ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TOMO_LAUNCH(ctx, "k_f2py_sparse", k_f2py_sparse<false>, dim3((n_rays + 63) / 64), dim3(64), 0, (const int32_t *)fp.p, (const double *)wf.p, nx, ny, nz,
                n_rays, n_points, (int64_t *)cnt.p, (const int64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr);
    std::vector<int64_t> start((size_t)n_rays);
    TOMO_HIP(ctx, hipMemcpyAsync(start.data(), cnt.p, (size_t)n_rays * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int64_t total = 0;
    for (int r = 0; r < n_rays; ++r) { const int64_t c = start[(size_t)r]; start[(size_t)r] = total; total += c; }      // emission order: ray-major
    if (total > 0) {
        F2PY_ALLOC(dat, (size_t)total * sizeof(int32_t)); F2PY_ALLOC(det, (size_t)total * sizeof(int32_t)); F2PY_ALLOC(wts, (size_t)total * sizeof(double));
        TOMO_HIP(ctx, hipMemcpyAsync(cnt.p, start.data(), (size_t)n_rays * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        TOMO_LAUNCH(ctx, "k_f2py_sparse", k_f2py_sparse<true>, dim3((n_rays + 63) / 64), dim3(64), 0, (const int32_t *)fp.p, (const double *)wf.p, nx, ny, nz,
                    n_rays, n_points, (int64_t *)nullptr, (const int64_t *)cnt.p, (int32_t *)dat.p, (int32_t *)det.p, (double *)wts.p);
        TOMO_HIP(ctx, hipMemcpyAsync(h_dat_inds, dat.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        TOMO_HIP(ctx, hipMemcpyAsync(h_det_inds, det.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        TOMO_HIP(ctx, hipMemcpyAsync(h_wts, wts.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *h_n_inds = (int32_t)total;
    return TOMO_OK;
}
