// tomo_reg.hip -- the vector kernels of the reference's regularised solvers: the soft-threshold of ISTA /
// LASSO (recon/regularized.py:433-440, used at :278,321,375) and the total-variation proximal step of TV-FISTA
// (utilities/tv_denoise.py:98-170 denoise_fista with its helpers gradient :34-59, div :20-31, _projector_on_dual :67-75,
// dual_gap :78-95; called from recon/regularized.py:93).  The solvers' drivers themselves are out of scope; these
// are the device-resident pieces they would call between two projector applications, so a regularised iteration never has to
// bring the volume back over PCIe.  HBM-streaming stencil kernels: lanes along z (the contiguous axis), +-1 neighbours in
// x / y come from rows the same work-group's neighbours touch (L2).
#include <algorithm>
#include <vector>

#include "tomo_ctx.h"

// out = sign(x) * max(|x| - lambda, 0) written as the reference does: x - l where x > l, x + l where x < -l, else 0
__global__ __launch_bounds__(256) void k_soft_threshold(float *__restrict__ out, const float *__restrict__ x, int64_t n, float l)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        out[i] = v > l ? v - l : (v < -l ? v + l : 0.f);
    }
}

extern "C" int tomo_vec_soft_threshold(tomo_ctx *ctx, float *d_out, const float *d_x, int64_t n, float lambda)
{
    if (!ctx) return tomo_fail(nullptr, TOMO_ERR_ARG, "null ctx");
    if (n < 0 || (n > 0 && (!d_out || !d_x))) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_vec_soft_threshold: bad args");
    if (n == 0) return TOMO_OK;
    TOMO_LAUNCH(ctx, "k_soft_threshold", k_soft_threshold, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, d_out, d_x, n, lambda);
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// TV-FISTA.  Volume [nx][ny][nz], z fastest.  Grid: (z chunks of 256, ny, nx).
// ------------------------------------------------------------------------------------------------
struct TvDims { int nx, ny, nz; int64_t sy, sx; };

// divergence of a 3-component field with the reference's boundary rule (utilities/tv_denoise.py:26-30):
//   d_a[i] = (i < n_a - 1 ? g_a[i] : 0) - (i > 0 ? g_a[i - 1] : 0), summed over the axes in the order x, y, z (adds in the
//   order the reference's in-place += / -= apply them)
__device__ __forceinline__ float tv_div_at(const float *gx, const float *gy, const float *gz, int64_t i, int ix, int iy, int iz, const TvDims &d)
{
    float r = 0.f;
    if (ix < d.nx - 1) r += gx[i];
    if (ix > 0) r -= gx[i - d.sx];
    if (iy < d.ny - 1) r += gy[i];
    if (iy > 0) r -= gy[i - d.sy];
    if (iz < d.nz - 1) r += gz[i];
    if (iz > 0) r -= gz[i - 1];
    return r;
}

// err = weight * div(grad_aux) - im                                                       tv_denoise.py:151
__global__ __launch_bounds__(256) void k_tv_error(const float *__restrict__ ax, const float *__restrict__ ay, const float *__restrict__ az,
                                                  const float *__restrict__ im, float *__restrict__ err, TvDims d, float weight)
{
    const int iz = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y, ix = blockIdx.z;
    if (iz >= d.nz) return;
    const int64_t i = (int64_t)ix * d.sx + (int64_t)iy * d.sy + iz;
    err[i] = __fsub_rn(__fmul_rn(weight, tv_div_at(ax, ay, az, i, ix, iy, iz, d)), im[i]);
}

// grad_tmp = gradient(err) / (factor * weight); grad_aux += grad_tmp; project on the unit ball; FISTA combination
//                                                                                          tv_denoise.py:152-158, :34-59, :67-75
__global__ __launch_bounds__(256) void k_tv_update(float *__restrict__ ax, float *__restrict__ ay, float *__restrict__ az,
                                                   float *__restrict__ px, float *__restrict__ py, float *__restrict__ pz,
                                                   const float *__restrict__ err, TvDims d, float c, float one_tf, float tf)
{
    const int iz = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y, ix = blockIdx.z;
    if (iz >= d.nz) return;
    const int64_t i = (int64_t)ix * d.sx + (int64_t)iy * d.sy + iz;
    const float e0 = err[i];
    const float gx = ix < d.nx - 1 ? __fsub_rn(err[i + d.sx], e0) : 0.f;      // forward differences, 0 at the last index of an axis
    const float gy = iy < d.ny - 1 ? __fsub_rn(err[i + d.sy], e0) : 0.f;
    const float gz = iz < d.nz - 1 ? __fsub_rn(err[i + 1], e0) : 0.f;
    const float a0 = __fadd_rn(ax[i], __fmul_rn(gx, c)), a1 = __fadd_rn(ay[i], __fmul_rn(gy, c)), a2 = __fadd_rn(az[i], __fmul_rn(gz, c));
    const float nrm = fmaxf(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(a0, a0), __fmul_rn(a1, a1)), __fmul_rn(a2, a2))), 1.f);
    const float p0 = __fdiv_rn(a0, nrm), p1 = __fdiv_rn(a1, nrm), p2 = __fdiv_rn(a2, nrm);
    ax[i] = __fsub_rn(__fmul_rn(one_tf, p0), __fmul_rn(tf, px[i]));             // (1 + t_factor) * grad_tmp - t_factor * grad_im   :158
    ay[i] = __fsub_rn(__fmul_rn(one_tf, p1), __fmul_rn(tf, py[i]));
    az[i] = __fsub_rn(__fmul_rn(one_tf, p2), __fmul_rn(tf, pz[i]));
    px[i] = p0; py[i] = p1; pz[i] = p2;                                            // grad_im = grad_tmp                               :159
}

__device__ __forceinline__ void tv_block_sum(double v, double *dst)
{
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, sh[0] + sh[1] + sh[2] + sh[3]);
    __syncthreads();
}

// gap = weight * div(grad_im); new = im - gap; sums of gap^2, new^2, im^2                  tv_denoise.py:162-163, :84,94
__global__ __launch_bounds__(256) void k_tv_new(const float *__restrict__ px, const float *__restrict__ py, const float *__restrict__ pz,
                                                const float *__restrict__ im, float *__restrict__ out, TvDims d, float weight, double *__restrict__ red)
{
    const int iz = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y, ix = blockIdx.z;
    double s_gap = 0.0, s_new = 0.0, s_im = 0.0;
    if (iz < d.nz) {
        const int64_t i = (int64_t)ix * d.sx + (int64_t)iy * d.sy + iz;
        const float gap = __fmul_rn(weight, tv_div_at(px, py, pz, i, ix, iy, iz, d));
        const float v = im[i], nw = __fsub_rn(v, gap);
        out[i] = nw;
        s_gap = (double)gap * gap; s_new = (double)nw * nw; s_im = (double)v * v;
    }
    tv_block_sum(s_gap, red + 0);
    tv_block_sum(s_new, red + 1);
    tv_block_sum(s_im, red + 2);
}

// sum over voxels of sqrt(gx^2 + gy^2 + gz^2) (ISO = 1: the isotropic TV of dual_gap, tv_denoise.py:85-92) or of gx^2 + gy^2 + gz^2
// (ISO = 0: tv_norm_3d = ||gradient(x)||_2, :62-64), forward differences with 0 at the last index
template <int ISO>
__global__ __launch_bounds__(256) void k_tv_norm(const float *__restrict__ x, TvDims d, double *__restrict__ red)
{
    const int iz = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y, ix = blockIdx.z;
    double s = 0.0;
    if (iz < d.nz) {
        const int64_t i = (int64_t)ix * d.sx + (int64_t)iy * d.sy + iz;
        const float v = x[i];
        const float gx = ix < d.nx - 1 ? x[i + d.sx] - v : 0.f, gy = iy < d.ny - 1 ? x[i + d.sy] - v : 0.f, gz = iz < d.nz - 1 ? x[i + 1] - v : 0.f;
        const float q = __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
        s = ISO ? (double)__fsqrt_rn(q) : (double)q;
    }
    tv_block_sum(s, red);
}

static int tv_dims(tomo_ctx *ctx, int nx, int ny, int nz, TvDims &d, dim3 &grid)
{
    if (nx < 2 || ny < 2 || nz < 2) return tomo_fail(ctx, TOMO_ERR_ARG, "tv: every axis needs at least 2 voxels (the reference's div indexes [-2])");
    if (ny > 65535 || nx > 65535) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tv: nx, ny <= 65535");
    d.nx = nx; d.ny = ny; d.nz = nz; d.sy = nz; d.sx = (int64_t)ny * nz;
    grid = dim3((nz + 255) / 256, ny, nx);
    return TOMO_OK;
}

extern "C" int tomo_tv_norm_3d(tomo_ctx *ctx, const float *d_x, int nx, int ny, int nz, double *h_norm)
{
    if (!ctx || !d_x || !h_norm) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_tv_norm_3d: bad args");
    TvDims d;
    dim3 grid;
    int rc = tv_dims(ctx, nx, ny, nz, d, grid);
    if (rc) return rc;
    rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, sizeof(double), ctx->stream));
    TOMO_LAUNCH(ctx, "k_tv_norm", k_tv_norm<0>, grid, dim3(256), 0, d_x, d, ctx->d_red);
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red, ctx->d_red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *h_norm = sqrt(ctx->h_red[0]);
    return TOMO_OK;
}

extern "C" int tomo_tv_denoise_fista(tomo_ctx *ctx, const float *d_im, float *d_out, int nx, int ny, int nz, double weight, int niter, double eps,
                                     int check_gap_frequency, int *h_iters, double *h_dual_gap)
{
    if (!ctx || !d_im || !d_out || niter < 0 || check_gap_frequency < 1 || !(weight > 0.0)) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_tv_denoise_fista: bad args");
    TvDims d;
    dim3 grid;
    int rc = tv_dims(ctx, nx, ny, nz, d, grid);
    if (rc) return rc;
    rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    const size_t n = (size_t)nx * ny * nz;
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    // grad_aux[3], grad_im[3], err: kept in the context across calls (grow-only, like d_stage / d_red) -- a proximal step per
    // outer iteration of a regularised solver used to pay a 7 n hipMalloc + hipFree (28 GB at 1024^3, each an implicit device sync)
    rc = tomo_ensure_ws(ctx, 7 * n);
    if (rc) return rc;
    float *ws = ctx->d_ws;
    float *ax = ws, *ay = ws + n, *az = ws + 2 * n, *px = ws + 3 * n, *py = ws + 4 * n, *pz = ws + 5 * n, *err = ws + 6 * n;
    double dgap = 0.0;
    int i = 0;
    if (h_iters) *h_iters = 0;
    if (h_dual_gap) *h_dual_gap = 0.0;
    TOMO_HIP(ctx, hipMemsetAsync(ws, 0, 6 * n * sizeof(float), ctx->stream));                                        // :144-145
    TOMO_HIP(ctx, hipMemcpyAsync(d_out, d_im, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));           // new = im.copy()  :148
    const float w = (float)weight, c = (float)(1.0 / (12.0 * weight));                                               // factor 12 for 3-D  :139-142,153
    double t = 1.0;
    while (i < niter) {                                                                                              // :149
        TOMO_LAUNCH(ctx, "k_tv_error", k_tv_error, grid, dim3(256), 0, (const float *)ax, (const float *)ay, (const float *)az, d_im, err, d, w);
        const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t * t)), t_factor = (t - 1.0) / t_new;                   // :156-157
        TOMO_LAUNCH(ctx, "k_tv_update", k_tv_update, grid, dim3(256), 0, ax, ay, az, px, py, pz, (const float *)err, d, c, (float)(1.0 + t_factor), (float)t_factor);
        t = t_new;
        if (i % check_gap_frequency == 0) {                                                                          // :161-166
            TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, 4 * sizeof(double), ctx->stream));
            TOMO_LAUNCH(ctx, "k_tv_new", k_tv_new, grid, dim3(256), 0, (const float *)px, (const float *)py, (const float *)pz, d_im, d_out, d, w, ctx->d_red);
            TOMO_LAUNCH(ctx, "k_tv_norm", k_tv_norm<1>, grid, dim3(256), 0, (const float *)d_out, d, ctx->d_red + 3);
            TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red, ctx->d_red, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const double s_gap = ctx->h_red[0], s_new = ctx->h_red[1], im_norm = ctx->h_red[2], tv_new = 2.0 * weight * ctx->h_red[3];
            dgap = im_norm > 0.0 ? 0.5 / im_norm * (s_gap + tv_new - im_norm + s_new) : 0.0;                        // :93-95
            if (dgap < eps) break;                                                                                   // :165-166 (i is not advanced)
        }
        ++i;
    }
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_iters) *h_iters = i;
    if (h_dual_gap) *h_dual_gap = dgap;
    return TOMO_OK;
}
