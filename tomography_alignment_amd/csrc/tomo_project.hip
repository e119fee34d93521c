// tomo_project.hip -- gfx950 kernels of the ray-driven projection hot path.
//
// Work decomposition (all kernels): one ray per work-item, the 64 lanes of a wavefront run along
// detector-z == the memory-fastest volume axis (src/ray_wt_grad.f90:38), so every corner row a wave
// touches is one contiguous ~256-B run; the 4 waves of a 256-thread work-group take 4 adjacent
// detector-x rays, whose corner rows overlap in the CU's L1.  No MFMA: this is gather/scatter ray
// marching, bounded by L1/LDS/HBM traffic, not a dense contraction.
//
// Kernels                              replaces (reference)
//   k_pad / k_unpad                    -- (zero-halo staging of the volume; removes the 8 per-corner
//                                         bounds tests of src/ray_wt_grad.f90:35-89 from the loop)
//   k_fwd_v1 / k_fwd_v2                A.x : recon/sirt.py:59 ; src/forward_projection.f90:1-68
//   k_adj_v1                           A^T.y : recon/sirt.py:61 (global float atomics, reference form)
//   k_bp_voxel                         src/back_projection.f90:1-34
//   k_proj_grad<FUSED>                 src/ray_wt_grad.f90:95-223 ; src/projection_gradient.f90:1-79 ;
//                                      utilities/alignment_functions.py:16-37,124,146 (FUSED)
#include <limits.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tomo_ctx.h"

// ------------------------------------------------------------------------------------------------
// wave helpers (64 lanes)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int64_t readfirstlane_i64(int64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------------
// zero-halo staging
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pad(const float *__restrict__ vol, float *__restrict__ vp, TomoGeomC g)
{
    const int row = blockIdx.x;            // ix*ny + iy
    const int ix = row / g.ny, iy = row - ix * g.ny;
    const float *src = vol + (size_t)row * g.nz;
    float *dst = vp + ((size_t)(ix + TOMO_HALO) * g.nyp + (iy + TOMO_HALO)) * g.nzp + TOMO_HALO;
    for (int z = threadIdx.x; z < g.nz; z += blockDim.x) dst[z] = src[z];
}

__global__ __launch_bounds__(256) void k_unpad(float *__restrict__ vol, const float *__restrict__ vp, TomoGeomC g, int accumulate)
{
    const int row = blockIdx.x;
    const int ix = row / g.ny, iy = row - ix * g.ny;
    float *dst = vol + (size_t)row * g.nz;
    const float *src = vp + ((size_t)(ix + TOMO_HALO) * g.nyp + (iy + TOMO_HALO)) * g.nzp + TOMO_HALO;
    if (accumulate)
        for (int z = threadIdx.x; z < g.nz; z += blockDim.x) dst[z] += src[z];
    else
        for (int z = threadIdx.x; z < g.nz; z += blockDim.x) dst[z] = src[z];
}

static int stage_volume(tomo_ctx *ctx, const float *d_vol)
{
    const TomoGeomC &g = ctx->g;
    if (ctx->reuse_staged && !ctx->halo_dirty && ctx->staged_src == (const void *)d_vol) return TOMO_OK;   // caller vouches: unchanged
    ctx->staged_src = (const void *)d_vol;
    if (ctx->halo_dirty) {
        TOMO_HIP(ctx, hipMemsetAsync(ctx->d_volpad, 0, ctx->volpad_elems * sizeof(float), ctx->stream));
        ctx->halo_dirty = false;
    }
    TOMO_LAUNCH(ctx, "k_pad", k_pad, dim3(g.nx * g.ny), dim3(256), 0, d_vol, ctx->d_volpad, g);
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// per-ray set-up shared by the ray-driven kernels
// ------------------------------------------------------------------------------------------------
struct RayCtx {
    double b[3], d[3];
    int j0, j1;
};

__device__ __forceinline__ void ray_setup(const ProjC &c, const TomoGeomC &g, int ix, int iz, bool valid, RayCtx &r)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.b[a] = c.p0[a] + (double)ix * c.u[a] + (double)iz * c.w[a];
        r.d[a] = c.d[a];
    }
    tomo_ray_range(r.b, r.d, c.n, g.nx, g.ny, g.nz, r.j0, r.j1);
    if (!valid) r.j0 = r.j1 = 0;
}

// trilinear value from the 8 loaded corners: v000 + wz*(v001-v000) ... == sum rec*wx*wy*wz of
// src/ray_wt_grad.f90:143-145 with wf = 1-wc (utilities/ray_voxel_utilities.py:98-99)
__device__ __forceinline__ float trilerp(float v000, float v001, float v010, float v011, float v100, float v101, float v110,
                                         float v111, float wx, float wy, float wz)
{
    float c00 = fmaf(wz, v001 - v000, v000);
    float c01 = fmaf(wz, v011 - v010, v010);
    float c10 = fmaf(wz, v101 - v100, v100);
    float c11 = fmaf(wz, v111 - v110, v110);
    float e0 = fmaf(wy, c01 - c00, c00);
    float e1 = fmaf(wy, c11 - c10, c10);
    return fmaf(wx, e1 - e0, e0);
}

// ------------------------------------------------------------------------------------------------
// forward projection, variant 1: plain 64-bit indexing (reference form of the algorithm)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_v1(const ProjC *__restrict__ pcs, const float *__restrict__ vp,
                                                float *__restrict__ proj, TomoGeomC g)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int iz = blockIdx.x * 64 + lane, ix = blockIdx.y * 4 + wv, ip = blockIdx.z;
    if (ix >= g.ndx || iz >= g.ndz) return;
    const ProjC &c = pcs[ip];
    RayCtx r;
    ray_setup(c, g, ix, iz, true, r);
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const int64_t sy = g.nzp, sx = (int64_t)g.nyp * g.nzp;
    double total = 0.0;
    for (int jb = r.j0; jb < r.j1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const float *base = vp + ((int64_t)(ia[0] + TOMO_HALO) * sx + (int64_t)(ia[1] + TOMO_HALO) * sy + (ia[2] + TOMO_HALO));
        const int cnt = min(TOMO_JB, r.j1 - jb);
        float acc = 0.f;
        for (int jj = 0; jj < cnt; ++jj) {
            const float t = (float)jj;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
            const float *q = base + ((int64_t)(int)fx * sx + (int64_t)(int)fy * sy + (int)fz);
            acc += trilerp(q[0], q[1], q[sy], q[sy + 1], q[sx], q[sx + 1], q[sx + sy], q[sx + sy + 1], x - fx, y - fy, z - fz);
        }
        total += (double)acc;
    }
    proj[((size_t)ip * g.ndx + ix) * g.ndz + iz] = (float)total;
}

// ------------------------------------------------------------------------------------------------
// forward projection, variant 2: wave-uniform 64-bit block base in SGPRs + one unsigned 32-bit byte
// offset per lane shared by all 8 corner loads (global_load_dword v, v_off, s[base], offset:0|4).
// The block loop runs over the wave-uniform union of the lanes' sample ranges so the cross-lane
// minimum is taken with every lane active.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_v2(const ProjC *__restrict__ pcs, const float *__restrict__ vp,
                                                float *__restrict__ proj, TomoGeomC g)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ix = blockIdx.y * 4 + wv, ip = blockIdx.z;
    if (ix >= g.ndx) return;                       // wave-uniform exit
    int iz = blockIdx.x * 64 + lane;
    const bool valid = iz < g.ndz;
    if (!valid) iz = g.ndz - 1;                    // keep the lane's arithmetic in range; it contributes nothing
    const ProjC &c = pcs[ip];
    RayCtx r;
    ray_setup(c, g, ix, iz, valid, r);
    const bool nonempty = r.j1 > r.j0;
    const int J0 = __builtin_amdgcn_readfirstlane(wave_min_i32(nonempty ? r.j0 : INT_MAX));
    const int J1 = __builtin_amdgcn_readfirstlane(wave_max_i32(nonempty ? r.j1 : 0));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const uint32_t sy4 = (uint32_t)g.nzp * 4u, sx4 = (uint32_t)g.nyp * (uint32_t)g.nzp * 4u;
    // dword gathers on purpose (a wave-wide dwordx2 costs 3.5x a dword in the L1 pipeline, tools/gather_bench.hip): the z + 1
    // bases carry an offset the compiler cannot see through, so it does not fuse the corner pairs
    int four;
    asm volatile("s_mov_b32 %0, 4" : "=s"(four));
    double total = 0.0;
    for (int jb = J0; jb < J1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const int64_t lin = ((int64_t)(ia[0] + TOMO_HALO) * g.nyp + (ia[1] + TOMO_HALO)) * g.nzp + (ia[2] + TOMO_HALO);
        const int64_t lin0 = readfirstlane_i64(lin);
        const int delta = (int)(lin - lin0);       // neighbouring rays at the same j: a few rows apart
        const int m = __builtin_amdgcn_readfirstlane(wave_min_i32(delta));
        const char *sb00 = (const char *)(vp + (lin0 + m));
        const char *sb01 = sb00 + sy4;
        const char *sb10 = sb00 + sx4;
        const char *sb11 = sb10 + sy4;
        const char *sc00 = sb00 + four, *sc01 = sb01 + four, *sc10 = sb10 + four, *sc11 = sb11 + four;
        const uint32_t off0 = (uint32_t)(delta - m) * 4u;
        const int lo = max(r.j0, jb) - jb, hi = min(r.j1, jb + TOMO_JB) - jb;
        float acc = 0.f;
        for (int jj = lo; jj < hi; ++jj) {
            const float t = (float)jj;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
            const uint32_t vo = off0 + __umul24((uint32_t)(int)fx, sx4) + __umul24((uint32_t)(int)fy, sy4) + ((uint32_t)(int)fz << 2);
            const float v000 = *(const float *)(sb00 + vo), v001 = *(const float *)(sc00 + vo);
            const float v010 = *(const float *)(sb01 + vo), v011 = *(const float *)(sc01 + vo);
            const float v100 = *(const float *)(sb10 + vo), v101 = *(const float *)(sc10 + vo);
            const float v110 = *(const float *)(sb11 + vo), v111 = *(const float *)(sc11 + vo);
            acc += trilerp(v000, v001, v010, v011, v100, v101, v110, v111, x - fx, y - fy, z - fz);
        }
        total += (double)acc;
    }
    if (valid) proj[((size_t)ip * g.ndx + ix) * g.ndz + iz] = (float)total;
}

// ------------------------------------------------------------------------------------------------
// adjoint, variant 1: the same traversal scattering w*y with global float atomics into the padded
// scratch volume (halo swallows the out-of-bounds corners).  Atomic-rate bound (~1.3 TB/s of added
// bytes, MI355X_MICROARCH 'Global float atomics'): kept as the simple reference form for parity.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adj_v1(const ProjC *__restrict__ pcs, const float *__restrict__ proj,
                                                float *__restrict__ vp, TomoGeomC g)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int iz = blockIdx.x * 64 + lane, ix = blockIdx.y * 4 + wv, ip = blockIdx.z;
    if (ix >= g.ndx || iz >= g.ndz) return;
    const ProjC &c = pcs[ip];
    RayCtx r;
    ray_setup(c, g, ix, iz, true, r);
    const float yv = proj[((size_t)ip * g.ndx + ix) * g.ndz + iz];
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const int64_t sy = g.nzp, sx = (int64_t)g.nyp * g.nzp;
    for (int jb = r.j0; jb < r.j1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        float *base = vp + ((int64_t)(ia[0] + TOMO_HALO) * sx + (int64_t)(ia[1] + TOMO_HALO) * sy + (ia[2] + TOMO_HALO));
        const int cnt = min(TOMO_JB, r.j1 - jb);
        for (int jj = 0; jj < cnt; ++jj) {
            const float t = (float)jj;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
            const float wcx = x - fx, wcy = y - fy, wcz = z - fz;
            const float wfx = 1.f - wcx, wfy = 1.f - wcy, wfz = 1.f - wcz;
            float *q = base + ((int64_t)(int)fx * sx + (int64_t)(int)fy * sy + (int)fz);
            const float a0 = yv * wfx, a1 = yv * wcx;
            const float b00 = a0 * wfy, b01 = a0 * wcy, b10 = a1 * wfy, b11 = a1 * wcy;
            atomicAdd(q, b00 * wfz);
            atomicAdd(q + 1, b00 * wcz);
            atomicAdd(q + sy, b01 * wfz);
            atomicAdd(q + sy + 1, b01 * wcz);
            atomicAdd(q + sx, b10 * wfz);
            atomicAdd(q + sx + 1, b10 * wcz);
            atomicAdd(q + sx + sy, b11 * wfz);
            atomicAdd(q + sx + sy + 1, b11 * wcz);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint, variant 2: volume-tile-owned scatter into LDS, fixed-point.
//
// A work-group owns the samples whose floor cell lies in an ATX x ATY x ATZ voxel tile (the tile grid
// starts at -1 so the floor = -1 shell is owned too) and accumulates their 8 corner contributions into
// a (ATX+1)(ATY+1)(ATZ+1) LDS image.  Measured on MI355X (tools/lds_atomic_bench.hip): ds_add_f32 costs
// ~170 cycles per wave-op, ds_add_u32 ~4 -- so contributions are converted to 32-bit fixed point
// (scale from the sinogram's abs-max, found on the device) and added with ds_add_u32; integer adds
// commute, so the LDS image does not depend on wave scheduling.  Every ADJ_BATCH projections the
// image is converted back and flushed with global float atomics (~1.2x the volume bytes per batch
// instead of 8 global atomics per sample at the chip-wide ~1.3 TB/s atomic rate).
// Lanes run along detector-z (consecutive LDS banks); the 8 waves take different detector-x rows.  A
// row's sample range comes from clipping its centre line against the tile box widened by the lanes'
// lateral spread; each lane then masks itself by exact ownership.  Cell indices and weights come
// from the same tile-independent block anchors as the forward kernel (tomo_block_anchor), so
// neighbouring tiles agree bit-for-bit on who owns a sample and A^T uses exactly A's weights.
// ------------------------------------------------------------------------------------------------
#define ATX 16
#define ATY 16
#define ATZ 60
#define ALX (ATX + 1)
#define ALY (ATY + 1)
#define ALZ 64            // LDS row of ATZ + 1 planes padded to 64 dwords (256-B aligned rows: measured 20 % faster LDS atomics)
#define ADJ_WAVES 8
#define ADJ_BATCH 64

struct AdjC {
    double p0[3], u[3], w[3], d[3];
    double minv[3][3];   // (ix, iz, j) = minv * (p - p0)
    int64_t fp0[3], fu[3], fw[3], fd[3];   // the same lattice in 32.32 fixed point (index space)
    int32_t n;
    int32_t slot;        // row block of the sinogram this projection reads / writes (its index in the caller's pose list)
};

__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ v, int64_t n, unsigned *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bit patterns
}

__device__ __forceinline__ int cvt_round_i32(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));   // floor(x + 0.5) in one instruction
    return r;
}

// trilinear value with the lerps ordered y -> x -> z so that the (z, z+1) register pairs ds_read2_b32 returns feed the
// packed ops directly: p00 = (v000, v001), p01 = (v010, v011), p10 = (v100, v101), p11 = (v110, v111)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float trilerp_pairs(f32x2 p00, f32x2 p01, f32x2 p10, f32x2 p11, float wx, float wy, float wz)
{
    const f32x2 c0 = p00 + wy * (p01 - p00);
    const f32x2 c1 = p10 + wy * (p11 - p10);
    const f32x2 e = c0 + wx * (c1 - c0);
    return fmaf(wz, e.y - e.x, e.x);
}

// FWD = true : the LDS image holds the volume tile (+1 high-side halo, zeros outside the volume); owned samples are
//              interpolated from it with ds_read and each (tile, projection, detector row) adds its partial ray sums to
//              proj with one 256-B global float atomic per wave -- the volume is read from HBM once per CALL, not per angle.
// FWD = false: the adjoint described above.
//
// Sample positions are 32.32 FIXED POINT (int64): p = fp0 + ix*fu + iz*fw + j*fd - tile_origin.  Integer arithmetic is
// exact and order-independent, so every tile computes the identical cell and fraction for a sample (consistent ownership,
// A^T uses exactly A's weights) without any float64 work in the kernel; resolution 2^-32 voxel, accumulated rounding of the
// lattice constants < 1e-6 voxel at 1024^3.
template <bool FWD>
__global__ __launch_bounds__(ADJ_WAVES * 64) void k_tile(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                         float *__restrict__ vol, TomoGeomC g, const unsigned *__restrict__ absmax_bits,
                                                         float weight_bound, int tile_x0)
{
    __shared__ int acc[ALX * ALY * ALZ];
    const float *img = (const float *)acc;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z0 = -1 + (int)blockIdx.x * ATZ, y0 = -1 + (int)blockIdx.y * ATY, x0 = -1 + ((int)blockIdx.z + tile_x0) * ATX;
    float scale = 1.f, inv_scale = 1.f;
    if (FWD) {
        bool any_nz = false;
        for (int e = threadIdx.x; e < ALX * ALY * ALZ; e += ADJ_WAVES * 64) {
            const int lz = e % ALZ, t2 = e / ALZ, ly = t2 % ALY, lx = t2 / ALY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
            float v = 0.f;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) v = vol[((size_t)gx * g.ny + gy) * g.nz + gz];
            ((float *)acc)[e] = v;
            any_nz |= (v != 0.f);
        }
        if (!__syncthreads_or(any_nz)) return;                   // an all-zero tile contributes nothing to any ray
    } else {
        const float ymax = __uint_as_float(*absmax_bits);
        if (!(ymax > 0.f)) return;                               // A^T 0 = 0 (vol already holds the right answer)
        // |image| <= ADJ_BATCH * ymax * weight_bound  ->  keep it below 2^30
        scale = 1073741824.f / ((float)min(n_proj, ADJ_BATCH) * ymax * weight_bound);
        inv_scale = 1.f / scale;
        for (int e = threadIdx.x; e < ALX * ALY * ALZ; e += ADJ_WAVES * 64) acc[e] = 0;
        __syncthreads();
    }
    const float bc[3] = {(float)x0 + 0.5f * ATX, (float)y0 + 0.5f * ATY, (float)z0 + 0.5f * ATZ};   // owned-box centre
    const float ext[3] = {(float)ATX, (float)ATY, (float)ATZ};
    const int64_t org[3] = {(int64_t)x0 << 32, (int64_t)y0 << 32, (int64_t)z0 << 32};
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const float two_m32 = 2.3283064365386963e-10f;

    const int batch = FWD ? n_proj : ADJ_BATCH;
    for (int ip0 = 0; ip0 < n_proj; ip0 += batch) {
        const int ip1 = min(n_proj, ip0 + batch);
        for (int ip = ip0 + wv; ip < ip1; ip += ADJ_WAVES) {      // one wave owns a whole (tile, projection): set-up runs once
            const AdjC &c = pcs[ip];
            // Range work is CONSERVATIVE set-up in float32 (coordinates < 2^11: float32 error < 1e-3 voxel, margins 2e-2): it
            // only has to cover the owned samples; exact ownership is decided per sample from the fixed-point position.
            // lattice-coordinate ranges of the owned box: a linear functional over a box = centre value +- sum |coef|*half-extent
            const float qx = bc[0] - (float)c.p0[0], qy = bc[1] - (float)c.p0[1], qz = bc[2] - (float)c.p0[2];
            const float m00 = (float)c.minv[0][0], m01 = (float)c.minv[0][1], m02 = (float)c.minv[0][2];
            const float m10 = (float)c.minv[1][0], m11 = (float)c.minv[1][1], m12 = (float)c.minv[1][2];
            const float ixc = m00 * qx + m01 * qy + m02 * qz;
            const float ixr = fabsf(m00) * (0.5f * ATX) + fabsf(m01) * (0.5f * ATY) + fabsf(m02) * (0.5f * ATZ) + 2e-2f;
            const float izm = m10 * qx + m11 * qy + m12 * qz;
            const float izr = fabsf(m10) * (0.5f * ATX) + fabsf(m11) * (0.5f * ATY) + fabsf(m12) * (0.5f * ATZ) + 2e-2f;
            const int ix_lo = max(0, (int)ceilf(fmaxf(ixc - ixr, -1.f)));
            const int ix_hi = min(g.ndx - 1, (int)floorf(fminf(ixc + ixr, (float)g.ndx)));
            if (ix_lo > ix_hi) continue;
            const float izl = fmaxf(izm - izr, 0.f), izh = fminf(izm + izr, (float)(g.ndz - 1));
            if (izl > izh + 1.f) continue;
            const float izc = 0.5f * (izl + izh), hs = 0.5f * (izh - izl) + 1.f;   // lanes' iz spread about the centre line
            const int n_rows_w = ix_hi - ix_lo + 1;
            const float fp0[3] = {(float)c.p0[0] - (float)x0, (float)c.p0[1] - (float)y0, (float)c.p0[2] - (float)z0};   // tile-relative
            const float fu[3] = {(float)c.u[0], (float)c.u[1], (float)c.u[2]}, fw[3] = {(float)c.w[0], (float)c.w[1], (float)c.w[2]};
            const float fd[3] = {(float)c.d[0], (float)c.d[1], (float)c.d[2]};
            // per-lane part of the fixed-point position: lane * fw  (the row adds the uniform rest)
            int64_t lw0 = (int64_t)lane * c.fw[0], lw1 = (int64_t)lane * c.fw[1], lw2 = (int64_t)lane * c.fw[2];
            asm volatile("" : "+v"(lw0), "+v"(lw1), "+v"(lw2));      // opaque: or the compiler rebuilds them with 64-bit multiplies per chunk

            for (int r0 = 0; r0 < n_rows_w; r0 += 64) {
                // row set-up, one detector row per LANE (row r0+lane of this wave), broadcast below with v_readlane:
                // sample range = centre line clipped against the box widened by the lanes' lateral spread; detector-z
                // lanes needed for ownership in z over that range
                int v_jlo = 0, v_jhi = 0, v_izf = 0, v_izl = -1;
                {
                    const int rix = ix_lo + r0 + lane;
                    const float frix = (float)rix;
                    float t0 = 0.f, t1 = (float)(c.n - 1);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float cb = fp0[a] + frix * fu[a] + izc * fw[a];          // tile-relative centre-line point at j = 0
                        const float h = fabsf(fw[a]) * hs + 2e-2f;
                        const float lo_a = -h, hi_a = ext[a] + h;
                        if (fd[a] != 0.f) {
                            const float inv = 1.f / fd[a];
                            const float ta = (lo_a - cb) * inv, tb = (hi_a - cb) * inv;
                            t0 = fmaxf(t0, fminf(ta, tb));
                            t1 = fminf(t1, fmaxf(ta, tb));
                        } else if (cb < lo_a || cb >= hi_a) {
                            t0 = 1.f; t1 = 0.f;
                        }
                    }
                    if (rix <= ix_hi && t0 <= t1) {
                        v_jlo = max(0, (int)ceilf(t0));                              // the 2e-2 box margin already covers float32 error
                        v_jhi = min(c.n, (int)floorf(t1) + 1);
                        const float czr = fp0[2] + frix * fu[2];                       // z0-relative
                        const float zj0 = (float)v_jlo * fd[2], zj1 = (float)(v_jhi - 1) * fd[2];
                        const float iw = 1.f / fw[2];
                        v_izf = max(0, (int)floorf((0.f - czr - fmaxf(zj0, zj1)) * iw - 2e-2f));
                        v_izl = min(g.ndz - 1, (int)ceilf((ext[2] - czr - fminf(zj0, zj1)) * iw + 2e-2f));
                    }
                }
                const int r_end = min(64, n_rows_w - r0);
                for (int r = 0; r < r_end; ++r) {
                    const int jlo = __builtin_amdgcn_readlane(v_jlo, r), jhi = __builtin_amdgcn_readlane(v_jhi, r);
                    if (jhi <= jlo) continue;
                    const int iz_first = __builtin_amdgcn_readlane(v_izf, r), iz_last = __builtin_amdgcn_readlane(v_izl, r);
                    const int ix = ix_lo + r0 + r;
                    // uniform part of the fixed-point position of sample jlo of this row (scalar 64-bit arithmetic)
                    const int64_t rb0 = c.fp0[0] + (int64_t)ix * c.fu[0] + (int64_t)jlo * c.fd[0] - org[0];
                    const int64_t rb1 = c.fp0[1] + (int64_t)ix * c.fu[1] + (int64_t)jlo * c.fd[1] - org[1];
                    const int64_t rb2 = c.fp0[2] + (int64_t)ix * c.fu[2] + (int64_t)jlo * c.fd[2] - org[2];
                    const int cnt = jhi - jlo;
                    for (int izb = iz_first; izb <= iz_last; izb += 64) {
                        const int iz = izb + lane;
                        const bool lane_ok = iz <= iz_last;
                        float *pr = proj + (size_t)c.slot * n_det + (size_t)ix * g.ndz + iz;
                        int64_t px = rb0 + (int64_t)izb * c.fw[0] + lw0;
                        int64_t py = rb1 + (int64_t)izb * c.fw[1] + lw1;
                        int64_t pz = rb2 + (int64_t)izb * c.fw[2] + lw2;
                        if (FWD) {
                            // branch-free body (lanes that do not own the sample read LDS word 0 and discard it), so the compiler
                            // can overlap the LDS latency of consecutive samples
                            float part = 0.f;
                            for (int jj = 0; jj < cnt; ++jj) {
                                const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32), lz = (unsigned)(pz >> 32);
                                static_assert(ATX == ATY && (ATX & (ATX - 1)) == 0, "ownership test uses (lx | ly) < ATX");
                                static_assert(ALY == 17 && ALZ == 64, "cell index is written with shifts");
                                const bool own = (lx | ly) < (unsigned)ATX && lz < (unsigned)ATZ;
                                const unsigned e = own ? ((((lx << 4) + lx + ly) << 6) + lz) : 0u;      // (lx * ALY + ly) * ALZ + lz without a quarter-rate multiply
                                const float wx = (float)(unsigned)px * two_m32, wy = (float)(unsigned)py * two_m32, wz = (float)(unsigned)pz * two_m32;
                                const float *q = img + e;
                                const f32x2 p00 = {q[0], q[1]}, p01 = {q[ALZ], q[ALZ + 1]};
                                const f32x2 p10 = {q[ALY * ALZ], q[ALY * ALZ + 1]}, p11 = {q[ALY * ALZ + ALZ], q[ALY * ALZ + ALZ + 1]};
                                const float v = trilerp_pairs(p00, p01, p10, p11, wx, wy, wz);
                                part += own ? v : 0.f;
                                px += c.fd[0]; py += c.fd[1]; pz += c.fd[2];
                            }
                            if (lane_ok) atomicAdd(pr, part);          // 64 consecutive floats per wave: the full-rate atomic shape
                        } else {
                            const float ys = (lane_ok ? *pr : 0.f) * scale;
                            for (int jj = 0; jj < cnt; ++jj) {
                                const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32), lz = (unsigned)(pz >> 32);
                                if (lane_ok && (lx | ly) < (unsigned)ATX && lz < (unsigned)ATZ) {
                                    const float wcx = (float)(unsigned)px * two_m32, wcy = (float)(unsigned)py * two_m32, wcz = (float)(unsigned)pz * two_m32;
                                    const float wfx = 1.f - wcx, wfy = 1.f - wcy, wfz = 1.f - wcz;
                                    const float a0 = ys * wfx, a1 = ys * wcx;
                                    const float b00 = a0 * wfy, b01 = a0 * wcy, b10 = a1 * wfy, b11 = a1 * wcy;
                                    int *q = &acc[(((lx << 4) + lx + ly) << 6) + lz];                    // (lx * ALY + ly) * ALZ + lz
                                    atomicAdd(q, cvt_round_i32(b00 * wfz));
                                    atomicAdd(q + 1, cvt_round_i32(b00 * wcz));
                                    atomicAdd(q + ALZ, cvt_round_i32(b01 * wfz));
                                    atomicAdd(q + ALZ + 1, cvt_round_i32(b01 * wcz));
                                    atomicAdd(q + ALY * ALZ, cvt_round_i32(b10 * wfz));
                                    atomicAdd(q + ALY * ALZ + 1, cvt_round_i32(b10 * wcz));
                                    atomicAdd(q + ALY * ALZ + ALZ, cvt_round_i32(b11 * wfz));
                                    atomicAdd(q + ALY * ALZ + ALZ + 1, cvt_round_i32(b11 * wcz));
                                }
                                px += c.fd[0]; py += c.fd[1]; pz += c.fd[2];
                            }
                        }
                    }
                }
            }
        }
        if (FWD) break;
        __syncthreads();
        // flush this batch: interior of the image is exclusively ours, the +1 faces are shared => global atomics
        for (int e = threadIdx.x; e < ALX * ALY * ALZ; e += ADJ_WAVES * 64) {
            const int v = acc[e];
            if (v != 0) {
                acc[e] = 0;
                const int lz = e % ALZ, t2 = e / ALZ, ly = t2 % ALY, lx = t2 / ALY;
                const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
                if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz)
                    atomicAdd(&vol[((size_t)gx * g.ny + gy) * g.nz + gz], (float)v * inv_scale);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// "flat" tile kernels for UNTILTED lattices (alpha = beta = 0, detector-z pitch 1; any phi, translation, COR shift):
//   fw = (0, 0, 1), fu_z = fd_z = 0  =>  x,y of a sample depend on (ix, j) only, z on iz only.
// Then for one detector row the cell (lx, ly), the x/y weights and the LDS address are the same in all 64 lanes, and
// every lane sees the same z fraction.  So: lane l is pinned to LDS plane l; one lane per SAMPLE precomputes
// (address, own, w00, w01, w10, w11) once per row; the sample loop broadcasts those 6 words with v_readlane and does
// 2 ds_read2_b32 + 4 FMA (forward) or 4 mul + 4 cvt + 4 ds_add_u32 (adjoint) per lane; the z-lerp is applied once per
// row (forward: to the accumulated plane sums S_l, S_{l+1}; adjoint: to the sinogram row before the loop).
// Same sums as k_tile, regrouped: ~11 VALU per sample instead of ~32.
// ------------------------------------------------------------------------------------------------
#define FTZ 63              // flat kernels: 63 owned planes + halo = all 64 lanes busy
#define FLZ (FTZ + 1)
#define FTAB 32             // entries of the forward kernel's per-wave sample table
#define FTAB_ALLOC (FTAB + 4) // + zero padding for the groups of four

template <bool FWD>
__global__ __launch_bounds__(ADJ_WAVES * 64) void k_tile_flat(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                              float *__restrict__ vol, TomoGeomC g, const unsigned *__restrict__ absmax_bits,
                                                              float weight_bound, int tile_x0)
{
    __shared__ int acc[ALX * ALY * FLZ];
    // forward only: per-wave table of the samples of the current row chunk that fall into this tile's x,y cells (compacted):
    // the four x,y weights and the byte offset of the cell in the image.  The sample loop fetches entries with broadcast
    // ds_reads at immediate offsets instead of six v_readlane per sample (PMC: the VALU was 94 % busy, LDS issue stalls 0.3 %).
    // 32 entries: a row crosses <= 24 cells of a 16 x 16 tile; + zero padding so that the loop runs in unmasked groups of four.
    __shared__ float4 tab_w[FWD ? ADJ_WAVES * FTAB_ALLOC : 1];
    __shared__ __attribute__((aligned(16))) unsigned tab_e[FWD ? ADJ_WAVES * FTAB_ALLOC : 4];
    const float *img = (const float *)acc;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z0 = -1 + (int)blockIdx.x * FTZ, y0 = -1 + (int)blockIdx.y * ATY, x0 = -1 + ((int)blockIdx.z + tile_x0) * ATX;
    float scale = 1.f, inv_scale = 1.f;
    if (FWD) {
        bool any_nz = false;
        for (int e = threadIdx.x; e < ALX * ALY * FLZ; e += ADJ_WAVES * 64) {
            const int lz = e % FLZ, t2 = e / FLZ, ly = t2 % ALY, lx = t2 / ALY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
            float v = 0.f;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) v = vol[((size_t)gx * g.ny + gy) * g.nz + gz];
            ((float *)acc)[e] = v;
            any_nz |= (v != 0.f);
        }
        if (!__syncthreads_or(any_nz)) return;
    } else {
        const float ymax = __uint_as_float(*absmax_bits);
        if (!(ymax > 0.f)) return;
        scale = 1073741824.f / ((float)min(n_proj, ADJ_BATCH) * ymax * weight_bound);
        inv_scale = 1.f / scale;
        for (int e = threadIdx.x; e < ALX * ALY * FLZ; e += ADJ_WAVES * 64) acc[e] = 0;
        __syncthreads();
    }
    const float bcx = (float)x0 + 0.5f * ATX, bcy = (float)y0 + 0.5f * ATY;
    const int64_t orgx = (int64_t)x0 << 32, orgy = (int64_t)y0 << 32;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const float two_m32 = 2.3283064365386963e-10f;
    const unsigned lane4 = (unsigned)min(lane, FLZ - 1) * 4u;      // lanes 61..63 alias the halo plane with zero weight

    const int batch = FWD ? n_proj : ADJ_BATCH;
    for (int ip0 = 0; ip0 < n_proj; ip0 += batch) {
        const int ip1 = min(n_proj, ip0 + batch);
        for (int ip = ip0 + wv; ip < ip1; ip += ADJ_WAVES) {      // one wave owns a whole (tile, projection): set-up runs once
            const AdjC &c = pcs[ip];
            // z: ray iz sits in plane lz = floor(p0z) + iz - z0 with the same fraction for every ray
            const int p0z_i = (int)(c.fp0[2] >> 32);
            const float wcz = (float)(unsigned)c.fp0[2] * two_m32, wfz = 1.f - wcz;
            const int izoff = z0 - p0z_i;                              // iz = lane + izoff
            if (izoff + FTZ <= 0 || izoff >= g.ndz) continue;          // no ray of this projection floors into the tile's z range
            // detector rows crossing the tile's x,y footprint (2-D: a linear functional over a rectangle)
            const float qx = bcx - (float)c.p0[0], qy = bcy - (float)c.p0[1];
            const float m00 = (float)c.minv[0][0], m01 = (float)c.minv[0][1];
            const float ixc = m00 * qx + m01 * qy;
            const float ixr = fabsf(m00) * (0.5f * ATX) + fabsf(m01) * (0.5f * ATY) + 2e-2f;
            const int ix_lo = max(0, (int)ceilf(fmaxf(ixc - ixr, -1.f)));
            const int ix_hi = min(g.ndx - 1, (int)floorf(fminf(ixc + ixr, (float)g.ndx)));
            if (ix_lo > ix_hi) continue;
            const int n_rows_w = ix_hi - ix_lo + 1;
            const float fp0x = (float)c.p0[0] - (float)x0, fp0y = (float)c.p0[1] - (float)y0;
            const float fux = (float)c.u[0], fuy = (float)c.u[1], fdx = (float)c.d[0], fdy = (float)c.d[1];
            const int iz = izoff + lane;
            const bool ray_ok = lane < FTZ && iz >= 0 && iz < g.ndz;   // the ray this lane owns (the last plane is halo only)
            int64_t ldx = (int64_t)lane * c.fd[0], ldy = (int64_t)lane * c.fd[1];   // sample `lane` of a chunk, relative to its first
            // The row loop below runs ~24 times per (tile, projection).  Keep what it needs in registers: left to itself the
            // compiler re-loaded the lattice constants from memory in every row (scalar loads + wait) and rebuilt lane * fd with
            // 64 x 64-bit multiplies.  The empty asm statements make the values opaque, so they can be neither rematerialised
            // nor folded back into a multiply.
            int64_t k_fux = c.fu[0], k_fuy = c.fu[1], k_fdx = c.fd[0], k_fdy = c.fd[1];
            asm volatile("" : "+v"(ldx), "+v"(ldy));
            asm volatile("" : "+s"(k_fux), "+s"(k_fuy), "+s"(k_fdx), "+s"(k_fdy));
            float *const proj_c = proj + (size_t)c.slot * n_det + iz;

            for (int r0 = 0; r0 < n_rows_w; r0 += 64) {
                int v_jlo = 0, v_jhi = 0;
                {
                    const int rix = ix_lo + r0 + lane;
                    const float frix = (float)rix;
                    float t0 = 0.f, t1 = (float)(c.n - 1);
                    {
                        const float cb = fp0x + frix * fux;
                        if (fdx != 0.f) {
                            const float inv = 1.f / fdx, ta = (-2e-2f - cb) * inv, tb = ((float)ATX + 2e-2f - cb) * inv;
                            t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                        } else if (cb < -2e-2f || cb >= (float)ATX + 2e-2f) { t0 = 1.f; t1 = 0.f; }
                    }
                    {
                        const float cb = fp0y + frix * fuy;
                        if (fdy != 0.f) {
                            const float inv = 1.f / fdy, ta = (-2e-2f - cb) * inv, tb = ((float)ATY + 2e-2f - cb) * inv;
                            t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                        } else if (cb < -2e-2f || cb >= (float)ATY + 2e-2f) { t0 = 1.f; t1 = 0.f; }
                    }
                    if (rix <= ix_hi && t0 <= t1) {
                        v_jlo = max(0, (int)ceilf(t0));
                        v_jhi = min(c.n, (int)floorf(t1) + 1);
                    }
                }
                const int r_end = min(64, n_rows_w - r0);
                // row bases advance incrementally: tile-relative 32.32 position of sample 0 and the row's sinogram pointer
                int64_t rbx = c.fp0[0] + (int64_t)(ix_lo + r0) * k_fux - orgx, rby = c.fp0[1] + (int64_t)(ix_lo + r0) * k_fuy - orgy;
                float *pr = proj_c + (size_t)(ix_lo + r0) * g.ndz;
                for (int r = 0; r < r_end; ++r, rbx += k_fux, rby += k_fuy, pr += g.ndz) {
                    const int jlo = __builtin_amdgcn_readlane(v_jlo, r), jhi = __builtin_amdgcn_readlane(v_jhi, r);
                    if (jhi <= jlo) continue;
                    float S = 0.f;            // forward: sum over samples of the x,y-interpolated plane `lane`
                    float yt = 0.f;           // adjoint: what this row adds to plane `lane` per unit x,y weight (fixed-point scaled)
                    if (!FWD) {
                        const float yv = ray_ok ? *pr : 0.f;
                        const float ym1 = __shfl_up(yv, 1, 64);        // ray of plane lane-1 (lane 0: belongs to the tile below)
                        yt = (wfz * yv + (lane > 0 ? wcz * ym1 : 0.f)) * scale;
                    }
                    for (int jc = jlo; jc < jhi; jc += (FWD ? FTAB : 64)) {
                        // one lane per SAMPLE: cell, ownership in x,y and the four x,y weights of sample jc + lane
                        const int64_t px = (rbx + (int64_t)jc * k_fdx) + ldx, py = (rby + (int64_t)jc * k_fdy) + ldy;   // uniform part on the SALU
                        const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32);
                        const bool own = (lx | ly) < (unsigned)ATX && jc + lane < jhi && (!FWD || lane < FTAB);
                        const unsigned t_e = own ? (__umul24(lx, ALY * FLZ) + __umul24(ly, FLZ)) * 4u : 0xffffffffu;
                        const float wx = (float)(unsigned)px * two_m32, wy = (float)(unsigned)py * two_m32;
                        const float t_w11 = wx * wy, t_w10 = wx - t_w11, t_w01 = wy - t_w11, t_w00 = 1.f - wx - t_w01;
#ifdef TOMO_ABLATE_FLAT_SAMPLES          // development build only: skip the sample loop at run time (keeps all set-up alive)
                        const int cnt = g.step < 0.0 ? min(64, jhi - jc) : 0;
#else
                        const int cnt = min(64, jhi - jc);
#endif
                        if (FWD) {
                            // compact the owned samples into the wave's table (LDS operations of a wave execute in order: no barrier);
                            // three zero entries behind them let the loop run in unmasked groups of four
                            const unsigned long long om = __ballot(own);
                            const int n_own = cnt > 0 ? (int)__builtin_popcountll(om) : 0;
                            float4 *tw = tab_w + wv * FTAB_ALLOC;
                            unsigned *te = tab_e + wv * FTAB_ALLOC;
                            if (own) {
                                const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(om >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)om, 0u));
                                tw[at] = make_float4(t_w00, t_w01, t_w10, t_w11);
                                te[at] = t_e;
                            }
                            if (lane >= n_own && lane < n_own + 3) {
                                tw[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                                te[lane] = 0u;
                            }
                            // (q[0], q[FLZ]) arrive as a register pair from one ds_read2st64, (w00, w01) as a pair of the table's
                            // float4: two packed FMAs per sample, no shuffles; .x collects the y-cell, .y the y+1-cell terms
                            f32x2 Sa = {0.f, 0.f}, Sb = {0.f, 0.f}, Sc = {0.f, 0.f}, Sd = {0.f, 0.f};
#pragma unroll
                            for (int j4 = 0; j4 < FTAB; j4 += 4) {
                                if (j4 < n_own) {                                          // wave-uniform
                                    const uint4 e = *(const uint4 *)(te + j4);             // broadcast reads at immediate offsets
                                    const float4 wa = tw[j4], wb = tw[j4 + 1], wc = tw[j4 + 2], wd = tw[j4 + 3];
                                    const float *qa = (const float *)((const char *)img + (e.x + lane4));
                                    const float *qb = (const float *)((const char *)img + (e.y + lane4));
                                    const float *qc = (const float *)((const char *)img + (e.z + lane4));
                                    const float *qd = (const float *)((const char *)img + (e.w + lane4));
                                    Sa += (f32x2){wa.x, wa.y} * (f32x2){qa[0], qa[FLZ]}; Sb += (f32x2){wb.x, wb.y} * (f32x2){qb[0], qb[FLZ]};
                                    Sc += (f32x2){wc.x, wc.y} * (f32x2){qc[0], qc[FLZ]}; Sd += (f32x2){wd.x, wd.y} * (f32x2){qd[0], qd[FLZ]};
                                    Sa += (f32x2){wa.z, wa.w} * (f32x2){qa[ALY * FLZ], qa[ALY * FLZ + FLZ]};
                                    Sb += (f32x2){wb.z, wb.w} * (f32x2){qb[ALY * FLZ], qb[ALY * FLZ + FLZ]};
                                    Sc += (f32x2){wc.z, wc.w} * (f32x2){qc[ALY * FLZ], qc[ALY * FLZ + FLZ]};
                                    Sd += (f32x2){wd.z, wd.w} * (f32x2){qd[ALY * FLZ], qd[ALY * FLZ + FLZ]};
                                }
                            }
                            const f32x2 St = (Sa + Sb) + (Sc + Sd);
                            S += St.x + St.y;
                            continue;
                        }
                        for (int jj = 0; jj < cnt; ++jj) {
                            const unsigned e4 = (unsigned)__builtin_amdgcn_readlane((int)t_e, jj);
                            if (e4 == 0xffffffffu) continue;                       // sample not in this tile's x,y cells (scalar branch)
                            const float w00 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t_w00), jj));
                            const float w01 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t_w01), jj));
                            const float w10 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t_w10), jj));
                            const float w11 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t_w11), jj));
                            if (FWD) {
                                const float *q = (const float *)((const char *)img + (e4 + lane4));
                                S = fmaf(w00, q[0], S);
                                S = fmaf(w01, q[FLZ], S);
                                S = fmaf(w10, q[ALY * FLZ], S);
                                S = fmaf(w11, q[ALY * FLZ + FLZ], S);
                            } else {
                                int *q = (int *)((char *)acc + (e4 + lane4));
                                atomicAdd(q, cvt_round_i32(yt * w00));
                                atomicAdd(q + FLZ, cvt_round_i32(yt * w01));
                                atomicAdd(q + ALY * FLZ, cvt_round_i32(yt * w10));
                                atomicAdd(q + ALY * FLZ + FLZ, cvt_round_i32(yt * w11));
                            }
                        }
                    }
                    if (FWD) {
                        const float Sp1 = __shfl_down(S, 1, 64);                   // plane lane+1
                        if (ray_ok) atomicAdd(pr, wfz * S + wcz * Sp1);
                    }
                }
            }
        }
        if (FWD) break;
        __syncthreads();
        for (int e = threadIdx.x; e < ALX * ALY * FLZ; e += ADJ_WAVES * 64) {
            const int v = acc[e];
            if (v != 0) {
                acc[e] = 0;
                const int lz = e % FLZ, t2 = e / FLZ, ly = t2 % ALY, lx = t2 / ALY;
                const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
                if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz)
                    atomicAdd(&vol[((size_t)gx * g.ny + gy) * g.nz + gz], (float)v * inv_scale);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Forward flat kernel over NZT z-adjacent tiles per work-group.  60 % of k_tile_flat<true>'s time is per-row set-up (sample
// table, row bases, compaction: ~115 issue slots per row against ~130 for the row's samples) and that set-up does not depend on
// z: here a work-group of FZ_WAVES waves holds the LDS images of NZT tiles stacked in z, builds each row's table once and runs
// the sample loop against every image.  NZT = 2 with 16 waves uses 148 KB of the 160 KB LDS for the two images, with the same
// number of waves per CU as two 8-wave work-groups of the one-image kernel.
// ------------------------------------------------------------------------------------------------
#define FZ_WAVES 16
template <int NZT>
__global__ __launch_bounds__(FZ_WAVES * 64) void k_fwd_flat_z(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                              const float *__restrict__ vol, TomoGeomC g)
{
    __shared__ float img[NZT][ALX * ALY * FLZ];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z0 = -1 + (int)blockIdx.x * (NZT * FTZ), y0 = -1 + (int)blockIdx.y * ATY, x0 = -1 + (int)blockIdx.z * ATX;
    bool live[NZT];
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < NZT; ++k) {
        bool any_nz = false;
        for (int e = threadIdx.x; e < ALX * ALY * FLZ; e += FZ_WAVES * 64) {
            const int lz = e % FLZ, t2 = e / FLZ, ly = t2 % ALY, lx = t2 / ALY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + k * FTZ + lz;
            float v = 0.f;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) v = vol[((size_t)gx * g.ny + gy) * g.nz + gz];
            img[k][e] = v;
            any_nz |= (v != 0.f);
        }
        live[k] = __syncthreads_or(any_nz) != 0;                      // an all-zero tile contributes nothing to any ray
        any_live |= live[k];
    }
    if (!any_live) return;
    const float bcx = (float)x0 + 0.5f * ATX, bcy = (float)y0 + 0.5f * ATY;
    const int64_t orgx = (int64_t)x0 << 32, orgy = (int64_t)y0 << 32;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const float two_m32 = 2.3283064365386963e-10f;
    const unsigned lane4 = (unsigned)min(lane, FLZ - 1) * 4u;      // lanes 61..63 alias the halo plane with zero weight

    for (int ip = wv; ip < n_proj; ip += FZ_WAVES) {               // one wave owns a whole (tile stack, projection)
        const AdjC &c = pcs[ip];
        // z: ray iz sits in plane lz = floor(p0z) + iz - z0 with the same fraction for every ray
        const int p0z_i = (int)(c.fp0[2] >> 32);
        const float wcz = (float)(unsigned)c.fp0[2] * two_m32, wfz = 1.f - wcz;
        bool zuse[NZT], ray_ok[NZT];
        bool any_use = false;
        const int iz0 = z0 - p0z_i + lane;                             // this lane's ray in the lowest tile; + FTZ per tile
#pragma unroll
        for (int k = 0; k < NZT; ++k) {
            const int izoff = z0 + k * FTZ - p0z_i;
            zuse[k] = live[k] && !(izoff + FTZ <= 0 || izoff >= g.ndz);   // some ray of this projection floors into the tile's z range
            any_use |= zuse[k];
            const int iz = iz0 + k * FTZ;
            ray_ok[k] = zuse[k] && lane < FTZ && iz >= 0 && iz < g.ndz;     // the last plane of an image is halo only
        }
        if (!any_use) continue;
        // detector rows crossing the tile's x,y footprint (2-D: a linear functional over a rectangle)
        const float qx = bcx - (float)c.p0[0], qy = bcy - (float)c.p0[1];
        const float m00 = (float)c.minv[0][0], m01 = (float)c.minv[0][1];
        const float ixc = m00 * qx + m01 * qy;
        const float ixr = fabsf(m00) * (0.5f * ATX) + fabsf(m01) * (0.5f * ATY) + 2e-2f;
        const int ix_lo = max(0, (int)ceilf(fmaxf(ixc - ixr, -1.f)));
        const int ix_hi = min(g.ndx - 1, (int)floorf(fminf(ixc + ixr, (float)g.ndx)));
        if (ix_lo > ix_hi) continue;
        const int n_rows_w = ix_hi - ix_lo + 1;
        const float fp0x = (float)c.p0[0] - (float)x0, fp0y = (float)c.p0[1] - (float)y0;
        const float fux = (float)c.u[0], fuy = (float)c.u[1], fdx = (float)c.d[0], fdy = (float)c.d[1];
        int64_t ldx = (int64_t)lane * c.fd[0], ldy = (int64_t)lane * c.fd[1];   // sample `lane` of a chunk, relative to its first
        int64_t k_fux = c.fu[0], k_fuy = c.fu[1], k_fdx = c.fd[0], k_fdy = c.fd[1];
        asm volatile("" : "+v"(ldx), "+v"(ldy));                       // see k_tile_flat: keep the row loop's inputs in registers
        asm volatile("" : "+s"(k_fux), "+s"(k_fuy), "+s"(k_fdx), "+s"(k_fdy));
        float *const proj_c = proj + (size_t)c.slot * n_det + iz0;

        for (int r0 = 0; r0 < n_rows_w; r0 += 64) {
            int v_jlo = 0, v_jhi = 0;
            {
                const int rix = ix_lo + r0 + lane;
                const float frix = (float)rix;
                float t0 = 0.f, t1 = (float)(c.n - 1);
                {
                    const float cb = fp0x + frix * fux;
                    if (fdx != 0.f) {
                        const float inv = 1.f / fdx, ta = (-2e-2f - cb) * inv, tb = ((float)ATX + 2e-2f - cb) * inv;
                        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                    } else if (cb < -2e-2f || cb >= (float)ATX + 2e-2f) { t0 = 1.f; t1 = 0.f; }
                }
                {
                    const float cb = fp0y + frix * fuy;
                    if (fdy != 0.f) {
                        const float inv = 1.f / fdy, ta = (-2e-2f - cb) * inv, tb = ((float)ATY + 2e-2f - cb) * inv;
                        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
                    } else if (cb < -2e-2f || cb >= (float)ATY + 2e-2f) { t0 = 1.f; t1 = 0.f; }
                }
                if (rix <= ix_hi && t0 <= t1) {
                    v_jlo = max(0, (int)ceilf(t0));
                    v_jhi = min(c.n, (int)floorf(t1) + 1);
                }
            }
            const int r_end = min(64, n_rows_w - r0);
            int64_t rbx = c.fp0[0] + (int64_t)(ix_lo + r0) * k_fux - orgx, rby = c.fp0[1] + (int64_t)(ix_lo + r0) * k_fuy - orgy;
            float *pr = proj_c + (size_t)(ix_lo + r0) * g.ndz;
            for (int r = 0; r < r_end; ++r, rbx += k_fux, rby += k_fuy, pr += g.ndz) {
                const int jlo = __builtin_amdgcn_readlane(v_jlo, r), jhi = __builtin_amdgcn_readlane(v_jhi, r);
                if (jhi <= jlo) continue;
                float S[NZT];
#pragma unroll
                for (int k = 0; k < NZT; ++k) S[k] = 0.f;
                for (int jc = jlo; jc < jhi; jc += 60) {
                    // one lane per SAMPLE: cell, ownership in x,y and the four x,y weights of sample jc + lane
                    const int64_t px = (rbx + (int64_t)jc * k_fdx) + ldx, py = (rby + (int64_t)jc * k_fdy) + ldy;
                    const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32);
                    const bool own = (lx | ly) < (unsigned)ATX && jc + lane < jhi && lane < 60;
                    const unsigned t_e = (__umul24(lx, ALY * FLZ) + __umul24(ly, FLZ)) * 4u;
                    const float wx = (float)(unsigned)px * two_m32, wy = (float)(unsigned)py * two_m32;
                    const float t_w11 = wx * wy, t_w10 = wx - t_w11, t_w01 = wy - t_w11, t_w00 = 1.f - wx - t_w01;
                    // compact the owned samples to lanes 0 .. n_own-1 IN REGISTERS (ds_permute: lane i sends to its rank among the
                    // owned; the others send to lane 63, which is never read: n_own <= 60, and in practice a row owns <= 24
                    // samples of a 16 x 16 tile; destination lanes nobody writes receive 0 = entries without effect).  The
                    // sample loop then broadcasts an entry with v_readlane: LDS cycles go to the image reads only (a table entry
                    // read from LDS cost 6.4 of the 16 LDS cycles per sample, tools/lds_read_bench.hip).
                    const unsigned long long om = __ballot(own);
                    const int n_own = (int)__builtin_popcountll(om);
                    const int dst4 = own ? 4 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(om >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)om, 0u)) : 4 * 63;
                    const int c_e = __builtin_amdgcn_ds_permute(dst4, (int)t_e);
                    const int c_w00 = __builtin_amdgcn_ds_permute(dst4, __float_as_int(t_w00)), c_w01 = __builtin_amdgcn_ds_permute(dst4, __float_as_int(t_w01));
                    const int c_w10 = __builtin_amdgcn_ds_permute(dst4, __float_as_int(t_w10)), c_w11 = __builtin_amdgcn_ds_permute(dst4, __float_as_int(t_w11));
                    f32x2 Sa[NZT], Sb[NZT];
#pragma unroll
                    for (int k = 0; k < NZT; ++k) { Sa[k] = (f32x2){0.f, 0.f}; Sb[k] = (f32x2){0.f, 0.f}; }
                    // an entry's five readlanes serve every image; (q[0], q[FLZ]) arrive as a register pair from one
                    // ds_read2st64 and the weights as SGPR pairs: two packed FMAs per sample and image.  Which images take part
                    // is decided outside the loop (an all-zero or out-of-range image is skipped).
#define FZ_ENTRY(T, J)                                                                                                      \
                        const unsigned T##e = (unsigned)__builtin_amdgcn_readlane(c_e, (J)) + lane4;                        \
                        const f32x2 T##0 = {__int_as_float(__builtin_amdgcn_readlane(c_w00, (J))), __int_as_float(__builtin_amdgcn_readlane(c_w01, (J)))}; \
                        const f32x2 T##1 = {__int_as_float(__builtin_amdgcn_readlane(c_w10, (J))), __int_as_float(__builtin_amdgcn_readlane(c_w11, (J)))};
#define FZ_READ(T, K)                                                                                                       \
                            const float *T##q = (const float *)((const char *)&img[0][0] + (T##e + (unsigned)(K) * (unsigned)(ALX * ALY * FLZ * 4))); \
                            const f32x2 T##v0 = {T##q[0], T##q[FLZ]}, T##v1 = {T##q[ALY * FLZ], T##q[ALY * FLZ + FLZ]};
#define FZ_SAMPLE_LOOP(K0, K1)                                                                                              \
                    for (int jj = 0; jj < n_own; ++jj) { /* one entry per trip: pairs measured the same, fours 6 % slower */ \
                        FZ_ENTRY(s0_, jj)                                                                                   \
                        _Pragma("unroll") for (int k = (K0); k < (K1); ++k) {                                               \
                            FZ_READ(s0_, k)                                                                                 \
                            Sa[k] += s0_0 * s0_v0; Sb[k] += s0_1 * s0_v1;                                                   \
                        }                                                                                                   \
                    }
                    if (NZT == 2 && zuse[0] && zuse[NZT - 1]) { FZ_SAMPLE_LOOP(0, NZT) }
                    else if (zuse[0]) { FZ_SAMPLE_LOOP(0, 1) }
                    else { FZ_SAMPLE_LOOP(NZT - 1, NZT) }
#undef FZ_SAMPLE_LOOP
#undef FZ_READ
#undef FZ_ENTRY
#pragma unroll
                    for (int k = 0; k < NZT; ++k) {
                        const f32x2 St = Sa[k] + Sb[k];
                        S[k] += St.x + St.y;
                    }
                }
#pragma unroll
                for (int k = 0; k < NZT; ++k) {
                    if (!zuse[k]) continue;
                    const float Sp1 = __shfl_down(S[k], 1, 64);                // plane lane+1
                    if (ray_ok[k]) atomicAdd(pr + k * FTZ, wfz * S[k] + wcz * Sp1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GATHER-form adjoint for untilted unit lattices (the poses of a plain parallel-beam scan: alpha = beta = 0, detector pitch =
// step = voxel; any phi, translation, COR shift).  For such a lattice the adjoint separates:
//     (A^T y)(X, Y, Z) = sum_ix  W(X, Y, ix) * Yz(ix, Z)
//     Yz(ix, Z)   = (1 - tau) y[ix, Z - zc] + tau y[ix, Z - zc - 1]                 (every sample has z = iz + zc + tau)
//     W(X, Y, ix) = sum_{j in [0, n)} tent(px(ix, j) - X) * tent(py(ix, j) - Y)       (tent(r) = 1 - |r| on [-1, 1))
// and W does not depend on Z.  A wave owns 8 x 8 voxel columns x 64 planes with the 64 accumulators of a lane (= column) in
// registers for ALL projections -- no atomics, no fixed-point image, no flush, each voxel written once (a lane finally
// stores its column's 64 consecutive floats):
//   1. lane = COLUMN: the <= 3 detector rows ix and <= 3 samples j per row that can reach the column are enumerated from the
//      column's lattice coordinates; their positions are exact 32.32 fixed point (the forward kernels' lattice), the tents
//      are evaluated from them, summed over j -> W0..W2 and the first row i0, per lane.  This table is the same for every
//      z chunk of the tile: the four waves of a work-group (four z chunks) each compute it for every fourth projection and
//      share it through a triple-buffered LDS table, one barrier per four projections;
//   2. lane = PLANE: the z-lerped sinogram rows the tile can touch (<= 14) are loaded once (coalesced) into wave-private LDS
//      rows (pitch 65 dwords, so that lanes reading different rows of one plane hit different banks);
//   3. lane = COLUMN again, 64 plane accumulators per lane (statically indexed registers): per plane 3 ds_read_b32 at
//      row(lane) + immediate plane offset and 3 FMA with the lane's own W0..W2 -- no broadcasts, no address arithmetic.
// Same sums as k_tile_flat<false> (which needs 4 ds_add_u32 per sample and lane), regrouped by voxel instead of by sample.
// ------------------------------------------------------------------------------------------------
#define GTX 8
#define GTY 8
#define GROWS 14          // rows a tile can touch: i0 spreads over <= 7 (|m00| + |m01|) <= 10.2 -> 11 values, + 3
#define GPITCH 65          // LDS row pitch in dwords: rows r, r+1, ... of one plane fall in different banks
#define GWAVES 4

struct GfC {
    int64_t fp0x, fp0y, fux, fuy, fdx, fdy;   // x, y of the 32.32 lattice  p = fp0 + ix fu + j fd
    float m00, m01, m10, m11;                 // (ix, j) = M ((x, y) - p0)
    float p0x, p0y, tau;
    int32_t n, zc, slot;
};

template <int NJ>      // samples per row that can reach a column: 3 for step >= 0.95 voxel, 6 for step >= 0.475
__global__ __launch_bounds__(GWAVES * 64) void k_adj_gather_flat(const GfC *__restrict__ cs, int n_proj, const float *__restrict__ proj,
                                                                 float *__restrict__ vol, TomoGeomC g, int xs, int xe)
{
    __shared__ float rows[GWAVES][GROWS * GPITCH];
    __shared__ float4 wtab[3][GWAVES][64];          // [group mod 3][projection of the group][column] = (i0, W0, W1, W2)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the work-group owns 8 x 8 voxel columns; its four waves take four consecutive 64-plane chunks of them
    const int x0 = xs + (int)blockIdx.z * GTX, y0 = (int)blockIdx.y * GTY, z0 = ((int)blockIdx.x * GWAVES + wv) * 64;
    if (x0 >= xe || y0 >= g.ny) return;                                 // uniform over the WORK-GROUP (barriers below)
    const bool zlive = z0 < g.nz;                                       // a wave past the volume still computes its share of tables
    // a lane is a voxel COLUMN (X, Y) with 64 plane accumulators, except while loading sinogram rows, where it is plane Zl
    const int X = x0 + (lane >> 3), Y = y0 + (lane & 7), Zl = z0 + lane;
    float *wrows = rows[wv];
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const float two_m32 = 2.3283064365386963e-10f;
    float acc[64];
#pragma unroll
    for (int p = 0; p < 64; ++p) acc[p] = 0.f;

    // ---- 1. the weight table of this lane's column for projection IPX -> wtab[GRP % 3][IPX % GWAVES][lane].  The table does not
    //         depend on z: the four waves share it, wave w computes the projections 4 g + w (one barrier per four projections).
    //   candidates: rows i0..i0+2, samples j0..j0+NJ-1 (the footprint |dx|,|dy| < 1 maps to |d ix| <= |m00|+|m01| < 1.5: three
    //   consecutive integers cover an interval shorter than 3; likewise |d j| <= |m10|+|m11| < NJ/2); W_k from exact 32.32
    //   positions relative to the voxel
#define G_TABLE(IPX)                                                                                                       \
    {                                                                                                                      \
        float4 t4 = {0.f, 0.f, 0.f, 0.f};                                                                                  \
        if ((IPX) < n_proj) {                                                                                              \
            const GfC &ct = cs[IPX];                                                                                       \
            const float qx = (float)X - ct.p0x, qy = (float)Y - ct.p0y;                                                    \
            const float a = ct.m00 * qx + ct.m01 * qy, b = ct.m10 * qx + ct.m11 * qy;                                      \
            const int i0 = (int)ceilf(a - (fabsf(ct.m00) + fabsf(ct.m01) + 5e-3f));                                        \
            const int j0 = (int)ceilf(b - (fabsf(ct.m10) + fabsf(ct.m11) + 5e-3f));                                        \
            int64_t rx = ct.fp0x + (int64_t)i0 * ct.fux + (int64_t)j0 * ct.fdx - ((int64_t)X << 32);                       \
            int64_t ry = ct.fp0y + (int64_t)i0 * ct.fuy + (int64_t)j0 * ct.fdy - ((int64_t)Y << 32);                       \
            float W[3];                                                                                                    \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                                \
                int64_t sx = rx, sy = ry;                                                                                  \
                float wsum = 0.f;                                                                                          \
                _Pragma("unroll") for (int mth = 0; mth < NJ; ++mth) {                                                     \
                    const int hx = (int)(sx >> 32), hy = (int)(sy >> 32);                                                  \
                    const float fx = (float)(unsigned)sx * two_m32, fy = (float)(unsigned)sy * two_m32;                    \
                    const float wx = hx == 0 ? 1.f - fx : (hx == -1 ? fx : 0.f); /* tent on [-1, 1) */                     \
                    const float wy = hy == 0 ? 1.f - fy : (hy == -1 ? fy : 0.f);                                           \
                    wsum += ((unsigned)(j0 + mth) < (unsigned)ct.n) ? wx * wy : 0.f;                                       \
                    sx += ct.fdx; sy += ct.fdy;                                                                            \
                }                                                                                                          \
                W[k] = ((unsigned)(i0 + k) < (unsigned)g.ndx) ? wsum : 0.f;                                                \
                rx += ct.fux; ry += ct.fuy;                                                                                \
            }                                                                                                              \
            t4.x = __builtin_bit_cast(float, i0); t4.y = W[0]; t4.z = W[1]; t4.w = W[2];                                   \
        }                                                                                                                  \
        wtab[((IPX) / GWAVES) % 3][(IPX) % GWAVES][lane] = t4;                                                             \
    }
    // ---- 2a. fetch projection IPX's table entry and ISSUE the 32 loads of the sinogram rows the tile can touch: rows
    //          ix_lo .. ix_lo+13 at this lane's PLANE (coalesced), from clamped -- always valid -- addresses, masked when used.
    //          Straight-line on purpose (with a branch per row every row waited for its own round trip to memory).  The
    //          loads are consumed one projection later: they fly while the previous projection accumulates.
    float4 tn;
    int ix_lo_n, nrows_n;
    float y0v[GROWS], y1v[GROWS];
#define G_SETUP(IPX)                                                                                                       \
    {                                                                                                                      \
        tn = wtab[((IPX) / GWAVES) % 3][(IPX) % GWAVES][lane];                                                             \
        const int i0s = __builtin_bit_cast(int, tn.x);                                                                     \
        ix_lo_n = __builtin_amdgcn_readfirstlane(wave_min_i32(i0s));                                                       \
        nrows_n = min(GROWS, __builtin_amdgcn_readfirstlane(wave_max_i32(i0s)) + 3 - ix_lo_n);                             \
        if (zlive) {                                                                                                       \
            const GfC &cn = cs[IPX];                                                                                       \
            const int iz0 = Zl - cn.zc;                                                                                    \
            const float *srow = proj + (size_t)cn.slot * n_det;                                                            \
            const float *p0 = srow + min(max(iz0, 0), g.ndz - 1), *p1 = srow + min(max(iz0 - 1, 0), g.ndz - 1);            \
            _Pragma("unroll") for (int r = 0; r < GROWS; ++r) {                                                            \
                const size_t ro = (size_t)min(max(ix_lo_n + r, 0), g.ndx - 1) * g.ndz; /* wave-uniform */                  \
                y0v[r] = p0[ro];                                                                                           \
                y1v[r] = p1[ro];                                                                                           \
            }                                                                                                              \
        }                                                                                                                  \
    }
    const int n_grp = (n_proj + GWAVES - 1) / GWAVES;
    if (n_grp > 0) {
        G_TABLE(wv)                                                     // group 0
        __syncthreads();
        G_SETUP(0)
    }
    for (int grp = 0; grp < n_grp; ++grp) {
        if (grp + 1 < n_grp) G_TABLE((grp + 1) * GWAVES + wv)           // next group's tables: a third buffer, nobody reads it yet
        __syncthreads();                                                // ... and everybody is done with group grp - 1's buffer
        for (int ip = grp * GWAVES; ip < min(n_proj, (grp + 1) * GWAVES); ++ip) {
            const GfC &c = cs[ip];
            const float4 t = tn;
            const int i0 = __builtin_bit_cast(int, t.x), ix_lo = ix_lo_n, nrows = nrows_n;
            const float W0 = t.y, W1 = t.z, W2 = t.w;
            const bool hit = zlive && __any(W0 != 0.f || W1 != 0.f || W2 != 0.f);   // else this projection's rays miss the tile
            // ---- 2b. z-lerp the rows loaded one projection ago into the wave's LDS rows (lane = plane)
            if (hit) {
                const int iz0 = Zl - c.zc, iz1 = iz0 - 1;
                const bool ok0 = iz0 >= 0 && iz0 < g.ndz, ok1 = iz1 >= 0 && iz1 < g.ndz;
#pragma unroll
                for (int r = 0; r < GROWS; ++r) {
                    const bool rowok = r < nrows && (unsigned)(ix_lo + r) < (unsigned)g.ndx;   // wave-uniform
                    const float a0 = (rowok && ok0) ? y0v[r] : 0.f, a1 = (rowok && ok1) ? y1v[r] : 0.f;
                    wrows[r * GPITCH + lane] = fmaf(c.tau, a1 - a0, a0);
                }
            }
            if (ip + 1 < n_proj) G_SETUP(ip + 1)                        // the next group's table is already published
            // ---- 3. accumulate, lane = column: its three rows start at slot0; plane p is an immediate offset.  The LDS rows were
            //         written by this same wave (LDS operations of a wave execute in order), no other wave touches them.
            if (hit) {
                const int slot0 = min(max(i0 - ix_lo, 0), GROWS - 3);       // <= nrows - 3 by construction; clamped for safety
                const float *q = wrows + slot0 * GPITCH;
#pragma unroll
                for (int p = 0; p < 64; ++p) {
                    acc[p] = fmaf(W0, q[p], acc[p]);
                    acc[p] = fmaf(W1, q[GPITCH + p], acc[p]);
                    acc[p] = fmaf(W2, q[2 * GPITCH + p], acc[p]);
                }
            }
        }
    }
#undef G_TABLE
#undef G_SETUP
    // ---- store: the lane's column is 64 consecutive floats of the volume
    if (zlive && X < xe && Y < g.ny) {
        float *dst = vol + ((size_t)X * g.ny + Y) * g.nz + z0;
        if (z0 + 64 <= g.nz && (g.nz & 3) == 0) {
#pragma unroll
            for (int p = 0; p < 64; p += 4) {
                float4 v = *(float4 *)(dst + p);
                v.x += acc[p]; v.y += acc[p + 1]; v.z += acc[p + 2]; v.w += acc[p + 3];
                *(float4 *)(dst + p) = v;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 64; ++p)
                if (z0 + p < g.nz) dst[p] += acc[p];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// voxel-driven bilinear back-projector (src/back_projection.f90:25-32): one voxel per work-item,
// lanes along z, loop over projections with the accumulator in a register; the voxel centre is
// transformed on the fly (the reference re-reads a (3,n_vox) voxel_centers array per projection).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bp_voxel(const BpC *__restrict__ cs, int n_proj, const float *__restrict__ det,
                                                  float *__restrict__ vol, TomoGeomC g)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int iz = blockIdx.x * 64 + lane, iy = blockIdx.y * 4 + wv, ix = blockIdx.z;
    if (iy >= g.ny || iz >= g.nz) return;
    const size_t img = (size_t)g.ndx * g.ndz;
    float acc = 0.f;
    for (int ip = 0; ip < n_proj; ++ip) {
        const BpC c = cs[ip];
        const double u = c.u0 + ix * c.ux + iy * c.uy + iz * c.uz;
        const double v = c.v0 + ix * c.vx + iy * c.vy + iz * c.vz;
        if (!(u >= -1.0 && u < (double)g.ndx && v >= -1.0 && v < (double)g.ndz)) continue;
        const double fu = floor(u), fv = floor(v);
        const int fx = (int)fu, fz = (int)fv;
        const float ax = (float)(u - fu), az = (float)(v - fv);      // external_back_projection.f90:47-48
        const float *im = det + (size_t)ip * img;
        const bool x0 = fx >= 0, x1 = fx + 1 < g.ndx, z0 = fz >= 0, z1 = fz + 1 < g.ndz;
        float s = 0.f;                                                 // :54-65, per-pixel bounds tests
        if (x0 && z0) s += im[(size_t)fx * g.ndz + fz] * (1.f - ax) * (1.f - az);
        if (x1 && z0) s += im[(size_t)(fx + 1) * g.ndz + fz] * ax * (1.f - az);
        if (x0 && z1) s += im[(size_t)fx * g.ndz + fz + 1] * (1.f - ax) * az;
        if (x1 && z1) s += im[(size_t)(fx + 1) * g.ndz + fz + 1] * ax * az;
        acc += s;                                                      // back_projection.f90:31
    }
    vol[((size_t)ix * g.ny + iy) * g.nz + iz] = acc;
}

// ------------------------------------------------------------------------------------------------
// projection + 6-DoF pose gradient.  Per sample only the interpolant's spatial gradient is formed
// (3 values); S0 = sum_j grad_j and S1 = sum_j sf_j*grad_j are accumulated and the per-ray 9x3 pose
// Jacobian is applied once (same algebra as src/ray_wt_grad.f90:136-149).
// FUSED: multiply by the residual and reduce to 7 numbers per projection.
// ------------------------------------------------------------------------------------------------
template <bool FUSED>
__global__ __launch_bounds__(256) void k_proj_grad(const ProjC *__restrict__ pcs, const GradC *__restrict__ gcs,
                                                   const float *__restrict__ vp, float *__restrict__ proj,
                                                   float *__restrict__ grad, const float *__restrict__ bvec,
                                                   float *__restrict__ resid, double *__restrict__ red, TomoGeomC g,
                                                   int row_order)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ix = blockIdx.y * 4 + wv, ip = blockIdx.z;
    int iz = blockIdx.x * 64 + lane;
    const bool valid = (ix < g.ndx) && (iz < g.ndz);
    const int ixc = min(ix, g.ndx - 1);
    if (iz >= g.ndz) iz = g.ndz - 1;
    const ProjC &c = pcs[ip];
    const GradC &gc = gcs[ip];
    RayCtx r;
    ray_setup(c, g, ixc, iz, valid, r);
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const int64_t sy = g.nzp, sx = (int64_t)g.nyp * g.nzp;
    const float sfs = (float)(g.step / c.rlen);      // sf_j = (j*step)/|r_0|   ray_voxel_utilities.py:151
    double val = 0.0, s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
    for (int jb = r.j0; jb < r.j1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const float *base = vp + ((int64_t)(ia[0] + TOMO_HALO) * sx + (int64_t)(ia[1] + TOMO_HALO) * sy + (ia[2] + TOMO_HALO));
        const int cnt = min(TOMO_JB, r.j1 - jb);
        float av = 0.f, a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
        for (int jj = 0; jj < cnt; ++jj) {
            const float t = (float)jj;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
            const float wx = x - fx, wy = y - fy, wz = z - fz;
            const float *q = base + ((int64_t)(int)fx * sx + (int64_t)(int)fy * sy + (int)fz);
            const float v000 = q[0], v001 = q[1], v010 = q[sy], v011 = q[sy + 1];
            const float v100 = q[sx], v101 = q[sx + 1], v110 = q[sx + sy], v111 = q[sx + sy + 1];
            const float d00 = v001 - v000, d01 = v011 - v010, d10 = v101 - v100, d11 = v111 - v110;
            const float c00 = fmaf(wz, d00, v000), c01 = fmaf(wz, d01, v010), c10 = fmaf(wz, d10, v100), c11 = fmaf(wz, d11, v110);
            const float dz0 = fmaf(wy, d01 - d00, d00), dz1 = fmaf(wy, d11 - d10, d10);
            const float gz = fmaf(wx, dz1 - dz0, dz0);
            const float dy0 = c01 - c00, dy1 = c11 - c10;
            const float e0 = fmaf(wy, dy0, c00), e1 = fmaf(wy, dy1, c10);
            const float gy = fmaf(wx, dy1 - dy0, dy0);
            const float gx = e1 - e0;
            av += fmaf(wx, gx, e0);
            const float sf = (float)(jb + jj) * sfs;
            a0x += gx; a0y += gy; a0z += gz;
            a1x = fmaf(sf, gx, a1x); a1y = fmaf(sf, gy, a1y); a1z = fmaf(sf, gz, a1z);
        }
        val += (double)av;
        s0[0] += (double)a0x; s0[1] += (double)a0y; s0[2] += (double)a0z;
        s1[0] += (double)a1x; s1[1] += (double)a1y; s1[2] += (double)a1z;
    }
    // per-ray pose Jacobian (utilities/ray_voxel_utilities.py:38-49)
    const double s[3] = {gc.s00[0] + ixc * gc.sdx, gc.s00[1], gc.s00[2] + iz * gc.sdz};
    double qv[3], gk[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) qv[a] = gc.ry[a][0] * s[0] + gc.ry[a][1] * s[1] + gc.ry[a][2] * s[2] + gc.t[a];
#pragma unroll
    for (int k = 0; k < 3; ++k) gk[k] = gc.rzx[0][k] * s0[0] + gc.rzx[1][k] * s0[1] + gc.rzx[2][k] * s0[2];
    gk[3] = gk[4] = gk[5] = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double d3 = gc.a3[a][0] * qv[0] + gc.a3[a][1] * qv[1] + gc.a3[a][2] * qv[2];
        const double d4 = gc.a4[a][0] * qv[0] + gc.a4[a][1] * qv[1] + gc.a4[a][2] * qv[2];
        const double d5 = gc.a5[a][0] * s[0] + gc.a5[a][1] * s[1] + gc.a5[a][2] * s[2];
        gk[3] += d3 * s0[a] + gc.app[0][a] * s1[a];
        gk[4] += d4 * s0[a] + gc.app[1][a] * s1[a];
        gk[5] += d5 * s0[a] + gc.app[2][a] * s1[a];
    }
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const size_t ray = (size_t)ixc * g.ndz + iz;
    if (!FUSED) {
        if (valid) {
            proj[ray] = (float)val;
            if (row_order == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) grad[k * n_det + ray] = (float)gk[k];
            } else {   // tx,ty,tz,alpha,beta,phi  (src/external_forward_projection.f90:56-69)
                grad[0 * n_det + ray] = (float)gk[0]; grad[1 * n_det + ray] = (float)gk[1]; grad[2 * n_det + ray] = (float)gk[2];
                grad[3 * n_det + ray] = (float)gk[4]; grad[4 * n_det + ray] = (float)gk[5]; grad[5 * n_det + ray] = (float)gk[3];
            }
        }
    } else {
        double part[7] = {0, 0, 0, 0, 0, 0, 0};
        if (valid) {
            const float pv = (float)val;                                  // projection_operators.py:119 cast
            const double res = (double)(bvec[(size_t)gc.b_row * n_det + ray] - pv);   // alignment_functions.py:23
            if (resid) resid[(size_t)gc.slot * n_det + ray] = (float)res;
            part[0] = 0.5 * res * res;                                    // :124
#pragma unroll
            for (int k = 0; k < 6; ++k) part[1 + k] = -(double)(float)gk[k] * res;   // :35,146
        }
        __shared__ double sh[4][7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double w = wave_sum_d(part[k]);
            if (lane == 0) sh[wv][k] = w;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const int k = threadIdx.x;
            atomicAdd(&red[(size_t)gc.slot * 7 + k], sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// projection + gradient, variant 2: the same sums as k_proj_grad with a cheaper sample (about 40 VALU instead of 59).
//   * addressing as in k_fwd_v2: the sample blocks are walked in wave-uniform steps, the block bases are SGPR pairs and each
//     lane carries ONE 32-bit byte offset for all eight corners (saddr + voffset loads): 3 integer ops instead of 14 64-bit ones;
//   * eight dword gathers instead of four dwordx2 (see the note in the kernel: 3.5x cheaper in the L1 pipeline);
//   * the lerps are written on (z, z+1) register pairs -- y first, then x, then z -- so that they map 1:1 onto
//     v_pk_add_f32 / v_pk_fma_f32 without register shuffles.
// (A version that loaded only the four lower-z corners and took the upper ones from the neighbouring lane by a lane shift was
// measured 30 % SLOWER than variant 1: the kernel is VALU-bound, not gather-bound, and the shifts cost more than the loads.)
// Only lanes inside their own [lo, hi) execute loads, all at addresses of samples inside the padded volume.
// ------------------------------------------------------------------------------------------------
template <bool FUSED>
__global__ __launch_bounds__(256) void k_proj_grad_v2(const ProjC *__restrict__ pcs, const GradC *__restrict__ gcs,
                                                      const float *__restrict__ vp, float *__restrict__ proj,
                                                      float *__restrict__ grad, const float *__restrict__ bvec,
                                                      float *__restrict__ resid, double *__restrict__ red, TomoGeomC g,
                                                      int row_order)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Grid = (ix groups, projections, detector-z chunks), z chunk SLOWEST: all projections of one 64-row detector slab run
    // back to back, so the volume slab they read (n^2 * 64 cells, 64 MB at 512^3) stays in the 256 MB Infinity Cache
    // instead of the whole volume streaming from HBM once per projection.  Workgroups are dealt to the 8 XCDs round-robin
    // in dispatch order, so the swizzle gives each XCD a contiguous range of ix groups (neighbouring rays share L2 lines).
    // (Volumes that fit the cache anyway keep the plain order row_order < 16: z chunk fastest, projection slowest.)
    int ix, ip, iz;
    if (row_order & 16) {
        const int nxg = gridDim.x;
        const int xg = ((nxg & 7) == 0) ? ((int)(blockIdx.x & 7) * (nxg >> 3) + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
        ix = xg * 4 + wv, ip = blockIdx.y, iz = blockIdx.z * 64 + lane;
    } else {
        ix = blockIdx.y * 4 + wv, ip = blockIdx.z, iz = blockIdx.x * 64 + lane;
    }
    row_order &= 15;
    const bool valid = (ix < g.ndx) && (iz < g.ndz);
    const int ixc = min(ix, g.ndx - 1);
    if (iz >= g.ndz) iz = g.ndz - 1;
    const ProjC &c = pcs[ip];
    const GradC &gc = gcs[ip];
    RayCtx r;
    ray_setup(c, g, ixc, iz, valid, r);
    const bool nonempty = r.j1 > r.j0;
    const int J0 = __builtin_amdgcn_readfirstlane(wave_min_i32(nonempty ? r.j0 : INT_MAX));
    const int J1 = __builtin_amdgcn_readfirstlane(wave_max_i32(nonempty ? r.j1 : 0));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const uint32_t sy4 = (uint32_t)g.nzp * 4u, sx4 = (uint32_t)g.nyp * (uint32_t)g.nzp * 4u;
    const float sfs = (float)(g.step / c.rlen);
    // Eight DWORD gathers per sample, on purpose: with lanes on consecutive z cells a wave-wide global_load_dword costs
    // 4.8 cycles of the CU's texture-address/L1 pipeline, a dwordx2 (or x4) 17 (tools/gather_bench.hip), and that pipeline
    // is what bounds this kernel (TA_BUSY = 100 %, profiles/).  The z + 1 bases are offset by an SGPR the compiler cannot
    // see through, or it would fuse each (z, z + 1) pair back into one dwordx2.
    int four;
    asm volatile("s_mov_b32 %0, 4" : "=s"(four));
    double val = 0.0, s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
    for (int jb = J0; jb < J1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const int64_t lin = ((int64_t)(ia[0] + TOMO_HALO) * g.nyp + (ia[1] + TOMO_HALO)) * g.nzp + (ia[2] + TOMO_HALO);
        const int64_t lin0 = readfirstlane_i64(lin);
        const int delta = (int)(lin - lin0);
        const int m = __builtin_amdgcn_readfirstlane(wave_min_i32(delta));
        const char *sb00 = (const char *)(vp + (lin0 + m));
        const char *sb01 = sb00 + sy4;
        const char *sb10 = sb00 + sx4;
        const char *sb11 = sb10 + sy4;
        const char *sc00 = sb00 + four, *sc01 = sb01 + four, *sc10 = sb10 + four, *sc11 = sb11 + four;   // the z + 1 corners
        const uint32_t off0 = (uint32_t)(delta - m) * 4u;
        const int lo = max(r.j0, jb) - jb, hi = min(r.j1, jb + TOMO_JB) - jb;      // this lane's samples of the block
        float av = 0.f;
        f32x2 a0xy = {0.f, 0.f}, a1xy = {0.f, 0.f}, az = {0.f, 0.f};              // (S0x, S0y), (S1x, S1y), (S0z, S1z)
        const float sfb = (float)jb * sfs;
        // two samples per trip: all 16 gathers are issued before the first value is used (the kernel waits on memory 3/4 of
        // the time; this doubles the loads in flight per wave).  An odd tail re-reads sample A's address and is masked out.
        for (int jj = lo; jj < hi; jj += 2) {
            const float ta = (float)jj, tb = ta + 1.f;
            const bool two = jj + 1 < hi;
            const float xa = fmaf(ta, dxf, f0[0]), ya = fmaf(ta, dyf, f0[1]), za = fmaf(ta, dzf, f0[2]);
            const float xb = fmaf(tb, dxf, f0[0]), yb = fmaf(tb, dyf, f0[1]), zb = fmaf(tb, dzf, f0[2]);
            const float fxa = floorf(xa), fya = floorf(ya), fza = floorf(za);
            const float fxb = floorf(xb), fyb = floorf(yb), fzb = floorf(zb);
            const uint32_t voa = off0 + __umul24((uint32_t)(int)fxa, sx4) + __umul24((uint32_t)(int)fya, sy4) + ((uint32_t)(int)fza << 2);
            const uint32_t vob_ = off0 + __umul24((uint32_t)(int)fxb, sx4) + __umul24((uint32_t)(int)fyb, sy4) + ((uint32_t)(int)fzb << 2);
            const uint32_t vob = two ? vob_ : voa;
            const f32x2 a00 = {*(const float *)(sb00 + voa), *(const float *)(sc00 + voa)};
            const f32x2 a01 = {*(const float *)(sb01 + voa), *(const float *)(sc01 + voa)};
            const f32x2 a10 = {*(const float *)(sb10 + voa), *(const float *)(sc10 + voa)};
            const f32x2 a11 = {*(const float *)(sb11 + voa), *(const float *)(sc11 + voa)};
            const f32x2 b00 = {*(const float *)(sb00 + vob), *(const float *)(sc00 + vob)};
            const f32x2 b01 = {*(const float *)(sb01 + vob), *(const float *)(sc01 + vob)};
            const f32x2 b10 = {*(const float *)(sb10 + vob), *(const float *)(sc10 + vob)};
            const f32x2 b11 = {*(const float *)(sb11 + vob), *(const float *)(sc11 + vob)};
            {
                const float wx = xa - fxa, wy = ya - fya, wz = za - fza;
                const f32x2 dy0 = a01 - a00, dy1 = a11 - a10;          // d/dy on the x = 0 / x = 1 faces, at z and z + 1
                const f32x2 c0 = a00 + wy * dy0, c1 = a10 + wy * dy1;  // y-lerped
                const f32x2 dx = c1 - c0;                              // d/dx at z, z + 1
                const f32x2 e = c0 + wx * dx;                          // x,y-lerped value at z, z + 1
                const f32x2 dyx = dy0 + wx * (dy1 - dy0);              // d/dy at z, z + 1
                const float gz = e.y - e.x;
                const float gx = fmaf(wz, dx.y - dx.x, dx.x), gy = fmaf(wz, dyx.y - dyx.x, dyx.x);
                av += fmaf(wz, gz, e.x);
                const float sf = fmaf(ta, sfs, sfb);                   // (jb + jj) * step / |r0|, one rounding
                const f32x2 gxy = {gx, gy}, one_sf = {1.f, sf};
                a0xy += gxy;
                a1xy += sf * gxy;
                az += one_sf * gz;
            }
            {
                const float keep = two ? 1.f : 0.f;
                const float wx = xb - fxb, wy = yb - fyb, wz = zb - fzb;
                const f32x2 dy0 = b01 - b00, dy1 = b11 - b10;
                const f32x2 c0 = b00 + wy * dy0, c1 = b10 + wy * dy1;
                const f32x2 dx = c1 - c0;
                const f32x2 e = c0 + wx * dx;
                const f32x2 dyx = dy0 + wx * (dy1 - dy0);
                const float gz = keep * (e.y - e.x);
                const float gx = keep * fmaf(wz, dx.y - dx.x, dx.x), gy = keep * fmaf(wz, dyx.y - dyx.x, dyx.x);
                av = fmaf(keep, fmaf(wz, e.y - e.x, e.x), av);
                const float sf = fmaf(tb, sfs, sfb);
                const f32x2 gxy = {gx, gy}, one_sf = {1.f, sf};
                a0xy += gxy;
                a1xy += sf * gxy;
                az += one_sf * gz;
            }
        }
        val += (double)av;
        s0[0] += (double)a0xy.x; s0[1] += (double)a0xy.y; s0[2] += (double)az.x;
        s1[0] += (double)a1xy.x; s1[1] += (double)a1xy.y; s1[2] += (double)az.y;
    }
    const double s[3] = {gc.s00[0] + ixc * gc.sdx, gc.s00[1], gc.s00[2] + iz * gc.sdz};
    double qv[3], gk[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) qv[a] = gc.ry[a][0] * s[0] + gc.ry[a][1] * s[1] + gc.ry[a][2] * s[2] + gc.t[a];
#pragma unroll
    for (int k = 0; k < 3; ++k) gk[k] = gc.rzx[0][k] * s0[0] + gc.rzx[1][k] * s0[1] + gc.rzx[2][k] * s0[2];
    gk[3] = gk[4] = gk[5] = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double d3 = gc.a3[a][0] * qv[0] + gc.a3[a][1] * qv[1] + gc.a3[a][2] * qv[2];
        const double d4 = gc.a4[a][0] * qv[0] + gc.a4[a][1] * qv[1] + gc.a4[a][2] * qv[2];
        const double d5 = gc.a5[a][0] * s[0] + gc.a5[a][1] * s[1] + gc.a5[a][2] * s[2];
        gk[3] += d3 * s0[a] + gc.app[0][a] * s1[a];
        gk[4] += d4 * s0[a] + gc.app[1][a] * s1[a];
        gk[5] += d5 * s0[a] + gc.app[2][a] * s1[a];
    }
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const size_t ray = (size_t)ixc * g.ndz + iz;
    if (!FUSED) {
        if (valid) {
            proj[ray] = (float)val;
            if (row_order == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) grad[k * n_det + ray] = (float)gk[k];
            } else {
                grad[0 * n_det + ray] = (float)gk[0]; grad[1 * n_det + ray] = (float)gk[1]; grad[2 * n_det + ray] = (float)gk[2];
                grad[3 * n_det + ray] = (float)gk[4]; grad[4 * n_det + ray] = (float)gk[5]; grad[5 * n_det + ray] = (float)gk[3];
            }
        }
    } else {
        double part[7] = {0, 0, 0, 0, 0, 0, 0};
        if (valid) {
            const float pv = (float)val;
            const double res = (double)(bvec[(size_t)gc.b_row * n_det + ray] - pv);
            if (resid) resid[(size_t)gc.slot * n_det + ray] = (float)res;
            part[0] = 0.5 * res * res;
#pragma unroll
            for (int k = 0; k < 6; ++k) part[1 + k] = -(double)(float)gk[k] * res;
        }
        __shared__ double sh[4][7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double w = wave_sum_d(part[k]);
            if (lane == 0) sh[wv][k] = w;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const int k = threadIdx.x;
            atomicAdd(&red[(size_t)gc.slot * 7 + k], sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// projection + gradient, variant 3: four gathers per sample instead of eight.  Lanes run along detector-z, so lane l's
// upper-z corners are normally lane l+1's lower-z corners: every lane gathers its four lower-z corners and receives the upper
// ones from its neighbour with a DPP wave shift (v_mov_b32_dpp wave_shl:1 -- a full-rate VALU op on gfx950, tools/dpp_check.hip);
// where the neighbour's address is not mine + 4 (tilt-induced row steps, lane 63) the lane loads them itself.  That decision
// needs only the ADDRESSES, so the fallback loads are issued together with the main ones.
// For the shift to read live registers the sample loop is wave-uniform over the union of the lanes' ranges; a lane outside
// its own range still loads -- at its own ray's nearest in-range sample (always inside the padded volume), or, with no sample
// in the block at all, at the first sample of the first lane that has one -- and its contribution is masked.  Since every
// lane's values really are the volume at the address it advertises, "neighbour address == mine + 4" is all a lane must check.
// The gathers are what bounds the gradient kernels under tilt (TA busy 100 %): time grows linearly with the tilt because
// a 16-lane group then straddles more volume rows; halving the gathers halves that term.
// ------------------------------------------------------------------------------------------------
// lane l <- lane l + 1; lane 63 <- 0 (bound_ctrl: no `old` register to initialise)
__device__ __forceinline__ int dpp_shl1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ float dpp_shl1_f(float v) { return __builtin_bit_cast(float, dpp_shl1_i(__builtin_bit_cast(int, v))); }

template <bool FUSED>
__global__ __launch_bounds__(256) void k_proj_grad_v3(const ProjC *__restrict__ pcs, const GradC *__restrict__ gcs,
                                                      const float *__restrict__ vp, float *__restrict__ proj,
                                                      float *__restrict__ grad, const float *__restrict__ bvec,
                                                      float *__restrict__ resid, double *__restrict__ red, TomoGeomC g,
                                                      int row_order)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ix, ip, iz;
    if (row_order & 16) {                              // cache-ordered grid, see k_proj_grad_v2
        const int nxg = gridDim.x;
        const int xg = ((nxg & 7) == 0) ? ((int)(blockIdx.x & 7) * (nxg >> 3) + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
        ix = xg * 4 + wv, ip = blockIdx.y, iz = blockIdx.z * 64 + lane;
    } else {
        ix = blockIdx.y * 4 + wv, ip = blockIdx.z, iz = blockIdx.x * 64 + lane;
    }
    row_order &= 15;
    const bool valid = (ix < g.ndx) && (iz < g.ndz);
    const int ixc = min(ix, g.ndx - 1);
    if (iz >= g.ndz) iz = g.ndz - 1;
    const ProjC &c = pcs[ip];
    const GradC &gc = gcs[ip];
    RayCtx r;
    ray_setup(c, g, ixc, iz, valid, r);
    const bool nonempty = r.j1 > r.j0;
    const int J0 = __builtin_amdgcn_readfirstlane(wave_min_i32(nonempty ? r.j0 : INT_MAX));
    const int J1 = __builtin_amdgcn_readfirstlane(wave_max_i32(nonempty ? r.j1 : 0));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const uint32_t sy4 = (uint32_t)g.nzp * 4u, sx4 = (uint32_t)g.nyp * (uint32_t)g.nzp * 4u;
    const float sfs = (float)(g.step / c.rlen);
    double val = 0.0, s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
    for (int jb = J0; jb < J1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const int64_t lin = ((int64_t)(ia[0] + TOMO_HALO) * g.nyp + (ia[1] + TOMO_HALO)) * g.nzp + (ia[2] + TOMO_HALO);
        const int64_t lin0 = readfirstlane_i64(lin);
        const int delta = (int)(lin - lin0);
        const int m = __builtin_amdgcn_readfirstlane(wave_min_i32(delta));
        const char *sb00 = (const char *)(vp + (lin0 + m));
        const char *sb01 = sb00 + sy4;
        const char *sb10 = sb00 + sx4;
        const char *sb11 = sb10 + sy4;
        const uint32_t off0 = (uint32_t)(delta - m) * 4u;
        const int lo = max(r.j0, jb) - jb, hi = min(r.j1, jb + TOMO_JB) - jb;      // this lane's samples of the block
        const bool has = hi > lo;
        const unsigned long long hm = __ballot(has);
        if (hm == 0ull) continue;                                                  // wave-uniform
        const int LO = __builtin_amdgcn_readfirstlane(wave_min_i32(has ? lo : INT_MAX));
        const int HI = __builtin_amdgcn_readfirstlane(wave_max_i32(has ? hi : 0));
        // a lane with no sample in this block gathers where the first lane that has one takes its first sample
        uint32_t borrow;
        {
            const float t = (float)lo;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const uint32_t mine = off0 + __umul24((uint32_t)(int)floorf(x), sx4) + __umul24((uint32_t)(int)floorf(y), sy4) + ((uint32_t)(int)floorf(z) << 2);
            borrow = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(hm));
        }
        const int lo_c = has ? lo : 0, hi_c = has ? hi - 1 : 0;
        const float sfb = (float)jb * sfs;
        float av = 0.f;
        f32x2 a0xy = {0.f, 0.f}, a1xy = {0.f, 0.f}, az = {0.f, 0.f};              // (S0x, S0y), (S1x, S1y), (S0z, S1z)
        // issue: addresses, the four gathers, and -- decided from the addresses alone -- the fallback gathers.
        // (Macros over plain scalars on purpose: a struct passed to helper lambdas was promoted to an LDS alloca, which put
        // a store of every loaded value -- hence a vmcnt(0) wait -- between the two samples' loads.)
#define GS_DECL(S) float S##v000, S##v010, S##v100, S##v110, S##f001, S##f011, S##f101, S##f111, S##wx, S##wy, S##wz, S##t; /* f*: set and read only where fb */ \
                   bool S##act, S##fb
#define GS_ISSUE(S, JJ)                                                                                                            \
    {                                                                                                                              \
        const int jc = min(max((JJ), lo_c), hi_c); /* own ray's nearest in-range sample */                                         \
        S##act = has && jc == (JJ);                                                                                                \
        S##t = (float)jc;                                                                                                          \
        const float x = fmaf(S##t, dxf, f0[0]), y = fmaf(S##t, dyf, f0[1]), z = fmaf(S##t, dzf, f0[2]);                            \
        const float fx = floorf(x), fy = floorf(y), fz = floorf(z);                                                                \
        S##wx = x - fx; S##wy = y - fy; S##wz = z - fz;                                                                            \
        const uint32_t vo_own = off0 + __umul24((uint32_t)(int)fx, sx4) + __umul24((uint32_t)(int)fy, sy4) + ((uint32_t)(int)fz << 2); \
        const uint32_t vo = has ? vo_own : borrow;                                                                                 \
        S##v000 = *(const float *)(sb00 + vo); S##v010 = *(const float *)(sb01 + vo);                                              \
        S##v100 = *(const float *)(sb10 + vo); S##v110 = *(const float *)(sb11 + vo);                                              \
        const uint32_t nb = (uint32_t)dpp_shl1_i((int)vo); /* lane 63 receives 0: never vo + 4 */                                 \
        const uint32_t vo4 = vo + 4u;                                                                                              \
        S##fb = S##act && nb != vo4;                                                                                               \
        if (S##fb) { /* my upper-z cell is not the neighbour's lower-z cell */                                                     \
            S##f001 = *(const float *)(sb00 + vo4); S##f011 = *(const float *)(sb01 + vo4);                                        \
            S##f101 = *(const float *)(sb10 + vo4); S##f111 = *(const float *)(sb11 + vo4);                                        \
        }                                                                                                                          \
    }
        // consume: the shifts run with every lane enabled (a DPP source lane that is masked off delivers nothing): take them
        // first, unconditionally, then select
#define GS_CONSUME(S)                                                                                                              \
    {                                                                                                                              \
        const float n001 = dpp_shl1_f(S##v000), n011 = dpp_shl1_f(S##v010), n101 = dpp_shl1_f(S##v100), n111 = dpp_shl1_f(S##v110);   \
        const float v001 = S##fb ? S##f001 : n001, v011 = S##fb ? S##f011 : n011, v101 = S##fb ? S##f101 : n101, v111 = S##fb ? S##f111 : n111; \
        const f32x2 p00 = {S##v000, v001}, p01 = {S##v010, v011}, p10 = {S##v100, v101}, p11 = {S##v110, v111};                    \
        const f32x2 dy0 = p01 - p00, dy1 = p11 - p10;             /* d/dy on the x = 0 / x = 1 faces, at z and z + 1 */            \
        const f32x2 c0 = p00 + S##wy * dy0, c1 = p10 + S##wy * dy1; /* y-lerped */                                                 \
        const f32x2 dx = c1 - c0;                                 /* d/dx at z, z + 1 */                                           \
        const f32x2 e = c0 + S##wx * dx;                          /* x,y-lerped value at z, z + 1 */                               \
        const f32x2 dyx = dy0 + S##wx * (dy1 - dy0);              /* d/dy at z, z + 1 */                                           \
        const float keep = S##act ? 1.f : 0.f;                                                                                     \
        const float gz = keep * (e.y - e.x);                                                                                       \
        const float gx = keep * fmaf(S##wz, dx.y - dx.x, dx.x), gy = keep * fmaf(S##wz, dyx.y - dyx.x, dyx.x);                     \
        av = fmaf(keep, fmaf(S##wz, e.y - e.x, e.x), av);                                                                          \
        const float sf = fmaf(S##t, sfs, sfb);                    /* (jb + jj) * step / |r0|, one rounding */                      \
        const f32x2 gxy = {gx, gy}, one_sf = {1.f, sf};                                                                            \
        a0xy += gxy;                                                                                                               \
        a1xy += sf * gxy;                                                                                                          \
        az += one_sf * gz;                                                                                                         \
    }
        for (int jj = LO; jj < HI; jj += 2) {                                      // wave-uniform trip count; two samples in flight
            GS_DECL(a_);
            GS_DECL(b_);
            GS_ISSUE(a_, jj)
            GS_ISSUE(b_, jj + 1)                                                   // past the end: clamped address, act = false
            GS_CONSUME(a_)
            GS_CONSUME(b_)
        }
#undef GS_DECL
#undef GS_ISSUE
#undef GS_CONSUME
        val += (double)av;
        s0[0] += (double)a0xy.x; s0[1] += (double)a0xy.y; s0[2] += (double)az.x;
        s1[0] += (double)a1xy.x; s1[1] += (double)a1xy.y; s1[2] += (double)az.y;
    }
    const double s[3] = {gc.s00[0] + ixc * gc.sdx, gc.s00[1], gc.s00[2] + iz * gc.sdz};
    double qv[3], gk[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) qv[a] = gc.ry[a][0] * s[0] + gc.ry[a][1] * s[1] + gc.ry[a][2] * s[2] + gc.t[a];
#pragma unroll
    for (int k = 0; k < 3; ++k) gk[k] = gc.rzx[0][k] * s0[0] + gc.rzx[1][k] * s0[1] + gc.rzx[2][k] * s0[2];
    gk[3] = gk[4] = gk[5] = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double d3 = gc.a3[a][0] * qv[0] + gc.a3[a][1] * qv[1] + gc.a3[a][2] * qv[2];
        const double d4 = gc.a4[a][0] * qv[0] + gc.a4[a][1] * qv[1] + gc.a4[a][2] * qv[2];
        const double d5 = gc.a5[a][0] * s[0] + gc.a5[a][1] * s[1] + gc.a5[a][2] * s[2];
        gk[3] += d3 * s0[a] + gc.app[0][a] * s1[a];
        gk[4] += d4 * s0[a] + gc.app[1][a] * s1[a];
        gk[5] += d5 * s0[a] + gc.app[2][a] * s1[a];
    }
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const size_t ray = (size_t)ixc * g.ndz + iz;
    if (!FUSED) {
        if (valid) {
            proj[ray] = (float)val;
            if (row_order == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) grad[k * n_det + ray] = (float)gk[k];
            } else {
                grad[0 * n_det + ray] = (float)gk[0]; grad[1 * n_det + ray] = (float)gk[1]; grad[2 * n_det + ray] = (float)gk[2];
                grad[3 * n_det + ray] = (float)gk[4]; grad[4 * n_det + ray] = (float)gk[5]; grad[5 * n_det + ray] = (float)gk[3];
            }
        }
    } else {
        double part[7] = {0, 0, 0, 0, 0, 0, 0};
        if (valid) {
            const float pv = (float)val;
            const double res = (double)(bvec[(size_t)gc.b_row * n_det + ray] - pv);
            if (resid) resid[(size_t)gc.slot * n_det + ray] = (float)res;
            part[0] = 0.5 * res * res;
#pragma unroll
            for (int k = 0; k < 6; ++k) part[1 + k] = -(double)(float)gk[k] * res;
        }
        __shared__ double sh[4][7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double w = wave_sum_d(part[k]);
            if (lane == 0) sh[wv][k] = w;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const int k = threadIdx.x;
            atomicAdd(&red[(size_t)gc.slot * 7 + k], sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// how far the detector-z direction leaves the volume's z rows: |w_x| + |w_y| voxels per detector row.  The gradient kernels'
// lanes run along detector z, so 64 x this is the drift of a wave across volume rows; beyond ~1 voxel the eight-gather kernel
// (v2) slows down linearly and the four-gather + lane-shift kernel (v3) wins (measured crossover: |alpha| + |beta| ~ 0.9 deg)
static inline bool grad_pose_is_tilted(const ProjC &c) { return fabs(c.w[0]) + fabs(c.w[1]) > 0.0157; }

// Stage the per-projection constants.  With `n_first` the poses are ordered [those for which !grad_pose_is_tilted ..., tilted
// ...] (stable), *n_first = size of the first group; GradC::slot keeps the caller's index.
static int upload_projc(tomo_ctx *ctx, const double *h_poses, int n, bool with_grad, ProjC **d_pc, GradC **d_gc,
                        const int32_t *h_rows = nullptr, int *n_first = nullptr)
{
    const size_t pc_bytes = sizeof(ProjC) * (size_t)n;
    const size_t gc_off = (pc_bytes + 255) & ~(size_t)255;
    const size_t total = gc_off + (with_grad ? sizeof(GradC) * (size_t)n : 0);
    // the staging buffer may still be in use by an earlier async launch: drain before rewriting it
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc = tomo_ensure_stage(ctx, total);
    if (rc) return rc;
    ProjC *hp = (ProjC *)ctx->h_stage;
    GradC *hg = (GradC *)((char *)ctx->h_stage + gc_off);
    int lo = 0, hi = n;                                  // next free entry of the first group / one past the last free of the second
    for (int pass = 0; pass < (n_first ? 2 : 1); ++pass) {
        for (int i = (pass == 0 ? 0 : n - 1); pass == 0 ? i < n : i >= 0; i += (pass == 0 ? 1 : -1)) {
            ProjC pc;
            GradC gc;
            tomo_make_projc(ctx->g, h_poses + (size_t)i * TOMO_POSE_STRIDE, pc, with_grad ? &gc : nullptr);
            int at = i;
            if (n_first) {
                const bool tilted = grad_pose_is_tilted(pc);
                if (pass == 0) { if (tilted) continue; at = lo++; }        // first group in ascending order
                else { if (!tilted) continue; at = --hi; }                 // second group filled from the back, walking backwards
            }
            hp[at] = pc;
            if (with_grad) {
                gc.b_row = h_rows ? h_rows[i] : i;
                gc.slot = i;
                hg[at] = gc;
            }
        }
    }
    if (n_first) *n_first = lo;
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_stage, ctx->h_stage, total, hipMemcpyHostToDevice, ctx->stream));
    *d_pc = (ProjC *)ctx->d_stage;
    if (d_gc) *d_gc = (GradC *)((char *)ctx->d_stage + gc_off);
    return TOMO_OK;
}

static inline dim3 ray_grid(const TomoGeomC &g, int n_proj) { return dim3((g.ndz + 63) / 64, (g.ndx + 3) / 4, n_proj); }
// k_proj_grad_v2 picks its block order from bit 4 of row_order: detector-z chunk slowest when the padded volume is larger than
// the Infinity Cache can keep (about 192 MB to leave room for the rest) and several projections share it
static inline bool grad_zslow(const TomoGeomC &g, int n_proj) { return n_proj > 1 && (size_t)g.nxp * g.nyp * g.nzp * 4 > ((size_t)192 << 20); }
static inline dim3 grad_grid(const TomoGeomC &g, int n_proj)
{
    return grad_zslow(g, n_proj) ? dim3((g.ndx + 3) / 4, n_proj, (g.ndz + 63) / 64) : ray_grid(g, n_proj);
}

#define TOMO_MAX_GRID_Z 65535

static bool invert3(const double m[3][3], double inv[3][3], double *det_out)
{
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    *det_out = det;
    if (!(fabs(det) > 1e-12)) return false;
    const double id = 1.0 / det;
    inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id;
    inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
    inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id;
    inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    return true;
}

// Per-projection constants of the tile kernels, staged to the device as [flat projections..., general projections...].
// all_ok = false (nothing staged) when some projection's detector-z axis does not map mostly onto volume z (tilt beyond
// ~45 deg) or its lattice is singular / out of fixed-point range: those calls take the ray-driven / atomic kernels.
// Stage the tile kernels' constants in the order [gather-eligible flat ..., other flat ..., general ...]; the first *n_gather
// also get a GfC record (for k_adj_gather_flat) in a second array behind the AdjC array (*d_gfc).
static int stage_tile_consts(tomo_ctx *ctx, const double *h_poses, int n_proj, bool *all_ok, double *weight_bound, int *n_flat,
                             int *n_gather = nullptr, const GfC **d_gfc = nullptr)
{
    const TomoGeomC &g = ctx->g;
    *all_ok = false;
    *weight_bound = 2.0;
    *n_flat = 0;
    if (n_gather) *n_gather = 0;
    const int opts = (ctx->tile_flat != 0 ? 1 : 0) | (ctx->adj_flat_gather != 0 ? 2 : 0);
    const size_t n_pose_doubles = (size_t)n_proj * TOMO_POSE_STRIDE;
    if (ctx->tile_cache_valid && ctx->tile_cache_opts == opts && ctx->tile_cache_poses.size() == n_pose_doubles && n_proj > 0 &&
        memcmp(ctx->tile_cache_poses.data(), h_poses, n_pose_doubles * sizeof(double)) == 0) {
        *all_ok = ctx->tile_cache_ok;                    // same poses, staging buffers untouched since: nothing to do
        *weight_bound = ctx->tile_cache_wb;
        *n_flat = ctx->tile_cache_nflat;
        if (n_gather) *n_gather = ctx->tile_cache_ngather;
        if (d_gfc) *d_gfc = (const GfC *)((char *)ctx->d_stage + ctx->tile_cache_gfoff);
        return TOMO_OK;
    }
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t gf_off = (sizeof(AdjC) * (size_t)std::max(n_proj, 1) + 255) & ~(size_t)255;
    int rc = tomo_ensure_stage(ctx, gf_off + sizeof(GfC) * (size_t)std::max(n_proj, 1));
    if (rc) return rc;
    std::vector<AdjC> gath, flat, gen;
    std::vector<GfC> gfc;
    double eb_max = 0.0;
    for (int i = 0; i < n_proj; ++i) {
        ProjC pc;
        tomo_make_projc(g, h_poses + (size_t)i * TOMO_POSE_STRIDE, pc, nullptr);
        AdjC a;
        double m[3][3], det = 0.0;
        bool fits = true;
        for (int r = 0; r < 3; ++r) {
            a.p0[r] = pc.p0[r]; a.u[r] = pc.u[r]; a.w[r] = pc.w[r]; a.d[r] = pc.d[r];
            m[r][0] = pc.u[r]; m[r][1] = pc.w[r]; m[r][2] = pc.d[r];
            const double lim = 1073741824.0;     // |coordinate| < 2^30 voxels
            fits = fits && fabs(pc.p0[r]) < lim && fabs(pc.u[r]) * g.ndx < lim && fabs(pc.w[r]) * g.ndz < lim && fabs(pc.d[r]) * pc.n < lim;
            a.fp0[r] = llround(pc.p0[r] * 4294967296.0);
            a.fu[r] = llround(pc.u[r] * 4294967296.0);
            a.fw[r] = llround(pc.w[r] * 4294967296.0);
            a.fd[r] = llround(pc.d[r] * 4294967296.0);
        }
        a.n = pc.n;
        a.slot = i;
        if (!fits) return TOMO_OK;
        if (!invert3(m, a.minv, &det)) return TOMO_OK;
        if (!(pc.w[2] > 0.7 * sqrt(pc.w[0] * pc.w[0] + pc.w[1] * pc.w[1] + pc.w[2] * pc.w[2]))) return TOMO_OK;
        // samples per unit volume = 1/|det[u w d]|; the tent weights a voxel collects from one projection sum to about
        // that density (exactly 1 for an axis-aligned unit lattice); x2 head-room for the fixed-point image
        *weight_bound = std::max(*weight_bound, 2.0 / fabs(det));
        const bool untilted = a.fw[0] == 0 && a.fw[1] == 0 && a.fw[2] == ((int64_t)1 << 32) && a.fu[2] == 0 && a.fd[2] == 0 &&
                              ctx->tile_flat != 0;
        // gather-form adjoint: additionally the x-y lattice must be (close to) a unit lattice, so that three consecutive rows
        // and three consecutive samples cover a voxel's footprint (2 * (|m_r0| + |m_r1| + 5e-3) < 3) -- true for detector
        // pitch = step = voxel at any phi (sum <= sqrt 2)
        const double ea = fabs(a.minv[0][0]) + fabs(a.minv[0][1]), eb = fabs(a.minv[2][0]) + fabs(a.minv[2][1]);
        const bool gatherable = untilted && ctx->adj_flat_gather != 0 && ea < 1.45 && eb < 2.99 &&      // ea: see GROWS; eb: NJ <= 6
                                fabs(a.minv[0][2]) < 1e-9 && fabs(a.minv[2][2]) < 1e-9;
        if (gatherable) {
            GfC q;
            q.fp0x = a.fp0[0]; q.fp0y = a.fp0[1]; q.fux = a.fu[0]; q.fuy = a.fu[1]; q.fdx = a.fd[0]; q.fdy = a.fd[1];
            q.m00 = (float)a.minv[0][0]; q.m01 = (float)a.minv[0][1]; q.m10 = (float)a.minv[2][0]; q.m11 = (float)a.minv[2][1];
            q.p0x = (float)a.p0[0]; q.p0y = (float)a.p0[1];
            q.zc = (int32_t)(a.fp0[2] >> 32);                                       // floor of the constant z offset
            q.tau = (float)((double)(uint32_t)(a.fp0[2] & 0xffffffffll) / 4294967296.0);
            q.n = a.n; q.slot = a.slot;
            gath.push_back(a);
            gfc.push_back(q);
            eb_max = std::max(eb_max, eb);
        } else
            (untilted ? flat : gen).push_back(a);
    }
    AdjC *h = (AdjC *)ctx->h_stage;
    size_t k = 0;
    for (const AdjC &a : gath) h[k++] = a;
    for (const AdjC &a : flat) h[k++] = a;
    for (const AdjC &a : gen) h[k++] = a;
    GfC *hg = (GfC *)((char *)ctx->h_stage + gf_off);
    for (size_t i = 0; i < gfc.size(); ++i) hg[i] = gfc[i];
    *n_flat = (int)(gath.size() + flat.size());
    if (n_gather) *n_gather = (int)gath.size();
    if (d_gfc) *d_gfc = (const GfC *)((char *)ctx->d_stage + gf_off);
    if (n_proj) TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_stage, h, gf_off + sizeof(GfC) * gfc.size(), hipMemcpyHostToDevice, ctx->stream));
    *all_ok = true;
    ctx->tile_cache_poses.assign(h_poses, h_poses + n_pose_doubles);
    ctx->tile_cache_opts = opts;
    ctx->tile_cache_ok = true;
    ctx->tile_cache_wb = *weight_bound;
    ctx->tile_cache_nflat = *n_flat;
    ctx->tile_cache_ngather = (int)gath.size();
    ctx->tile_cache_gfoff = gf_off;
    ctx->tile_cache_eb_max = eb_max;
    ctx->tile_cache_valid = true;
    return TOMO_OK;
}

static inline dim3 tile_grid(const TomoGeomC &g, int tz = ATZ) { return dim3((g.nz + 1 + tz - 1) / tz, (g.ny + 1 + ATY - 1) / ATY, (g.nx + 1 + ATX - 1) / ATX); }

extern "C" int tomo_forward(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_vol, float *d_proj)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_vol || !d_proj || n_proj < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_forward: bad args");
    if (n_proj == 0) return TOMO_OK;
    const TomoGeomC &g = ctx->g;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    int rc;
    if (ctx->fwd_variant == 3) {
        const dim3 grid = tile_grid(g);
        bool ok = false;
        double wb = 2.0;
        int n_flat = 0;
        if (grid.y <= 65535 && grid.z <= 65535) {
            rc = stage_tile_consts(ctx, h_poses, n_proj, &ok, &wb, &n_flat);
            if (rc) return rc;
        }
        if (ok) {
            const AdjC *d_c = (const AdjC *)ctx->d_stage;
            TOMO_HIP(ctx, hipMemsetAsync(d_proj, 0, n_det * (size_t)n_proj * sizeof(float), ctx->stream));
            if (n_flat > 0) {
                const dim3 fg = tile_grid(g, FTZ);
                if (fg.x >= 2 && ctx->fwd_flat_ztiles >= 2)         // two z-adjacent tiles per work-group share the per-row set-up
                    TOMO_LAUNCH(ctx, "k_fwd_tile_flat", k_fwd_flat_z<2>, dim3((fg.x + 1) / 2, fg.y, fg.z), dim3(FZ_WAVES * 64), 0, d_c, n_flat,
                                d_proj, d_vol, g);
                else
                    TOMO_LAUNCH(ctx, "k_fwd_tile_flat", k_tile_flat<true>, fg, dim3(ADJ_WAVES * 64), 0, d_c, n_flat, d_proj,
                                (float *)d_vol, g, (const unsigned *)nullptr, 1.f, 0);
            }
            if (n_proj > n_flat)
                TOMO_LAUNCH(ctx, "k_fwd_tile", k_tile<true>, grid, dim3(ADJ_WAVES * 64), 0, d_c + n_flat, n_proj - n_flat, d_proj, (float *)d_vol,
                            g, (const unsigned *)nullptr, 1.f, 0);
            return TOMO_OK;
        }
    }
    rc = stage_volume(ctx, d_vol);
    if (rc) return rc;
    for (int p0 = 0; p0 < n_proj; p0 += TOMO_MAX_GRID_Z) {
        const int np = std::min(TOMO_MAX_GRID_Z, n_proj - p0);
        ProjC *d_pc = nullptr;
        rc = upload_projc(ctx, h_poses + (size_t)p0 * TOMO_POSE_STRIDE, np, false, &d_pc, nullptr);
        if (rc) return rc;
        float *out = d_proj + (size_t)p0 * n_det;
        if (ctx->fwd_variant == 1)
            TOMO_LAUNCH(ctx, "k_fwd_v1", k_fwd_v1, ray_grid(g, np), dim3(256), 0, d_pc, ctx->d_volpad, out, g);
        else
            TOMO_LAUNCH(ctx, "k_fwd_v2", k_fwd_v2, ray_grid(g, np), dim3(256), 0, d_pc, ctx->d_volpad, out, g);
    }
    return TOMO_OK;
}

static int adjoint_atomic(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol, int accumulate)
{
    const TomoGeomC &g = ctx->g;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_volpad, 0, ctx->volpad_elems * sizeof(float), ctx->stream));
    ctx->halo_dirty = true;   // the halo collects the out-of-bounds corners
    ctx->staged_src = nullptr;
    for (int p0 = 0; p0 < n_proj; p0 += TOMO_MAX_GRID_Z) {
        const int np = std::min(TOMO_MAX_GRID_Z, n_proj - p0);
        ProjC *d_pc = nullptr;
        int rc = upload_projc(ctx, h_poses + (size_t)p0 * TOMO_POSE_STRIDE, np, false, &d_pc, nullptr);
        if (rc) return rc;
        TOMO_LAUNCH(ctx, "k_adj_v1", k_adj_v1, ray_grid(g, np), dim3(256), 0, d_pc, d_proj + (size_t)p0 * n_det, ctx->d_volpad, g);
    }
    TOMO_LAUNCH(ctx, "k_unpad", k_unpad, dim3(g.nx * g.ny), dim3(256), 0, d_vol, ctx->d_volpad, g, accumulate);
    return TOMO_OK;
}

// tile adjoint restricted to the x tile columns [xt0, xt1): adds into d_vol (the caller zeroes it).  *done = false when the
// poses do not qualify for the tile kernels (nothing launched).
static int adjoint_tiles(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol, int xt0, int xt1, bool *done)
{
    const TomoGeomC &g = ctx->g;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    dim3 grid = tile_grid(g);
    *done = false;
    bool ok = false;
    double weight_bound = 2.0;
    int n_flat = 0;
    if (ctx->adj_variant == 1 || n_proj <= 0 || grid.y > 65535 || grid.z > 65535) return TOMO_OK;
    int n_gather = 0;
    const GfC *d_gfc = nullptr;
    int rc = stage_tile_consts(ctx, h_poses, n_proj, &ok, &weight_bound, &n_flat, &n_gather, &d_gfc);
    if (rc) return rc;
    if (!ok) return TOMO_OK;
    const int n_xt = (int)grid.z;
    xt0 = std::max(xt0, 0);
    xt1 = std::min(xt1, n_xt);
    *done = true;
    if (xt1 <= xt0) return TOMO_OK;
    grid.z = (unsigned)(xt1 - xt0);
    const AdjC *d_c = (const AdjC *)ctx->d_stage;
    if (n_gather > 0) {
        // the voxel x range the tile columns [xt0, xt1) finalise (the tile grid starts at x = -1)
        const int xs = std::max(0, ATX * xt0 - 1), xe = (xt1 == n_xt) ? g.nx : std::min(g.nx, ATX * xt1 - 1);
        if (xe > xs) {
            const dim3 ggrid((g.nz + 64 * GWAVES - 1) / (64 * GWAVES), (g.ny + GTY - 1) / GTY, (xe - xs + GTX - 1) / GTX);
            if (ctx->tile_cache_eb_max < 1.49)
                TOMO_LAUNCH(ctx, "k_adj_gather_flat", k_adj_gather_flat<3>, ggrid, dim3(GWAVES * 64), 0, d_gfc, n_gather, d_proj, d_vol, g, xs, xe);
            else        // finer sampling along the rays (step down to ~0.475 voxel): six samples per row can reach a column
                TOMO_LAUNCH(ctx, "k_adj_gather_flat", k_adj_gather_flat<6>, ggrid, dim3(GWAVES * 64), 0, d_gfc, n_gather, d_proj, d_vol, g, xs, xe);
        }
    }
    if (n_proj == n_gather) return TOMO_OK;
    rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    unsigned *d_absmax = (unsigned *)(ctx->d_red + 4);
    TOMO_HIP(ctx, hipMemsetAsync(d_absmax, 0, sizeof(unsigned), ctx->stream));
    const int64_t n_y = (int64_t)n_det * n_proj;
    TOMO_LAUNCH(ctx, "k_absmax", k_absmax, dim3((unsigned)std::min<int64_t>((n_y + 255) / 256, 2048)), dim3(256), 0, d_proj, n_y, d_absmax);
    if (n_flat > n_gather) {
        dim3 fgrid = tile_grid(g, FTZ);
        fgrid.z = grid.z;
        TOMO_LAUNCH(ctx, "k_adj_tile_flat", k_tile_flat<false>, fgrid, dim3(ADJ_WAVES * 64), 0, d_c + n_gather, n_flat - n_gather, (float *)d_proj,
                    d_vol, g, (const unsigned *)d_absmax, (float)weight_bound, xt0);
    }
    if (n_proj > n_flat)
        TOMO_LAUNCH(ctx, "k_adj_tile", k_tile<false>, grid, dim3(ADJ_WAVES * 64), 0, d_c + n_flat, n_proj - n_flat, (float *)d_proj, d_vol, g,
                    (const unsigned *)d_absmax, (float)weight_bound, xt0);
    return TOMO_OK;
}

extern "C" int tomo_adjoint(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol, int accumulate)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_vol || !d_proj || n_proj < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_adjoint: bad args");
    const TomoGeomC &g = ctx->g;
    const size_t n_vox = (size_t)g.nx * g.ny * g.nz;
    if (n_proj == 0) {
        if (!accumulate) TOMO_HIP(ctx, hipMemsetAsync(d_vol, 0, n_vox * sizeof(float), ctx->stream));
        return TOMO_OK;
    }
    // the tile kernels add into d_vol: zero it first; if the poses turn out not to qualify, the atomic path overwrites anyway
    if (!accumulate && ctx->adj_variant != 1) TOMO_HIP(ctx, hipMemsetAsync(d_vol, 0, n_vox * sizeof(float), ctx->stream));
    bool done = false;
    int rc = adjoint_tiles(ctx, h_poses, n_proj, d_proj, d_vol, 0, INT_MAX, &done);
    if (rc) return rc;
    if (done) return TOMO_OK;
    return adjoint_atomic(ctx, h_poses, n_proj, d_proj, d_vol, accumulate);
}

// x-slab form for pipelining the all-reduce with the back-projection (volume is x-major: the tile columns [xt0, xt1) finalise
// the contiguous voxel range x in [ATX*xt0 - 1, ATX*xt1 - 1) clipped to [0, nx); the last column also finalises up to nx).
extern "C" int tomo_adjoint_xslab_info(tomo_ctx *ctx, int *n_xtiles, int *tile_width)
{
    TOMO_NEED_GEOM(ctx);
    if (!n_xtiles || !tile_width) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_adjoint_xslab_info: bad args");
    *n_xtiles = (int)tile_grid(ctx->g).z;
    *tile_width = ATX;
    return TOMO_OK;
}

extern "C" int tomo_adjoint_xslab(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol, int xt0, int xt1)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_vol || !d_proj || n_proj < 0 || xt0 < 0 || xt1 < xt0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_adjoint_xslab: bad args");
    bool done = false;
    int rc = adjoint_tiles(ctx, h_poses, n_proj, d_proj, d_vol, xt0, xt1, &done);
    if (rc) return rc;
    if (!done) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_adjoint_xslab: these poses do not take the tile kernels");
    return TOMO_OK;
}

extern "C" int tomo_backproject_voxel(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_det, float *d_vol)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_det || !d_vol || n_proj < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_backproject_voxel: bad args");
    const TomoGeomC &g = ctx->g;
    if (g.nx > TOMO_MAX_GRID_Z) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_backproject_voxel: nx > 65535");
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc = tomo_ensure_stage(ctx, sizeof(BpC) * (size_t)std::max(n_proj, 1));
    if (rc) return rc;
    BpC *h = (BpC *)ctx->h_stage;
    for (int i = 0; i < n_proj; ++i) {
        const double *p = h_poses + (size_t)i * TOMO_POSE_STRIDE;
        // x' = Ry(beta) (Rx(alpha) Rz(phi) c + t)            src/external_back_projection.f90:20-25
        TomoM3 Rz = tomo_rz(p[0]), Rx = tomo_rx(p[1]), Ry = tomo_ry(p[2]);
        TomoM3 B = tomo_mm(Ry, tomo_mm(Rx, Rz));
        const double t[3] = {p[3], p[4], p[5]};
        double rt[3], bo[3];
        tomo_mv(Ry, t, rt);
        tomo_mv(B, g.org, bo);
        h[i].u0 = bo[0] + rt[0] - g.org[0];                      // :45 (u = x'_1 - origin_1)
        h[i].ux = B.m[0][0] * ctx->vox_pitch[0]; h[i].uy = B.m[0][1] * ctx->vox_pitch[1]; h[i].uz = B.m[0][2] * ctx->vox_pitch[2];
        h[i].v0 = bo[2] + rt[2] - g.org[2];                      // :46 (v = x'_3 - origin_3)
        h[i].vx = B.m[2][0] * ctx->vox_pitch[0]; h[i].vy = B.m[2][1] * ctx->vox_pitch[1]; h[i].vz = B.m[2][2] * ctx->vox_pitch[2];
    }
    if (n_proj) TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_stage, h, sizeof(BpC) * (size_t)n_proj, hipMemcpyHostToDevice, ctx->stream));
    TOMO_LAUNCH(ctx, "k_bp_voxel", k_bp_voxel, dim3((g.nz + 63) / 64, (g.ny + 3) / 4, g.nx), dim3(256), 0, (const BpC *)ctx->d_stage,
                n_proj, d_det, d_vol, g);
    return TOMO_OK;
}

extern "C" int tomo_proj_grad(tomo_ctx *ctx, const double *h_pose, const float *d_vol, float *d_proj, float *d_grad, int row_order)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_pose || !d_vol || !d_proj || !d_grad || (row_order != 0 && row_order != 1))
        return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_proj_grad: bad args");
    const TomoGeomC &g = ctx->g;
    int rc = stage_volume(ctx, d_vol);
    if (rc) return rc;
    ProjC *d_pc = nullptr;
    GradC *d_gc = nullptr;
    int n_plain = 0;
    rc = upload_projc(ctx, h_pose, 1, true, &d_pc, &d_gc, nullptr, &n_plain);
    if (rc) return rc;
    const int variant = ctx->grad_variant == 4 ? (n_plain ? 2 : 3) : ctx->grad_variant;       // 4 = by tilt
    if (variant == 1)
        TOMO_LAUNCH(ctx, "k_proj_grad", k_proj_grad<false>, ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
                    (const float *)nullptr, (float *)nullptr, (double *)nullptr, g, row_order);
    else if (variant == 2)
        TOMO_LAUNCH(ctx, "k_proj_grad", k_proj_grad_v2<false>, ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
                    (const float *)nullptr, (float *)nullptr, (double *)nullptr, g, row_order);
    else
        TOMO_LAUNCH(ctx, "k_proj_grad", k_proj_grad_v3<false>, ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
                    (const float *)nullptr, (float *)nullptr, (double *)nullptr, g, row_order);
    return TOMO_OK;
}

extern "C" int tomo_cost_grad(tomo_ctx *ctx, const double *h_poses, int n, const float *d_vol, const float *d_b, double *h_cost,
                              double *h_grad6, float *d_resid)
{
    return tomo_cost_grad_rows(ctx, h_poses, n, d_vol, d_b, nullptr, n, h_cost, h_grad6, d_resid);
}

extern "C" int tomo_cost_grad_rows(tomo_ctx *ctx, const double *h_poses, int n, const float *d_vol, const float *d_b,
                                   const int32_t *h_rows, int n_rows_total, double *h_cost, double *h_grad6, float *d_resid)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_vol || !d_b || !h_cost || !h_grad6 || n < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_cost_grad: bad args");
    if (h_rows)
        for (int i = 0; i < n; ++i)
            if (h_rows[i] < 0 || h_rows[i] >= n_rows_total) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_cost_grad_rows: row index outside the table");
    if (n == 0) return TOMO_OK;
    if (n > TOMO_MAX_GRID_Z) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_cost_grad: more than 65535 projections per call");
    const TomoGeomC &g = ctx->g;
    int rc = stage_volume(ctx, d_vol);
    if (rc) return rc;
    rc = tomo_ensure_red(ctx, (size_t)n * 7 + 8);
    if (rc) return rc;
    ProjC *d_pc = nullptr;
    GradC *d_gc = nullptr;
    int n_plain = 0;                                      // staged order: [near-untilted poses ..., tilted poses ...]
    rc = upload_projc(ctx, h_poses, n, true, &d_pc, &d_gc, h_rows, &n_plain);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, sizeof(double) * (size_t)n * 7, ctx->stream));
    // up to two launches: the first group with one kernel variant, the second with another (grad_variant 4: v2 / v3 by tilt)
    const int var_a = ctx->grad_variant == 4 ? 2 : ctx->grad_variant, var_b = ctx->grad_variant == 4 ? 3 : ctx->grad_variant;
    for (int part = 0; part < 2; ++part) {
        const int first = part == 0 ? 0 : n_plain, cnt = part == 0 ? n_plain : n - n_plain, variant = part == 0 ? var_a : var_b;
        if (cnt == 0) continue;
        const ProjC *pc = d_pc + first;
        const GradC *gc = d_gc + first;
        if (variant == 1)
            TOMO_LAUNCH(ctx, "k_cost_grad", k_proj_grad<true>, ray_grid(g, cnt), dim3(256), 0, pc, gc, ctx->d_volpad, (float *)nullptr,
                        (float *)nullptr, d_b, d_resid, ctx->d_red, g, 0);
        else if (variant == 2)
            TOMO_LAUNCH(ctx, "k_cost_grad", k_proj_grad_v2<true>, grad_grid(g, cnt), dim3(256), 0, pc, gc, ctx->d_volpad, (float *)nullptr,
                        (float *)nullptr, d_b, d_resid, ctx->d_red, g, grad_zslow(g, cnt) ? 16 : 0);
        else
            TOMO_LAUNCH(ctx, "k_cost_grad", k_proj_grad_v3<true>, grad_grid(g, cnt), dim3(256), 0, pc, gc, ctx->d_volpad, (float *)nullptr,
                        (float *)nullptr, d_b, d_resid, ctx->d_red, g, grad_zslow(g, cnt) ? 16 : 0);
    }
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red, ctx->d_red, sizeof(double) * (size_t)n * 7, hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) {
        h_cost[i] = ctx->h_red[(size_t)i * 7];
        for (int k = 0; k < 6; ++k) h_grad6[(size_t)i * 6 + k] = ctx->h_red[(size_t)i * 7 + 1 + k];
    }
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// COO triplets of one projection (src/ray_wt_grad.f90:1-92 trilinear_ray_sparse): count -> scan -> fill, float64
// weights, emission order ray-major / sample / corner (x slowest, z fastest, floor before ceil).  Small volumes only
// (8 slots per sample): this is the "materialise a real scipy CSR" path, not a hot path.
// ------------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void k_triplets(const ProjC *__restrict__ pcs, TomoGeomC g, const int64_t *__restrict__ offsets,
                                                  int32_t *__restrict__ counts, int32_t *__restrict__ dat, int32_t *__restrict__ det,
                                                  double *__restrict__ wts)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_det = g.ndx * g.ndz;
    if (r >= n_det) return;
    const int ix = r / g.ndz, iz = r - ix * g.ndz;
    const ProjC &c = pcs[0];
    double b[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) b[a] = c.p0[a] + (double)ix * c.u[a] + (double)iz * c.w[a];
    int j0, j1;
    tomo_ray_range(b, c.d, c.n, g.nx, g.ny, g.nz, j0, j1);
    int64_t o = FILL ? offsets[r] : 0;
    int cnt = 0;
    for (int j = j0; j < j1; ++j) {
        double p[3], f[3], wf[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            p[a] = b[a] + (double)j * c.d[a];                 // utilities/ray_voxel_utilities.py:93
            f[a] = floor(p[a]);                               // :96
            wf[a] = 1.0 - (p[a] - f[a]);                      // :98-99
        }
        const int fx = (int)f[0], fy = (int)f[1], fz = (int)f[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int x = fx + (k >> 2), y = fy + ((k >> 1) & 1), z = fz + (k & 1);
            if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {      // src/ray_wt_grad.f90:35-89
                if (FILL) {
                    const double wx = (k >> 2) ? 1.0 - wf[0] : wf[0], wy = ((k >> 1) & 1) ? 1.0 - wf[1] : wf[1], wz = (k & 1) ? 1.0 - wf[2] : wf[2];
                    det[o] = r;
                    dat[o] = (x * g.ny + y) * g.nz + z;
                    wts[o] = wx * wy * wz;
                    ++o;
                }
                ++cnt;
            }
        }
    }
    if (!FILL) counts[r] = cnt;
}

extern "C" int tomo_triplets(tomo_ctx *ctx, const double *h_pose, int64_t capacity, int32_t *h_dat, int32_t *h_det, double *h_wts,
                             int64_t *n_inds)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_pose || !n_inds) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_triplets: bad args");
    const TomoGeomC &g = ctx->g;
    const int n_det = g.ndx * g.ndz;
    if ((size_t)g.nx * g.ny * g.nz >= ((size_t)1 << 31)) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_triplets: int32 voxel indices");
    ProjC *d_pc = nullptr;
    int rc = upload_projc(ctx, h_pose, 1, false, &d_pc, nullptr);
    if (rc) return rc;
    int32_t *d_counts = nullptr;
    int64_t *d_off = nullptr;
    TOMO_HIP(ctx, hipMalloc((void **)&d_counts, sizeof(int32_t) * (size_t)n_det));
    TOMO_HIP(ctx, hipMalloc((void **)&d_off, sizeof(int64_t) * (size_t)n_det));
    const dim3 grid((n_det + 255) / 256);
    std::vector<int32_t> counts(n_det);
    std::vector<int64_t> off(n_det);
    int result = TOMO_OK;
    int32_t *d_dat = nullptr, *d_det = nullptr;
    double *d_w = nullptr;
    do {
        hipLaunchKernelGGL(k_triplets<false>, grid, dim3(256), 0, ctx->stream, (const ProjC *)d_pc, g, (const int64_t *)nullptr, d_counts,
                           (int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr);
        if (hipMemcpyAsync(counts.data(), d_counts, sizeof(int32_t) * (size_t)n_det, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { result = tomo_fail(ctx, TOMO_ERR_HIP, "tomo_triplets: count pass failed"); break; }
        int64_t total = 0;
        for (int r = 0; r < n_det; ++r) { off[r] = total; total += counts[r]; }
        *n_inds = total;
        if (!h_dat || !h_det || !h_wts) break;                  // count query
        if (capacity < total) { result = tomo_fail(ctx, TOMO_ERR_ARG, "tomo_triplets: capacity too small"); break; }
        if (total == 0) break;
        if (hipMalloc((void **)&d_dat, sizeof(int32_t) * (size_t)total) != hipSuccess || hipMalloc((void **)&d_det, sizeof(int32_t) * (size_t)total) != hipSuccess ||
            hipMalloc((void **)&d_w, sizeof(double) * (size_t)total) != hipSuccess) { result = tomo_fail(ctx, TOMO_ERR_HIP, "tomo_triplets: out of device memory"); break; }
        (void)hipMemcpyAsync(d_off, off.data(), sizeof(int64_t) * (size_t)n_det, hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL(k_triplets<true>, grid, dim3(256), 0, ctx->stream, (const ProjC *)d_pc, g, (const int64_t *)d_off, (int32_t *)nullptr, d_dat,
                           d_det, d_w);
        (void)hipMemcpyAsync(h_dat, d_dat, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(h_det, d_det, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(h_wts, d_w, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) result = tomo_fail(ctx, TOMO_ERR_HIP, "tomo_triplets: fill pass failed");
    } while (0);
    (void)hipStreamSynchronize(ctx->stream);
    if (d_dat) (void)hipFree(d_dat);
    if (d_det) (void)hipFree(d_det);
    if (d_w) (void)hipFree(d_w);
    (void)hipFree(d_counts);
    (void)hipFree(d_off);
    return result;
}

// ------------------------------------------------------------------------------------------------
// voxel-driven bilinear splat (src/vox_wt_grad.f90, utilities/voxel_utilities.py): one voxel per work-item, lanes
// along z; 4 (+24 with the Jacobian) global float atomics per voxel.  Dead code in the reference
// (utilities/projection_operators.py:54 hard-wires the ray path), built for completeness of SURVEY 8a row A9.
// ------------------------------------------------------------------------------------------------
struct VoxC {
    double m[3][3];     // Ry Rx Rz
    double off[3];      // Ry t
    double org[3];      // vox_origin - cor_shift
    float rb[3][3];     // Ry           (der rows 0-2 are its columns)
    double a3[3][3];    // Ry Rx dRz    (der row 3 = a3 c)
    double a4[3][3];    // Ry dRx Rz    (der row 4 = a4 c)
    double drb[3][3];   // dRy          (der row 5 = dRy (Rx Rz c + t))
    double rxz[3][3];   // Rx Rz
    double t[3];
};

template <int MODE>   // 0 splat value, 1 splat value + gradient, 2 triplets
__global__ __launch_bounds__(256) void k_vox_splat(VoxC c, TomoGeomC g, double px, double py, double pz, const float *__restrict__ vol,
                                                   float *__restrict__ img, float *__restrict__ grad, int32_t *__restrict__ det4,
                                                   float *__restrict__ wts4)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int iz = blockIdx.x * 64 + lane, iy = blockIdx.y * 4 + wv, ix = blockIdx.z;
    if (iy >= g.ny || iz >= g.nz) return;
    const size_t i = ((size_t)ix * g.ny + iy) * g.nz + iz;
    const double cx = g.org[0] + ix * px, cy = g.org[1] + iy * py, cz = g.org[2] + iz * pz;      // geometry.py:82-86
    const double rx = c.m[0][0] * cx + c.m[0][1] * cy + c.m[0][2] * cz + c.off[0];
    const double rz = c.m[2][0] * cx + c.m[2][1] * cy + c.m[2][2] * cz + c.off[2];
    const double u = rx - c.org[0], v = rz - c.org[2];
    const double fu = floor(u), fv = floor(v);                                                   // voxel_utilities.py:64-65
    const float ax = (float)(u - fu), az = (float)(v - fv);                                      // :66-67
    const bool sane = fabs(fu) < 2e9 && fabs(fv) < 2e9;
    const int fx = sane ? (int)fu : -10, fz = sane ? (int)fv : -10;
    const int ndx = g.ndx, ndz = g.ndz;
    const size_t n_pix = (size_t)ndx * ndz;
    float rec = 0.f, der0[6], der2[6];
    if (MODE != 2) rec = vol[i];
    if (MODE == 1) {
        // der rows (voxel_utilities.py:38-46), only columns 0 and 2 (x', z') are used by the splat
        double q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = c.rxz[a][0] * cx + c.rxz[a][1] * cy + c.rxz[a][2] * cz + c.t[a];
#pragma unroll
        for (int k = 0; k < 3; ++k) { der0[k] = c.rb[0][k]; der2[k] = c.rb[2][k]; }
        der0[3] = (float)(c.a3[0][0] * cx + c.a3[0][1] * cy + c.a3[0][2] * cz);
        der2[3] = (float)(c.a3[2][0] * cx + c.a3[2][1] * cy + c.a3[2][2] * cz);
        der0[4] = (float)(c.a4[0][0] * cx + c.a4[0][1] * cy + c.a4[0][2] * cz);
        der2[4] = (float)(c.a4[2][0] * cx + c.a4[2][1] * cy + c.a4[2][2] * cz);
        der0[5] = (float)(c.drb[0][0] * q[0] + c.drb[0][1] * q[1] + c.drb[0][2] * q[2]);
        der2[5] = (float)(c.drb[2][0] * q[0] + c.drb[2][1] * q[1] + c.drb[2][2] * q[2]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {          // emission order (fx,fz),(fx+1,fz),(fx,fz+1),(fx+1,fz+1): src/vox_wt_grad.f90:80-106
        const int a = k & 1, b = k >> 1;
        const int x = fx + a, z = fz + b;
        const bool ok = x >= 0 && x < ndx && z >= 0 && z < ndz;
        const float wx = a ? ax : 1.f - ax, wz = b ? az : 1.f - az;
        if (MODE == 2) {
            det4[4 * i + k] = ok ? x + ndx * z : -1;
            wts4[4 * i + k] = wx * wz;
        } else if (ok) {
            const size_t o = (size_t)z * ndx + x;
            atomicAdd(&img[o], rec * wx * wz);
            if (MODE == 1) {
                // src/vox_wt_grad.f90:27-28,33-34,39-40,45-46 (signs as written there)
                const float f0 = (a ? -1.f : 1.f) * (b ? az : 1.f - az);
                const float f2 = (b ? -1.f : 1.f) * (a ? ax : 1.f - ax);
#pragma unroll
                for (int r = 0; r < 6; ++r) atomicAdd(&grad[(size_t)r * n_pix + o], der0[r] * f0 * rec + der2[r] * f2 * rec);
            }
        }
    }
}

static void make_voxc(const tomo_ctx *ctx, const double *pose, const double *cor3, VoxC &c)
{
    const TomoGeomC &g = ctx->g;
    TomoM3 Rz = tomo_rz(pose[0]), Rx = tomo_rx(pose[1]), Ry = tomo_ry(pose[2]);
    TomoM3 dRz = tomo_drz(pose[0]), dRx = tomo_drx(pose[1]), dRy = tomo_dry(pose[2]);
    TomoM3 Rxz = tomo_mm(Rx, Rz), M = tomo_mm(Ry, Rxz);
    TomoM3 A3 = tomo_mm(tomo_mm(Ry, Rx), dRz), A4 = tomo_mm(Ry, tomo_mm(dRx, Rz));
    const double t[3] = {pose[3], pose[4], pose[5]};
    tomo_mv(Ry, t, c.off);
    for (int i = 0; i < 3; ++i) {
        c.org[i] = g.org[i] - (cor3 ? cor3[i] : 0.0);
        c.t[i] = t[i];
        for (int j = 0; j < 3; ++j) {
            c.m[i][j] = M.m[i][j]; c.rb[i][j] = (float)Ry.m[i][j]; c.a3[i][j] = A3.m[i][j]; c.a4[i][j] = A4.m[i][j];
            c.drb[i][j] = dRy.m[i][j]; c.rxz[i][j] = Rxz.m[i][j];
        }
    }
}

extern "C" int tomo_vox_splat(tomo_ctx *ctx, const double *h_pose, const double *h_cor3, const float *d_vol, float *d_img, float *d_grad)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_pose || !d_vol || !d_img) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_vox_splat: bad args");
    const TomoGeomC &g = ctx->g;
    if (g.nx > TOMO_MAX_GRID_Z) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_vox_splat: nx > 65535");
    VoxC c;
    make_voxc(ctx, h_pose, h_cor3, c);
    const size_t n_pix = (size_t)g.ndx * g.ndz;
    TOMO_HIP(ctx, hipMemsetAsync(d_img, 0, n_pix * sizeof(float), ctx->stream));
    const dim3 grid((g.nz + 63) / 64, (g.ny + 3) / 4, g.nx);
    if (d_grad) {
        TOMO_HIP(ctx, hipMemsetAsync(d_grad, 0, 6 * n_pix * sizeof(float), ctx->stream));
        TOMO_LAUNCH(ctx, "k_vox_splat", k_vox_splat<1>, grid, dim3(256), 0, c, g, ctx->vox_pitch[0], ctx->vox_pitch[1], ctx->vox_pitch[2], d_vol, d_img,
                    d_grad, (int32_t *)nullptr, (float *)nullptr);
    } else {
        TOMO_LAUNCH(ctx, "k_vox_splat", k_vox_splat<0>, grid, dim3(256), 0, c, g, ctx->vox_pitch[0], ctx->vox_pitch[1], ctx->vox_pitch[2], d_vol, d_img,
                    (float *)nullptr, (int32_t *)nullptr, (float *)nullptr);
    }
    return TOMO_OK;
}

extern "C" int tomo_vox_triplets(tomo_ctx *ctx, const double *h_pose, const double *h_cor3, int32_t *h_det4, float *h_wts4)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_pose || !h_det4 || !h_wts4) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_vox_triplets: bad args");
    const TomoGeomC &g = ctx->g;
    if (g.nx > TOMO_MAX_GRID_Z) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_vox_triplets: nx > 65535");
    const size_t n4 = 4 * (size_t)g.nx * g.ny * g.nz;
    VoxC c;
    make_voxc(ctx, h_pose, h_cor3, c);
    int32_t *d_det = nullptr;
    float *d_w = nullptr;
    TOMO_HIP(ctx, hipMalloc((void **)&d_det, n4 * sizeof(int32_t)));
    if (hipMalloc((void **)&d_w, n4 * sizeof(float)) != hipSuccess) { (void)hipFree(d_det); return tomo_fail(ctx, TOMO_ERR_HIP, "tomo_vox_triplets: out of device memory"); }
    const dim3 grid((g.nz + 63) / 64, (g.ny + 3) / 4, g.nx);
    hipLaunchKernelGGL(k_vox_splat<2>, grid, dim3(256), 0, ctx->stream, c, g, ctx->vox_pitch[0], ctx->vox_pitch[1], ctx->vox_pitch[2], (const float *)nullptr,
                       (float *)nullptr, (float *)nullptr, d_det, d_w);
    hipError_t e = hipMemcpyAsync(h_det4, d_det, n4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_wts4, d_w, n4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_det);
    (void)hipFree(d_w);
    if (e != hipSuccess) return tomo_fail(ctx, TOMO_ERR_HIP, std::string("tomo_vox_triplets: ") + hipGetErrorString(e));
    return TOMO_OK;
}
