// kernels_ray.hip.h -- wave helpers, zero-halo staging, the ray-driven forward / adjoint kernels and the voxel-driven back-projector
// Part of the single translation unit tomo_project.hip (included there, in this order: kernels_ray, kernels_tile,
// kernels_grad); not compiled on its own.

// ------------------------------------------------------------------------------------------------
// wave helpers (64 lanes)
// ------------------------------------------------------------------------------------------------
// Wave-wide integer min / max, the result in every lane (wave-uniform): quad and row butterflies with DPP (full-rate VALU:
// quad_perm xor 1, xor 2, row_half_mirror, row_mirror make each row of 16 lanes uniform), then the four rows meet on the scalar
// unit.  No LDS traffic -- the __shfl_xor form costs six ds_bpermute round trips, and the gather adjoint takes two of these per
// projection and wave.  ALL 64 LANES MUST BE ACTIVE (v_readlane reads lanes 0 / 16 / 32 / 48 whatever EXEC says).
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));    // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));    // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
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
// The padded copy is followed by 8 ints: the box [lo, hi) of the volume's non-zero voxels (x, y, z lo; x, y, z hi; 2 spare),
// found while the data streams through: k_pad leaves each (x, y) row's z range of non-zero values in `rowz`, k_box reduces the
// rows (one work-group; no contended atomics -- on a dense volume a million rows would all want to update the same six words).
// The ray-driven kernels clip every ray to the box (tomo_ray_range_box).
#define TOMO_BOX_INTS 8
__global__ __launch_bounds__(256) void k_pad(const float *__restrict__ vol, float *__restrict__ vp, TomoGeomC g, int2 *__restrict__ rowz)
{
    const int row = blockIdx.x;            // ix*ny + iy
    const int ix = row / g.ny, iy = row - ix * g.ny;
    const float *src = vol + (size_t)row * g.nz;
    float *dst = vp + ((size_t)(ix + TOMO_HALO) * g.nyp + (iy + TOMO_HALO)) * g.nzp + TOMO_HALO;
    int zlo = INT_MAX, zhi = 0;
    for (int z = threadIdx.x; z < g.nz; z += blockDim.x) {
        const float v = src[z];
        dst[z] = v;
        if (v != 0.f) { zlo = min(zlo, z); zhi = max(zhi, z + 1); }       // NaN != 0: kept
    }
    __shared__ int sh[8];
    zlo = wave_min_i32(zlo);
    zhi = wave_max_i32(zhi);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = zlo; sh[4 + (threadIdx.x >> 6)] = zhi; }
    __syncthreads();
    if (threadIdx.x == 0) rowz[row] = make_int2(min(min(sh[0], sh[1]), min(sh[2], sh[3])), max(max(sh[4], sh[5]), max(sh[6], sh[7])));
}

__global__ __launch_bounds__(1024) void k_box(const int2 *__restrict__ rowz, TomoGeomC g, int *__restrict__ box)
{
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {0, 0, 0};
    const int n_rows = g.nx * g.ny;
    for (int row = threadIdx.x; row < n_rows; row += 1024) {
        const int2 z = rowz[row];
        if (z.y > 0) {
            const int ix = row / g.ny, iy = row - ix * g.ny;
            lo[0] = min(lo[0], ix); hi[0] = max(hi[0], ix + 1);
            lo[1] = min(lo[1], iy); hi[1] = max(hi[1], iy + 1);
            lo[2] = min(lo[2], z.x); hi[2] = max(hi[2], z.y);
        }
    }
    __shared__ int sh[16][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int l = wave_min_i32(lo[a]), h = wave_max_i32(hi[a]);
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][a] = l; sh[threadIdx.x >> 6][3 + a] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = sh[0][threadIdx.x];
        for (int w = 1; w < 16; ++w) v = threadIdx.x < 3 ? min(v, sh[w][threadIdx.x]) : max(v, sh[w][threadIdx.x]);
        box[threadIdx.x] = v;
    }
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
    int *box = (int *)(ctx->d_volpad + ctx->volpad_elems);
    int2 *rowz = (int2 *)(box + TOMO_BOX_INTS);
    TOMO_LAUNCH(ctx, "k_pad", k_pad, dim3(g.nx * g.ny), dim3(256), 0, d_vol, ctx->d_volpad, g, rowz);
    TOMO_LAUNCH(ctx, "k_box", k_box, dim3(1), dim3(1024), 0, (const int2 *)rowz, g, box);
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// per-ray set-up shared by the ray-driven kernels
// ------------------------------------------------------------------------------------------------
struct RayCtx {
    double b[3], d[3];
    int j0, j1;
};

// `box`: the 6 ints behind the staged volume (k_pad) for kernels that READ it -- rays are clipped to the non-zero voxels; nullptr
// for the scatter kernel, which must visit every sample inside the volume
__device__ __forceinline__ void ray_setup(const ProjC &c, const TomoGeomC &g, int ix, int iz, bool valid, RayCtx &r, const int *box = nullptr)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.b[a] = c.p0[a] + (double)ix * c.u[a] + (double)iz * c.w[a];
        r.d[a] = c.d[a];
    }
    if (box) {
        const int lo[3] = {box[0], box[1], box[2]}, hi[3] = {box[3], box[4], box[5]};      // wave-uniform: scalar loads
        tomo_ray_range_box(r.b, r.d, c.n, lo, hi, r.j0, r.j1);
    } else
        tomo_ray_range(r.b, r.d, c.n, g.nx, g.ny, g.nz, r.j0, r.j1);
    if (!valid) r.j0 = r.j1 = 0;
}
__device__ __forceinline__ const int *staged_box(const float *vp, const TomoGeomC &g) { return (const int *)(vp + (size_t)g.nxp * g.nyp * g.nzp); }

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
    ray_setup(c, g, ix, iz, true, r, staged_box(vp, g));
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
    ray_setup(c, g, ix, iz, valid, r, staged_box(vp, g));
    const bool nonempty = r.j1 > r.j0;
    const int J0 = __builtin_amdgcn_readfirstlane(wave_min_i32(nonempty ? r.j0 : INT_MAX));
    const int J1 = __builtin_amdgcn_readfirstlane(wave_max_i32(nonempty ? r.j1 : 0));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const uint32_t sy4 = (uint32_t)g.nzp * 4u, sx4 = (uint32_t)g.nyp * (uint32_t)g.nzp * 4u;     // < 2^23 (tomo_check_geometry): signed 24-bit multiplies
    const uint32_t abias4 = tomo_abias_bytes(sx4, sy4) + tomo_lbias_bytes(sx4, sy4);
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
        const int delta = (int)(lin - lin0);       // neighbouring rays at the same j: a few rows apart (< 66 cells per axis: lbias4)
        // cells relative to the (middle-of-block) anchor are signed and so is delta: base lowered, lane offset raised by a fixed bias
        const char *sb00 = (const char *)(vp + lin0) - abias4;
        const char *sb01 = sb00 + sy4;
        const char *sb10 = sb00 + sx4;
        const char *sb11 = sb10 + sy4;
        const char *sc00 = sb00 + four, *sc01 = sb01 + four, *sc10 = sb10 + four, *sc11 = sb11 + four;
        const uint32_t off0 = (uint32_t)(delta * 4) + abias4;
        const int lo = max(r.j0, jb) - jb, hi = min(r.j1, jb + TOMO_JB) - jb;
        float acc = 0.f;
        for (int jj = lo; jj < hi; ++jj) {
            const float t = (float)jj;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
            const uint32_t vo = off0 + (uint32_t)__mul24((int)fx, (int)sx4) + (uint32_t)__mul24((int)fy, (int)sy4) + ((uint32_t)(int)fz << 2);
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
// voxel-driven bilinear back-projector (src/back_projection.f90:25-32): one voxel per work-item, lanes along z, loop over
// projections with the accumulator in a register; the voxel centre is transformed on the fly (the reference re-reads a
// (3, n_vox) voxel_centers array per projection).
// FLOAT32 IN THE REFERENCE'S OWN OPERATION ORDER: the Fortran is real(kind=4) throughout -- x' = Ry (Rx (Rz c) + t) as three
// float32 matrix-vector products (src/external_back_projection.f90:20-25), u = x'_1 - origin_1, alpha = u - floor(u), the four
// products det * wx * wz added in the order of :54-65 -- and a float32 voxel coordinate at |x| ~ N/2 carries an ulp of N * 6e-8
// voxel, so a float64 transform gives weights that differ from the reference's by ~1e-5.  Every operation below is the
// reference's, unfused (__fmul_rn / __fadd_rn: no FMA contraction), with the rotation matrices built on the host by the same
// cosf / sinf: parity with back_project_ to float32 rounding instead of 3e-5.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bp_mv3(const float m[3][3], const float x[3], float o[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = __fadd_rn(__fadd_rn(__fmul_rn(m[i][0], x[0]), __fmul_rn(m[i][1], x[1])), __fmul_rn(m[i][2], x[2]));
}

__global__ __launch_bounds__(256) void k_bp_voxel(const BpC *__restrict__ cs, int n_proj, const float *__restrict__ det,
                                                  float *__restrict__ vol, TomoGeomC g, double px, double py, double pz)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int iz = blockIdx.x * 64 + lane, iy = blockIdx.y * 4 + wv, ix = blockIdx.z;
    if (iy >= g.ny || iz >= g.nz) return;
    const size_t img = (size_t)g.ndx * g.ndz;
    // voxel centre as the caller of the reference hands it over: float64 grid (utilities/geometry.py:82-87) cast to float32
    const float c[3] = {(float)(g.org[0] + ix * px), (float)(g.org[1] + iy * py), (float)(g.org[2] + iz * pz)};
    const float o0 = (float)g.org[0], o2 = (float)g.org[2];
    float acc = 0.f;
    for (int ip = 0; ip < n_proj; ++ip) {
        const BpC &b = cs[ip];
        float a1[3], a2[3], a3[3];
        bp_mv3(b.rp, c, a1);                                           // external_back_projection.f90:20
        bp_mv3(b.ra, a1, a2);                                          // :21
#pragma unroll
        for (int k = 0; k < 3; ++k) a2[k] = __fadd_rn(a2[k], b.t[k]); // :22-24
        bp_mv3(b.rb, a2, a3);                                          // :25
        const float u = __fsub_rn(a3[0], o0), v = __fsub_rn(a3[2], o2);
        const float fu = floorf(u), fv = floorf(v);
        if (!(fu >= -2.f && fu <= (float)g.ndx && fv >= -2.f && fv <= (float)g.ndz)) continue;      // no pixel in reach (also NaN)
        const int fx = (int)fu, fz = (int)fv;
        const float ax = __fsub_rn(u, fu), az = __fsub_rn(v, fv);     // :47-48
        const float bx = __fsub_rn(1.f, ax), bz = __fsub_rn(1.f, az);
        const float *im = det + (size_t)ip * img;
        const bool x0 = fx >= 0 && fx < g.ndx, x1 = fx + 1 >= 0 && fx + 1 < g.ndx, z0 = fz >= 0 && fz < g.ndz, z1 = fz + 1 >= 0 && fz + 1 < g.ndz;
        float s = 0.f;                                                 // :54-65, per-pixel bounds tests, products left to right
        if (x0 && z0) s = __fadd_rn(s, __fmul_rn(__fmul_rn(im[(size_t)fx * g.ndz + fz], bx), bz));
        if (x1 && z0) s = __fadd_rn(s, __fmul_rn(__fmul_rn(im[(size_t)(fx + 1) * g.ndz + fz], ax), bz));
        if (x0 && z1) s = __fadd_rn(s, __fmul_rn(__fmul_rn(im[(size_t)fx * g.ndz + fz + 1], bx), az));
        if (x1 && z1) s = __fadd_rn(s, __fmul_rn(__fmul_rn(im[(size_t)(fx + 1) * g.ndz + fz + 1], ax), az));
        acc = __fadd_rn(acc, s);                                       // back_projection.f90:31
    }
    vol[((size_t)ix * g.ny + iy) * g.nz + iz] = acc;
}

