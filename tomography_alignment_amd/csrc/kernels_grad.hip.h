// kernels_grad.hip.h -- projection + 6-DoF pose gradient kernels (plain, dword-gather, DPP neighbour-shift), optionally fused with the residual reduction
// Part of the single translation unit tomo_project.hip (included there, in this order: kernels_ray, kernels_tile,
// kernels_grad); not compiled on its own.

// floor(x) as an integer and x - floor(x), one instruction each (v_floor + v_sub + v_cvt before).  v_fract_f32 never returns
// 1.0: for x a hair below an integer it gives 1 - 2^-24 where x - floorf(x) rounds to 1 -- a 6e-8 change of one weight.
__device__ __forceinline__ int cvt_floor_i32(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float fract_f32(float x) { return __builtin_amdgcn_fractf(x); }
__device__ __forceinline__ int med3_i32_s(int a_uniform, int b, int c)      // median of a wave-uniform value and two per-lane ones
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "s"(a_uniform), "v"(b), "v"(c));
    return r;
}

// work-groups of the gradient kernels per projection (4 detector-x rows x 64 detector-z pixels each), and a work-group's index among them
__host__ __device__ __forceinline__ int grad_wg_per_proj(const TomoGeomC &g) { return ((g.ndx + 3) / 4) * ((g.ndz + 63) / 64); }

// ------------------------------------------------------------------------------------------------
// Shared tail of the gradient kernels: the per-ray 9x3 pose Jacobian applied ONCE to the accumulated S0 = sum_j grad_j and
// S1 = sum_j sf_j grad_j (utilities/ray_voxel_utilities.py:38-49; same algebra as src/ray_wt_grad.f90:136-149), then either the
// plain outputs (proj[n_det], grad[6][n_det]; row_order 1 = the Fortran twin's tx,ty,tz,alpha,beta,phi,
// src/external_forward_projection.f90:56-69) or, FUSED, the residual and the 7 reductions of
// utilities/alignment_functions.py:23-37,124,146: wave shuffles -> LDS -> the work-group's seven float64 partial sums are WRITTEN to
// red[(ipl * 7 + k) * n_wg + w] (ipl = the projection's index in this launch, w = the work-group's index inside the projection,
// fixed by the rays it owns, not by the grid order) and k_cost_grad_reduce adds them in the order of w.  No atomics: the reference's
// sums (utilities/alignment_functions.py:16-37) are deterministic and so are these -- the same pose gives the same seven numbers
// bit for bit, in whatever batch it is evaluated (round 6; until round 5 one float64 atomicAdd per work-group, whose completion
// order L-BFGS-B amplified to 1e-4 px between two identical passes).
// Must be reached by every thread of the work-group (FUSED ends in a barrier).
// ------------------------------------------------------------------------------------------------
template <bool FUSED>
__device__ __forceinline__ void grad_finish(const GradC &gc, const TomoGeomC &g, int ixc, int iz, bool valid, double val, const double s0[3],
                                            const double s1[3], float *__restrict__ proj, float *__restrict__ grad, const float *__restrict__ bvec,
                                            float *__restrict__ resid, double *__restrict__ red, int row_order, int lane, int wv, int ipl, int w)
{
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
            const double ws = wave_sum_d(part[k]);
            if (lane == 0) sh[wv][k] = ws;
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const int k = threadIdx.x;
            const size_t n_wg = (size_t)grad_wg_per_proj(g);
            red[((size_t)ipl * 7 + k) * n_wg + (size_t)w] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
        }
    }
}

// second stage of the fused reduction: one work-group per projection of the launch adds the n_wg partials of each of the seven sums in
// a FIXED order (thread t takes w = t, t + 256, ... in ascending order, then a fixed LDS tree) and stores them at the caller's slot.
__global__ __launch_bounds__(256) void k_cost_grad_reduce(const GradC *__restrict__ gcs, const double *__restrict__ part, double *__restrict__ red, int n_wg)
{
    __shared__ double sh[256];
    const int ipl = blockIdx.x, t = threadIdx.x;
    const int slot = gcs[ipl].slot;
    for (int k = 0; k < 7; ++k) {
        const double *p = part + ((size_t)ipl * 7 + k) * (size_t)n_wg;
        double acc = 0.0;
        for (int w = t; w < n_wg; w += 256) acc += p[w];
        sh[t] = acc;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if (t < h) sh[t] += sh[t + h];
            __syncthreads();
        }
        if (t == 0) red[(size_t)slot * 7 + k] = sh[0];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// projection + 6-DoF pose gradient.  Per sample only the interpolant's spatial gradient is formed
// (3 values); S0 = sum_j grad_j and S1 = sum_j sf_j*grad_j are accumulated and the per-ray 9x3 pose
// Jacobian is applied once (same algebra as src/ray_wt_grad.f90:136-149).
// FUSED: multiply by the residual and reduce to 7 numbers per projection.
// ------------------------------------------------------------------------------------------------
// GRAD ACCURACY (round 4; tools/grad_error_model.py, profiles/round4_grad_error_model.md).  The pose gradient is a sum over a ray of
// spatial gradients that largely cancel (along the beam the sum telescopes: sum_j|g_j| / |sum_j g_j| reached 1e4 on the round-3 soak
// failure, a smooth 59 x 71 x 61 volume with a 21 x 5 detector), so float32 rounding AT THE SIZE OF THE VOXEL VALUES in the per-sample
// gradient (a difference of two lerped values: 6e-8 x |v| per sample) and float32 partial sums of 32 samples showed as 1.06e-5 of
// the row-group maximum -- the sample positions (1e-6 voxel) contribute 5e-7 and were not it.  All three kernels therefore
//   * lerp the corners MINUS corner 000 (two packed subtractions and one add per sample): every rounding then scales with the local
//     differences; the value is v000 + lerp(differences);
//   * accumulate the seven sums in two float32 levels: TOMO_JS samples, then the block of TOMO_JB, then float64.
// Model and GPU agree on the failing geometry (1.06e-5 before, 4.8e-6 after); the remaining error is the float32 rounding of the terms.
#define TOMO_JS 8
// PREC (option grad_v1_prec, a diagnostic of WHERE float32 costs accuracy; 0 = the production arithmetic): bit 0 = sample positions,
// cells and fractions in float64 (no float32 in-block offsets), bit 1 = lerps and per-block sums in float64.
#ifndef GRAD_MIN_WG
#define GRAD_MIN_WG 1      // 256-thread work-groups per CU the register allocation of the v2 / v3 kernels must leave room for (= waves per SIMD)
#endif
template <bool FUSED, int PREC = 0>
__global__ __launch_bounds__(256) void k_proj_grad(const ProjC *__restrict__ pcs, const GradC *__restrict__ gcs,
                                                   const float *__restrict__ vp, float *__restrict__ proj,
                                                   float *__restrict__ grad, const float *__restrict__ bvec,
                                                   float *__restrict__ resid, double *__restrict__ red, TomoGeomC g,
                                                   int row_order)
{
    typedef typename std::conditional<(PREC & 2) != 0, double, float>::type T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ix = blockIdx.y * 4 + wv, ip = blockIdx.z;
    int iz = blockIdx.x * 64 + lane;
    const bool valid = (ix < g.ndx) && (iz < g.ndz);
    const int ixc = min(ix, g.ndx - 1);
    if (iz >= g.ndz) iz = g.ndz - 1;
    const ProjC &c = pcs[ip];
    const GradC &gc = gcs[ip];
    RayCtx r;
    ray_setup(c, g, ixc, iz, valid, r, staged_box(vp, g));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const int64_t sy = g.nzp, sx = (int64_t)g.nyp * g.nzp;
    const T sfs = (T)(g.step / c.rlen);      // sf_j = (j*step)/|r_0|   ray_voxel_utilities.py:151
    double val = 0.0, s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
    for (int jb = r.j0; jb < r.j1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const float *base = vp + ((int64_t)(ia[0] + TOMO_HALO) * sx + (int64_t)(ia[1] + TOMO_HALO) * sy + (ia[2] + TOMO_HALO));
        const int cnt = min(TOMO_JB, r.j1 - jb);
        T av = 0, a0x = 0, a0y = 0, a0z = 0, a1x = 0, a1y = 0, a1z = 0;
        for (int jq = 0; jq < cnt; jq += TOMO_JS) {          // two-level sums, see GRAD ACCURACY above
            T qv = 0, q0x = 0, q0y = 0, q0z = 0, q1x = 0, q1y = 0, q1z = 0;
            const int qe = min(jq + TOMO_JS, cnt);
            for (int jj = jq; jj < qe; ++jj) {
                T wx, wy, wz;
                const float *q;
                if (PREC & 1) {
                    const double xd = r.b[0] + (double)(jb + jj) * r.d[0], yd = r.b[1] + (double)(jb + jj) * r.d[1], zd = r.b[2] + (double)(jb + jj) * r.d[2];
                    const double fx = floor(xd), fy = floor(yd), fz = floor(zd);
                    wx = (T)(xd - fx), wy = (T)(yd - fy), wz = (T)(zd - fz);
                    q = vp + ((int64_t)((int)fx + TOMO_HALO) * sx + (int64_t)((int)fy + TOMO_HALO) * sy + ((int)fz + TOMO_HALO));
                } else {
                    const float t = (float)jj;
                    const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
                    const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
                    wx = x - fx, wy = y - fy, wz = z - fz;
                    q = base + ((int64_t)(int)fx * sx + (int64_t)(int)fy * sy + (int)fz);
                }
                // the corners relative to corner 000 (GRAD ACCURACY): every lerp below then rounds at the size of the local differences
                const T v0 = q[0];
                const T v001 = (T)q[1] - v0, v010 = (T)q[sy] - v0, v011 = (T)q[sy + 1] - v0;
                const T v100 = (T)q[sx] - v0, v101 = (T)q[sx + 1] - v0, v110 = (T)q[sx + sy] - v0, v111 = (T)q[sx + sy + 1] - v0;
                const T d00 = v001, d01 = v011 - v010, d10 = v101 - v100, d11 = v111 - v110;
                const T c00 = wz * d00, c01 = fma(wz, d01, v010), c10 = fma(wz, d10, v100), c11 = fma(wz, d11, v110);
                const T dz0 = fma(wy, d01 - d00, d00), dz1 = fma(wy, d11 - d10, d10);
                const T gz = fma(wx, dz1 - dz0, dz0);
                const T dy0 = c01 - c00, dy1 = c11 - c10;
                const T e0 = fma(wy, dy0, c00), e1 = fma(wy, dy1, c10);
                const T gy = fma(wx, dy1 - dy0, dy0);
                const T gx = e1 - e0;
                qv += v0 + fma(wx, gx, e0);
                const T sf = (T)(jb + jj) * sfs;
                q0x += gx; q0y += gy; q0z += gz;
                q1x = fma(sf, gx, q1x); q1y = fma(sf, gy, q1y); q1z = fma(sf, gz, q1z);
            }
            av += qv; a0x += q0x; a0y += q0y; a0z += q0z; a1x += q1x; a1y += q1y; a1z += q1z;
        }
        val += (double)av;
        s0[0] += (double)a0x; s0[1] += (double)a0y; s0[2] += (double)a0z;
        s1[0] += (double)a1x; s1[1] += (double)a1y; s1[2] += (double)a1z;
    }
    grad_finish<FUSED>(gc, g, ixc, iz, valid, val, s0, s1, proj, grad, bvec, resid, red, row_order, lane, wv, ip,
                       (int)blockIdx.y * ((g.ndz + 63) / 64) + (int)blockIdx.x);
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
__global__ __launch_bounds__(256, GRAD_MIN_WG) void k_proj_grad_v2(const ProjC *__restrict__ pcs, const GradC *__restrict__ gcs,
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
    int ix, ip, iz, wg;                               // wg: this work-group's index inside its projection = x group * z chunks + z chunk
    if (row_order & 16) {
        const int nxg = gridDim.x;
        const int xg = ((nxg & 7) == 0) ? ((int)(blockIdx.x & 7) * (nxg >> 3) + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
        ix = xg * 4 + wv, ip = blockIdx.y, iz = blockIdx.z * 64 + lane;
        wg = xg * (int)gridDim.z + (int)blockIdx.z;
    } else {
        ix = blockIdx.y * 4 + wv, ip = blockIdx.z, iz = blockIdx.x * 64 + lane;
        wg = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    }
    row_order &= 15;
    const bool valid = (ix < g.ndx) && (iz < g.ndz);
    const int ixc = min(ix, g.ndx - 1);
    if (iz >= g.ndz) iz = g.ndz - 1;
    const ProjC &c = pcs[ip];
    const GradC &gc = gcs[ip];
    RayCtx r;
    ray_setup(c, g, ixc, iz, valid, r, staged_box(vp, g));
    const bool nonempty = r.j1 > r.j0;
    const int J0 = __builtin_amdgcn_readfirstlane(wave_min_i32(nonempty ? r.j0 : INT_MAX));
    const int J1 = __builtin_amdgcn_readfirstlane(wave_max_i32(nonempty ? r.j1 : 0));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const uint32_t sy4 = (uint32_t)g.nzp * 4u, sx4 = (uint32_t)g.nyp * (uint32_t)g.nzp * 4u;     // < 2^23 (tomo_check_geometry): signed 24-bit multiplies
    const uint32_t abias4 = tomo_abias_bytes(sx4, sy4) + tomo_lbias_bytes(sx4, sy4);
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
        const char *sb00 = (const char *)(vp + lin0) - abias4;            // cells relative to the anchor, and delta, are signed: fixed biases (tomo_raycore.h)
        const char *sb01 = sb00 + sy4;
        const char *sb10 = sb00 + sx4;
        const char *sb11 = sb10 + sy4;
        const char *sc00 = sb00 + four, *sc01 = sb01 + four, *sc10 = sb10 + four, *sc11 = sb11 + four;   // the z + 1 corners
        const uint32_t off0 = (uint32_t)(delta * 4) + abias4;
        const int lo = max(r.j0, jb) - jb, hi = min(r.j1, jb + TOMO_JB) - jb;      // this lane's samples of the block
        float av = 0.f;
        f32x2 a0xy = {0.f, 0.f}, a1xy = {0.f, 0.f}, az = {0.f, 0.f};              // (S0x, S0y), (S1x, S1y), (S0z, S1z)
        const float sfb = (float)jb * sfs;
        for (int jq = lo; jq < hi; jq += TOMO_JS) {                               // two-level sums (GRAD ACCURACY)
        float qv = 0.f;
        f32x2 q0xy = {0.f, 0.f}, q1xy = {0.f, 0.f}, qz = {0.f, 0.f};
        const int qe = min(jq + TOMO_JS, hi);
        // two samples per trip: all 16 gathers are issued before the first value is used (the kernel waits on memory 3/4 of
        // the time; this doubles the loads in flight per wave).  An odd tail re-reads sample A's address and is masked out.
        for (int jj = jq; jj < qe; jj += 2) {
            const float ta = (float)jj, tb = ta + 1.f;
            const bool two = jj + 1 < qe;
            const float xa = fmaf(ta, dxf, f0[0]), ya = fmaf(ta, dyf, f0[1]), za = fmaf(ta, dzf, f0[2]);
            const float xb = fmaf(tb, dxf, f0[0]), yb = fmaf(tb, dyf, f0[1]), zb = fmaf(tb, dzf, f0[2]);
            const uint32_t voa = off0 + (uint32_t)__mul24(cvt_floor_i32(xa), (int)sx4) + (uint32_t)__mul24(cvt_floor_i32(ya), (int)sy4) + ((uint32_t)cvt_floor_i32(za) << 2);
            const uint32_t vob_ = off0 + (uint32_t)__mul24(cvt_floor_i32(xb), (int)sx4) + (uint32_t)__mul24(cvt_floor_i32(yb), (int)sy4) + ((uint32_t)cvt_floor_i32(zb) << 2);
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
                const float wx = fract_f32(xa), wy = fract_f32(ya), wz = fract_f32(za);
                const f32x2 dy0 = a01 - a00, dy1 = a11 - a10;          // d/dy on the x = 0 / x = 1 faces, at z and z + 1
                const f32x2 o00 = a00 - a00.x, o10 = a10 - a00.x;      // the corners relative to corner 000 (GRAD ACCURACY)
                const f32x2 c0 = o00 + wy * dy0, c1 = o10 + wy * dy1;  // y-lerped
                const f32x2 dx = c1 - c0;                              // d/dx at z, z + 1
                const f32x2 e = c0 + wx * dx;                          // x,y-lerped value at z, z + 1
                const f32x2 dyx = dy0 + wx * (dy1 - dy0);              // d/dy at z, z + 1
                const float gz = e.y - e.x;
                const float gx = fmaf(wz, dx.y - dx.x, dx.x), gy = fmaf(wz, dyx.y - dyx.x, dyx.x);
                qv += a00.x + fmaf(wz, gz, e.x);
                const float sf = fmaf(ta, sfs, sfb);                   // (jb + jj) * step / |r0|, one rounding
                const f32x2 gxy = {gx, gy}, one_sf = {1.f, sf};
                q0xy += gxy;
                q1xy += sf * gxy;
                qz += one_sf * gz;
            }
            {
                const float keep = two ? 1.f : 0.f;
                const float wx = fract_f32(xb), wy = fract_f32(yb), wz = fract_f32(zb);
                const f32x2 dy0 = b01 - b00, dy1 = b11 - b10;
                const f32x2 o00 = b00 - b00.x, o10 = b10 - b00.x;
                const f32x2 c0 = o00 + wy * dy0, c1 = o10 + wy * dy1;
                const f32x2 dx = c1 - c0;
                const f32x2 e = c0 + wx * dx;
                const f32x2 dyx = dy0 + wx * (dy1 - dy0);
                const float gz = keep * (e.y - e.x);
                const float gx = keep * fmaf(wz, dx.y - dx.x, dx.x), gy = keep * fmaf(wz, dyx.y - dyx.x, dyx.x);
                qv = fmaf(keep, b00.x + fmaf(wz, e.y - e.x, e.x), qv);
                const float sf = fmaf(tb, sfs, sfb);
                const f32x2 gxy = {gx, gy}, one_sf = {1.f, sf};
                q0xy += gxy;
                q1xy += sf * gxy;
                qz += one_sf * gz;
            }
        }
        av += qv; a0xy += q0xy; a1xy += q1xy; az += qz;
        }
        val += (double)av;
        s0[0] += (double)a0xy.x; s0[1] += (double)a0xy.y; s0[2] += (double)az.x;
        s1[0] += (double)a1xy.x; s1[1] += (double)a1xy.y; s1[2] += (double)az.y;
    }
    grad_finish<FUSED>(gc, g, ixc, iz, valid, val, s0, s1, proj, grad, bvec, resid, red, row_order, lane, wv, ip, wg);
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
// The gathers are what bounds the gradient kernels under tilt (TA busy 83-100 %; cutting the VALU work by 22 % in round 2 bought
// 1-6 %): time grows linearly with the tilt because
// a 16-lane group then straddles more volume rows; halving the gathers halves that term.
// ------------------------------------------------------------------------------------------------
// lane l <- lane l + 1; lane 63 <- 0 (bound_ctrl: no `old` register to initialise)
__device__ __forceinline__ int dpp_shl1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ float dpp_shl1_f(float v) { return __builtin_bit_cast(float, dpp_shl1_i(__builtin_bit_cast(int, v))); }

template <bool FUSED>
__global__ __launch_bounds__(256, GRAD_MIN_WG) void k_proj_grad_v3(const ProjC *__restrict__ pcs, const GradC *__restrict__ gcs,
                                                      const float *__restrict__ vp, float *__restrict__ proj,
                                                      float *__restrict__ grad, const float *__restrict__ bvec,
                                                      float *__restrict__ resid, double *__restrict__ red, TomoGeomC g,
                                                      int row_order)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ix, ip, iz, wg;
    if (row_order & 16) {                              // cache-ordered grid, see k_proj_grad_v2
        const int nxg = gridDim.x;
        const int xg = ((nxg & 7) == 0) ? ((int)(blockIdx.x & 7) * (nxg >> 3) + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
        ix = xg * 4 + wv, ip = blockIdx.y, iz = blockIdx.z * 64 + lane;
        wg = xg * (int)gridDim.z + (int)blockIdx.z;
    } else {
        ix = blockIdx.y * 4 + wv, ip = blockIdx.z, iz = blockIdx.x * 64 + lane;
        wg = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    }
    row_order &= 15;
    const bool valid = (ix < g.ndx) && (iz < g.ndz);
    const int ixc = min(ix, g.ndx - 1);
    if (iz >= g.ndz) iz = g.ndz - 1;
    const ProjC &c = pcs[ip];
    const GradC &gc = gcs[ip];
    RayCtx r;
    ray_setup(c, g, ixc, iz, valid, r, staged_box(vp, g));
    const bool nonempty = r.j1 > r.j0;
    const int J0 = __builtin_amdgcn_readfirstlane(wave_min_i32(nonempty ? r.j0 : INT_MAX));
    const int J1 = __builtin_amdgcn_readfirstlane(wave_max_i32(nonempty ? r.j1 : 0));
    const float dxf = (float)r.d[0], dyf = (float)r.d[1], dzf = (float)r.d[2];
    const uint32_t sy4 = (uint32_t)g.nzp * 4u, sx4 = (uint32_t)g.nyp * (uint32_t)g.nzp * 4u;     // < 2^23 (tomo_check_geometry): signed 24-bit multiplies
    const uint32_t abias4 = tomo_abias_bytes(sx4, sy4) + tomo_lbias_bytes(sx4, sy4);
    const float sfs = (float)(g.step / c.rlen);
#if defined(TOMO_ABLATE_GRAD_LOADS) && TOMO_ABLATE_GRAD_LOADS == 1
    __shared__ float abl_lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) abl_lds[i] = vp[i];
    __syncthreads();
#endif
    double val = 0.0, s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
    for (int jb = J0; jb < J1; jb += TOMO_JB) {
        int ia[3];
        float f0[3];
        tomo_block_anchor(r.b, r.d, jb, ia, f0);
        const int64_t lin = ((int64_t)(ia[0] + TOMO_HALO) * g.nyp + (ia[1] + TOMO_HALO)) * g.nzp + (ia[2] + TOMO_HALO);
        const int64_t lin0 = readfirstlane_i64(lin);
        const int delta = (int)(lin - lin0);
        const char *sb00 = (const char *)(vp + lin0) - abias4;            // cells relative to the anchor, and delta, are signed: fixed biases (tomo_raycore.h)
        const char *sb01 = sb00 + sy4;
        const char *sb10 = sb00 + sx4;
        const char *sb11 = sb10 + sy4;
        const uint32_t off0 = (uint32_t)(delta * 4) + abias4;
        const int lo = max(r.j0, jb) - jb, hi = min(r.j1, jb + TOMO_JB) - jb;      // this lane's samples of the block
        const bool has = hi > lo;
        const unsigned long long hm = __ballot(has);
        if (hm == 0ull) continue;                                                  // wave-uniform
        // the wave's trip range from the uniform [J0, J1) (a lane that owns no sample of a trip is masked there): no per-block
        // cross-lane min / max
        const int LO = max(J0, jb) - jb, HI = min(J1, jb + TOMO_JB) - jb;
        // a lane with no sample in this block gathers where the first lane that has one takes its first sample
        uint32_t borrow;
        {
            const float t = (float)lo;
            const float x = fmaf(t, dxf, f0[0]), y = fmaf(t, dyf, f0[1]), z = fmaf(t, dzf, f0[2]);
            const uint32_t mine = off0 + (uint32_t)__mul24((int)floorf(x), (int)sx4) + (uint32_t)__mul24((int)floorf(y), (int)sy4) + ((uint32_t)(int)floorf(z) << 2);
            borrow = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(hm));
        }
        const int lo_c = has ? lo : 0, hi_c = has ? hi - 1 : 0;
        const float sfb = (float)jb * sfs;
        float av = 0.f, a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
        // issue: addresses, the four gathers, and -- decided from the addresses alone -- the fallback gathers.
        // (Macros over plain scalars on purpose: a struct passed to helper lambdas was promoted to an LDS alloca, which put
        // a store of every loaded value -- hence a vmcnt(0) wait -- between the two samples' loads.)
        // TOMO_ABLATE_GRAD_LOADS (timing experiments only, results are wrong by construction; profiles/round4_grad_lds_ablation.md):
        // 1 = every gather becomes an LDS read of a resident 32 KB array at an address derived from the gather's own -- what an
        //     LDS-staged form of this kernel could reach at best, with staging, halos and ownership for free;
        // 2 = no loads at all (a value made from the address bits): the floor set by the kernel's VALU work.
#if defined(TOMO_ABLATE_GRAD_LOADS) && TOMO_ABLATE_GRAD_LOADS == 1
#define GS_LOAD(base, off) abl_lds[((uint32_t)(uintptr_t)(base) + (off)) >> 2 & 8191u]
#elif defined(TOMO_ABLATE_GRAD_LOADS) && TOMO_ABLATE_GRAD_LOADS == 2
#define GS_LOAD(base, off) __uint_as_float(0x3f800000u | ((uint32_t)(uintptr_t)(base) + (off) & 0x7fffffu))
#else
#define GS_LOAD(base, off) (*(const float *)((base) + (off)))
#endif
#define GS_DECL(S) float S##v000, S##v010, S##v100, S##v110, S##f001, S##f011, S##f101, S##f111, S##wx, S##wy, S##wz, S##t; /* f*: set, and selected, only where fb */ \
                   unsigned long long S##actm, S##fbm /* lane masks: the sample is the lane's own; it loads its own upper corners */
#define GS_ISSUE(S, JJ)                                                                                                            \
    {                                                                                                                              \
        const int jc = med3_i32_s((JJ), lo_c, hi_c); /* own ray's nearest in-range sample (lo_c <= hi_c) */                          \
        S##actm = hm & __builtin_amdgcn_ballot_w64(jc == (JJ));                                                                    \
        S##t = (float)jc;                                                                                                          \
        const float x = fmaf(S##t, dxf, f0[0]), y = fmaf(S##t, dyf, f0[1]), z = fmaf(S##t, dzf, f0[2]);                            \
        S##wx = fract_f32(x); S##wy = fract_f32(y); S##wz = fract_f32(z);                                                          \
        const uint32_t vo_own = off0 + (uint32_t)__mul24(cvt_floor_i32(x), (int)sx4) + (uint32_t)__mul24(cvt_floor_i32(y), (int)sy4) + ((uint32_t)cvt_floor_i32(z) << 2); \
        const uint32_t vo = has ? vo_own : borrow;                                                                                 \
        S##v000 = GS_LOAD(sb00, vo); S##v010 = GS_LOAD(sb01, vo);                                                                  \
        S##v100 = GS_LOAD(sb10, vo); S##v110 = GS_LOAD(sb11, vo);                                                                  \
        const uint32_t nb = (uint32_t)dpp_shl1_i((int)vo); /* lane 63 receives 0: never vo + 4 */                                 \
        const uint32_t vo4 = vo + 4u;                                                                                              \
        const bool fb = has && jc == (JJ) && nb != vo4;                                                                            \
        S##fbm = S##actm & __builtin_amdgcn_ballot_w64(nb != vo4);                                                                 \
        asm("" : "=v"(S##f001), "=v"(S##f011), "=v"(S##f101), "=v"(S##f111)); /* defined (no instruction); only fb lanes' values are selected */ \
        if (fb) { /* my upper-z cell is not the neighbour's lower-z cell */                                                        \
            S##f001 = GS_LOAD(sb00, vo4); S##f011 = GS_LOAD(sb01, vo4);                                                            \
            S##f101 = GS_LOAD(sb10, vo4); S##f111 = GS_LOAD(sb11, vo4);                                                            \
        }                                                                                                                          \
    }
        // consume: the shifts run with every lane enabled (a DPP source lane that is masked off delivers nothing): take them
        // first, unconditionally, then select
#define GS_CONSUME(S)                                                                                                              \
    {                                                                                                                              \
        const float n001 = dpp_shl1_f(S##v000), n011 = dpp_shl1_f(S##v010), n101 = dpp_shl1_f(S##v100), n111 = dpp_shl1_f(S##v110);   \
        /* the lerps in z and y run on (x = 0, x = 1) register pairs -- the selected upper corners can be written to any       */ \
        /* register, so pairing them is free -- as packed ops; the x stage is scalar                                            */ \
        const f32x2 lo0 = {S##v000, S##v100}, lo1 = {S##v010, S##v110};                                                            \
        const f32x2 up0 = {select_lanes2(n001, S##f001, S##fbm), select_lanes2(n101, S##f101, S##fbm)};                                  \
        const f32x2 up1 = {select_lanes2(n011, S##f011, S##fbm), select_lanes2(n111, S##f111, S##fbm)};                                  \
        const f32x2 d0 = up0 - lo0, d1 = up1 - lo1;                                  /* d/dz at y = 0, y = 1 */                    \
        const f32x2 o0 = lo0 - S##v000, o1 = lo1 - S##v000;                          /* relative to corner 000 (GRAD ACCURACY) */  \
        const f32x2 c0 = o0 + S##wz * d0, c1 = o1 + S##wz * d1;                      /* z-lerped corners */                        \
        const f32x2 dz = d0 + S##wy * (d1 - d0);                                                                                   \
        const f32x2 dy = c1 - c0;                                                                                                  \
        const f32x2 e = c0 + S##wy * dy;                                                                                           \
        const float keep = select_lanes(1.f, S##actm);                                                                             \
        const float gz = fmaf(S##wx, dz.y - dz.x, dz.x);                                                                           \
        const float gy = fmaf(S##wx, dy.y - dy.x, dy.x);                                                                           \
        const float gx = e.y - e.x;                                                                                                \
        qv = fmaf(keep, S##v000 + fmaf(S##wx, gx, e.x), qv);                                                                       \
        const float sf = keep * fmaf(S##t, sfs, sfb);             /* (jb + jj) * step / |r0|, one rounding; 0 where masked */      \
        q0x = fmaf(keep, gx, q0x); q0y = fmaf(keep, gy, q0y); q0z = fmaf(keep, gz, q0z);                                           \
        q1x = fmaf(sf, gx, q1x); q1y = fmaf(sf, gy, q1y); q1z = fmaf(sf, gz, q1z);                                                 \
    }
        for (int jq = LO; jq < HI; jq += TOMO_JS) {                                // two-level sums (GRAD ACCURACY)
            float qv = 0.f, q0x = 0.f, q0y = 0.f, q0z = 0.f, q1x = 0.f, q1y = 0.f, q1z = 0.f;
            const int qe = min(jq + TOMO_JS, HI);
            for (int jj = jq; jj < qe; jj += 2) {                                  // wave-uniform trip count; two samples in flight
                GS_DECL(a_);
                GS_DECL(b_);
                GS_ISSUE(a_, jj)
                GS_ISSUE(b_, jj + 1)                                               // past the end: clamped address, act = false
                GS_CONSUME(a_)
                GS_CONSUME(b_)
            }
            av += qv; a0x += q0x; a0y += q0y; a0z += q0z; a1x += q1x; a1y += q1y; a1z += q1z;
        }
#undef GS_DECL
#undef GS_LOAD
#undef GS_ISSUE
#undef GS_CONSUME
        val += (double)av;
        s0[0] += (double)a0x; s0[1] += (double)a0y; s0[2] += (double)a0z;
        s1[0] += (double)a1x; s1[1] += (double)a1y; s1[2] += (double)a1z;
    }
    grad_finish<FUSED>(gc, g, ixc, iz, valid, val, s0, s1, proj, grad, bvec, resid, red, row_order, lane, wv, ip, wg);
}


