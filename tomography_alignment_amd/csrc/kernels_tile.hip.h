// kernels_tile.hip.h -- the GENERAL (tilted-pose) LDS tile kernels, forward and adjoint (k_tile<FWD>), and what every tile kernel
// shares: the tile shape, the per-projection constants (AdjC), k_absmax and the small inline-asm helpers.
// The flat (untilted-pose) tile kernels are in kernels_tile_flat.hip.h, the gather-form flat adjoint in kernels_tile_gather.hip.h.
// Part of the single translation unit tomo_project.hip (included there, in this order: kernels_ray, kernels_tile, kernels_tile_flat,
// kernels_tile_gather, kernels_grad); not compiled on its own.

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
// Waves per work-group of the GENERAL tile kernels (k_tile) and the adjoint's projections per flush -- tuned in round 5 (1024^3 x 1024 angles,
// +-1 deg, dense volume; profiles/round5_tile_waves_sweep.md).  Until then both ran 8 waves (two work-groups per CU by the 74 KB image):
//   forward   8: 932 ms   10: 1108   12: 842   14: 915   16: 837   (16 with <= 64 VGPRs, two work-groups per CU: 833)
//   adjoint   8 waves x 64 projections per flush: 1371 ms   10 x 80: 1989   12 x 96: 1239   12 x 192: 1226   12 x 384: 1218   14 x 112: 1730   16 x 128: 1539
// Wave counts that are not a multiple of the 4 SIMDs leave SIMDs unevenly loaded (10, 14: slower than 8).  The forward has no barrier after
// staging: ONE 16-wave work-group per CU beats two of 8; the adjoint flushes behind barriers every batch: two work-groups of 12 overlap
// each other's flushes, and a longer batch means fewer of them.
#ifndef TILE_FWD_WAVES
#define TILE_FWD_WAVES 16
#endif
#ifndef TILE_ADJ_WAVES
#define TILE_ADJ_WAVES 12
#endif
#ifndef TILE_ADJ_BATCH
#define TILE_ADJ_BATCH 192
#endif
#ifndef TILE_MIN_WG
#define TILE_MIN_WG 1
#endif
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
typedef __attribute__((address_space(3))) float lds_cfloat;
// ... the same up to the z lerp: (value at z, value at z + 1)
__device__ __forceinline__ f32x2 bilerp_pairs(f32x2 p00, f32x2 p01, f32x2 p10, f32x2 p11, float wx, float wy)
{
    const f32x2 c0 = p00 + wy * (p01 - p00);
    const f32x2 c1 = p10 + wy * (p11 - p10);
    return c0 + wx * (c1 - c0);
}
__device__ __forceinline__ float trilerp_pairs(f32x2 p00, f32x2 p01, f32x2 p10, f32x2 p11, float wx, float wy, float wz)
{
    const f32x2 c0 = p00 + wy * (p01 - p00);
    const f32x2 c1 = p10 + wy * (p11 - p10);
    const f32x2 e = c0 + wx * (c1 - c0);
    return fmaf(wz, e.y - e.x, e.x);
}

// v if the lane's bit is set in the wave-uniform mask m, else 0.  Written as the SGPR-pair form of v_cndmask_b32: the form that reads
// VCC retires one per ~13 clk per SIMD on gfx950 against ~3 for this one (tools/issue_bench.hip, profiles/round2_issue_bench.log).
__device__ __forceinline__ float select_lanes(float v, unsigned long long m)
{
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}

__device__ __forceinline__ unsigned select_lanes_u(unsigned v, unsigned long long m)
{
    unsigned r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}

// acc + v in the lanes whose bit is set in the wave-uniform mask m, acc unchanged in the others: the add runs under EXEC & m (three
// scalar instructions, no branch) instead of a v_cndmask + v_add pair -- one VALU instruction less per sample in a VALU-bound loop
__device__ __forceinline__ float add_lanes(float acc, float v, unsigned long long m)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tv_add_f32 %0, %0, %2\n\ts_mov_b64 exec, %1"
                 : "+v"(acc), "=&s"(saved) : "v"(v), "s"(m) : "scc");      // s_and_b64 writes SCC (round 3: without the clobber the compiler kept a
                                                                           // loop condition in SCC across this statement -- the loop never ended)
    return acc;
}

// The last (z) lerp and the accumulation of a sample, for the lanes of mask m only:  acc += e0 ;  accz += fz * (e1 - e0)  with fz the
// z fraction as it comes out of v_cvt_f32_u32 (x 2^32): the caller adds accz * 2^-32 once per row chunk instead of scaling every
// sample's fraction -- with the select gone too (add_lanes) that is 3 VALU instructions where there were 5.
__device__ __forceinline__ void zlerp_acc_lanes(float &acc, float &accz, float e0, float e1, float fz, unsigned long long m)
{
    unsigned long long saved;
    const float d = e1 - e0;
    asm volatile("s_mov_b64 %2, exec\n\ts_and_b64 exec, exec, %6\n\tv_add_f32 %0, %0, %3\n\tv_fmac_f32 %1, %4, %5\n\ts_mov_b64 exec, %2"
                 : "+v"(acc), "+v"(accz), "=&s"(saved) : "v"(e0), "v"(fz), "v"(d), "s"(m) : "scc");   // SCC: see add_lanes
}

// b where the lane's bit is set in m, else a
__device__ __forceinline__ float select_lanes2(float a, float b, unsigned long long m)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}

// lane l <- lane l - 1 (lane 0 <- 0)
__device__ __forceinline__ float dpp_shr1_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// per-lane 64-bit position + wave-uniform 64-bit step in one VALU instruction, kept opaque so that the low word (the fraction)
// and the high word (the cell) are used as they come (the compiler otherwise re-forms base + scalar offset and adds the low
// words a second time)
__device__ __forceinline__ int64_t add64_vs(int64_t p, int64_t step)
{
    asm("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p) : "s"(step));
    return p;
}

// the same sum into a NEW register pair (where the first operand is needed again: no copy in front of the in-place form)
__device__ __forceinline__ int64_t sum64_vs(int64_t p, int64_t step)
{
    int64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(p), "s"(step));
    return r;
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
// dword index of cell (cx, cy, cz) of the LDS image, (cx * ALY + cy) * ALZ + cz, in two full-rate instructions (v_lshl_add_u32,
// v_mad_u32_u24 -- left to itself the compiler picked the quarter-rate v_mad_u64_u32 once the cells were no longer provably small).
// Round 3: the cells of a lane that does NOT own the sample are not clamped into the image any more (two v_min_u32 and one v_and
// per sample).  Whatever (cx, cy) are, cz's low five bits -- the lanes are consecutive detector-z rays -- set the bank, so the 32
// lanes of a bank group still hit 32 banks; the dword lies either somewhere in the image (read and discarded by the ownership
// select / added an integer 0) or beyond the work-group's LDS allocation, where DS reads return 0 and DS writes are dropped (LDS
// accesses are bounds-checked in hardware; they never fault).
__device__ __forceinline__ unsigned tile_cell_dword(unsigned cx, unsigned cy, unsigned cz)
{
    static_assert(ALY * ALZ == 1088 && ALZ == 64, "immediates below");
    unsigned u, r;
    asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(u) : "v"(cy), "v"(cz));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(cx), "s"((unsigned)(ALY * ALZ)), "v"(u));      // VOP3 takes no 32-bit literal: the pitch sits in an SGPR
    return r;
}

#ifdef TOMO_TILE_CLAMP          // A/B switch (tools/gpu_r3c.sh): the round-2 form with the non-owners' cells clamped into the image
#define TILE_CLAMP(v, hi) min((v), (unsigned)(hi))
#define TILE_ZMASK(v) ((v) & 63u)
#else
// HARDWARE BEHAVIOUR RELIED ON: out-of-range DS reads return 0 and out-of-range DS writes / atomics are dropped on gfx9 / CDNA.
// Any other target must build with -DTOMO_TILE_CLAMP (kept in the test matrix: tests/test_gpu_fuzz.py runs against either build).
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "k_tile without TOMO_TILE_CLAMP relies on the LDS bounds check of gfx9-family targets"
#endif
#define TILE_CLAMP(v, hi) (v)
#define TILE_ZMASK(v) (v)
#endif
// The forward kernels skip volume tiles that hold no non-zero voxel -- but a work-group that only finds that out and returns still has to
// be dispatched, in launch order, to a CU with room for it (kernels_tile_flat.hip.h, "live-block list": 248 against 198 ms for the same live
// work on the flat forward).  k_tile_live classifies the general forward's tiles (flag = 1: some voxel of the 17 x 17 x 61 image is not zero);
// k_fwd_compact (kernels_tile_flat.hip.h) compacts the live ones in launch order and work-group i of k_tile<true> takes tile list[1 + i].
__global__ __launch_bounds__(256) void k_tile_live(const float *__restrict__ vol, TomoGeomC g, int tile_x0, int nzt, int nty, unsigned char *__restrict__ flags)
{
    const int b = (int)blockIdx.x;
    const int z0 = -1 + (b % nzt) * ATZ, y0 = -1 + ((b / nzt) % nty) * ATY, x0 = -1 + (b / (nzt * nty) + tile_x0) * ATX;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gz = z0 + lane;
    const bool inz = lane <= ATZ && gz >= 0 && gz < g.nz;
    bool nz = false;
    for (int c = wv; c < ALX * ALY; c += 4) {
        const int gx = x0 + c / ALY, gy = y0 + c % ALY;
        if (gx < 0 || gx >= g.nx || gy < 0 || gy >= g.ny) continue;
        if (inz) nz |= vol[((size_t)gx * g.ny + gy) * g.nz + gz] != 0.f;
        if (__builtin_amdgcn_ballot_w64(nz)) break;
    }
    const int l = __syncthreads_or(nz);
    if (threadIdx.x == 0) flags[b] = (unsigned char)(l ? 1 : 0);
}

// detector-z range [izl, izh] (clipped to the detector) of the rays that can own a sample in the tile whose owned box has centre bc: a linear
// functional over a box = centre value +- sum |coef| * half-extent (+ 2e-2 for float32).  ONE definition for k_tile and k_tile_adj_live.
__device__ __forceinline__ void tile_iz_range(const AdjC &c, const float bc[3], int ndz, float &izl, float &izh)
{
    const float qx = bc[0] - (float)c.p0[0], qy = bc[1] - (float)c.p0[1], qz = bc[2] - (float)c.p0[2];
    const float m10 = (float)c.minv[1][0], m11 = (float)c.minv[1][1], m12 = (float)c.minv[1][2];
    const float izm = m10 * qx + m11 * qy + m12 * qz;
    const float izr = fabsf(m10) * (0.5f * ATX) + fabsf(m11) * (0.5f * ATY) + fabsf(m12) * (0.5f * ATZ) + 2e-2f;
    izl = fmaxf(izm - izr, 0.f);
    izh = fminf(izm + izr, (float)(ndz - 1));
}
// "some detector-z plane in reach of the tile holds a non-zero sinogram value" (zcum: prefix counts of k_sino_zflags' flags)
__device__ __forceinline__ bool tile_iz_any(const int *__restrict__ zcum, int ndz, float izl, float izh)
{
    const int za = (int)izl, zb = min(ndz - 1, (int)izh + 1);
    return zcum[zb + 1] != zcum[za];
}

// The general ADJOINT's live tiles: a tile none of whose projections can bring a non-zero sinogram value receives nothing.  One wave per
// tile, lanes over the projections; the flagged tiles are compacted by k_fwd_compact and k_tile<false> takes them from the list -- like the
// forward's all-zero tiles, a tile that only returns would still wait in launch order for room on a CU.
__global__ __launch_bounds__(64) void k_tile_adj_live(const AdjC *__restrict__ pcs, int n_proj, TomoGeomC g, int tile_x0, int nzt, int nty,
                                                      const int *__restrict__ zcum, unsigned char *__restrict__ flags)
{
    const int b = (int)blockIdx.x;
    const int z0 = -1 + (b % nzt) * ATZ, y0 = -1 + ((b / nzt) % nty) * ATY, x0 = -1 + (b / (nzt * nty) + tile_x0) * ATX;
    const float bc[3] = {(float)x0 + 0.5f * ATX, (float)y0 + 0.5f * ATY, (float)z0 + 0.5f * ATZ};
    bool any = false;
    for (int ip = threadIdx.x; ip < n_proj && !any; ip += 64) {
        float izl, izh;
        tile_iz_range(pcs[ip], bc, g.ndz, izl, izh);
        any = !(izl > izh + 1.f) && tile_iz_any(zcum, g.ndz, izl, izh);
    }
    const bool live = __builtin_amdgcn_ballot_w64(any) != 0;
    if (threadIdx.x == 0) flags[b] = live ? 1 : 0;
}

// list != nullptr: a 1-D grid, work-group i takes the i-th live tile (nzt, nty: the tile grid the ids index); nullptr: a 3-D grid of tiles
template <bool FWD, int NW>
__global__ __launch_bounds__(NW * 64, TILE_MIN_WG) void k_tile(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                         float *__restrict__ vol, TomoGeomC g, const unsigned *__restrict__ absmax_bits,
                                                         float weight_bound, int tile_x0, const int *__restrict__ list, int nzt, int nty,
                                                         const int *__restrict__ zcum)
{
    __shared__ int acc[ALX * ALY * ALZ + 4];        // + pad: a masked-out lane may read one dword past the last row (its value is discarded)
    const lds_cfloat *img3 = (const lds_cfloat *)acc;       // explicit LDS pointer: offsets made opaque below must still give ds_read
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tz = (int)blockIdx.x, ty = (int)blockIdx.y, tx = (int)blockIdx.z;
    if (list) {
        if ((int)blockIdx.x >= list[0]) return;
        const int b = list[1 + blockIdx.x];
        tz = b % nzt; ty = (b / nzt) % nty; tx = b / (nzt * nty);
    }
    const int z0 = -1 + tz * ATZ, y0 = -1 + ty * ATY, x0 = -1 + (tx + tile_x0) * ATX;
    float scale = 1.f, inv_scale = 1.f;
    if (FWD) {
        bool any_nz = false;
        for (int e = threadIdx.x; e < ALX * ALY * ALZ; e += NW * 64) {
            const int lz = e % ALZ, t2 = e / ALZ, ly = t2 % ALY, lx = t2 / ALY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
            float v = 0.f;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) v = vol[((size_t)gx * g.ny + gy) * g.nz + gz];
            ((float *)acc)[e] = v;
            any_nz |= (v != 0.f);
        }
        if (!__syncthreads_or(any_nz)) return;                   // an all-zero tile contributes nothing to any ray (none is left when the list is used)
    } else {
        const float ymax = __uint_as_float(*absmax_bits);
        if (!(ymax > 0.f)) return;                               // A^T 0 = 0 (vol already holds the right answer)
        // |image| <= TILE_ADJ_BATCH * ymax * weight_bound  ->  keep it below 2^30
        scale = 1073741824.f / ((float)min(n_proj, TILE_ADJ_BATCH) * ymax * weight_bound);
        inv_scale = 1.f / scale;
        for (int e = threadIdx.x; e < ALX * ALY * ALZ; e += NW * 64) acc[e] = 0;
        __syncthreads();
    }
    const float bc[3] = {(float)x0 + 0.5f * ATX, (float)y0 + 0.5f * ATY, (float)z0 + 0.5f * ATZ};   // owned-box centre
    const float ext[3] = {(float)ATX, (float)ATY, (float)ATZ};
    const int64_t org[3] = {(int64_t)x0 << 32, (int64_t)y0 << 32, (int64_t)z0 << 32};
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const float two_m32 = 2.3283064365386963e-10f;

    const int batch = FWD ? n_proj : TILE_ADJ_BATCH;
    for (int ip0 = 0; ip0 < n_proj; ip0 += batch) {
        const int ip1 = min(n_proj, ip0 + batch);
        for (int ip = ip0 + wv; ip < ip1; ip += NW) {      // one wave owns a whole (tile, projection): set-up runs once
            const AdjC &c = pcs[ip];
            // Range work is CONSERVATIVE set-up in float32 (coordinates < 2^11: float32 error < 1e-3 voxel, margins 2e-2): it
            // only has to cover the owned samples; exact ownership is decided per sample from the fixed-point position.
            // lattice-coordinate ranges of the owned box: a linear functional over a box = centre value +- sum |coef|*half-extent
            const float qx = bc[0] - (float)c.p0[0], qy = bc[1] - (float)c.p0[1], qz = bc[2] - (float)c.p0[2];
            const float m00 = (float)c.minv[0][0], m01 = (float)c.minv[0][1], m02 = (float)c.minv[0][2];
            const float ixc = m00 * qx + m01 * qy + m02 * qz;
            const float ixr = fabsf(m00) * (0.5f * ATX) + fabsf(m01) * (0.5f * ATY) + fabsf(m02) * (0.5f * ATZ) + 2e-2f;
            const int ix_lo = max(0, (int)ceilf(fmaxf(ixc - ixr, -1.f)));
            const int ix_hi = min(g.ndx - 1, (int)floorf(fminf(ixc + ixr, (float)g.ndx)));
            if (ix_lo > ix_hi) continue;
            float izl, izh;
            tile_iz_range(c, bc, g.ndz, izl, izh);
            if (izl > izh + 1.f) continue;
            // adjoint: every sample this tile owns belongs to a ray with iz in [izl, izh]; zcum[i] = number of detector-z planes < i of the
            // call's sinogram that hold a non-zero value (tomo_project.hip: k_sino_zflags, k_zflags_prefix).  None in reach: the rays that
            // own samples here carry zeros, the others own nothing here -- this (tile, projection) adds nothing
            if (!FWD && zcum && !tile_iz_any(zcum, g.ndz, izl, izh)) continue;
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
                    // TOMO_ABLATE_TILE (timing experiments only, results are wrong by construction; profiles/round6_tilted_split.md): bit 0 = no sample
                    // loop (what is left is staging / zeroing, the per-(tile, projection) and per-row set-up, the row's global atomic or sinogram
                    // read, and the flushes); bit 1 = adjoint without the flushes; bit 2 = forward without its atomics; bit 3 = forward with
                    // plain stores instead of atomics
#if defined(TOMO_ABLATE_TILE) && (TOMO_ABLATE_TILE & 1)
                    const int cnt = 0;
#else
                    const int cnt = jhi - jlo;
#endif
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
                            float part = 0.f, partz = 0.f;
                            for (int jj = 0; jj < cnt; ++jj) {
                                const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32), lz = (unsigned)(pz >> 32);
                                static_assert(ATX == ATY && (ATX & (ATX - 1)) == 0, "ownership test uses (lx | ly) < ATX");
                                static_assert(ALY == 17 && ALZ == 64, "cell index is written with shifts");
                                const unsigned long long own = __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATX) & __builtin_amdgcn_ballot_w64(lz < (unsigned)ATZ);
                                // Lanes that do not own the sample still read (and discard): from a cell clamped into the image whose LDS
                                // bank is the one their own z would have -- lanes sit on consecutive z, so the 32 lanes of a bank group keep
                                // 32 different banks.  (They used to read word 0: every such lane then hit bank 0 together with whichever
                                // owner lane mapped there -- SQ_LDS_BANK_CONFLICT was 47 % of the kernel's LDS cycles, LDS 81 % busy.)
                                const unsigned cx = TILE_CLAMP(lx, ATX - 1), cy = TILE_CLAMP(ly, ATY - 1), cz = TILE_ZMASK(lz);      // see tile_cell_dword
                                const unsigned eb = tile_cell_dword(cx, cy, cz) << 2;                   // byte offset of the cell
                                const f32x2 wxy = f32x2{(float)(unsigned)px, (float)(unsigned)py} * two_m32;          // one packed multiply for two of the three fractions
                                const float wx = wxy.x, wy = wxy.y, fz32 = (float)(unsigned)pz;           // z fraction x 2^32 (scaled once per chunk)
                                unsigned eb1 = eb + ALY * ALZ * 4;
                                asm("" : "+v"(eb1));                      // one add for the x+1 face; its y+1 rows sit within ds_read2's offset range
                                const lds_cfloat *q = (const lds_cfloat *)((const __attribute__((address_space(3))) char *)img3 + eb);
                                const lds_cfloat *q1 = (const lds_cfloat *)((const __attribute__((address_space(3))) char *)img3 + eb1);
                                const f32x2 p00 = {q[0], q[1]}, p01 = {q[ALZ], q[ALZ + 1]};
                                const f32x2 p10 = {q1[0], q1[1]}, p11 = {q1[ALZ], q1[ALZ + 1]};
                                const f32x2 e = bilerp_pairs(p00, p01, p10, p11, wx, wy);
                                zlerp_acc_lanes(part, partz, e.x, e.y, fz32, own);
                                px = add64_vs(px, c.fd[0]); py = add64_vs(py, c.fd[1]); pz = add64_vs(pz, c.fd[2]);
                            }                                          // (two samples per trip, 8 reads in flight: measured no faster -- the loop is VALU-issue bound)
                            part = fmaf(partz, two_m32, part);
#if defined(TOMO_ABLATE_TILE) && (TOMO_ABLATE_TILE & 4)
                            if (lane_ok && part == 12345.678f) *pr = part;         // timing experiment: no atomic (the address is still formed)
#elif defined(TOMO_ABLATE_TILE) && (TOMO_ABLATE_TILE & 8)
                            if (lane_ok) *pr = part;                               // timing experiment: a plain store instead of the atomic
#else
                            if (lane_ok) atomicAdd(pr, part);          // 64 consecutive floats per wave: the full-rate atomic shape
#endif
                        } else {
                            const float ys = (lane_ok ? *pr : 0.f) * scale;
                            // straight-line body: a lane that does not own the sample adds integer 0 to a cell clamped into the image on
                            // its own z bank (see the forward) -- no exec-mask switches between samples (two s_cbranch_execz per sample
                            // before: 1.52 -> 1.41 ms/angle at 1024^3)
                            f32x2 ksc = {-two_m32, two_m32}, kof = {1.f, 0.f};
                            asm("" : "+v"(kof));                        // held in a VGPR pair (else re-materialised per sample)
                            // 64 rays whose sinogram values are all zero add integer zeros: skip them (wave-uniform; a residual sinogram is
                            // exactly zero wherever the rays miss the object's support -- a third of the rows in the benchmark's SIRT step)
                            const int cnt_a = __builtin_amdgcn_ballot_w64(ys != 0.f) ? cnt : 0;
                            for (int jj = 0; jj < cnt_a; ++jj) {
                                const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32), lz = (unsigned)(pz >> 32);
                                const float yo = select_lanes(ys, __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATX) & __builtin_amdgcn_ballot_w64(lz < (unsigned)ATZ));
                                const unsigned cx = TILE_CLAMP(lx, ATX - 1), cy = TILE_CLAMP(ly, ATY - 1), cz = TILE_ZMASK(lz);      // unclamped: a non-owner adds integer 0 (tile_cell_dword)
                                // (1 - w, w) per axis as one packed pair, products as packed multiplies: 3 + 7 packed instructions for what
                                // were 6 + 14 scalar ones (same operations in the same order)
                                const float fx = (float)(unsigned)px, fy = (float)(unsigned)py, fz = (float)(unsigned)pz;
                                const f32x2 wxp = f32x2{fx, fx} * ksc + kof, wyp = f32x2{fy, fy} * ksc + kof, wzp = f32x2{fz, fz} * ksc + kof;
                                const f32x2 a = yo * wxp;
                                const f32x2 b0 = a.x * wyp, b1 = a.y * wyp;
                                const f32x2 c00 = b0.x * wzp, c01 = b0.y * wzp, c10 = b1.x * wzp, c11 = b1.y * wzp;
                                int *q = &acc[tile_cell_dword(cx, cy, cz)];
                                atomicAdd(q, cvt_round_i32(c00.x));
                                atomicAdd(q + 1, cvt_round_i32(c00.y));
                                atomicAdd(q + ALZ, cvt_round_i32(c01.x));
                                atomicAdd(q + ALZ + 1, cvt_round_i32(c01.y));
                                atomicAdd(q + ALY * ALZ, cvt_round_i32(c10.x));
                                atomicAdd(q + ALY * ALZ + 1, cvt_round_i32(c10.y));
                                atomicAdd(q + ALY * ALZ + ALZ, cvt_round_i32(c11.x));
                                atomicAdd(q + ALY * ALZ + ALZ + 1, cvt_round_i32(c11.y));
                                px = add64_vs(px, c.fd[0]); py = add64_vs(py, c.fd[1]); pz = add64_vs(pz, c.fd[2]);
                            }
                        }
                    }
                }
            }
        }
        if (FWD) break;
        __syncthreads();
#if defined(TOMO_ABLATE_TILE) && (TOMO_ABLATE_TILE & 2)
        continue;
#endif
        // flush this batch: interior of the image is exclusively ours, the +1 faces are shared => global atomics
        for (int e = threadIdx.x; e < ALX * ALY * ALZ; e += NW * 64) {
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

