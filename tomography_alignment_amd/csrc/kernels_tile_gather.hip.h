// kernels_tile_gather.hip.h -- the GATHER-form adjoint for untilted unit lattices (k_adj_gather_flat<NJ>): no atomics, each voxel
// written once.  Needs kernels_tile.hip.h (helpers) before it; part of the translation unit tomo_project.hip.

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
//      rows (pitch 68 dwords: 16-byte aligned plane quads, lanes reading different rows hit different bank quads);
//   3. lane = COLUMN again, 64 plane accumulators per lane (statically indexed registers, as plane pairs): per four planes
//      3 ds_read_b128 at row(lane) + immediate plane offset and 6 v_pk_fma_f32 with the lane's own W0..W2 -- no broadcasts, no
//      address arithmetic.
// Same sums as k_tile_flat<false> (which needs 4 ds_add_u32 per sample and lane), regrouped by voxel instead of by sample.
// ------------------------------------------------------------------------------------------------
#define GTX 8
#define GTY 8
#define GROWS 14          // rows a tile can touch: i0 spreads over <= 7 (|m00| + |m01|) <= 10.2 -> 11 values, + 3
#define GPITCH 68          // LDS row pitch in dwords: a multiple of 4, so that a row's plane quads (p .. p+3) are 16-byte aligned for ds_read_b128; rows r, r+1, ...
                           // of one plane quad fall in different bank quads (4 r + p mod 64)
#ifndef GWAVES
#define GWAVES 4
#endif
#ifndef GPX
#define GPX 8              // (x, y) tile patch that one XCD's resident work-groups cover together
#define GPY 12
#endif

struct GfC {
    int64_t fp0x, fp0y, fux, fuy, fdx, fdy;   // x, y of the 32.32 lattice  p = fp0 + ix fu + j fd
    float m00, m01, m10, m11;                 // (ix, j) = M ((x, y) - p0)
    float p0x, p0y, tau;
    int32_t n, zc, slot;
};

// flags[iz] = 1 when ANY value of the sinogram in detector-z plane iz is not zero (all projections, all rows of the call).  Thread = plane
// (coalesced), work-groups stride over the n_rows = n_proj * ndx rows; flags pre-zeroed, benign races (everybody stores 1).
__global__ __launch_bounds__(256) void k_sino_zflags(const float *__restrict__ proj, long long n_rows, int ndz, unsigned char *__restrict__ flags)
{
    const int iz = (int)(blockIdx.x * 256 + threadIdx.x);
    if (iz >= ndz) return;
    bool nz = false;
    for (long long r = blockIdx.y; r < n_rows; r += gridDim.y) nz |= proj[(size_t)r * ndz + iz] != 0.f;
    if (nz) flags[iz] = 1;
}

// cum[i] = number of flagged planes below i, i = 0 .. n (one work-group; the general adjoint asks "any flagged plane in [a, b]?" with two loads)
__global__ __launch_bounds__(1024) void k_zflags_prefix(const unsigned char *__restrict__ flags, int n, int *__restrict__ cum)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        const bool f = i < n && flags[i] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        if (lane == 0) wsum[wv] = (int)__builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (i < n) cum[i] = off + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) cum[n] = base;
}

// How the 64-plane z chunks are grouped into work-groups of GWAVES: *shift = (GWAVES - first live chunk mod GWAVES) mod GWAVES, so that the
// first chunk that can receive anything starts a work-group (work-group q holds chunks q * GWAVES - shift ...).  With the chunks grouped from 0
// a live range that starts mid-group leaves a work-group with dead waves at each end; its live waves then compute the dead ones' share of the
// weight tables.  A chunk c is live when a flagged sinogram plane lies in [64 c - zc_hi - 1, 64 c + 63 - zc_lo] (as in the kernel).
__global__ __launch_bounds__(64) void k_zchunk_shift(const unsigned char *__restrict__ flags, int ndz, int nz, int zc_lo, int zc_hi, int *__restrict__ shift)
{
    const int n_chunk = (nz + 63) / 64;
    int first = n_chunk;
    for (int c = threadIdx.x; c < n_chunk; c += 64) {
        bool any = false;
        for (int i = max(0, 64 * c - zc_hi - 1); i <= min(ndz - 1, 64 * c + 63 - zc_lo) && !any; ++i) any = flags[i] != 0;
        if (any) first = min(first, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    if (threadIdx.x == 0) *shift = first >= n_chunk ? 0 : (GWAVES - first % GWAVES) % GWAVES;
}

template <int NJ>      // samples per row that can reach a column: 3 for step >= 0.95 voxel, 6 for step >= 0.475
__global__ __launch_bounds__(GWAVES * 64, 16 / GWAVES) void k_adj_gather_flat(const GfC *__restrict__ cs, int n_proj, const float *__restrict__ proj,
                                                                 float *__restrict__ vol, TomoGeomC g, int xs, int xe, int patched,
                                                                 const unsigned char *__restrict__ zflags, int zc_lo, int zc_hi, const int *__restrict__ zshift)
{
    __shared__ __attribute__((aligned(16))) float rows[GWAVES][GROWS * GPITCH];
    __shared__ float4 wtab[3][GWAVES][64];          // [group mod 3][projection of the group][column] = (i0, W0, W1, W2)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the work-group owns 8 x 8 voxel columns; its four waves take four consecutive 64-plane chunks of them.
    // Work-group -> tile mapping: the ~96 work-groups resident on one XCD (32 CUs x 3) should share sinogram rows in that XCD's
    // L2 -- with a plain (z, y, x) grid they formed a 128 x 1.5-tile strip with almost no common rows and every row load went
    // to the fabric (0.88 TB per launch at 1024^3).  Work-groups are dealt to the XCDs round-robin in dispatch order, so XCD k
    // sees the linear ids k, k+8, ...: those are mapped to compact GPX x GPY patches of (x, y) tiles of one z quad (patch
    // p*8 + k for the p-th group of 96 of them): ~10x row reuse within a patch.
    // (Small grids keep the plain order, patched = 0: the patch grid is padded to 8 x 96 work-groups, which costs more than
    // the reuse gains below ~256 patches.  Measured at 1024^3: same speed; fabric traffic -64 % on a 64-angle launch, -15 %
    // (0.89 -> 0.76 TB) over 1024 angles, where the work-groups of a patch drift apart in angle index.)
    const int ntx = (xe - xs + GTX - 1) / GTX, nty = (g.ny + GTY - 1) / GTY, nzq = (g.nz + 64 * GWAVES - 1) / (64 * GWAVES) + 1;
    // (+ 1: the chunks are grouped from -shift, k_zchunk_shift above; with shift = 0 the last layer of work-groups lies past the volume and ends at once)
    int tx, ty, zq;
    if (patched) {
        const int npx = (ntx + GPX - 1) / GPX, npy = (nty + GPY - 1) / GPY;
        const int slot = (int)(blockIdx.x >> 3), patch = (slot / (GPX * GPY)) * 8 + (int)(blockIdx.x & 7), within = slot % (GPX * GPY);
        const int pxy = patch % (npx * npy);
        zq = patch / (npx * npy);
        tx = (pxy / npy) * GPX + within / GPY;
        ty = (pxy % npy) * GPY + within % GPY;
    } else {
        zq = (int)(blockIdx.x % (unsigned)nzq);
        ty = (int)((blockIdx.x / (unsigned)nzq) % (unsigned)nty);
        tx = (int)(blockIdx.x / ((unsigned)nzq * (unsigned)nty));
    }
    const int x0 = xs + tx * GTX, y0 = ty * GTY, z0 = (zq * GWAVES + wv - *zshift) * 64;
    if (tx >= ntx || ty >= nty || zq >= nzq) return;                    // uniform over the WORK-GROUP (barriers below)
    // a wave past the volume still computes its share of tables; so does one whose 64 planes Z can only receive zeros: plane Z gathers from the
    // sinogram planes Z - zc and Z - zc - 1 (zc in [zc_lo, zc_hi] over the projections), and k_sino_zflags marked the planes that hold anything
    bool zlive = z0 >= 0 && z0 < g.nz;
    if (zlive) {
        bool any = false;
        for (int i = z0 - zc_hi - 1 + lane; i <= z0 + 63 - zc_lo; i += 64) any |= i >= 0 && i < g.ndz && zflags[i] != 0;
        zlive = __builtin_amdgcn_ballot_w64(any) != 0;
    }
    // Only the LIVE waves stay: nl of them, this one the rk-th.  They share the weight tables among themselves (a wave that has ended is not
    // waited for by s_barrier); a work-group none of whose z chunks can receive anything ends here (its voxels keep what they hold).
    __shared__ int wlive[GWAVES];
    if (lane == 0) wlive[wv] = zlive ? 1 : 0;
    __syncthreads();
    int nl = 0, rk = 0;
#pragma unroll
    for (int w = 0; w < GWAVES; ++w) { nl += wlive[w]; rk += w < wv ? wlive[w] : 0; }
    // HARDWARE BEHAVIOUR RELIED ON (outside the HIP programming model): on gfx9 / CDNA s_barrier counts only the waves of the
    // work-group that have not terminated, so the live waves may keep calling __syncthreads() after the dead ones returned.
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "k_adj_gather_flat: divergent exit before barriers is only known to be safe on gfx9-family targets (s_barrier ignores ended waves)"
#endif
    if (!zlive) return;
    // a lane is a voxel COLUMN (X, Y) with 64 plane accumulators, except while loading sinogram rows, where it is plane Zl
    const int X = x0 + (lane >> 3), Y = y0 + (lane & 7), Zl = z0 + lane;
    float *wrows = rows[wv];
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const uint32_t pitch4 = (uint32_t)g.ndz * 4u;                     // one projection's sinogram is < 4 GiB (host check)
    const float two_m32 = 2.3283064365386963e-10f;
    f32x2 acc2[32];                                                   // plane pairs (2k, 2k + 1)
#pragma unroll
    for (int p = 0; p < 32; ++p) acc2[p] = f32x2{0.f, 0.f};

    // ---- 1. the weight table of this lane's column for projection IPX -> wtab[GRP % 3][IPX % GWAVES][lane].  The table does not
    //         depend on z: the nl live waves share it, the rk-th computes the projections nl g + rk (one barrier per nl projections).
    //   candidates: rows i0..i0+2, samples j0..j0+NJ-1 (the footprint |dx|,|dy| < 1 maps to |d ix| <= |m00|+|m01| < 1.5: three
    //   consecutive integers cover an interval shorter than 3; likewise |d j| <= |m10|+|m11| < NJ/2); W_k from exact 32.32
    //   positions relative to the voxel
#define G_TABLE(IPX, BUF, SLOT)                                                                                            \
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
                    /* tent on [-1, 1); selects in the SGPR-mask form (select_lanes) */                                    \
                    const float wx = select_lanes2(select_lanes(fx, __builtin_amdgcn_ballot_w64(hx == -1)), 1.f - fx, __builtin_amdgcn_ballot_w64(hx == 0)); \
                    const float wy = select_lanes2(select_lanes(fy, __builtin_amdgcn_ballot_w64(hy == -1)), 1.f - fy, __builtin_amdgcn_ballot_w64(hy == 0)); \
                    wsum += select_lanes(wx * wy, __builtin_amdgcn_ballot_w64((unsigned)(j0 + mth) < (unsigned)ct.n));     \
                    sx += ct.fdx; sy += ct.fdy;                                                                            \
                }                                                                                                          \
                W[k] = select_lanes(wsum, __builtin_amdgcn_ballot_w64((unsigned)(i0 + k) < (unsigned)g.ndx));              \
                rx += ct.fux; ry += ct.fuy;                                                                                \
            }                                                                                                              \
            t4.x = __builtin_bit_cast(float, i0); t4.y = W[0]; t4.z = W[1]; t4.w = W[2];                                   \
        }                                                                                                                  \
        wtab[BUF][SLOT][lane] = t4;                                                                                        \
    }
    // ---- 2a. fetch projection IPX's table entry and ISSUE the 15 loads of the sinogram rows the tile can touch: rows
    //          ix_lo .. ix_lo+13 at this lane's PLANE (coalesced) plus one gather of their values one plane below the wave's
    //          first, from clamped -- always valid -- addresses, masked when used.  Straight-line on purpose (with a branch per
    //          row every row waited for its own round trip to memory).  The loads are consumed one projection later: they fly
    //          while the previous projection accumulates.
    float4 tn;
    int ix_lo_n;
    float y0v[GROWS], yedge = 0.f;                                         // yedge: lane r holds row r one plane below the wave's first
#ifdef TOMO_ABLATE_GATHER_LOADS     /* measurement builds only: the kernel without its sinogram loads (wrong sums) */
#define G_ROW_LOAD(P) __int_as_float(0x3f800000 + (int)(size_t)(P))
#else
#define G_ROW_LOAD(P) (*(const float *)(P))
#endif
#define G_SETUP(IPX, BUF, SLOT)                                                                                            \
    {                                                                                                                      \
        tn = wtab[BUF][SLOT][lane];                                                                                        \
        const int i0s = __builtin_bit_cast(int, tn.x);                                                                     \
        ix_lo_n = __builtin_amdgcn_readfirstlane(wave_min_i32(i0s));                                                       \
        if (zlive) {                                                                                                       \
            const GfC &cn = cs[IPX];                                                                                       \
            const int iz0 = Zl - cn.zc;                                                                                    \
            const char *srow = (const char *)(proj + (size_t)cn.slot * n_det);                          /* wave-uniform */ \
            /* addresses: the projection's base is a wave-uniform SGPR pair (saddr); the 32-bit voffset is the lane's plane     */ \
            /* offset + the row's byte offset.  The 14 clamped row offsets are computed by 14 LANES at once and handed out    */ \
            /* with v_readlane: per row one readlane and one add, no scalar clamp / multiply / 64-bit add (the kernel issued */ \
            /* 335 SALU instructions per projection and wave against 290 VALU -- the scalar unit, one per CU, was the limit) */ \
            const uint32_t o0 = (uint32_t)min(max(iz0, 0), g.ndz - 1) * 4u;                                                 \
            const uint32_t rowoff = (uint32_t)min(max(ix_lo_n + min(lane, GROWS - 1), 0), g.ndx - 1) * pitch4;              \
            _Pragma("unroll") for (int r = 0; r < GROWS; ++r)                                                              \
                y0v[r] = G_ROW_LOAD(srow + (o0 + (uint32_t)__builtin_amdgcn_readlane((int)rowoff, r)));                     \
            /* the plane below (iz0 - 1) is the neighbouring lane's value (DPP shift when used); lane 0 has no neighbour: one  */ \
            /* more load, lane r fetching row r at the wave's first plane - 1 -- 15 loads per projection instead of 28      */ \
            if (cn.tau != 0.f) {                                            /* (tau = 0: the plane below is never used) */ \
                const uint32_t oe = (uint32_t)min(max(z0 - cn.zc - 1, 0), g.ndz - 1) * 4u;                                  \
                yedge = *(const float *)(srow + (rowoff + oe));                                                            \
            }                                                                                                              \
        }                                                                                                                  \
    }
    const int n_grp = (n_proj + nl - 1) / nl;
    int buf = 0;                                                        // table buffer of the current group (three in rotation)
    if (n_grp > 0) {
        G_TABLE(rk, 0, rk)                                              // group 0
        __syncthreads();
        G_SETUP(0, 0, 0)
    }
    for (int grp = 0; grp < n_grp; ++grp) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
        if (grp + 1 < n_grp) G_TABLE((grp + 1) * nl + rk, nbuf, rk)     // next group's tables: a third buffer, nobody reads it yet
        __syncthreads();                                                // ... and everybody is done with group grp - 1's buffer
        const int g0 = grp * nl, g1 = min(n_proj, g0 + nl);
        for (int ip = g0; ip < g1; ++ip) {
            const GfC &c = cs[ip];
            const float4 t = tn;
            const int i0 = __builtin_bit_cast(int, t.x), ix_lo = ix_lo_n;
            const float W0 = t.y, W1 = t.z, W2 = t.w;
            const bool hit = zlive && __any(W0 != 0.f || W1 != 0.f || W2 != 0.f);   // else this projection's rays miss the tile
            // ---- 2b. z-lerp the rows loaded one projection ago into the wave's LDS rows (lane = plane)
            if (hit) {
                const int iz0 = Zl - c.zc, iz1 = iz0 - 1;
                // (no per-row validity test: a row outside the detector was loaded from a clamped, valid address and every lane's
                //  weight for it is 0 (G_TABLE); rows past the last one a lane needs are never read)
                const unsigned long long m0 = __builtin_amdgcn_ballot_w64(iz0 >= 0) & __builtin_amdgcn_ballot_w64(iz0 < g.ndz);
                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(iz1 >= 0) & __builtin_amdgcn_ballot_w64(iz1 < g.ndz);
#define G_ZLERP(MASKED)                                                                                                           \
    _Pragma("unroll") for (int r = 0; r < GROWS; ++r) {                                                                           \
        float y1 = dpp_shr1_f(y0v[r]);                                            /* lane l <- lane l - 1: y(ix, iz0 - 1) for l >= 1 */ \
        asm("v_writelane_b32 %0, %1, 0" : "+v"(y1) : "s"(__builtin_amdgcn_readlane(__builtin_bit_cast(int, yedge), r)));   /* lane 0 <- row r's edge value */ \
        const float a0 = (MASKED) ? select_lanes(y0v[r], m0) : y0v[r], a1 = (MASKED) ? select_lanes(y1, m1) : y1;                \
        wrows[r * GPITCH + lane] = fmaf(c.tau, a1 - a0, a0);                                                                      \
    }
                if (c.tau == 0.f) {                                         // samples sit exactly on detector rows (integer z shift: the nominal geometry
                    if (m0 == ~0ull) {                                      //  before alignment): Yz = y -- no neighbour plane, no lerp (fma(0, a1 - a0, a0) = a0)
#pragma unroll
                        for (int r = 0; r < GROWS; ++r) wrows[r * GPITCH + lane] = y0v[r];            // all 64 planes on the detector: no select either
                    } else {
#pragma unroll
                        for (int r = 0; r < GROWS; ++r) wrows[r * GPITCH + lane] = select_lanes(y0v[r], m0);
                    }
                }
                else if ((m0 & m1) == ~0ull) { G_ZLERP(false) }             // all 64 planes and their lower neighbours on the detector: the usual case
                else { G_ZLERP(true) }
#undef G_ZLERP
            }
            if (ip + 1 < g1) G_SETUP(ip + 1, buf, ip + 1 - g0)
            else if (ip + 1 < n_proj) G_SETUP(ip + 1, nbuf, 0)          // the next group's table is already published
            // ---- 3. accumulate, lane = column: its three rows start at slot0; plane p is an immediate offset.  The LDS rows were
            //         written by this same wave (LDS operations of a wave execute in order), no other wave touches them.
            if (hit) {
                const int slot0 = min(max(i0 - ix_lo, 0), GROWS - 3);       // an 8 x 8 column tile touches <= 13 rows (7 (|cos| + |sin|) + 3); clamped for safety
                // plane QUADS with ds_read_b128 (round 2): 256 B/clk where ds_read_b32 moves 128 B/clk -- the kernel was LDS-bound
                // (SQ_LDS_IDX_ACTIVE = 0.79 of its cycles) on 3 x 64 dword reads per projection
                const float4 *q = (const float4 *)__builtin_assume_aligned(wrows + slot0 * GPITCH, 16);
                // two quads of reads in flight: quad p + 4 is issued before quad p is used (24 temporaries; with one quad in
                // flight a wave had 6 packed FMAs to cover each LDS round trip)
                float4 n0 = q[0], n1 = q[GPITCH / 4], n2 = q[2 * GPITCH / 4];
#pragma unroll
                for (int p = 0; p < 64; p += 4) {
                    const float4 r0 = n0, r1 = n1, r2 = n2;
                    if (p + 4 < 64) { n0 = q[(p + 4) / 4]; n1 = q[(GPITCH + p + 4) / 4]; n2 = q[(2 * GPITCH + p + 4) / 4]; }
                    // plane pairs as packed FMAs (v_pk_fma_f32: two planes per instruction at 0.83 of the scalar rate)
                    acc2[p / 2] = W2 * f32x2{r2.x, r2.y} + (W1 * f32x2{r1.x, r1.y} + (W0 * f32x2{r0.x, r0.y} + acc2[p / 2]));
                    acc2[p / 2 + 1] = W2 * f32x2{r2.z, r2.w} + (W1 * f32x2{r1.z, r1.w} + (W0 * f32x2{r0.z, r0.w} + acc2[p / 2 + 1]));
                    __builtin_amdgcn_sched_barrier(0);                       // keeps the reads from all being hoisted to the top (192 temporaries)
                }
            }
        }
        buf = nbuf;
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
                v.x += acc2[p / 2].x; v.y += acc2[p / 2].y; v.z += acc2[p / 2 + 1].x; v.w += acc2[p / 2 + 1].y;
                *(float4 *)(dst + p) = v;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 64; ++p)
                if (z0 + p < g.nz) dst[p] += (p & 1) ? acc2[p / 2].y : acc2[p / 2].x;
        }
    }
}

