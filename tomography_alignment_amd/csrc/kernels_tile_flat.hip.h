// kernels_tile_flat.hip.h -- the FLAT tile kernels for untilted poses: k_tile_flat<FWD> (one tile per work-group; the adjoint form
// scatters with LDS atomics) and k_fwd_flat_z<NZT> (forward over NZT z-adjacent tiles per work-group, the default forward).
// Needs kernels_tile.hip.h (tile shape, AdjC, helpers) before it; part of the translation unit tomo_project.hip.

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
// Row set-up shared by the flat kernels: for detector row `rix` of an untilted projection, the sample range [jlo, jhi) whose x, y
// cells can fall into the tile's 16 x 16 footprint -- the row's line (tile-relative, sample 0 at (cbx, cby), direction (fdx, fdy))
// clipped against the footprint widened by 2e-2 (conservative float32; exact ownership is decided per sample from the
// fixed-point position).  One row per LANE; the callers broadcast the results with v_readlane.
__device__ __forceinline__ void flat_row_range(float cbx, float cby, float fdx, float fdy, int n, bool row_ok, int &jlo, int &jhi, float xext = (float)ATX)
{
    float t0 = 0.f, t1 = (float)(n - 1);
    if (fdx != 0.f) {
        const float inv = 1.f / fdx, ta = (-2e-2f - cbx) * inv, tb = (xext + 2e-2f - cbx) * inv;
        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
    } else if (cbx < -2e-2f || cbx >= xext + 2e-2f) { t0 = 1.f; t1 = 0.f; }
    if (fdy != 0.f) {
        const float inv = 1.f / fdy, ta = (-2e-2f - cby) * inv, tb = ((float)ATY + 2e-2f - cby) * inv;
        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
    } else if (cby < -2e-2f || cby >= (float)ATY + 2e-2f) { t0 = 1.f; t1 = 0.f; }
    jlo = jhi = 0;
    if (row_ok && t0 <= t1) {
        jlo = max(0, (int)ceilf(t0));
        jhi = min(n, (int)floorf(t1) + 1);
    }
}

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
                int v_jlo, v_jhi;
                {
                    const int rix = ix_lo + r0 + lane;
                    flat_row_range(fp0x + (float)rix * fux, fp0y + (float)rix * fuy, fdx, fdy, c.n, rix <= ix_hi, v_jlo, v_jhi);
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
                        const int cnt = min(64, jhi - jc);
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
#ifndef FZ_WAVES
#define FZ_WAVES 16
#endif
// TX = x width of the tile footprint: 16 (with NZT = 2 z-stacked images: the default) or 32 (with NZT = 1: the "32 x 16 footprint"
// of DESIGN.md section 4, built in round 3 to MEASURE what halving the tile crossings -- hence the float atomics -- costs in the
// sample loop, whose per-entry broadcasts then serve one image instead of two; option fwd_flat_wide).
template <int NZT, int TX = ATX>
__global__ __launch_bounds__(FZ_WAVES * 64) void k_fwd_flat_z(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                              const float *__restrict__ vol, TomoGeomC g, int tile_x0)
{
    // the images of the NZT stacked tiles are INTERLEAVED per (x, y) cell: [x][y][tile][64 planes] -- every corner of every image of a
    // sample then lies within ds_read2st64_b32's offset range (units of 256 B, < 256) of ONE address register
    static_assert((TX + 1) * ALY * NZT * FLZ * 4 <= 160 * 1024, "LDS");
    __shared__ float img[(TX + 1) * ALY * NZT * FLZ];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z0 = -1 + (int)blockIdx.x * (NZT * FTZ), y0 = -1 + (int)blockIdx.y * ATY, x0 = -1 + ((int)blockIdx.z + tile_x0) * TX;
    bool live[NZT];
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < NZT; ++k) {
        bool any_nz = false;
        for (int e = threadIdx.x; e < (TX + 1) * ALY * FLZ; e += FZ_WAVES * 64) {
            const int lz = e % FLZ, t2 = e / FLZ, ly = t2 % ALY, lx = t2 / ALY;
            const int gx = x0 + lx, gy = y0 + ly, gz = z0 + k * FTZ + lz;
            float v = 0.f;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz >= 0 && gz < g.nz) v = vol[((size_t)gx * g.ny + gy) * g.nz + gz];
            img[(t2 * NZT + k) * FLZ + lz] = v;
            any_nz |= (v != 0.f);
        }
        live[k] = __syncthreads_or(any_nz) != 0;                      // an all-zero tile contributes nothing to any ray
        any_live |= live[k];
    }
    if (!any_live) return;
    const float bcx = (float)x0 + 0.5f * TX, bcy = (float)y0 + 0.5f * ATY;
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
        const float ixr = fabsf(m00) * (0.5f * TX) + fabsf(m01) * (0.5f * ATY) + 2e-2f;
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
            int v_jlo, v_jhi;
            {
                const int rix = ix_lo + r0 + lane;
                flat_row_range(fp0x + (float)rix * fux, fp0y + (float)rix * fuy, fdx, fdy, c.n, rix <= ix_hi, v_jlo, v_jhi, (float)TX);
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
                    // one lane per SAMPLE: cell, ownership in x,y and the four x,y weights of sample jc + lane.  The uniform part of
                    // the position (row base + jc steps) is formed on the scalar unit and kept opaque -- the compiler otherwise
                    // folds it into two per-lane 64-bit multiply-adds (v_mad_u64_u32)
                    int64_t ux = rbx + (int64_t)jc * k_fdx, uy = rby + (int64_t)jc * k_fdy;
                    asm volatile("" : "+s"(ux), "+s"(uy));
                    const int64_t px = add64_vs(ldx, ux), py = add64_vs(ldy, uy);
                    const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32);
                    const unsigned long long own_m = (TX == ATY ? __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATY) : (__builtin_amdgcn_ballot_w64(lx < (unsigned)TX) & __builtin_amdgcn_ballot_w64(ly < (unsigned)ATY))) &
                                                     __builtin_amdgcn_ballot_w64(jc + lane < min(jhi, jc + 60));
                    const unsigned t_e = (__umul24(lx, ALY * NZT * FLZ) + __umul24(ly, NZT * FLZ)) * 4u;
                    const float wx = (float)(unsigned)px * two_m32, wy = (float)(unsigned)py * two_m32;
                    const float t_w11 = wx * wy, t_w10 = wx - t_w11, t_w01 = wy - t_w11, t_w00 = 1.f - wx - t_w01;
                    // NO compaction: the owned samples of a row are CONTIGUOUS lanes (a line meets the tile's convex footprint in one
                    // interval of j, and the cells come from exact fixed-point positions), so the sample loop simply broadcasts lanes
                    // first .. first + n_own - 1 with v_readlane.  (Until round 2 the entries were compacted to lanes 0.. with five
                    // ds_permute per row -- an LDS round trip in front of every row's loop; an ablation without the sample loop still took
                    // 71 % of the kernel's time: the row set-up, not the samples, was the cost.)  Lanes that own nothing carry address 0 and
                    // weights 0: the loop's look-ahead may read one or two of them.
                    const unsigned long long om = own_m;
                    const int n_own = (int)__builtin_popcountll(om);
                    const int first = om ? (int)__builtin_ctzll(om) : 0;
                    const int c_e = (int)select_lanes_u(t_e, om);
                    const int c_w00 = __float_as_int(select_lanes(t_w00, om)), c_w01 = __float_as_int(select_lanes(t_w01, om));
                    const int c_w10 = __float_as_int(select_lanes(t_w10, om)), c_w11 = __float_as_int(select_lanes(t_w11, om));
                    f32x2 Sa[NZT], Sb[NZT];
#pragma unroll
                    for (int k = 0; k < NZT; ++k) { Sa[k] = (f32x2){0.f, 0.f}; Sb[k] = (f32x2){0.f, 0.f}; }
                    // an entry's five readlanes serve every image.  One image in use: (y, y + 1) corners arrive as a register pair from
                    // one ds_read2st64 and the weights as SGPR pairs, two packed FMAs per sample.  Both in use: see below.  Which images
                    // take part is decided outside the loop (an all-zero or out-of-range image is skipped).
                    // SOFTWARE-PIPELINED (round 2): the reads of entry jj + 1 are issued before entry jj's values are used.  The
                    // one-entry-per-trip loop drained the LDS queue (s_waitcnt lgkmcnt(0)) before its last FMA, so every entry cost
                    // a wave a full LDS round trip, and with 4 waves per SIMD the kernel sat at 56 % LDS / ~30 % VALU utilisation:
                    // latency-bound.  Entries past n_own exist (ds_permute leaves 0 in lanes nobody wrote: address 0, weights 0), so
                    // the look-ahead needs no guard.  (The images used to lie one behind the other, 73 984 B apart -- beyond the DS offset
                    // fields, a second address register per entry; interleaved per cell, one register reaches all eight corners.)
                    {
                        int r_e = c_e, r_w00 = c_w00, r_w01 = c_w01, r_w10 = c_w10, r_w11 = c_w11;
#define FZ_LOAD(T, J, K0, K1)                                                                                               \
                        {                                                                                                   \
                            const unsigned e_ = (unsigned)__builtin_amdgcn_readlane(r_e, first + (J)) + lane4;                      \
                            const float *q_ = (const float *)((const char *)&img[0] + e_);                                  \
                            _Pragma("unroll") for (int k = (K0); k < (K1); ++k) {                                           \
                                T##v0[k] = (f32x2){q_[k * FLZ], q_[(NZT + k) * FLZ]};                                       \
                                T##v1[k] = (f32x2){q_[(ALY * NZT + k) * FLZ], q_[(ALY * NZT + NZT + k) * FLZ]};             \
                            }                                                                                               \
                        }
#define FZ_USE(T, J, K0, K1)                                                                                                \
                        {                                                                                                   \
                            const f32x2 w0_ = {__int_as_float(__builtin_amdgcn_readlane(r_w00, first + (J))), __int_as_float(__builtin_amdgcn_readlane(r_w01, first + (J)))}; \
                            const f32x2 w1_ = {__int_as_float(__builtin_amdgcn_readlane(r_w10, first + (J))), __int_as_float(__builtin_amdgcn_readlane(r_w11, first + (J)))}; \
                            _Pragma("unroll") for (int k = (K0); k < (K1); ++k) { Sa[k] += w0_ * T##v0[k]; Sb[k] += w1_ * T##v1[k]; } \
                        }
#define FZ_SAMPLE_LOOP(K0, K1)                                                                                              \
                        {                                                                                                   \
                            f32x2 A_v0[NZT], A_v1[NZT], B_v0[NZT], B_v1[NZT];                                               \
                            FZ_LOAD(A_, 0, K0, K1)                                                                          \
                            for (int jj = 0; jj < n_own; jj += 2) {                                                         \
                                FZ_LOAD(B_, jj + 1, K0, K1)                                                                 \
                                FZ_USE(A_, jj, K0, K1)                                                                      \
                                FZ_LOAD(A_, jj + 2, K0, K1)                                                                 \
                                FZ_USE(B_, jj + 1, K0, K1)                                                                  \
                            }                                                                                               \
                        }
                        if (NZT == 2 && zuse[0] && zuse[NZT - 1]) {
                            // both images: a register pair = the SAME corner of image 0 and image 1 (adjacent 256-B units of the interleaved
                            // layout, one ds_read2st64_b32), the weight a scalar for both halves: 4 reads + 4 packed FMAs per sample on one
                            // address register
                            f32x2 Pa = {0.f, 0.f}, Pb = {0.f, 0.f};
                            f32x2 A_00, A_01, A_10, A_11, B_00, B_01, B_10, B_11;
#define FB_LOAD(T, J)                                                                                                      \
                            {                                                                                                   \
                                const float *q_ = (const float *)((const char *)&img[0] + ((unsigned)__builtin_amdgcn_readlane(r_e, first + (J)) + lane4)); \
                                T##00 = (f32x2){q_[0], q_[FLZ]}; T##01 = (f32x2){q_[2 * FLZ], q_[3 * FLZ]};                     \
                                T##10 = (f32x2){q_[2 * ALY * FLZ], q_[(2 * ALY + 1) * FLZ]};                                    \
                                T##11 = (f32x2){q_[(2 * ALY + 2) * FLZ], q_[(2 * ALY + 3) * FLZ]};                              \
                            }
#define FB_USE(T, J)                                                                                                       \
                            {                                                                                                   \
                                Pa += __int_as_float(__builtin_amdgcn_readlane(r_w00, first + (J))) * T##00;                            \
                                Pb += __int_as_float(__builtin_amdgcn_readlane(r_w10, first + (J))) * T##10;                            \
                                Pa += __int_as_float(__builtin_amdgcn_readlane(r_w01, first + (J))) * T##01;                            \
                                Pb += __int_as_float(__builtin_amdgcn_readlane(r_w11, first + (J))) * T##11;                            \
                            }
                            static_assert(NZT <= 2, "the pair layout is written for two images");
                            FB_LOAD(A_, 0)
                            for (int jj = 0; jj < n_own; jj += 2) {
                                FB_LOAD(B_, jj + 1)
                                FB_USE(A_, jj)
                                FB_LOAD(A_, jj + 2)
                                FB_USE(B_, jj + 1)
                            }
#undef FB_LOAD
#undef FB_USE
                            const f32x2 Pt = Pa + Pb;
                            Sa[0] = (f32x2){Pt.x, 0.f}; Sa[NZT - 1] = (f32x2){Sa[NZT - 1].x + (NZT == 2 ? Pt.y : 0.f), 0.f};
                        }
                        else if (zuse[0]) { FZ_SAMPLE_LOOP(0, 1) }
                        else { FZ_SAMPLE_LOOP(NZT - 1, NZT) }
#undef FZ_SAMPLE_LOOP
#undef FZ_USE
#undef FZ_LOAD
                    }
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
// Forward flat kernel, round-3 form: sample table in LDS + image PAIRS.
// Round 2 measured two halves of this separately and each moved the bound to the other unit: (a) the per-row sample table in a
// wave-private LDS table (no v_readlane): 0.523 ms/angle, LDS-bound at 5 + 16 LDS clk per entry; (b) data reads as ds_read_b64
// (half the LDS cycles) with the table still broadcast by five v_readlane per entry: 0.444 vs 0.448, VALU-bound.  Together:
//   * the two z-stacked images are interleaved per PLANE, [x][y][64 planes][image]: the values a lane needs from both images for a
//     corner are one aligned 8-byte word -> 4 ds_read_b64 per entry at immediate offsets of ONE address (y + 1: +512 B, x + 1:
//     +8704 B), 2 LDS clk each, the register pair (image 0, image 1) feeds v_pk_fma_f32 directly;
//   * an entry's four weights come back from a wave-private LDS table by broadcast ds_read_b128 (4 LDS clk), its cell address by one
//     v_readlane: 4 v_pk_fma_f32 + 1 v_add_u32 + 1 v_readlane per entry.
// LDS 12 clk and VALU ~6.3 instructions per entry against 17.6 / ~12 (+ set-up) of k_fwd_flat_z<2>.
// Same sums as k_fwd_flat_z<2>, in the same order per row (entries ascending, four accumulator pairs -> two).
//   * NO HALO PLANE (second half of round 3): an image is 64 owned planes, a work-group owns planes [128 b, 128 b + 128) and all 64 lanes
//     of both images carry a ray.  The plane above a lane's own (the z-lerp's upper neighbour) is the next lane's sum (DPP shift), for
//     lane 63 of image 0 it is image 1's lane 0, for lane 63 of image 1 it belongs to the NEXT work-group in z -- which adds its
//     w_c * S[0] to that ray itself (one single-lane atomic per row, only when the projection's z fraction w_c is not 0: the nominal
//     geometry has w_c = 0).  Why: at 1024 angles per launch (a 4.3 GB sinogram) the memory-side float atomics bound this kernel --
//     0.327 ms/angle with them, 0.273 without (profiles/round3_fwd_tab_variants.md) -- and they are priced per 64-B unit touched
//     (tools/gatomic_scope_bench.hip: a 63-float row at an arbitrary float offset runs at 1.07 TB/s, 64-B aligned at 1.29).  With
//     128-plane work-groups a row instruction is 64 floats starting at ray 128 b + 64 k - floor(p0z): aligned whenever floor(p0z) is a
//     multiple of 16 (0 in the nominal geometry) -- 4 units instead of 4.9 -- and a 1024-plane volume takes 8 work-groups in z, not 9.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// The LIVE-BLOCK LIST of k_fwd_flat_tab.  A block (16 x 16 x 128 voxels + x, y halo) whose voxels are all zero adds nothing to any ray.
// Until round 3 such a work-group found that out itself after staging its block and returned -- but a returning work-group still has to
// be DISPATCHED: it waits for a whole free CU (1024 threads, 157 KB of LDS) like any other, in launch order.  Measured on one MI355X, 1024
// angles per launch: a volume of ones in 768 of 1024 planes (6 of 8 blocks in z live) took 248 ms where the same 6 blocks alone
// (a 1024 x 1024 x 768 volume) take 198 ms; the benchmark's SIRT iterate is such a volume (planes 159 .. 864 of 1024).  So the blocks are
// classified first (k_fwd_live: one small work-group per block, stops at the first non-zero of each image; 5 GB of coalesced reads at
// worst), the live ones compacted in launch order (k_fwd_compact: one work-group, ballot + prefix counts) and the forward's work-group i
// takes block list[1 + i]; work-groups past list[0] return -- they sit at the END of the grid, where nothing queues behind them.
// flags[b]: bit 0 = image 0 (planes z0 .. z0 + 63) holds a non-zero voxel, bit 1 = image 1.  b = zb + nzb * (ty + nty * tx).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fwd_live(const float *__restrict__ vol, TomoGeomC g, int tile_x0, int nzb, int nty, unsigned char *__restrict__ flags)
{
    const int b = (int)blockIdx.x;
    const int zb = b % nzb, ty = (b / nzb) % nty, tx = b / (nzb * nty);
    const int z0 = zb * (2 * FLZ), y0 = -1 + ty * ATY, x0 = -1 + (tx + tile_x0) * ATX;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool in0 = z0 + lane < g.nz, in1 = z0 + FLZ + lane < g.nz;
    bool nz0 = false, nz1 = false;
    for (int c = wv; c < ALX * ALY; c += 4) {                        // a wave reads a column's 2 x 64 planes: 2 x 256 B, coalesced
        const int gx = x0 + c / ALY, gy = y0 + c % ALY;
        if (gx < 0 || gx >= g.nx || gy < 0 || gy >= g.ny) continue;
        const float *col = vol + ((size_t)gx * g.ny + gy) * g.nz + z0 + lane;
        if (in0) nz0 |= col[0] != 0.f;
        if (in1) nz1 |= col[FLZ] != 0.f;
        if (__builtin_amdgcn_ballot_w64(nz0) && __builtin_amdgcn_ballot_w64(nz1)) break;      // wave-uniform: both images known to be live
    }
    const int l0 = __syncthreads_or(nz0), l1 = __syncthreads_or(nz1);
    if (threadIdx.x == 0) flags[b] = (unsigned char)((l0 ? 1 : 0) | (l1 ? 2 : 0));
}

__global__ __launch_bounds__(1024) void k_fwd_compact(const unsigned char *__restrict__ flags, int n, int *__restrict__ list)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        const bool live = i < n && flags[i] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
        if (lane == 0) wsum[wv] = (int)__builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (live) list[1 + off + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;      // launch order kept
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) list[0] = base;
}

#define FT2_TAB 32
#define FT2_TAB_ALLOC (FT2_TAB + 4)      // a wave's table: the owners' entries of one pass (<= FT2_TAB) + padding to a 64-B multiple
__global__ __launch_bounds__(FZ_WAVES * 64) void k_fwd_flat_tab(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                                const float *__restrict__ vol, TomoGeomC g, int tile_x0,
                                                                const int *__restrict__ list, const unsigned char *__restrict__ flags, int nzb, int nty)
{
    __shared__ __attribute__((aligned(16))) float img[ALX * ALY * FLZ * 2];                  // [x][y][plane][image]
    __shared__ float4 tab_w[FZ_WAVES * FT2_TAB_ALLOC];
    static_assert(sizeof(float) * ALX * ALY * FLZ * 2 + 16 * FZ_WAVES * FT2_TAB_ALLOC <= 160 * 1024, "LDS");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // work-group i takes the i-th LIVE block (k_fwd_live / k_fwd_compact above).  Consecutive work-groups go to consecutive XCDs: in list order
    // (z fastest) the work-groups in flight add to neighbouring 512-B pieces of the same sinogram rows, and an empty z range idles no XCD.
    if ((int)blockIdx.x >= list[0]) return;
    const int blk = list[1 + blockIdx.x];
    const int z0 = (blk % nzb) * (2 * FLZ), y0 = -1 + ((blk / nzb) % nty) * ATY, x0 = -1 + (blk / (nzb * nty) + tile_x0) * ATX;
    (void)flags;        // (per-image flags: an all-zero image of a live block yields zero sums, which the row tail does not send)
    for (int e = threadIdx.x; e < ALX * ALY * FLZ * 2; e += FZ_WAVES * 64) {          // e = ((lx * ALY + ly) * FLZ + lz) * 2 + image
        const int k = e & 1, e1 = e >> 1;
        const int lz = e1 % FLZ, t2 = e1 / FLZ, ly = t2 % ALY, lx = t2 / ALY;
        const int gx = x0 + lx, gy = y0 + ly, gz = z0 + k * FLZ + lz;
        float v = 0.f;
        if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny && gz < g.nz) v = vol[((size_t)gx * g.ny + gy) * g.nz + gz];
        img[e] = v;
    }
    __syncthreads();
    const float bcx = (float)x0 + 0.5f * ATX, bcy = (float)y0 + 0.5f * ATY;
    const int64_t orgx = (int64_t)x0 << 32, orgy = (int64_t)y0 << 32;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    const float two_m32 = 2.3283064365386963e-10f;
    float4 *const tw = tab_w + wv * FT2_TAB_ALLOC;
    const lds_cfloat *const img_l = (const lds_cfloat *)img;            // explicit LDS pointer (address space 3): ds_read, not flat loads

    for (int ip = wv; ip < n_proj; ip += FZ_WAVES) {               // one wave owns a whole (tile stack, projection)
        const AdjC &c = pcs[ip];
        const int p0z_i = (int)(c.fp0[2] >> 32);
        const float wcz = (float)(unsigned)c.fp0[2] * two_m32, wfz = 1.f - wcz;
        // Rays and lanes.  The work-group's planes z0 .. z0 + 127 give to the rays izb - 1 .. izb + 127 (izb = z0 - floor(p0z): the ray whose lower
        // plane is plane 0; the ray below it only when the z fraction w_c is not 0).  The row's sums go out as up to THREE atomic instructions
        // over the 64-ray windows that start at w0 = the multiple of 16 at or below the first of those rays, so every instruction is 64-B
        // aligned whatever the projection's z offset (64-B units are what the memory side prices, header).  To that end lane L holds plane
        // (L - t) mod 64 of both images, t = izb - w0 in 0 .. 16: the sums arrive already rotated into window order -- window 0 = lanes >= t of
        // image 0 (+ the ray below in lane t - 1), window 1 = lanes < t of image 0 (its top planes) and lanes >= t of image 1, window 2 =
        // lanes < t of image 1.  t = 0 in the nominal geometry: two full windows, as before.
        const int izb = z0 - p0z_i;
        const bool lerp = wcz != 0.f;
        const int w0 = (lerp ? izb - 1 : izb) & ~15;                   // (two's complement: floors negative rays too)
        const int t = izb - w0;
        const int lane_wrap = (t - 1) & 63;                             // the lane that holds plane 63 of an image
        const unsigned lane8 = (unsigned)((lane - t) & 63) * 8u;
        const int ray0 = w0 + lane;                                     // this lane's ray in window 0; + 64, + 128 in windows 1, 2
        const bool m0 = (lane >= t || (lerp && lane == t - 1)) && ray0 >= 0 && ray0 < g.ndz;
        const bool m1 = ray0 + 64 >= 0 && ray0 + 64 < g.ndz;
        const bool m2 = lane < t && ray0 + 128 >= 0 && ray0 + 128 < g.ndz;
        if (!__builtin_amdgcn_ballot_w64(m0 || m1 || m2)) continue;     // none of its rays is on the detector
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
        int64_t ldx = (int64_t)lane * c.fd[0], ldy = (int64_t)lane * c.fd[1];
        int64_t k_fux = c.fu[0], k_fuy = c.fu[1], k_fdx = c.fd[0], k_fdy = c.fd[1];
        asm volatile("" : "+v"(ldx), "+v"(ldy));                       // see k_tile_flat: keep the row loop's inputs in registers
        asm volatile("" : "+s"(k_fux), "+s"(k_fuy), "+s"(k_fdx), "+s"(k_fdy));
        float *const proj_c = proj + (size_t)c.slot * n_det + ray0;

        for (int r0 = 0; r0 < n_rows_w; r0 += 64) {
            int v_jlo, v_jhi;
            {
                const int rix = ix_lo + r0 + lane;
                flat_row_range(fp0x + (float)rix * fux, fp0y + (float)rix * fuy, fdx, fdy, c.n, rix <= ix_hi, v_jlo, v_jhi);
            }
            const int r_end = min(64, n_rows_w - r0);
            int64_t rbx = c.fp0[0] + (int64_t)(ix_lo + r0) * k_fux - orgx, rby = c.fp0[1] + (int64_t)(ix_lo + r0) * k_fuy - orgy;
            float *pr = proj_c + (size_t)(ix_lo + r0) * g.ndz;
            for (int r = 0; r < r_end; ++r, rbx += k_fux, rby += k_fuy, pr += g.ndz) {
                const int jlo = __builtin_amdgcn_readlane(v_jlo, r), jhi = __builtin_amdgcn_readlane(v_jhi, r);
                if (jhi <= jlo) continue;
                f32x2 Pa = {0.f, 0.f}, Pb = {0.f, 0.f};               // .x: image 0 (lower tile), .y: image 1
                for (int jc = jlo; jc < jhi; jc += FT2_TAB) {
                    int64_t ux = rbx + (int64_t)jc * k_fdx, uy = rby + (int64_t)jc * k_fdy;
                    asm volatile("" : "+s"(ux), "+s"(uy));
                    const int64_t px = sum64_vs(ldx, ux), py = sum64_vs(ldy, uy);
                    const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32);
                    const unsigned long long om = __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATX) &
                                                  __builtin_amdgcn_ballot_w64(lane < min(jhi - jc, FT2_TAB));
                    if (om == 0) continue;
                    const unsigned t_e = (__umul24(lx, ALY * FLZ * 2) + __umul24(ly, FLZ * 2)) * 4u;      // byte offset of cell (lx, ly), plane 0, image 0
                    const float wx = (float)(unsigned)px * two_m32, wy = (float)(unsigned)py * two_m32;
                    const float t_w11 = wx * wy, t_w10 = wx - t_w11, t_w01 = wy - t_w11, t_w00 = 1.f - wx - t_w01;
                    // the owned samples of a row are CONTIGUOUS lanes first .. first + n_own - 1 (see k_fwd_flat_z): entry i of the table =
                    // lane first + i.  LDS operations of one wave execute in order: no barrier between these writes and the reads below,
                    // nor against the previous chunk's.
                    const int n_own = (int)__builtin_popcountll(om);
                    const int first = (int)__builtin_ctzll(om);
                    const unsigned slot = (unsigned)(lane - first);
                    if (slot < (unsigned)n_own) tw[slot] = make_float4(t_w00, t_w01, t_w10, t_w11);      // owners only: no padding entries (see the loops)
                    // The cell ADDRESS of an entry is broadcast with one v_readlane (lanes that own nothing carry address 0), the four weights come
                    // from the wave's LDS table (broadcast ds_read_b128): the data reads of a group do not wait for an LDS round trip of the table
                    // -- they are issued beside the weight reads, which are needed only at the FMAs.  4 + 8 LDS clk per entry.
                    const int c_e = (int)t_e;                       // only the owners' lanes first .. first + n_own - 1 are ever read
                    // Entries in groups of four (four sets of reads in flight per wave), the 0 .. 3 left over one by one: no zero-padding entries.
                    // Variants measured in round 3 on a dense 1024^3 volume, ms per angle (profiles/round3_fwd_tab_variants.md): addresses in the
                    // LDS table too, zero-padded groups of four 0.352; this form with zero-padded groups of four 0.345, of eight 0.388, of
                    // two 0.336, one by one 0.335; groups of four + tail (this code) 0.329; eight + four + tail 0.333; the next group's table
                    // words fetched ahead 0.375; hand software-pipelined with two register sets 0.355.  Round 2's kernel: 0.434.
#define FT2_ENTRY(J, W)                                                                                                    \
                        {                                                                                                   \
                            /* explicit LDS pointer; volatile keeps four ds_read_b64 (2 LDS clk each): merged into ds_read2st64_b64 */ \
                            /* they cost 8 clk per pair (MI355X_MICROARCH.md, LDS table)                                           */ \
                            typedef __attribute__((address_space(3))) const volatile f32x2 lds_v2;                          \
                            const unsigned e_ = (unsigned)__builtin_amdgcn_readlane(c_e, first + j4 + (J));                 \
                            const __attribute__((address_space(3))) char *q_ = (const __attribute__((address_space(3))) char *)img_l + (e_ + lane8); \
                            const f32x2 v00 = *(lds_v2 *)(q_), v01 = *(lds_v2 *)(q_ + FLZ * 8);                             \
                            const f32x2 v10 = *(lds_v2 *)(q_ + ALY * FLZ * 8), v11 = *(lds_v2 *)(q_ + (ALY + 1) * FLZ * 8); \
                            /* weight = one half of an aligned register pair, broadcast to both images by op_sel (the compiler   */ \
                            /* copies W.w to a fresh register first: one v_mov per entry)                                          */ \
                            FT2_FMAS(J, W)                                                                                  \
                        }
                    // (round 3 also measured the weights broadcast with v_readlane instead of read from the table, wholly or by halves:
                    //  slower, profiles/round3_fwd_tab_variants.md, profiles/round3_fwd_weight_source_ab.log -- those variants are in git history only)
#define FT2_FMAS(J, W)                                                                                                     \
                            const f32x2 w01_ = {(W).x, (W).y}, w23_ = {(W).z, (W).w};                                       \
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(Pa) : "v"(w01_), "v"(v00));          \
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(Pb) : "v"(w01_), "v"(v01));             \
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(Pa) : "v"(w23_), "v"(v10));          \
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(Pb) : "v"(w23_), "v"(v11));
#define FT2_W(J) tw[j4 + (J)]
                    typedef float4 ft2_w_t;
                    int j4 = 0;
                    for (; j4 + 4 <= n_own; j4 += 4) {                                         // wave-uniform; four entries in flight, no padding
                        const ft2_w_t wa = FT2_W(0), wb = FT2_W(1), wc = FT2_W(2), wd = FT2_W(3);
                        FT2_ENTRY(0, wa) FT2_ENTRY(1, wb) FT2_ENTRY(2, wc) FT2_ENTRY(3, wd)
                    }
                    for (; j4 < n_own; j4 += 1) {                                              // the 0 .. 3 entries left
                        const ft2_w_t wa = FT2_W(0);
                        FT2_ENTRY(0, wa)
                    }
#undef FT2_ENTRY
#undef FT2_FMAS
#undef FT2_W
                }
                const f32x2 Pt = Pa + Pb;
                const float s0 = Pt.x, s1 = Pt.y;                              // image 0 / image 1, plane (lane - t) mod 64
                // the plane above: the next lane's (DPP wave rotate: a VALU move, not a ds_bpermute round trip) -- except above plane 63, where
                // it is the next image's plane 0 (image 1's for image 0; the next work-group's for image 1: that one adds its share itself,
                // as this one does for the ray below its plane 0 in window 0's lane t - 1)
                const float r0n = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0x134, 0xf, 0xf, false));       // wave_rol:1: lane l <- lane l + 1, lane 63 <- lane 0
                const float r1n = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0x134, 0xf, 0xf, false));
                const bool wrap = lane == lane_wrap;
                const float val0 = wfz * s0 + wcz * (wrap ? r1n : r0n);
                const float val1 = wfz * s1 + wcz * (wrap ? 0.f : r1n);
                // lanes whose value is 0 (rays that crossed only zero voxels, all-zero images) do not take part; an empty mask skips the instruction
                const float o0 = lane >= t ? val0 : wcz * r0n, o1 = lane < t ? val0 : val1;
#ifdef TOMO_ABLATE_FWD_ATOMICS          // measurement builds only (tools/gpu_r3l.sh): what the kernel costs without its atomics
                asm volatile("" :: "v"(o0), "v"(o1), "v"(val1));
#else
                if (m0 && o0 != 0.f) atomicAdd(pr, o0);
                if (m1 && o1 != 0.f) atomicAdd(pr + 64, o1);
                if (m2 && val1 != 0.f) atomicAdd(pr + 128, val1);
#endif
            }
        }
    }
}
