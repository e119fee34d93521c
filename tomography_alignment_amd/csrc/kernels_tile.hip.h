// kernels_tile.hip.h -- the LDS tile kernels (general and flat, forward and adjoint) and the gather-form flat adjoint
// Part of the single translation unit tomo_project.hip (included there, in this order: kernels_ray, kernels_tile,
// kernels_grad); not compiled on its own.

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
typedef __attribute__((address_space(3))) float lds_cfloat;
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
    __shared__ int acc[ALX * ALY * ALZ + 4];        // + pad: a masked-out lane may read one dword past the last row (its value is discarded)
    const lds_cfloat *img3 = (const lds_cfloat *)acc;       // explicit LDS pointer: offsets made opaque below must still give ds_read
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
                                const unsigned long long own = __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATX) & __builtin_amdgcn_ballot_w64(lz < (unsigned)ATZ);
                                // Lanes that do not own the sample still read (and discard): from a cell clamped into the image whose LDS
                                // bank is the one their own z would have -- lanes sit on consecutive z, so the 32 lanes of a bank group keep
                                // 32 different banks.  (They used to read word 0: every such lane then hit bank 0 together with whichever
                                // owner lane mapped there -- SQ_LDS_BANK_CONFLICT was 47 % of the kernel's LDS cycles, LDS 81 % busy.)
                                const unsigned cx = min(lx, (unsigned)(ATX - 1)), cy = min(ly, (unsigned)(ATY - 1)), cz = lz & 63u;
                                const unsigned eb = (((cx << 4) + cx + cy) << 8) + (cz << 2);          // byte offset of cell (cx, cy, cz): (cx * ALY + cy) * ALZ + cz, no quarter-rate multiply
                                const float wx = (float)(unsigned)px * two_m32, wy = (float)(unsigned)py * two_m32, wz = (float)(unsigned)pz * two_m32;
                                unsigned eb1 = eb + ALY * ALZ * 4;
                                asm("" : "+v"(eb1));                      // one add for the x+1 face; its y+1 rows sit within ds_read2's offset range
                                const lds_cfloat *q = (const lds_cfloat *)((const __attribute__((address_space(3))) char *)img3 + eb);
                                const lds_cfloat *q1 = (const lds_cfloat *)((const __attribute__((address_space(3))) char *)img3 + eb1);
                                const f32x2 p00 = {q[0], q[1]}, p01 = {q[ALZ], q[ALZ + 1]};
                                const f32x2 p10 = {q1[0], q1[1]}, p11 = {q1[ALZ], q1[ALZ + 1]};
                                part += select_lanes(trilerp_pairs(p00, p01, p10, p11, wx, wy, wz), own);
                                px = add64_vs(px, c.fd[0]); py = add64_vs(py, c.fd[1]); pz = add64_vs(pz, c.fd[2]);
                            }                                          // (two samples per trip, 8 reads in flight: measured no faster -- the loop is VALU-issue bound)
                            if (lane_ok) atomicAdd(pr, part);          // 64 consecutive floats per wave: the full-rate atomic shape
                        } else {
                            const float ys = (lane_ok ? *pr : 0.f) * scale;
                            // straight-line body: a lane that does not own the sample adds integer 0 to a cell clamped into the image on
                            // its own z bank (see the forward) -- no exec-mask switches between samples (two s_cbranch_execz per sample
                            // before: 1.52 -> 1.41 ms/angle at 1024^3)
                            f32x2 ksc = {-two_m32, two_m32}, kof = {1.f, 0.f};
                            asm("" : "+v"(kof));                        // held in a VGPR pair (else re-materialised per sample)
                            for (int jj = 0; jj < cnt; ++jj) {
                                const unsigned lx = (unsigned)(px >> 32), ly = (unsigned)(py >> 32), lz = (unsigned)(pz >> 32);
                                const float yo = select_lanes(ys, __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATX) & __builtin_amdgcn_ballot_w64(lz < (unsigned)ATZ));
                                const unsigned cx = min(lx, (unsigned)(ATX - 1)), cy = min(ly, (unsigned)(ATY - 1)), cz = lz & 63u;
                                // (1 - w, w) per axis as one packed pair, products as packed multiplies: 3 + 7 packed instructions for what
                                // were 6 + 14 scalar ones (same operations in the same order)
                                const float fx = (float)(unsigned)px, fy = (float)(unsigned)py, fz = (float)(unsigned)pz;
                                const f32x2 wxp = f32x2{fx, fx} * ksc + kof, wyp = f32x2{fy, fy} * ksc + kof, wzp = f32x2{fz, fz} * ksc + kof;
                                const f32x2 a = yo * wxp;
                                const f32x2 b0 = a.x * wyp, b1 = a.y * wyp;
                                const f32x2 c00 = b0.x * wzp, c01 = b0.y * wzp, c10 = b1.x * wzp, c11 = b1.y * wzp;
                                int *q = &acc[(((cx << 4) + cx + cy) << 6) + cz];
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
// Row set-up shared by the flat kernels: for detector row `rix` of an untilted projection, the sample range [jlo, jhi) whose x, y
// cells can fall into the tile's 16 x 16 footprint -- the row's line (tile-relative, sample 0 at (cbx, cby), direction (fdx, fdy))
// clipped against the footprint widened by 2e-2 (conservative float32; exact ownership is decided per sample from the
// fixed-point position).  One row per LANE; the callers broadcast the results with v_readlane.
__device__ __forceinline__ void flat_row_range(float cbx, float cby, float fdx, float fdy, int n, bool row_ok, int &jlo, int &jhi)
{
    float t0 = 0.f, t1 = (float)(n - 1);
    if (fdx != 0.f) {
        const float inv = 1.f / fdx, ta = (-2e-2f - cbx) * inv, tb = ((float)ATX + 2e-2f - cbx) * inv;
        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
    } else if (cbx < -2e-2f || cbx >= (float)ATX + 2e-2f) { t0 = 1.f; t1 = 0.f; }
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
#define FZ_WAVES 16
template <int NZT>
__global__ __launch_bounds__(FZ_WAVES * 64) void k_fwd_flat_z(const AdjC *__restrict__ pcs, int n_proj, float *__restrict__ proj,
                                                              const float *__restrict__ vol, TomoGeomC g, int tile_x0)
{
    // the images of the NZT stacked tiles are INTERLEAVED per (x, y) cell: [x][y][tile][64 planes] -- every corner of every image of a
    // sample then lies within ds_read2st64_b32's offset range (units of 256 B, < 256) of ONE address register
    __shared__ float img[ALX * ALY * NZT * FLZ];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z0 = -1 + (int)blockIdx.x * (NZT * FTZ), y0 = -1 + (int)blockIdx.y * ATY, x0 = -1 + ((int)blockIdx.z + tile_x0) * ATX;
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
            img[(t2 * NZT + k) * FLZ + lz] = v;
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
                    const unsigned long long own_m = __builtin_amdgcn_ballot_w64((lx | ly) < (unsigned)ATX) & __builtin_amdgcn_ballot_w64(jc + lane < min(jhi, jc + 60));
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
#define GWAVES 4
#define GPX 8              // (x, y) tile patch that one XCD's resident work-groups cover together
#define GPY 12

struct GfC {
    int64_t fp0x, fp0y, fux, fuy, fdx, fdy;   // x, y of the 32.32 lattice  p = fp0 + ix fu + j fd
    float m00, m01, m10, m11;                 // (ix, j) = M ((x, y) - p0)
    float p0x, p0y, tau;
    int32_t n, zc, slot;
};

template <int NJ>      // samples per row that can reach a column: 3 for step >= 0.95 voxel, 6 for step >= 0.475
__global__ __launch_bounds__(GWAVES * 64, 4) void k_adj_gather_flat(const GfC *__restrict__ cs, int n_proj, const float *__restrict__ proj,
                                                                 float *__restrict__ vol, TomoGeomC g, int xs, int xe, int patched)
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
    const int ntx = (xe - xs + GTX - 1) / GTX, nty = (g.ny + GTY - 1) / GTY, nzq = (g.nz + 64 * GWAVES - 1) / (64 * GWAVES);
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
    const int x0 = xs + tx * GTX, y0 = ty * GTY, z0 = (zq * GWAVES + wv) * 64;
    if (tx >= ntx || ty >= nty || zq >= nzq) return;                    // uniform over the WORK-GROUP (barriers below)
    const bool zlive = z0 < g.nz;                                       // a wave past the volume still computes its share of tables
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
        wtab[((IPX) / GWAVES) % 3][(IPX) % GWAVES][lane] = t4;                                                             \
    }
    // ---- 2a. fetch projection IPX's table entry and ISSUE the 15 loads of the sinogram rows the tile can touch: rows
    //          ix_lo .. ix_lo+13 at this lane's PLANE (coalesced) plus one gather of their values one plane below the wave's
    //          first, from clamped -- always valid -- addresses, masked when used.  Straight-line on purpose (with a branch per
    //          row every row waited for its own round trip to memory).  The loads are consumed one projection later: they fly
    //          while the previous projection accumulates.
    float4 tn;
    int ix_lo_n;
    float y0v[GROWS], yedge;                                           // yedge: lane r holds row r one plane below the wave's first
#define G_SETUP(IPX)                                                                                                       \
    {                                                                                                                      \
        tn = wtab[((IPX) / GWAVES) % 3][(IPX) % GWAVES][lane];                                                             \
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
                y0v[r] = *(const float *)(srow + (o0 + (uint32_t)__builtin_amdgcn_readlane((int)rowoff, r)));              \
            /* the plane below (iz0 - 1) is the neighbouring lane's value (DPP shift when used); lane 0 has no neighbour: one  */ \
            /* more load, lane r fetching row r at the wave's first plane - 1 -- 15 loads per projection instead of 28      */ \
            const uint32_t oe = (uint32_t)min(max(z0 - cn.zc - 1, 0), g.ndz - 1) * 4u;                                      \
            yedge = *(const float *)(srow + (rowoff + oe));                                                                \
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
#pragma unroll                                                              //  before alignment): Yz = y -- no neighbour plane, no lerp (fma(0, a1 - a0, a0) = a0)
                    for (int r = 0; r < GROWS; ++r) wrows[r * GPITCH + lane] = select_lanes(y0v[r], m0);
                }
                else if ((m0 & m1) == ~0ull) { G_ZLERP(false) }             // all 64 planes and their lower neighbours on the detector: the usual case
                else { G_ZLERP(true) }
#undef G_ZLERP
            }
            if (ip + 1 < n_proj) G_SETUP(ip + 1)                        // the next group's table is already published
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

