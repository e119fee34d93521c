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
//
// The kernels live in five included files (one translation unit): kernels_ray.hip.h (k_pad/k_unpad, k_fwd_v1/v2, k_adj_v1,
// k_bp_voxel), kernels_tile.hip.h (tile shape, AdjC, helpers, k_absmax, the general k_tile), kernels_tile_flat.hip.h (k_tile_flat,
// k_fwd_flat_z), kernels_tile_gather.hip.h (k_adj_gather_flat), kernels_grad.hip.h (k_proj_grad, k_proj_grad_v2, k_proj_grad_v3);
// this file holds the host side of the C-ABI and the small-N kernels (k_triplets, k_vox_splat).
#include <limits.h>
#include <string.h>

#include <algorithm>
#include <climits>
#include <vector>

#include "tomo_ctx.h"

#include "kernels_ray.hip.h"
#include "kernels_tile.hip.h"
#include "kernels_tile_flat.hip.h"
#include "kernels_tile_gather.hip.h"
#include "kernels_grad.hip.h"

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
#define TOMO_RED_PART_BYTES ((size_t)512 << 20)

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
    int zc_lo = INT_MAX, zc_hi = INT_MIN;          // range of the gather projections' integer z offsets (k_sino_zflags / zlive in the kernel)
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
            zc_lo = std::min(zc_lo, (int)q.zc);
            zc_hi = std::max(zc_hi, (int)q.zc);
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
    ctx->tile_cache_zc_lo = zc_lo;
    ctx->tile_cache_zc_hi = zc_hi;
    ctx->tile_cache_valid = true;
    return TOMO_OK;
}

static inline dim3 tile_grid(const TomoGeomC &g, int tz = ATZ) { return dim3((g.nz + 1 + tz - 1) / tz, (g.ny + 1 + ATY - 1) / ATY, (g.nx + 1 + ATX - 1) / ATX); }

// Tile forward restricted to the x tile columns [xt0, xt1): ADDS the partial ray sums of those tiles into d_proj (the caller
// zeroes it).  *done = false when the poses do not qualify for the tile kernels (nothing launched).
static int forward_tiles(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_vol, float *d_proj, int xt0, int xt1, bool *done)
{
    const TomoGeomC &g = ctx->g;
    *done = false;
    if (ctx->fwd_variant != 3 || n_proj <= 0) return TOMO_OK;
    dim3 grid = tile_grid(g);
    if (grid.y > 65535 || grid.z > 65535) return TOMO_OK;
    bool ok = false;
    double wb = 2.0;
    int n_flat = 0;
    int rc = stage_tile_consts(ctx, h_poses, n_proj, &ok, &wb, &n_flat);
    if (rc) return rc;
    if (!ok) return TOMO_OK;
    *done = true;
    xt0 = std::max(xt0, 0);
    xt1 = std::min(xt1, (int)grid.z);
    if (xt1 <= xt0) return TOMO_OK;
    grid.z = (unsigned)(xt1 - xt0);
    const AdjC *d_c = (const AdjC *)ctx->d_stage;
    ctx->fwd_blk_flat_ints = 0;
    ctx->zf_src = nullptr;                  // the block lists below overwrite the back-projection's cached plane flags in d_blk
    if (n_flat > 0) {
        dim3 fg = tile_grid(g, FTZ);
        fg.z = grid.z;
#ifdef TOMO_MEASUREMENT_VARIANTS
        if (ctx->fwd_flat_wide && xt0 == 0 && xt1 == (int)tile_grid(g).z)    // measurement variant: 32 x 16 footprint, one image (whole-volume calls only)
            TOMO_LAUNCH(ctx, "k_fwd_tile_flat", (k_fwd_flat_z<1, 32>), dim3(fg.x, fg.y, (g.nx + 1 + 31) / 32), dim3(FZ_WAVES * 64), 0, d_c, n_flat,
                        d_proj, d_vol, g, 0);
        else
#endif
        if (fg.x >= 2 && ctx->fwd_flat_ztiles >= 2 && ctx->fwd_flat_tab) {    // round 3: LDS sample table + image pairs, live blocks only
            const int nzb = (g.nz + 2 * FLZ - 1) / (2 * FLZ), nty = (int)fg.y;
            const size_t n_blk = (size_t)nzb * nty * fg.z;
            if (n_blk >= (size_t)1 << 30) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_forward: volume too large for the flat forward's block list");
            ctx->fwd_blk_flat_ints = 1 + n_blk + (n_blk + 3) / 4;       // the general kernel's tile list of the same call goes behind it
            rc = tomo_ensure_blk(ctx, ctx->fwd_blk_flat_ints + (n_proj > n_flat ? 1 + (size_t)grid.x * grid.y * grid.z * 5 / 4 + 1 : 0));
            if (rc) return rc;
            int *d_list = ctx->d_blk;
            unsigned char *d_flags = (unsigned char *)(ctx->d_blk + 1 + n_blk);
            TOMO_LAUNCH(ctx, "k_fwd_live", k_fwd_live, dim3((unsigned)n_blk), dim3(256), 0, d_vol, g, xt0, nzb, nty, d_flags);
            TOMO_LAUNCH(ctx, "k_fwd_live", k_fwd_compact, dim3(1), dim3(1024), 0, d_flags, (int)n_blk, d_list);
            TOMO_LAUNCH(ctx, "k_fwd_tile_flat", k_fwd_flat_tab, dim3((unsigned)n_blk), dim3(FZ_WAVES * 64), 0, d_c, n_flat, d_proj, d_vol, g, xt0,
                        (const int *)d_list, (const unsigned char *)d_flags, nzb, nty);
        }
        else if (fg.x >= 2 && ctx->fwd_flat_ztiles >= 2)         // two z-adjacent tiles per work-group share the per-row set-up
            TOMO_LAUNCH(ctx, "k_fwd_tile_flat", k_fwd_flat_z<2>, dim3((fg.x + 1) / 2, fg.y, fg.z), dim3(FZ_WAVES * 64), 0, d_c, n_flat,
                        d_proj, d_vol, g, xt0);
        else
            TOMO_LAUNCH(ctx, "k_fwd_tile_flat", k_tile_flat<true>, fg, dim3(ADJ_WAVES * 64), 0, d_c, n_flat, d_proj,
                        (float *)d_vol, g, (const unsigned *)nullptr, 1.f, xt0);
    }
    if (n_proj > n_flat) {
        // the general kernel over the LIVE tiles only (kernels_tile.hip.h, k_tile_live); its list sits behind the flat forward's in d_blk
        const size_t n_tile = (size_t)grid.x * grid.y * grid.z;
        const size_t flat_ints = ctx->fwd_blk_flat_ints;
        if (n_tile >= (size_t)1 << 30) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_forward: volume too large for the tile list");
        rc = tomo_ensure_blk(ctx, flat_ints + 1 + n_tile + (n_tile + 3) / 4);
        if (rc) return rc;
        int *t_list = ctx->d_blk + flat_ints;
        unsigned char *t_flags = (unsigned char *)(t_list + 1 + n_tile);
        TOMO_LAUNCH(ctx, "k_fwd_live", k_tile_live, dim3((unsigned)n_tile), dim3(256), 0, d_vol, g, xt0, (int)grid.x, (int)grid.y, t_flags);
        TOMO_LAUNCH(ctx, "k_fwd_live", k_fwd_compact, dim3(1), dim3(1024), 0, (const unsigned char *)t_flags, (int)n_tile, t_list);
        TOMO_LAUNCH(ctx, "k_fwd_tile", (k_tile<true, TILE_FWD_WAVES>), dim3((unsigned)n_tile), dim3(TILE_FWD_WAVES * 64), 0, d_c + n_flat, n_proj - n_flat, d_proj, (float *)d_vol,
                    g, (const unsigned *)nullptr, 1.f, xt0, (const int *)t_list, (int)grid.x, (int)grid.y, (const int *)nullptr);
    }
    return TOMO_OK;
}

extern "C" int tomo_forward(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_vol, float *d_proj)
{
    TomoRange roctx_range("tomo_forward");
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_vol || !d_proj || n_proj < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_forward: bad args");
    if (n_proj == 0) return TOMO_OK;
    const TomoGeomC &g = ctx->g;
    const size_t n_det = (size_t)g.ndx * g.ndz;
    int rc;
    if (ctx->fwd_variant == 3) {
        // the tile kernels add into d_proj: zero it first; if the poses turn out not to qualify, the ray-driven path overwrites anyway
        TOMO_HIP(ctx, hipMemsetAsync(d_proj, 0, n_det * (size_t)n_proj * sizeof(float), ctx->stream));
        bool done = false;
        rc = forward_tiles(ctx, h_poses, n_proj, d_vol, d_proj, 0, INT_MAX, &done);
        if (rc) return rc;
        if (done) return TOMO_OK;
    }
    rc = stage_volume(ctx, d_vol);
    if (rc) return rc;
    for (int p0 = 0; p0 < n_proj; p0 += TOMO_MAX_GRID_Z) {
        const int np = std::min(TOMO_MAX_GRID_Z, n_proj - p0);
        ProjC *d_pc = nullptr;
        rc = upload_projc(ctx, h_poses + (size_t)p0 * TOMO_POSE_STRIDE, np, false, &d_pc, nullptr);
        if (rc) return rc;
        float *out = d_proj + (size_t)p0 * n_det;
        if (ctx->fwd_variant == 1 || ctx->wide_rows)      // wide rows: 24-bit offset multiplies would truncate (tomo_check_geometry)
            TOMO_LAUNCH(ctx, "k_fwd_v1", k_fwd_v1, ray_grid(g, np), dim3(256), 0, d_pc, ctx->d_volpad, out, g);
        else
            TOMO_LAUNCH(ctx, "k_fwd_v2", k_fwd_v2, ray_grid(g, np), dim3(256), 0, d_pc, ctx->d_volpad, out, g);
    }
    return TOMO_OK;
}

// x-slab form of the forward projection, the counterpart of tomo_adjoint_xslab: the tile columns [xt0, xt1) read only the voxels
// x in [ATX*xt0 - 1, ATX*xt1] (clipped to the volume), so a sharded solver can start them as soon as that range of the volume is
// final while the all-reduces of the other slabs are still in flight.  ADDS into d_proj (zero it before the first slab).
extern "C" int tomo_forward_xslab(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_vol, float *d_proj, int xt0, int xt1)
{
    TomoRange roctx_range("tomo_forward_xslab");
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !d_vol || !d_proj || n_proj < 0 || xt0 < 0 || xt1 < xt0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_forward_xslab: bad args");
    bool done = false;
    int rc = forward_tiles(ctx, h_poses, n_proj, d_vol, d_proj, xt0, xt1, &done);
    if (rc) return rc;
    if (!done) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_forward_xslab: these poses do not take the tile kernels");
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
    // Which detector-z planes of this call's sinogram hold a non-zero value at all (any projection, any row)?  A residual sinogram is exactly
    // zero wherever the rays miss the object's support: a wave of the gather kernel whose 64 voxel planes can only receive from all-zero
    // planes skips its loads and its plane loop (zlive), the general kernel skips a (tile, projection) whose rays all lie in such planes
    // (zcum: prefix counts of the flags).  One coalesced pass over the sinogram (0.8 ms for 4.3 GB); flags and counts live in d_blk.
    unsigned char *d_zf = nullptr;
    int *d_zcum = nullptr, *d_zshift = nullptr;
    if (n_gather > 0 || n_proj > n_flat) {
        const size_t zf_ints = ((size_t)g.ndz + 3) / 4;
        const size_t n_tile_all = (size_t)grid.x * grid.y * grid.z;
        rc = tomo_ensure_blk(ctx, zf_ints + (size_t)g.ndz + 2 + (n_proj > n_flat ? 1 + n_tile_all + (n_tile_all + 3) / 4 : 0));
        if (rc) return rc;
        d_zf = (unsigned char *)ctx->d_blk;
        d_zcum = ctx->d_blk + zf_ints;
        d_zshift = d_zcum + g.ndz + 1;
        const bool want_cum = n_proj > n_flat;
        // (option "reuse_sino_flags": the x-slab calls of one back-projection pass scan the sinogram once, not once per slab)
        // The key holds everything the cached words depend on: the sinogram and its shape (layout of d_zf / d_zcum), and the
        // volume height and the poses' integer z offsets the chunk shift was computed from.
        const bool cached = ctx->reuse_sino_flags && ctx->zf_src == (const void *)d_proj && ctx->zf_nproj == n_proj && (ctx->zf_has_cum || !want_cum) &&
                            (ctx->zf_has_shift || n_gather == 0) && ctx->zf_ndz == g.ndz && ctx->zf_ndx == g.ndx && ctx->zf_nz == g.nz &&
                            ctx->zf_zc_lo == ctx->tile_cache_zc_lo && ctx->zf_zc_hi == ctx->tile_cache_zc_hi;
        if (!cached) {
            TOMO_HIP(ctx, hipMemsetAsync(d_zf, 0, (size_t)g.ndz, ctx->stream));
            const long long n_rows = (long long)n_proj * g.ndx;
            const unsigned gy = (unsigned)std::min<long long>(n_rows, 4096);
            TOMO_LAUNCH(ctx, "k_sino_zflags", k_sino_zflags, dim3((unsigned)((g.ndz + 255) / 256), gy), dim3(256), 0, d_proj, n_rows, g.ndz, d_zf);
            if (want_cum) TOMO_LAUNCH(ctx, "k_sino_zflags", k_zflags_prefix, dim3(1), dim3(1024), 0, (const unsigned char *)d_zf, g.ndz, d_zcum);
            if (n_gather > 0)
                TOMO_LAUNCH(ctx, "k_sino_zflags", k_zchunk_shift, dim3(1), dim3(64), 0, (const unsigned char *)d_zf, g.ndz, g.nz, ctx->tile_cache_zc_lo,
                            ctx->tile_cache_zc_hi, d_zshift);
            ctx->zf_src = (const void *)d_proj;
            ctx->zf_nproj = n_proj;
            ctx->zf_has_cum = want_cum;
            ctx->zf_has_shift = n_gather > 0;
            ctx->zf_ndz = g.ndz; ctx->zf_ndx = g.ndx; ctx->zf_nz = g.nz;
            ctx->zf_zc_lo = ctx->tile_cache_zc_lo; ctx->zf_zc_hi = ctx->tile_cache_zc_hi;
        }
    }
    if (n_gather > 0) {
        // the voxel x range the tile columns [xt0, xt1) finalise (the tile grid starts at x = -1)
        const int xs = std::max(0, ATX * xt0 - 1), xe = (xt1 == n_xt) ? g.nx : std::min(g.nx, ATX * xt1 - 1);
        if (xe > xs) {
            // one-dimensional grid of patches (see the kernel): ceil(#patches / 8) groups of 8 patches x GPX*GPY tiles
            const int ntx = (xe - xs + GTX - 1) / GTX, nty = (g.ny + GTY - 1) / GTY, nzq = (g.nz + 64 * GWAVES - 1) / (64 * GWAVES) + 1;   // + 1: see the kernel
            const long long n_patch = (long long)nzq * ((ntx + GPX - 1) / GPX) * ((nty + GPY - 1) / GPY);
            const int patched = n_patch >= 64 ? 1 : 0;       // (256 until round 3: the sharded solver's x slabs -- 88 .. 176 patches at 1024^3 -- ran unpatched, 10 % slower)
            const long long n_wg = patched ? ((n_patch + 7) / 8) * 8 * (GPX * GPY) : (long long)nzq * ntx * nty;
            if (n_wg >= ((long long)1 << 31)) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "gather adjoint: grid too large");
            const dim3 ggrid((unsigned)n_wg);
            const int zc_lo = ctx->tile_cache_zc_lo, zc_hi = ctx->tile_cache_zc_hi;
            if (ctx->tile_cache_eb_max < 1.49)
                TOMO_LAUNCH(ctx, "k_adj_gather_flat", k_adj_gather_flat<3>, ggrid, dim3(GWAVES * 64), 0, d_gfc, n_gather, d_proj, d_vol, g, xs, xe, patched,
                            (const unsigned char *)d_zf, zc_lo, zc_hi, (const int *)d_zshift);
            else        // finer sampling along the rays (step down to ~0.475 voxel): six samples per row can reach a column
                TOMO_LAUNCH(ctx, "k_adj_gather_flat", k_adj_gather_flat<6>, ggrid, dim3(GWAVES * 64), 0, d_gfc, n_gather, d_proj, d_vol, g, xs, xe, patched,
                            (const unsigned char *)d_zf, zc_lo, zc_hi, (const int *)d_zshift);
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
    if (n_proj > n_flat) {
        // the tiles some projection can bring something to, compacted in launch order (kernels_tile.hip.h: k_tile_adj_live); list and flags
        // sit behind the plane flags and their prefix counts in d_blk
        const size_t n_tile = (size_t)grid.x * grid.y * grid.z;
        if (n_tile >= (size_t)1 << 30) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_adjoint: volume too large for the tile list");
        const size_t head = ((size_t)g.ndz + 3) / 4 + (size_t)g.ndz + 2;
        rc = tomo_ensure_blk(ctx, head + 1 + n_tile + (n_tile + 3) / 4);      // (grows at most once per geometry: d_zf / d_zcum are re-derived below)
        if (rc) return rc;
        if ((unsigned char *)ctx->d_blk != d_zf) return tomo_fail(ctx, TOMO_ERR_STATE, "tomo_adjoint: block buffer moved");
        int *t_list = ctx->d_blk + head;
        unsigned char *t_flags = (unsigned char *)(t_list + 1 + n_tile);
        TOMO_LAUNCH(ctx, "k_sino_zflags", k_tile_adj_live, dim3((unsigned)n_tile), dim3(64), 0, d_c + n_flat, n_proj - n_flat, g, xt0, (int)grid.x,
                    (int)grid.y, (const int *)d_zcum, t_flags);
        TOMO_LAUNCH(ctx, "k_sino_zflags", k_fwd_compact, dim3(1), dim3(1024), 0, (const unsigned char *)t_flags, (int)n_tile, t_list);
        TOMO_LAUNCH(ctx, "k_adj_tile", (k_tile<false, TILE_ADJ_WAVES>), dim3((unsigned)n_tile), dim3(TILE_ADJ_WAVES * 64), 0, d_c + n_flat, n_proj - n_flat, (float *)d_proj, d_vol, g,
                    (const unsigned *)d_absmax, (float)weight_bound, xt0, (const int *)t_list, (int)grid.x, (int)grid.y, (const int *)d_zcum);
    }
    return TOMO_OK;
}

extern "C" int tomo_adjoint(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_proj, float *d_vol, int accumulate)
{
    TomoRange roctx_range("tomo_adjoint");
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
    TomoRange roctx_range("tomo_adjoint_xslab");
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
    TomoRange roctx_range("tomo_backproject_voxel");
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
        // float32 matrices exactly as src/rotations_module.f90:6-54 builds them (angles arrive as real(kind=4))
        const float ph = (float)p[0], al = (float)p[1], be = (float)p[2];
        const float cp = cosf(ph), sp = sinf(ph), ca = cosf(al), sa = sinf(al), cb = cosf(be), sb = sinf(be);
        const float rp[3][3] = {{cp, -sp, 0.f}, {sp, cp, 0.f}, {0.f, 0.f, 1.f}};
        const float ra[3][3] = {{1.f, 0.f, 0.f}, {0.f, ca, -sa}, {0.f, sa, ca}};
        const float rb[3][3] = {{cb, 0.f, sb}, {0.f, 1.f, 0.f}, {-sb, 0.f, cb}};
        memcpy(h[i].rp, rp, sizeof(rp));
        memcpy(h[i].ra, ra, sizeof(ra));
        memcpy(h[i].rb, rb, sizeof(rb));
        for (int k = 0; k < 3; ++k) h[i].t[k] = (float)p[3 + k];
    }
    if (n_proj) TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_stage, h, sizeof(BpC) * (size_t)n_proj, hipMemcpyHostToDevice, ctx->stream));
    TOMO_LAUNCH(ctx, "k_bp_voxel", k_bp_voxel, dim3((g.nz + 63) / 64, (g.ny + 3) / 4, g.nx), dim3(256), 0, (const BpC *)ctx->d_stage,
                n_proj, d_det, d_vol, g, ctx->vox_pitch[0], ctx->vox_pitch[1], ctx->vox_pitch[2]);
    return TOMO_OK;
}

extern "C" int tomo_proj_grad(tomo_ctx *ctx, const double *h_pose, const float *d_vol, float *d_proj, float *d_grad, int row_order)
{
    TomoRange roctx_range("tomo_proj_grad");
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
    const int variant = ctx->wide_rows ? 1 : ctx->grad_variant == 4 ? (n_plain ? 2 : 3) : ctx->grad_variant;       // 4 = by tilt
    if (variant == 1 && ctx->grad_v1_prec == 0)
        TOMO_LAUNCH(ctx, "k_proj_grad", k_proj_grad<false>, ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
                    (const float *)nullptr, (float *)nullptr, (double *)nullptr, g, row_order);
    else if (variant == 1 && ctx->grad_v1_prec == 1)      // diagnostics: float64 sample positions / float64 lerps and sums / both
        TOMO_LAUNCH(ctx, "k_proj_grad", (k_proj_grad<false, 1>), ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
                    (const float *)nullptr, (float *)nullptr, (double *)nullptr, g, row_order);
    else if (variant == 1 && ctx->grad_v1_prec == 2)
        TOMO_LAUNCH(ctx, "k_proj_grad", (k_proj_grad<false, 2>), ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
                    (const float *)nullptr, (float *)nullptr, (double *)nullptr, g, row_order);
    else if (variant == 1)
        TOMO_LAUNCH(ctx, "k_proj_grad", (k_proj_grad<false, 3>), ray_grid(g, 1), dim3(256), 0, d_pc, d_gc, ctx->d_volpad, d_proj, d_grad,
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
    TomoRange roctx_range("tomo_cost_grad_rows");
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
    // the work-groups' partial sums (7 float64 each), added in a fixed order by k_cost_grad_reduce: at most TOMO_RED_PART_BYTES of them at
    // a time (512^2 detector: 56 KB per pose; 1024^2: 224 KB), larger batches go through in several launches
    const int n_wg = grad_wg_per_proj(g);
    const size_t per_pose = (size_t)n_wg * 7;
    const int max_poses = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, (TOMO_RED_PART_BYTES / sizeof(double)) / per_pose));
    rc = tomo_ensure_red_part(ctx, per_pose * (size_t)max_poses);
    if (rc) return rc;
    ProjC *d_pc = nullptr;
    GradC *d_gc = nullptr;
    int n_plain = 0;                                      // staged order: [near-untilted poses ..., tilted poses ...]
    rc = upload_projc(ctx, h_poses, n, true, &d_pc, &d_gc, h_rows, &n_plain);
    if (rc) return rc;
    // up to two groups: the first with one kernel variant, the second with another (grad_variant 4: v2 / v3 by tilt)
    const int var_a = ctx->wide_rows ? 1 : ctx->grad_variant == 4 ? 2 : ctx->grad_variant;
    const int var_b = ctx->wide_rows ? 1 : ctx->grad_variant == 4 ? 3 : ctx->grad_variant;
    for (int part = 0; part < 2; ++part) {
        const int gfirst = part == 0 ? 0 : n_plain, gcnt = part == 0 ? n_plain : n - n_plain, variant = part == 0 ? var_a : var_b;
        for (int first = gfirst; first < gfirst + gcnt; first += max_poses) {
            const int cnt = std::min(max_poses, gfirst + gcnt - first);
            const ProjC *pc = d_pc + first;
            const GradC *gc = d_gc + first;
            // block order by the size of the GROUP, not of this launch: the per-ray arithmetic does not depend on it, the cache behaviour does
            const bool zslow = grad_zslow(g, gcnt) && cnt > 1;
            const dim3 gg = zslow ? dim3((g.ndx + 3) / 4, cnt, (g.ndz + 63) / 64) : ray_grid(g, cnt);
            if (variant == 1)
                TOMO_LAUNCH(ctx, "k_cost_grad(v1)", k_proj_grad<true>, ray_grid(g, cnt), dim3(256), 0, pc, gc, ctx->d_volpad, (float *)nullptr,
                            (float *)nullptr, d_b, d_resid, ctx->d_red_part, g, 0);
            else if (variant == 2)
                TOMO_LAUNCH(ctx, "k_cost_grad(v2)", k_proj_grad_v2<true>, gg, dim3(256), 0, pc, gc, ctx->d_volpad, (float *)nullptr,
                            (float *)nullptr, d_b, d_resid, ctx->d_red_part, g, zslow ? 16 : 0);
            else
                TOMO_LAUNCH(ctx, "k_cost_grad(v3)", k_proj_grad_v3<true>, gg, dim3(256), 0, pc, gc, ctx->d_volpad, (float *)nullptr,
                            (float *)nullptr, d_b, d_resid, ctx->d_red_part, g, zslow ? 16 : 0);
            TOMO_LAUNCH(ctx, "k_cost_grad_reduce", k_cost_grad_reduce, dim3(cnt), dim3(256), 0, gc, (const double *)ctx->d_red_part, ctx->d_red, n_wg);
        }
    }
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red, ctx->d_red, sizeof(double) * (size_t)n * 7, hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) {
        h_cost[i] = ctx->h_red[(size_t)i * 7];
        for (int k = 0; k < 6; ++k) h_grad6[(size_t)i * 6 + k] = ctx->h_red[(size_t)i * 7 + 1 + k];
    }
    return TOMO_OK;
}

// the lattice constants of `n_proj` poses on the device, for tomo_csr.hip (valid until the next call that stages poses)
int tomo_upload_projc_for_csr(tomo_ctx *ctx, const double *h_poses, int n_proj, ProjC **d_pc)
{
    return upload_projc(ctx, h_poses, n_proj, false, d_pc, nullptr);
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
