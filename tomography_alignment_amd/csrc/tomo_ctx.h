// tomo_ctx.h -- internal context shared by the translation units of libtomo_hip.so
#ifndef TOMO_CTX_H_
#define TOMO_CTX_H_
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <deque>
#include <map>
#include <string>
#include <vector>

#include "../../include/tomo.h"
#include "tomo_raycore.h"

struct BpC {   // voxel-driven back-projector constants of one projection: the reference's three float32 rotation matrices and
               // the translation (src/external_back_projection.f90:17-25, src/rotations_module.f90:6-54)
    float rp[3][3], ra[3][3], rb[3][3];   // Rz(phi), Rx(alpha), Ry(beta)
    float t[3];
};

struct ProfRec { int64_t n = 0; double ms = 0.0; };
struct ProfPending { std::string name; hipEvent_t e0, e1; };

struct tomo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool has_geom = false;
    TomoGeomC g{};
    double vox_pitch[3] = {1, 1, 1};
    // padded scratch volume (zero halo) and its state
    float *d_volpad = nullptr;
    size_t volpad_elems = 0;
    size_t volpad_rows = 0;             // nx * ny the row scratch behind the padded copy was sized for
    bool halo_dirty = true;
    bool wide_rows = false;             // TOMO_GEOM_WIDE_ROWS: padded x-row pitch >= 2^23 bytes, detector-z pitch or sample step > 1 voxel -- the 24-bit-multiply / fixed-bias kernels are not used
    const void *staged_src = nullptr;   // device pointer whose contents the padded copy currently holds
    int reuse_staged = 0;               // caller vouches: contents of staged_src unchanged since it was staged
    // per-projection constants (pinned host staging + device)
    void *h_stage = nullptr;
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    // the tile kernels' constants staged last (stage_tile_consts): reused while the same poses come again and nobody else has
    // used the staging buffers -- the x-slab calls of one pipelined back-projection, forward + back-projection of a solver
    // iteration, every iteration of a solver (a re-stage drains the stream: the GPU would idle between slab kernels)
    bool tile_cache_valid = false;
    std::vector<double> tile_cache_poses;
    int tile_cache_opts = 0, tile_cache_nflat = 0, tile_cache_ngather = 0;
    bool tile_cache_ok = false;
    double tile_cache_wb = 2.0;
    int tile_cache_zc_lo = 0, tile_cache_zc_hi = 0;   // range of the gather-eligible projections' integer z offsets (GfC::zc)
    double tile_cache_eb_max = 0.0;     // largest |m10| + |m11| of the gather-eligible projections staged (picks NJ)
    size_t tile_cache_gfoff = 0;
    // reduction scratch
    double *d_red = nullptr;
    double *h_red = nullptr;
    size_t red_cap = 0;     // doubles
    // measurement hook: synthetic copy traffic beside the asynchronous collectives (tomo_ctx.hip: comm_async)
    int comm_test_copy_eighths = 0, comm_test_copy_wgs = 0;
    void *d_comm_scratch = nullptr;
    size_t comm_scratch_bytes = 0;
    // work-group partial sums of the fused cost / gradient kernels (tomo_cost_grad_rows: added in a fixed order, no atomics); grow-only,
    // handed back by tomo_release_workspace
    double *d_red_part = nullptr;
    size_t red_part_cap = 0;     // doubles
    // TOMO_N_ACC double accumulators (tomo_acc_zero / tomo_vec_dot_acc / tomo_acc_fetch) + their pinned host mirror
    double *d_acc = nullptr;
    double *h_acc = nullptr;
    // live-block list of the flat forward (grow-only): [0] = number of live blocks, [1 ..] their ids in launch order, then one flag byte per block
    int *d_blk = nullptr;
    size_t blk_ints = 0;
    int reuse_sino_flags = 0;           // option: the caller vouches the sinogram of the previous back-projection call is unchanged
    const void *zf_src = nullptr;       // sinogram whose plane flags (+ prefix counts when zf_has_cum) d_blk currently holds; nullptr: none
    int zf_nproj = 0;
    bool zf_has_cum = false, zf_has_shift = false;
    int zf_ndz = 0, zf_ndx = 0, zf_nz = 0, zf_zc_lo = 0, zf_zc_hi = 0;   // ... and the geometry / pose z offsets they were computed for
    size_t fwd_blk_flat_ints = 0;       // ints of d_blk the flat forward of the current call uses (the general kernel's tile list follows)
    // general float workspace (grow-only): the TV-FISTA proximal step keeps its 7 fields here across calls
    float *d_ws = nullptr;
    size_t ws_elems = 0;
    // assembled CSR kept between tomo_csr_assemble and tomo_csr_fetch (tomo_csr.hip)
    void *csr_data = nullptr, *csr_indices = nullptr, *csr_indptr = nullptr;
    int64_t csr_nnz = 0, csr_rows = 0;
    int csr_value_bytes = 4;
    // options
    int fwd_variant = 3;      // 1 ray-driven plain, 2 ray-driven SGPR-base, 3 LDS tile (default)
    int adj_variant = 2;      // 1 global float atomics, 2 LDS tile fixed-point (default)
    int grad_variant = 4;     // 1 plain, 2 eight dword gathers + packed lerps, 3 four gathers + DPP neighbour shift, 4 (default) 2 or 3 per pose by tilt
    int grad_v1_prec = 0;     // diagnostic for grad_variant 1 (per-ray tomo_proj_grad only): bit 0 float64 sample positions, bit 1 float64 lerps and sums
    int tile_flat = 1;      // 1: untilted projections take the flat tile kernels
    int adj_flat_gather = 1;  // 1: untilted unit lattices take the gather-form adjoint (k_adj_gather_flat) instead of the LDS-atomic flat kernel
    int fwd_flat_tab = 1;     // 1 (default): the flat forward with the sample table in LDS and the two images interleaved per plane (k_fwd_flat_tab); 0: the round-2 kernel (k_fwd_flat_z<2>: entries broadcast with v_readlane)
    int fwd_flat_wide = 0;    // 1: measurement variant of the flat forward -- 32 x 16 x 63 footprint, one image per work-group (k_fwd_flat_z<1, 32>); only in builds with -DTOMO_MEASUREMENT_VARIANTS
    int fwd_flat_ztiles = 2;  // 2: the flat forward processes two z-adjacent tiles per work-group (k_fwd_flat_z<2>); 1: one tile (k_tile_flat<true>)
    // timing / profiling
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profile_on = false;
    std::map<std::string, ProfRec> prof;
    std::vector<ProfPending> pending;
    std::vector<hipEvent_t> ev_pool;
    // comm
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_compute = nullptr, ev_comm = nullptr;
    bool comm_pending = false;
    std::deque<hipEvent_t> comm_done;    // one event per asynchronous all-reduce / reduce-scatter the compute stream has not yet waited for (issue order)
    std::deque<hipEvent_t> comm_done_g;  // ... and per asynchronous all-gather
    int comm_test_poison_us = 0;         // test hook, see comm_async
    std::vector<hipEvent_t> comm_ev_pool;
    int n_ranks = 1, rank = 0;
    std::string err;
};

int tomo_fail(tomo_ctx *ctx, int code, const std::string &msg);

// roctx ranges around the C-ABI's projector / gradient / collective entry points (the reference has print timings only).  Off by
// default and without a link-time dependency: TOMO_ROCTX=1 in the environment, or tomo_set_option(ctx, "roctx", 1), dlopens
// librocprofiler-sdk-roctx.so (rocprofv3 --marker-trace) -- or libroctx64.so -- on first use; the ranges are named after the entry points.
struct TomoRange {
    bool on;
    explicit TomoRange(const char *name);
    ~TomoRange();
    TomoRange(const TomoRange &) = delete;
    TomoRange &operator=(const TomoRange &) = delete;
};
int tomo_roctx_enable(int on);      // 0 ok, -1: no roctx library could be loaded
int tomo_ensure_stage(tomo_ctx *ctx, size_t bytes);
int tomo_ensure_red(tomo_ctx *ctx, size_t n_doubles);
int tomo_ensure_red_part(tomo_ctx *ctx, size_t n_doubles);
int tomo_ensure_ws(tomo_ctx *ctx, size_t n_floats);
int tomo_ensure_blk(tomo_ctx *ctx, size_t n_ints);
void tomo_csr_release(tomo_ctx *ctx);
void tomo_prof_begin(tomo_ctx *ctx, const char *name);
void tomo_prof_end(tomo_ctx *ctx);
void tomo_prof_begin_on(tomo_ctx *ctx, const char *name, hipStream_t stream);
void tomo_prof_end_on(tomo_ctx *ctx, hipStream_t stream);

#define TOMO_HIP(ctx, call)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return tomo_fail((ctx), TOMO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define TOMO_NEED_GEOM(ctx)                                                              \
    do {                                                                                 \
        if (!(ctx)) return tomo_fail(nullptr, TOMO_ERR_ARG, "null ctx");                 \
        if (!(ctx)->has_geom) return tomo_fail((ctx), TOMO_ERR_STATE, "geometry not set"); \
        {   /* the current device is per THREAD: a caller's helper thread starts on device 0 */                     \
            hipError_t e_ = hipSetDevice((ctx)->device);                                                         \
            if (e_ != hipSuccess) return tomo_fail((ctx), TOMO_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e_)); \
        }                                                                                                        \
    } while (0)

// launch with optional event bracketing (tomo_profile_enable) and launch-error check
#define TOMO_LAUNCH(ctx, name, kern, grid, block, shmem, ...)                                   \
    do {                                                                                        \
        tomo_prof_begin((ctx), (name));                                                         \
        hipLaunchKernelGGL(kern, (grid), (block), (shmem), (ctx)->stream, __VA_ARGS__);         \
        tomo_prof_end((ctx));                                                                   \
        TOMO_HIP((ctx), hipGetLastError());                                                     \
    } while (0)

#endif
