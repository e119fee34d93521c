// tomo_ctx.hip -- context, device memory, geometry, profiling, solver vector kernels, RCCL.
// (Projection kernels live in tomo_project.hip.)
#include <string.h>

#include <algorithm>

#include "tomo_ctx.h"

static std::string g_last_error;

int tomo_fail(tomo_ctx *ctx, int code, const std::string &msg)
{
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}

extern "C" int tomo_abi_version(void) { return TOMO_ABI_VERSION; }

extern "C" const char *tomo_last_error(const tomo_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int tomo_device_count(int *n)
{
    if (!n) return tomo_fail(nullptr, TOMO_ERR_ARG, "null n");
    TOMO_HIP(nullptr, hipGetDeviceCount(n));
    return TOMO_OK;
}

extern "C" int tomo_ctx_create(int device, tomo_ctx **out)
{
    if (!out) return tomo_fail(nullptr, TOMO_ERR_ARG, "null out");
    *out = nullptr;
    TOMO_HIP(nullptr, hipSetDevice(device));
    tomo_ctx *c = new tomo_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        std::string m = std::string("ctx_create: ") + hipGetErrorString(e);
        delete c;
        return tomo_fail(nullptr, TOMO_ERR_HIP, m);
    }
    *out = c;
    return TOMO_OK;
}

extern "C" int tomo_ctx_destroy(tomo_ctx *c)
{
    if (!c) return TOMO_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->ev_compute) (void)hipEventDestroy(c->ev_compute);
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    for (auto e : c->comm_done) (void)hipEventDestroy(e);
    for (auto e : c->comm_done_g) (void)hipEventDestroy(e);
    for (auto e : c->comm_ev_pool) (void)hipEventDestroy(e);
    for (auto &p : c->pending) { c->ev_pool.push_back(p.e0); c->ev_pool.push_back(p.e1); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->d_volpad) (void)hipFree(c->d_volpad);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->h_red) (void)hipHostFree(c->h_red);
    if (c->d_red_part) (void)hipFree(c->d_red_part);
    if (c->d_comm_scratch) (void)hipFree(c->d_comm_scratch);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->h_acc) (void)hipHostFree(c->h_acc);
    if (c->d_ws) (void)hipFree(c->d_ws);
    tomo_csr_release(c);
    if (c->d_blk) (void)hipFree(c->d_blk);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TOMO_OK;
}

extern "C" int tomo_device_name(tomo_ctx *ctx, char *buf, size_t n)
{
    if (!ctx || !buf || !n) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    hipDeviceProp_t p;
    TOMO_HIP(ctx, hipGetDeviceProperties(&p, ctx->device));
    snprintf(buf, n, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return TOMO_OK;
}

extern "C" int tomo_malloc(tomo_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    TOMO_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 4));
    return TOMO_OK;
}

extern "C" int tomo_free(tomo_ctx *ctx, void *d_ptr)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    if (!d_ptr) return TOMO_OK;
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    TOMO_HIP(ctx, hipFree(d_ptr));
    return TOMO_OK;
}

extern "C" int tomo_memcpy_h2d(tomo_ctx *ctx, void *d, const void *h, size_t bytes)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));   // caller may reuse h immediately
    return TOMO_OK;
}

extern "C" int tomo_memcpy_d2h(tomo_ctx *ctx, void *h, const void *d, size_t bytes)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_memcpy_d2d(tomo_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_memset0(tomo_ctx *ctx, void *d, size_t bytes)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipMemsetAsync(d, 0, bytes, ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_sync(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TOMO_OK;
}

// HIP's current device is a per-thread setting and a new thread starts on device 0: a caller that uses a context from a thread other
// than the one that created it (alignment.py evaluates its batches from a helper thread) binds the thread first.  Every entry point
// that needs a geometry does the same on its own (TOMO_NEED_GEOM), this covers the rest (allocation, copies, collectives).
extern "C" int tomo_ctx_make_current(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    return TOMO_OK;
}

extern "C" int tomo_ctx_set_cu_mask(tomo_ctx *ctx, const uint32_t *mask, int n_words)
{
    if (!ctx || n_words < 0 || (n_words > 0 && !mask)) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_ctx_set_cu_mask: bad args");
    if (n_words > 0) {
        bool any = false;
        for (int i = 0; i < n_words; ++i) any |= mask[i] != 0;
        if (!any) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_ctx_set_cu_mask: empty mask (n_words = 0 restores the unrestricted stream)");
    }
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipStream_t s = nullptr;
    if (n_words == 0) TOMO_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else TOMO_HIP(ctx, hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
    (void)hipStreamDestroy(ctx->stream);
    ctx->stream = s;
    return TOMO_OK;
}

extern "C" int tomo_set_option(tomo_ctx *ctx, const char *key, int value)
{
    if (!ctx || !key) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    if (!strcmp(key, "fwd_variant")) ctx->fwd_variant = value;
    else if (!strcmp(key, "adj_variant")) ctx->adj_variant = value;
    else if (!strcmp(key, "tile_flat")) ctx->tile_flat = value;
    else if (!strcmp(key, "grad_variant")) ctx->grad_variant = value;
    else if (!strcmp(key, "grad_v1_prec")) ctx->grad_v1_prec = value & 3;
    else if (!strcmp(key, "comm_test_poison_us")) ctx->comm_test_poison_us = value;
    else if (!strcmp(key, "comm_test_copy_eighths")) ctx->comm_test_copy_eighths = value < 0 ? 0 : value;
    else if (!strcmp(key, "comm_test_copy_wgs")) ctx->comm_test_copy_wgs = value < 0 ? 0 : value;
    else if (!strcmp(key, "adj_flat_gather")) ctx->adj_flat_gather = value;
    else if (!strcmp(key, "fwd_flat_ztiles")) ctx->fwd_flat_ztiles = value;
    else if (!strcmp(key, "fwd_flat_wide")) {
#ifdef TOMO_MEASUREMENT_VARIANTS
        ctx->fwd_flat_wide = value;
#else
        if (value) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "fwd_flat_wide is a measurement variant: build with EXTRA=-DTOMO_MEASUREMENT_VARIANTS (csrc/Makefile)");
#endif
    }
    else if (!strcmp(key, "fwd_flat_tab")) ctx->fwd_flat_tab = value;
    else if (!strcmp(key, "reuse_staged_volume")) ctx->reuse_staged = value;
    else if (!strcmp(key, "reuse_sino_flags")) ctx->reuse_sino_flags = value;
    else if (!strcmp(key, "roctx")) {
        if (tomo_roctx_enable(value)) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "roctx: neither librocprofiler-sdk-roctx.so nor libroctx64.so could be loaded");
    }
    else return tomo_fail(ctx, TOMO_ERR_ARG, std::string("unknown option ") + key);
    return TOMO_OK;
}

int tomo_ensure_stage(tomo_ctx *ctx, size_t bytes)
{
    ctx->tile_cache_valid = false;                       // whoever asks for the staging buffers is about to overwrite them
    if (bytes <= ctx->stage_bytes) return TOMO_OK;
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->d_stage = ctx->h_stage = nullptr;
    ctx->stage_bytes = 0;
    size_t cap = std::max<size_t>(bytes, 1 << 16);
    TOMO_HIP(ctx, hipMalloc(&ctx->d_stage, cap));
    TOMO_HIP(ctx, hipHostMalloc(&ctx->h_stage, cap, hipHostMallocDefault));
    ctx->stage_bytes = cap;
    return TOMO_OK;
}

int tomo_ensure_red(tomo_ctx *ctx, size_t n)
{
    if (n <= ctx->red_cap) return TOMO_OK;
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_red) (void)hipFree(ctx->d_red);
    if (ctx->h_red) (void)hipHostFree(ctx->h_red);
    ctx->d_red = ctx->h_red = nullptr;
    ctx->red_cap = 0;
    size_t cap = std::max<size_t>(n, 4096);
    TOMO_HIP(ctx, hipMalloc((void **)&ctx->d_red, cap * sizeof(double)));
    TOMO_HIP(ctx, hipHostMalloc((void **)&ctx->h_red, cap * sizeof(double), hipHostMallocDefault));
    ctx->red_cap = cap;
    return TOMO_OK;
}

int tomo_ensure_red_part(tomo_ctx *ctx, size_t n)
{
    if (n <= ctx->red_part_cap) return TOMO_OK;
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_red_part) (void)hipFree(ctx->d_red_part);
    ctx->d_red_part = nullptr;
    ctx->red_part_cap = 0;
    TOMO_HIP(ctx, hipMalloc((void **)&ctx->d_red_part, n * sizeof(double)));
    ctx->red_part_cap = n;
    return TOMO_OK;
}

// The grow-only workspaces a context keeps between calls can be large (the TV-FISTA proximal step holds 7 volumes: 28 GB at 1024^3):
// a long-lived context hands them back with this.  The next call that needs one allocates it again.
extern "C" int tomo_release_workspace(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    ctx->d_ws = nullptr;
    ctx->ws_elems = 0;
    if (ctx->d_red_part) (void)hipFree(ctx->d_red_part);
    ctx->d_red_part = nullptr;
    ctx->red_part_cap = 0;
    return TOMO_OK;
}

int tomo_ensure_ws(tomo_ctx *ctx, size_t n)
{
    if (n <= ctx->ws_elems) return TOMO_OK;
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    ctx->d_ws = nullptr;
    ctx->ws_elems = 0;
    TOMO_HIP(ctx, hipMalloc((void **)&ctx->d_ws, n * sizeof(float)));
    ctx->ws_elems = n;
    return TOMO_OK;
}

int tomo_ensure_blk(tomo_ctx *ctx, size_t n)
{
    if (n <= ctx->blk_ints) return TOMO_OK;
    ctx->zf_src = nullptr;                              // the buffer is replaced: cached plane flags go with it
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_blk) (void)hipFree(ctx->d_blk);
    ctx->d_blk = nullptr;
    ctx->blk_ints = 0;
    TOMO_HIP(ctx, hipMalloc((void **)&ctx->d_blk, n * sizeof(int)));
    ctx->blk_ints = n;
    return TOMO_OK;
}

// Pure host check shared by tomo_set_geometry and callers without a context (tests run it on CPU).
extern "C" int tomo_check_geometry(const tomo_geom *g, int *flags)
{
    if (flags) *flags = 0;
    if (!g) return tomo_fail(nullptr, TOMO_ERR_ARG, "check_geometry: null geometry");
    if (g->nx < 1 || g->ny < 1 || g->nz < 1 || g->ndx < 1 || g->ndz < 1 || !(g->step > 0.0) || !(g->det_y > g->src_y))
        return tomo_fail(nullptr, TOMO_ERR_ARG, "set_geometry: non-positive shape/step or det_y <= src_y");
    const size_t nxp = (size_t)g->nx + 2 * TOMO_HALO, nyp = (size_t)g->ny + 2 * TOMO_HALO, nzp = (size_t)g->nz + 2 * TOMO_HALO;
    if (nxp * nyp * nzp >= ((size_t)1 << 31)) return tomo_fail(nullptr, TOMO_ERR_UNSUPPORTED, "set_geometry: padded volume exceeds 2^31 voxels");
    // the SGPR-base kernels (k_fwd_v2, k_proj_grad_v2/_v3) form lane offsets with SIGNED 24-bit multiplies: cell * (row pitch in bytes)
    // ... bias a wave's lane offsets by TOMO_LBIAS = 66 cells per axis, which covers 64 rays only while the detector-z pitch <= 1 voxel
    // ... and address a block's cells relative to its mid-block anchor with a bias of TOMO_ABIAS = 18 cells: in-block cells lie
    // within +-(15.5 |d_a| + 1) of the anchor, i.e. within +-17 only while the sample step is <= 1 voxel (with step 1.3
    // the range is +-21 and only the slack of the lane bias kept the unsigned offsets from wrapping; beyond ~5 voxels they
    // would).  Such geometries take the plain 64-bit kernels (variant 1), which have no such assumption.
    if (nyp * nzp * 4 >= ((size_t)1 << 23) || nzp * 4 >= ((size_t)1 << 23) || !(fabs(g->det_dz) <= 1.0 + 1e-9) || !(g->step <= 1.0 + 1e-9)) { if (flags) *flags |= TOMO_GEOM_WIDE_ROWS; }
    return TOMO_OK;
}

extern "C" int tomo_set_geometry(tomo_ctx *ctx, const tomo_geom *g)
{
    if (!ctx || !g) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    int gflags = 0;
    int grc = tomo_check_geometry(g, &gflags);
    if (grc) return tomo_fail(ctx, grc, tomo_last_error(nullptr));
    ctx->tile_cache_valid = false;
    TomoGeomC c{};
    c.nx = g->nx; c.ny = g->ny; c.nz = g->nz; c.ndx = g->ndx; c.ndz = g->ndz;
    c.nxp = g->nx + 2 * TOMO_HALO; c.nyp = g->ny + 2 * TOMO_HALO; c.nzp = g->nz + 2 * TOMO_HALO;
    for (int a = 0; a < 3; ++a) c.org[a] = g->vox_origin[a];
    c.det_x0 = g->det_x0; c.det_z0 = g->det_z0; c.det_dx = g->det_dx; c.det_dz = g->det_dz;
    c.src_y = g->src_y; c.det_y = g->det_y; c.step = g->step;
    size_t pe = (size_t)c.nxp * c.nyp * c.nzp;
    ctx->wide_rows = (gflags & TOMO_GEOM_WIDE_ROWS) != 0;
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    if (pe != ctx->volpad_elems || (size_t)g->nx * g->ny != ctx->volpad_rows) {
        TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_volpad) (void)hipFree(ctx->d_volpad);
        ctx->d_volpad = nullptr;
        ctx->volpad_elems = 0;
        // + the non-zero box behind it (TOMO_BOX_INTS = 8 ints) + one int2 per (x, y) row for its reduction (k_pad, k_box)
        TOMO_HIP(ctx, hipMalloc((void **)&ctx->d_volpad, (pe + 8 + 2 * (size_t)g->nx * g->ny) * sizeof(float)));
        ctx->volpad_elems = pe;
        ctx->volpad_rows = (size_t)g->nx * g->ny;
    }
    ctx->halo_dirty = true;
    ctx->staged_src = nullptr;
    ctx->zf_src = nullptr;                  // cached sinogram plane flags belong to the geometry they were scanned under
    ctx->g = c;
    for (int a = 0; a < 3; ++a) ctx->vox_pitch[a] = g->vox_pitch[a];
    ctx->has_geom = true;
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// timing / profiling
// ------------------------------------------------------------------------------------------------
static hipEvent_t prof_event(tomo_ctx *ctx)
{
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// ---- roctx (see tomo_ctx.h): process-wide, loaded on demand
#include <dlfcn.h>
namespace {
struct Roctx {
    int state = 0;                  // 0 not tried, 1 on, -1 unavailable, 2 loaded but switched off
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
} g_roctx;
bool roctx_load()
{
    for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop");
        if (g_roctx.push && g_roctx.pop) return true;
    }
    return false;
}
bool roctx_on()
{
    if (g_roctx.state == 0) {
        const char *e = getenv("TOMO_ROCTX");
        g_roctx.state = (e && *e && *e != '0') ? (roctx_load() ? 1 : -1) : 2;
    }
    return g_roctx.state == 1;
}
}  // namespace
int tomo_roctx_enable(int on)
{
    if (!on) { if (g_roctx.state == 1) g_roctx.state = 2; return 0; }
    if (!g_roctx.push && !roctx_load()) { g_roctx.state = -1; return -1; }
    g_roctx.state = 1;
    return 0;
}
TomoRange::TomoRange(const char *name) : on(roctx_on()) { if (on) (void)g_roctx.push(name); }
TomoRange::~TomoRange() { if (on) (void)g_roctx.pop(); }

void tomo_prof_begin_on(tomo_ctx *ctx, const char *name, hipStream_t stream)
{
    if (!ctx->profile_on) return;
    ProfPending p;
    p.name = name;
    p.e0 = prof_event(ctx);
    p.e1 = prof_event(ctx);
    (void)hipEventRecord(p.e0, stream);
    ctx->pending.push_back(p);
}

void tomo_prof_end_on(tomo_ctx *ctx, hipStream_t stream)
{
    if (!ctx->profile_on || ctx->pending.empty()) return;
    (void)hipEventRecord(ctx->pending.back().e1, stream);
}

void tomo_prof_begin(tomo_ctx *ctx, const char *name) { tomo_prof_begin_on(ctx, name, ctx->stream); }
void tomo_prof_end(tomo_ctx *ctx) { tomo_prof_end_on(ctx, ctx->stream); }

static void prof_drain(tomo_ctx *ctx)
{
    if (ctx->pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);     // asynchronous all-reduces are bracketed on their own stream
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ProfRec &r = ctx->prof[p.name];
            r.n += 1;
            r.ms += ms;
        }
        ctx->ev_pool.push_back(p.e0);
        ctx->ev_pool.push_back(p.e1);
    }
    ctx->pending.clear();
}

extern "C" int tomo_timer_start(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    TOMO_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_timer_stop(tomo_ctx *ctx, float *ms)
{
    if (!ctx || !ms) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    TOMO_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    TOMO_HIP(ctx, hipEventSynchronize(ctx->ev1));
    TOMO_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return TOMO_OK;
}

extern "C" int tomo_profile_enable(tomo_ctx *ctx, int on)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    if (!on) prof_drain(ctx);
    ctx->profile_on = on != 0;
    return TOMO_OK;
}

extern "C" int tomo_profile_reset(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    prof_drain(ctx);
    ctx->prof.clear();
    return TOMO_OK;
}

extern "C" int tomo_profile_get(tomo_ctx *ctx, const char *kernel, int64_t *n, double *ms)
{
    if (!ctx || !kernel) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    prof_drain(ctx);
    // exact name, plus every record named `kernel(...)`: "k_cost_grad" sums its variants "k_cost_grad(v2)", "k_cost_grad(v3)"
    int64_t cnt = 0;
    double tot = 0.0;
    const std::string base(kernel), pre = base + "(";
    for (const auto &kv : ctx->prof)
        if (kv.first == base || kv.first.compare(0, pre.size(), pre) == 0) { cnt += kv.second.n; tot += kv.second.ms; }
    if (n) *n = cnt;
    if (ms) *ms = tot;
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// solver vector kernels: HBM-streaming, coalesced dword grid-stride loops, <= 2048 work-groups
// (they move ~0.6 % of a SIRT iteration's bytes: recon/sirt.py:60-73 vs :59,61).
// Reductions: wave shuffle -> LDS -> one double atomic per work-group.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void block_atomic_sum(double v, double *dst)
{
    __shared__ double sh[4];
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = sh[0] + sh[1] + sh[2] + sh[3];
        atomicAdd(dst, s);
    }
}

enum VecOp { VOP_RECIP_STRICT, VOP_RECIP_THRESH, VOP_FILL, VOP_AXPY, VOP_XPAY, VOP_SUB, VOP_MUL };

template <int OP>
__global__ __launch_bounds__(256) void k_vec(float *__restrict__ y, const float *__restrict__ a, const float *__restrict__ b,
                                             float s, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (OP == VOP_RECIP_STRICT) { float v = y[i]; y[i] = (v == 0.f) ? 0.f : 1.f / v; }      // recon/sirt.py:37-40 (x -> inf -> 1/inf = 0)
        else if (OP == VOP_RECIP_THRESH) { float v = y[i]; y[i] = (v < s) ? 0.f : 1.f / v; }     // recon/sirt_mpi.py:69-72
        else if (OP == VOP_FILL) y[i] = s;
        else if (OP == VOP_AXPY) y[i] = fmaf(s, a[i], y[i]);
        else if (OP == VOP_XPAY) y[i] = fmaf(s, y[i], a[i]);
        else if (OP == VOP_SUB) y[i] = a[i] - b[i];
        else if (OP == VOP_MUL) y[i] *= a[i];
    }
}

static inline int vec_grid(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }

#define VEC_CHECK(ctx, n)                                                  \
    do {                                                                   \
        if (!(ctx)) return tomo_fail(nullptr, TOMO_ERR_ARG, "null ctx");   \
        if ((n) < 0) return tomo_fail((ctx), TOMO_ERR_ARG, "negative n");  \
        if ((n) == 0) return TOMO_OK;                                      \
    } while (0)

extern "C" int tomo_vec_recip_guard(tomo_ctx *ctx, float *v, int64_t n, float thresh, int strict_zero)
{
    VEC_CHECK(ctx, n);
    if (strict_zero) TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_RECIP_STRICT>, dim3(vec_grid(n)), dim3(256), 0, v, nullptr, nullptr, 0.f, n);
    else TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_RECIP_THRESH>, dim3(vec_grid(n)), dim3(256), 0, v, nullptr, nullptr, thresh, n);
    return TOMO_OK;
}
extern "C" int tomo_vec_fill(tomo_ctx *ctx, float *v, int64_t n, float value)
{
    VEC_CHECK(ctx, n);
    TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_FILL>, dim3(vec_grid(n)), dim3(256), 0, v, nullptr, nullptr, value, n);
    return TOMO_OK;
}
extern "C" int tomo_vec_axpy(tomo_ctx *ctx, float *y, const float *x, float a, int64_t n)
{
    VEC_CHECK(ctx, n);
    TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_AXPY>, dim3(vec_grid(n)), dim3(256), 0, y, x, nullptr, a, n);
    return TOMO_OK;
}
extern "C" int tomo_vec_xpay(tomo_ctx *ctx, float *y, const float *x, float a, int64_t n)
{
    VEC_CHECK(ctx, n);
    TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_XPAY>, dim3(vec_grid(n)), dim3(256), 0, y, x, nullptr, a, n);
    return TOMO_OK;
}
extern "C" int tomo_vec_sub(tomo_ctx *ctx, float *out, const float *a, const float *b, int64_t n)
{
    VEC_CHECK(ctx, n);
    TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_SUB>, dim3(vec_grid(n)), dim3(256), 0, out, a, b, 0.f, n);
    return TOMO_OK;
}
extern "C" int tomo_vec_mul(tomo_ctx *ctx, float *y, const float *x, int64_t n)
{
    VEC_CHECK(ctx, n);
    TOMO_LAUNCH(ctx, "k_vec", k_vec<VOP_MUL>, dim3(vec_grid(n)), dim3(256), 0, y, x, nullptr, 0.f, n);
    return TOMO_OK;
}

// out = w*(b-ax), sumsq += (b-ax)^2          recon/sirt.py:60-61,69
__global__ __launch_bounds__(256) void k_residual_scale(const float *__restrict__ b, const float *__restrict__ ax,
                                                        const float *__restrict__ w, float *__restrict__ out, int64_t n, double *sumsq)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float r = b[i] - ax[i];
        acc += (double)r * (double)r;
        out[i] = w ? w[i] * r : r;
    }
    block_atomic_sum(acc, sumsq);
}

// rec += v*bp ; rec = max(rec,0) if positivity ; err += (gt-rec)^2     recon/sirt.py:63-67,73
__global__ __launch_bounds__(256) void k_update(float *__restrict__ rec, const float *__restrict__ bp, const float *__restrict__ v,
                                                int64_t n, int positivity, const float *__restrict__ gt, double *sumsq)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float r = rec[i] + (v ? bp[i] * v[i] : bp[i]);
        if (positivity && r < 0.f) r = 0.f;
        rec[i] = r;
        if (gt) { float e = gt[i] - r; acc += (double)e * (double)e; }
    }
    if (gt) block_atomic_sum(acc, sumsq);
}

__global__ __launch_bounds__(256) void k_dot(const float *__restrict__ a, const float *__restrict__ b, int64_t n, int diff, double *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (diff) { float e = a[i] - b[i]; acc += (double)e * (double)e; }
        else acc += (double)a[i] * (double)b[i];
    }
    block_atomic_sum(acc, out);
}

static int red_fetch(tomo_ctx *ctx, double *h_out)
{
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red, ctx->d_red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *h_out = ctx->h_red[0];
    return TOMO_OK;
}

extern "C" int tomo_vec_residual_scale(tomo_ctx *ctx, const float *b, const float *ax, const float *w, float *out, int64_t n,
                                       double *h_sumsq)
{
    VEC_CHECK(ctx, n);
    int rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, sizeof(double), ctx->stream));
    TOMO_LAUNCH(ctx, "k_residual_scale", k_residual_scale, dim3(vec_grid(n)), dim3(256), 0, b, ax, w, out, n, ctx->d_red);
    return h_sumsq ? red_fetch(ctx, h_sumsq) : TOMO_OK;
}

extern "C" int tomo_vec_update(tomo_ctx *ctx, float *rec, const float *bp, const float *v, int64_t n, int positivity,
                               const float *gt, double *h_sumsq_err)
{
    VEC_CHECK(ctx, n);
    int rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, sizeof(double), ctx->stream));
    TOMO_LAUNCH(ctx, "k_update", k_update, dim3(vec_grid(n)), dim3(256), 0, rec, bp, v, n, positivity, gt, ctx->d_red);
    return (gt && h_sumsq_err) ? red_fetch(ctx, h_sumsq_err) : TOMO_OK;
}

// The update of recon/sirt.py:63-67,73 applied to ONE x slab of a pipelined iteration: no host synchronisation; the error sum
// accumulates across the slabs in a device scalar (zeroed when `first`), fetched once by tomo_vec_update_acc_fetch.
#define TOMO_RED_UPDATE_ACC 2      // slot of d_red (0: one-shot reductions, 4: the tile adjoint's abs-max, 8..: host all-reduces)
extern "C" int tomo_vec_update_acc(tomo_ctx *ctx, float *rec, const float *bp, const float *v, int64_t n, int positivity, const float *gt, int first)
{
    if (!ctx) return tomo_fail(nullptr, TOMO_ERR_ARG, "null ctx");
    if (n < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "negative n");
    int rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    if (first) TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red + TOMO_RED_UPDATE_ACC, 0, sizeof(double), ctx->stream));
    if (n == 0) return TOMO_OK;
    TOMO_LAUNCH(ctx, "k_update", k_update, dim3(vec_grid(n)), dim3(256), 0, rec, bp, v, n, positivity, gt, ctx->d_red + TOMO_RED_UPDATE_ACC);
    return TOMO_OK;
}

extern "C" int tomo_vec_update_acc_fetch(tomo_ctx *ctx, double *h_sumsq_err)
{
    if (!ctx || !h_sumsq_err) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    int rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red + TOMO_RED_UPDATE_ACC, ctx->d_red + TOMO_RED_UPDATE_ACC, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *h_sumsq_err = ctx->h_red[TOMO_RED_UPDATE_ACC];
    return TOMO_OK;
}

extern "C" int tomo_vec_dot(tomo_ctx *ctx, const float *a, const float *b, int64_t n, double *h_dot)
{
    if (!h_dot) return tomo_fail(ctx, TOMO_ERR_ARG, "null out");
    *h_dot = 0.0;
    VEC_CHECK(ctx, n);
    int rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, sizeof(double), ctx->stream));
    TOMO_LAUNCH(ctx, "k_dot", k_dot, dim3(vec_grid(n)), dim3(256), 0, a, b, n, 0, ctx->d_red);
    return red_fetch(ctx, h_dot);
}

extern "C" int tomo_vec_diff_sumsq(tomo_ctx *ctx, const float *a, const float *b, int64_t n, double *h_sumsq)
{
    if (!h_sumsq) return tomo_fail(ctx, TOMO_ERR_ARG, "null out");
    *h_sumsq = 0.0;
    VEC_CHECK(ctx, n);
    int rc = tomo_ensure_red(ctx, 8);
    if (rc) return rc;
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_red, 0, sizeof(double), ctx->stream));
    TOMO_LAUNCH(ctx, "k_dot", k_dot, dim3(vec_grid(n)), dim3(256), 0, a, b, n, 1, ctx->d_red);
    return red_fetch(ctx, h_sumsq);
}

// ------------------------------------------------------------------------------------------------
// synthetic phantom (bench / test input): utilities/generate_phantom.py:81-179
// ------------------------------------------------------------------------------------------------
struct EllC { double A, inv[3], m[3], R[3][3]; };

__global__ __launch_bounds__(256) void k_phantom(float *__restrict__ vol, int nx, int ny, int nz, const EllC *__restrict__ tab, int n_rows)
{
    const int iz = blockIdx.x * 256 + threadIdx.x, iy = blockIdx.y, ix = blockIdx.z;
    if (iz >= nz) return;
    const double x = nx > 1 ? -1.0 + 2.0 * ix / (nx - 1) : -1.0;
    const double y = ny > 1 ? -1.0 + 2.0 * iy / (ny - 1) : -1.0;
    const double z = nz > 1 ? -1.0 + 2.0 * iz / (nz - 1) : -1.0;
    float acc = 0.f;
    for (int k = 0; k < n_rows; ++k) {
        const EllC e = tab[k];
        double r2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double q = (e.R[a][0] * x + e.R[a][1] * y + e.R[a][2] * z - e.m[a]) * e.inv[a];
            r2 += q * q;
        }
        if (r2 <= 1.0) acc = (float)((double)acc + e.A);
    }
    vol[((size_t)ix * ny + iy) * nz + iz] = fmaxf(acc, 0.f);
}

extern "C" int tomo_phantom_ellipsoids(tomo_ctx *ctx, float *d_vol, int nx, int ny, int nz, const double *h_table, int n_rows)
{
    if (!ctx || !d_vol || !h_table || nx < 1 || ny < 1 || nz < 1 || n_rows < 0 || ny > 65535 || nx > 65535)
        return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_phantom_ellipsoids: bad args");
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc = tomo_ensure_stage(ctx, sizeof(EllC) * (size_t)std::max(n_rows, 1));
    if (rc) return rc;
    EllC *h = (EllC *)ctx->h_stage;
    const double d2r = 3.14159265358979323846 / 180.0;
    for (int k = 0; k < n_rows; ++k) {
        const double *r = h_table + (size_t)k * 10;
        h[k].A = r[0];
        for (int a = 0; a < 3; ++a) { h[k].inv[a] = 1.0 / r[1 + a]; h[k].m[a] = r[4 + a]; }
        const double cp = cos(r[7] * d2r), sp = sin(r[7] * d2r), ct = cos(r[8] * d2r), st = sin(r[8] * d2r), cs = cos(r[9] * d2r), ss = sin(r[9] * d2r);
        const double R[3][3] = {{cs * cp - ct * sp * ss, cs * sp + ct * cp * ss, ss * st},
                                {-ss * cp - ct * sp * cs, -ss * sp + ct * cp * cs, cs * st},
                                {st * sp, -st * cp, ct}};
        memcpy(h[k].R, R, sizeof(R));
    }
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_stage, h, sizeof(EllC) * (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream));
    TOMO_LAUNCH(ctx, "k_phantom", k_phantom, dim3((nz + 255) / 256, ny, nx), dim3(256), 0, d_vol, nx, ny, nz, (const EllC *)ctx->d_stage, n_rows);
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL over xGMI: one process per GPU; replaces mpi4py Allreduce (recon/sirt_mpi.py:68,103,110;
// recon/cgls_mpi.py:55,75-76,98,107).
// ------------------------------------------------------------------------------------------------
#define TOMO_NCCL(ctx, call)                                                                          \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return tomo_fail((ctx), TOMO_ERR_RCCL, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)

extern "C" int tomo_comm_get_unique_id(void *h_id128)
{
    if (!h_id128) return tomo_fail(nullptr, TOMO_ERR_ARG, "null id");
    static_assert(sizeof(ncclUniqueId) <= TOMO_COMM_ID_BYTES, "id size");
    ncclUniqueId id;
    TOMO_NCCL(nullptr, ncclGetUniqueId(&id));
    memset(h_id128, 0, TOMO_COMM_ID_BYTES);
    memcpy(h_id128, &id, sizeof(id));
    return TOMO_OK;
}

extern "C" int tomo_comm_init(tomo_ctx *ctx, const void *h_id128, int n_ranks, int rank)
{
    if (!ctx || !h_id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    if (ctx->comm) return tomo_fail(ctx, TOMO_ERR_STATE, "comm already initialised");
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    TOMO_NCCL(ctx, ncclCommInitRank(&ctx->comm, n_ranks, id, rank));
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return TOMO_OK;
}

extern "C" int tomo_comm_destroy(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    if (ctx->comm) {
        if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamSynchronize(ctx->stream);
        ncclCommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return TOMO_OK;
}

extern "C" int tomo_allreduce_sum_f32(tomo_ctx *ctx, float *d_buf, int64_t n)
{
    TomoRange roctx_range("tomo_allreduce_sum_f32");
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    if (!ctx->comm) return ctx->n_ranks == 1 ? TOMO_OK : tomo_fail(ctx, TOMO_ERR_STATE, "comm not initialised");
    tomo_prof_begin(ctx, "allreduce_f32");
    ncclResult_t r = ncclAllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, ctx->comm, ctx->stream);
    tomo_prof_end(ctx);
    if (r != ncclSuccess) return tomo_fail(ctx, TOMO_ERR_RCCL, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return TOMO_OK;
}

// Test hook (option comm_test_poison_us, tests only): before an asynchronous collective starts, the communication stream doubles
// the collective's buffer, idles for that many microseconds and halves it again (exact in float32).  A compute-stream kernel that
// reads or writes the buffer without having waited for the collective then sees doubled values -- so a ONE-rank run, whose
// collectives change nothing, can tell a missing wait from a correct one (tests/test_gpu_dist.py).
__global__ void k_comm_test_scale(float *__restrict__ v, int64_t n, float f)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] *= f;
}
__global__ void k_comm_test_idle(long long ticks)      // one wave; wall_clock64 ticks at 100 MHz
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// Measurement hook (options comm_test_copy_eighths / comm_test_copy_wgs; tools/contention_probe.py, profiles/round6_contention_probe.md): on a
// ONE-rank communicator the collectives move nothing, so the kernels of a rank's share have the chip to themselves -- unlike at P = 8, where
// a ring reduce-scatter / all-gather of a slab reads and writes (P-1)/P of its bytes in this GPU's HBM through RCCL's own work-groups while
// the back-projection of the next slab runs.  With comm_test_copy_eighths = k every asynchronous collective also copies k/8 of the buffer
// it touches to a scratch buffer on the communication stream, inside its profile record: by hipMemcpyAsync (comm_test_copy_wgs = 0: the
// copy engine / blit kernel at full rate) or by a copy kernel of comm_test_copy_wgs work-groups (RCCL-like: a fixed, small number of
// work-groups that hold CUs and stream at a link-like rate).
__global__ __launch_bounds__(256) void k_comm_test_copy(float4 *__restrict__ dst, const float4 *__restrict__ src, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// The three asynchronous collectives share everything but the RCCL call: they run on the communication stream after everything
// queued so far on the compute stream, and leave an event in one of two FIFO queues (0: reductions -- all-reduce, reduce-scatter;
// 1: all-gathers) that tomo_comm_wait_next / tomo_comm_wait_next_gather consume in issue order.
enum { COLL_ALLREDUCE = 0, COLL_REDUCE_SCATTER = 1, COLL_ALLGATHER = 2 };
static int comm_async(tomo_ctx *ctx, int kind, float *d_buf, int64_t n)
{
    TomoRange roctx_range(kind == COLL_ALLREDUCE ? "tomo_allreduce_sum_f32_async" : kind == COLL_REDUCE_SCATTER ? "tomo_reduce_scatter_sum_f32_async" : "tomo_allgather_f32_async");
    if (!ctx || n < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    if (!ctx->comm) return ctx->n_ranks == 1 ? TOMO_OK : tomo_fail(ctx, TOMO_ERR_STATE, "comm not initialised");
    if (!ctx->comm_stream) {
        TOMO_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        TOMO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_compute, hipEventDisableTiming));
        TOMO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming));
    }
    TOMO_HIP(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));
    TOMO_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_compute, 0));
    const int64_t n_all = kind == COLL_ALLREDUCE ? n : n * ctx->n_ranks;     // the whole buffer the collective touches
    float *mine = d_buf + (kind == COLL_ALLREDUCE ? 0 : n * ctx->rank);      // in-place forms: this rank's piece of it
    if (ctx->comm_test_poison_us > 0 && n_all > 0) {
        const int grid = (int)std::min<int64_t>((n_all + 255) / 256, 2048);
        hipLaunchKernelGGL(k_comm_test_scale, dim3(grid), dim3(256), 0, ctx->comm_stream, d_buf, n_all, 2.0f);
        hipLaunchKernelGGL(k_comm_test_idle, dim3(1), dim3(64), 0, ctx->comm_stream, (long long)ctx->comm_test_poison_us * 100);
        hipLaunchKernelGGL(k_comm_test_scale, dim3(grid), dim3(256), 0, ctx->comm_stream, d_buf, n_all, 0.5f);
        TOMO_HIP(ctx, hipGetLastError());
    }
    // profile records on the communication stream: they hold the collective's own duration (peers' arrival skew included) whether
    // or not the compute stream ever waits for it -- "comm_join_wait" is the exposed part
    static const char *names[3] = {"allreduce_f32", "reduce_scatter_f32", "allgather_f32"};
    tomo_prof_begin_on(ctx, names[kind], ctx->comm_stream);
    ncclResult_t r = ncclSuccess;
    if (n > 0) {
        if (kind == COLL_ALLREDUCE) r = ncclAllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, ctx->comm, ctx->comm_stream);
        else if (kind == COLL_REDUCE_SCATTER) r = ncclReduceScatter(d_buf, mine, (size_t)n, ncclFloat32, ncclSum, ctx->comm, ctx->comm_stream);
        else r = ncclAllGather(mine, d_buf, (size_t)n, ncclFloat32, ctx->comm, ctx->comm_stream);
    }
    if (ctx->comm_test_copy_eighths > 0 && n_all > 0) {
        const size_t bytes = ((size_t)n_all * 4 * (size_t)ctx->comm_test_copy_eighths / 8) & ~(size_t)15;
        if (bytes > ctx->comm_scratch_bytes) {
            TOMO_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
            if (ctx->d_comm_scratch) (void)hipFree(ctx->d_comm_scratch);
            ctx->d_comm_scratch = nullptr;
            ctx->comm_scratch_bytes = 0;
            TOMO_HIP(ctx, hipMalloc(&ctx->d_comm_scratch, bytes));
            ctx->comm_scratch_bytes = bytes;
        }
        if (bytes && ((uintptr_t)d_buf & 15) == 0 && ctx->comm_test_copy_wgs > 0)
            hipLaunchKernelGGL(k_comm_test_copy, dim3(ctx->comm_test_copy_wgs), dim3(256), 0, ctx->comm_stream, (float4 *)ctx->d_comm_scratch, (const float4 *)d_buf,
                               (int64_t)(bytes / 16));
        else if (bytes)
            TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_comm_scratch, d_buf, bytes, hipMemcpyDeviceToDevice, ctx->comm_stream));
    }
    tomo_prof_end_on(ctx, ctx->comm_stream);
    if (r != ncclSuccess) return tomo_fail(ctx, TOMO_ERR_RCCL, std::string(names[kind]) + ": " + ncclGetErrorString(r));
    TOMO_HIP(ctx, hipEventRecord(ctx->ev_comm, ctx->comm_stream));
    ctx->comm_pending = true;
    // ... and an event of its own, for callers that consume the collectives one by one
    hipEvent_t e = nullptr;
    if (!ctx->comm_ev_pool.empty()) { e = ctx->comm_ev_pool.back(); ctx->comm_ev_pool.pop_back(); }
    else TOMO_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    TOMO_HIP(ctx, hipEventRecord(e, ctx->comm_stream));
    (kind == COLL_ALLGATHER ? ctx->comm_done_g : ctx->comm_done).push_back(e);
    return TOMO_OK;
}

extern "C" int tomo_allreduce_sum_f32_async(tomo_ctx *ctx, float *d_buf, int64_t n) { return comm_async(ctx, COLL_ALLREDUCE, d_buf, n); }
extern "C" int tomo_reduce_scatter_sum_f32_async(tomo_ctx *ctx, float *d_buf, int64_t n_per_rank) { return comm_async(ctx, COLL_REDUCE_SCATTER, d_buf, n_per_rank); }
extern "C" int tomo_allgather_f32_async(tomo_ctx *ctx, float *d_buf, int64_t n_per_rank) { return comm_async(ctx, COLL_ALLGATHER, d_buf, n_per_rank); }

// The compute stream waits for the OLDEST asynchronous collective of a queue it has not waited for yet (issue order).  With this a
// caller consumes the x slabs of a pipelined update one by one -- slab s is updated, and the next iteration's forward projection
// of that slab started, while the collectives of the later slabs are still on the links (recon/sirt_mpi.py).  No-op when none is pending.
static int comm_wait_next(tomo_ctx *ctx, std::deque<hipEvent_t> &q)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    if (q.empty()) return TOMO_OK;
    hipEvent_t e = q.front();
    q.pop_front();
    tomo_prof_begin(ctx, "comm_join_wait");                  // the same record as tomo_comm_join: exposed communication, summed per step
    hipError_t r = hipStreamWaitEvent(ctx->stream, e, 0);
    tomo_prof_end(ctx);
    ctx->comm_ev_pool.push_back(e);                          // a wait already queued keeps the state the event had when it was queued
    if (r != hipSuccess) return tomo_fail(ctx, TOMO_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(r));
    if (ctx->comm_done.empty() && ctx->comm_done_g.empty()) ctx->comm_pending = false;
    return TOMO_OK;
}
extern "C" int tomo_comm_wait_next(tomo_ctx *ctx) { return ctx ? comm_wait_next(ctx, ctx->comm_done) : tomo_fail(ctx, TOMO_ERR_ARG, "null ctx"); }
extern "C" int tomo_comm_wait_next_gather(tomo_ctx *ctx) { return ctx ? comm_wait_next(ctx, ctx->comm_done_g) : tomo_fail(ctx, TOMO_ERR_ARG, "null ctx"); }

extern "C" int tomo_comm_join(tomo_ctx *ctx)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "null ctx");
    if (ctx->comm_pending) {
        // "comm_join_wait": time the compute stream sits between the end of its own work and the end of the last asynchronous
        // all-reduce = the communication the pipeline failed to hide
        tomo_prof_begin(ctx, "comm_join_wait");
        TOMO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_comm, 0));   // events on one stream complete in order: the last covers all
        tomo_prof_end(ctx);
        ctx->comm_pending = false;
    }
    for (auto e : ctx->comm_done) ctx->comm_ev_pool.push_back(e);
    ctx->comm_done.clear();
    for (auto e : ctx->comm_done_g) ctx->comm_ev_pool.push_back(e);
    ctx->comm_done_g.clear();
    return TOMO_OK;
}

// ---- device accumulators (include/tomo.h: tomo_acc_*): scalars of a solver iteration with one host synchronisation
static int acc_ensure(tomo_ctx *ctx)
{
    TOMO_HIP(ctx, hipSetDevice(ctx->device));            // every tomo_acc_* entry point comes through here: HIP's current device is per thread
    if (ctx->d_acc && ctx->h_acc) return TOMO_OK;
    // both or neither: a failed second allocation must not leave a half-made pair that the next call takes for a whole one
    if (ctx->d_acc) { (void)hipFree(ctx->d_acc); ctx->d_acc = nullptr; }
    if (ctx->h_acc) { (void)hipHostFree(ctx->h_acc); ctx->h_acc = nullptr; }
    TOMO_HIP(ctx, hipMalloc((void **)&ctx->d_acc, TOMO_N_ACC * sizeof(double)));
    if (hipHostMalloc((void **)&ctx->h_acc, TOMO_N_ACC * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(ctx->d_acc);
        ctx->d_acc = ctx->h_acc = nullptr;
        return tomo_fail(ctx, TOMO_ERR_HIP, "tomo_acc: no pinned host memory for the accumulators' mirror");
    }
    TOMO_HIP(ctx, hipMemsetAsync(ctx->d_acc, 0, TOMO_N_ACC * sizeof(double), ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_acc_zero(tomo_ctx *ctx, int slot0, int n)
{
    if (!ctx || slot0 < 0 || n < 0 || slot0 + n > TOMO_N_ACC) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_acc_zero: bad slots");
    int rc = acc_ensure(ctx);
    if (rc) return rc;
    if (n) TOMO_HIP(ctx, hipMemsetAsync(ctx->d_acc + slot0, 0, n * sizeof(double), ctx->stream));
    return TOMO_OK;
}

extern "C" int tomo_vec_dot_acc(tomo_ctx *ctx, const float *a, const float *b, int64_t n, int diff, int slot)
{
    if (!ctx || slot < 0 || slot >= TOMO_N_ACC || n < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_vec_dot_acc: bad args");
    int rc = acc_ensure(ctx);
    if (rc) return rc;
    if (n == 0) return TOMO_OK;
    if (!a || !b) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_vec_dot_acc: null operand");
    TOMO_LAUNCH(ctx, "k_dot", k_dot, dim3(vec_grid(n)), dim3(256), 0, a, b, n, diff ? 1 : 0, ctx->d_acc + slot);
    return TOMO_OK;
}

extern "C" int tomo_acc_fetch(tomo_ctx *ctx, int slot0, int n, int allreduce, double *h_out)
{
    if (!ctx || !h_out || slot0 < 0 || n < 0 || slot0 + n > TOMO_N_ACC) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_acc_fetch: bad args");
    int rc = acc_ensure(ctx);
    if (rc) return rc;
    if (n == 0) return TOMO_OK;
    // a sum over the ranks was asked for: without a communicator that is only right on one rank (as tomo_allreduce_sum_f32 rules)
    if (allreduce && !ctx->comm && ctx->n_ranks > 1) return tomo_fail(ctx, TOMO_ERR_STATE, "tomo_acc_fetch: comm not initialised");
    if (allreduce && ctx->comm) {
        tomo_prof_begin(ctx, "allreduce_scalars");
        ncclResult_t r = ncclAllReduce(ctx->d_acc + slot0, ctx->d_acc + slot0, (size_t)n, ncclFloat64, ncclSum, ctx->comm, ctx->stream);
        tomo_prof_end(ctx);
        if (r != ncclSuccess) return tomo_fail(ctx, TOMO_ERR_RCCL, std::string("ncclAllReduce (accumulators): ") + ncclGetErrorString(r));
    }
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_acc + slot0, ctx->d_acc + slot0, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) h_out[i] = ctx->h_acc[slot0 + i];
    return TOMO_OK;
}

static int host_allreduce_f64(tomo_ctx *ctx, double *h_vals, int n, ncclRedOp_t op)
{
    if (!ctx || !h_vals || n < 0) return tomo_fail(ctx, TOMO_ERR_ARG, "bad args");
    if (!ctx->comm || n == 0) return TOMO_OK;
    int rc = tomo_ensure_red(ctx, (size_t)n + 8);
    if (rc) return rc;
    memcpy(ctx->h_red + 8, h_vals, sizeof(double) * n);
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->d_red + 8, ctx->h_red + 8, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    TOMO_NCCL(ctx, ncclAllReduce(ctx->d_red + 8, ctx->d_red + 8, (size_t)n, ncclFloat64, op, ctx->comm, ctx->stream));
    TOMO_HIP(ctx, hipMemcpyAsync(ctx->h_red + 8, ctx->d_red + 8, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(h_vals, ctx->h_red + 8, sizeof(double) * n);
    return TOMO_OK;
}

extern "C" int tomo_allreduce_sum_f64_host(tomo_ctx *ctx, double *h_vals, int n) { return host_allreduce_f64(ctx, h_vals, n, ncclSum); }
extern "C" int tomo_allreduce_max_f64_host(tomo_ctx *ctx, double *h_vals, int n) { return host_allreduce_f64(ctx, h_vals, n, ncclMax); }
