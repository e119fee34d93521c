// tomo_csr.hip -- the assembled CSR of the reference's projection_matrix, built ON THE DEVICE:
//   utilities/projection_operators.py:54-76   _forward_ray (per-projection triplets of src/ray_wt_grad.f90:1-92, weights cast to
//                                             `precision`, detector index + iproj * n_det) -> optional voxel-mask filter (:60-70) ->
//                                             coo_matrix -> csr_matrix (duplicates summed, explicit zeros kept, indices sorted)
// Until round 3 the triplets were emitted on the device and scipy did the rest on the host (sort + duplicate merge: the 280 s of the
// reference's own build at 128^3 x 64 are mostly that).  Here: count -> scan -> fill of (row << 32 | column, weight) pairs for ALL
// projections, rocPRIM radix sort by key (stable: duplicates keep the reference's emission order), reduce-by-key (the duplicate sums,
// in `precision` like scipy's; a parallel reduction, so the last bit of a sum of 2-4 duplicates may differ from scipy's sequential one), row pointers by a histogram + scan.  Two calls: tomo_csr_assemble builds and keeps the result in the
// context and says how many entries it has, tomo_csr_fetch copies it into the caller's arrays and frees it.
// Small volumes only (8 slots per sample; N <= 128), like the reference's matrix.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_scan.hpp>

#include "tomo_ctx.h"

int tomo_upload_projc_for_csr(tomo_ctx *ctx, const double *h_poses, int n_proj, ProjC **d_pc);      // tomo_project.hip

namespace {

// pass 0 (FILL = false): in-bounds, unmasked corners per ray; pass 1: the pairs, ray-major / sample / corner within a projection
template <bool FILL, typename V>
__global__ __launch_bounds__(256) void k_csr_triplets(const ProjC *__restrict__ pcs, int n_proj, TomoGeomC g, const float *__restrict__ mask,
                                                      const int64_t *__restrict__ offsets, int64_t *__restrict__ counts, uint64_t *__restrict__ keys,
                                                      V *__restrict__ vals)
{
    const int n_det = g.ndx * g.ndz;
    const int64_t gr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // global ray = ip * n_det + r
    if (gr >= (int64_t)n_proj * n_det) return;
    const int ip = (int)(gr / n_det), r = (int)(gr - (int64_t)ip * n_det);
    const int ix = r / g.ndz, iz = r - ix * g.ndz;
    const ProjC &c = pcs[ip];
    double b[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) b[a] = c.p0[a] + (double)ix * c.u[a] + (double)iz * c.w[a];
    int j0, j1;
    tomo_ray_range(b, c.d, c.n, g.nx, g.ny, g.nz, j0, j1);
    int64_t o = FILL ? offsets[gr] : 0, cnt = 0;
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
            if (x < 0 || x >= g.nx || y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;      // src/ray_wt_grad.f90:35-89
            const uint32_t col = (uint32_t)((x * g.ny + y) * g.nz + z);
            if (mask && mask[col] == 0.f) continue;                                              // projection_operators.py:61-70
            if (FILL) {
                const double wx = (k >> 2) ? 1.0 - wf[0] : wf[0], wy = ((k >> 1) & 1) ? 1.0 - wf[1] : wf[1], wz = (k & 1) ? 1.0 - wf[2] : wf[2];
                keys[o] = ((uint64_t)gr << 32) | col;
                vals[o] = (V)(wx * wy * wz);                  // float64 weight (Fortran), cast to `precision` (:106)
                ++o;
            }
            ++cnt;
        }
    }
    if (!FILL) counts[gr] = cnt;
}

__global__ __launch_bounds__(256) void k_csr_rows(const uint64_t *__restrict__ ukeys, int64_t n_unique, int32_t *__restrict__ indices,
                                                  unsigned long long *__restrict__ row_counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_unique) return;
    const uint64_t k = ukeys[i];
    indices[i] = (int32_t)(uint32_t)k;
    atomicAdd(&row_counts[k >> 32], 1ull);
}

template <typename V>
__global__ __launch_bounds__(256) void k_csr_zero(V *__restrict__ v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (V)0;
}

struct Dev {      // scoped device allocation
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    template <typename T> T *as() { return (T *)p; }
};

#define CSR_ALLOC(buf, bytes) TOMO_HIP(ctx, hipMalloc(&(buf).p, (bytes) ? (size_t)(bytes) : 1))

template <typename V>
int assemble(tomo_ctx *ctx, const ProjC *d_pc, int n_proj, const float *d_mask, int64_t *h_nnz)
{
    const TomoGeomC &g = ctx->g;
    const int64_t n_rays = (int64_t)n_proj * g.ndx * g.ndz;
    const dim3 grid((unsigned)((n_rays + 255) / 256));
    hipStream_t st = ctx->stream;
    Dev counts, offs, tmp, keys_a, keys_b, vals_a, vals_b, ukeys, uvals, n_unique_d;
    CSR_ALLOC(counts, sizeof(int64_t) * (n_rays + 1));
    CSR_ALLOC(offs, sizeof(int64_t) * (n_rays + 1));
    const float *mask = d_mask;
    int64_t total = 0;
    bool all_masked = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        TOMO_HIP(ctx, hipMemsetAsync(counts.p, 0, sizeof(int64_t) * (n_rays + 1), st));
        hipLaunchKernelGGL((k_csr_triplets<false, V>), grid, dim3(256), 0, st, d_pc, n_proj, g, mask, (const int64_t *)nullptr, counts.as<int64_t>(),
                           (uint64_t *)nullptr, (V *)nullptr);
        size_t tb = 0;      // exclusive scan over n_rays + 1 counts: the last offset is the total
        TOMO_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, counts.as<int64_t>(), offs.as<int64_t>(), (int64_t)0, (size_t)(n_rays + 1), rocprim::plus<int64_t>(), st));
        Dev scan_tmp;
        CSR_ALLOC(scan_tmp, tb);
        TOMO_HIP(ctx, rocprim::exclusive_scan(scan_tmp.p, tb, counts.as<int64_t>(), offs.as<int64_t>(), (int64_t)0, (size_t)(n_rays + 1), rocprim::plus<int64_t>(), st));
        TOMO_HIP(ctx, hipMemcpyAsync(&total, offs.as<int64_t>() + n_rays, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        TOMO_HIP(ctx, hipStreamSynchronize(st));
        if (total == 0 && mask) {      // "entire object is masked": the reference keeps EVERY entry, with weight 0 (:63-65)
            all_masked = true;
            mask = nullptr;
            continue;
        }
        break;
    }
    if (total >= ((int64_t)1 << 31)) return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_csr_assemble: 2^31 or more triplets; keep the operator matrix-free");
    int64_t n_unique = 0;
    const int64_t n_rows = n_rays;
    Dev indices, indptr, row_counts;
    CSR_ALLOC(indptr, sizeof(int64_t) * (n_rows + 1));
    if (total > 0) {
        CSR_ALLOC(keys_a, sizeof(uint64_t) * total); CSR_ALLOC(keys_b, sizeof(uint64_t) * total);
        CSR_ALLOC(vals_a, sizeof(V) * total); CSR_ALLOC(vals_b, sizeof(V) * total);
        hipLaunchKernelGGL((k_csr_triplets<true, V>), grid, dim3(256), 0, st, d_pc, n_proj, g, mask, (const int64_t *)offs.p, (int64_t *)nullptr,
                           keys_a.as<uint64_t>(), vals_a.as<V>());
        if (all_masked) hipLaunchKernelGGL((k_csr_zero<V>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, vals_a.as<V>(), total);
        // rows < 2^31 and columns < 2^31: sort on the bits that can be set
        unsigned row_bits = 1;
        while (((int64_t)1 << row_bits) < n_rows) ++row_bits;
        size_t tb = 0;
        TOMO_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<V>(), vals_b.as<V>(), (size_t)total, 0u, 32u + row_bits, st));
        CSR_ALLOC(tmp, tb);
        TOMO_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<V>(), vals_b.as<V>(), (size_t)total, 0u, 32u + row_bits, st));
        // duplicates (the same voxel reached from several samples of a ray) summed in `precision`
        CSR_ALLOC(n_unique_d, sizeof(int64_t));
        size_t tb2 = 0;
        TOMO_HIP(ctx, rocprim::reduce_by_key(nullptr, tb2, keys_b.as<uint64_t>(), vals_b.as<V>(), (size_t)total, keys_a.as<uint64_t>(), vals_a.as<V>(),
                                             n_unique_d.as<int64_t>(), rocprim::plus<V>(), rocprim::equal_to<uint64_t>(), st));
        Dev tmp2;
        CSR_ALLOC(tmp2, tb2);
        TOMO_HIP(ctx, rocprim::reduce_by_key(tmp2.p, tb2, keys_b.as<uint64_t>(), vals_b.as<V>(), (size_t)total, keys_a.as<uint64_t>(), vals_a.as<V>(),
                                             n_unique_d.as<int64_t>(), rocprim::plus<V>(), rocprim::equal_to<uint64_t>(), st));
        TOMO_HIP(ctx, hipMemcpyAsync(&n_unique, n_unique_d.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        TOMO_HIP(ctx, hipStreamSynchronize(st));
        CSR_ALLOC(indices, sizeof(int32_t) * n_unique);
        CSR_ALLOC(row_counts, sizeof(int64_t) * (n_rows + 1));
        TOMO_HIP(ctx, hipMemsetAsync(row_counts.p, 0, sizeof(int64_t) * (n_rows + 1), st));
        hipLaunchKernelGGL(k_csr_rows, dim3((unsigned)((n_unique + 255) / 256)), dim3(256), 0, st, (const uint64_t *)keys_a.p, n_unique, indices.as<int32_t>(),
                           (unsigned long long *)row_counts.p);
        size_t tb3 = 0;
        TOMO_HIP(ctx, rocprim::exclusive_scan(nullptr, tb3, row_counts.as<int64_t>(), indptr.as<int64_t>(), (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
        Dev tmp3;
        CSR_ALLOC(tmp3, tb3);
        TOMO_HIP(ctx, rocprim::exclusive_scan(tmp3.p, tb3, row_counts.as<int64_t>(), indptr.as<int64_t>(), (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
        TOMO_HIP(ctx, hipStreamSynchronize(st));
        TOMO_HIP(ctx, hipGetLastError());
    } else {
        TOMO_HIP(ctx, hipMemsetAsync(indptr.p, 0, sizeof(int64_t) * (n_rows + 1), st));
        TOMO_HIP(ctx, hipStreamSynchronize(st));
    }
    // keep the result in the context until tomo_csr_fetch (ownership moves out of the scoped holders)
    ctx->csr_data = vals_a.p; vals_a.p = nullptr;
    ctx->csr_indices = indices.p; indices.p = nullptr;
    ctx->csr_indptr = indptr.p; indptr.p = nullptr;
    ctx->csr_nnz = n_unique;
    ctx->csr_rows = n_rows;
    ctx->csr_value_bytes = (int)sizeof(V);
    *h_nnz = n_unique;
    return TOMO_OK;
}

void csr_release(tomo_ctx *ctx)
{
    if (ctx->csr_data) (void)hipFree(ctx->csr_data);
    if (ctx->csr_indices) (void)hipFree(ctx->csr_indices);
    if (ctx->csr_indptr) (void)hipFree(ctx->csr_indptr);
    ctx->csr_data = ctx->csr_indices = ctx->csr_indptr = nullptr;
    ctx->csr_nnz = ctx->csr_rows = 0;
}

}  // namespace

void tomo_csr_release(tomo_ctx *ctx) { csr_release(ctx); }

extern "C" int tomo_csr_assemble(tomo_ctx *ctx, const double *h_poses, int n_proj, const float *d_mask, int precision_bits, int64_t *h_nnz)
{
    TOMO_NEED_GEOM(ctx);
    if (!h_poses || !h_nnz || n_proj < 1 || (precision_bits != 32 && precision_bits != 64)) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_csr_assemble: bad args");
    const TomoGeomC &g = ctx->g;
    if ((size_t)g.nx * g.ny * g.nz >= ((size_t)1 << 31) || (int64_t)n_proj * g.ndx * g.ndz >= ((int64_t)1 << 31))
        return tomo_fail(ctx, TOMO_ERR_UNSUPPORTED, "tomo_csr_assemble: int32 row / column indices (as the reference's matrix)");
    csr_release(ctx);
    ProjC *d_pc = nullptr;
    int rc = tomo_upload_projc_for_csr(ctx, h_poses, n_proj, &d_pc);
    if (rc) return rc;
    tomo_prof_begin(ctx, "csr_assemble");
    rc = precision_bits == 32 ? assemble<float>(ctx, d_pc, n_proj, d_mask, h_nnz) : assemble<double>(ctx, d_pc, n_proj, d_mask, h_nnz);
    tomo_prof_end(ctx);
    return rc;
}

extern "C" int tomo_csr_fetch(tomo_ctx *ctx, void *h_data, int32_t *h_indices, int64_t *h_indptr)
{
    if (!ctx) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_csr_fetch: null ctx");
    if (!h_data && !h_indices && !h_indptr) {      // discard: the caller decided against the download (max_nnz is checked BEFORE host arrays exist)
        TOMO_HIP(ctx, hipSetDevice(ctx->device));
        TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        csr_release(ctx);
        return TOMO_OK;
    }
    if (!h_indptr) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_csr_fetch: bad args");
    if (!ctx->csr_indptr) return tomo_fail(ctx, TOMO_ERR_STATE, "tomo_csr_fetch: nothing assembled");
    TOMO_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->csr_nnz > 0) {
        if (!h_data || !h_indices) return tomo_fail(ctx, TOMO_ERR_ARG, "tomo_csr_fetch: null output");
        TOMO_HIP(ctx, hipMemcpyAsync(h_data, ctx->csr_data, (size_t)ctx->csr_nnz * ctx->csr_value_bytes, hipMemcpyDeviceToHost, ctx->stream));
        TOMO_HIP(ctx, hipMemcpyAsync(h_indices, ctx->csr_indices, (size_t)ctx->csr_nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    TOMO_HIP(ctx, hipMemcpyAsync(h_indptr, ctx->csr_indptr, (size_t)(ctx->csr_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    TOMO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    csr_release(ctx);
    return TOMO_OK;
}
