// micro-benchmark: LDS accumulate variants (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int MODE> __global__ __launch_bounds__(512) void k(float *out, int stride)
{
    __shared__ float acc[17 * 17 * 61];
    for (int e = threadIdx.x; e < 17 * 17 * 61; e += 512) acc[e] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base = (wv * 17 + 3) * 61 + (lane * stride) % 60;
    float v = 1.0f + lane;
    for (int it = 0; it < N_IT; ++it) {
        int a = base + (it & 7) * 61;
        if (MODE == 0) { atomicAdd(&acc[a], v); atomicAdd(&acc[a + 1], v); }
        else if (MODE == 1) { atomicAdd((unsigned *)&acc[a], (unsigned)it); atomicAdd((unsigned *)&acc[a + 1], (unsigned)it); }
        else if (MODE == 2) { acc[a] += v; acc[a + 1] += v; }
        else if (MODE == 3) { unsigned long long *p = (unsigned long long *)&acc[(a & ~1)]; atomicAdd(p, (unsigned long long)it); atomicAdd(p + 1, (unsigned long long)it); }
        else if (MODE == 4) { __hip_atomic_fetch_add(&acc[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&acc[a + 1], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[base];
}
template <int MODE> void run(const char *name, int stride)
{
    float *d; hipMalloc(&d, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<512, 512>>>(d, stride); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<512, 512>>>(d, stride); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveops = 512.0 * 8 * N_IT * 2;   // wave-level LDS ops
    // 512 blocks over 256 CUs, 2 per CU concurrently: per-CU wave-ops = waveops/256
    printf("%-28s stride %d: %8.3f ms  -> %.1f cycles(2.1GHz) per wave-op per CU\n", name, stride, ms, ms * 1e-3 * 2.1e9 / (waveops / 256));
    hipFree(d);
}
int main()
{
    for (int s : {1, 2}) {
        run<0>("ds_add_f32 (atomicAdd)", s);
        run<4>("hip_atomic wg-scope f32", s);
        run<1>("ds_add_u32", s);
        run<3>("ds_add_u64", s);
        run<2>("plain read+add+write", s);
    }
    return 0;
}
