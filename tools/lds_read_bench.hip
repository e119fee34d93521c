// micro-benchmark (development aid): LDS read cost per wave instruction for the access shapes of the flat forward kernel:
// per-lane reads of an image plane (ds_read_b32 / ds_read2st64_b32) and wave-uniform (broadcast) reads of a small table
// (ds_read_b32 / b64 / b128).  16 waves per CU hammer the LDS; cycles are per wave instruction per CU at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(512) void k(float *out, int zero)
{
    __shared__ float img[17 * 17 * 64];
    __shared__ float4 tab[64];
    for (int e = threadIdx.x; e < 17 * 17 * 64; e += 512) img[e] = 1.f;
    if (threadIdx.x < 64) tab[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < N_IT; ++it) {
        const int cell = ((it * 7) & 255) * 64 + zero;              // wave-uniform cell, changes every iteration
        const int t = (it & 31) + zero;
        if (MODE == 0) acc += img[cell + lane];                                                  // ds_read_b32 per lane
        else if (MODE == 1) { acc += img[cell + lane] + img[cell + 64 + lane]; }                 // ds_read2st64_b32
        else if (MODE == 2) acc += ((const float *)tab)[t];                                      // broadcast b32
        else if (MODE == 3) { f2 v = ((const f2 *)tab)[t]; acc += v.x + v.y; }                   // broadcast b64
        else if (MODE == 4) { float4 v = tab[t]; acc += v.x + v.y + v.z + v.w; }                 // broadcast b128
        else if (MODE == 5) { acc += img[cell + lane] + img[cell + 64 + lane] + img[cell + 1088 + lane] + img[cell + 1152 + lane]; }   // 2 x read2st64
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<512, 512>>>(out, 0); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<512, 512>>>(out, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveops = 512.0 * 8 * N_IT;
    printf("%-40s %8.3f ms  -> %6.2f cycles per wave-iteration per CU\n", name, ms, ms * 1e-3 * 2.4e9 / (waveops / 256));
}
int main()
{
    float *out; hipMalloc(&out, 512 * 512 * 4);
    run<0>("ds_read_b32 per lane", out);
    run<1>("ds_read2st64_b32 per lane (2 dwords)", out);
    run<5>("2 x ds_read2st64_b32 per lane (4 dwords)", out);
    run<2>("broadcast ds_read_b32", out);
    run<3>("broadcast ds_read_b64", out);
    run<4>("broadcast ds_read_b128", out);
    return 0;
}
