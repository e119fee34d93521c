// micro-benchmark (development aid, round 6): does a ds_read2_b32 cost the same wherever in the 160 KB LDS of gfx950 its address lies?
// One 16-wave work-group per CU (139 KB of LDS, as k_tile<true> after the A/B-row experiment), every lane reads (z, z + 1) pairs at
// base + 4 * lane from a window of the image: [0, 64 KB), [64 KB, 128 KB), [75 KB, 139 KB), or the whole 139 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 8192
#define IMG (34816 + 64)
typedef __attribute__((address_space(3))) float lds_f;
__global__ __launch_bounds__(1024) void k(float *out, unsigned lo, unsigned span, int four)
{
    __shared__ float img[IMG];
    for (int e = threadIdx.x; e < IMG; e += 1024) img[e] = 1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const lds_f *p = (const lds_f *)img;
    float acc = 0.f;
    unsigned r = wv * 977u;
    for (int it = 0; it < N_IT; ++it) {
        r = r * 1664525u + 1013904223u;
        const unsigned row = lo + ((r >> 8) % span);            // a 64-dword row of the window (wave-uniform)
        const lds_f *q = p + row * 64 + lane;
        float a = q[0] + q[1];
        if (four) a += q[64] + q[65] + q[2176] + q[2177] + q[2240] + q[2241];
        acc += a;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
static void run(const char *name, float *out, unsigned lo, unsigned span, int four)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<1024, 1024>>>(out, lo, span, four); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<1024, 1024>>>(out, lo, span, four); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveops = 1024.0 * 16 * N_IT;
    printf("%-58s %8.3f ms  -> %6.2f cycles per wave-iteration per CU\n", name, ms, ms * 1e-3 * 2.4e9 / (waveops / 256));
}
int main()
{
    float *out; hipMalloc(&out, 1024 * 1024 * 4);
    for (int four = 0; four < 2; ++four) {
        printf(four ? "-- 4 x ds_read2_b32 per iteration (the sample's four corner pairs)\n" : "-- 1 x ds_read2_b32 per iteration\n");
        run("rows in [0, 64 KB)", out, 0, 256 - 36, four);
        run("rows in [64 KB, 128 KB)", out, 256, 256 - 36, four);
        run("rows in [75 KB, 130 KB)", out, 300, 220 - 36, four);
        run("rows anywhere in the 139 KB", out, 0, 544 - 36, four);
        run("rows in [0, 74 KB) (the old image)", out, 0, 289 - 36, four);
    }
    return 0;
}
