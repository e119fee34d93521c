// micro-benchmark (development aid): cost in the texture-address / L1 pipeline of one wave-wide gather whose lanes read
// consecutive z cells of a volume row -- the access shape of the ray-driven projection/gradient kernels -- as a function
// of the load width.  All addresses hit L1 (8 rows reused), so this is the address-processing rate, not bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
#define ROW 1028
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k(const float *__restrict__ buf, float *out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float *base = buf + (size_t)(blockIdx.x % 64) * 16 * ROW + wv * 2 * ROW;
    float acc = 0.f;
    for (int it = 0; it < N_IT; ++it) {
        const float *row = base + (it & 7) * ROW + (it & 3);      // unaligned start like a real ray
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *q = row + u * 8 * ROW;
            if (MODE == 0) acc += q[lane];                                                 // dword, stride 4 B
            else if (MODE == 1) { f2 v = *(const f2 *)(q + lane); acc += v.x + v.y; }      // dwordx2 overlapping, stride 4 B
            else if (MODE == 2) { f2 v = *(const f2 *)(q + 2 * lane); acc += v.x + v.y; }  // dwordx2, stride 8 B
            else if (MODE == 3) { f4 v = *(const f4 *)(q + 4 * lane); acc += v.x + v.y + v.z + v.w; }   // dwordx4, stride 16 B
            else if (MODE == 4) { acc += q[lane] + q[lane + 1]; }                          // two dword loads
            else if (MODE == 5) { f4 v = *(const f4 *)(q + lane); acc += v.x + v.w; }      // dwordx4 overlapping, stride 4 B
            else if (MODE == 6) { if (lane < 17) { f4 v = *(const f4 *)(q + 4 * lane); acc += v.x + v.y + v.z + v.w; } }   // 17 lanes x 16 B = one row segment
            else if (MODE == 7) { if (lane < 33) { f2 v = *(const f2 *)(q + 2 * lane); acc += v.x + v.y; } }               // 33 lanes x 8 B
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name, const float *buf, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    k<MODE><<<grid, 256>>>(buf, out); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(buf, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gathers = (double)grid * 4 * N_IT * 4 * (MODE == 4 ? 1 : 1);   // wave-level "corner-pair fetches"
    printf("%-44s %8.3f ms  -> %6.1f cycles (2.4 GHz) per wave-level fetch per CU\n", name, ms, ms * 1e-3 * 2.4e9 / (gathers / 256));
}
int main()
{
    float *buf, *out;
    const size_t n = (size_t)64 * 16 * ROW + 64 * ROW;
    hipMalloc(&buf, n * 4 * 2); hipMemset(buf, 0, n * 4 * 2); hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("dword, lane stride 4 B", buf, out);
    run<1>("dwordx2 overlapping, lane stride 4 B", buf, out);
    run<2>("dwordx2, lane stride 8 B", buf, out);
    run<3>("dwordx4, lane stride 16 B", buf, out);
    run<4>("2 x dword (z, z+1)", buf, out);
    run<5>("dwordx4 overlapping, lane stride 4 B", buf, out);
    run<6>("dwordx4, 17 active lanes (260 B segment)", buf, out);
    run<7>("dwordx2, 33 active lanes (264 B segment)", buf, out);
    return 0;
}
