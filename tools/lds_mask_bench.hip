// micro-benchmark (development aid, round 3): what does an LDS atomic / read cost when only SOME lanes of the wave are active?
// Question behind it (HISTORY.md section 4, tilted adjoint): schemes that merge the upper-plane contributions of lane l with the
// lower-plane ones of lane l+1 need a fallback pass for the few lanes whose neighbour sits in another cell.  That pass only pays if
// a ds_add_u32 with 2-4 active lanes is cheaper than one with 64.
// Every CU busy (1024 work-groups of 512 threads, 2 resident per CU), conflict-free addresses (lanes on consecutive dwords).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048

// MODE 0: ds_add_u32   1: ds_read_b32 (+ use)   2: ds_read2_b32   3: ds_add_u64   4: ds_write_b32
template <int MODE> __global__ __launch_bounds__(512) void k(unsigned *out, unsigned long long mask)
{
    __shared__ unsigned acc[17 * 17 * 64];
    for (int e = threadIdx.x; e < 17 * 17 * 64; e += 512) acc[e] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool on = (mask >> lane) & 1ull;
    unsigned base = (wv * 20 + 3) * 64 + lane;
    unsigned sum = 0;
    if (on) {                                      // exec mask = `mask` for the whole loop
        for (int it = 0; it < N_IT; ++it) {
            unsigned a = base + (it & 7) * 64 * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {          // 8 LDS instructions per trip, like the tilted adjoint's sample
                unsigned *p = &acc[a + r * 64];
                if (MODE == 0) { atomicAdd(p, (unsigned)it); atomicAdd(p + 17 * 64, (unsigned)it); }
                else if (MODE == 1) { sum += p[0]; sum += p[17 * 64]; }
                else if (MODE == 2) { sum += p[0] + p[1]; sum += p[17 * 64] + p[17 * 64 + 1]; }
                else if (MODE == 3) { unsigned long long *q = (unsigned long long *)acc + ((a - lane + r * 64) >> 1) + lane; atomicAdd(q, (unsigned long long)it); atomicAdd(q + 17 * 32, (unsigned long long)it); }
                else if (MODE == 4) { p[0] = (unsigned)it; p[17 * 64] = (unsigned)it; }
            }
            if (MODE == 1 || MODE == 2) asm volatile("" : "+v"(sum));
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc[base] + sum;
}

template <int MODE> void run(const char *name, unsigned long long mask, const char *mname)
{
    unsigned *d;
    (void)hipMalloc(&d, 1024 * 64 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<1024, 512>>>(d, mask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k<MODE><<<1024, 512>>>(d, mask);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double waveops = 1024.0 * 8 * N_IT * 8;          // wave-level LDS instructions
    printf("%-14s lanes %-22s %8.3f ms  %6.1f G wave-instr/s  %5.2f clk(2.4 GHz) per instr per CU\n", name, mname, ms, waveops / ms / 1e6,
           ms * 1e-3 * 2.4e9 / (waveops / 256));
    (void)hipFree(d);
}

int main()
{
    struct { unsigned long long m; const char *n; } masks[] = {
        {~0ull, "all 64"}, {0xffffffffull, "0-31 (one half)"}, {0x5555555555555555ull, "every 2nd (32)"}, {0x00000000ffff0000ull, "16-31 (16)"},
        {0x0101010101010101ull, "every 8th (8)"}, {0x0000000100000001ull, "0 and 32 (2)"}, {0x1ull, "lane 0 (1)"}, {0x0000000300000000ull, "32,33 (2)"}};
    for (auto &mk : masks) run<0>("ds_add_u32", mk.m, mk.n);
    for (auto &mk : masks) run<3>("ds_add_u64", mk.m, mk.n);
    for (auto &mk : masks) run<1>("ds_read_b32", mk.m, mk.n);
    for (auto &mk : masks) run<2>("ds_read2_b32", mk.m, mk.n);
    for (auto &mk : masks) run<4>("ds_write_b32", mk.m, mk.n);
    return 0;
}
