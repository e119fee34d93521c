// micro-benchmark (development aid, round 3): how fast can the SCALAR data cache stream data that is read once?
// Question behind it (HISTORY.md section 7): the flat forward spends 5 of its 10 VALU instructions per sample entry on v_readlane
// broadcasts of a per-row table; feeding that table through s_load_dwordx8 instead would free them -- but every entry is read
// exactly once, i.e. every scalar load misses the scalar cache: ~3e10 entries/s x 32 B = ~1 TB/s chip-wide would be needed.
// Each wave streams its own region with s_load_dwordx8 (DEPTH loads in flight), 16 waves per CU, every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));

template <int DEPTH> __global__ __launch_bounds__(1024) void k(const int *__restrict__ base, size_t bytes_per_wave, int n_iter, int *out)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 16 + (threadIdx.x >> 6));
    uint64_t p = (uint64_t)base + (uint64_t)wave * bytes_per_wave;
    {   // wave-uniform: SGPR pair.  (__builtin_amdgcn_readfirstlane returns int: without the casts to unsigned a low word with bit 31 set
        //  sign-extends into the high word -- the first version of this file faulted on exactly that address)
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p & 0xffffffffull));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
        p = ((uint64_t)hi << 32) | (uint64_t)lo;
    }
    int acc = 0;
    for (int it = 0; it < n_iter; ++it) {
        v8i r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(r[d]) : "s"(p), "i"(d * 32));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            int t;
            asm volatile("s_add_i32 %0, %1, %2" : "=s"(t) : "s"(r[d].x), "s"(r[d].w));
            acc += t;
        }
        p += DEPTH * 32;
    }
    if ((threadIdx.x & 63) == 0) out[wave] = acc;
}

template <int DEPTH> void run(size_t bytes_per_wave)
{
    const int blocks = 256 * 2, waves = blocks * 16;
    int *buf, *out;
    (void)hipMalloc(&buf, waves * bytes_per_wave);
    (void)hipMalloc(&out, waves * 4);
    (void)hipMemset(buf, 1, waves * bytes_per_wave);
    const int n_iter = (int)(bytes_per_wave / (DEPTH * 32));
    if ((size_t)n_iter * DEPTH * 32 > bytes_per_wave) { printf("bad sizes\n"); return; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<DEPTH><<<blocks, 1024>>>(buf, bytes_per_wave, n_iter, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<DEPTH><<<blocks, 1024>>>(buf, bytes_per_wave, n_iter, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * n_iter * DEPTH * 32;
    printf("s_load_dwordx8, %d in flight per wave, %5.0f KB per wave (%6.0f MB in all): %8.3f ms  %7.1f GB/s chip-wide  %6.2f G loads/s\n", DEPTH,
           bytes_per_wave / 1024.0, waves * bytes_per_wave / 1048576.0, ms, bytes / ms / 1e6, bytes / 32 / ms / 1e6);
    (void)hipFree(buf); (void)hipFree(out);
}

int main()
{
    for (size_t kb : {16, 64}) {
        run<1>(kb << 10);
        run<2>(kb << 10);
        run<4>(kb << 10);
        run<8>(kb << 10);
    }
    return 0;
}
