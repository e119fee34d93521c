// micro-benchmark (development aid, round 3): global float atomics by SCOPE and by the XCD placement of the adders.
// MI355X_MICROARCH.md: no-return global_atomic_add_f32 at agent scope executes at the memory side, ~1.3 TB/s chip-wide whatever
// the locality -- the co-bound of the flat forward projector (275-368 GB of 256-B row atomics per launch).  Question: do the
// narrower scopes (wavefront / workgroup: no sc bits) execute in the XCD's L2 instead, and at what rate?  If so, a launch whose
// adders to one sinogram row all sit on ONE XCD could use them.  Also measured: plain read-modify-write (load, add, store) of the
// same rows by an exclusive owner, and plain stores, as the ceilings.
//   table: R rows of 64 floats (256 B); every wave adds to pseudo-random rows (one 256-B wave-instruction each), 16 in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_IT 512

template <int MODE> __global__ __launch_bounds__(256) void k(float *tab, unsigned n_rows_mask, int xcd_local, float *sink)
{
    const int lane = threadIdx.x & 63;
    unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned h = wid * 2654435761u + 12345u;
    // xcd_local: rows are partitioned by the adder's XCD (blockIdx % 8 under round-robin dispatch): every row has adders on one XCD only
    const unsigned xcd = blockIdx.x & 7u;
    float acc = 0.f;
    for (int it = 0; it < N_IT; ++it) {
        h = h * 1664525u + 1013904223u;
        unsigned row = (h >> 8) & n_rows_mask;
        if (xcd_local) row = (row & ~7u) | xcd;
        if (MODE == 8 || MODE == 9) row &= ~1u;             // 512 B = two rows of the table
        float *p = tab + (size_t)row * 64 + lane;
        const float v = 1.0f;
        if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == 4) *p = v + (float)it;                              // plain store
        else if (MODE == 5) { float o = *p; *p = o + v; }                    // plain read-modify-write (NOT safe with several adders per row)
        else if (MODE == 6) acc += *p;                                        // plain load
        else if (MODE == 7) acc += __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // RETURNING atomic (old value used)
        else if (MODE == 8) __hip_atomic_fetch_add((unsigned long long *)(tab + (size_t)row * 64) + lane, 0x0000000100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // u64: 8 B per lane, 512 B per wave-instruction (two rows)
        else if (MODE == 9) __hip_atomic_fetch_add((double *)(tab + (size_t)row * 64) + lane, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE >= 10) {
            // ALIGNMENT of a row instruction (the flat forward adds 63 consecutive floats at an arbitrary float offset): 10 = 63 lanes at a
            // pseudo-random float offset, 11 = 63 lanes aligned to 256 B, 12 = 64 lanes offset by 32 floats (128-B aligned), 13 = 64 lanes
            // offset by 16 floats (64-B aligned), 14 = 64 lanes offset by 1 float, 15 = 32 lanes aligned to 128 B
            const unsigned off = MODE == 10 ? ((h >> 2) & 63u) : MODE == 12 ? 32u : MODE == 13 ? 16u : MODE == 14 ? 1u : 0u;
            float *q = tab + (size_t)(row & (n_rows_mask >> 1)) * 64 + off + lane;          // the offset row stays inside the table
            const int n_act = (MODE == 10 || MODE == 11) ? 63 : MODE == 15 ? 32 : 64;
            if (lane < n_act) __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (MODE == 6 || MODE == 7) sink[wid * 64 + lane] = acc;
}

template <int MODE> void run(const char *name, size_t table_bytes, int xcd_local, int check)
{
    float *tab, *sink;
    const unsigned n_rows = (unsigned)(table_bytes / 256);
    (void)hipMalloc(&tab, table_bytes);
    (void)hipMalloc(&sink, 4096 * 4 * 64 * 4);
    (void)hipMemset(tab, 0, table_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 4096;
    k<MODE><<<blocks, 256>>>(tab, n_rows - 1, xcd_local, sink);
    (void)hipDeviceSynchronize();
    (void)hipMemset(tab, 0, table_bytes);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(tab, n_rows - 1, xcd_local, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * N_IT * ((MODE == 8 || MODE == 9) ? 512 : (MODE == 10 || MODE == 11) ? 252 : MODE == 15 ? 128 : 256);
    double total = -1;
    if (MODE == 8 || MODE == 9) check = 0;
    if (check) {                 // no add may be lost: the table must sum to the number of adds
        float *h = (float *)malloc(table_bytes);
        (void)hipMemcpy(h, tab, table_bytes, hipMemcpyDeviceToHost);
        total = 0;
        for (size_t i = 0; i < table_bytes / 4; ++i) total += h[i];
        free(h);
    }
    printf("%-34s table %6.0f MB %s: %8.3f ms  %7.1f GB/s of added/stored bytes%s", name, table_bytes / 1048576.0, xcd_local ? "rows by XCD " : "rows shared ", ms,
           bytes / ms / 1e6, check ? "" : "\n");
    if (check) printf("   sum %.0f of %.0f %s\n", total, bytes / 4, total == bytes / 4 ? "(exact)" : "(ADDS LOST)");
    (void)hipFree(tab); (void)hipFree(sink);
}

int main()
{
    for (size_t mb : {64, 1024}) {
        const size_t b = mb << 20;
        for (int loc : {0, 1}) {
            run<0>("atomic add f32, agent scope", b, loc, 1);
            run<1>("atomic add f32, workgroup scope", b, loc, 1);
            run<2>("atomic add f32, wavefront scope", b, loc, 1);
            run<3>("atomic add f32, system scope", b, loc, 1);
        }
        run<7>("atomic add f32, agent, RETURNING", b, 0, 1);
        run<7>("atomic add f32, agent, RETURNING", b, 1, 1);
        run<8>("atomic add u64 (8 B per lane)", b, 0, 0);
        run<9>("atomic add f64 (8 B per lane)", b, 0, 0);
        run<11>("atomic f32, 63 lanes, 256-B aligned", b, 0, 1);
        run<10>("atomic f32, 63 lanes, any float offset", b, 0, 1);
        run<12>("atomic f32, 64 lanes, +128 B", b, 0, 1);
        run<13>("atomic f32, 64 lanes, +64 B", b, 0, 1);
        run<14>("atomic f32, 64 lanes, +4 B", b, 0, 1);
        run<15>("atomic f32, 32 lanes, 128-B aligned", b, 0, 1);
        run<4>("plain store", b, 0, 0);
        run<5>("plain load + add + store", b, 0, 0);
        run<6>("plain load", b, 0, 0);
    }
    return 0;
}
