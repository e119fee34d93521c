// development aid: semantics of the DPP wave shifts on gfx950 (which lane feeds which, what lane 63 / lane 0 receive)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int v = 100 + (int)threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);        // wave_shl:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
int main()
{
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shl:1  lane0 <- %d  lane1 <- %d  lane15 <- %d lane16 <- %d lane31 <- %d lane62 <- %d  lane63 <- %d\n", h[0], h[1], h[15], h[16], h[31], h[62], h[63]);
    printf("wave_shr:1  lane0 <- %d  lane1 <- %d  lane16 <- %d lane32 <- %d lane63 <- %d\n", h[64], h[65], h[80], h[96], h[127]);
    int ok = 1;
    for (int i = 0; i < 63; ++i) ok &= (h[i] == 101 + i);
    printf("wave_shl:1 is lane i <- lane i+1 for all i < 63: %s; lane 63 keeps old: %s\n", ok ? "yes" : "NO", h[63] == -1 ? "yes" : "NO");
    return 0;
}
