// tools/issue_bench.hip -- what one CU of an MI355X can issue per clock, measured: the denominators of bench.py's roofline
// for the LDS-tile kernels (VALU wave-instructions, LDS wave-instructions) and the numbers behind HISTORY.md's cycle budgets.
//
//   hipcc -O3 --offload-arch=gfx950 tools/issue_bench.hip -o /tmp/issue_bench && /tmp/issue_bench
//
// Every test runs a loop of 16 independent copies of one instruction, on every CU, with 1 / 2 / 4 waves per SIMD
// (256-thread work-groups, 1 / 2 / 4 of them per CU), and prints
//   cyc/instr/SIMD  = shader cycles (s_memtime) the loop took / (instructions per wave * waves per SIMD)
//   Ginstr/s chip   = wave-instructions per second over the whole chip from the wall time (HIP events)
// so "1 wave-instruction per N cycles per SIMD" can be read off directly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

enum Op { FMA, PK_FMA, ADD_U32, ADD64, READLANE, CVT, MUL_LO, MAD_U24, DPP_MOV, FLOOR, FRACT, CNDMASK, CNDMASK_IND, CNDMASK_SGPR, MUL_HI_I32, MUL_HI_U32, MIN_U32,
          CVT_RPI, PK_MUL, LSHL_ADD, BFE_I32, CMP_LT, MIX_FMA_CND, CVT_FLR, MED3, CND_DPP, ADD64_ONE, PERMUTE,
          DS_READ_B32, DS_READ2_B32, DS_READ2ST64, DS_READ_B64, DS_READ_B128, DS_ADD_U32, DS_ADD_U64, DS_ADD_F32, DS_WRITE_B32, DS_ADD_F32_ZERO, DS_ADD_F32_HALF, DS_ADD_RTN_F32, DS_MAX_F32, DS_ADD_F64, N_OPS };
static const char *op_name[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_add_co+v_addc (64-bit add)", "v_readlane_b32", "v_cvt_f32_u32",
                                     "v_mul_lo_u32", "v_mad_u32_u24", "v_mov_b32 dpp wave_shl:1", "v_floor_f32", "v_fract_f32", "v_cndmask_b32",
                                     "v_cndmask_b32 (indep. dst, vcc set)", "v_cndmask_b32 (sgpr-pair mask)", "v_mul_hi_i32", "v_mul_hi_u32", "v_min_u32",
                                     "v_cvt_rpi_i32_f32", "v_pk_mul_f32", "v_lshl_add_u32", "v_bfe_i32", "v_cmp_lt_u32 (to sgpr pair)",
                                     "3 v_fma_f32 + 1 v_cndmask vcc", "v_cvt_flr_i32_f32", "v_med3_i32", "v_cndmask_b32_dpp wave_shl:1 (vcc)", "v_lshl_add_u64 (v + s pair)",
                                     "ds_permute_b32", "ds_read_b32", "ds_read2_b32", "ds_read2st64_b32", "ds_read_b64", "ds_read_b128",
                                     "ds_add_u32", "ds_add_u64", "ds_add_f32", "ds_write_b32",
                                     "ds_add_f32 (+0.0 onto zeros)", "ds_add_f32 (+0.5, sums stay < 2^24)", "ds_add_rtn_f32", "ds_max_f32", "ds_add_f64"};

template <int OP>
__global__ __launch_bounds__(256) void k_issue(int iters, unsigned long long *cycles, float *sink)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
    __syncthreads();
    float a0 = lane, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f};
    f4 q0 = {0, 0, 0, 0}, q1 = q0;
    unsigned u0 = lane, u1 = lane + 1, u2 = lane + 2, u3 = lane + 3;
    unsigned long long w0 = lane, w1 = lane * 3;
    double d0 = 1.0;
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    unsigned long long msk = __ballot(lane & 1), m0 = 0, m1 = 0;
    asm volatile("v_cmp_lt_u32 vcc, 7, %0" : : "v"(lane) : "vcc");
    // LDS address: lane along consecutive dwords (conflict-free), wave-private 4 KB region
    const unsigned base = ((threadIdx.x >> 6) * 4096u) + lane * 4u;
    const unsigned base8 = ((threadIdx.x >> 6) * 4096u) + lane * 8u;
    const unsigned base16 = ((threadIdx.x >> 6) * 4096u) + lane * 16u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) {
            R4(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (OP == PK_FMA) {
            R4(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (OP == ADD_U32) {
            R4(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == ADD64) {      // 16 64-bit adds = 32 VALU instructions (counted as 16 "instructions" below)
            R4(R4(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3) : "vcc");))
        } else if (OP == READLANE) {
            R4(asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9"
                            : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(u0));)
        } else if (OP == CVT) {
            R4(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %4\n v_cvt_f32_u32 %2, %4\n v_cvt_f32_u32 %3, %4"
                            : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(u0));)
        } else if (OP == MUL_LO) {
            R4(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == MAD_U24) {
            R4(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == DPP_MOV) {
            R4(asm volatile("v_mov_b32_dpp %0, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                            "v_mov_b32_dpp %2, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                            : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(lane));)
        } else if (OP == FLOOR) {
            R4(asm volatile("v_floor_f32 %0, %4\n v_floor_f32 %1, %4\n v_floor_f32 %2, %4\n v_floor_f32 %3, %4" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a4));)
        } else if (OP == FRACT) {
            R4(asm volatile("v_fract_f32 %0, %4\n v_fract_f32 %1, %4\n v_fract_f32 %2, %4\n v_fract_f32 %3, %4" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a4));)
        } else if (OP == CNDMASK) {
            R4(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : );)
        } else if (OP == CNDMASK_IND) {
            R4(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc"
                            : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a4), "v"(a5) : );)
        } else if (OP == CNDMASK_SGPR) {
            R4(asm volatile("v_cndmask_b32 %0, %4, %5, %6\n v_cndmask_b32 %1, %5, %4, %6\n v_cndmask_b32 %2, %4, %5, %6\n v_cndmask_b32 %3, %5, %4, %6"
                            : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a4), "v"(a5), "s"(msk) : );)
        } else if (OP == MUL_HI_I32) {
            R4(asm volatile("v_mul_hi_i32 %0, %0, %4\n v_mul_hi_i32 %1, %1, %4\n v_mul_hi_i32 %2, %2, %4\n v_mul_hi_i32 %3, %3, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == MUL_HI_U32) {
            R4(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == MIN_U32) {
            R4(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == CVT_RPI) {
            R4(asm volatile("v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %4\n v_cvt_rpi_i32_f32 %2, %4\n v_cvt_rpi_i32_f32 %3, %4"
                            : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a4));)
        } else if (OP == PK_MUL) {
            R4(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (OP == LSHL_ADD) {
            R4(asm volatile("v_lshl_add_u32 %0, %0, 4, %4\n v_lshl_add_u32 %1, %1, 4, %4\n v_lshl_add_u32 %2, %2, 4, %4\n v_lshl_add_u32 %3, %3, 4, %4"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));)
        } else if (OP == BFE_I32) {
            R4(asm volatile("v_bfe_i32 %0, %0, 31, 1\n v_bfe_i32 %1, %1, 31, 1\n v_bfe_i32 %2, %2, 31, 1\n v_bfe_i32 %3, %3, 31, 1"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : );)
        } else if (OP == CMP_LT) {
            R4(asm volatile("v_cmp_lt_u32 %0, %2, %3\n v_cmp_lt_u32 %1, %3, %2\n v_cmp_lt_u32 %0, %3, %2\n v_cmp_lt_u32 %1, %2, %3"
                            : "=s"(m0), "=s"(m1) : "v"(u0), "v"(u1));)
        } else if (OP == MIX_FMA_CND) {
            R4(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (OP == CVT_FLR) {
            R4(asm volatile("v_cvt_flr_i32_f32 %0, %4\n v_cvt_flr_i32_f32 %1, %4\n v_cvt_flr_i32_f32 %2, %4\n v_cvt_flr_i32_f32 %3, %4"
                            : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a4));)
        } else if (OP == MED3) {
            R4(asm volatile("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5"
                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane), "v"(base));)
        } else if (OP == CND_DPP) {
            R4(asm volatile("v_cndmask_b32_dpp %0, %4, %5, vcc wave_shl:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %1, %5, %4, vcc wave_shl:1 row_mask:0xf bank_mask:0xf\n"
                            "v_cndmask_b32_dpp %2, %4, %5, vcc wave_shl:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %5, %4, vcc wave_shl:1 row_mask:0xf bank_mask:0xf"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : );)
        } else if (OP == ADD64_ONE) {
            R4(asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2"
                            : "+v"(w0), "+v"(w1) : "s"(msk));)
        } else if (OP == PERMUTE) {
            R4(asm volatile("ds_permute_b32 %0, %4, %5\n ds_permute_b32 %1, %4, %5\n ds_permute_b32 %2, %4, %5\n ds_permute_b32 %3, %4, %5\n s_waitcnt lgkmcnt(0)"
                            : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(base), "v"(lane));)
        } else if (OP == DS_READ_B32) {
            R4(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(base));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_READ2_B32) {
            R4(asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n ds_read2_b32 %1, %4 offset0:64 offset1:65\n ds_read2_b32 %2, %4 offset0:128 offset1:129\n ds_read2_b32 %3, %4 offset0:192 offset1:193\n"
                            : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(base));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_READ2ST64) {
            R4(asm volatile("ds_read2st64_b32 %0, %4 offset0:0 offset1:1\n ds_read2st64_b32 %1, %4 offset0:2 offset1:3\n ds_read2st64_b32 %2, %4 offset0:4 offset1:5\n ds_read2st64_b32 %3, %4 offset0:6 offset1:7\n"
                            : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(base));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_READ_B64) {
            R4(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(base8));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_READ_B128) {
            R4(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072\n" : "=v"(q0), "=v"(q1) : "v"(base16));)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_ADD_U32) {
            R4(asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:256\n ds_add_u32 %0, %1 offset:512\n ds_add_u32 %0, %1 offset:768\n" : : "v"(base), "v"(u0) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_ADD_U64) {
            R4(asm volatile("ds_add_u64 %0, %1\n ds_add_u64 %0, %1 offset:512\n ds_add_u64 %0, %1 offset:1024\n ds_add_u64 %0, %1 offset:1536\n" : : "v"(base8), "v"(w0) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_ADD_F32) {
            R4(asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1 offset:256\n ds_add_f32 %0, %1 offset:512\n ds_add_f32 %0, %1 offset:768\n" : : "v"(base), "v"(a1) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_ADD_F32_ZERO || OP == DS_ADD_F32_HALF) {
            // round 5: is ds_add_f32's 768 cycles a property of the operands?  LDS is zeroed above and every operand is finite and
            // normal in all three variants: +1.0 (DS_ADD_F32; sums reach 32 000), +0.0 (sums stay 0), +0.5 (sums reach 16 000)
            const float inc = OP == DS_ADD_F32_ZERO ? 0.0f : 0.5f;
            R4(asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1 offset:256\n ds_add_f32 %0, %1 offset:512\n ds_add_f32 %0, %1 offset:768\n" : : "v"(base), "v"(inc) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_ADD_RTN_F32) {
            R4(asm volatile("ds_add_rtn_f32 %0, %4, %5\n ds_add_rtn_f32 %1, %4, %5 offset:256\n ds_add_rtn_f32 %2, %4, %5 offset:512\n ds_add_rtn_f32 %3, %4, %5 offset:768\n"
                            : "=v"(a0), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(base), "v"(a1) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_MAX_F32) {
            R4(asm volatile("ds_max_f32 %0, %1\n ds_max_f32 %0, %1 offset:256\n ds_max_f32 %0, %1 offset:512\n ds_max_f32 %0, %1 offset:768\n" : : "v"(base), "v"(a1) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_ADD_F64) {
            R4(asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:512\n ds_add_f64 %0, %1 offset:1024\n ds_add_f64 %0, %1 offset:1536\n" : : "v"(base8), "v"(d0) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == DS_WRITE_B32) {
            R4(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n" : : "v"(base), "v"(a1) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float r = a0 + a1 + a2 + a3 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + q0.x + q1.w + (float)(u0 + u1 + u2 + u3) + (float)(w0 + w1) + (float)(s0 + s1 + s2 + s3) + (float)(m0 ^ m1) + lds[lane];
    if (r == 123.456f) sink[0] = r;
}

template <int OP>
static void run(int n_cu, double clk_ghz)
{
    const int iters = 2000;
    unsigned long long *d_cyc;
    float *d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * n_cu * 8 * 4));
    CHECK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-30s", op_name[OP]);
    for (int bpc = 1; bpc <= 4; bpc *= 2) {           // work-groups per CU = waves per SIMD
        const int grid = n_cu * bpc;
        hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(256), 0, 0, 10, d_cyc, d_sink);          // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(256), 0, 0, iters, d_cyc, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long *h = (unsigned long long *)malloc(sizeof(unsigned long long) * grid * 4);
        CHECK(hipMemcpy(h, d_cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int i = 0; i < grid * 4; ++i) sum += (double)h[i];
        free(h);
        const double per_wave_cycles = sum / (grid * 4);              // s_memtime ticks = shader cycles (MI355X_MICROARCH)
        const double instr_per_wave = 16.0 * iters;
        const double cyc_per_instr_simd = per_wave_cycles / (instr_per_wave * bpc);
        const double ginstr = instr_per_wave * grid * 4 / (ms * 1e-3) / 1e9;
        printf(" | %dw/SIMD %6.2f cyc/instr/SIMD %8.1f Ginstr/s", bpc, cyc_per_instr_simd, ginstr);
    }
    printf("\n");
    CHECK(hipFree(d_cyc));
    CHECK(hipFree(d_sink));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", p.name, n_cu, p.clockRate / 1e3);
    printf("(16 independent instructions per loop trip; LDS tests wait lgkmcnt(0) once per 16; 64-bit add = 2 VALU per counted instruction)\n");
    const double clk = p.clockRate / 1e6;
    run<FMA>(n_cu, clk); run<PK_FMA>(n_cu, clk); run<ADD_U32>(n_cu, clk); run<ADD64>(n_cu, clk); run<READLANE>(n_cu, clk); run<CVT>(n_cu, clk);
    run<MUL_LO>(n_cu, clk); run<MAD_U24>(n_cu, clk); run<DPP_MOV>(n_cu, clk); run<FLOOR>(n_cu, clk); run<FRACT>(n_cu, clk); run<CNDMASK>(n_cu, clk);
    run<CNDMASK_IND>(n_cu, clk); run<CNDMASK_SGPR>(n_cu, clk); run<MUL_HI_I32>(n_cu, clk); run<MUL_HI_U32>(n_cu, clk); run<MIN_U32>(n_cu, clk);
    run<CVT_RPI>(n_cu, clk); run<PK_MUL>(n_cu, clk); run<LSHL_ADD>(n_cu, clk); run<BFE_I32>(n_cu, clk); run<CMP_LT>(n_cu, clk);
    run<MIX_FMA_CND>(n_cu, clk); run<CVT_FLR>(n_cu, clk); run<MED3>(n_cu, clk); run<CND_DPP>(n_cu, clk); run<ADD64_ONE>(n_cu, clk);
    run<PERMUTE>(n_cu, clk);
    run<DS_READ_B32>(n_cu, clk); run<DS_READ2_B32>(n_cu, clk); run<DS_READ2ST64>(n_cu, clk); run<DS_READ_B64>(n_cu, clk); run<DS_READ_B128>(n_cu, clk);
    run<DS_ADD_U32>(n_cu, clk); run<DS_ADD_U64>(n_cu, clk); run<DS_ADD_F32>(n_cu, clk); run<DS_WRITE_B32>(n_cu, clk);
    run<DS_ADD_F32_ZERO>(n_cu, clk); run<DS_ADD_F32_HALF>(n_cu, clk); run<DS_ADD_RTN_F32>(n_cu, clk); run<DS_MAX_F32>(n_cu, clk); run<DS_ADD_F64>(n_cu, clk);
    return 0;
}
