// Issue cost of vector instructions with every SIMD holding 8 waves (cycles per
// wave-instruction and SIMD, from the elapsed time at the reported clock; compare
// the rows with each other rather than with a nominal figure).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY4(i0, i1, i2, i3) asm volatile(i0 "\n" i1 "\n" i2 "\n" i3 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1) : "v"(b) : "s20", "s21", "s22", "s23", "scc", "vcc");
#define DEFK(NAME, i0, i1, i2, i3)                                                                   \
    __global__ void __launch_bounds__(512) NAME(float* out, int iters) {                              \
        typedef float v2 __attribute__((ext_vector_type(2)));                                         \
        v2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, b = {0.999f, 1.001f}; \
        unsigned long long u0 = threadIdx.x, u1 = threadIdx.x * 3;                                    \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");     \
        for (int i = 0; i < iters; ++i) { REP16(BODY4(i0, i1, i2, i3)) }                              \
        v2 s = a0 + a1 + a2 + a3;                                                                     \
        if (s.x + s.y + (float)(u0 + u1) == 12345.678f) out[0] = s.x;                                 \
    }
// %0..%3: float pairs a0..a3 (sub-registers by modifier are not available, so plain forms use the pair's low half via v_*_e32 on %0 etc. being 64-bit is not allowed: use separate kernels below)
DEFK(k_pk_fma, "v_pk_fma_f32 %0, %0, %6, %6", "v_pk_fma_f32 %1, %1, %6, %6", "v_pk_fma_f32 %2, %2, %6, %6", "v_pk_fma_f32 %3, %3, %6, %6")
DEFK(k_pk_mul, "v_pk_mul_f32 %0, %0, %6", "v_pk_mul_f32 %1, %1, %6", "v_pk_mul_f32 %2, %2, %6", "v_pk_mul_f32 %3, %3, %6")
DEFK(k_pk_mov, "v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]", "v_pk_mov_b32 %1, %1, %1 op_sel:[1,0]", "v_pk_mov_b32 %2, %2, %2 op_sel:[1,0]", "v_pk_mov_b32 %3, %3, %3 op_sel:[1,0]")
DEFK(k_lshl_add_u64, "v_lshl_add_u64 %4, %4, 3, %5", "v_lshl_add_u64 %5, %5, 3, %4", "v_lshl_add_u64 %4, %4, 2, %5", "v_lshl_add_u64 %5, %5, 2, %4")

#define BODY1(i0, i1, i2, i3) asm volatile(i0 "\n" i1 "\n" i2 "\n" i3 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g), "v"(h) : "s22", "s23", "s24", "s25", "scc", "vcc");
#define DEFS(NAME, i0, i1, i2, i3)                                                                   \
    __global__ void __launch_bounds__(512) NAME(float* out, int iters) {                              \
        float f0 = 1.0f + threadIdx.x, f1 = f0 + 1.0f, f2 = f0 + 2.0f, f3 = f0 + 3.0f, g = 0.999f, h = 1.5f; \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");     \
        for (int i = 0; i < iters; ++i) { REP16(BODY1(i0, i1, i2, i3)) }                              \
        if (f0 + f1 + f2 + f3 == 12345.678f) out[0] = f0;                                             \
    }
DEFS(k_fma, "v_fma_f32 %0, %0, %4, %4", "v_fma_f32 %1, %1, %4, %4", "v_fma_f32 %2, %2, %4, %4", "v_fma_f32 %3, %3, %4, %4")
DEFS(k_mul, "v_mul_f32 %0, %0, %4", "v_mul_f32 %1, %1, %4", "v_mul_f32 %2, %2, %4", "v_mul_f32 %3, %3, %4")
DEFS(k_add_u32, "v_add_u32 %0, %0, %4", "v_add_u32 %1, %1, %4", "v_add_u32 %2, %2, %4", "v_add_u32 %3, %3, %4")
DEFS(k_min, "v_min_f32 %0, %0, %4", "v_min_f32 %1, %1, %4", "v_min_f32 %2, %2, %4", "v_min_f32 %3, %3, %4")
DEFS(k_max3, "v_max3_f32 %0, %0, %4, %5", "v_max3_f32 %1, %1, %4, %5", "v_max3_f32 %2, %2, %4, %5", "v_max3_f32 %3, %3, %4, %5")
DEFS(k_fma3, "v_fma_f32 %0, %0, %4, %5", "v_fma_f32 %1, %1, %4, %5", "v_fma_f32 %2, %2, %4, %5", "v_fma_f32 %3, %3, %4, %5")
DEFS(k_cnd_vcc, "v_cndmask_b32 %0, %0, %4, vcc", "v_cndmask_b32 %1, %1, %4, vcc", "v_cndmask_b32 %2, %2, %4, vcc", "v_cndmask_b32 %3, %3, %4, vcc")
DEFS(k_cnd_sgpr, "v_cndmask_b32 %0, %0, %4, s[20:21]", "v_cndmask_b32 %1, %1, %4, s[20:21]", "v_cndmask_b32 %2, %2, %4, s[20:21]", "v_cndmask_b32 %3, %3, %4, s[20:21]")
DEFS(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %4", "v_cmp_lt_f32 vcc, %1, %4", "v_cmp_lt_f32 vcc, %2, %4", "v_cmp_lt_f32 vcc, %3, %4")
DEFS(k_cmp_sgpr, "v_cmp_lt_f32 s[22:23], %0, %4", "v_cmp_lt_f32 s[24:25], %1, %4", "v_cmp_lt_f32 s[22:23], %2, %4", "v_cmp_lt_f32 s[24:25], %3, %4")
DEFS(k_cmp_cnd_vcc, "v_cmp_lt_f32 vcc, %0, %4", "v_cndmask_b32 %1, %1, %4, vcc", "v_cmp_lt_f32 vcc, %2, %4", "v_cndmask_b32 %3, %3, %4, vcc")
DEFS(k_cmp_cnd_sgpr, "v_cmp_lt_f32 s[22:23], %0, %4", "v_cndmask_b32 %1, %1, %4, s[22:23]", "v_cmp_lt_f32 s[24:25], %2, %4", "v_cndmask_b32 %3, %3, %4, s[24:25]")
DEFS(k_cnd_vcc_e64, "v_cndmask_b32_e64 %0, %0, %4, vcc", "v_cndmask_b32_e64 %1, %1, %4, vcc", "v_cndmask_b32_e64 %2, %2, %4, vcc", "v_cndmask_b32_e64 %3, %3, %4, vcc")
DEFS(k_cnd_vcc_other, "v_cndmask_b32 %0, %5, %4, vcc", "v_cndmask_b32 %1, %5, %4, vcc", "v_cndmask_b32 %2, %5, %4, vcc", "v_cndmask_b32 %3, %5, %4, vcc")
DEFS(k_cmp_3cnd_vcc, "v_cmp_lt_f32 vcc, %0, %4", "v_cndmask_b32 %1, %1, %4, vcc", "v_cndmask_b32 %2, %2, %4, vcc", "v_cndmask_b32 %3, %3, %4, vcc")
DEFS(k_cmp_3cnd_sgpr, "v_cmp_lt_f32 s[22:23], %0, %4", "v_cndmask_b32 %1, %1, %4, s[22:23]", "v_cndmask_b32 %2, %2, %4, s[22:23]", "v_cndmask_b32 %3, %3, %4, s[22:23]")
DEFS(k_cmp_3cnd_vcc64, "v_cmp_lt_f32 vcc, %0, %4", "v_cndmask_b32_e64 %1, %1, %4, vcc", "v_cndmask_b32_e64 %2, %2, %4, vcc", "v_cndmask_b32_e64 %3, %3, %4, vcc")
DEFS(k_cnd_mul_vcc, "v_cndmask_b32 %0, %0, %4, vcc", "v_mul_f32 %1, %1, %4", "v_cndmask_b32 %2, %2, %4, vcc", "v_mul_f32 %3, %3, %4")
DEFS(k_min_i32, "v_min_i32 %0, %0, %4", "v_min_i32 %1, %1, %4", "v_min_i32 %2, %2, %4", "v_min_i32 %3, %3, %4")
DEFS(k_min_u32, "v_min_u32 %0, %0, %4", "v_min_u32 %1, %1, %4", "v_min_u32 %2, %2, %4", "v_min_u32 %3, %3, %4")
DEFS(k_min3_i32, "v_min3_i32 %0, %0, %4, %5", "v_min3_i32 %1, %1, %4, %5", "v_min3_i32 %2, %2, %4, %5", "v_min3_i32 %3, %3, %4, %5")
DEFS(k_med3, "v_med3_f32 %0, %0, %4, %5", "v_med3_f32 %1, %1, %4, %5", "v_med3_f32 %2, %2, %4, %5", "v_med3_f32 %3, %3, %4, %5")
DEFS(k_sub, "v_sub_f32 %0, %0, %4", "v_sub_f32 %1, %1, %4", "v_sub_f32 %2, %2, %4", "v_sub_f32 %3, %3, %4")
DEFS(k_and_or, "v_and_or_b32 %0, %0, %4, %5", "v_and_or_b32 %1, %1, %4, %5", "v_and_or_b32 %2, %2, %4, %5", "v_and_or_b32 %3, %3, %4, %5")
DEFS(k_xor, "v_xor_b32 %0, %0, %4", "v_xor_b32 %1, %1, %4", "v_xor_b32 %2, %2, %4", "v_xor_b32 %3, %3, %4")
DEFS(k_cmp_i32, "v_cmp_lt_i32 s[22:23], %0, %4", "v_cmp_lt_i32 s[24:25], %1, %4", "v_cmp_lt_i32 s[22:23], %2, %4", "v_cmp_lt_i32 s[24:25], %3, %4")
DEFS(k_mov, "v_mov_b32 %0, %4", "v_mov_b32 %1, %4", "v_mov_b32 %2, %4", "v_mov_b32 %3, %4")
DEFS(k_rcp, "v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3")
DEFS(k_dpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
DEFS(k_salu, "s_add_u32 s22, s22, 1", "s_add_u32 s23, s23, 1", "s_add_u32 s24, s24, 1", "s_add_u32 s25, s25, 1")
DEFS(k_salu_valu, "s_add_u32 s22, s22, 1", "v_fma_f32 %1, %1, %4, %4", "s_add_u32 s24, s24, 1", "v_fma_f32 %3, %3, %4, %4")
DEFS(k_nop, "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0")

typedef void (*Kern)(float*, int);
void run(const char* name, Kern kern, float* d_out, double ghz) {
    const int iters = 2000, blocks = 256 * 4;  // 4 workgroups of 8 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d_out, 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d_out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 8.0 * iters * 64.0;  // 8 waves x iters x 64 instructions
    fflush(stdout);
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction and SIMD at %.2f GHz\n", name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
}
int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* d_out; hipMalloc(&d_out, 64);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
#define RUN(k) run(#k, k, d_out, ghz)
    RUN(k_fma); RUN(k_fma3); RUN(k_mul); RUN(k_add_u32); RUN(k_min); RUN(k_max3); RUN(k_mov); RUN(k_pk_fma); RUN(k_pk_mul); RUN(k_pk_mov);
    RUN(k_cnd_vcc); RUN(k_cnd_sgpr); RUN(k_cnd_vcc_e64); RUN(k_cnd_vcc_other); RUN(k_cmp_cnd_vcc); RUN(k_cmp_cnd_sgpr); RUN(k_cmp_3cnd_vcc); RUN(k_cmp_3cnd_sgpr); RUN(k_cmp_3cnd_vcc64); RUN(k_cnd_mul_vcc); RUN(k_min_i32); RUN(k_min_u32); RUN(k_min3_i32); RUN(k_med3); RUN(k_sub); RUN(k_and_or); RUN(k_xor); RUN(k_cmp_i32); RUN(k_cmp_vcc); RUN(k_cmp_sgpr); RUN(k_rcp); RUN(k_dpp); RUN(k_lshl_add_u64);
    RUN(k_salu); RUN(k_salu_valu); RUN(k_nop);
    return 0;
}
