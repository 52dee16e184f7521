// Single-wave VALU timing probe for gfx950: cycles per instruction of dependent / independent chains, DPP forms, s_nop, transcendental
// ops.  One wave64 on one SIMD (the regime of the step kernels at 4096 envs/GPU).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int CASE>
__global__ void probe(float* out, unsigned long long* ticks, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0f, d = 2.0f, e = 3.0f, f = 4.0f, g = 5.0f, h = 6.0f;
    unsigned long long t0 = 0, t1 = 0;
    for (int warm = 0; warm < 2; ++warm) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 16; ++it) {
            if (CASE == 0) asm volatile(REP64("v_fmac_f32 %0, %1, %1\n") : "+v"(a) : "v"(b));                                   // dependent chain
            if (CASE == 1) asm volatile(REP16("v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));   // 4 independent chains
            if (CASE == 2) asm volatile(REP64("v_fmac_f32_dpp %0, %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a) : "v"(b));       // dependent acc, dpp src constant
            if (CASE == 3) asm volatile(REP64("s_nop 1\n v_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a) : "v"(b));   // dpp src = previous result
            if (CASE == 4) asm volatile(REP64("v_mov_b32_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %0, %1, %2\n") : "+v"(a), "+v"(c) : "v"(b));   // mov_dpp + fmac
            if (CASE == 5) asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a));                                                    // dependent transcendental
            if (CASE == 6) asm volatile(REP16("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %4\n v_rcp_f32 %2, %4\n v_rcp_f32 %3, %4\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
            if (CASE == 7) asm volatile(REP64("s_nop 1\n"));
            if (CASE == 8) asm volatile(REP64("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n") : "+v"(a));  // scan step chain
            if (CASE == 9) asm volatile(REP64("v_pk_fma_f32 %0, %1, %1, %0\n") : "+v"(*(double*)&c) : "v"(*(double*)&e));           // dependent packed fma (register pair)
            if (CASE == 10) asm volatile(REP16("v_fmac_f32 %0, %2, %2\n v_fmac_f32 %1, %2, %2\n v_fmac_f32 %0, %2, %2\n v_fmac_f32 %1, %2, %2\n") : "+v"(a), "+v"(c) : "v"(b));   // 2 independent chains
            if (CASE == 11) asm volatile(REP64("v_mul_hi_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
            if (CASE == 12) asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b));
            if (CASE == 13) asm volatile(REP64("v_mad_u64_u32 %0, vcc, %1, %1, %0\n") : "+v"(*(double*)&c) : "v"(b) : "vcc");
            if (CASE == 14) asm volatile(REP64("v_sqrt_f32 %0, %0\n") : "+v"(a));
            if (CASE == 15) asm volatile(REP64("v_exp_f32 %0, %0\n") : "+v"(a));
        }
        t1 = __builtin_readcyclecounter();
    }
    out[threadIdx.x + CASE * 64] = a + c + d + e + f + g + h;
    if (threadIdx.x == 0) ticks[CASE] = t1 - t0;
}
int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 64 * 32 * sizeof(float)); hipMalloc(&ticks, 32 * sizeof(unsigned long long));
    hipMemset(ticks, 0, 32 * sizeof(unsigned long long));
#define RUN(C) hipLaunchKernelGGL(probe<C>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0f);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    hipDeviceSynchronize();
    unsigned long long h[32]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    const char* names[16] = {"fmac dependent chain", "fmac 4 independent chains", "fmac_dpp dependent acc (dpp src constant)", "s_nop 1 + fmac_dpp (dpp src = acc)",
                             "mov_dpp + fmac pair", "rcp dependent", "rcp 4 independent", "s_nop 1", "add_dpp row_shr chain + s_nop 1", "pk_fma dependent", "fmac 2 independent chains",
                             "mul_hi_u32 dependent", "cndmask dependent", "mad_u64_u32 dependent", "sqrt dependent", "exp dependent"};
    const int per[16] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
    for (int c = 0; c < 16; ++c) printf("%-44s %8.2f ticks per asm line (%llu total)\n", names[c], (double)h[c] / (16.0 * per[c]), h[c]);
    return 0;
}
