// Micro-benchmark (diagnostic, not product): what a wave-uniform branch costs on gfx950 next to plain VALU work, at one and two
// wavefronts per SIMD: not taken / taken forward, condition from the scalar unit (s_cmp) or from a vector compare (v_cmp -> vcc).
//   hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip && ./branch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define FMA8 "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" \
             "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
template <int MODE>
__global__ void k(float* out, int n, int z)
{
    float a0 = threadIdx.x, a1 = a0 + 1;
    const float m = 0.999f, c = 0.001f;
    float one = z ? 0.f : 1.f; // > 0 in every lane
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { // 8 fma, no branch
            REP8(asm volatile(FMA8 : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));)
        } else if (MODE == 1) { // 8 fma + scalar compare + branch not taken
            REP8(asm volatile(FMA8 "s_cmp_lg_u32 %4, 0\n s_cbranch_scc1 1f\n 1:\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c), "s"(z) : "scc");)
        } else if (MODE == 2) { // 8 fma + scalar compare + branch taken over one instruction
            REP8(asm volatile(FMA8 "s_cmp_eq_u32 %4, 0\n s_cbranch_scc1 1f\n v_fma_f32 %0, %0, %2, %3\n 1:\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c), "s"(z) : "scc");)
        } else if (MODE == 3) { // 8 fma + v_cmp -> vcc + branch not taken (vccz false: some lane set)
            REP8(asm volatile(FMA8 "v_cmp_lt_f32 vcc, 0, %4\n s_cbranch_vccz 1f\n 1:\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c), "v"(one) : "vcc");)
        } else if (MODE == 4) { // 8 fma + v_cmp -> vcc + branch taken (no lane set) over one instruction
            REP8(asm volatile(FMA8 "v_cmp_gt_f32 vcc, 0, %4\n s_cbranch_vccz 1f\n v_fma_f32 %0, %0, %2, %3\n 1:\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c), "v"(one) : "vcc");)
        } else if (MODE == 5) { // 8 fma + v_cmp + v_cndmask (the branch-free alternative: select instead of skip)
            REP8(asm volatile(FMA8 "v_cmp_gt_f32 vcc, 0, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c), "v"(one) : "vcc");)
        } else if (MODE == 6) { // 8 fma + ds_bpermute round trip used at once
            REP8(asm volatile(FMA8 "ds_bpermute_b32 %0, %4, %0\n s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c), "v"((int)(threadIdx.x & 63) * 4));)
        } else if (MODE == 7) { // 8 fma + 3 dependent DPP adds (a class sum's row part)
            REP8(asm volatile(FMA8 "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                                   "v_add_f32_dpp %0, %0, %0 row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                                   "v_add_f32_dpp %0, %0, %0 row_shr:12 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));)
        } else if (MODE == 8) { // 8 fma + v_readlane -> SGPR used by a VALU op (a reduction's broadcast)
            REP8(asm volatile(FMA8 "v_readlane_b32 s20, %0, 63\n s_nop 3\n v_mul_f32 %1, s20, %1\n" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c) : "s20");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
}
template <int MODE>
void run(const char* name, float* out)
{
    const int n = 4000;
    double base[3] = {0, 0, 0};
    for (int wps = 1; wps <= 2; wps *= 2) {
        const int block = 64 * 4 * wps, blocks = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, 10, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, n, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc_per_group_simd = ms * 1e6 * 2.4 / ((double)n * 8 * wps); // cycles of SIMD time per (8 fma + extra) group
        printf("%-64s waves/SIMD %d: %8.3f ms  -> %6.2f SIMD cycles per group of 8 fma + extra (@2.4GHz)\n", name, wps, ms, cyc_per_group_simd);
        (void)base;
    }
}
int main()
{
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    run<0>("8 fma", out);
    run<1>("8 fma + s_cmp + s_cbranch not taken", out);
    run<2>("8 fma + s_cmp + s_cbranch taken (skips 1 instr)", out);
    run<3>("8 fma + v_cmp + s_cbranch_vccz not taken", out);
    run<4>("8 fma + v_cmp + s_cbranch_vccz taken (skips 1 instr)", out);
    run<5>("8 fma + v_cmp + s_nop 1 + v_cndmask", out);
    run<6>("8 fma + ds_bpermute + wait", out);
    run<7>("8 fma + 3 dependent DPP adds with their s_nops", out);
    run<8>("8 fma + v_readlane + s_nop 3 + v_mul with the SGPR", out);
    return 0;
}
