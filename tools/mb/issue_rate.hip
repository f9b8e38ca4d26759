// Micro-benchmark (diagnostic, not product): VALU issue rate per SIMD of gfx950 as a function of resident wavefronts,
// for plain and packed fp32 FMAs, DPP adds and LDS reads -- the numbers DESIGN.md's issue-bound estimates rest on.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float* out, int n)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f, c = 0.001f;
    const float2_ pm = {m, m}, pc = {c, c};
    __shared__ float sh[1024];
    sh[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int idx = threadIdx.x & 63;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { // 8 independent v_fma_f32 per pass x 8
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 1) { // 8 independent v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        } else if (MODE == 2) { // one dependent chain of v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));)
        } else if (MODE == 3) { // dependent chain of v_add_f32_dpp (row_shr:1) with the required s_nop 1
            REP8(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0));)
        } else if (MODE == 4) { // 8 independent ds_read_b128 then one wait
            for (int r = 0; r < 8; ++r) {
                float4 q0 = *reinterpret_cast<float4*>(&sh[(idx * 4) & 1020]);
                float4 q1 = *reinterpret_cast<float4*>(&sh[(idx * 4 + 256) & 1020]);
                float4 q2 = *reinterpret_cast<float4*>(&sh[(idx * 4 + 512) & 1020]);
                float4 q3 = *reinterpret_cast<float4*>(&sh[(idx * 4 + 768) & 1020]);
                a0 += q0.x + q1.y + q2.z + q3.w;
                a1 += q0.y + q1.z + q2.w + q3.x;
                idx = (idx + 1) & 63;
            }
        } else if (MODE == 5) { // mixed: 6 fma + v_max + v_cndmask-ish, mimicking the ellipse body's flavour (compiler-scheduled)
            for (int r = 0; r < 8; ++r) {
                const float dx = a0 - a4, dy = a1 - a5;
                const float tx = a2 * dx + a3 * dy, ty = a3 * dx + a6 * dy;
                const float h = fmaxf(0.f, 1.f - (dx * tx + dy * ty));
                a7 += h * h * c;
                a4 += -4.f * c * h * tx;
                a5 += -4.f * c * h * ty;
                a0 = a0 * m + c;
                a1 = a1 * m - c;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
void run(const char* name, float* out, int instr_per_iter)
{
    const int n = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int block = 64 * 4 * (wps > 4 ? 4 : wps); // up to 16 waves per workgroup
        const int blocks = 256 * (wps > 4 ? wps / 4 : 1);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ns_per_instr_simd = ms * 1e6 / ((double)n * instr_per_iter * wps);
        printf("%-34s waves/SIMD %d: %8.3f ms  -> %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
               ns_per_instr_simd, ns_per_instr_simd * 2.4);
    }
}
int main()
{
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    run<0>("v_fma_f32 x8 independent", out, 64);
    run<1>("v_pk_fma_f32 x8 independent", out, 64);
    run<2>("v_fma_f32 dependent chain", out, 64);
    run<3>("v_add_f32_dpp dep chain (+s_nop 1)", out, 64);
    run<4>("ds_read_b128 x4 + adds", out, 8 * 12);
    run<5>("ellipse-like mixed body", out, 8 * 22);
    return 0;
}
