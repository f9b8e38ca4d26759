// Microbenchmark: does s_setprio change how two wavefronts that share a SIMD are served?
// One workgroup of 8 wavefronts per CU (wave i -> SIMD i % 4: waves 0 and 4 share a SIMD); the waves of the first half
// optionally raise their priority; every wave runs the same dependent-FMA loop and records when it finished.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(float* out, long long* t_end, int raise, int iters)
{
    const int wave = threadIdx.x >> 6;
    if (raise && wave < 4) __builtin_amdgcn_s_setprio(3);
    // eight independent chains: one wavefront alone can keep the SIMD's issue port busy
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        x0 = x0 * b + c; x1 = x1 * b + c; x2 = x2 * b + c; x3 = x3 * b + c; x4 = x4 * b + c; x5 = x5 * b + c; x6 = x6 * b + c; x7 = x7 * b + c;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if ((threadIdx.x & 63) == 0) t_end[blockIdx.x * 8 + wave] = t1 - t0;
}
int main()
{
    const int blocks = 256, iters = 200000;
    float* out; long long* t;
    hipMalloc(&out, blocks * 512 * sizeof(float)); hipMalloc(&t, blocks * 8 * sizeof(long long));
    std::vector<long long> h(blocks * 8);
    for (int raise = 0; raise < 2; ++raise) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, t, raise, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), t, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double lo = 0, hi = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
        printf("s_setprio(3) on waves 0-3: %s   mean duration waves 0-3: %.0f ticks, waves 4-7: %.0f ticks (%d x 8 FMA)\n", raise ? "yes" : "no ", lo / (blocks * 4), hi / (blocks * 4), iters);
    }
    return 0;
}
