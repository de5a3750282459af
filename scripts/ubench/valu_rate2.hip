// VALU issue-rate micro-benchmark #2 (compile with -fno-slp-vectorize): plain fp32 ops, transcendental ops,
// compare+select.  Prints cycles per wave-instruction per SIMD assuming a given clock.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed, unsigned long long *clk) {
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x + j;
    const float m = 0.999f, c = 0.001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) a[j] = fmaf(a[j], m, c);
            if (MODE == 1) a[j] = a[j] * m;
            if (MODE == 2) a[j] = a[j] + c;
            if (MODE == 3) a[j] = __builtin_amdgcn_rcpf(a[j]);
            if (MODE == 4) a[j] = __builtin_amdgcn_exp2f(a[j]);
            if (MODE == 5) a[j] = a[j] > c ? a[j] : m;          // v_cmp + v_cndmask
            if (MODE == 6) a[j] = fmaxf(a[j], c);                // v_max
            if (MODE == 7) a[j] = __builtin_amdgcn_sqrtf(a[j]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int j = 0; j < 8; ++j) r += a[j];
    if (r == 12345.678f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE>
void run(const char *name, int instr_per_op) {
    float *out; unsigned long long *clk, hclk[2];
    (void)hipMalloc(&out, 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 4, 8}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f, clk);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0) ; // memrealtime ticks at 100 MHz
        double instr_per_simd = (double)wps * iters * 8.0 * instr_per_op;
        printf("%-16s waves/SIMD=%d  %.3f ms  in-kernel clock %.2f GHz  cycles/instr/SIMD(actual clock)=%.2f\n", name, wps, ms, ghz,
               ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
}
int main() {
    run<0>("v_fma_f32", 1); run<1>("v_mul_f32", 1); run<2>("v_add_f32", 1); run<3>("v_rcp_f32", 1);
    run<4>("v_exp_f32", 1); run<5>("v_cmp+v_cndmask", 2); run<6>("v_max_f32", 1); run<7>("v_sqrt_f32", 1);
    return 0;
}
