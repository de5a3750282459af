// Micro-benchmark: VALU issue rates on gfx950 (plain vs packed fp32, transcendental), by occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f, c = 0.001f;
    const v2f pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 8 independent v_fma_f32
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        } else if (MODE == 1) { // 8 independent v_pk_fma_f32
            p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
            p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            p4 = __builtin_elementwise_fma(p4, pm, pc); p5 = __builtin_elementwise_fma(p5, pm, pc);
            p6 = __builtin_elementwise_fma(p6, pm, pc); p7 = __builtin_elementwise_fma(p7, pm, pc);
        } else if (MODE == 2) { // 8 v_rsq_f32
            a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
            a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
        } else if (MODE == 3) { // 8 v_pk_mul_f32
            p0 = p0 * pm; p1 = p1 * pm; p2 = p2 * pm; p3 = p3 * pm; p4 = p4 * pm; p5 = p5 * pm; p6 = p6 * pm; p7 = p7 * pm;
        } else if (MODE == 4) { // 8 v_cndmask via compare
            a0 = a0 > c ? a0 * m : a1; a1 = a1 > c ? a1 * m : a2; a2 = a2 > c ? a2 * m : a3; a3 = a3 > c ? a3 * m : a0;
            a4 = a4 > c ? a4 * m : a5; a5 = a5 > c ? a5 * m : a6; a6 = a6 > c ? a6 * m : a7; a7 = a7 > c ? a7 * m : a4;
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x +
              p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const char *name, int ops_per_iter, int flops_per_op) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpb : {1, 2, 4, 8}) { // waves per SIMD: blocks of 256 threads = 4 waves = 1 wave/SIMD per block
        int blocks = 256 * wpb;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double waves = blocks * 4.0;
        double wave_instr = waves * iters * (double)ops_per_iter;
        double cyc_per_instr_per_simd = (ms * 1e-3 * 2.4e9) / (wave_instr / 1024.0);
        double tflops = wave_instr * 64.0 * flops_per_op / (ms * 1e-3) / 1e12;
        printf("%-14s waves/SIMD=%d  %.3f ms  cycles(2.4GHz)/wave-instr/SIMD=%.2f  %.1f TFLOP/s\n", name, wpb, ms,
               cyc_per_instr_per_simd, tflops);
    }
}

int main() {
    run<0>("v_fma_f32", 8, 2);
    run<1>("v_pk_fma_f32", 8, 4);
    run<2>("v_rsq_f32", 8, 1);
    run<3>("v_pk_mul_f32", 8, 2);
    run<4>("cmp+mul+cndmask", 24, 1);
    return 0;
}
