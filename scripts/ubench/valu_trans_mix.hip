// Does a transcendental (quarter-rate) op overlap with plain VALU ops of the same wave / of other waves on gfx950?
// Loop body: 1 v_rcp_f32 (chain a) + N independent v_fma_f32 (chains b[]).  If the cost is additive the time grows as
// T_rcp + N*T_fma; if the trans unit overlaps it stays at max(T_rcp, N*T_fma).  (compile with -fno-slp-vectorize)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[4], b[8];
    for (int j = 0; j < 4; ++j) a[j] = seed + threadIdx.x + j;
    for (int j = 0; j < 8; ++j) b[j] = seed * 0.5f + threadIdx.x + j;
    const float m = 0.999f, c = 0.001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = __builtin_amdgcn_rcpf(a[t]);
#pragma unroll
            for (int j = 0; j < N; ++j) b[(t * N + j) & 7] = fmaf(b[(t * N + j) & 7], m, c);
        }
    }
    float r = 0;
    for (int j = 0; j < 4; ++j) r += a[j];
    for (int j = 0; j < 8; ++j) r += b[j];
    if (r == 12345.678f) out[0] = r;
}
template <int N>
void run() {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 4, 8}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves, iters*4 groups of (1 rcp + N fma)
        double ns_per_group = ms * 1e6 / ((double)wps * iters * 4.0);
        printf("1 rcp + %d fma  waves/SIMD=%d  %.3f ms  %.2f ns per group per SIMD (= %.1f cycles at 2.4 GHz)\n", N, wps, ms,
               ns_per_group, ns_per_group * 2.4);
    }
}
int main() { run<0>(); run<1>(); run<2>(); run<4>(); run<6>(); run<8>(); return 0; }
