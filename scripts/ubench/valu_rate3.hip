// VALU issue-rate micro-benchmark #3: which "cheap" ops are really full rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed, unsigned long long *clk, float thr) {
    float a[8]; int b[8];
    for (int j = 0; j < 8; ++j) { a[j] = seed + threadIdx.x + j; b[j] = threadIdx.x * 7 + j; }
    const float m = 0.999f, c = 0.001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        const bool cond = (i & 1);  // scalar condition
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) a[j] = __saturatef(fmaf(a[j], m, c));                 // v_fma clamp
            if (MODE == 1) a[j] = cond ? a[j] : a[(j + 1) & 7];                  // v_cndmask with scalar mask
            if (MODE == 2) b[j] = b[j] & 0x7fff7;                                // v_and_b32
            if (MODE == 3) b[j] = (b[j] << 2) + j;                               // v_lshl_add_u32
            if (MODE == 4) a[j] = a[j] - c;                                      // v_sub_f32
            if (MODE == 5) a[j] = __builtin_fminf(a[j], thr);                    // v_min_f32 (maybe + canonicalize)
            if (MODE == 6) b[j] = __builtin_amdgcn_alignbit(b[j], b[(j + 1) & 7], 31); // v_alignbit
            if (MODE == 7) b[j] += (a[j] > thr) ? 1 : 0;                          // v_cmp + v_addc/cndmask
            if (MODE == 8) a[j] = __builtin_amdgcn_fmed3f(a[j], c, thr);          // v_med3_f32
            if (MODE == 9) b[j] = b[j] + 0x1234;                                  // v_add_u32
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int j = 0; j < 8; ++j) r += a[j] + b[j];
    if (r == 12345.678f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE>
void run(const char *name) {
    float *out; unsigned long long *clk, hclk[2];
    (void)hipMalloc(&out, 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {8}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f, clk, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, clk, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0);
        double per_simd = (double)wps * iters * 8.0;
        printf("%-22s waves/SIMD=%d  %.3f ms  clock %.2f GHz  cycles per source-op per SIMD=%.2f\n", name, wps, ms, ghz,
               ms * 1e-3 * ghz * 1e9 / per_simd);
    }
}
int main() {
    run<0>("fma+clamp"); run<1>("cndmask(scalar cond)"); run<2>("v_and_b32"); run<3>("v_lshl_add_u32"); run<4>("v_sub_f32");
    run<5>("v_min_f32"); run<6>("v_alignbit"); run<7>("cmp+add-carry"); run<8>("v_med3_f32"); run<9>("v_add_u32");
    return 0;
}
