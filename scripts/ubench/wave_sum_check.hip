// Bitwise check of the DPP / permlane-swap butterfly (mmx_common.hpp: wave_sum, wave_min, wave_max) against the __shfl_xor
// butterfly it replaces: same pairings in the same order, so every lane must hold the same bits.
//   hipcc -O3 --offload-arch=gfx950 -I multimm_amd/csrc -o scripts/ubench/bin/wave_sum_check scripts/ubench/wave_sum_check.hip
#include "mmx_common.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace mmx;
__device__ __forceinline__ double ref_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ref_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ref_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float ref_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int ref_max_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__global__ void k(const double *d, const float *f, const int *ii, unsigned long long *bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double a = wave_sum(d[i]), b = ref_sum(d[i]);
    const float c = wave_sum(f[i]), e = ref_sum(f[i]);
    const float mn = wave_min(f[i]), mn2 = ref_min(f[i]), mx = wave_max(f[i]), mx2 = ref_max(f[i]);
    const int im = wave_max_i(ii[i]), im2 = ref_max_i(ii[i]);
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(bad, 1ull);
    if (__float_as_int(c) != __float_as_int(e)) atomicAdd(bad + 1, 1ull);
    if (__float_as_int(mn) != __float_as_int(mn2)) atomicAdd(bad + 2, 1ull);
    if (__float_as_int(mx) != __float_as_int(mx2)) atomicAdd(bad + 3, 1ull);
    if (im != im2) atomicAdd(bad + 4, 1ull);
    // lane_xor against __shfl_xor, every distance, 32- and 64-bit
    const unsigned long long key = ((unsigned long long)(unsigned)ii[i] << 32) | (unsigned)__float_as_int(f[i]);
    bool okx = true;
    okx = okx && lane_xor<1>(ii[i]) == __shfl_xor(ii[i], 1, 64) && lane_xor<2>(ii[i]) == __shfl_xor(ii[i], 2, 64);
    okx = okx && lane_xor<4>(ii[i]) == __shfl_xor(ii[i], 4, 64) && lane_xor<8>(ii[i]) == __shfl_xor(ii[i], 8, 64);
    okx = okx && lane_xor<16>(ii[i]) == __shfl_xor(ii[i], 16, 64) && lane_xor<32>(ii[i]) == __shfl_xor(ii[i], 32, 64);
    for (int j = 1; j < 64; j <<= 1) okx = okx && lane_xor_rt(key, j) == __shfl_xor(key, j, 64);
    if (!okx) atomicAdd(bad + 5, 1ull);
    // wave_sum12 against twelve wave_sum calls
    double v12[12], t12[3];
    for (int q = 0; q < 12; ++q) v12[q] = d[(i + 977 * q) & ((1 << 20) - 1)] * (q + 1);
    wave_sum12(v12, t12);
    bool ok12 = true;
    for (int q = 0; q < 12; ++q) {
        const double want = wave_sum(v12[q]);
        const int lane = threadIdx.x & 63;
        if ((lane >> 4) == q / 3 && __double_as_longlong(t12[q % 3]) != __double_as_longlong(want)) ok12 = false;
    }
    if (!ok12) atomicAdd(bad + 6, 1ull);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> d(n);
    std::vector<float> f(n);
    std::vector<int> ii(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const double m = (double)rand() / RAND_MAX - 0.5;
        d[i] = m * pow(10.0, rand() % 40 - 20);
        f[i] = (float)(((double)rand() / RAND_MAX - 0.5) * pow(10.0, rand() % 20 - 10));
        ii[i] = rand() - RAND_MAX / 2;
    }
    double *dd; float *df; int *di; unsigned long long *bad;
    hipMalloc(&dd, n * 8); hipMalloc(&df, n * 4); hipMalloc(&di, n * 4); hipMalloc(&bad, 7 * 8);
    hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(df, f.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(di, ii.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 7 * 8);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dd, df, di, bad);
    unsigned long long h[7];
    hipMemcpy(h, bad, 7 * 8, hipMemcpyDeviceToHost);
    printf("lanes that differ: sum f64 %llu, sum f32 %llu, min %llu, max %llu, max int %llu, lane_xor %llu, wave_sum12 %llu (of %d)\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], n);
    return (h[0] | h[1] | h[2] | h[3] | h[4] | h[5] | h[6]) ? 1 : 0;
}
