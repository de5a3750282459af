// Micro-benchmark: LDS accumulate forms on gfx950, per wave instruction and per CU.
//   mode 0  ds_add_f32            (no return)
//   mode 1  ds_add_u32            (no return)
//   mode 2  ds_add_u64            (no return)
//   mode 3  ds_read_b32 + v_add_f32 + ds_write_b32   (not atomic)
//   mode 4  ds_pk_add_f16 ... not used
// Addresses: every lane a different word of a 16 KB window, pseudo-random permutation per step (bank conflicts as in
// the pair kernel's j-side flush), or lane-linear (conflict-free).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    __shared__ float s_f[4096 * 2];
    for (int i = threadIdx.x; i < 8192; i += 512) s_f[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned a = (threadIdx.x * 2654435761u) >> 8;
    float v = 1.0f + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int idx;
            if (RANDOM) { // every lane a random word of the window (banks collide as they do in the pair kernel)
                a = a * 1664525u + 1013904223u;
                idx = (int)((a >> 12) & 4095u);
            } else {
                idx = (u * 64 + lane) & 4095;
            }
            if (MODE == 0) atomicAdd(&s_f[idx], v);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(s_f) + idx, (unsigned)lane);
            else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(s_f) + idx, (unsigned long long)lane);
            else if (MODE == 3) s_f[idx] += v;
        }
    }
    __syncthreads();
    float r = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 512) r += s_f[i];
    if (r == 12345.678f) out[0] = r;
}

template <int MODE, bool RANDOM>
void run(const char *name) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int bpc : {1, 2}) { // 512-thread blocks per CU: 2 / 4 waves per SIMD
        int blocks = 256 * bpc;
        hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(512), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)bpc * 8.0 * iters * 8.0;
        printf("%-28s %s blocks/CU=%d  %.3f ms  cycles(2.4GHz) per wave-instruction per CU = %.1f\n", name,
               RANDOM ? "scattered" : "linear   ", bpc, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
}

int main() {
    run<0, false>("ds_add_f32");
    run<0, true>("ds_add_f32");
    run<1, false>("ds_add_u32");
    run<1, true>("ds_add_u32");
    run<2, false>("ds_add_u64");
    run<2, true>("ds_add_u64");
    run<3, false>("read+add+write (non-atomic)");
    run<3, true>("read+add+write (non-atomic)");
    return 0;
}
