// mmx_md.hpp -- SURVEY 8(f4): MD integrators on the same force kernels.
// Reference call sites: integrator choice model.py:768-808 (LangevinIntegrator(T, friction, dt) is the
// default, config.py:258-266), context.setVelocitiesToTemperature model.py:878, simulation.step model.py:931.
// Update rules restate OpenMM's leap-frog integrators [upstream: openmm 8.5.1 ReferenceStochasticDynamics /
// ReferenceVerletDynamics / ReferenceBrownianDynamics; no constraints exist in this system]:
//   langevin : v' = a v + (1-a)/gamma F/m + sqrt(kT (1-a^2)/m) N(0,1),  a = exp(-gamma dt);  x' = x + v' dt
//   verlet   : v' = v + dt F/m;                                                               x' = x + v' dt
//   brownian : x' = x + dt/(gamma m) F + sqrt(2 kT dt/(gamma m)) N(0,1);                      v' = (x'-x)/dt
// Random numbers: Philox4x32-10 keyed by the seed, counter = {bead, step_lo, step_hi, stream}: the noise of
// a bead at a step does not depend on the launch geometry or on how beads are split over GPUs.
// (OpenMM's own generator cannot be reproduced bit-for-bit: trajectories agree with it in distribution only.)
#pragma once
#include "mmx_cells.hpp"

namespace mmx {

enum MdKind { MD_LANGEVIN = 0, MD_VERLET = 1, MD_BROWNIAN = 2, MD_AMD = 3 };

struct MdParams {
    float dt;       // ps
    float vscale;   // langevin: exp(-gamma dt)
    float fscale;   // force -> velocity (langevin: (1-a)/(gamma m); verlet: dt/m) or -> displacement (brownian)
    float noise;    // per-component noise amplitude (velocity for langevin, displacement for brownian)
    float inv_dt;
    uint32_t key0, key1;       // seed
    uint32_t step_lo, step_hi; // step index of this launch
    double amd_alpha, amd_e;   // aMD boost: f' = f (alpha / (alpha + E - U))^2 while U < E (kJ/mol)
};

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Three standard normals from one Philox block (Box-Muller on 24-bit uniforms, exactly representable in fp32).
__device__ __forceinline__ void normal3(uint32_t bead, uint32_t step_lo, uint32_t step_hi, uint32_t stream,
                                        uint32_t k0, uint32_t k1, float z[3]) {
    uint32_t r[4];
    philox4x32_10(bead, step_lo, step_hi, stream, k0, k1, r);
    const float s = 1.0f / 16777216.0f;
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * s, u2 = ((float)(r[1] >> 8) + 0.5f) * s;
    const float u3 = ((float)(r[2] >> 8) + 0.5f) * s, u4 = ((float)(r[3] >> 8) + 0.5f) * s;
    const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float sn, cs;
    sincosf(6.2831853071795865f * u2, &sn, &cs);
    z[0] = ra * cs;
    z[1] = ra * sn;
    z[2] = rb * cosf(6.2831853071795865f * u4);
}

// One integrator step fused with the position pack of the next force evaluation:
// reads the gradient of the current positions, advances v and x (x carried as hi + lo: a step moves a bead
// by ~1e-5 nm while an fp32 ulp at 10 nm is 1e-6 nm), writes pos4 and the per-block bounding box.
// Algorithmic traffic: read 12 B x + 12 B xlo + 12 B v + 12 B g + 1 B label, write 12+12+12+16 B = 101 B/bead.
template <int KIND, bool COUNT = false>
__global__ __launch_bounds__(256) void k_md_pack(int n_own, const Own own, float *__restrict__ x, float *__restrict__ xlo,
                                                 float *__restrict__ v, const float *__restrict__ g,
                                                 const int8_t *__restrict__ labels, float4 *__restrict__ pos4,
                                                 float *__restrict__ bbox_part, const MdParams M,
                                                 const double *__restrict__ epot, // aMD: U of the current positions
                                                 const GridParams *__restrict__ grid = nullptr,
                                                 int *__restrict__ cell_of = nullptr, int *__restrict__ rank = nullptr,
                                                 int *__restrict__ count = nullptr) {
    __shared__ float s_bb[6][4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < n_own;
    float p[3] = {0.f, 0.f, 0.f};
    float boost = 1.f;
    if (KIND == MD_AMD) { // mm.amd.AMDIntegrator: modify = step(E - energy); (alpha / (alpha + E - energy))^2
        const double u = *epot;
        if (M.amd_e - u >= 0.0) {
            const double r = M.amd_alpha / (M.amd_alpha + M.amd_e - u);
            boost = (float)(r * r);
        }
    }
    if (act) {
        const int bead = own.bead(i);
        float z[3] = {0.f, 0.f, 0.f};
        if (KIND == MD_LANGEVIN || KIND == MD_BROWNIAN) normal3((uint32_t)bead, M.step_lo, M.step_hi, 0u, M.key0, M.key1, z);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float f = -g[3 * i + k];
            float vel = v[3 * i + k], dx;
            if (KIND == MD_LANGEVIN) {
                vel = fmaf(M.vscale, vel, fmaf(M.fscale, f, M.noise * z[k]));
                dx = vel * M.dt;
            } else if (KIND == MD_VERLET) {
                vel = fmaf(M.fscale, f, vel);
                dx = vel * M.dt;
            } else if (KIND == MD_AMD) {
                vel = fmaf(M.fscale * boost, f, vel);
                dx = vel * M.dt;
            } else {
                dx = fmaf(M.fscale, f, M.noise * z[k]);
                vel = dx * M.inv_dt;
            }
            const float hi = x[3 * i + k], lo = xlo[3 * i + k] + dx;
            const float nh = hi + lo;
            p[k] = nh;
            x[3 * i + k] = nh;
            xlo[3 * i + k] = lo - (nh - hi);
            v[3 * i + k] = vel;
        }
        pos4[bead] = make_float4(p[0], p[1], p[2], __int_as_float((bead << 3) | ((int)labels[bead] + 2)));
    }
    if (COUNT) { // single GPU: cell assignment of k_cell_count fused in (see k_pack)
        const GridParams G = *grid;
        int c = 0;
        if (act) {
            c = (cell_coord(p[2], G.oz, G.inv_h, G.nz) * G.ny + cell_coord(p[1], G.oy, G.inv_h, G.ny)) * G.nx +
                cell_coord(p[0], G.ox, G.inv_h, G.nx);
            cell_of[i] = c;
        }
        cell_rank(act, c, i, rank, count);
    }
    const float big = 3.0e38f;
    const bool fin = act && fabsf(p[0]) < big && fabsf(p[1]) < big && fabsf(p[2]) < big;
    float bb[6] = {fin ? p[0] : big, fin ? p[1] : big, fin ? p[2] : big, fin ? p[0] : -big, fin ? p[1] : -big,
                   fin ? p[2] : -big};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bb[k] = wave_min(bb[k]);
        bb[k + 3] = wave_max(bb[k + 3]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_bb[k][wave] = bb[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float r = s_bb[k][0];
        for (int w = 1; w < 4; ++w) r = k < 3 ? fminf(r, s_bb[k][w]) : fmaxf(r, s_bb[k][w]);
        bbox_part[k * gridDim.x + blockIdx.x] = r;
    }
}

// context.setVelocitiesToTemperature(T, seed): v = sqrt(kT/m) N(0,1) per component (stream 1, step 0).
__global__ __launch_bounds__(256) void k_md_init_velocities(int n_own, const Own own, float sigma, uint32_t key0,
                                                            uint32_t key1, float *__restrict__ v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_own) return;
    float z[3];
    normal3((uint32_t)own.bead(i), 0u, 0u, 1u, key0, key1, z);
    v[3 * i] = sigma * z[0];
    v[3 * i + 1] = sigma * z[1];
    v[3 * i + 2] = sigma * z[2];
}

// Kinetic energy 1/2 m |v + shift F/m|^2 summed in fp64 (OpenMM reports leap-frog velocities shifted by
// half a step: shift = dt/2 for langevin/verlet, 0 for brownian).  One partial per block.
__global__ __launch_bounds__(256) void k_md_kinetic(int n_own, const float *__restrict__ v, const float *__restrict__ g,
                                                    float shift_over_m, double half_m, double *__restrict__ part) {
    __shared__ double s_w[4];
    double acc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_own; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double w = (double)v[3 * i + k] - (double)shift_over_m * (double)g[3 * i + k];
            acc += w * w;
        }
    }
    const double r = block_sum<256>(acc, s_w);
    if (threadIdx.x == 0) part[blockIdx.x] = half_m * r;
}

// Folds the kinetic partials of one rank into out[0] (single block; fixed order => deterministic).
__global__ __launch_bounds__(256) void k_md_kinetic_fold(int nblk, const double *__restrict__ part,
                                                         double *__restrict__ out) {
    __shared__ double s_w[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) acc += part[b];
    const double r = block_sum<256>(acc, s_w);
    if (threadIdx.x == 0) out[0] = r;
}

} // namespace mmx
