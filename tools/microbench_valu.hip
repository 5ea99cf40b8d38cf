// tools/microbench_valu.hip -- hipcc --offload-arch=gfx950 -O3 -o build/valu tools/microbench_valu.hip
// What does one SIMD of gfx950 sustain in wave64 VALU instructions per cycle, for the instruction kinds the traversal
// loop is made of (fp32 fma / min / max3 / compare + select), with 1 .. 8 waves per SIMD? The traversal's PMC profile
// reads SQ_ACTIVE_INST_VALU ~= SQ_INSTS_VALU (in quad-cycles): is the kernel at the VALU issue ceiling or at half of it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kUnroll = 16; // independent chains per lane

// MODE 0: v_fma_f32   1: v_min_f32 / v_max_f32   2: v_max3_f32 / v_min3_f32   3: v_cmp + v_cndmask   4: the slab-test mix
template <int MODE> __global__ __launch_bounds__(256) void kern(float *out, uint32_t iters, float a, float b) {
    float x[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) x[k] = a + static_cast<float>(threadIdx.x + k);
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            if (MODE == 0) x[k] = __builtin_fmaf(x[k], a, b);
            if (MODE == 1) x[k] = (k & 1) ? __builtin_fminf(x[k], a + x[(k + 1) % kUnroll]) : __builtin_fmaxf(x[k], b);
            if (MODE == 2) x[k] = __builtin_fmaxf(__builtin_fmaxf(x[k], a), x[(k + 3) % kUnroll]);
            if (MODE == 3) x[k] = x[k] > x[(k + 5) % kUnroll] ? a : x[k] + b;
            if (MODE == 4) { // fma, fma, min, max per axis pair as in hit_bvh_node_fma
                const float t0 = __builtin_fmaf(x[k], a, b), t1 = __builtin_fmaf(x[(k + 1) % kUnroll], a, b);
                x[k] = __builtin_fminf(t0, t1) + __builtin_fmaxf(t0, t1) * 1e-9f;
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, int insts_per_iter_per_chain, float *d_out, int cus) {
    const uint32_t iters = 4096;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        // 256-thread blocks = 4 waves = one per SIMD; waves_per_simd blocks per CU
        const int blocks = cus * waves_per_simd;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 64u, 1.0001f, 0.5f); // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_insts = static_cast<double>(iters) * kUnroll * insts_per_iter_per_chain * waves_per_simd; // per SIMD
        // cycles per wave64 instruction per SIMD, at the nominal 2.4 GHz (the chip may clock lower under load)
        printf("%-28s waves/SIMD %d: %8.3f ms  %6.2f cycles/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
               ms * 1e-3 * 2.4e9 / wave_insts);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *d_out;
    CK(hipMalloc(&d_out, sizeof(float) * 256 * cus * 8));
    printf("device: %s, %d CUs\n", prop.name, cus);
    if (run<0>("v_fma_f32", 1, d_out, cus)) return 1;
    if (run<1>("v_min/v_max (+add)", 1, d_out, cus)) return 1;  // odd chains carry an extra add: ~1.5 per chain step
    if (run<2>("v_max3_f32", 1, d_out, cus)) return 1;
    if (run<3>("v_cmp + v_add + v_cndmask", 3, d_out, cus)) return 1;
    if (run<4>("2 fma + min + max + fma", 5, d_out, cus)) return 1;
    return 0;
}
