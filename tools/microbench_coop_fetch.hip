// tools/microbench_coop_fetch.hip -- hipcc --offload-arch=gfx950 -O3 -o build/coop_fetch tools/microbench_coop_fetch.hip
// What limits a wave whose 64 lanes each fetch their own random 64-byte record (a quantised four-wide BVH node) with four
// 16-byte loads? Every such load instruction touches 64 different cache lines. "coop" fetches the same 64 records with
// the same four instructions, but lane l of instruction k loads quarter (l & 3) of the record of lane 16 k + (l >> 2):
// an instruction now touches 16 lines, 64 contiguous bytes each; the quarters go through LDS to the lanes that own them.
// Tables: 16 KB (L1-resident), 2 MB (L2), 64 MB (Infinity Cache). Dependent chains, as in the traversal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t next_rec(uint32_t rec, float s, uint32_t n_rec) { return (rec * 1664525u + 1013904223u + __float_as_uint(s)) % n_rec; }

__global__ __launch_bounds__(512) void chase_direct(const float4 *table, const uint32_t *idx, uint32_t n_rec, uint32_t iters, float *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rec = idx[tid] % n_rec;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; ++it) {
        const float4 *p = table + 4u * static_cast<size_t>(rec);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        const float s = (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        acc += s;
        rec = next_rec(rec, s, n_rec);
    }
    out[tid] = acc;
}

// the same chase with a record of N x 16 bytes per lane (N = 3: a 48-byte node), records still 64 bytes apart
template <int N> __global__ __launch_bounds__(512) void chase_n(const float4 *table, const uint32_t *idx, uint32_t n_rec, uint32_t iters, float *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rec = idx[tid] % n_rec;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; ++it) {
        const float4 *p = table + 4u * static_cast<size_t>(rec);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) { const float4 v = p[k]; s += (v.x + v.y) + (v.z + v.w); }
        acc += s;
        rec = next_rec(rec, s, n_rec);
    }
    out[tid] = acc;
}

__global__ __launch_bounds__(512) void chase_coop(const float4 *table, const uint32_t *idx, uint32_t n_rec, uint32_t iters, float *out) {
    __shared__ float4 stage[8][256]; // per wave: 64 records x 4 quarters
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4 *st = stage[wave];
    uint32_t rec = idx[tid] % n_rec;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t r = __shfl(rec, 16u * k + (lane >> 2), 64);
            st[64u * k + lane] = table[4u * static_cast<size_t>(r) + (lane & 3u)]; // record 16 k + (lane >> 2), quarter lane & 3
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // record of lane j sits at st[4 j .. 4 j + 3]; rotate the quarter order by lane so that the 16 lanes of a phase spread over the banks
        const float4 a = st[4u * lane + ((0u + (lane >> 1)) & 3u)], b = st[4u * lane + ((1u + (lane >> 1)) & 3u)],
                     c = st[4u * lane + ((2u + (lane >> 1)) & 3u)], d = st[4u * lane + ((3u + (lane >> 1)) & 3u)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float s = (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        acc += s;
        rec = next_rec(rec, s, n_rec);
    }
    out[tid] = acc;
}

int main() {
    const uint32_t blocks = 1024, threads = 512, iters = 400;
    const size_t max_bytes = 64u << 20;
    std::vector<float4> h(max_bytes / 16, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> hi(blocks * threads);
    for (auto &v : hi) v = rand();
    float4 *t; uint32_t *idx; float *out;
    CK(hipMalloc(&t, max_bytes)); CK(hipMalloc(&idx, 4 * hi.size())); CK(hipMalloc(&out, 4 * hi.size()));
    CK(hipMemcpy(t, h.data(), max_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx, hi.data(), 4 * hi.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t bytes : {size_t(16) << 10, size_t(2) << 20, size_t(64) << 20}) {
        const uint32_t n_rec = static_cast<uint32_t>(bytes / 64u);
        for (int nload = 1; nload <= 3; ++nload) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (nload == 1) hipLaunchKernelGGL(chase_n<1>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
                if (nload == 2) hipLaunchKernelGGL(chase_n<2>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
                if (nload == 3) hipLaunchKernelGGL(chase_n<3>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double steps = double(blocks) * threads * iters;
            printf("table %6zu KB  %d x 16 B: %8.3f ms  %6.1f G records/s\n", bytes >> 10, nload, best, steps / best / 1e6);
        }
        for (int coop = 0; coop < 2; ++coop) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (coop) hipLaunchKernelGGL(chase_coop, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
                else hipLaunchKernelGGL(chase_direct, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double steps = double(blocks) * threads * iters;
            printf("table %6zu KB  %-6s: %8.3f ms  %6.1f G records/s  (%.2f records per clock per CU at 2.1 GHz)\n", bytes >> 10, coop ? "coop" : "direct", best,
                   steps / best / 1e6, steps / best / 1e6 / 256.0 / 2.1);
        }
    }
    return 0;
}
