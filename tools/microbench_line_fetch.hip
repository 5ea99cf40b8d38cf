// tools/microbench_line_fetch.hip -- hipcc --offload-arch=gfx950 -O3 -o line_fetch tools/microbench_line_fetch.hip
// Microbenchmark: is a 16-byte-per-lane load cheaper when the 4 lanes of a quad read one contiguous 64-byte line
// than when every lane reads its own line? (config 5's extend is bound by such loads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: lane reads the 4 pieces of its own line (4 loads, divergent lines)
// mode 1: for k in 0..3: the quad reads line of its lane k, lane j takes piece j (4 loads, contiguous per quad)
template <int MODE> __global__ void kern(const float4 *table, const uint32_t *idx, uint32_t n_lines, uint32_t iters, float *out, uint32_t active_mask) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t line = idx[tid] % n_lines;
    float acc = 0.0f;
    const bool active = (active_mask >> (lane & 31u)) & 1u;
    for (uint32_t it = 0; it < iters; ++it) {
        if (MODE == 0) {
            if (active) {
                const float4 *p = table + 4u * line;
                const float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x + b.y + c.z + d.w;
                line = (line * 1664525u + 1013904223u + __float_as_uint(a.x)) % n_lines;
            }
        } else {
            float s = 0.0f, mine = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t src = __shfl(line, (lane & ~3u) + k);
                const bool src_active = __shfl(active ? 1 : 0, (lane & ~3u) + k);
                if (src_active) {
                    const float4 v = table[4u * src + (lane & 3u)];
                    s += v.x + v.y + v.z + v.w;
                    if ((lane & 3u) == 0 && false) mine = v.x;
                    if (k == (int)(lane & 3u)) mine = v.x; // stands in for the transpose: keep one value of the own line
                }
            }
            acc += s;
            if (active) line = (line * 1664525u + 1013904223u + __float_as_uint(mine)) % n_lines;
        }
    }
    out[tid] = acc;
}

int main() {
    const uint32_t n_lines = 1u << 20; // 64 MB table: lives in Infinity Cache, like the BVH bottom
    const uint32_t blocks = 1024, threads = 512, iters = 2000;
    std::vector<float4> h(4u * n_lines);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    std::vector<uint32_t> hi(blocks * threads);
    for (auto &v : hi) v = rand();
    float4 *t; uint32_t *idx; float *out;
    CK(hipMalloc(&t, sizeof(float4) * h.size())); CK(hipMalloc(&idx, 4 * hi.size())); CK(hipMalloc(&out, 4 * hi.size()));
    CK(hipMemcpy(t, h.data(), sizeof(float4) * h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(idx, hi.data(), 4 * hi.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t masks[3] = {0xffffffffu, 0x55555555u, 0x11111111u}; // all lanes, every 2nd, every 4th (one per quad)
    for (int m = 0; m < 3; ++m) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_lines, iters, out, masks[m]);
                else hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_lines, iters, out, masks[m]);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double active = (m == 0 ? 1.0 : (m == 1 ? 0.5 : 0.25));
            const double steps = double(blocks) * threads * iters * active;
            printf("active lanes %.0f%%  mode %s: %.3f ms  %.1f G lane-steps/s\n", active * 100, mode == 0 ? "own-line (4 divergent loads)" : "quad-cooperative", best, steps / best / 1e6);
        }
    }
    return 0;
}
