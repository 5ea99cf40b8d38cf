// tools/microbench_node_fetch.hip -- hipcc --offload-arch=gfx950 -O3 -o build/node_fetch tools/microbench_node_fetch.hip
// Config 5's extend is a chain of dependent random node fetches out of a table that lives in the Infinity Cache. How
// does the rate of such fetches change with the record size -- 64 B (a sibling pair of the binary BVH, what extend reads
// per visit today) against 128 B (a four-wide node: half as many visits per ray) and 256 B (eight-wide)? Every lane
// chases its own chain; the next index depends on the data just read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VEC4> __global__ __launch_bounds__(512) void chase(const float4 *table, const uint32_t *idx, uint32_t n_rec, uint32_t iters, float *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rec = idx[tid] % n_rec;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; ++it) {
        const float4 *p = table + static_cast<size_t>(VEC4) * rec;
        float4 v[VEC4];
#pragma unroll
        for (int k = 0; k < VEC4; ++k) v[k] = p[k];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < VEC4; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
        acc += s;
        rec = (rec * 1664525u + 1013904223u + __float_as_uint(s)) % n_rec;
    }
    out[tid] = acc;
}

int main() {
    const size_t table_bytes = 64u << 20; // 64 MB: Infinity-Cache resident, like the bottom of the 1M-triangle BVH
    const uint32_t blocks = 1024, threads = 512, iters = 1000;
    std::vector<float4> h(table_bytes / 16, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> hi(blocks * threads);
    for (auto &v : hi) v = rand();
    float4 *t; uint32_t *idx; float *out;
    CK(hipMalloc(&t, table_bytes)); CK(hipMalloc(&idx, 4 * hi.size())); CK(hipMalloc(&out, 4 * hi.size()));
    CK(hipMemcpy(t, h.data(), table_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx, hi.data(), 4 * hi.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int vec4 : {4, 8, 16}) {
        float best = 1e30f;
        const uint32_t n_rec = static_cast<uint32_t>(table_bytes / (16u * vec4));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (vec4 == 4) hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
            if (vec4 == 8) hipLaunchKernelGGL(chase<8>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
            if (vec4 == 16) hipLaunchKernelGGL(chase<16>, dim3(blocks), dim3(threads), 0, 0, t, idx, n_rec, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double steps = double(blocks) * threads * iters;
        printf("record %3d B: %8.3f ms  %6.1f G records/s  %6.2f TB/s\n", 16 * vec4, best, steps / best / 1e6, steps * 16 * vec4 / best / 1e9);
    }
    return 0;
}
