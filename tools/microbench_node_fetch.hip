// tools/microbench_node_fetch.hip -- hipcc --offload-arch=gfx950 -O3 -o build/node_fetch tools/microbench_node_fetch.hip
// Config 5's extend is a chain of dependent random node fetches out of a table that lives in the Infinity Cache. How
// does the rate of such fetches change with the record size -- 64 B (a sibling pair of the binary BVH, what extend reads
// per visit today) against 128 B (a four-wide node: half as many visits per ray) and 256 B (eight-wide)? Every lane
// chases its own chain; the next index depends on the data just read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
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

// ---- `node_fetch calib`: known byte counts in the refill traversal's access shape, for calibrating rocprofv3's FETCH_SIZE
// (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every lane reads `iters` independent random records of a 1 GiB table (32 x the L2s: practically every record misses L2) as
// VEC4 x 16-byte loads -- gather64: 64-byte records, what refill_kernel fetches per node visit; gather128: 128-byte records (a whole
// L2 line); stream16: the coalesced 16 B-per-lane streaming read the guide's factor 2 was measured on.
template <int VEC4> __global__ __launch_bounds__(512) void gather_known(const float4 *table, uint32_t n_rec, uint32_t iters, float *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    uint32_t h = tid * 2654435761u + 12345u;
    for (uint32_t it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const uint32_t rec = (h >> 4) % n_rec;
        const float4 *p = table + static_cast<size_t>(VEC4) * rec;
        float4 v[VEC4];
#pragma unroll
        for (int k = 0; k < VEC4; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < VEC4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    out[tid] = acc;
}
__global__ __launch_bounds__(256) void stream_known(const float4 *table, size_t n_vec, float *out) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n_vec; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const float4 v = table[i];
        acc += v.x + v.y + v.z + v.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static int calib() {
    const size_t table_bytes = 1ull << 30;
    const uint32_t blocks = 2048, threads = 512, iters = 64;
    float4 *t; float *out;
    CK(hipMalloc(&t, table_bytes)); CK(hipMalloc(&out, 4ull * blocks * threads));
    CK(hipMemset(t, 0, table_bytes));
    hipLaunchKernelGGL(gather_known<4>, dim3(blocks), dim3(threads), 0, 0, t, static_cast<uint32_t>(table_bytes / 64), iters, out);
    hipLaunchKernelGGL(gather_known<8>, dim3(blocks), dim3(threads), 0, 0, t, static_cast<uint32_t>(table_bytes / 128), iters, out);
    hipLaunchKernelGGL(stream_known, dim3(4096), dim3(256), 0, 0, t, table_bytes / 16, out);
    CK(hipDeviceSynchronize());
    const double reads = double(blocks) * threads * iters;
    printf("{\"gather64_bytes\": %.0f, \"gather128_bytes\": %.0f, \"stream16_bytes\": %.0f, \"table_bytes\": %.0f, \"records_read\": %.0f}\n",
           reads * 64, reads * 128, double(table_bytes), double(table_bytes), reads);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "calib") return calib();
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
