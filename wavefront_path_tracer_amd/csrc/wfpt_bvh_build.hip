// wfpt_bvh_build.hip -- the reference's BVH builder (wavefront_common/src/bvh.rs:50-210) on the device.
//
// Build extension. Same inputs and, byte for byte, the same outputs as the host
// builder in wfpt_host.cpp (node array in the reference's depth-first numbering, primitives reordered in
// place). bvh.rs is a sequential recursion; what makes a parallel restatement exact:
//   * boxes are min/max reductions and bin counts are integer sums: order-free;
//   * the SAH sweep picks the FIRST strict minimum in (axis, plane) order: an arg-min with index tie-break;
//   * the in-place partition `while i <= j { if left(i) { i += 1 } else { swap(i, j); j -= 1 } }`
//     (bvh.rs:176-186) produces a permutation with a closed form (partition_dest below), so every element's
//     destination follows from prefix counts;
//   * node numbers are the order of split events in a depth-first walk: the tree is built breadth-first with
//     provisional numbers and renumbered from subtree sizes at the end.
// Large nodes are processed level by level, 1024-primitive chunks per workgroup over per-primitive records that move
// with the partition, bins reduced with atomics on order-preserving integer images of the floats, one wave per node
// for the sweep. A node with <= 64 primitives (and <= 64 bins) is finished by ONE wave that runs bvh.rs's recursion
// for the whole subtree out of LDS.
// Zeros: the minimum of (+0, -0) is -0 and the maximum +0, here (the integer image orders them that way) and in the host
// builder alike (wfpt_host.cpp zmin / zmax); the reference's f32::min / max leave that choice open.
#include "wfpt_kernels.h"

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

namespace wfpt {
void set_last_error(const std::string &msg); // wfpt_api.hip

namespace {

constexpr uint32_t kBuildThreads = 256;
constexpr uint32_t kBuildChunk = 1024; // primitives per workgroup pass
constexpr uint32_t kSmallPrims = 64;   // subtrees of at most this many primitives are finished by one wave
constexpr uint32_t kWaveBins = 64;     // ... when the bin count fits one wave
constexpr uint32_t kDirectPrims = 8;   // inside such a subtree, nodes this small skip the bins (3 * 9 candidate lanes)
constexpr uint32_t kMaxBins = 4096;    // bvh.rs:4

// ---- order-preserving integer image of a float, so that min / max become integer atomics
__device__ __forceinline__ uint32_t ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
// Boxes under construction: lower bounds as ~ord (so atomicMax tracks the minimum), upper bounds as ord; an
// all-zero word is "empty" (+inf / -inf, Bin::default, bvh.rs:12-20), so a memset initialises any number of boxes.
__device__ __forceinline__ uint32_t enc_lo(float f) { return ~ord(f); }
__device__ __forceinline__ uint32_t enc_hi(float f) { return ord(f); }
__device__ __forceinline__ float dec_lo(uint32_t v) { return v == 0u ? __builtin_inff() : unord(~v); }
__device__ __forceinline__ float dec_hi(uint32_t v) { return v == 0u ? -__builtin_inff() : unord(v); }
// the same order on floats: -0 below +0 (wfpt_host.cpp zmin / zmax); operands are never NaN
__device__ __forceinline__ float zmin(float a, float b) { return a < b ? a : (b < a ? b : (__float_as_uint(a) >> 31 ? a : b)); }
__device__ __forceinline__ float zmax(float a, float b) { return a > b ? a : (b > a ? b : (__float_as_uint(a) >> 31 ? b : a)); }

struct BuildNode { // provisional (breadth-first) node
    uint32_t lo[3]; // enc_lo
    uint32_t first;
    uint32_t hi[3]; // enc_hi
    uint32_t count;
    uint32_t child; // provisional id of the left child; the right one is child + 1
    uint32_t kind;  // 0 leaf, 1 split, 2 root of a wave-built subtree
    uint32_t sub;   // kind 2: slot in the subtree table
    uint32_t _pad;
};

struct Decision {
    float plane;
    uint32_t axis;
    uint32_t partition; // 0: leaf by cost (bvh.rs:172-174), primitives untouched; 1: the partition loop runs
    uint32_t n_left;
    uint32_t split;     // children were made (bvh.rs:188-209)
    uint32_t child;
};

struct BuildCtl {
    uint32_t n_nodes;    // provisional nodes allocated
    uint32_t n_next;     // nodes queued for the next level
    uint32_t n_small;    // wave-built subtree roots queued
    uint32_t n_chunks;   // chunks of the level that was set up last
    uint32_t n_active;   // ... and its node count
    uint32_t sub_alloc;  // subtree node records allocated
    uint32_t overflow;   // a capacity was exceeded (nothing was written out of bounds)
};

struct BuildArgs {
    // per-primitive records in POSITION order (they move with the partition, so every pass streams them):
    // bin key, box; `ids` = which input primitive sits at each position. `t_*`: the scatter's destination copy.
    float *key[3], *plo[3], *phi[3];
    float *t_key[3], *t_plo[3], *t_phi[3];
    uint32_t n, n_bins, use_wave;
    uint32_t *ids, *tmp;
    BuildNode *nodes;
    uint32_t node_cap;
    BuildCtl *ctl;
    // current level
    const uint32_t *active; // provisional ids
    uint32_t n_active;
    uint32_t *next_active, *small_roots;
    uint32_t *chunk_first;  // [n_active + 1]
    uint32_t *chunk_node, *chunk_off;
    uint32_t chunk_cap;
    uint32_t *bins;         // [n_active][3][7][n_bins]: count, lo xyz, hi xyz
    Decision *dec;          // [n_active]
    uint32_t *chunk_left, *chunk_left_before;
    uint32_t *head_r, *tail_l; // [n + 2] rank -> position tables of the partition
    // wave-built subtrees
    wfpt_bvh_node *sub_nodes;
    uint32_t sub_cap;
    uint32_t *sub_base, *sub_pairs;
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Exclusive prefix sum over the 256 threads of a workgroup; `total` is the workgroup sum.
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *lds4, uint32_t &total) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads(); // lds4 may still be read from a previous call
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBuildThreads / 64; ++w) {
        const uint32_t t = lds4[w];
        before += (w < wave) ? t : 0u;
        total += t;
    }
    return before + incl - v;
}

__device__ __forceinline__ int bin_of(float key, float lo_bound, float scale, uint32_t n_bins) { // bvh.rs:95-99
    const float pos = (key - lo_bound) * scale;
    // Rust `as usize` saturates (negative and NaN -> 0), then .min(BINS - 1)
    return pos > 0.0f ? (pos >= static_cast<float>(n_bins) ? static_cast<int>(n_bins) - 1 : static_cast<int>(pos)) : 0;
}

__device__ __forceinline__ float half_area(float lx, float ly, float lz, float hx, float hy, float hz) { // Bin::get_area, bvh.rs:29-35
    if (!(isfinite(hx) && isfinite(hy) && isfinite(hz))) return 0.0f;
    const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
    return (ex * ey + ey * ez) + ez * ex;
}

// Where the element at position s (0-based inside its node) ends up after bvh.rs:176-186, given its class, the
// number of left elements before it, the node's totals and the two rank tables:
//   head_r[k] = position of the k-th right element among positions 0..n_left   (k = 1 + rights before it)
//   tail_l[k] = position of the k-th left element from the END among positions >= n_left
// Derivation: the loop examines every element once at cursor i; the rights it meets go to n-1, n-2, ... in
// the order met, and that order is: the next right of the head, then the elements fetched from the tail for as
// long as they are rights. (Checked exhaustively against the loop for all class strings up to length 12.)
__device__ __forceinline__ uint32_t partition_dest(bool is_left, uint32_t s, uint32_t left_before, uint32_t n_left,
                                                   uint32_t count, const uint32_t *head_r, const uint32_t *tail_l) {
    if (is_left) return s < n_left ? s : head_r[n_left - left_before];
    if (s > n_left) return s - 1u;
    const uint32_t k = 1u + (s - left_before);
    return k == 1u ? count - 1u : tail_l[k - 1u] - 1u;
}

// ---------------------------------------------------------------------------------------------- primitives
template <int PRIM>
__global__ __launch_bounds__(kBuildThreads) void prep_prims_kernel(const void *prims, uint32_t n, float *key0, float *key1,
                                                                  float *key2, float *lo0, float *lo1, float *lo2, float *hi0,
                                                                  float *hi1, float *hi2, uint32_t *ids, BuildNode *root) {
    __shared__ uint32_t s_box[6];
    if (threadIdx.x < 6) s_box[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * kBuildThreads + threadIdx.x;
    if (i < n) {
        float k[3], lo[3], hi[3];
        if (PRIM == 0) { // sphere.rs:22-26; binned by its centre (bvh.rs:97)
            const wfpt_sphere s = static_cast<const wfpt_sphere *>(prims)[i];
            for (int a = 0; a < 3; ++a) {
                k[a] = s.center[a];
                lo[a] = s.center[a] - s.radius;
                hi[a] = s.center[a] + s.radius;
            }
        } else { // build extension: box of the three vertices, binned by the centroid (wfpt_host.cpp TrianglePrims)
            const wfpt_triangle t = static_cast<const wfpt_triangle *>(prims)[i];
            for (int a = 0; a < 3; ++a) {
                const float va = t.v0[a], vb = t.v0[a] + t.e1[a], vc = t.v0[a] + t.e2[a];
                lo[a] = zmin(zmin(va, vb), vc);
                hi[a] = zmax(zmax(va, vb), vc);
                k[a] = t.v0[a] + (t.e1[a] + t.e2[a]) * 0.33333334f;
            }
        }
        key0[i] = k[0]; key1[i] = k[1]; key2[i] = k[2];
        lo0[i] = lo[0]; lo1[i] = lo[1]; lo2[i] = lo[2];
        hi0[i] = hi[0]; hi1[i] = hi[1]; hi2[i] = hi[2];
        ids[i] = i;
        for (int a = 0; a < 3; ++a) { // update_node_bounds of the root, bvh.rs:58-70
            atomicMax(&s_box[a], enc_lo(lo[a]));
            atomicMax(&s_box[3 + a], enc_hi(hi[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMax(&root->lo[threadIdx.x], s_box[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&root->hi[threadIdx.x - 3], s_box[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------- level set-up
// One workgroup: cut every active node into chunks of kBuildChunk primitives.
// `from_queue`: the level is the one the previous level's offsets_kernel has just queued (count in ctl->n_next), so a
// level costs one host round trip (reading back n_active / n_chunks for the launch sizes) instead of two.
__global__ __launch_bounds__(kBuildThreads) void setup_level_kernel(BuildArgs A, uint32_t from_queue) {
    __shared__ uint32_t s_scan[kBuildThreads / 64];
    const uint32_t n_active = from_queue ? A.ctl->n_next : A.n_active;
    __syncthreads(); // everyone has read n_next before thread 0 resets it below
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_active; base += kBuildThreads) {
        const uint32_t a = base + threadIdx.x;
        const uint32_t count = a < n_active ? A.nodes[A.active[a]].count : 0u;
        const uint32_t chunks = (count + kBuildChunk - 1) / kBuildChunk;
        uint32_t total;
        const uint32_t first = carry + block_scan_excl(chunks, s_scan, total);
        if (a < n_active) {
            A.chunk_first[a] = first;
            for (uint32_t k = 0; k < chunks && first + k < A.chunk_cap; ++k) {
                A.chunk_node[first + k] = a;
                A.chunk_off[first + k] = k * kBuildChunk;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        A.chunk_first[n_active] = carry;
        A.ctl->n_chunks = carry;
        A.ctl->n_active = n_active;
        if (carry > A.chunk_cap) A.ctl->overflow = 1;
        A.ctl->n_next = 0;
    }
}

// ---------------------------------------------------------------------------------------------- binning
// bvh.rs:86-102 for one chunk: every primitive grows the bin of its centre on each axis that is wide enough.
__global__ __launch_bounds__(kBuildThreads) void bin_kernel(BuildArgs A) {
    __shared__ uint32_t s_bins[3 * 7 * kWaveBins];
    const uint32_t c = blockIdx.x;
    const uint32_t a = A.chunk_node[c];
    const BuildNode nd = A.nodes[A.active[a]];
    const uint32_t off = A.chunk_off[c];
    const uint32_t cnt = min(kBuildChunk, nd.count - off);
    const uint32_t nb = A.n_bins;
    const bool local = nb <= kWaveBins;
    float lo_bound[3], scale[3];
    bool wide[3];
    for (int ax = 0; ax < 3; ++ax) {
        lo_bound[ax] = dec_lo(nd.lo[ax]);
        const float extent = dec_hi(nd.hi[ax]) - lo_bound[ax];
        wide[ax] = !(extent < 0.00001f); // bvh.rs:83-85 `continue`s when it is narrower
        scale[ax] = static_cast<float>(nb) / extent;
    }
    if (local) {
        for (uint32_t i = threadIdx.x; i < 3 * 7 * nb; i += kBuildThreads) s_bins[i] = 0;
        __syncthreads();
    }
    uint32_t *g_bins = A.bins + static_cast<size_t>(a) * 3 * 7 * nb;
    for (uint32_t i = threadIdx.x; i < cnt; i += kBuildThreads) {
        const uint32_t pos = nd.first + off + i;
        uint32_t elo[3], ehi[3];
        for (int k = 0; k < 3; ++k) {
            elo[k] = enc_lo(A.plo[k][pos]);
            ehi[k] = enc_hi(A.phi[k][pos]);
        }
        for (int ax = 0; ax < 3; ++ax) {
            if (!wide[ax]) continue;
            const uint32_t b = static_cast<uint32_t>(bin_of(A.key[ax][pos], lo_bound[ax], scale[ax], nb));
            uint32_t *bins = (local ? s_bins : g_bins) + static_cast<size_t>(ax) * 7 * nb;
            atomicAdd(&bins[b], 1u);
            for (int k = 0; k < 3; ++k) {
                atomicMax(&bins[(1 + k) * nb + b], elo[k]);
                atomicMax(&bins[(4 + k) * nb + b], ehi[k]);
            }
        }
    }
    if (local) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 3 * nb; i += kBuildThreads) { // (axis, bin)
            const uint32_t ax = i / nb, b = i % nb;
            const uint32_t *src = s_bins + ax * 7 * nb;
            if (src[b] == 0) continue;
            uint32_t *dst = g_bins + static_cast<size_t>(ax) * 7 * nb;
            atomicAdd(&dst[b], src[b]);
            for (int k = 1; k < 7; ++k) atomicMax(&dst[k * nb + b], src[k * nb + b]);
        }
    }
}

// ---------------------------------------------------------------------------------------------- SAH sweep
struct SweepBox {
    float lx, ly, lz, hx, hy, hz;
    uint32_t n;
    __device__ __forceinline__ void clear() {
        lx = ly = lz = __builtin_inff();
        hx = hy = hz = -__builtin_inff();
        n = 0;
    }
    __device__ __forceinline__ void grow(const SweepBox &o) {
        lx = zmin(lx, o.lx); ly = zmin(ly, o.ly); lz = zmin(lz, o.lz);
        hx = zmax(hx, o.hx); hy = zmax(hy, o.hy); hz = zmax(hz, o.hz);
        n += o.n;
    }
    __device__ __forceinline__ float area() const { return half_area(lx, ly, lz, hx, hy, hz); }
};

__device__ __forceinline__ SweepBox load_bin(const uint32_t *bins, uint32_t nb, uint32_t b) {
    SweepBox x;
    x.n = bins[b];
    x.lx = dec_lo(bins[1 * nb + b]); x.ly = dec_lo(bins[2 * nb + b]); x.lz = dec_lo(bins[3 * nb + b]);
    x.hx = dec_hi(bins[4 * nb + b]); x.hy = dec_hi(bins[5 * nb + b]); x.hz = dec_hi(bins[6 * nb + b]);
    return x;
}

// find_best_split_plane + the leaf test of subdivide (bvh.rs:73-139, 167-174): one workgroup per active node.
__global__ __launch_bounds__(kBuildThreads) void split_kernel(BuildArgs A) {
    __shared__ float s_box[2][6][kBuildThreads]; // [prefix | suffix] scans of the per-thread segment boxes
    __shared__ uint32_t s_cnt[2][kBuildThreads];
    __shared__ float s_left_area[kMaxBins];
    __shared__ uint32_t s_left_count[kMaxBins];
    __shared__ float s_cost[kBuildThreads];
    __shared__ uint32_t s_idx[kBuildThreads];
    const uint32_t a = blockIdx.x, t = threadIdx.x, nb = A.n_bins;
    const BuildNode nd = A.nodes[A.active[a]];
    const uint32_t seg = (nb + kBuildThreads - 1) / kBuildThreads;
    const uint32_t b0 = min(nb, t * seg), b1 = min(nb, b0 + seg);
    float best_cost = __builtin_inff(), best_plane = 0.0f;
    uint32_t best_axis = 0;
    float extent[3];
    for (int ax = 0; ax < 3; ++ax) extent[ax] = dec_hi(nd.hi[ax]) - dec_lo(nd.lo[ax]);
    // a single primitive never splits (see small_subtree_kernel): no sweep
    const bool lone = nd.count == 1u && isfinite(extent[0]) && isfinite(extent[1]) && isfinite(extent[2]);
    for (int ax = 0; ax < 3 && !lone; ++ax) {
        if (extent[ax] < 0.00001f) continue; // workgroup-uniform
        const uint32_t *bins = A.bins + (static_cast<size_t>(a) * 3 + ax) * 7 * nb;
        // per-thread segment, then inclusive prefix and suffix scans over the threads
        SweepBox mine;
        mine.clear();
        for (uint32_t b = b0; b < b1; ++b) mine.grow(load_bin(bins, nb, b));
        __syncthreads();
        for (int side = 0; side < 2; ++side) {
            s_box[side][0][t] = mine.lx; s_box[side][1][t] = mine.ly; s_box[side][2][t] = mine.lz;
            s_box[side][3][t] = mine.hx; s_box[side][4][t] = mine.hy; s_box[side][5][t] = mine.hz;
            s_cnt[side][t] = mine.n;
        }
        __syncthreads();
        for (uint32_t d = 1; d < kBuildThreads; d <<= 1) {
            SweepBox p, q;
            p.clear();
            q.clear();
            if (t >= d) {
                p.lx = s_box[0][0][t - d]; p.ly = s_box[0][1][t - d]; p.lz = s_box[0][2][t - d];
                p.hx = s_box[0][3][t - d]; p.hy = s_box[0][4][t - d]; p.hz = s_box[0][5][t - d];
                p.n = s_cnt[0][t - d];
            }
            if (t + d < kBuildThreads) {
                q.lx = s_box[1][0][t + d]; q.ly = s_box[1][1][t + d]; q.lz = s_box[1][2][t + d];
                q.hx = s_box[1][3][t + d]; q.hy = s_box[1][4][t + d]; q.hz = s_box[1][5][t + d];
                q.n = s_cnt[1][t + d];
            }
            __syncthreads();
            s_box[0][0][t] = zmin(s_box[0][0][t], p.lx); s_box[0][1][t] = zmin(s_box[0][1][t], p.ly);
            s_box[0][2][t] = zmin(s_box[0][2][t], p.lz); s_box[0][3][t] = zmax(s_box[0][3][t], p.hx);
            s_box[0][4][t] = zmax(s_box[0][4][t], p.hy); s_box[0][5][t] = zmax(s_box[0][5][t], p.hz);
            s_cnt[0][t] += p.n;
            s_box[1][0][t] = zmin(s_box[1][0][t], q.lx); s_box[1][1][t] = zmin(s_box[1][1][t], q.ly);
            s_box[1][2][t] = zmin(s_box[1][2][t], q.lz); s_box[1][3][t] = zmax(s_box[1][3][t], q.hx);
            s_box[1][4][t] = zmax(s_box[1][4][t], q.hy); s_box[1][5][t] = zmax(s_box[1][5][t], q.hz);
            s_cnt[1][t] += q.n;
            __syncthreads();
        }
        // forward walk: what lies left of plane i (bins 0..i), bvh.rs:107-113
        SweepBox run;
        run.clear();
        if (t > 0) {
            run.lx = s_box[0][0][t - 1]; run.ly = s_box[0][1][t - 1]; run.lz = s_box[0][2][t - 1];
            run.hx = s_box[0][3][t - 1]; run.hy = s_box[0][4][t - 1]; run.hz = s_box[0][5][t - 1];
            run.n = s_cnt[0][t - 1];
        }
        for (uint32_t b = b0; b < b1; ++b) {
            run.grow(load_bin(bins, nb, b));
            s_left_count[b] = run.n;
            s_left_area[b] = run.area();
        }
        // backward walk: what lies right of plane i (bins i+1..), bvh.rs:114-119, and the cost, bvh.rs:125-126
        run.clear();
        if (t + 1 < kBuildThreads) {
            run.lx = s_box[1][0][t + 1]; run.ly = s_box[1][1][t + 1]; run.lz = s_box[1][2][t + 1];
            run.hx = s_box[1][3][t + 1]; run.hy = s_box[1][4][t + 1]; run.hz = s_box[1][5][t + 1];
            run.n = s_cnt[1][t + 1];
        }
        for (uint32_t b = b1; b > b0; --b) {
            const uint32_t i = b - 1;
            if (i + 1 < nb)
                s_left_area[i] = static_cast<float>(s_left_count[i]) * s_left_area[i] + static_cast<float>(run.n) * run.area();
            run.grow(load_bin(bins, nb, i));
        }
        // first strict minimum over i = 0..nb-2 (bvh.rs:127-132); each thread only touched its own segment so far
        float my_cost = __builtin_inff();
        uint32_t my_idx = 0xffffffffu;
        for (uint32_t i = b0; i < b1 && i + 1 < nb; ++i) {
            const float cost = s_left_area[i];
            if (cost < my_cost) {
                my_cost = cost;
                my_idx = i;
            }
        }
        s_cost[t] = my_cost;
        s_idx[t] = my_idx;
        __syncthreads();
        for (uint32_t d = kBuildThreads / 2; d > 0; d >>= 1) {
            if (t < d) {
                const float oc = s_cost[t + d];
                const uint32_t oi = s_idx[t + d];
                if (oc < s_cost[t] || (oc == s_cost[t] && oi < s_idx[t])) {
                    s_cost[t] = oc;
                    s_idx[t] = oi;
                }
            }
            __syncthreads();
        }
        const float cost = s_cost[0];
        const uint32_t idx = s_idx[0];
        if (idx != 0xffffffffu && cost < best_cost) {
            best_cost = cost;
            best_axis = static_cast<uint32_t>(ax);
            const float step = 1.0f / static_cast<float>(nb);
            best_plane = dec_lo(nd.lo[ax]) + extent[ax] * step * (1.0f + static_cast<float>(idx)); // bvh.rs:130
        }
        __syncthreads();
    }
    if (t == 0) {
        const float leaf_cost = static_cast<float>(nd.count) * ((extent[0] * extent[1] + extent[1] * extent[2]) + extent[2] * extent[0]);
        Decision d;
        d.plane = best_plane;
        d.axis = best_axis;
        d.partition = (lone || leaf_cost <= best_cost) ? 0u : 1u; // bvh.rs:172-174
        d.n_left = 0;
        d.split = 0;
        d.child = 0;
        A.dec[a] = d;
    }
}

// ---------------------------------------------------------------------------------------------- partition
__global__ __launch_bounds__(kBuildThreads) void classify_kernel(BuildArgs A) {
    __shared__ uint32_t s_scan[kBuildThreads / 64];
    const uint32_t c = blockIdx.x;
    const uint32_t a = A.chunk_node[c];
    const Decision d = A.dec[a];
    if (!d.partition) return;
    const BuildNode nd = A.nodes[A.active[a]];
    const uint32_t off = A.chunk_off[c];
    const uint32_t cnt = min(kBuildChunk, nd.count - off);
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < cnt; i += kBuildThreads)
        mine += A.key[d.axis][nd.first + off + i] < d.plane ? 1u : 0u; // bvh.rs:179
    uint32_t total;
    block_scan_excl(mine, s_scan, total);
    if (threadIdx.x == 0) A.chunk_left[c] = total;
}

// Per node (one thread each): prefix of its chunks' left counts, the split decision, the children (bvh.rs:188-209).
__global__ __launch_bounds__(kBuildThreads) void offsets_kernel(BuildArgs A) {
    const uint32_t a = blockIdx.x * kBuildThreads + threadIdx.x;
    if (a >= A.n_active) return;
    Decision d = A.dec[a];
    if (!d.partition) return;
    const uint32_t self = A.active[a];
    uint32_t carry = 0;
    for (uint32_t c = A.chunk_first[a]; c < A.chunk_first[a + 1]; ++c) {
        A.chunk_left_before[c] = carry;
        carry += A.chunk_left[c];
    }
    const BuildNode nd = A.nodes[self];
    d.n_left = carry;
    if (carry != 0 && carry != nd.count) { // bvh.rs:188-190
        const uint32_t child = atomicAdd(&A.ctl->n_nodes, 2u);
        if (child + 2u <= A.node_cap) {
            d.split = 1;
            d.child = child;
            for (uint32_t side = 0; side < 2; ++side) {
                BuildNode ch{};
                ch.first = side == 0 ? nd.first : nd.first + carry;
                ch.count = side == 0 ? carry : nd.count - carry;
                if (A.use_wave && ch.count <= kSmallPrims) {
                    ch.kind = 2;
                    ch.sub = atomicAdd(&A.ctl->n_small, 1u);
                    A.small_roots[ch.sub] = child + side; // n_small <= n: cannot overflow
                } else {
                    A.next_active[atomicAdd(&A.ctl->n_next, 1u)] = child + side; // <= n entries
                }
                A.nodes[child + side] = ch;
            }
            A.nodes[self].child = child;
            A.nodes[self].kind = 1;
        } else {
            A.ctl->overflow = 1;
        }
    }
    A.dec[a] = d;
}

// Walks one chunk in position order and hands every element (position in node, class, lefts before it) to f.
template <typename F> __device__ __forceinline__ void for_each_classified(const BuildArgs &A, uint32_t c, const BuildNode &nd,
                                                                          const Decision &d, uint32_t *s_scan, F f) {
    const uint32_t off = A.chunk_off[c];
    const uint32_t cnt = min(kBuildChunk, nd.count - off);
    uint32_t carry = A.chunk_left_before[c];
    for (uint32_t base = 0; base < cnt; base += kBuildThreads) { // workgroup-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < cnt;
        const bool is_left = valid && A.key[d.axis][nd.first + off + i] < d.plane;
        uint32_t total;
        const uint32_t before = carry + block_scan_excl(is_left ? 1u : 0u, s_scan, total);
        if (valid) f(off + i, is_left, before);
        carry += total;
    }
}

__global__ __launch_bounds__(kBuildThreads) void rank_kernel(BuildArgs A) {
    __shared__ uint32_t s_scan[kBuildThreads / 64];
    const uint32_t c = blockIdx.x;
    const uint32_t a = A.chunk_node[c];
    const Decision d = A.dec[a];
    if (!d.partition) return;
    const BuildNode nd = A.nodes[A.active[a]];
    uint32_t *head_r = A.head_r + nd.first, *tail_l = A.tail_l + nd.first;
    for_each_classified(A, c, nd, d, s_scan, [&](uint32_t s, bool is_left, uint32_t left_before) {
        if (is_left) {
            if (s >= d.n_left) tail_l[d.n_left - left_before] = s;
        } else if (s <= d.n_left) {
            head_r[1u + (s - left_before)] = s;
        }
    });
}

__global__ __launch_bounds__(kBuildThreads) void scatter_kernel(BuildArgs A) {
    __shared__ uint32_t s_scan[kBuildThreads / 64];
    __shared__ uint32_t s_box[2][6];
    const uint32_t c = blockIdx.x;
    const uint32_t a = A.chunk_node[c];
    const Decision d = A.dec[a];
    if (!d.partition) return;
    const BuildNode nd = A.nodes[A.active[a]];
    if (threadIdx.x < 12) s_box[threadIdx.x / 6][threadIdx.x % 6] = 0;
    __syncthreads();
    const uint32_t *head_r = A.head_r + nd.first, *tail_l = A.tail_l + nd.first;
    for_each_classified(A, c, nd, d, s_scan, [&](uint32_t s, bool is_left, uint32_t left_before) {
        const uint32_t dest = nd.first + partition_dest(is_left, s, left_before, d.n_left, nd.count, head_r, tail_l);
        const uint32_t src = nd.first + s;
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            lo[k] = A.plo[k][src];
            hi[k] = A.phi[k][src];
            A.t_plo[k][dest] = lo[k];
            A.t_phi[k][dest] = hi[k];
            A.t_key[k][dest] = A.key[k][src];
        }
        A.tmp[dest] = A.ids[src];
        if (d.split) { // update_node_bounds of the child it lands in, bvh.rs:196,202
            const uint32_t side = dest - nd.first < d.n_left ? 0u : 1u;
            for (int k = 0; k < 3; ++k) {
                atomicMax(&s_box[side][k], enc_lo(lo[k]));
                atomicMax(&s_box[side][3 + k], enc_hi(hi[k]));
            }
        }
    });
    __syncthreads();
    if (d.split && threadIdx.x < 12) {
        const uint32_t side = threadIdx.x / 6, k = threadIdx.x % 6;
        BuildNode &ch = A.nodes[d.child + side];
        atomicMax(k < 3 ? &ch.lo[k] : &ch.hi[k - 3], s_box[side][k]);
    }
}

__global__ __launch_bounds__(kBuildThreads) void copy_back_kernel(BuildArgs A) {
    const uint32_t c = blockIdx.x;
    const uint32_t a = A.chunk_node[c];
    if (!A.dec[a].partition) return;
    const BuildNode nd = A.nodes[A.active[a]];
    const uint32_t off = A.chunk_off[c];
    const uint32_t cnt = min(kBuildChunk, nd.count - off);
    for (uint32_t i = threadIdx.x; i < cnt; i += kBuildThreads) {
        const uint32_t pos = nd.first + off + i;
        A.ids[pos] = A.tmp[pos];
        for (int k = 0; k < 3; ++k) {
            A.key[k][pos] = A.t_key[k][pos];
            A.plo[k][pos] = A.t_plo[k][pos];
            A.phi[k][pos] = A.t_phi[k][pos];
        }
    }
}

// ---------------------------------------------------------------------------------------------- small subtrees
// One wave = one workgroup = one subtree of <= 64 primitives: bvh.rs's recursion (subdivide, bvh.rs:166-210) as an
// explicit depth-first loop, lanes = primitives while binning / partitioning, lanes = bins while sweeping.
// Nodes come out in the reference's own numbering relative to the subtree (pair j at records 2j, 2j+1).
struct WaveBox {
    float v[6]; // lo xyz, hi xyz
};

__device__ __forceinline__ SweepBox shfl_box_up(const SweepBox &b, uint32_t d) {
    SweepBox o;
    o.lx = __shfl_up(b.lx, d); o.ly = __shfl_up(b.ly, d); o.lz = __shfl_up(b.lz, d);
    o.hx = __shfl_up(b.hx, d); o.hy = __shfl_up(b.hy, d); o.hz = __shfl_up(b.hz, d);
    o.n = __shfl_up(b.n, d);
    return o;
}
__device__ __forceinline__ SweepBox shfl_box_down(const SweepBox &b, uint32_t d) {
    SweepBox o;
    o.lx = __shfl_down(b.lx, d); o.ly = __shfl_down(b.ly, d); o.lz = __shfl_down(b.lz, d);
    o.hx = __shfl_down(b.hx, d); o.hy = __shfl_down(b.hy, d); o.hz = __shfl_down(b.hz, d);
    o.n = __shfl_down(b.n, d);
    return o;
}

// The sweep of find_best_split_plane (bvh.rs:104-133) by one wave. Lane = (axis slot `group`, bin `bin`) holds that
// bin (`mine`; empty for bins >= nb and unused slots); `seg` (a power of two) bins per slot, `in_pass` slots in use.
// Returns the first strict minimum below `best_cost` in (slot, plane) order -- the order the reference breaks cost
// ties in -- as (cost, lane), or lane 0xffffffff if no plane beats `best_cost`.
__device__ __forceinline__ void wave_sweep(const SweepBox &mine, uint32_t bin, uint32_t group, uint32_t seg, uint32_t nb, uint32_t in_pass,
                                           float best_cost, float &c, uint32_t &ci) {
    const uint32_t lane = lane_id();
    SweepBox left = mine, right = mine; // inclusive prefix / suffix inside the segment
    for (uint32_t d = 1; d < seg; d <<= 1) {
        const SweepBox pl = shfl_box_up(left, d), pr = shfl_box_down(right, d);
        if (bin >= d) left.grow(pl);
        if (bin + d < seg) right.grow(pr);
    }
    const SweepBox beyond = shfl_box_down(right, 1); // bins bin+1 .. of the same axis: right of plane `bin`
    const bool plane_here = bin + 1u < nb && group < in_pass;
    float cost = __builtin_inff();
    if (plane_here) cost = static_cast<float>(left.n) * left.area() + static_cast<float>(beyond.n) * beyond.area();
    // NaN and +inf never win (bvh.rs:127)
    const bool cand = plane_here && (cost < best_cost);
    c = cand ? cost : __builtin_inff();
    ci = cand ? lane : 0xffffffffu;
#pragma unroll
    for (uint32_t d = 32; d > 0; d >>= 1) {
        const float oc = __shfl_xor(c, d);
        const uint32_t oi = __shfl_xor(ci, d);
        if (oc < c || (oc == c && oi < ci)) {
            c = oc;
            ci = oi;
        }
    }
}

// split_kernel for at most 64 bins: one WAVE per active node (four nodes per workgroup), bins straight from global memory
// into lanes, no LDS, no barriers.
__global__ __launch_bounds__(kBuildThreads) void split_wave_kernel(BuildArgs A) {
    const uint32_t a = blockIdx.x * (kBuildThreads / 64) + (threadIdx.x >> 6);
    if (a >= A.n_active) return; // wave-uniform
    const uint32_t lane = lane_id(), nb = A.n_bins;
    const BuildNode nd = A.nodes[A.active[a]];
    const float lo[3] = {dec_lo(nd.lo[0]), dec_lo(nd.lo[1]), dec_lo(nd.lo[2])};
    const float ex = dec_hi(nd.hi[0]) - lo[0], ey = dec_hi(nd.hi[1]) - lo[1], ez = dec_hi(nd.hi[2]) - lo[2];
    uint32_t seg = 2;
    while (seg < nb) seg <<= 1;
    const uint32_t per_pass = min(3u, 64u / seg);
    const uint32_t bin = lane & (seg - 1u), group = lane / seg;
    uint32_t axes = 0, n_axes = 0; // wide axes, two bits each, in increasing order
    if (!(ex < 0.00001f)) { axes |= 0u << (2u * n_axes); n_axes += 1; }
    if (!(ey < 0.00001f)) { axes |= 1u << (2u * n_axes); n_axes += 1; }
    if (!(ez < 0.00001f)) { axes |= 2u << (2u * n_axes); n_axes += 1; }
    const bool lone = nd.count == 1u && isfinite(ex) && isfinite(ey) && isfinite(ez); // see small_subtree_kernel
    float best_cost = __builtin_inff(), best_plane = 0.0f;
    uint32_t best_axis = 0;
    for (uint32_t g0 = 0; g0 < n_axes && !lone; g0 += per_pass) {
        const uint32_t in_pass = min(per_pass, n_axes - g0);
        SweepBox mine;
        mine.clear();
        if (group < in_pass && bin < nb) {
            const uint32_t ax = (axes >> (2u * (g0 + group))) & 3u;
            mine = load_bin(A.bins + (static_cast<size_t>(a) * 3 + ax) * 7 * nb, nb, bin);
        }
        float c;
        uint32_t ci;
        wave_sweep(mine, bin, group, seg, nb, in_pass, best_cost, c, ci);
        if (ci != 0xffffffffu) {
            const uint32_t ax = (axes >> (2u * (g0 + ci / seg))) & 3u;
            const float extent = ax == 0 ? ex : (ax == 1 ? ey : ez);
            best_cost = c;
            best_axis = ax;
            const float step = 1.0f / static_cast<float>(nb);
            best_plane = (ax == 0 ? lo[0] : (ax == 1 ? lo[1] : lo[2])) + extent * step * (1.0f + static_cast<float>(ci & (seg - 1u))); // bvh.rs:130
        }
    }
    if (lane == 0) {
        const float leaf_cost = static_cast<float>(nd.count) * ((ex * ey + ey * ez) + ez * ex);
        Decision d;
        d.plane = best_plane;
        d.axis = best_axis;
        d.partition = (lone || leaf_cost <= best_cost) ? 0u : 1u; // bvh.rs:172-174
        d.n_left = 0;
        d.split = 0;
        d.child = 0;
        A.dec[a] = d;
    }
}

__global__ __launch_bounds__(64) void small_subtree_kernel(BuildArgs A, uint32_t n_small) {
    __shared__ float s_key[3][64], s_lo[3][64], s_hi[3][64];
    __shared__ uint32_t s_id[64], s_perm[64];
    __shared__ uint32_t s_bins[7][kWaveBins];
    __shared__ uint32_t s_head_r[66], s_tail_l[66];
    __shared__ uint32_t s_child[2][6];
    __shared__ uint32_t s_stack_first[64], s_stack_count[64], s_stack_self[64];
    __shared__ float s_stack_box[6][64];
    const uint32_t s = blockIdx.x, lane = threadIdx.x, nb = A.n_bins;
    if (s >= n_small) return;
    const uint32_t root_id = A.small_roots[s];
    const BuildNode root = A.nodes[root_id];
    const uint32_t n = root.count; // <= 64
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&A.ctl->sub_alloc, 2u * n);
    base = __shfl(base, 0);
    if (base + 2u * n > A.sub_cap) { // cannot happen (sub_cap = 2 * primitives); never write out of bounds
        if (lane == 0) {
            A.ctl->overflow = 1;
            A.sub_base[s] = 0;
            A.sub_pairs[s] = 0;
        }
        return;
    }
    if (lane < n) {
        const uint32_t pos = root.first + lane;
        s_id[lane] = A.ids[pos];
        for (int k = 0; k < 3; ++k) {
            s_key[k][lane] = A.key[k][pos];
            s_lo[k][lane] = A.plo[k][pos];
            s_hi[k][lane] = A.phi[k][pos];
        }
    }
    s_perm[lane] = lane;
    __syncthreads();

    // current node (wave-uniform)
    uint32_t first = 0, count = n, self = 0xffffffffu; // self: record index inside the subtree, or "the root"
    float box[6];
    for (int k = 0; k < 3; ++k) {
        box[k] = dec_lo(root.lo[k]);
        box[3 + k] = dec_hi(root.hi[k]);
    }
    uint32_t sp = 0, n_pairs = 0;
    uint32_t seg = 2; // bins per axis rounded up to a power of two (<= 64)
    while (seg < nb) seg <<= 1;
    const uint32_t per_pass = min(3u, 64u / seg);
    for (;;) {
        // ---- find_best_split_plane, bvh.rs:73-139. The 64 lanes hold `per_pass` axes of `seg` bins each (seg = bin
        // count rounded up to a power of two), so 32 bins take two passes instead of three; scans never cross a
        // segment, and lane order = (axis, plane) order, which is the order the reference breaks cost ties in.
        float best_cost = __builtin_inff(), best_plane = 0.0f;
        uint32_t best_axis = 0;
        const float ex = box[3] - box[0], ey = box[4] - box[1], ez = box[5] - box[2];
        const uint32_t slot = lane < count ? s_perm[first + lane] : 0u;
        // A single primitive never splits: every plane costs exactly 1 * area(its box), which is also the leaf
        // cost, and bvh.rs:172 keeps the leaf on `<=` (finite extents keep NaN out of that comparison). Half of
        // all nodes are such leaves, so they skip the sweep.
        const bool lone = count == 1u && isfinite(ex) && isfinite(ey) && isfinite(ez);
        uint32_t axes = 0, n_axes = 0; // wide axes, two bits each, in increasing order (bvh.rs:83-85 skips the narrow ones)
        if (!(ex < 0.00001f)) { axes |= 0u << (2u * n_axes); n_axes += 1; }
        if (!(ey < 0.00001f)) { axes |= 1u << (2u * n_axes); n_axes += 1; }
        if (!(ez < 0.00001f)) { axes |= 2u << (2u * n_axes); n_axes += 1; }
        const uint32_t bin = lane & (seg - 1u), group = lane / seg;
        const bool direct = count <= kDirectPrims && !lone;
        if (direct) {
            // Few primitives: the cost only changes at planes where a primitive's bin ends, and the reference takes
            // the first plane of each run of equal costs, so the candidates are plane 0 and the primitives' own bins:
            // (count + 1) lanes per axis evaluate them straight from the primitives' boxes. Left of plane i = the
            // primitives whose bin is <= i, exactly the bins the sweep would have merged; same cost expression.
            if (lane < count) {
                for (uint32_t j = 0; j < n_axes; ++j) {
                    const uint32_t ax = (axes >> (2u * j)) & 3u;
                    const float extent = ax == 0 ? ex : (ax == 1 ? ey : ez);
                    const float lo_bound = ax == 0 ? box[0] : (ax == 1 ? box[1] : box[2]);
                    s_bins[j][lane] = static_cast<uint32_t>(bin_of(s_key[ax][slot], lo_bound, static_cast<float>(nb) / extent, nb));
                }
            }
            __syncthreads();
            const uint32_t per_axis = count + 1u;
            const uint32_t j = lane / per_axis, k = lane % per_axis; // axis slot, candidate
            const bool in_range = j < n_axes;
            const uint32_t plane = (in_range && k > 0u) ? s_bins[j][k - 1u] : 0u;
            SweepBox left, right;
            left.clear();
            right.clear();
            for (uint32_t p = 0; p < count; ++p) { // wave-uniform trip count
                const uint32_t ps = s_perm[first + p];
                SweepBox one;
                one.lx = s_lo[0][ps]; one.ly = s_lo[1][ps]; one.lz = s_lo[2][ps];
                one.hx = s_hi[0][ps]; one.hy = s_hi[1][ps]; one.hz = s_hi[2][ps];
                one.n = 1;
                const bool on_left = in_range && s_bins[j][p] <= plane;
                SweepBox none;
                none.clear();
                left.grow(on_left ? one : none);
                right.grow(on_left ? none : one);
            }
            const float cost = static_cast<float>(left.n) * left.area() + static_cast<float>(right.n) * right.area();
            const bool cand = in_range && plane + 1u < nb && cost < best_cost; // NaN and +inf never win (bvh.rs:127)
            float c = cand ? cost : __builtin_inff();
            uint32_t ci = cand ? j * 64u + plane : 0xffffffffu; // (axis, plane) order; nb <= 64 here
#pragma unroll
            for (uint32_t d = 32; d > 0; d >>= 1) {
                const float oc = __shfl_xor(c, d);
                const uint32_t oi = __shfl_xor(ci, d);
                if (oc < c || (oc == c && oi < ci)) {
                    c = oc;
                    ci = oi;
                }
            }
            if (ci != 0xffffffffu) {
                const uint32_t ax = (axes >> (2u * (ci / 64u))) & 3u;
                const float extent = ax == 0 ? ex : (ax == 1 ? ey : ez);
                const float lo_bound = ax == 0 ? box[0] : (ax == 1 ? box[1] : box[2]);
                best_cost = c;
                best_axis = ax;
                const float step = 1.0f / static_cast<float>(nb);
                best_plane = lo_bound + extent * step * (1.0f + static_cast<float>(ci & 63u)); // bvh.rs:130
            }
            __syncthreads();
        }
        for (uint32_t g0 = 0; g0 < n_axes && !lone && !direct; g0 += per_pass) { // wave-uniform
            const uint32_t in_pass = min(per_pass, n_axes - g0);
            for (int k = 0; k < 7; ++k) s_bins[k][lane] = 0;
            __syncthreads();
            if (lane < count) {
                for (uint32_t j = 0; j < in_pass; ++j) {
                    const uint32_t ax = (axes >> (2u * (g0 + j))) & 3u;
                    const float extent = ax == 0 ? ex : (ax == 1 ? ey : ez);
                    const float lo_bound = ax == 0 ? box[0] : (ax == 1 ? box[1] : box[2]);
                    const float scale = static_cast<float>(nb) / extent;
                    const uint32_t b = j * seg + static_cast<uint32_t>(bin_of(s_key[ax][slot], lo_bound, scale, nb));
                    atomicAdd(&s_bins[0][b], 1u);
                    for (int k = 0; k < 3; ++k) {
                        atomicMax(&s_bins[1 + k][b], enc_lo(s_lo[k][slot]));
                        atomicMax(&s_bins[4 + k][b], enc_hi(s_hi[k][slot]));
                    }
                }
            }
            __syncthreads();
            SweepBox mine; // lane = (axis slot, bin); bins >= nb and unused slots are empty
            mine.n = s_bins[0][lane];
            mine.lx = dec_lo(s_bins[1][lane]); mine.ly = dec_lo(s_bins[2][lane]); mine.lz = dec_lo(s_bins[3][lane]);
            mine.hx = dec_hi(s_bins[4][lane]); mine.hy = dec_hi(s_bins[5][lane]); mine.hz = dec_hi(s_bins[6][lane]);
            float c;
            uint32_t ci;
            wave_sweep(mine, bin, group, seg, nb, in_pass, best_cost, c, ci);
            if (ci != 0xffffffffu) {
                const uint32_t ax = (axes >> (2u * (g0 + ci / seg))) & 3u;
                const float extent = ax == 0 ? ex : (ax == 1 ? ey : ez);
                const float lo_bound = ax == 0 ? box[0] : (ax == 1 ? box[1] : box[2]);
                best_cost = c;
                best_axis = ax;
                const float step = 1.0f / static_cast<float>(nb);
                best_plane = lo_bound + extent * step * (1.0f + static_cast<float>(ci & (seg - 1u))); // bvh.rs:130
            }
            __syncthreads();
        }
        // ---- subdivide, bvh.rs:166-210
        const float leaf_cost = static_cast<float>(count) * ((ex * ey + ey * ez) + ez * ex);
        bool split = false;
        uint32_t n_left = 0;
        if (!lone && !(leaf_cost <= best_cost)) {
            const bool valid = lane < count;
            const bool is_left = valid && s_key[best_axis][slot] < best_plane;
            const unsigned long long left_mask = __ballot(is_left);
            n_left = static_cast<uint32_t>(__popcll(left_mask));
            const uint32_t left_before = static_cast<uint32_t>(__popcll(left_mask & ((1ull << lane) - 1ull)));
            if (valid) {
                if (is_left) {
                    if (lane >= n_left) s_tail_l[n_left - left_before] = lane;
                } else if (lane <= n_left) {
                    s_head_r[1u + (lane - left_before)] = lane;
                }
            }
            __syncthreads();
            uint32_t dest = 0;
            if (valid) dest = partition_dest(is_left, lane, left_before, n_left, count, s_head_r, s_tail_l);
            split = n_left != 0 && n_left != count;
            if (lane < 12) s_child[lane / 6][lane % 6] = 0;
            __syncthreads();
            if (valid) {
                s_perm[first + dest] = slot;
                if (split) {
                    const uint32_t side = dest < n_left ? 0u : 1u;
                    for (int k = 0; k < 3; ++k) {
                        atomicMax(&s_child[side][k], enc_lo(s_lo[k][slot]));
                        atomicMax(&s_child[side][3 + k], enc_hi(s_hi[k][slot]));
                    }
                }
            }
            __syncthreads();
        }
        if (split) {
            const uint32_t pair = n_pairs++;
            float lbox[6], rbox[6];
            for (int k = 0; k < 3; ++k) {
                lbox[k] = dec_lo(s_child[0][k]); lbox[3 + k] = dec_hi(s_child[0][3 + k]);
                rbox[k] = dec_lo(s_child[1][k]); rbox[3 + k] = dec_hi(s_child[1][3 + k]);
            }
            if (lane == 0) {
                wfpt_bvh_node l{}, r{};
                for (int k = 0; k < 3; ++k) {
                    l.aabb_min[k] = lbox[k]; l.aabb_max[k] = lbox[3 + k];
                    r.aabb_min[k] = rbox[k]; r.aabb_max[k] = rbox[3 + k];
                }
                l.left_first = root.first + first;          l.prim_count = n_left;
                r.left_first = root.first + first + n_left; r.prim_count = count - n_left;
                A.sub_nodes[base + 2u * pair] = l;
                A.sub_nodes[base + 2u * pair + 1u] = r;
                if (self != 0xffffffffu) { // this node becomes an inner node pointing at the new pair (relative)
                    A.sub_nodes[base + self].left_first = 2u * pair;
                    A.sub_nodes[base + self].prim_count = 0;
                }
                // the right child waits on the stack, the left one is next (bvh.rs:208-209)
                s_stack_first[sp] = first + n_left;
                s_stack_count[sp] = count - n_left;
                s_stack_self[sp] = 2u * pair + 1u;
                for (int k = 0; k < 6; ++k) s_stack_box[k][sp] = rbox[k];
            }
            sp += 1;
            count = n_left;
            self = 2u * pair;
            for (int k = 0; k < 6; ++k) box[k] = lbox[k];
            __syncthreads();
            continue;
        }
        if (sp == 0) break;
        sp -= 1;
        first = s_stack_first[sp];
        count = s_stack_count[sp];
        self = s_stack_self[sp];
        for (int k = 0; k < 6; ++k) box[k] = s_stack_box[k][sp];
        __syncthreads();
    }
    if (lane < n) A.ids[root.first + lane] = s_id[s_perm[lane]];
    if (lane == 0) {
        A.sub_base[s] = base;
        A.sub_pairs[s] = n_pairs;
    }
}

// ---------------------------------------------------------------------------------------------- final numbering
__global__ __launch_bounds__(kBuildThreads) void emit_nodes_kernel(const BuildNode *nodes, uint32_t n_prov, const uint32_t *final_index,
                                                                  const uint32_t *pair_index, wfpt_bvh_node *out) {
    const uint32_t x = blockIdx.x * kBuildThreads + threadIdx.x;
    if (x >= n_prov) return;
    const BuildNode nd = nodes[x];
    wfpt_bvh_node o{};
    for (int k = 0; k < 3; ++k) {
        o.aabb_min[k] = dec_lo(nd.lo[k]);
        o.aabb_max[k] = dec_hi(nd.hi[k]);
    }
    const uint32_t pair = pair_index[x]; // 0 for a leaf
    o.left_first = pair ? pair : nd.first;
    o.prim_count = pair ? 0u : nd.count;
    out[final_index[x]] = o;
    if (x == 0) out[1] = wfpt_bvh_node{}; // bvh.rs:160-161: the placeholder that keeps siblings at (2k, 2k+1)
}

__global__ __launch_bounds__(64) void emit_subtrees_kernel(const uint32_t *small_roots, const uint32_t *sub_base, const uint32_t *sub_pairs,
                                                           const uint32_t *pair_index, const wfpt_bvh_node *sub_nodes,
                                                           wfpt_bvh_node *out) {
    const uint32_t s = blockIdx.x;
    const uint32_t dst = pair_index[small_roots[s]];
    const uint32_t n = 2u * sub_pairs[s], base = sub_base[s];
    for (uint32_t j = threadIdx.x; j < n; j += 64) {
        wfpt_bvh_node nd = sub_nodes[base + j];
        if (nd.prim_count == 0) nd.left_first += dst;
        out[dst + j] = nd;
    }
}

template <typename T> __global__ __launch_bounds__(kBuildThreads) void gather_prims_kernel(const T *in, const uint32_t *ids, uint32_t n, T *out) {
    const uint32_t i = blockIdx.x * kBuildThreads + threadIdx.x;
    if (i < n) out[i] = in[ids[i]];
}

// ---------------------------------------------------------------------------------------------- host driver
struct DeviceBuffers {
    std::vector<void *> owned;
    ~DeviceBuffers() {
        for (void *p : owned) (void)hipFree(p);
    }
    template <typename T> hipError_t alloc(T **p, size_t n) {
        void *raw = nullptr;
        const hipError_t e = hipMalloc(&raw, sizeof(T) * (n ? n : 1));
        if (e == hipSuccess) owned.push_back(raw);
        *p = static_cast<T *>(raw);
        return e;
    }
};

int fail_build(int status, const std::string &msg) {
    set_last_error(msg);
    return status;
}

#define BUILD_HIP(call)                                                                                     \
    do {                                                                                                    \
        const hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                               \
            return fail_build(e_ == hipErrorOutOfMemory ? WFPT_ERR_OUT_OF_MEMORY : WFPT_ERR_HIP,            \
                              std::string("device BVH build: ") + #call + ": " + hipGetErrorString(e_));   \
    } while (0)

template <int PRIM, typename Prim>
int build_on_device(Prim *prims, uint32_t n, wfpt_bvh_node *out_nodes, uint32_t cap, uint32_t *n_nodes, uint32_t n_bins, int device,
                    float *device_ms) {
    if (!prims || !out_nodes || !n_nodes || n == 0) return fail_build(WFPT_ERR_INVALID_ARGUMENT, "device BVH build: null or empty argument");
    if (cap < 2u * n || n > (1u << 30)) return fail_build(WFPT_ERR_INVALID_ARGUMENT, "device BVH build: node capacity must be >= 2 n"); // bvh.rs:148-150
    if (n_bins < 2) n_bins = 2;
    if (n_bins > kMaxBins) return fail_build(WFPT_ERR_UNSUPPORTED, "device BVH build: at most 4096 bins");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev)
        return fail_build(WFPT_ERR_NO_DEVICE, "device BVH build: no HIP device (there is no CPU fallback; use wfpt_build_bvh)");
    BUILD_HIP(hipSetDevice(device));

    const bool use_wave = n_bins <= kWaveBins;
    // A level's active nodes: with the wave path every active node has > 64 primitives.
    const uint32_t max_active = use_wave ? n / (kSmallPrims + 1u) + 1u : n;
    const size_t bin_words = static_cast<size_t>(max_active) * 3 * 7 * n_bins;
    if (bin_words * 4 > (static_cast<size_t>(16) << 30))
        return fail_build(WFPT_ERR_UNSUPPORTED, "device BVH build: this many bins for this many primitives needs > 16 GiB of bin storage");
    const uint32_t node_cap = 2u * n + 2u;
    const uint32_t chunk_cap = n / kBuildChunk + max_active + 1u;

    DeviceBuffers mem;
    Prim *d_prims = nullptr, *d_sorted = nullptr;
    float *soa = nullptr;
    BuildArgs A{};
    uint32_t *active[2] = {nullptr, nullptr};
    uint32_t *final_index = nullptr, *pair_index = nullptr;
    wfpt_bvh_node *d_out = nullptr;
    BUILD_HIP(mem.alloc(&d_prims, n));
    BUILD_HIP(mem.alloc(&d_sorted, n));
    BUILD_HIP(mem.alloc(&soa, 18 * static_cast<size_t>(n)));
    BUILD_HIP(mem.alloc(&A.ids, n));
    BUILD_HIP(mem.alloc(&A.tmp, n));
    BUILD_HIP(mem.alloc(&A.nodes, node_cap));
    BUILD_HIP(mem.alloc(&A.ctl, 1));
    BUILD_HIP(mem.alloc(&active[0], max_active + 1u));
    BUILD_HIP(mem.alloc(&active[1], max_active + 1u));
    BUILD_HIP(mem.alloc(&A.small_roots, n));
    BUILD_HIP(mem.alloc(&A.chunk_first, max_active + 2u));
    BUILD_HIP(mem.alloc(&A.chunk_node, chunk_cap));
    BUILD_HIP(mem.alloc(&A.chunk_off, chunk_cap));
    BUILD_HIP(mem.alloc(&A.bins, bin_words));
    BUILD_HIP(mem.alloc(&A.dec, max_active + 1u));
    BUILD_HIP(mem.alloc(&A.chunk_left, chunk_cap));
    BUILD_HIP(mem.alloc(&A.chunk_left_before, chunk_cap));
    BUILD_HIP(mem.alloc(&A.head_r, static_cast<size_t>(n) + 2));
    BUILD_HIP(mem.alloc(&A.tail_l, static_cast<size_t>(n) + 2));
    A.sub_cap = use_wave ? 2u * n : 0u;
    BUILD_HIP(mem.alloc(&A.sub_nodes, A.sub_cap));
    BUILD_HIP(mem.alloc(&A.sub_base, n));
    BUILD_HIP(mem.alloc(&A.sub_pairs, n));
    BUILD_HIP(mem.alloc(&d_out, cap));
    for (int k = 0; k < 3; ++k) {
        A.key[k] = soa + static_cast<size_t>(k) * n;
        A.plo[k] = soa + static_cast<size_t>(3 + k) * n;
        A.phi[k] = soa + static_cast<size_t>(6 + k) * n;
        A.t_key[k] = soa + static_cast<size_t>(9 + k) * n;
        A.t_plo[k] = soa + static_cast<size_t>(12 + k) * n;
        A.t_phi[k] = soa + static_cast<size_t>(15 + k) * n;
    }
    A.n = n;
    A.n_bins = n_bins;
    A.use_wave = use_wave ? 1u : 0u;
    A.node_cap = node_cap;
    A.chunk_cap = chunk_cap;

    hipStream_t stream = nullptr; // the default stream: this entry point is synchronous
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    BUILD_HIP(hipEventCreate(&ev0));
    BUILD_HIP(hipEventCreate(&ev1));
    struct EventGuard {
        hipEvent_t a, b;
        ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    } guard{ev0, ev1};

    BUILD_HIP(hipMemcpy(d_prims, prims, sizeof(Prim) * n, hipMemcpyHostToDevice));
    BUILD_HIP(hipEventRecord(ev0, stream));
    // root (bvh.rs:152-158)
    BuildNode root{};
    root.first = 0;
    root.count = n;
    const bool root_small = use_wave && n <= kSmallPrims;
    root.kind = root_small ? 2u : 0u;
    BuildCtl ctl{};
    ctl.n_nodes = 1;
    ctl.n_small = root_small ? 1u : 0u;
    BUILD_HIP(hipMemcpyAsync(A.nodes, &root, sizeof root, hipMemcpyHostToDevice, stream));
    BUILD_HIP(hipMemcpyAsync(A.ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice, stream));
    const uint32_t zero = 0;
    BUILD_HIP(hipMemcpyAsync(active[0], &zero, sizeof zero, hipMemcpyHostToDevice, stream));
    BUILD_HIP(hipMemcpyAsync(A.small_roots, &zero, sizeof zero, hipMemcpyHostToDevice, stream));
    const uint32_t prim_blocks = (n + kBuildThreads - 1) / kBuildThreads;
    hipLaunchKernelGGL(prep_prims_kernel<PRIM>, dim3(prim_blocks), dim3(kBuildThreads), 0, stream, d_prims, n, soa, soa + n, soa + 2 * static_cast<size_t>(n),
                       soa + 3 * static_cast<size_t>(n), soa + 4 * static_cast<size_t>(n), soa + 5 * static_cast<size_t>(n),
                       soa + 6 * static_cast<size_t>(n), soa + 7 * static_cast<size_t>(n), soa + 8 * static_cast<size_t>(n), A.ids, A.nodes);
    BUILD_HIP(hipGetLastError());

    // ---- large nodes, level by level; one host round trip per level (the launch sizes of the next one)
    uint32_t n_active = root_small ? 0u : 1u, n_chunks = 0;
    int cur = 0;
    uint32_t levels = 0;
    if (n_active) {
        A.active = active[0];
        A.n_active = 1;
        hipLaunchKernelGGL(setup_level_kernel, dim3(1), dim3(kBuildThreads), 0, stream, A, 0u);
        n_chunks = (n + kBuildChunk - 1) / kBuildChunk; // the root's chunks
    }
    while (n_active > 0) {
        if (++levels > 4096) return fail_build(WFPT_ERR_UNSUPPORTED, "device BVH build: tree deeper than 4096 levels");
        A.active = active[cur];
        A.next_active = active[cur ^ 1];
        A.n_active = n_active;
        BUILD_HIP(hipMemsetAsync(A.bins, 0, sizeof(uint32_t) * static_cast<size_t>(n_active) * 3 * 7 * n_bins, stream));
        hipLaunchKernelGGL(bin_kernel, dim3(n_chunks), dim3(kBuildThreads), 0, stream, A);
        if (use_wave) hipLaunchKernelGGL(split_wave_kernel, dim3((n_active + 3u) / 4u), dim3(kBuildThreads), 0, stream, A);
        else hipLaunchKernelGGL(split_kernel, dim3(n_active), dim3(kBuildThreads), 0, stream, A);
        hipLaunchKernelGGL(classify_kernel, dim3(n_chunks), dim3(kBuildThreads), 0, stream, A);
        hipLaunchKernelGGL(offsets_kernel, dim3((n_active + kBuildThreads - 1) / kBuildThreads), dim3(kBuildThreads), 0, stream, A);
        hipLaunchKernelGGL(rank_kernel, dim3(n_chunks), dim3(kBuildThreads), 0, stream, A);
        hipLaunchKernelGGL(scatter_kernel, dim3(n_chunks), dim3(kBuildThreads), 0, stream, A);
        hipLaunchKernelGGL(copy_back_kernel, dim3(n_chunks), dim3(kBuildThreads), 0, stream, A);
        BuildArgs next = A; // cut the level that offsets_kernel has just queued
        next.active = active[cur ^ 1];
        hipLaunchKernelGGL(setup_level_kernel, dim3(1), dim3(kBuildThreads), 0, stream, next, 1u);
        BUILD_HIP(hipGetLastError());
        BuildCtl now{};
        BUILD_HIP(hipMemcpyAsync(&now, A.ctl, sizeof now, hipMemcpyDeviceToHost, stream));
        BUILD_HIP(hipStreamSynchronize(stream));
        if (now.overflow || now.n_active > max_active || now.n_nodes > node_cap || now.n_chunks > chunk_cap ||
            (now.n_active > 0 && now.n_chunks == 0))
            return fail_build(WFPT_ERR_HIP, "device BVH build: internal capacity exceeded while splitting a level");
        n_active = now.n_active;
        n_chunks = now.n_chunks;
        cur ^= 1;
    }
    // ---- small subtrees
    BuildCtl after{};
    BUILD_HIP(hipMemcpy(&after, A.ctl, sizeof after, hipMemcpyDeviceToHost));
    const uint32_t n_small = after.n_small;
    if (n_small > 0) {
        hipLaunchKernelGGL(small_subtree_kernel, dim3(n_small), dim3(64), 0, stream, A, n_small);
        BUILD_HIP(hipGetLastError());
    }
    // ---- the reference's numbering: node ids are the order of split events of a depth-first walk (bvh.rs:192-209)
    BUILD_HIP(hipMemcpy(&after, A.ctl, sizeof after, hipMemcpyDeviceToHost));
    if (after.overflow) return fail_build(WFPT_ERR_HIP, "device BVH build: internal capacity exceeded in a subtree");
    const uint32_t n_prov = after.n_nodes;
    std::vector<BuildNode> prov(n_prov);
    std::vector<uint32_t> sub_pairs(n_small);
    BUILD_HIP(hipMemcpy(prov.data(), A.nodes, sizeof(BuildNode) * n_prov, hipMemcpyDeviceToHost));
    if (n_small) BUILD_HIP(hipMemcpy(sub_pairs.data(), A.sub_pairs, sizeof(uint32_t) * n_small, hipMemcpyDeviceToHost));
    std::vector<uint32_t> inner(n_prov, 0), h_final(n_prov, 0), h_pair(n_prov, 0);
    for (uint32_t x = n_prov; x-- > 0;) { // children always have larger provisional ids than their parent
        const BuildNode &nd = prov[x];
        if (nd.kind == 1) inner[x] = 1u + inner[nd.child] + inner[nd.child + 1u];
        else if (nd.kind == 2) inner[x] = sub_pairs[nd.sub];
    }
    const uint64_t total_nodes = 2ull + 2ull * inner[0];
    if (total_nodes > cap) return fail_build(WFPT_ERR_INVALID_ARGUMENT, "device BVH build: node capacity too small");
    h_final[0] = 0;
    h_pair[0] = inner[0] ? 2u : 0u;
    for (uint32_t x = 0; x < n_prov; ++x) {
        const BuildNode &nd = prov[x];
        if (nd.kind != 1) continue;
        const uint32_t l = nd.child, r = nd.child + 1u, p = h_pair[x];
        h_final[l] = p;
        h_final[r] = p + 1u;
        h_pair[l] = inner[l] ? p + 2u : 0u;                   // the left subtree's pairs follow immediately,
        h_pair[r] = inner[r] ? p + 2u + 2u * inner[l] : 0u;   // the right subtree's after all of them
    }
    BUILD_HIP(mem.alloc(&final_index, n_prov));
    BUILD_HIP(mem.alloc(&pair_index, n_prov));
    BUILD_HIP(hipMemcpyAsync(final_index, h_final.data(), sizeof(uint32_t) * n_prov, hipMemcpyHostToDevice, stream));
    BUILD_HIP(hipMemcpyAsync(pair_index, h_pair.data(), sizeof(uint32_t) * n_prov, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(emit_nodes_kernel, dim3((n_prov + kBuildThreads - 1) / kBuildThreads), dim3(kBuildThreads), 0, stream, A.nodes, n_prov,
                       final_index, pair_index, d_out);
    if (n_small)
        hipLaunchKernelGGL(emit_subtrees_kernel, dim3(n_small), dim3(64), 0, stream, A.small_roots, A.sub_base, A.sub_pairs, pair_index,
                           A.sub_nodes, d_out);
    hipLaunchKernelGGL(gather_prims_kernel<Prim>, dim3(prim_blocks), dim3(kBuildThreads), 0, stream, d_prims, A.ids, n, d_sorted);
    BUILD_HIP(hipGetLastError());
    BUILD_HIP(hipEventRecord(ev1, stream));
    BUILD_HIP(hipStreamSynchronize(stream));
    if (device_ms) BUILD_HIP(hipEventElapsedTime(device_ms, ev0, ev1));
    BUILD_HIP(hipMemcpy(out_nodes, d_out, sizeof(wfpt_bvh_node) * total_nodes, hipMemcpyDeviceToHost));
    BUILD_HIP(hipMemcpy(prims, d_sorted, sizeof(Prim) * n, hipMemcpyDeviceToHost));
    *n_nodes = static_cast<uint32_t>(total_nodes);
    return WFPT_OK;
}

} // namespace
} // namespace wfpt

extern "C" {

int wfpt_build_bvh_device(wfpt_sphere *spheres, uint32_t n, wfpt_bvh_node *nodes, uint32_t cap, uint32_t *n_nodes, int device,
                          float *device_ms) {
    return wfpt::build_on_device<0>(spheres, n, nodes, cap, n_nodes, wfpt::kMaxBins, device, device_ms);
}

int wfpt_build_bvh_triangles_device(wfpt_triangle *tris, uint32_t n, wfpt_bvh_node *nodes, uint32_t cap, uint32_t *n_nodes,
                                    uint32_t n_bins, int device, float *device_ms) {
    return wfpt::build_on_device<1>(tris, n, nodes, cap, n_nodes, n_bins, device, device_ms);
}

} // extern "C"
