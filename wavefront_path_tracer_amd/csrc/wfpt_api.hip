// wfpt_api.hip -- context and C ABI (include/wfpt.h) over the gfx950 kernels.
//
// Mirrors what PathTracer::new allocates and what PathTracer::run drives (gpu_wavefront_pt/src/
// path_tracer.rs:43-371) and the per-stage Kernel::run / get_timing API (gpu_wavefront_pt/src/kernel.rs).
// There is no CPU fallback: without a HIP device every device entry point fails with WFPT_ERR_NO_DEVICE.
#include "wfpt_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: the library is opened at run time by the gather entry points

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

using namespace wfpt;

namespace {
thread_local std::string g_last_error;
thread_local int g_last_status = 0; // status of the last failure on this thread (wfpt_create returns a pointer, not a code)

struct StageTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    bool pending = false;
    std::deque<double> window; // QueryResults::running_avg, query_gpu.rs:16,26-43 (10 deep)
};
} // namespace

namespace wfpt {
void set_last_error(const std::string &msg) { g_last_error = msg; } // for entry points without a context (wfpt_bvh_build.hip)
} // namespace wfpt

// 8-row bands of an image of `height` rows owned by rank `rank` of `world` (band k belongs to rank k % world)
static uint32_t bands_of(uint32_t height, uint32_t rank, uint32_t world) {
    const uint32_t n_bands = (height + 7u) / 8u;
    return n_bands > rank ? (n_bands - rank + world - 1u) / world : 0u;
}

struct wfpt_ctx {
    wfpt_params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t width = 0, height = 0;
    uint32_t tiles_x = 0, tiles_y_local = 0;
    uint32_t n_pixels = 0;      // pixels in this context's slab
    uint32_t pixel_capacity = 0;
    uint32_t capacity = 0;      // ray-queue slots (multiple of kChunk)
    uint32_t n_chunks_max = 0;
    uint32_t batch_max = 1;     // samples kept in flight by the device-resident loop
    size_t image_floats = 0;    // floats per image slice (one float4 per pixel: r, g, b, unused)
    size_t acc_floats = 0;      // floats of `accumulated` (the reference's stride-12 layout)
    uint32_t cus = 0, blocks_per_cu = 1;
    uint32_t last_slot = 0;     // batch slice that holds the most recent sample's bounce table
    bool has_inactive = false;
    Tiling tile{0, 1};

    float *ray_mem[2] = {nullptr, nullptr};
    RayQueue q[2]{};
    int cur = 0; // which queue is "ray_buffer"; the other one is "extension_ray_buffer"
    uint32_t *hit_mem = nullptr, *miss_mem = nullptr; // 3 planes each: (t, prim, ray index) / (ray index, dir.y, pixel)
    float4 *hit_rec = nullptr; // [classic_batch][capacity][2]: the path record extend leaves beside the hit queue for shade to stream
    bool hit_rec_valid = false; // slice 0's records describe the hit queue AND the ray queue as they are (stage API: a host may touch either between stages)
    HitQueue hq{};
    MissQueue mq{};
    float4 *d_shade_rec = nullptr;
    uint32_t *chunk_hits = nullptr, *chunk_miss = nullptr, *chunk_hit_base = nullptr, *chunk_miss_base = nullptr;
    uint16_t *mat_list = nullptr; // [3][batch][capacity] per-material hit lists
    uint32_t *chunk_mat = nullptr; // [3][batch][segments]
    // fused bounce path (default device-resident loop): path records and miss payloads ping-pong between wavefronts
    float4 *rec_mem[2] = {nullptr, nullptr};   // [batch][capacity][2]
    uint32_t *f_miss_mem[2] = {nullptr, nullptr};
    MissQueue f_mq[2]{};
    uint32_t *f_chunk_hits[2] = {nullptr, nullptr}, *f_chunk_miss[2] = {nullptr, nullptr}, *first_seg = nullptr;
    // class-binned fused loop (LDS-resident scenes; bounce_binned_kernel): per-segment class totals, the scan's per-class tables, the
    // work-item plan, and -- dispatch-keyed RNG -- extend's hit flags per thread index with the rank table made of them
    bool bin_capable = false;     // buffers exist (decided at wfpt_create: fused loop, pixel-keyed RNG, sizes within the record packing, flag clear)
    uint32_t *f_cls[2] = {nullptr, nullptr}, *first_seg_cls = nullptr, *plan = nullptr;
    uint2 *cls_table = nullptr;
    uint32_t bounce_binned_blocks_per_cu = 1;
    // multi-GPU gather of the band-sharded frame (RCCL over xGMI)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    float *gather_stage = nullptr; // root: [world - 1] slabs received from the peers
    float *gather_frame = nullptr; // root: assembled frame, whole bands (ceil(height / 8) * 8 rows)
    unsigned long long *d_stamps = nullptr; // diagnostic builds: 16 counters (wfpt_debug_read_stamps)
    float4 *rec_dense = nullptr; // HBM-resident scenes: per-ray results of the refill traversal, [batch][capacity][2]
    uint32_t classic_batch = 1; // slices of the stage-by-stage queues: batch_max when the loop runs unfused, else 1 (stage API)
    uint32_t bounce_blocks_per_cu = 1;
    bool fused = true;
    float *image = nullptr, *accumulated = nullptr;
    Control *ctl = nullptr;
    CameraDev *camera = nullptr;
    wfpt_gpu_camera h_camera{};      // host copy (the conservative traversal's range check, wfpt_update_scene)
    wfpt_bvh_node *d_nodes = nullptr;
    float4 *d_sphere_geom = nullptr;
    uint16_t *d_pair_parent = nullptr;
    uint32_t *d_pair_parent32 = nullptr;
    float4 *d_nodes4 = nullptr;      // HBM-resident scenes: the tree collapsed into four-wide nodes
    float4 *d_nodes_ch = nullptr;    // LDS-resident scenes: conservative (centre, half-extent) boxes, see build_nodes_ch
    float extent[3] = {0, 0, 0};     // per axis: the |coordinate| bound the conservative margin was sized for (origins up to 4x)
    bool ch_ok = false;              // nodes_ch exists (LDS scene, finite boxes)
    bool far_rays = false;           // wfpt_write_rays injected an origin beyond 4 x extent: the stage API's extend goes exact
    size_t gather_frame_floats = 0, gather_stage_floats = 0; // allocated sizes of the two gather buffers
    uint32_t *d_stack_spill = nullptr;
    uint32_t depth4 = 0;
    wfpt_sphere *d_spheres = nullptr;
    wfpt_triangle *d_triangles = nullptr;
    wfpt_material *d_materials = nullptr;
    SceneDev scene{};
    std::vector<uint32_t> h_prim_mat_type; // material_type per primitive, for wfpt_read_hits (extend.wgsl:199)

    uint32_t accumulate_grid = 0;
    uint32_t progress_frame = 0, accumulated_samples = 0;
    bool dev_frame_valid = false;

    std::map<uint32_t, std::pair<hipGraph_t, hipGraphExec_t>> graphs; // captured chain, keyed by samples per launch

    StageTimer timers[WFPT_STAGE_COUNT];
    std::vector<hipEvent_t> sample_events; // for wfpt_render_sample_timed
    std::string err;
};

namespace {

int fail(wfpt_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    g_last_error = msg;
    g_last_status = code;
    return code;
}
int hip_fail(wfpt_ctx *c, hipError_t e, const char *what) {
    return fail(c, e == hipErrorOutOfMemory ? WFPT_ERR_OUT_OF_MEMORY : WFPT_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
#define WFPT_HIP(c, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(c, e_, #call); \
    } while (0)

template <typename T> hipError_t dmalloc(T **p, size_t n) {
    return hipMalloc(reinterpret_cast<void **>(p), sizeof(T) * (n ? n : 1));
}

void set_queue(RayQueue &q, float *base, size_t cap) {
    q.base = base;
    q.cap = static_cast<uint32_t>(cap);
}

// Geometry of the viewport for this context (band sharding included).
void set_viewport(wfpt_ctx *c, uint32_t w, uint32_t h) {
    c->width = w;
    c->height = h;
    c->tiles_x = (w + 7) / 8;
    const uint32_t tiles_y = (h + 7) / 8;
    const uint32_t rank = c->tile.rank, world = c->tile.world;
    c->tiles_y_local = tiles_y > rank ? (tiles_y - rank + world - 1) / world : 0;
    c->n_pixels = world <= 1 ? w * h : c->tiles_y_local * 8u * w;
    c->has_inactive = (w % 8u) != 0 || (h % 8u) != 0;
}

// The traversal needs siblings at (2k, 2k+1), which BVHTree::subdivide guarantees (bvh.rs:160-161,
// 191-206), and at most 63 levels. Returns the tree depth or a negative status.
int validate_bvh(wfpt_ctx *c, const wfpt_bvh_node *nodes, uint32_t n_nodes, uint32_t n_spheres,
                 std::vector<uint32_t> &pair_parent) {
    if (n_nodes == 0 || n_nodes > (1u << 30)) return fail(c, WFPT_ERR_UNSUPPORTED, "BVH must have 1..2^30 nodes");
    pair_parent.assign(((n_nodes / 2u + 1u) + 7u) / 8u * 8u, 0);
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};
    uint32_t max_depth = 0, visited = 0;
    while (!todo.empty()) {
        auto [i, d] = todo.back();
        todo.pop_back();
        if (++visited > n_nodes) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "BVH has a cycle");
        max_depth = std::max(max_depth, d);
        const wfpt_bvh_node &nd = nodes[i];
        if (nd.prim_count > 0) {
            if (static_cast<uint64_t>(nd.left_first) + nd.prim_count > n_spheres)
                return fail(c, WFPT_ERR_INVALID_ARGUMENT, "BVH leaf references spheres out of range");
        } else {
            const uint32_t l = nd.left_first;
            if (l < 2 || (l & 1u) || l + 1 >= n_nodes)
                return fail(c, WFPT_ERR_UNSUPPORTED, "BVH children must sit at (2k, 2k+1), k >= 1 (bvh.rs layout)");
            pair_parent[l >> 1] = i;
            todo.push_back({l, d + 1});
            todo.push_back({l + 1, d + 1});
        }
    }
    if (max_depth > static_cast<uint32_t>(kMaxTrailDepth))
        return fail(c, WFPT_ERR_UNSUPPORTED, "BVH deeper than 63 levels");
    return static_cast<int>(max_depth);
}

void destroy_graph(wfpt_ctx *c) {
    for (auto &kv : c->graphs) {
        if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
        if (kv.second.first) (void)hipGraphDestroy(kv.second.first);
    }
    c->graphs.clear();
}

Batch batch_of(const wfpt_ctx *c, uint32_t n) {
    Batch b{};
    b.n = n;
    b.ctl_stride = static_cast<uint32_t>(sizeof(Control) / sizeof(uint32_t));
    b.ray_stride = 7 * static_cast<size_t>(c->capacity);
    b.queue_stride = c->capacity;
    b.chunk_stride = c->n_chunks_max;
    b.image_stride = c->image_floats;
    return b;
}
uint32_t extend_grid(const wfpt_ctx *c, uint32_t n) {
    const uint64_t items = static_cast<uint64_t>(c->n_chunks_max) * n;
    return static_cast<uint32_t>(std::min<uint64_t>(items, static_cast<uint64_t>(c->cus) * c->blocks_per_cu));
}
uint32_t consumer_grid(const wfpt_ctx *c, uint32_t n) {
    return std::min(c->n_chunks_max, std::max(64u, (c->cus * 8u + n - 1) / n));
}

// ---------------------------------------------------------------- argument builders
GenerateArgs generate_args(wfpt_ctx *c, uint32_t gx, uint32_t gy, bool fused, uint32_t nb = 1) {
    GenerateArgs a{};
    a.batch = batch_of(c, nb);
    a.q = c->q[c->cur];
    a.image = c->image;
    a.ctl = c->ctl;
    a.camera = c->camera;
    a.gx = gx; a.gy = gy;
    a.true_size = fused ? 1u : 0u;
    a.reset_image = fused ? 1u : 0u;
    a.set_n_in = fused ? 1u : 0u;
    a.capacity = c->capacity;
    a.tile = c->tile;
    return a;
}
ExtendArgs extend_args(wfpt_ctx *c, int qi, const uint32_t *n_in, uint32_t limit, uint32_t nb = 1, bool partition = true) {
    ExtendArgs a{};
    a.batch = batch_of(c, nb);
    a.partition = partition ? 1u : 0u;
    a.mat_list = c->mat_list;
    a.chunk_mat = c->chunk_mat;
    a.mat_list_mstride = static_cast<size_t>(c->classic_batch) * c->capacity;
    a.chunk_mat_mstride = static_cast<size_t>(c->classic_batch) * c->n_chunks_max;
    a.q = c->q[qi];
    a.hq = c->hq;
    a.mq = c->mq;
    a.chunk_hits = c->chunk_hits;
    a.chunk_miss = c->chunk_miss;
    a.rec_out = c->hit_rec;
    a.ctl = c->ctl;
    a.n_in = n_in;
    a.limit = std::min(limit, c->capacity);
    a.has_inactive = c->has_inactive ? 1u : 0u;
    a.scene = c->scene;
    return a;
}
// `first`: the first sample of the launch's slices (0: a batch is one chain of launches)
ScanArgs scan_args(wfpt_ctx *c, const uint32_t *n_in, uint32_t limit, bool fused, uint32_t bounce, uint32_t nb = 1, int fused_parity = -1, uint32_t first = 0) {
    ScanArgs a{};
    a.batch = batch_of(c, nb);
    const size_t co = static_cast<size_t>(first) * c->n_chunks_max;
    a.chunk_hits = c->chunk_hits + co; a.chunk_miss = c->chunk_miss + co;
    if (fused_parity >= 0) { // counts written by the fused bounce kernel of this wavefront
        a.chunk_hits = c->f_chunk_hits[fused_parity] + co;
        a.chunk_miss = c->f_chunk_miss[fused_parity] + co;
        a.first_seg = c->first_seg + co;
    }
    a.chunk_hit_base = c->chunk_hit_base + co; a.chunk_miss_base = c->chunk_miss_base + co;
    a.ctl = c->ctl + first;
    a.n_in = n_in + static_cast<size_t>(first) * a.batch.ctl_stride;
    a.limit = std::min(limit, c->capacity);
    a.fused = fused ? 1u : 0u;
    a.miss_floor = c->p.miss_floor;
    a.bounce = bounce;
    return a;
}
ShadeArgs shade_args(wfpt_ctx *c, int qi, const uint32_t *n_hits, uint32_t limit, uint32_t gx, uint32_t material,
                     bool count_out, uint32_t nb = 1) {
    ShadeArgs a{};
    a.batch = batch_of(c, nb);
    a.split = (material != 0xffffffffu || (c->p.flags & WFPT_FLAG_SPLIT_SHADE)) ? 1u : 0u;
    a.mat_list = c->mat_list;
    a.chunk_mat = c->chunk_mat;
    a.mat_list_mstride = static_cast<size_t>(c->classic_batch) * c->capacity;
    a.chunk_mat_mstride = static_cast<size_t>(c->classic_batch) * c->n_chunks_max;
    a.q = c->q[qi];
    a.ext = c->q[qi ^ 1];
    a.hq = c->hq;
    a.chunk_hits = c->chunk_hits; a.chunk_hit_base = c->chunk_hit_base;
    a.rec_in = c->hit_rec_valid ? c->hit_rec : nullptr;
    a.image = c->image;
    a.ctl = c->ctl;
    a.n_hits = n_hits;
    a.limit = std::min(limit, c->capacity);
    a.gx = gx;
    a.rng_mode = c->p.rng_mode;
    a.material = material;
    a.count_out = count_out ? 1u : 0u;
    a.image_width = c->width;
    a.scene = c->scene;
    a.tile = c->tile;
    return a;
}
MissArgs miss_args(wfpt_ctx *c, int qi, const uint32_t *n_miss, uint32_t limit, uint32_t nb = 1) {
    MissArgs a{};
    a.batch = batch_of(c, nb);
    a.q = c->q[qi];
    a.mq = c->mq;
    a.chunk_miss = c->chunk_miss; a.chunk_miss_base = c->chunk_miss_base;
    a.image = c->image;
    a.ctl = c->ctl;
    a.n_miss = n_miss;
    a.limit = std::min(limit, c->capacity);
    a.image_width = c->width;
    a.tile = c->tile;
    return a;
}
AccumulateArgs accumulate_args(wfpt_ctx *c, uint32_t n_pixels, bool bookkeeping, uint32_t nb = 1) {
    AccumulateArgs a{};
    a.batch = batch_of(c, nb);
    a.image = c->image;
    a.accumulated = c->accumulated;
    a.ctl = c->ctl;
    a.n_pixels = std::min(n_pixels, c->n_pixels);
    a.bookkeeping = bookkeeping ? 1u : 0u;
    return a;
}

constexpr uint32_t kPlanWords = kMaxBatch * (kBinClasses + 2) + 8; // work-item plan of the class-binned loop
BounceArgs bounce_args(wfpt_ctx *c, int in_parity, int out_parity, uint32_t nb, uint32_t first = 0) {
    BounceArgs a{};
    a.batch = batch_of(c, nb);
    const size_t co = static_cast<size_t>(first) * c->n_chunks_max, qo = static_cast<size_t>(first) * c->capacity;
    a.stamps = c->d_stamps;
    a.rec_in = c->rec_mem[in_parity] + 2 * qo;
    a.rec_out = c->rec_mem[out_parity] + 2 * qo;
    a.in_hits = c->f_chunk_hits[in_parity] + co;
    a.in_miss = c->f_chunk_miss[in_parity] + co;
    a.in_hit_base = c->chunk_hit_base + co;
    a.in_first_seg = c->first_seg + co;
    a.out_hits = c->f_chunk_hits[out_parity] + co;
    a.out_miss = c->f_chunk_miss[out_parity] + co;
    a.mq_in = c->f_mq[in_parity]; a.mq_in.base += qo;
    a.mq_out = c->f_mq[out_parity]; a.mq_out.base += qo;
    if (c->bin_capable) {
        constexpr size_t kWords = ClsPack<kBinClasses>::kWords;
        a.plan = c->plan;
        a.plan_seg_off = nb * kBinClasses + 1u;
        a.plan_miss_off = a.plan_seg_off + nb + 1u;
        a.cls_table = c->cls_table + co * kBinClasses;
        a.first_seg_cls = c->first_seg_cls + co * kBinClasses;
        a.out_cls = c->f_cls[out_parity] + co * kWords;
    }
    a.image = c->image + static_cast<size_t>(first) * c->image_floats;
    a.ctl = c->ctl + first;
    a.camera = c->camera;
    a.gx = c->tiles_x;
    a.gy = c->tiles_y_local;
    a.capacity = c->capacity;
    a.rng_mode = c->p.rng_mode;
    a.image_width = c->width;
    a.tile = c->tile;
    a.scene = c->scene;
    return a;
}
RefillArgs refill_args(wfpt_ctx *c, int in_parity, uint32_t nb, uint32_t first = 0) {
    RefillArgs a{};
    a.batch = batch_of(c, nb);
    const size_t co = static_cast<size_t>(first) * c->n_chunks_max, qo = static_cast<size_t>(first) * c->capacity;
    a.stamps = c->d_stamps;
    a.rec_in = c->rec_mem[in_parity] + 2 * qo;
    a.dense_out = c->rec_dense + 2 * qo;
    a.in_hits = c->f_chunk_hits[in_parity] + co;
    a.in_hit_base = c->chunk_hit_base + co;
    a.in_first_seg = c->first_seg + co;
    a.image = c->image + static_cast<size_t>(first) * c->image_floats;
    a.ctl = c->ctl + first;
    a.camera = c->camera;
    a.gx = c->tiles_x;
    a.gy = c->tiles_y_local;
    a.capacity = c->capacity;
    a.rng_mode = c->p.rng_mode;
    a.image_width = c->width;
    a.tile = c->tile;
    a.scene = c->scene;
    return a;
}
CompactArgs compact_args(wfpt_ctx *c, int out_parity, uint32_t nb, uint32_t first = 0) {
    CompactArgs a{};
    a.batch = batch_of(c, nb);
    const size_t co = static_cast<size_t>(first) * c->n_chunks_max, qo = static_cast<size_t>(first) * c->capacity;
    a.dense_in = c->rec_dense + 2 * qo;
    a.rec_out = c->rec_mem[out_parity] + 2 * qo;
    a.mq_out = c->f_mq[out_parity]; a.mq_out.base += qo;
    a.out_hits = c->f_chunk_hits[out_parity] + co;
    a.out_miss = c->f_chunk_miss[out_parity] + co;
    a.ctl = c->ctl + first;
    a.capacity = c->capacity;
    return a;
}
MissArgs fused_miss_args(wfpt_ctx *c, int parity, uint32_t nb, uint32_t first = 0) { // miss_kernel over the fused loop's miss queue of one wavefront
    MissArgs a = miss_args(c, 0, &c->ctl->miss_n, c->capacity, nb);
    const size_t co = static_cast<size_t>(first) * c->n_chunks_max, qo = static_cast<size_t>(first) * c->capacity;
    a.mq = c->f_mq[parity]; a.mq.base += qo;
    a.chunk_miss = c->f_chunk_miss[parity] + co;
    a.chunk_miss_base = c->chunk_miss_base + co;
    a.image = c->image + static_cast<size_t>(first) * c->image_floats;
    a.ctl = c->ctl + first;
    a.n_miss = &c->ctl[first].miss_n;
    return a;
}
ScanBinnedArgs scan_binned_args(wfpt_ctx *c, uint32_t bounce, uint32_t nb, int parity, uint32_t first = 0) {
    ScanBinnedArgs a{};
    a.batch = batch_of(c, nb);
    constexpr size_t kWords = ClsPack<kBinClasses>::kWords;
    const size_t co = static_cast<size_t>(first) * c->n_chunks_max;
    a.chunk_hits = c->f_chunk_hits[parity] + co;
    a.chunk_miss = c->f_chunk_miss[parity] + co;
    a.chunk_cls = c->f_cls[parity] + co * kWords;
    a.cls_table = c->cls_table + co * kBinClasses;
    a.first_seg_cls = c->first_seg_cls + co * kBinClasses;
    a.ctl = c->ctl + first;
    a.n_in = &c->ctl[first].n_in;
    a.limit = c->capacity;
    a.miss_floor = c->p.miss_floor;
    a.bounce = bounce;
    return a;
}
PlanArgs plan_args(wfpt_ctx *c, uint32_t nb, bool last, uint32_t first = 0) {
    PlanArgs a{};
    a.batch = batch_of(c, nb);
    a.ctl = c->ctl + first;
    a.plan = c->plan;
    a.plan_seg_off = nb * kBinClasses + 1u;
    a.plan_miss_off = a.plan_seg_off + nb + 1u;
    a.last = last ? 1u : 0u;
    return a;
}
// the class-binned loop runs when its buffers exist and the scene at hand is one it is built for: in LDS, primitive indices within
// the record's 16 bits
bool use_binned(const wfpt_ctx *c) {
    return c->bin_capable && c->fused && c->scene.lds_scene && !c->rec_dense && c->scene.n_spheres <= (1u << 16) &&
           c->batch_max * static_cast<uint32_t>(kBinClasses) <= 1024u && c->p.rng_mode == WFPT_RNG_PIXEL;
}
uint32_t bounce_grid(const wfpt_ctx *c, uint32_t n) {
    // hit items + miss items never exceed 1.25 work items per segment
    const uint64_t items = (static_cast<uint64_t>(c->n_chunks_max) * 5u / 4u + 1u) * n;
    return static_cast<uint32_t>(std::min<uint64_t>(items, static_cast<uint64_t>(c->cus) * c->bounce_blocks_per_cu));
}

// One batch of fused samples on the stream (pt:291-368). `ev`: optional (stage, start, stop) event recorder.
struct EventRec { int stage; hipEvent_t start, stop; };

// The wavefront chain of the samples [first, first + nb) of a batch on `st`, up to (not including) accumulate. Returns false if this
// context's loop is not one of the fused ones (the stage kernels one by one: enqueue_batch).
template <typename Timed>
int enqueue_fused_chain(wfpt_ctx *c, Timed &timed, uint32_t nb, uint32_t first, hipStream_t st) {
    if (c->fused && c->rec_dense && !c->scene.exact) {
        // HBM-resident scene: traversal with dynamic lane refill. Per wavefront: (miss_kernel of the previous one) |
        // refill-trace into dense per-ray records | compact into the queues | scan; then shade+miss of the last one.
        const uint32_t grid = c->cus * c->bounce_blocks_per_cu;
        if (WFPT_PRESHADE) // generate_rays at full waves into the dense array, then the traversal refills from it
            WFPT_HIP(c, timed(WFPT_STAGE_GENERATE_RAYS, [&] { return launch_generate_dense(refill_args(c, 1, nb, first), st); }));
        WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE_FIRST, [&] { return launch_refill(refill_args(c, 1, nb, first), kBounceFirst, grid, st, WFPT_PRESHADE != 0); }));
        for (uint32_t b = 0; b < c->p.max_wavefronts; ++b) {
            const int par = static_cast<int>(b & 1u);
            WFPT_HIP(c, timed(WFPT_STAGE_COMPACT, [&] { return launch_compact(compact_args(c, par, nb, first), c->n_chunks_max, st); }));
            WFPT_HIP(c, timed(WFPT_STAGE_SCAN,
                              [&] { return launch_scan(scan_args(c, &c->ctl->n_in, c->capacity, true, b, nb, par, first), st); }));
            if (b + 1 < c->p.max_wavefronts) {
                WFPT_HIP(c, timed(WFPT_STAGE_MISS, [&] { return launch_miss(fused_miss_args(c, par, nb, first), consumer_grid(c, nb), st); }));
                if (WFPT_PRESHADE) // shade at full waves into the dense array, then the traversal refills from it
                    WFPT_HIP(c, timed(WFPT_STAGE_SHADE, [&] { return launch_shade_rays(refill_args(c, par, nb, first), c->n_chunks_max, st); }));
                WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE, [&] { return launch_refill(refill_args(c, par, nb, first), kBounceMiddle, grid, st, WFPT_PRESHADE != 0); }));
            } else {
                WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE_LAST, [&] { return launch_bounce(bounce_args(c, par, par ^ 1, nb, first), kBounceLast, bounce_grid(c, nb), st); }));
            }
        }
        return WFPT_OK;
    }
    if (use_binned(c)) {
        // the same chain with the hit queue binned by cost class: generate+extend | scan, plan | (shade+extend+miss | scan, plan) x (max - 1) | shade+miss
        const uint64_t items = (static_cast<uint64_t>(c->n_chunks_max) * 5u / 4u + 1u) * nb;
        const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(items, static_cast<uint64_t>(c->cus) * c->bounce_binned_blocks_per_cu));
        WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE_FIRST, [&] { return launch_bounce_binned(bounce_args(c, 1, 0, nb, first), kBounceFirst, grid, st); }));
        for (uint32_t b = 0; b < c->p.max_wavefronts; ++b) {
            const int par = static_cast<int>(b & 1u);
            const bool last = b + 1 >= c->p.max_wavefronts;
            WFPT_HIP(c, timed(WFPT_STAGE_SCAN, [&] {
                         hipError_t e = launch_scan_binned(scan_binned_args(c, b, nb, par, first), st);
                         return e != hipSuccess ? e : launch_plan(plan_args(c, nb, last, first), st);
                     }));
            WFPT_HIP(c, timed(last ? WFPT_STAGE_BOUNCE_LAST : WFPT_STAGE_BOUNCE,
                              [&] { return launch_bounce_binned(bounce_args(c, par, par ^ 1, nb, first), last ? kBounceLast : kBounceMiddle, grid, st); }));
        }
        return WFPT_OK;
    }
    // generate+extend | scan | (shade+extend+miss | scan) x (max_wavefronts - 1) | shade+miss
    const uint32_t grid = bounce_grid(c, nb);
    WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE_FIRST, [&] { return launch_bounce(bounce_args(c, 1, 0, nb, first), kBounceFirst, grid, st); }));
    for (uint32_t b = 0; b < c->p.max_wavefronts; ++b) {
        const int par = static_cast<int>(b & 1u);
        WFPT_HIP(c, timed(WFPT_STAGE_SCAN,
                          [&] { return launch_scan(scan_args(c, &c->ctl->n_in, c->capacity, true, b, nb, par, first), st); }));
        if (b + 1 < c->p.max_wavefronts)
            WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE, [&] { return launch_bounce(bounce_args(c, par, par ^ 1, nb, first), kBounceMiddle, grid, st); }));
        else
            WFPT_HIP(c, timed(WFPT_STAGE_BOUNCE_LAST, [&] { return launch_bounce(bounce_args(c, par, par ^ 1, nb, first), kBounceLast, grid, st); }));
    }
    return WFPT_OK;
}

int enqueue_batch(wfpt_ctx *c, std::vector<EventRec> *ev, uint32_t nb) {
    size_t next_event = 0;
    auto timed = [&](int stage, auto &&launch) -> hipError_t {
        if (!ev) return launch();
        if (next_event + 2 > c->sample_events.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e;
                hipError_t r = hipEventCreate(&e);
                if (r != hipSuccess) return r;
                c->sample_events.push_back(e);
            }
        }
        hipEvent_t s = c->sample_events[next_event], t = c->sample_events[next_event + 1];
        next_event += 2;
        hipError_t r = hipEventRecord(s, c->stream);
        if (r != hipSuccess) return r;
        r = launch();
        if (r != hipSuccess) return r;
        r = hipEventRecord(t, c->stream);
        ev->push_back({stage, s, t});
        return r;
    };
    c->cur = 0;
    const bool split = (c->p.flags & WFPT_FLAG_SPLIT_SHADE) != 0;
    if (c->fused) {
        if (int r = enqueue_fused_chain(c, timed, nb, 0, c->stream); r != WFPT_OK) return r;
        WFPT_HIP(c, timed(WFPT_STAGE_ACCUMULATE, [&] {
                     return launch_accumulate(accumulate_args(c, c->n_pixels, true, nb), c->accumulate_grid, c->stream);
                 }));
        return WFPT_OK;
    }
    WFPT_HIP(c, timed(WFPT_STAGE_GENERATE_RAYS,
                      [&] { return launch_generate(generate_args(c, c->tiles_x, c->tiles_y_local, true, nb), c->stream); }));
    for (uint32_t b = 0; b < c->p.max_wavefronts; ++b) {
        const int qi = static_cast<int>(b & 1u);
        WFPT_HIP(c, timed(WFPT_STAGE_EXTEND, [&] {
                     return launch_extend(extend_args(c, qi, &c->ctl->n_in, c->capacity, nb, split), extend_grid(c, nb), c->stream);
                 }));
        WFPT_HIP(c, timed(WFPT_STAGE_SCAN,
                          [&] { return launch_scan(scan_args(c, &c->ctl->n_in, c->capacity, true, b, nb), c->stream); }));
        if (split) { // one launch, blockIdx.z = material class (README.md:19's by-material shade kernels)
            WFPT_HIP(c, timed(WFPT_STAGE_SHADE_LAMBERTIAN, [&] {
                         ShadeArgs sa = shade_args(c, qi, &c->ctl->shade_n, c->capacity, 0, 0xffffffffu, false, nb);
                         sa.rec_in = c->hit_rec; // this wavefront's extend has just written them
                         return launch_shade(sa, consumer_grid(c, nb), c->stream);
                     }));
        } else {
            WFPT_HIP(c, timed(WFPT_STAGE_SHADE, [&] {
                         ShadeArgs sa = shade_args(c, qi, &c->ctl->shade_n, c->capacity, 0, 0xffffffffu, false, nb);
                         sa.rec_in = c->hit_rec; // this wavefront's extend has just written them
                         return launch_shade(sa, consumer_grid(c, nb), c->stream);
                     }));
        }
        WFPT_HIP(c, timed(WFPT_STAGE_MISS, [&] {
                     return launch_miss(miss_args(c, qi, &c->ctl->miss_n, c->capacity, nb), consumer_grid(c, nb), c->stream);
                 }));
    }
    WFPT_HIP(c, timed(WFPT_STAGE_ACCUMULATE, [&] {
                 return launch_accumulate(accumulate_args(c, c->n_pixels, true, nb), c->accumulate_grid, c->stream);
             }));
    return WFPT_OK;
}

int ensure_device_frame(wfpt_ctx *c) {
    if (c->dev_frame_valid) return WFPT_OK;
    const wfpt_frame_buffer f{c->width, c->height, c->progress_frame + 1u, 0u}; // parameters.rs:78-83; pt:296
    WFPT_HIP(c, launch_set_frame(c->ctl, f, c->stream)); // on the stream: frames can be queued back to back without a host sync
    c->dev_frame_valid = true;
    return WFPT_OK;
}

// Renders `nb` samples (frames progress_frame+1 ...) with one pass of the chain; 1 <= nb <= batch_max.
int render_batch(wfpt_ctx *c, uint32_t nb) {
    WFPT_HIP(c, hipSetDevice(c->device));
    if (int r = ensure_device_frame(c); r != WFPT_OK) return r;
    if (c->p.flags & WFPT_FLAG_NO_GRAPH) {
        if (int r = enqueue_batch(c, nullptr, nb); r != WFPT_OK) return r;
    } else {
        auto it = c->graphs.find(nb);
        if (it == c->graphs.end()) {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            WFPT_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int r = enqueue_batch(c, nullptr, nb);
            const hipError_t e = hipStreamEndCapture(c->stream, &g);
            if (r != WFPT_OK || e != hipSuccess) {
                if (g) (void)hipGraphDestroy(g); // a failed capture must not leak its partial graph
                return r != WFPT_OK ? r : hip_fail(c, e, "hipStreamEndCapture");
            }
            if (const hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); ie != hipSuccess) {
                (void)hipGraphDestroy(g);
                return hip_fail(c, ie, "hipGraphInstantiate");
            }
            it = c->graphs.emplace(nb, std::make_pair(g, ge)).first;
        }
        WFPT_HIP(c, hipGraphLaunch(it->second.second, c->stream));
    }
    c->cur = static_cast<int>(c->p.max_wavefronts & 1u);
    c->hit_rec_valid = false;     // (the stage API starts from the queues as the loop left them)
    c->progress_frame += nb;      // the device advanced ctl->frame.frame itself
    c->accumulated_samples += nb; // pt:363
    c->last_slot = nb - 1;
    return WFPT_OK;
}

int render_many(wfpt_ctx *c, uint32_t n_samples) {
    while (n_samples > 0) { // full batches, then one smaller batch for the remainder
        const uint32_t nb = std::min(n_samples, c->batch_max);
        if (int r = render_batch(c, nb); r != WFPT_OK) return r;
        n_samples -= nb;
    }
    return WFPT_OK;
}

int stage_begin(wfpt_ctx *c, int stage) {
    StageTimer &t = c->timers[stage];
    if (!t.start) {
        WFPT_HIP(c, hipEventCreate(&t.start));
        WFPT_HIP(c, hipEventCreate(&t.stop));
    }
    WFPT_HIP(c, hipEventRecord(t.start, c->stream));
    return WFPT_OK;
}
int stage_end(wfpt_ctx *c, int stage) {
    StageTimer &t = c->timers[stage];
    WFPT_HIP(c, hipEventRecord(t.stop, c->stream));
    t.pending = true;
    return WFPT_OK;
}

// ---------------------------------------------------------------- scene upload (wfpt_create, wfpt_update_scene)
// The conservative boxes of trace_ray_conservative (wfpt_kernels.hip): node i as (centre | left_first), (half-extent |
// prim_count). [c - h, c + h] encloses the caller's box, and h is then grown by margin = 2^-19 * extent per axis, where
// extent >= every |coordinate| of the scene on that axis and >= a quarter of the camera's. The device computes, per axis,
//   tc = fl(c b + nox), nox = -fl(o b);  t_entry = fl(tc - h |b|), t_exit = fl(tc + h |b|)      (b = clamped 1 / d)
// whose errors against the exact ((c -+ h) - o) b are at most 2^-24 (3 |o| + 2 |c| + h) |b| <= 2^-24 * 15 extent * |b| for
// |o| <= 4 extent, |c|, h <= extent: less than half of margin * |b|. So the computed entry distance never exceeds the
// exact box's and the computed exit distance never falls below it: the test can only say "enter" more often than the
// reference's (ex:164-183), never less. Returns false (=> the exact test is used) when a box is not finite.
// Inner boxes of the free walks are grown by margin = 2^kMarginLog2 * extent per axis. The margin pays for two things:
//  (a) the rounding error of the one-fma plane distances, < 2^-20 * extent (build_nodes_ch), and
//  (b) the rounding slack of the PRIMITIVE test: the reference's sphere test (ex:185-210) accepts a ray that passes up to
//      about 6 * 2^-24 * D^2 / r outside a sphere of radius r whose centre is D away from the ray's origin (its discriminant
//      b^2 - a c cancels ~12 * 2^-24 * a |o - c|^2). Such a "hit" is the reference's hit whenever the reference tests the
//      primitive -- through a leaf box that passes, or blindly (visit_leaf) -- so the free walk must at least REACH every
//      leaf the ray passes that close to: margin >= slack, i.e. D <= sqrt(margin * r / (6 * 2^-24)). safe_region() turns this
//      into one ball of origins per scene; rays from outside it are traced by the reference's own walk (far_origin).
constexpr int kMarginLog2 = -17;

// per axis: a bound on every |coordinate| of the scene and on a quarter of the camera's reach; false if a box is not finite
bool scene_extent(const wfpt_bvh_node *nodes, uint32_t n_nodes, const float cam_reach[3], float extent[3]) {
    for (int ax = 0; ax < 3; ++ax) {
        float e = 0.25f * cam_reach[ax];
        for (uint32_t i = 0; i < n_nodes; ++i) {
            if (i == 1) continue; // the pad slot (bvh.rs:160-161)
            const float lo = nodes[i].aabb_min[ax], hi = nodes[i].aabb_max[ax];
            if (!std::isfinite(lo) || !std::isfinite(hi) || hi < lo) return false;
            e = std::max(e, std::max(std::fabs(lo), std::fabs(hi)));
        }
        if (!(e < 1e30f)) return false;
        extent[ax] = e;
    }
    return true;
}
bool build_nodes_ch(const wfpt_bvh_node *nodes, uint32_t n_nodes, const float cam_reach[3], std::vector<float4> &out, float extent[3]) {
    auto up = [](double v) { // smallest float >= v
        float f = static_cast<float>(v);
        if (static_cast<double>(f) < v) f = std::nextafterf(f, INFINITY);
        return f;
    };
    if (!scene_extent(nodes, n_nodes, cam_reach, extent)) return false;
    out.resize(2 * static_cast<size_t>(n_nodes));
    for (uint32_t i = 0; i < n_nodes; ++i) {
        float c3[3], h3[3];
        for (int ax = 0; ax < 3; ++ax) {
            const double lo = nodes[i].aabb_min[ax], hi = nodes[i].aabb_max[ax];
            const float c = static_cast<float>(0.5 * (lo + hi));
            const double h = std::max(static_cast<double>(c) - lo, hi - static_cast<double>(c)); // exact in double
            c3[ax] = c;
            h3[ax] = up(static_cast<double>(up(h)) + std::ldexp(static_cast<double>(extent[ax]), kMarginLog2));
        }
        float lf, pc;
        std::memcpy(&lf, &nodes[i].left_first, 4);
        std::memcpy(&pc, &nodes[i].prim_count, 4);
        out[2 * i] = make_float4(c3[0], c3[1], c3[2], lf);
        out[2 * i + 1] = make_float4(h3[0], h3[1], h3[2], pc);
    }
    return true;
}

// The traversals that prune inner boxes freely (trace_ray_conservative, trace_ray4) test a LEAF's box with the reference's
// arithmetic on a box they recompute from the leaf's primitives, and rely on the boxes of a BVH nesting. Both are facts about
// the caller's tree, checked here: every leaf box equals the union of its primitives' boxes as the builder computes them
// (sphere.rs:22-26 / the triangle policy of wfpt_host.cpp; update_node_bounds, bvh.rs:58-70 -- values compared with ==, so
// the sign of a zero bound is free), and every inner box contains its children's. A tree that fails is walked with the
// reference's own test everywhere (decide_exact).
bool tree_is_recomputable(const wfpt_bvh_node *nodes, uint32_t n_nodes, const wfpt_sphere *spheres, const wfpt_triangle *triangles) {
    for (uint32_t i = 0; i < n_nodes; ++i) {
        if (i == 1) continue;
        const wfpt_bvh_node &nd = nodes[i];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (nd.prim_count > 0) {
            for (uint32_t k = 0; k < nd.prim_count; ++k) {
                const uint32_t p = nd.left_first + k;
                for (int ax = 0; ax < 3; ++ax) {
                    float a, b;
                    if (spheres) {
                        a = spheres[p].center[ax] - spheres[p].radius;
                        b = spheres[p].center[ax] + spheres[p].radius;
                    } else {
                        const float v0 = triangles[p].v0[ax], v1 = triangles[p].v0[ax] + triangles[p].e1[ax], v2 = triangles[p].v0[ax] + triangles[p].e2[ax];
                        a = std::fmin(std::fmin(v0, v1), v2);
                        b = std::fmax(std::fmax(v0, v1), v2);
                    }
                    lo[ax] = std::fmin(lo[ax], a);
                    hi[ax] = std::fmax(hi[ax], b);
                }
            }
            for (int ax = 0; ax < 3; ++ax)
                if (!(lo[ax] == nd.aabb_min[ax]) || !(hi[ax] == nd.aabb_max[ax])) return false;
        } else {
            for (uint32_t c = nd.left_first; c <= nd.left_first + 1u; ++c)
                for (int ax = 0; ax < 3; ++ax)
                    if (!(nodes[c].aabb_min[ax] >= nd.aabb_min[ax]) || !(nodes[c].aabb_max[ax] <= nd.aabb_max[ax])) return false;
        }
    }
    return true;
}

// The ball of ray origins for which every sphere's test slack stays within 7/8 of the smallest margin (1/8 is the plane
// distances' own rounding): centre = the middle of the box of all sphere centres but the largest sphere's, radius = the minimum
// over spheres of D_safe(r) - |centre - c|. Triangles (build extension): no closed bound is claimed, the region is unbounded.
void safe_region(const wfpt_sphere *spheres, uint32_t n, const float extent[3], float centre[3], float *r2) {
    centre[0] = centre[1] = centre[2] = 0.0f;
    *r2 = INFINITY;
    if (!spheres || n == 0) return;
    uint32_t biggest = 0;
    for (uint32_t i = 1; i < n; ++i)
        if (spheres[i].radius > spheres[biggest].radius) biggest = i;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i) {
        if (i == biggest && n > 1) continue;
        for (int ax = 0; ax < 3; ++ax) {
            lo[ax] = std::min(lo[ax], static_cast<double>(spheres[i].center[ax]));
            hi[ax] = std::max(hi[ax], static_cast<double>(spheres[i].center[ax]));
        }
    }
    for (int ax = 0; ax < 3; ++ax) centre[ax] = static_cast<float>(0.5 * (lo[ax] + hi[ax]));
    const double margin = 0.875 * std::ldexp(static_cast<double>(std::min(extent[0], std::min(extent[1], extent[2]))), kMarginLog2);
    double radius = INFINITY;
    for (uint32_t i = 0; i < n; ++i) {
        const double r = std::fabs(static_cast<double>(spheres[i].radius));
        const double d_safe = std::sqrt(margin * r / (6.0 * std::ldexp(1.0, -24)));
        double dist2 = 0.0;
        for (int ax = 0; ax < 3; ++ax) dist2 += (spheres[i].center[ax] - centre[ax]) * (spheres[i].center[ax] - centre[ax]);
        radius = std::min(radius, d_safe - std::sqrt(dist2));
    }
    *r2 = radius > 0.0 ? static_cast<float>(radius * radius) : -1.0f; // -1: no origin is safe, every ray takes the reference's walk
}

// ray origins the camera can produce: |position| + the lens radius (gr:73-79), per axis
void camera_reach(const wfpt_gpu_camera &cam, float reach[3]) {
    const float r = cam.defocus_radius > 0.0f ? cam.defocus_radius : 0.0f;
    for (int ax = 0; ax < 3; ++ax) reach[ax] = std::fabs(cam.position[ax]) + r;
}

// Which box test the traversal kernels run: the reference's own (WFPT_FLAG_EXACT_TRAVERSAL, or whenever the conservative
// boxes' error bound does not cover the rays at hand) or the conservative one.
void decide_exact(wfpt_ctx *c, const float cam_reach[3]) {
    bool exact = (c->p.flags & WFPT_FLAG_EXACT_TRAVERSAL) != 0 || (c->scene.lds_scene && !c->ch_ok) || c->far_rays;
    // A scene beyond LDS without four-wide nodes (WFPT_FLAG_BINARY_BVH, a tree that is not recomputable, a leaf collapse_bvh4
    // refuses) walks the caller's binary tree: with the reference's own test and its 1e30 miss value, i.e. with its blind descent
    // (ex:124). trace_ray<EXACT = false> -- the reference's arithmetic but kBoxMiss -- skips that descent and with it hits the
    // reference reports (tests/test_traversal_model.py); it is never the walk of a context.
    if (!c->scene.lds_scene && !c->scene.nodes4) exact = true;
    for (int ax = 0; ax < 3 && !exact && c->ch_ok; ++ax)
        if (!(cam_reach[ax] <= 4.0f * c->extent[ax])) exact = true;
    c->scene.exact = exact ? 1u : 0u; // (scenes beyond LDS: the four-wide nodes exist only when the tree passed tree_is_recomputable)
}

void free_scene(wfpt_ctx *c) {
    void *bufs[] = {c->d_shade_rec, c->d_nodes, c->d_nodes4, c->d_nodes_ch, c->d_stack_spill, c->d_sphere_geom, c->d_pair_parent,
                    c->d_pair_parent32, c->d_spheres, c->d_triangles, c->d_materials};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    c->d_shade_rec = nullptr; c->d_nodes = nullptr; c->d_nodes4 = nullptr; c->d_nodes_ch = nullptr; c->d_stack_spill = nullptr;
    c->d_sphere_geom = nullptr; c->d_pair_parent = nullptr; c->d_pair_parent32 = nullptr; c->d_spheres = nullptr; c->d_triangles = nullptr;
    c->d_materials = nullptr;
    c->scene = SceneDev{};
    c->ch_ok = false;
}

// Uploads the scene in the reference's layouts (pt:120-128) plus the traversal's derivatives of it: primitive geometry,
// the sibling-pair parent table, per-primitive shade records, conservative boxes (LDS-resident scenes) or four-wide nodes
// (scenes beyond LDS), and sizes the launches for it. `pair_parent` / `bvh_depth` come from validate_bvh.
int upload_scene(wfpt_ctx *c, const wfpt_sphere *spheres, const wfpt_triangle *triangles, uint32_t n_spheres, const wfpt_material *materials,
                 uint32_t n_materials, const wfpt_bvh_node *nodes, uint32_t n_nodes, const std::vector<uint32_t> &pair_parent,
                 uint32_t bvh_depth, const wfpt_gpu_camera *camera) {
    free_scene(c);
    const uint32_t prim_kind = triangles ? 1u : 0u;
    c->h_prim_mat_type.resize(n_spheres);
    for (uint32_t i = 0; i < n_spheres; ++i)
        c->h_prim_mat_type[i] = spheres ? spheres[i].material_type : triangles[i].material_type;
    WFPT_HIP(c, dmalloc(&c->d_nodes, n_nodes));
    WFPT_HIP(c, dmalloc(&c->d_materials, n_materials));
    WFPT_HIP(c, hipMemcpy(c->d_nodes, nodes, sizeof(wfpt_bvh_node) * n_nodes, hipMemcpyHostToDevice));
    WFPT_HIP(c, hipMemcpy(c->d_materials, materials, sizeof(wfpt_material) * n_materials, hipMemcpyHostToDevice));
    if (spheres) {
        std::vector<float4> geom(n_spheres);
        for (uint32_t i = 0; i < n_spheres; ++i)
            geom[i] = make_float4(spheres[i].center[0], spheres[i].center[1], spheres[i].center[2], spheres[i].radius);
        WFPT_HIP(c, dmalloc(&c->d_sphere_geom, n_spheres));
        WFPT_HIP(c, dmalloc(&c->d_spheres, n_spheres));
        WFPT_HIP(c, hipMemcpy(c->d_sphere_geom, geom.data(), sizeof(float4) * n_spheres, hipMemcpyHostToDevice));
        WFPT_HIP(c, hipMemcpy(c->d_spheres, spheres, sizeof(wfpt_sphere) * n_spheres, hipMemcpyHostToDevice));
        c->scene.prim_geom = c->d_sphere_geom;
    } else {
        WFPT_HIP(c, dmalloc(&c->d_triangles, n_spheres));
        WFPT_HIP(c, hipMemcpy(c->d_triangles, triangles, sizeof(wfpt_triangle) * n_spheres, hipMemcpyHostToDevice));
        c->scene.prim_geom = reinterpret_cast<const float4 *>(c->d_triangles); // 3 x float4 per triangle
    }
    // per-primitive shade records (sphere / triangle fields merged with the material they point at)
    {
        std::vector<ShadeRec> recs(n_spheres);
        for (uint32_t i = 0; i < n_spheres; ++i) {
            ShadeRec &r = recs[i];
            const wfpt_material &m = materials[spheres ? spheres[i].material_idx : triangles[i].material_idx];
            if (spheres) {
                r.v[0] = spheres[i].center[0]; r.v[1] = spheres[i].center[1]; r.v[2] = spheres[i].center[2];
            } else { // normalize(cross(e1, e2)): fixed operation order, v * (1 / length) like the kernels' normalize3, no contraction (-ffp-contract=off)
                const float *e1 = triangles[i].e1, *e2 = triangles[i].e2;
                const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
                const float inv_len = 1.0f / std::sqrt((nx * nx + ny * ny) + nz * nz);
                r.v[0] = nx * inv_len; r.v[1] = ny * inv_len; r.v[2] = nz * inv_len;
            }
            r.fuzz = m.fuzz;
            r.albedo[0] = m.albedo[0]; r.albedo[1] = m.albedo[1]; r.albedo[2] = m.albedo[2];
            r.refract_index = m.refract_index;
            r.mat_type = c->h_prim_mat_type[i]; // sphere.material_type, which extend copies into the payload (ex:199)
            r.cost_class = 1u + std::min<uint32_t>(r.mat_type > 2u ? 0u : r.mat_type, static_cast<uint32_t>(kBinClasses) - 2u); // `case 0u, default` of sh:102
            r._pad[0] = r._pad[1] = 0;
        }
        // cost class 0: the scene's DOMINANT primitives -- those whose own box has at least a quarter of the surface area of the box of
        // everything (the Shirley scene's ground sphere; none in a triangle soup). Rays that leave such a primitive cost differently from
        // rays that leave the clutter on it, whatever the material (tests/model_binning.py: the split that buys most of the traversal's gain).
        {
            auto area = [](const float lo[3], const float hi[3]) {
                const double dx = static_cast<double>(hi[0]) - lo[0], dy = static_cast<double>(hi[1]) - lo[1], dz = static_cast<double>(hi[2]) - lo[2];
                return 2.0 * (dx * dy + dy * dz + dz * dx);
            };
            const double root_area = area(nodes[0].aabb_min, nodes[0].aabb_max);
            for (uint32_t i = 0; i < n_spheres && n_spheres > 1; ++i) {
                float lo[3], hi[3];
                for (int ax = 0; ax < 3; ++ax) {
                    if (spheres) {
                        lo[ax] = spheres[i].center[ax] - spheres[i].radius; hi[ax] = spheres[i].center[ax] + spheres[i].radius;
                    } else {
                        const float v0 = triangles[i].v0[ax], v1 = v0 + triangles[i].e1[ax], v2 = v0 + triangles[i].e2[ax];
                        lo[ax] = std::fmin(std::fmin(v0, v1), v2); hi[ax] = std::fmax(std::fmax(v0, v1), v2);
                    }
                }
                if (area(lo, hi) >= 0.25 * root_area) recs[i].cost_class = 0u;
            }
        }
        static_assert(sizeof(ShadeRec) == 48, "ShadeRec is read as three float4");
        WFPT_HIP(c, dmalloc(&c->d_shade_rec, 3 * static_cast<size_t>(n_spheres)));
        WFPT_HIP(c, hipMemcpy(c->d_shade_rec, recs.data(), sizeof(ShadeRec) * n_spheres, hipMemcpyHostToDevice));
        c->scene.shade_rec = c->d_shade_rec;
    }
    // LDS variant when nodes + primitives + parents fit comfortably (>= 2 workgroups per CU); otherwise the
    // scene stays in HBM / Infinity Cache and extend reads it through L2.
    const uint32_t lds_need = extend_lds_bytes(n_nodes, n_spheres, prim_kind, true);
    const bool lds_scene = n_nodes <= 65536u && lds_need <= 80u * 1024u && !(c->p.flags & WFPT_FLAG_NO_LDS_SCENE);
    const bool flag_exact = (c->p.flags & WFPT_FLAG_EXACT_TRAVERSAL) != 0;
    float reach[3];
    camera_reach(*camera, reach);
    const bool recomputable = !flag_exact && tree_is_recomputable(nodes, n_nodes, spheres, triangles);
    if (lds_scene) {
        std::vector<uint16_t> p16(pair_parent.begin(), pair_parent.end());
        WFPT_HIP(c, dmalloc(&c->d_pair_parent, p16.size()));
        WFPT_HIP(c, hipMemcpy(c->d_pair_parent, p16.data(), sizeof(uint16_t) * p16.size(), hipMemcpyHostToDevice));
        std::vector<float4> ch;
        if (recomputable && build_nodes_ch(nodes, n_nodes, reach, ch, c->extent)) {
            WFPT_HIP(c, dmalloc(&c->d_nodes_ch, ch.size()));
            WFPT_HIP(c, hipMemcpy(c->d_nodes_ch, ch.data(), sizeof(float4) * ch.size(), hipMemcpyHostToDevice));
            c->scene.nodes_ch = c->d_nodes_ch;
            c->ch_ok = true;
        }
    } else {
        WFPT_HIP(c, dmalloc(&c->d_pair_parent32, pair_parent.size()));
        WFPT_HIP(c, hipMemcpy(c->d_pair_parent32, pair_parent.data(), sizeof(uint32_t) * pair_parent.size(), hipMemcpyHostToDevice));
    }
    // four-wide nodes for the HBM-resident traversal (not the reference's walk: WFPT_FLAG_EXACT_TRAVERSAL keeps the binary tree)
    if (!lds_scene && !(c->p.flags & WFPT_FLAG_BINARY_BVH) && recomputable) {
        std::vector<Node4> n4;
        float margin[3];
        if (scene_extent(nodes, n_nodes, reach, c->extent)) {
            for (int ax = 0; ax < 3; ++ax) margin[ax] = std::nextafterf(std::ldexp(c->extent[ax], kMarginLog2), INFINITY);
            if (collapse_bvh4(nodes, n_nodes, margin, n4, c->depth4)) {
                WFPT_HIP(c, dmalloc(&c->d_nodes4, 4 * n4.size()));
                WFPT_HIP(c, hipMemcpy(c->d_nodes4, n4.data(), sizeof(Node4) * n4.size(), hipMemcpyHostToDevice));
                c->scene.nodes4 = c->d_nodes4;
                c->scene.tile_n = static_cast<uint32_t>(std::min<size_t>(n4.size(), kTileNodesMax));
                c->ch_ok = true; // the one-fma plane distances of visit4 carry the same rounding allowance as the LDS walk's
            }
        }
    }
    const bool want_dense = c->fused && c->scene.nodes4 && !(c->p.flags & WFPT_FLAG_NO_REFILL);
    if (want_dense && !c->rec_dense) WFPT_HIP(c, dmalloc(&c->rec_dense, 2 * static_cast<size_t>(c->batch_max) * c->capacity));
    if (!want_dense && c->rec_dense) {
        (void)hipFree(c->rec_dense);
        c->rec_dense = nullptr;
    }
    c->scene.nodes = c->d_nodes;
    c->scene.pair_parent = c->d_pair_parent;
    c->scene.pair_parent32 = c->d_pair_parent32;
    c->scene.spheres = c->d_spheres;
    c->scene.triangles = c->d_triangles;
    c->scene.materials = c->d_materials;
    c->scene.n_nodes = n_nodes;
    c->scene.n_spheres = n_spheres;
    c->scene.n_materials = n_materials;
    c->scene.prim_kind = prim_kind;
    c->scene.lds_scene = lds_scene ? 1u : 0u;
    c->scene.lds_bytes = extend_lds_bytes(n_nodes, n_spheres, prim_kind, lds_scene);
    c->scene.depth = bvh_depth;
    c->scene.root_leaf = nodes[0].prim_count > 0 ? 1u : 0u;
    safe_region(spheres, n_spheres, c->extent, c->scene.safe_c, &c->scene.safe_r2);
    c->far_rays = false;
    decide_exact(c, reach);

    int blocks_per_cu = 1;
    WFPT_HIP(c, extend_blocks_per_cu(c->scene, &blocks_per_cu));
    c->blocks_per_cu = static_cast<uint32_t>(std::max(blocks_per_cu, 1));
    if (c->bin_capable && lds_scene) {
        int bb = 1;
        WFPT_HIP(c, bounce_binned_blocks_per_cu(c->scene, &bb));
        c->bounce_binned_blocks_per_cu = static_cast<uint32_t>(std::max(bb, 1));
    }
    int bounce_blocks = 1;
    WFPT_HIP(c, bounce_blocks_per_cu(c->scene, &bounce_blocks));
    c->bounce_blocks_per_cu = static_cast<uint32_t>(std::max(bounce_blocks, 1));
    if (c->scene.nodes4) { // spill area of the four-wide traversal's stack: at most 3 pushes per level
        const uint32_t need = 3u * (c->depth4 + 1u);
        const uint32_t spill_entries = need > kStack4Lds ? need - kStack4Lds : 1u;
        c->scene.spill_stride = c->cus * std::max(c->blocks_per_cu, c->bounce_blocks_per_cu) * static_cast<uint32_t>(kExtendThreads);
        WFPT_HIP(c, dmalloc(&c->d_stack_spill, static_cast<size_t>(spill_entries) * c->scene.spill_stride));
        c->scene.stack_spill = c->d_stack_spill;
    }
    return WFPT_OK;
}

// Rebuilds the reference-order queue from the segment-compacted one: segment c's entries go to
// positions [base[c], base[c] + count[c]).
template <typename Emit>
int walk_segments(wfpt_ctx *c, bool hits, uint32_t n, Emit emit) {
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    Control ctl;
    WFPT_HIP(c, hipMemcpy(&ctl, c->ctl, sizeof ctl, hipMemcpyDeviceToHost));
    const uint32_t n_chunks = (ctl.seg_n + kChunk - 1) / kChunk;
    const uint32_t total = hits ? ctl.hits : ctl.misses;
    if (n > total) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "read queue: n exceeds the entries the last extend produced");
    std::vector<uint32_t> count(n_chunks), base(n_chunks);
    if (n_chunks) {
        WFPT_HIP(c, hipMemcpy(count.data(), hits ? c->chunk_hits : c->chunk_miss, sizeof(uint32_t) * n_chunks, hipMemcpyDeviceToHost));
        WFPT_HIP(c, hipMemcpy(base.data(), hits ? c->chunk_hit_base : c->chunk_miss_base, sizeof(uint32_t) * n_chunks, hipMemcpyDeviceToHost));
    }
    const size_t span = static_cast<size_t>(n_chunks) * kChunk;
    std::vector<uint32_t> ridx(span), prim;
    std::vector<float> t;
    if (span) {
        WFPT_HIP(c, hipMemcpy(ridx.data(), hits ? c->hq.ridx() : c->mq.ridx(), sizeof(uint32_t) * span, hipMemcpyDeviceToHost));
        if (hits) {
            prim.resize(span);
            t.resize(span);
            WFPT_HIP(c, hipMemcpy(prim.data(), c->hq.prim(), sizeof(uint32_t) * span, hipMemcpyDeviceToHost));
            WFPT_HIP(c, hipMemcpy(t.data(), c->hq.t(), sizeof(float) * span, hipMemcpyDeviceToHost));
        }
    }
    for (uint32_t ch = 0; ch < n_chunks; ++ch)
        for (uint32_t r = 0; r < count[ch]; ++r) {
            const uint32_t pos = base[ch] + r;
            if (pos >= n) continue;
            const size_t slot = static_cast<size_t>(ch) * kChunk + r;
            emit(pos, ridx[slot], hits ? t[slot] : 0.0f, hits ? prim[slot] : 0u);
        }
    return WFPT_OK;
}


} // namespace

extern "C" {

static int alloc_gather_buffers(wfpt_ctx *c); // with the RCCL gather, below

const char *wfpt_build_info(void) {
    static const std::string info = std::string("arch=gfx950;chunk=") + std::to_string(kChunk) +
                                    ";extend_threads=" + std::to_string(kExtendThreads) + ";max_batch=" + std::to_string(kMaxBatch) + ";loop=fused-bounce;fp=ieee-no-contract";
    return info.c_str();
}

int wfpt_loop_kind_of(const wfpt_ctx *c) {
    if (!c) return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_loop_kind_of: null context");
    if (!c->fused) return WFPT_LOOP_STAGES;
    if (c->rec_dense && !c->scene.exact) return WFPT_LOOP_REFILL;
    return use_binned(c) ? WFPT_LOOP_FUSED_BINNED : WFPT_LOOP_FUSED;
}

int wfpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *wfpt_last_error(const wfpt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

static wfpt_ctx *create_impl(const wfpt_params *params, const wfpt_sphere *spheres, const wfpt_triangle *triangles,
                             uint32_t n_spheres, const wfpt_material *materials, uint32_t n_materials,
                             const wfpt_bvh_node *nodes, uint32_t n_nodes, const wfpt_gpu_camera *camera,
                             const float inv_proj[16], const float view[16]) {
    if (!params || !(spheres || triangles) || !materials || !nodes || !camera || !inv_proj || !view || n_spheres == 0 ||
        n_materials == 0 || params->width == 0 || params->height == 0) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: null or empty argument");
        return nullptr;
    }
    if ((params->flags & WFPT_FLAG_BINNING) != 0 && params->rng_mode != WFPT_RNG_PIXEL) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: WFPT_FLAG_BINNING needs WFPT_RNG_PIXEL (the class-binned loop reorders the hit queue, which "
                                                 "shade.wgsl:72's RNG key, the dispatch's thread index, does not allow)");
        return nullptr;
    }
    if (params->max_wavefronts == 0 || params->max_wavefronts > static_cast<uint32_t>(kMaxRows)) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: max_wavefronts must be in 1..64");
        return nullptr;
    }
    if (static_cast<uint64_t>(params->width) * params->height > (1ull << 31)) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: image too large");
        return nullptr;
    }
    for (uint32_t i = 0; i < n_spheres; ++i)
        if ((spheres ? spheres[i].material_idx : triangles[i].material_idx) >= n_materials) {
            fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: primitive material_idx out of range");
            return nullptr;
        }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || params->device < 0 || params->device >= n_dev) {
        fail(nullptr, WFPT_ERR_NO_DEVICE, "wfpt_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    auto *c = new wfpt_ctx();
    c->p = *params;
    c->device = params->device;
    c->tile.world = params->tile_world == 0 ? 1u : params->tile_world;
    c->tile.rank = params->tile_world == 0 ? 0u : params->tile_rank;
    if (c->tile.rank >= c->tile.world) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: tile_rank >= tile_world");
        delete c;
        return nullptr;
    }
    std::vector<uint32_t> pair_parent;
    const int bvh_depth = validate_bvh(c, nodes, n_nodes, n_spheres, pair_parent);
    if (bvh_depth < 0) {
        g_last_error = c->err;
        delete c;
        return nullptr;
    }
    auto bail = [&](hipError_t e, const char *what) -> wfpt_ctx * {
        hip_fail(c, e, what);
        g_last_error = c->err;
        wfpt_destroy(c);
        return nullptr;
    };
#define CREATE_HIP(call)                                  \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return bail(e_, #call);     \
    } while (0)

    CREATE_HIP(hipSetDevice(c->device));
    CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    set_viewport(c, params->width, params->height);
    c->pixel_capacity = std::max(c->n_pixels, c->tile.world <= 1 ? params->max_pixels : 0u);
    // ray slots: whole tiles (partial tiles carry padding lanes), rounded up to whole segments
    const uint64_t tile_slots = static_cast<uint64_t>(c->tiles_x) * c->tiles_y_local * 64u;
    uint64_t cap = std::max<uint64_t>(tile_slots, c->pixel_capacity);
    if (c->pixel_capacity > c->n_pixels) cap += 16ull * (static_cast<uint64_t>(c->pixel_capacity) / 8u + 64u) ; // room for partial tiles after a resize
    cap = (cap + kChunk - 1) / kChunk * kChunk;
    if (cap == 0) cap = kChunk;
    // The class-binned loop cuts every class's hits into work items of kChunk, so a wavefront may fill up to kBinClasses - 1 more
    // (partly filled) segments than its rays would: room for them. WFPT_RNG_PIXEL only: the order of the queue is free there
    // (include/wfpt.h, WFPT_FLAG_BINNING).
    // Opt-in (WFPT_FLAG_BINNING) since the end of round 5: rounds 4-5 ran it by default on contexts of >= 3/4 Mpixel, where it was 2-3 % ahead
    // of the thread-ordered loop; with the miss items handed out among the hit items and the compaction offsets by a DPP scan the two are level
    // at 1920x1080 (23.02 against 23.04 Grays/s), and small slabs always lost to the partly filled work items at the end of every class
    // (per rank of N = 4 / 8: 5.28 / 3.08 ms binned against 5.14 / 2.90). WFPT_FLAG_NO_BINNING is accepted and changes nothing.
    const bool want_binning = (params->flags & (WFPT_FLAG_UNFUSED | WFPT_FLAG_SPLIT_SHADE | WFPT_FLAG_NO_LDS_SCENE | WFPT_FLAG_NO_BINNING)) == 0 &&
                              params->rng_mode == WFPT_RNG_PIXEL && (params->flags & WFPT_FLAG_BINNING) != 0;
    if (want_binning && cap + kBinClasses * kChunk <= 0xffffffffull) {
        cap += kBinClasses * kChunk;
        c->bin_capable = true;
    }
    c->capacity = static_cast<uint32_t>(cap);
    c->n_chunks_max = c->capacity / kChunk;
    // the fused bounce launches keep their per-sample work-item counts as u16 in LDS: larger images stay on the stage kernels
    c->fused = (params->flags & (WFPT_FLAG_UNFUSED | WFPT_FLAG_SPLIT_SHADE)) == 0 && c->n_chunks_max <= 65535u;
    c->bin_capable = c->bin_capable && c->fused;
    c->batch_max = params->batch == 0 ? 16u : std::min<uint32_t>(params->batch, c->fused ? kMaxBatch : kMaxBatchClassic);
    c->classic_batch = c->fused ? 1u : c->batch_max; // the stage API works on slice 0 only
    const size_t nb_all = c->batch_max;
    const size_t nb = c->classic_batch;
    // The kernels keep slice strides as 32-bit element counts (Stride32: one scalar register each): 7 * capacity floats between
    // ray-queue slices, 4 * pixel_capacity floats between image slices. Refuse a context whose strides would wrap.
    if (7ull * cap > 0xffffffffull || 4ull * c->pixel_capacity > 0xffffffffull) {
        fail(nullptr, WFPT_ERR_UNSUPPORTED, "wfpt_create: ray capacity beyond 2^32 / 7 slots (or 2^30 pixels): slice strides are 32-bit");
        wfpt_destroy(c);
        return nullptr;
    }

    for (int k = 0; k < 2; ++k) {
        CREATE_HIP(dmalloc(&c->ray_mem[k], nb * 7 * static_cast<size_t>(c->capacity)));
        CREATE_HIP(hipMemsetAsync(c->ray_mem[k], 0, sizeof(float) * nb * 7 * static_cast<size_t>(c->capacity), c->stream));
        set_queue(c->q[k], c->ray_mem[k], c->capacity);
    }
    CREATE_HIP(dmalloc(&c->hit_mem, 3 * nb * c->capacity));
    CREATE_HIP(dmalloc(&c->hit_rec, 2 * nb * c->capacity));
    CREATE_HIP(dmalloc(&c->miss_mem, 3 * nb * c->capacity));
    if (nb * c->capacity > 0xffffffffull || nb_all * c->capacity > 0xffffffffull) {
        fail(nullptr, WFPT_ERR_UNSUPPORTED, "wfpt_create: samples in flight x ray capacity beyond 2^32 queue slots");
        wfpt_destroy(c);
        return nullptr;
    }
    c->hq.base = c->hit_mem; c->hq.plane = nb * c->capacity;
    c->mq.base = c->miss_mem; c->mq.plane = nb * c->capacity;
    const size_t n_counts = nb_all * c->n_chunks_max; // the scan's bases serve both paths
    CREATE_HIP(dmalloc(&c->chunk_hits, n_counts));
    CREATE_HIP(dmalloc(&c->chunk_miss, n_counts));
    CREATE_HIP(dmalloc(&c->chunk_hit_base, n_counts));
    CREATE_HIP(dmalloc(&c->chunk_miss_base, n_counts));
    CREATE_HIP(hipMemsetAsync(c->chunk_hits, 0, sizeof(uint32_t) * n_counts, c->stream));
    CREATE_HIP(hipMemsetAsync(c->chunk_miss, 0, sizeof(uint32_t) * n_counts, c->stream));
    CREATE_HIP(hipMemsetAsync(c->chunk_hit_base, 0, sizeof(uint32_t) * n_counts, c->stream));
    CREATE_HIP(hipMemsetAsync(c->chunk_miss_base, 0, sizeof(uint32_t) * n_counts, c->stream));
    CREATE_HIP(dmalloc(&c->mat_list, 3 * nb * c->capacity));
    CREATE_HIP(dmalloc(&c->chunk_mat, 3 * n_counts));
    CREATE_HIP(hipMemsetAsync(c->chunk_mat, 0, sizeof(uint32_t) * 3 * n_counts, c->stream));
    c->image_floats = 4 * static_cast<size_t>(c->pixel_capacity); // one float4 per pixel (kernels: pixel_of)
    c->acc_floats = (3 * static_cast<size_t>(c->pixel_capacity) + 7) / 4 * 4;
    CREATE_HIP(dmalloc(&c->image, nb_all * c->image_floats));
    CREATE_HIP(dmalloc(&c->accumulated, c->acc_floats));
    CREATE_HIP(launch_fill(c->image, 1.0f, nb_all * c->image_floats, c->stream));                        // pt:53-58
    if (c->fused) {
        const size_t slots = nb_all * c->capacity, counts = nb_all * c->n_chunks_max;
        for (int k = 0; k < 2; ++k) {
            CREATE_HIP(dmalloc(&c->rec_mem[k], 2 * slots));
            CREATE_HIP(dmalloc(&c->f_miss_mem[k], 3 * slots));
            c->f_mq[k].base = c->f_miss_mem[k]; c->f_mq[k].plane = slots;
            CREATE_HIP(dmalloc(&c->f_chunk_hits[k], counts));
            CREATE_HIP(dmalloc(&c->f_chunk_miss[k], counts));
            CREATE_HIP(hipMemsetAsync(c->f_chunk_hits[k], 0, sizeof(uint32_t) * counts, c->stream));
            CREATE_HIP(hipMemsetAsync(c->f_chunk_miss[k], 0, sizeof(uint32_t) * counts, c->stream));
        }
        CREATE_HIP(dmalloc(&c->first_seg, counts));
        CREATE_HIP(hipMemsetAsync(c->first_seg, 0, sizeof(uint32_t) * counts, c->stream));
        if (c->bin_capable) {
            constexpr size_t kWords = ClsPack<kBinClasses>::kWords;
            for (int k = 0; k < 2; ++k) {
                CREATE_HIP(dmalloc(&c->f_cls[k], counts * kWords));
                CREATE_HIP(hipMemsetAsync(c->f_cls[k], 0, sizeof(uint32_t) * counts * kWords, c->stream));
            }
            CREATE_HIP(dmalloc(&c->cls_table, counts * kBinClasses));
            CREATE_HIP(hipMemsetAsync(c->cls_table, 0, sizeof(uint2) * counts * kBinClasses, c->stream));
            CREATE_HIP(dmalloc(&c->first_seg_cls, counts * kBinClasses));
            CREATE_HIP(hipMemsetAsync(c->first_seg_cls, 0, sizeof(uint32_t) * counts * kBinClasses, c->stream));
            CREATE_HIP(dmalloc(&c->plan, kPlanWords));
            CREATE_HIP(hipMemsetAsync(c->plan, 0, sizeof(uint32_t) * kPlanWords, c->stream));
        }
    }
    CREATE_HIP(hipMemsetAsync(c->accumulated, 0, sizeof(float) * c->acc_floats, c->stream));         // pt:60-65
    CREATE_HIP(dmalloc(&c->ctl, kMaxBatch));
    CREATE_HIP(hipMemsetAsync(c->ctl, 0, sizeof(Control) * kMaxBatch, c->stream));
    CREATE_HIP(dmalloc(&c->camera, 1));
    CREATE_HIP(dmalloc(&c->d_stamps, 48));
    CREATE_HIP(hipMemsetAsync(c->d_stamps, 0, sizeof(unsigned long long) * 48, c->stream));

    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, c->device));
    c->cus = static_cast<uint32_t>(prop.multiProcessorCount);
    const uint32_t cus = c->cus;
    // scene upload (pt:120-128) plus the traversal's own copies of it
    c->h_camera = *camera;
    if (upload_scene(c, spheres, triangles, n_spheres, materials, n_materials, nodes, n_nodes, pair_parent, static_cast<uint32_t>(bvh_depth),
                     camera) != WFPT_OK) {
        g_last_error = c->err;
        wfpt_destroy(c);
        return nullptr;
    }

    CameraDev cam{};
    cam.cam = *camera;
    std::memcpy(cam.inv_proj, inv_proj, sizeof cam.inv_proj);
    std::memcpy(cam.view, view, sizeof cam.view);
    CREATE_HIP(hipMemcpy(c->camera, &cam, sizeof cam, hipMemcpyHostToDevice));
    const wfpt_frame_buffer f{c->width, c->height, 0u, 0u};
    CREATE_HIP(hipMemcpy(&c->ctl->frame, &f, sizeof f, hipMemcpyHostToDevice));

    c->accumulate_grid = std::min<uint32_t>((c->pixel_capacity + 255u) / 256u, cus * 32u); // one thread per pixel
    if (c->accumulate_grid == 0) c->accumulate_grid = 1;
    CREATE_HIP(hipStreamSynchronize(c->stream));
#undef CREATE_HIP
    return c;
}

wfpt_ctx *wfpt_create(const wfpt_params *params, const wfpt_sphere *spheres, uint32_t n_spheres,
                      const wfpt_material *materials, uint32_t n_materials, const wfpt_bvh_node *nodes,
                      uint32_t n_nodes, const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16]) {
    if (!spheres) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create: null or empty argument");
        return nullptr;
    }
    return create_impl(params, spheres, nullptr, n_spheres, materials, n_materials, nodes, n_nodes, camera, inv_proj, view);
}

wfpt_ctx *wfpt_create_mesh(const wfpt_params *params, const wfpt_triangle *triangles, uint32_t n_triangles,
                           const wfpt_material *materials, uint32_t n_materials, const wfpt_bvh_node *nodes,
                           uint32_t n_nodes, const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16]) {
    if (!triangles) {
        fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_create_mesh: null or empty argument");
        return nullptr;
    }
    return create_impl(params, nullptr, triangles, n_triangles, materials, n_materials, nodes, n_nodes, camera, inv_proj, view);
}

// Image chunking (README.md:20 "split rendering of image into chunks so that the buffers aren't so big"): the band cut of
// the multi-GPU path on ONE device. Chunk k of `chunks` is a context with tile_rank = k, tile_world = chunks, so every
// queue and image buffer is sized for 1/chunks of the frame; the chunks render one after the other and their slabs are
// interleaved into the caller's frame.
static int render_chunked_impl(const wfpt_params *params, const wfpt_sphere *spheres, const wfpt_triangle *triangles, uint32_t n_prims,
                               const wfpt_material *materials, uint32_t n_materials, const wfpt_bvh_node *nodes, uint32_t n_nodes,
                               const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16], uint32_t n_samples,
                               uint32_t chunks, float *rgb) {
    if (!params || !rgb || chunks == 0) return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_render_chunked: null or empty argument");
    if (chunks > 1 && params->rng_mode != WFPT_RNG_PIXEL)
        return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT,
                    "wfpt_render_chunked: WFPT_RNG_DISPATCH keys shade's RNG on a ray's queue position, which depends on the cut; "
                    "use WFPT_RNG_PIXEL to get the unchunked image");
    const uint32_t w = params->width, h = params->height;
    const size_t band_floats = 8u * static_cast<size_t>(w) * 3u;
    std::vector<float> slab;
    for (uint32_t k = 0; k < chunks; ++k) {
        wfpt_params p = *params;
        p.tile_rank = k;
        p.tile_world = chunks;
        p.max_pixels = 0;
        if (bands_of(h, k, chunks) == 0) continue; // more chunks than bands: nothing to render for this one
        g_last_status = WFPT_ERR_HIP;
        wfpt_ctx *c = create_impl(&p, spheres, triangles, n_prims, materials, n_materials, nodes, n_nodes, camera, inv_proj, view);
        if (!c) return g_last_status; // the status create_impl failed with; wfpt_last_error(NULL) holds the reason
        int r = wfpt_render(c, n_samples);
        if (r == WFPT_OK) {
            slab.resize(3 * static_cast<size_t>(c->n_pixels));
            r = wfpt_read_accumulated(c, slab.data(), slab.size());
        }
        if (r != WFPT_OK) {
            g_last_error = c->err;
            wfpt_destroy(c);
            return r;
        }
        const uint32_t nb = bands_of(h, k, chunks);
        for (uint32_t j = 0; j < nb; ++j) { // band j of this chunk is band j * chunks + k of the frame (the last one may be partial)
            const uint32_t y0 = (j * chunks + k) * 8u, rows = std::min(8u, h - y0);
            std::memcpy(rgb + static_cast<size_t>(y0) * w * 3u, slab.data() + j * band_floats, sizeof(float) * rows * w * 3u);
        }
        wfpt_destroy(c);
    }
    return WFPT_OK;
}

int wfpt_render_chunked(const wfpt_params *params, const wfpt_sphere *spheres, uint32_t n_spheres, const wfpt_material *materials,
                        uint32_t n_materials, const wfpt_bvh_node *nodes, uint32_t n_nodes, const wfpt_gpu_camera *camera,
                        const float inv_proj[16], const float view[16], uint32_t n_samples, uint32_t chunks, float *rgb) {
    if (!spheres) return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_render_chunked: null argument");
    return render_chunked_impl(params, spheres, nullptr, n_spheres, materials, n_materials, nodes, n_nodes, camera, inv_proj, view, n_samples, chunks, rgb);
}

int wfpt_render_chunked_mesh(const wfpt_params *params, const wfpt_triangle *triangles, uint32_t n_triangles, const wfpt_material *materials,
                             uint32_t n_materials, const wfpt_bvh_node *nodes, uint32_t n_nodes, const wfpt_gpu_camera *camera,
                             const float inv_proj[16], const float view[16], uint32_t n_samples, uint32_t chunks, float *rgb) {
    if (!triangles) return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_render_chunked_mesh: null argument");
    return render_chunked_impl(params, nullptr, triangles, n_triangles, materials, n_materials, nodes, n_nodes, camera, inv_proj, view, n_samples, chunks, rgb);
}

void wfpt_destroy(wfpt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)wfpt_comm_destroy(c);
    destroy_graph(c);
    for (auto &t : c->timers) {
        if (t.start) (void)hipEventDestroy(t.start);
        if (t.stop) (void)hipEventDestroy(t.stop);
    }
    for (auto e : c->sample_events) (void)hipEventDestroy(e);
    free_scene(c);
    void *bufs[] = {c->rec_dense, c->rec_mem[0], c->rec_mem[1], c->f_miss_mem[0], c->f_miss_mem[1], c->f_chunk_hits[0], c->f_chunk_hits[1],
                    c->f_chunk_miss[0], c->f_chunk_miss[1], c->first_seg, c->f_cls[0], c->f_cls[1], c->first_seg_cls, c->plan, c->cls_table,
                    c->ray_mem[0], c->ray_mem[1], c->hit_mem, c->hit_rec, c->miss_mem, c->chunk_hits,
                    c->chunk_miss, c->chunk_hit_base, c->chunk_miss_base, c->mat_list, c->chunk_mat, c->image, c->accumulated, c->ctl,
                    c->camera, c->d_stamps};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int wfpt_set_frame(wfpt_ctx *c, const wfpt_frame_buffer *frame) {
    if (!c || !frame) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_set_frame: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipMemcpyAsync(&c->ctl->frame, frame, sizeof *frame, hipMemcpyHostToDevice, c->stream));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    c->dev_frame_valid = false;
    return WFPT_OK;
}

int wfpt_update_render_parameters(wfpt_ctx *c, uint32_t width, uint32_t height, const wfpt_gpu_camera *camera,
                                  const float inv_proj[16], const float view[16]) {
    if (!c || !camera || !inv_proj || !view || width == 0 || height == 0)
        return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_update_render_parameters: null or empty argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    c->hit_rec_valid = false;
    wfpt_ctx probe;
    probe.tile = c->tile;
    set_viewport(&probe, width, height);
    const uint64_t slots = static_cast<uint64_t>(probe.tiles_x) * probe.tiles_y_local * 64u;
    if (probe.n_pixels > c->pixel_capacity || slots > c->capacity)
        return fail(c, WFPT_ERR_INVALID_ARGUMENT, "viewport exceeds the capacity given at wfpt_create (max_pixels)");
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    set_viewport(c, width, height);
    CameraDev cam{};
    cam.cam = *camera;
    std::memcpy(cam.inv_proj, inv_proj, sizeof cam.inv_proj);
    std::memcpy(cam.view, view, sizeof cam.view);
    WFPT_HIP(c, hipMemcpy(c->camera, &cam, sizeof cam, hipMemcpyHostToDevice));       // pt:259-272
    WFPT_HIP(c, hipMemsetAsync(c->accumulated, 0, sizeof(float) * c->acc_floats, c->stream)); // pt:248-250
    c->progress_frame = 0;      // RenderProgress::reset, pt:276
    c->accumulated_samples = 0;
    c->dev_frame_valid = false;
    destroy_graph(c); // grid shapes (and the kernel variant) are baked into the captured graph
    c->h_camera = *camera;
    float reach[3];
    camera_reach(*camera, reach);
    decide_exact(c, reach); // a camera far outside the scene leaves the conservative box test's error bound
    return alloc_gather_buffers(c);
}

// Dynamic scenes: the scene of a live context is replaced in place. The BVH is rebuilt on the
// context's device by the device builder (byte-identical to bvh.rs:147-210, so the oracle chain stays the checker), the
// traversal's derived data (shade records, conservative boxes or four-wide nodes, parent table) re-derived, and the
// accumulation reset as update_buffers does for a parameter change (pt:240-277): the next sample is frame 1 again.
static int update_scene_impl(wfpt_ctx *c, wfpt_sphere *spheres, wfpt_triangle *triangles, uint32_t n_prims, const wfpt_material *materials,
                             uint32_t n_materials, uint32_t n_bins) {
    if (!c || !(spheres || triangles) || !materials || n_prims == 0 || n_materials == 0)
        return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_update_scene: null or empty argument");
    for (uint32_t i = 0; i < n_prims; ++i)
        if ((spheres ? spheres[i].material_idx : triangles[i].material_idx) >= n_materials)
            return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_update_scene: primitive material_idx out of range");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream)); // nothing in flight may still read the old scene
    c->hit_rec_valid = false;
    std::vector<wfpt_bvh_node> nodes(2 * static_cast<size_t>(n_prims));
    uint32_t n_nodes = 0;
    const int st = spheres ? wfpt_build_bvh_device(spheres, n_prims, nodes.data(), static_cast<uint32_t>(nodes.size()), &n_nodes, c->device, nullptr)
                           : wfpt_build_bvh_triangles_device(triangles, n_prims, nodes.data(), static_cast<uint32_t>(nodes.size()), &n_nodes,
                                                             n_bins ? n_bins : 32u, c->device, nullptr);
    if (st != WFPT_OK) return fail(c, st, std::string("wfpt_update_scene: BVH build failed: ") + g_last_error);
    std::vector<uint32_t> pair_parent;
    const int depth = validate_bvh(c, nodes.data(), n_nodes, n_prims, pair_parent);
    if (depth < 0) return depth;
    destroy_graph(c); // scene pointers and launch shapes are baked into the captured graphs
    if (int r = upload_scene(c, spheres, triangles, n_prims, materials, n_materials, nodes.data(), n_nodes, pair_parent,
                             static_cast<uint32_t>(depth), &c->h_camera);
        r != WFPT_OK)
        return r;
    return wfpt_reset_progress(c);
}

int wfpt_update_scene(wfpt_ctx *c, wfpt_sphere *spheres, uint32_t n_spheres, const wfpt_material *materials, uint32_t n_materials) {
    if (!spheres) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_update_scene: null argument");
    return update_scene_impl(c, spheres, nullptr, n_spheres, materials, n_materials, 0);
}

int wfpt_update_scene_mesh(wfpt_ctx *c, wfpt_triangle *triangles, uint32_t n_triangles, const wfpt_material *materials, uint32_t n_materials,
                           uint32_t n_bins) {
    if (!triangles) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_update_scene_mesh: null argument");
    return update_scene_impl(c, nullptr, triangles, n_triangles, materials, n_materials, n_bins);
}

int wfpt_set_counters(wfpt_ctx *c, const uint32_t counters[16]) {
    if (!c || !counters) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_set_counters: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipMemcpyAsync(c->ctl->counters, counters, sizeof(uint32_t) * 16, hipMemcpyHostToDevice, c->stream));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    return WFPT_OK;
}

int wfpt_read_counters(wfpt_ctx *c, uint32_t counters[16]) {
    if (!c || !counters) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_counters: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    WFPT_HIP(c, hipMemcpy(counters, c->ctl->counters, sizeof(uint32_t) * 16, hipMemcpyDeviceToHost));
    return WFPT_OK;
}

int wfpt_reset_image(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, launch_fill(c->image, 1.0f, c->image_floats, c->stream));
    return WFPT_OK;
}

int wfpt_reset_accumulated(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipMemsetAsync(c->accumulated, 0, sizeof(float) * 3 * static_cast<size_t>(c->pixel_capacity), c->stream));
    return WFPT_OK;
}

int wfpt_reset_progress(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipMemsetAsync(c->accumulated, 0, sizeof(float) * 3 * static_cast<size_t>(c->pixel_capacity), c->stream)); // pt:248-250
    c->progress_frame = 0; // RenderProgress::reset, pt:276
    c->accumulated_samples = 0;
    c->dev_frame_valid = false;
    return WFPT_OK;
}

int wfpt_clear_ray_queues(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    c->hit_rec_valid = false;
    WFPT_HIP(c, hipSetDevice(c->device));
    for (int k = 0; k < 2; ++k)
        WFPT_HIP(c, hipMemsetAsync(c->ray_mem[k], 0, sizeof(float) * 7 * static_cast<size_t>(c->capacity), c->stream));
    return WFPT_OK; // slice 0 is the stage API's ray_buffer / extension_ray_buffer
}

int wfpt_swap_ray_queues(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    c->hit_rec_valid = false; // shade.wgsl:76-78 reads ray_buffer[hit.ray_idx] as it is when shade runs: after a swap, not extend's rays
    c->cur ^= 1;
    return WFPT_OK;
}

int wfpt_kernel_run(wfpt_ctx *c, int stage, uint32_t gx, uint32_t gy) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    if (stage < 0 || stage >= WFPT_STAGE_SCAN) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_kernel_run: unknown stage");
    const uint64_t threads64 = static_cast<uint64_t>(gx) * gy * 64u;
    if (threads64 == 0) return WFPT_OK; // an empty dispatch
    const uint32_t threads = threads64 > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(threads64);
    WFPT_HIP(c, hipSetDevice(c->device));
    if (stage == WFPT_STAGE_GENERATE_RAYS) { // argument checks come before the start event is recorded
        if (threads64 > c->capacity)
            return fail(c, WFPT_ERR_INVALID_ARGUMENT, "generate_rays dispatch exceeds the ray buffer (max_pixels)");
        // The literal dispatch writes pixel_idx = x + y * (8 gx) (gr:55-57), and shade / miss_kernel index the image with
        // it: the reference relies on monitor-sized buffers and robust buffer access, here the dispatch must stay inside
        // the image allocation (e.g. ceil(100/8)^2 tiles on a 100x100 context without max_pixels is refused).
        if (threads64 > c->pixel_capacity)
            return fail(c, WFPT_ERR_INVALID_ARGUMENT, "generate_rays dispatch exceeds the image buffer: 8*gx x 8*gy pixels > max_pixels");
        if (static_cast<uint64_t>(gx) * 8u * gy * 8u * (c->tile.world) > (1ull << 32))
            return fail(c, WFPT_ERR_INVALID_ARGUMENT, "generate_rays dispatch too large");
    }
    if (int r = stage_begin(c, stage); r != WFPT_OK) return r;
    switch (stage) {
    case WFPT_STAGE_GENERATE_RAYS:
        c->hit_rec_valid = false;
        WFPT_HIP(c, launch_generate(generate_args(c, gx, gy, false), c->stream));
        break;
    case WFPT_STAGE_EXTEND:
        WFPT_HIP(c, launch_extend(extend_args(c, c->cur, &c->ctl->counters[2], threads), extend_grid(c, 1), c->stream));
        WFPT_HIP(c, launch_scan(scan_args(c, &c->ctl->counters[2], threads, false, 0), c->stream));
        c->hit_rec_valid = true; // shade may stream extend's path records until the host touches the ray queue (generate_rays, write, swap, clear)
        break;
    case WFPT_STAGE_SHADE:
        WFPT_HIP(c, launch_shade(shade_args(c, c->cur, &c->ctl->counters[1], threads, gx, 0xffffffffu, true),
                                 consumer_grid(c, 1), c->stream));
        break;
    case WFPT_STAGE_SHADE_LAMBERTIAN:
    case WFPT_STAGE_SHADE_METAL:
    case WFPT_STAGE_SHADE_DIELECTRIC:
        WFPT_HIP(c, launch_shade(shade_args(c, c->cur, &c->ctl->counters[1], threads, gx,
                                            static_cast<uint32_t>(stage - WFPT_STAGE_SHADE_LAMBERTIAN), true),
                                 consumer_grid(c, 1), c->stream));
        break;
    case WFPT_STAGE_MISS:
        WFPT_HIP(c, launch_miss(miss_args(c, c->cur, &c->ctl->counters[0], threads), consumer_grid(c, 1), c->stream));
        break;
    case WFPT_STAGE_ACCUMULATE:
        WFPT_HIP(c, launch_accumulate(accumulate_args(c, threads, false), c->accumulate_grid, c->stream));
        break;
    default: break;
    }
    return stage_end(c, stage);
}

float wfpt_kernel_timing_us(wfpt_ctx *c, int stage) {
    if (!c || stage < 0 || stage >= WFPT_STAGE_COUNT) return 0.0f;
    StageTimer &t = c->timers[stage];
    if (t.pending) {
        (void)hipSetDevice(c->device);
        if (hipEventSynchronize(t.stop) == hipSuccess) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
                if (t.window.size() == 10) t.window.pop_back(); // query_gpu.rs:33-38
                t.window.push_front(static_cast<double>(ms) * 1000.0);
            }
        }
        t.pending = false;
    }
    if (t.window.empty()) return 0.0f;
    double sum = 0.0;
    for (double v : t.window) sum += v;
    return static_cast<float>(sum / static_cast<double>(t.window.size()));
}

int wfpt_render_sample(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    return render_batch(c, 1);
}

int wfpt_render(wfpt_ctx *c, uint32_t n_samples) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    return render_many(c, n_samples);
}

int wfpt_render_timed(wfpt_ctx *c, uint32_t n_samples, float *stage_ms, uint32_t *stage_launches) {
    if (!c || !stage_ms) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_render_timed: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    while (n_samples > 0) {
        const uint32_t nb = std::min(n_samples, c->batch_max); // same batching as wfpt_render
        if (int r = ensure_device_frame(c); r != WFPT_OK) return r;
        std::vector<EventRec> ev;
        if (int r = enqueue_batch(c, &ev, nb); r != WFPT_OK) return r;
        WFPT_HIP(c, hipStreamSynchronize(c->stream));
        for (const EventRec &e : ev) {
            float ms = 0.0f;
            WFPT_HIP(c, hipEventElapsedTime(&ms, e.start, e.stop));
            stage_ms[e.stage] += ms;
            if (stage_launches) stage_launches[e.stage] += 1;
        }
        c->cur = static_cast<int>(c->p.max_wavefronts & 1u);
        c->hit_rec_valid = false;
        c->progress_frame += nb;
        c->accumulated_samples += nb;
        c->last_slot = nb - 1;
        n_samples -= nb;
    }
    return WFPT_OK;
}

int wfpt_render_sample_timed(wfpt_ctx *c, float *stage_ms, uint32_t *stage_launches) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    const uint32_t keep = c->batch_max;
    c->batch_max = 1;
    const int r = wfpt_render_timed(c, 1, stage_ms, stage_launches);
    c->batch_max = keep;
    return r;
}

int wfpt_synchronize(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    return WFPT_OK;
}

uint32_t wfpt_frame(const wfpt_ctx *c) { return c ? c->progress_frame : 0; }
uint32_t wfpt_accumulated_samples(const wfpt_ctx *c) { return c ? c->accumulated_samples : 0; }
float wfpt_progress(const wfpt_ctx *c, uint32_t spp) {
    return (c && spp) ? static_cast<float>(c->accumulated_samples) / static_cast<float>(spp) : 0.0f;
}
uint32_t wfpt_n_pixels(const wfpt_ctx *c) { return c ? c->n_pixels : 0; }
uint32_t wfpt_ray_capacity(const wfpt_ctx *c) { return c ? c->capacity : 0; }

int wfpt_read_accumulated(wfpt_ctx *c, float *rgb, size_t n_floats) {
    if (!c || !rgb) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_accumulated: null argument");
    if (n_floats > 3 * static_cast<size_t>(c->n_pixels)) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "n_floats exceeds the image");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    WFPT_HIP(c, hipMemcpy(rgb, c->accumulated, sizeof(float) * n_floats, hipMemcpyDeviceToHost));
    return WFPT_OK;
}

int wfpt_read_image(wfpt_ctx *c, float *rgb, size_t n_floats) {
    if (!c || !rgb) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_image: null argument");
    if (n_floats > 3 * static_cast<size_t>(c->n_pixels)) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "n_floats exceeds the image");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    // the device keeps one float4 per pixel; the reference layout (stride 12, sh:6-10) is what crosses the ABI
    const size_t n_px = (n_floats + 2) / 3;
    std::vector<float> tmp(4 * n_px);
    WFPT_HIP(c, hipMemcpy(tmp.data(), c->image, sizeof(float) * tmp.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_floats; ++i) rgb[i] = tmp[4 * (i / 3) + i % 3];
    return WFPT_OK;
}

int wfpt_copy_accumulated_to_device(wfpt_ctx *c, void *device_ptr, size_t n_bytes) {
    if (!c || !device_ptr) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_copy_accumulated_to_device: null argument");
    if (n_bytes > sizeof(float) * 3 * static_cast<size_t>(c->n_pixels))
        return fail(c, WFPT_ERR_INVALID_ARGUMENT, "n_bytes exceeds the image");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipMemcpyAsync(device_ptr, c->accumulated, n_bytes, hipMemcpyDeviceToDevice, c->stream));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    return WFPT_OK;
}

static int read_ray_queue(wfpt_ctx *c, int qi, wfpt_ray *rays, uint32_t n) {
    if (!c || !rays) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "read rays: null argument");
    if (n > c->capacity) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "read rays: n exceeds the queue capacity");
    if (n == 0) return WFPT_OK;
    WFPT_HIP(c, hipSetDevice(c->device));
    wfpt_ray *tmp = nullptr;
    WFPT_HIP(c, dmalloc(&tmp, n));
    hipError_t e = launch_rays_to_aos(c->q[qi], tmp, n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(rays, tmp, sizeof(wfpt_ray) * n, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(c, e, "read rays");
    return WFPT_OK;
}

int wfpt_read_rays(wfpt_ctx *c, wfpt_ray *rays, uint32_t n) { return read_ray_queue(c, c ? c->cur : 0, rays, n); }
int wfpt_read_extension_rays(wfpt_ctx *c, wfpt_ray *rays, uint32_t n) {
    return read_ray_queue(c, c ? (c->cur ^ 1) : 0, rays, n);
}

int wfpt_write_rays(wfpt_ctx *c, const wfpt_ray *rays, uint32_t n) {
    if (!c || !rays) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_write_rays: null argument");
    if (n > c->capacity) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_write_rays: n exceeds the queue capacity");
    c->hit_rec_valid = false;
    if (n == 0) return WFPT_OK;
    // shade and miss_kernel index the image with the ray's pixel_idx: it must name a pixel this context holds
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t px = rays[i].pixel_idx;
        if (px == WFPT_INACTIVE_PIXEL) continue;
        bool ok = px < c->pixel_capacity;
        if (c->tile.world > 1) ok = px < c->width * c->height && ((px / c->width) >> 3) % c->tile.world == c->tile.rank;
        if (!ok) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_write_rays: pixel_idx outside the image this context holds");
        // an origin beyond the range the conservative box test's margin was sized for (or not a number): the stage API's
        // extend switches to the reference's own box test for this context
        for (int ax = 0; ax < 3 && c->ch_ok && !c->far_rays; ++ax)
            if (!(std::fabs(rays[i].origin[ax]) <= 4.0f * c->extent[ax])) c->far_rays = true;
    }
    if (c->far_rays && !c->scene.exact) {
        c->scene.exact = 1u;
        destroy_graph(c);
    }
    WFPT_HIP(c, hipSetDevice(c->device));
    wfpt_ray *tmp = nullptr;
    WFPT_HIP(c, dmalloc(&tmp, n));
    hipError_t e = hipMemcpy(tmp, rays, sizeof(wfpt_ray) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_rays_from_aos(c->q[c->cur], tmp, n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(c, e, "wfpt_write_rays");
    return WFPT_OK;
}

int wfpt_read_hits(wfpt_ctx *c, wfpt_hit_payload *out, uint32_t n) {
    if (!c || !out) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_hits: null argument");
    return walk_segments(c, true, n, [&](uint32_t pos, uint32_t ridx, float t, uint32_t prim) {
        out[pos].t = t;
        out[pos].ray_idx = ridx;
        out[pos].sphere_idx = prim;
        out[pos].mat_type = prim < c->h_prim_mat_type.size() ? c->h_prim_mat_type[prim] : 0u; // ex:199
    });
}

int wfpt_read_misses(wfpt_ctx *c, uint32_t *out, uint32_t n) {
    if (!c || !out) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_misses: null argument");
    return walk_segments(c, false, n, [&](uint32_t pos, uint32_t ridx, float, uint32_t) { out[pos] = ridx; });
}

int wfpt_read_bounce_table(wfpt_ctx *c, uint32_t *rows4, uint32_t max_rows, uint32_t *n_rows) {
    if (!c || !rows4 || !n_rows) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_bounce_table: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    Control ctl;
    WFPT_HIP(c, hipMemcpy(&ctl, c->ctl + c->last_slot, sizeof ctl, hipMemcpyDeviceToHost));
    // rows of wavefronts whose extend really ran (the device keeps writing empty rows after the loop exits)
    uint32_t n = 0;
    const uint32_t rows = std::min<uint32_t>(ctl.bounce, kMaxRows);
    for (uint32_t b = 0; b < rows && n < max_rows; ++b) {
        if (ctl.rows[b][0] == 0) break;
        std::memcpy(rows4 + 4 * n, ctl.rows[b], sizeof(uint32_t) * 4);
        ++n;
    }
    *n_rows = n;
    return WFPT_OK;
}

int wfpt_read_totals(wfpt_ctx *c, uint64_t totals[3]) {
    if (!c || !totals) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_totals: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    Control ctl;
    WFPT_HIP(c, hipMemcpy(&ctl, c->ctl, sizeof ctl, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) totals[k] = ctl.totals[k];
    return WFPT_OK;
}

int wfpt_read_wavefront_totals(wfpt_ctx *c, uint64_t *rows3, uint32_t max_rows, uint32_t *n_rows) {
    if (!c || !rows3 || !n_rows) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_wavefront_totals: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    Control ctl;
    WFPT_HIP(c, hipMemcpy(&ctl, c->ctl, sizeof ctl, hipMemcpyDeviceToHost));
    uint32_t n = 0;
    for (uint32_t b = 0; b < static_cast<uint32_t>(kMaxRows) && b < c->p.max_wavefronts && n < max_rows; ++b, ++n)
        for (int k = 0; k < 3; ++k) rows3[3 * n + k] = ctl.wave_totals[b][k];
    *n_rows = n;
    return WFPT_OK;
}

int wfpt_device_info(int device, uint32_t *compute_units, uint32_t *memory_clock_khz, uint32_t *memory_bus_width_bits,
                     uint64_t *total_memory_bytes) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return fail(nullptr, WFPT_ERR_NO_DEVICE, "wfpt_device_info: no such HIP device");
    hipDeviceProp_t prop;
    WFPT_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (compute_units) *compute_units = static_cast<uint32_t>(prop.multiProcessorCount);
    if (memory_clock_khz) *memory_clock_khz = static_cast<uint32_t>(prop.memoryClockRate);
    if (memory_bus_width_bits) *memory_bus_width_bits = static_cast<uint32_t>(prop.memoryBusWidth);
    if (total_memory_bytes) *total_memory_bytes = static_cast<uint64_t>(prop.totalGlobalMem);
    return WFPT_OK;
}

// ------------------------------------------------------------------ RCCL gather
// RCCL is opened at run time, so single-GPU users of libwfpt.so carry no dependency on it and a host process that has
// already loaded a RCCL (e.g. PyTorch's bundled copy) shares that one instead of mapping a second.
extern "C++" {
namespace {
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl &rccl() {
    static Rccl r = [] {
        Rccl l;
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !l.handle; ++pass) // first a copy the process already holds, then a fresh load
            for (const char *n : names)
                if (!l.handle) l.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (!l.handle) { l.error = std::string("cannot open librccl.so: ") + dlerror(); return l; }
#define WFPT_RCCL_SYM(field, sym)                                                     \
        l.field = reinterpret_cast<decltype(l.field)>(dlsym(l.handle, #sym));          \
        if (!l.field && l.error.empty()) l.error = "librccl.so lacks " #sym;
        WFPT_RCCL_SYM(GetUniqueId, ncclGetUniqueId)
        WFPT_RCCL_SYM(CommInitRank, ncclCommInitRank)
        WFPT_RCCL_SYM(CommDestroy, ncclCommDestroy)
        WFPT_RCCL_SYM(GroupStart, ncclGroupStart)
        WFPT_RCCL_SYM(GroupEnd, ncclGroupEnd)
        WFPT_RCCL_SYM(Send, ncclSend)
        WFPT_RCCL_SYM(Recv, ncclRecv)
        WFPT_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef WFPT_RCCL_SYM
        return l;
    }();
    return r;
}
int rccl_fail(wfpt_ctx *c, ncclResult_t r, const char *what) {
    return fail(c, WFPT_ERR_HIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}
#define WFPT_RCCL(c, call)                                         \
    do {                                                           \
        ncclResult_t r_ = (call);                                  \
        if (r_ != ncclSuccess) return rccl_fail(c, r_, #call);     \
    } while (0)

} // namespace
} // extern "C++"

// Root side of the gather: the assembled frame (whole bands) and the staging area the peers' slabs land in, sized for the
// CURRENT viewport. Called by wfpt_comm_init and again by wfpt_update_render_parameters: a wider, shorter viewport of the same
// pixel count needs more whole-band floats than the one the communicator was created with.
static int alloc_gather_buffers(wfpt_ctx *c) {
    if (!c->comm || c->comm_rank != 0) return WFPT_OK;
    const size_t band_floats = 8u * static_cast<size_t>(c->width) * 3u;
    const size_t frame_floats = static_cast<size_t>((c->height + 7u) / 8u) * band_floats;
    size_t stage_floats = 0;
    for (int r = 1; r < c->comm_world; ++r)
        stage_floats += bands_of(c->height, static_cast<uint32_t>(r), static_cast<uint32_t>(c->comm_world)) * band_floats;
    if (frame_floats > c->gather_frame_floats || !c->gather_frame) {
        if (c->gather_frame) (void)hipFree(c->gather_frame);
        c->gather_frame = nullptr;
        c->gather_frame_floats = 0;
        WFPT_HIP(c, dmalloc(&c->gather_frame, frame_floats));
        c->gather_frame_floats = frame_floats;
    }
    if (stage_floats > c->gather_stage_floats || !c->gather_stage) {
        if (c->gather_stage) (void)hipFree(c->gather_stage);
        c->gather_stage = nullptr;
        c->gather_stage_floats = 0;
        WFPT_HIP(c, dmalloc(&c->gather_stage, stage_floats));
        c->gather_stage_floats = stage_floats;
    }
    return WFPT_OK;
}

int wfpt_comm_unique_id(void *id128) {
    if (!id128) return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_comm_unique_id: null argument");
    if (!rccl().error.empty()) return fail(nullptr, WFPT_ERR_UNSUPPORTED, rccl().error);
    static_assert(sizeof(ncclUniqueId) == WFPT_COMM_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    WFPT_RCCL(nullptr, rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return WFPT_OK;
}

int wfpt_comm_init(wfpt_ctx *c, const void *id128, int rank, int world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_comm_init: bad argument");
    if (static_cast<uint32_t>(world) != c->tile.world || static_cast<uint32_t>(rank) != c->tile.rank)
        return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_comm_init: (rank, world) must equal the context's (tile_rank, tile_world)");
    if (c->comm) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_comm_init: communicator already initialised");
    if (!rccl().error.empty()) return fail(c, WFPT_ERR_UNSUPPORTED, rccl().error);
    WFPT_HIP(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    WFPT_RCCL(c, rccl().CommInitRank(&c->comm, world, id, rank)); // collective: every rank of the job calls it
    c->comm_rank = rank;
    c->comm_world = world;
    return alloc_gather_buffers(c); // the root assembles whole bands; peers' slabs land in a staging area first
}

int wfpt_comm_destroy(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
    }
    if (c->gather_stage) (void)hipFree(c->gather_stage);
    if (c->gather_frame) (void)hipFree(c->gather_frame);
    c->gather_stage = c->gather_frame = nullptr;
    c->gather_stage_floats = c->gather_frame_floats = 0;
    return WFPT_OK;
}

// One flat gather: every peer sends its slab of whole bands straight to the root (each peer has its own xGMI link to
// the root, so the transfers run side by side; a ring would make the same bytes hop link after link), the root receives
// them in one group and de-interleaves with one strided device copy per rank. Ordered after the renders on the
// context's stream; nothing touches the host.
int wfpt_gather_accumulated(wfpt_ctx *c) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "null context");
    if (!c->comm) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_gather_accumulated: call wfpt_comm_init first");
    WFPT_HIP(c, hipSetDevice(c->device));
    const uint32_t world = static_cast<uint32_t>(c->comm_world), rank = static_cast<uint32_t>(c->comm_rank);
    const size_t band_floats = 8u * static_cast<size_t>(c->width) * 3u;
    if (rank != 0) {
        const size_t count = bands_of(c->height, rank, world) * band_floats;
        if (count > c->acc_floats) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_gather_accumulated: the slab exceeds the accumulation buffer");
        if (count) WFPT_RCCL(c, rccl().Send(c->accumulated, count, ncclFloat, 0, c->comm, c->stream));
        return WFPT_OK;
    }
    {
        size_t stage_need = 0;
        for (uint32_t r = 1; r < world; ++r) stage_need += bands_of(c->height, r, world) * band_floats;
        if (static_cast<size_t>((c->height + 7u) / 8u) * band_floats > c->gather_frame_floats || stage_need > c->gather_stage_floats)
            return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_gather_accumulated: gather buffers are smaller than the current viewport needs");
    }
    if (world > 1) {
        WFPT_RCCL(c, rccl().GroupStart());
        size_t off = 0;
        for (uint32_t r = 1; r < world; ++r) {
            const size_t count = bands_of(c->height, r, world) * band_floats;
            if (count) WFPT_RCCL(c, rccl().Recv(c->gather_stage + off, count, ncclFloat, static_cast<int>(r), c->comm, c->stream));
            off += count;
        }
        WFPT_RCCL(c, rccl().GroupEnd());
    }
    // band k of the frame is band k / world of rank k % world's slab: one strided device copy per rank
    size_t off = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const uint32_t nb = bands_of(c->height, r, world);
        const float *src = r == 0 ? c->accumulated : c->gather_stage + off;
        if (r != 0) off += nb * band_floats;
        const size_t n_valid = r == 0 ? 3u * static_cast<size_t>(c->n_pixels) : nb * band_floats;
        WFPT_HIP(c, launch_band_scatter(c->gather_frame, src, n_valid, band_floats, world, r, c->stream));
    }
    return WFPT_OK;
}

int wfpt_gather_accumulated_timed(wfpt_ctx *c, float *ms) {
    if (!c || !ms) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_gather_accumulated_timed: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream)); // the gather's own time, not the tail of what was queued before it
    if (int r = stage_begin(c, WFPT_STAGE_ACCUMULATE); r != WFPT_OK) return r; // (borrows that stage's event pair; nothing is pending after the synchronize)
    const int r = wfpt_gather_accumulated(c);
    StageTimer &t = c->timers[WFPT_STAGE_ACCUMULATE];
    WFPT_HIP(c, hipEventRecord(t.stop, c->stream));
    WFPT_HIP(c, hipEventSynchronize(t.stop));
    if (r != WFPT_OK) return r;
    WFPT_HIP(c, hipEventElapsedTime(ms, t.start, t.stop));
    return WFPT_OK;
}

int wfpt_read_gathered(wfpt_ctx *c, float *rgb, size_t n_floats) {
    if (!c || !rgb) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_gathered: null argument");
    if (!c->gather_frame) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_read_gathered: only the root rank of an initialised communicator holds the frame");
    if (n_floats > 3u * static_cast<size_t>(c->width) * c->height) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "n_floats exceeds the frame");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    WFPT_HIP(c, hipMemcpy(rgb, c->gather_frame, sizeof(float) * n_floats, hipMemcpyDeviceToHost));
    return WFPT_OK;
}

static int read_frame(wfpt_ctx *c, std::vector<float> &acc, const char *who) {
    if (!c) return fail(c, WFPT_ERR_INVALID_ARGUMENT, std::string(who) + ": null argument");
    if (c->tile.world > 1) return fail(c, WFPT_ERR_UNSUPPORTED, std::string(who) + ": this context holds only its own pixel bands");
    if (c->accumulated_samples == 0) return fail(c, WFPT_ERR_INVALID_ARGUMENT, std::string(who) + ": nothing accumulated yet");
    acc.resize(3 * static_cast<size_t>(c->n_pixels));
    return wfpt_read_accumulated(c, acc.data(), acc.size());
}

int wfpt_save_ppm(wfpt_ctx *c, const char *path) {
    std::vector<float> acc;
    if (int r = read_frame(c, acc, "wfpt_save_ppm"); r != WFPT_OK) return r;
    if (!path) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_save_ppm: null path");
    std::vector<uint8_t> rgb(acc.size());
    wfpt_tonemap_rgb8(acc.data(), c->n_pixels, c->accumulated_samples, rgb.data());
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(c, WFPT_ERR_INVALID_ARGUMENT, std::string("wfpt_save_ppm: cannot open ") + path);
    std::fprintf(f, "P6\n%u %u\n255\n", c->width, c->height);
    const bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok ? WFPT_OK : fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_save_ppm: short write");
}

int wfpt_save_png(wfpt_ctx *c, const char *path) {
    std::vector<float> acc;
    if (int r = read_frame(c, acc, "wfpt_save_png"); r != WFPT_OK) return r;
    if (!path) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_save_png: null path");
    std::vector<uint8_t> rgb(acc.size());
    wfpt_tonemap_rgb8(acc.data(), c->n_pixels, c->accumulated_samples, rgb.data());
    const int st = wfpt_write_png_rgb8(path, rgb.data(), c->width, c->height);
    return st == WFPT_OK ? WFPT_OK : fail(c, st, std::string("wfpt_save_png: cannot write ") + path);
}

int wfpt_save_pfm(wfpt_ctx *c, const char *path) {
    std::vector<float> acc;
    if (int r = read_frame(c, acc, "wfpt_save_pfm"); r != WFPT_OK) return r;
    if (!path) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_save_pfm: null path");
    const float inv_n = 1.0f / static_cast<float>(c->accumulated_samples);
    for (float &v : acc) v = inv_n * v; // display_shader.wgsl:50: invN * color, before the sqrt
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(c, WFPT_ERR_INVALID_ARGUMENT, std::string("wfpt_save_pfm: cannot open ") + path);
    std::fprintf(f, "PF\n%u %u\n-1.0\n", c->width, c->height); // negative scale = little-endian
    bool ok = true;
    const size_t row = 3 * static_cast<size_t>(c->width);
    for (uint32_t y = c->height; y-- > 0 && ok;) ok = std::fwrite(acc.data() + y * row, sizeof(float), row, f) == row;
    std::fclose(f);
    return ok ? WFPT_OK : fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_save_pfm: short write");
}

int wfpt_debug_read_stamps(wfpt_ctx *c, uint64_t out[16], int reset) {
    // Diagnostic builds of the library (-DWFPT_STAMPS=1, tools/build_variant.sh) add up, over the waves of the middle bounce
    // launches, the shader cycles per phase of a work item: [0] item start -> walk start (shade), [1] the walk, [2] waiting for the
    // other waves at the barrier, [3] compaction + stores, [4] wave-items, [5] live rays; [8] wave-level inner visits, [9]
    // wave-level leaf rounds, [10] lane-level inner visits. The shipped library leaves them zero.
    if (!c || !out) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_debug_read_stamps: null argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    WFPT_HIP(c, hipMemcpy(out, c->d_stamps, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost));
    if (reset) WFPT_HIP(c, hipMemset(c->d_stamps, 0, sizeof(uint64_t) * 16));
    return WFPT_OK;
}

int wfpt_debug_read_stamps_ex(wfpt_ctx *c, int which, uint64_t out[16], int reset) {
    // which = 1 / 2: the refill traversal's first / middle launches (scenes beyond LDS), per wave summed: [0] loop iterations, [1] lanes holding a
    // ray, [2] four-box visit steps, [3] lanes in them, [4] leaf rounds, [5] lanes in them, [6] refill passes, [7] lanes refilled, [8] lanes that sat
    // at a leaf through an iteration without a leaf round; shader cycles: [9] refill, [10] visit step, [11] leaf round + result, [12] whole loop;
    // [13] waves. which = 0: wfpt_debug_read_stamps.
    if (which == 0) return wfpt_debug_read_stamps(c, out, reset);
    if (!c || !out || which < 0 || which > 2) return fail(c, WFPT_ERR_INVALID_ARGUMENT, "wfpt_debug_read_stamps_ex: bad argument");
    WFPT_HIP(c, hipSetDevice(c->device));
    WFPT_HIP(c, hipStreamSynchronize(c->stream));
    WFPT_HIP(c, hipMemcpy(out, c->d_stamps + 16 * which, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost));
    if (reset) WFPT_HIP(c, hipMemset(c->d_stamps + 16 * which, 0, sizeof(uint64_t) * 16));
    return WFPT_OK;
}

int wfpt_debug_extend_blocks_per_cu(int device, uint32_t lds_bytes) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return fail(nullptr, WFPT_ERR_NO_DEVICE, "wfpt_debug_extend_blocks_per_cu: no HIP device");
    WFPT_HIP(nullptr, hipSetDevice(device));
    SceneDev sc{};
    sc.lds_scene = 1;
    sc.lds_bytes = lds_bytes;
    int blocks = 0;
    WFPT_HIP(nullptr, extend_blocks_per_cu(sc, &blocks));
    return blocks;
}

int wfpt_selftest_math(int device, int op, const float *a, const float *b, float *out, size_t n) {
    if (!a || !out) return fail(nullptr, WFPT_ERR_INVALID_ARGUMENT, "wfpt_selftest_math: null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return fail(nullptr, WFPT_ERR_NO_DEVICE, "wfpt_selftest_math: no HIP device");
    WFPT_HIP(nullptr, hipSetDevice(device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    hipError_t e = dmalloc(&da, n);
    if (e == hipSuccess) e = dmalloc(&dout, n);
    if (e == hipSuccess && b) e = dmalloc(&db, n);
    if (e == hipSuccess) e = hipMemcpy(da, a, sizeof(float) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db, b, sizeof(float) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_selftest_math(op, da, db, dout, n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(float) * n, hipMemcpyDeviceToHost);
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return hip_fail(nullptr, e, "wfpt_selftest_math");
    return WFPT_OK;
}

} // extern "C"
