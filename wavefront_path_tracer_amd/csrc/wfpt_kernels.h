// wfpt_kernels.h -- device-side data layout and kernel launchers shared by wfpt_kernels.hip (kernels)
// and wfpt_api.hip (context, C ABI). Internal; the public surface is include/wfpt.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "wfpt.h"
#include "wfpt_bvh4.h"

namespace wfpt {

// A ray queue is cut into segments of kChunk consecutive slots. One extend workgroup traces one
// segment at a time and leaves that segment's hits and misses compacted (stable) at the front of the
// matching segment of the hit / miss queues; a one-workgroup scan then turns the per-segment counts
// into global queue positions. See DESIGN.md "Queues".
#ifndef WFPT_CHUNK
#define WFPT_CHUNK 512 // rays per queue segment = threads of an extend workgroup (tuning builds: 256 / 512 / 1024)
#endif
constexpr int kChunk = WFPT_CHUNK;
constexpr int kExtendThreads = kChunk;
constexpr int kExtendWaves = kExtendThreads / 64;
#ifndef WFPT_CONSUMER_THREADS
#define WFPT_CONSUMER_THREADS 256
#endif
constexpr int kConsumerThreads = WFPT_CONSUMER_THREADS;
constexpr int kScanThreads = 1024;
constexpr int kMaxRows = 64;       // per-bounce table rows kept on the device
constexpr int kMaxTrailDepth = 63; // traversal keeps one pending bit per tree level in a u64
constexpr int kMaxBatch = 128;     // samples kept in flight by one launch of the device-resident loop (fused bounce launches)
constexpr int kClsMax = 8;         // cost classes of the class-binned fused loop (Control::cls_n)
constexpr int kMaxBatchClassic = 64; // ... by the stage kernels one by one (WFPT_FLAG_UNFUSED / WFPT_FLAG_SPLIT_SHADE): their LDS tables are per sample

// SoA ray queue: 28 B per ray (origin, direction, pixel); inverse direction is recomputed. The seven planes of a
// slice sit `cap` elements apart behind one base pointer (3 SGPRs per queue in a kernel instead of 14).
// An element count that fits 32 bits but is used in 64-bit address arithmetic: kept as ONE scalar register in the kernels' argument
// blocks (a size_t costs two, and the fused kernels spill scalar registers to vector lanes), widened where it is used.
struct Stride32 {
    uint32_t v;
    __host__ __device__ operator size_t() const { return v; }
    __host__ __device__ Stride32 &operator=(size_t x) { v = static_cast<uint32_t>(x); return *this; }
};

struct RayQueue {
    float *base;
    uint32_t cap;
    __host__ __device__ float *ox() const { return base; }
    __host__ __device__ float *oy() const { return base + cap; }
    __host__ __device__ float *oz() const { return base + 2u * static_cast<size_t>(cap); }
    __host__ __device__ float *dx() const { return base + 3u * static_cast<size_t>(cap); }
    __host__ __device__ float *dy() const { return base + 4u * static_cast<size_t>(cap); }
    __host__ __device__ float *dz() const { return base + 5u * static_cast<size_t>(cap); }
    __host__ __device__ uint32_t *pixel() const { return reinterpret_cast<uint32_t *>(base + 6u * static_cast<size_t>(cap)); }
};

// Hit queue (12 B per hit: t, primitive, ray index) and miss queue (12 B per miss: ray index, direction.y, pixel),
// segment-compacted. Three planes `plane` elements apart behind one base pointer each.
struct HitQueue {
    uint32_t *base;
    Stride32 plane; // elements between planes (= samples in flight * capacity)
    __host__ __device__ float *t() const { return reinterpret_cast<float *>(base); }
    __host__ __device__ uint32_t *prim() const { return base + static_cast<size_t>(plane); }
    __host__ __device__ uint32_t *ridx() const { return base + 2u * static_cast<size_t>(plane); }
};
struct MissQueue {
    uint32_t *base;
    Stride32 plane;
    __host__ __device__ uint32_t *ridx() const { return base; }
    __host__ __device__ float *dy() const { return reinterpret_cast<float *>(base + static_cast<size_t>(plane)); }
    __host__ __device__ uint32_t *pixel() const { return base + 2u * static_cast<size_t>(plane); }
};

// Device-resident control block. `counters` is the reference's counter_buffer (extend.wgsl:41).
struct Control {
    uint32_t counters[16];
    wfpt_frame_buffer frame;
    uint32_t n_in;       // rays the next fused extend traces
    uint32_t seg_n;      // rays the last extend traced (segments the consumers walk)
    uint32_t hits;       // totals of the last extend
    uint32_t misses;
    uint32_t shade_n;    // fused loop: hits to shade (0 once the loop has exited)
    uint32_t miss_n;     // fused loop: misses to shade
    uint32_t shade_gx;   // x extent of workgroup_size_64(hits): the dispatch shape shade.wgsl:72 keys its RNG on
    uint32_t done;       // fused loop exited (path_tracer.rs:332)
    uint32_t ticket;     // extend's dynamic segment ticket
    uint32_t bounce;     // rows written this sample
    uint32_t samples;    // fused samples accumulated
    uint32_t _pad;
    // class-binned fused loop (bounce_binned_kernel): hits of the last extend per cost class (0 once the loop has
    // exited), the segments that extend wrote, and the segments the next one will write (= its hit work items)
    uint32_t cls_n[kClsMax];
    uint32_t n_segs, next_segs, _pad2[2];
    uint32_t rows[kMaxRows][4]; // (rays_in, hits, misses, shaded) per bounce of the current sample
    unsigned long long totals[4]; // rays traced, hits, misses, samples
    unsigned long long wave_totals[kMaxRows][3]; // per wavefront, over all fused samples: rays traced, hits, misses
};

// What shade needs to know about a primitive, merged into one 48-byte record (three float4) so that one
// gather replaces the reference's dependent sphere -> material look-ups (shade.wgsl:80-84):
// v = sphere centre, or for a triangle its unit normal normalize(cross(e1, e2)) (computed once on the host
// with the same operation order the oracle uses per hit).
struct ShadeRec {
    float v[3];
    float fuzz;
    float albedo[3];
    float refract_index;
    uint32_t mat_type;
    uint32_t cost_class; // class-binned loop: 0 = the scene's dominant primitive, 1 + material class otherwise (upload_scene)
    uint32_t _pad[2];
};

struct SceneDev {
    const wfpt_bvh_node *nodes;   // reference layout, 32 B
    const float4 *prim_geom;      // spheres: (cx, cy, cz, r), 16 B each; triangles: the wfpt_triangle array, 3 x 16 B each
    const uint16_t *pair_parent;  // LDS variant: parent node of the sibling pair (2k, 2k+1); padded to 8 entries
    const uint32_t *pair_parent32; // HBM variant: same table, 32-bit
    const wfpt_sphere *spheres;   // reference layout (shade reads material_idx / material_type)
    const wfpt_triangle *triangles;
    const wfpt_material *materials;
    const float4 *shade_rec;      // ShadeRec per primitive, as 3 x float4
    uint32_t n_nodes, n_spheres, n_materials; // n_spheres = primitive count
    uint32_t prim_kind;           // 0 spheres (the reference), 1 triangles (build extension)
    uint32_t lds_scene;           // 1: nodes + primitives + parents are staged in LDS by extend
    uint32_t lds_bytes;           // dynamic LDS the extend kernel needs for this scene
    uint32_t depth;               // levels below the root (validated <= kMaxTrailDepth)
    // HBM-resident scenes: the binary tree collapsed into four-wide nodes (64-byte quantised nodes); null = walk
    // the binary tree. Stack entries beyond the LDS column spill to `stack_spill`, entry k of global thread g at
    // [k * spill_stride + g].
    const float4 *nodes4;
    uint32_t tile_n;              // nodes4[0 .. tile_n) = the top of the tree (breadth-first numbering), staged in LDS by refill_kernel
    uint32_t *stack_spill;
    uint32_t spill_stride;
    // LDS-resident scenes, default traversal: every node as (centre.xyz | left_first), (half-extent.xyz | prim_count), the
    // half-extent grown by more than the box test's rounding error (trace_ray_conservative); staged instead of `nodes`
    const float4 *nodes_ch;
    float safe_c[3], safe_r2; // free walks: origins within sqrt(safe_r2) of safe_c are covered by the margin's bound (far_origin)
    uint32_t exact; // 1: WFPT_FLAG_EXACT_TRAVERSAL (or a fallback to it): the reference's box test and 1e30 miss value
    uint32_t root_leaf; // the root is a leaf: its box is never tested (ex:84), so neither is it by the leaf-box test of the free walks
};

constexpr uint32_t kStack4Lds = 8;  // stack entries of the four-wide traversal kept in LDS per lane (0.3 % of the pushes go deeper: they spill)
constexpr uint32_t kTileNodesMax = 341; // four-wide nodes staged in LDS by the refill traversal: five full levels (1 + 4 + 16 + 64 + 256), 21.8 KB

struct CameraDev {
    wfpt_gpu_camera cam;
    float inv_proj[16];
    float view[16];
};

struct Tiling {
    uint32_t rank, world; // this context owns 8-row bands k with k % world == rank
};

// Sample batching. One launch serves `n` independent samples (frames f0, f0+1, ...): every per-sample
// buffer is an array of `n` identically laid out slices, `*_stride` elements apart. Sample s keeps its
// own queues, counts and Control block, so results are exactly those of n sequential samples.
struct Batch {
    uint32_t n;            // samples in this launch (1 for the stage API)
    uint32_t ctl_stride;   // u32 words between Control blocks
    Stride32 ray_stride;   // floats between ray-queue slices (7 * capacity)
    Stride32 queue_stride; // elements between hit / miss queue slices (capacity)
    Stride32 chunk_stride; // elements between per-segment count arrays
    Stride32 image_stride; // floats between image slices (a slice holds one float4 per pixel)
};

struct GenerateArgs {
    Batch batch;
    RayQueue q;
    float *image;            // reset to 1 when reset_image != 0
    Control *ctl;            // frame uniform; fused loop: n_in <- rays generated
    const CameraDev *camera;
    uint32_t gx, gy;         // dispatch (gy counts this rank's bands)
    uint32_t true_size;      // 0: width/height = 8*gx, 8*gy (generate_rays.wgsl:55-56); 1: from the frame uniform
    uint32_t reset_image;
    uint32_t set_n_in;       // fused loop: ctl->n_in = gx*gy*64 (pt:313-316)
    uint32_t capacity;
    Tiling tile;
};

struct ExtendArgs {
    Batch batch;
    RayQueue q;
    HitQueue hq;
    MissQueue mq;         // payload: direction.y and pixel of the missing ray, so miss_kernel needs no gather
    uint32_t *chunk_hits, *chunk_miss;
    // Beside the reference's hit queue (t, primitive, ray index: what wfpt_read_hits returns) extend leaves the path record the fused loop
    // carries -- (hit point | pixel), (incoming direction | primitive), two float4 at the hit's slot -- so that shade STREAMS 32 bytes per hit
    // instead of gathering seven planes of the ray queue through the ray index (round 5; null = not written)
    float4 *rec_out;
    Control *ctl;
    const uint32_t *n_in; // rays to trace = min(*n_in, limit)
    uint32_t limit;
    uint32_t has_inactive; // ray queue may hold WFPT_INACTIVE_PIXEL padding rays
    // per-material partition of each segment's hits (README.md:19 "split shade into by-material shade kernels"):
    // list m of segment c holds, ascending, the in-segment ranks of its hits with material_type m
    uint32_t partition;
    uint16_t *mat_list;      // [3][batch][capacity]
    uint32_t *chunk_mat;     // [3][batch][segments]
    size_t mat_list_mstride; // elements between materials
    size_t chunk_mat_mstride;
    SceneDev scene;
};

struct ScanArgs {
    Batch batch;
    const uint32_t *chunk_hits, *chunk_miss;
    uint32_t *chunk_hit_base, *chunk_miss_base;
    Control *ctl;
    const uint32_t *n_in;
    uint32_t limit;
    uint32_t *first_seg; // fused bounce kernel: first_seg[s] = segment that holds hit 512*s (may be null)
    uint32_t fused;      // 1: drive the device-resident loop; 0: stage API (counters protocol only)
    uint32_t miss_floor;
    uint32_t bounce;
};

// ---- fused bounce kernel of the device-resident loop (DESIGN.md "The loop"): one launch per wavefront does
//   shade(hit of wavefront b-1) -> extend(the extension ray, straight from registers) -> compaction,
// plus miss_kernel for wavefront b-1's misses. The wavefront's queue IS the hit queue: a *path record* is the hit
// (point, incoming direction, primitive, pixel), 32 B as two float4, segment-compacted like the hit queue.
constexpr int kBounceFirst = 0;  // generate_rays -> extend                 (wavefront 0)
constexpr int kBounceMiddle = 1; // shade -> extend, miss_kernel            (wavefronts 1 .. max-1)
constexpr int kBounceLast = 2;   // shade (throughput only), miss_kernel    (after the last extend)
#ifndef WFPT_MISS_SEGS
#define WFPT_MISS_SEGS 16
#endif
constexpr int kMissSegsPerItem = WFPT_MISS_SEGS; // miss work item = this many input segments
#ifndef WFPT_MISS_EVERY
#define WFPT_MISS_EVERY 0
#endif
constexpr uint32_t kMissEvery = WFPT_MISS_EVERY; // fused bounce launches: every kMissEvery-th ticket is a miss item while both kinds are left; 0 = hit items / miss items + 1, per launch

struct BounceArgs {
    Batch batch;
    unsigned long long *stamps; // diagnostic builds (-DWFPT_STAMPS=1): per-phase wave-cycle sums, see wfpt_debug_read_stamps
    const float4 *rec_in;   // [batch][capacity][2]: (p.xyz | pixel), (d.xyz | prim) of the previous wavefront's hits
    float4 *rec_out;
    const uint32_t *in_hits, *in_hit_base, *in_miss; // per-segment counts / bases of the previous wavefront (scan's output)
    const uint32_t *in_first_seg;                    // segment holding hit 512*s, per output segment s (scan's output)
    uint32_t *out_hits, *out_miss;                   // per-segment counts of this wavefront
    MissQueue mq_in, mq_out;                         // (dy, pixel) payload of the misses; the ridx plane is unused here
    float *image;
    Control *ctl;
    const CameraDev *camera;
    uint32_t gx, gy;       // first wavefront: tiles of this context (gy counts this rank's bands)
    uint32_t capacity;
    uint32_t rng_mode;
    uint32_t image_width;
    Tiling tile;
    SceneDev scene;
    // ---- class-binned loop (bounce_binned_kernel; WFPT_RNG_PIXEL only, where the order of the queue is free): the hits of a segment are
    // stored sorted by COST CLASS (ShadeRec::cost_class of the primitive hit), and a work item of the next launch is kChunk hits of ONE
    // class (of one sample), found through the per-class tables the scan leaves.
    const uint32_t *plan;           // [n * K + 1] first hit item of each (sample, class) | [n + 1] first segment item of each sample | [n + 1] first miss item
    uint32_t plan_seg_off, plan_miss_off;
    const uint2 *cls_table;         // in:  [n][segments][K] {class-k hits before this segment, first slot of the segment's class-k run}
    const uint32_t *first_seg_cls;  // in:  [n][K][segments]: segment that holds class-k hit number kChunk * run
    uint32_t *out_cls;              // out: [n][segments][words] per-segment class totals, packed 10 bits each (ClsPack)
};

// Packed counters of the class-binned compaction: field f of a word array sits in word f / 3 at bit 10 * (f % 3); a field holds at most
// kChunk = 512 < 1024, so packed words add without carries between fields. Fields 0 .. K-1: hits per class, K: misses, K + 1: all hits.
template <int K> struct ClsPack {
    static constexpr int kFields = K + 2, kWords = (kFields + 2) / 3;
};
constexpr int kBinClasses = 4; // classes of the binned loop: 0 = the dominant primitive (the Shirley scene's ground), 1 + material type otherwise
static_assert(kBinClasses <= kClsMax && kChunk <= 1023, "a packed field must hold a segment's count");

struct ScanBinnedArgs {
    Batch batch;
    const uint32_t *chunk_hits, *chunk_miss, *chunk_cls; // per-segment totals of this wavefront (chunk_cls: ClsPack words)
    uint2 *cls_table;
    uint32_t *first_seg_cls;
    Control *ctl;
    const uint32_t *n_in;
    uint32_t limit, miss_floor, bounce;
};

struct PlanArgs {
    Batch batch;
    const Control *ctl;
    uint32_t *plan;
    uint32_t plan_seg_off, plan_miss_off;
    uint32_t last; // 1: the next launch is the last one (shade without extend): its hit items are whole segments
};

// ---- HBM-resident scenes: traversal with dynamic lane refill. Rays of such scenes take very
// different numbers of steps (1M-triangle soup: mean 188, p99 650), so a wave that keeps its 64 rays until the longest one
// ends runs at 13-17 % lane utilisation. Here every wave is an independent worker: a lane that finishes its ray takes
// the next ray index from a global cursor (in groups, one atomic per group), sets the ray up itself (generate_rays, or
// shade of hit h of the previous wavefront) and traces it. Results go to a DENSE per-ray array; `compact_kernel` then
// builds the segment-compacted path-record / miss queues in ray order, so everything downstream (scan, RNG keying by
// queue position, the next wavefront) sees exactly the queues the fused bounce kernel would have written.
constexpr uint32_t kDenseMiss = 0xffffffffu, kDenseInactive = 0xfffffffeu; // primitive word of a dense record that is not a hit
#ifndef WFPT_REFILL_IDLE
#define WFPT_REFILL_IDLE 40
#endif
#ifndef WFPT_REFILL_IDLE_FIRST
#define WFPT_REFILL_IDLE_FIRST 16
#endif
// refill when at least this many lanes of a wave are idle: what a refill costs (shade: ~550 instructions; generate_rays of the
// first wavefront: ~250) against the idle lanes the traversal drags along until then
#ifndef WFPT_TICKET_BLOCK
#define WFPT_TICKET_BLOCK 64
#endif
constexpr uint32_t kTicketBlock = WFPT_TICKET_BLOCK; // ray indices a wave of the refill traversal reserves per atomic
#ifndef WFPT_REFILL_IDLE_PRESHADED
#define WFPT_REFILL_IDLE_PRESHADED 24
#endif
#ifndef WFPT_REFILL_IDLE_FIRST_PRE
#define WFPT_REFILL_IDLE_FIRST_PRE 16 // first wavefront with pre-generated primary rays (generate_dense_kernel)
#endif
#ifndef WFPT_PRESHADE
#define WFPT_PRESHADE 1 // middle wavefronts of the refill traversal: shade in a kernel of its own (shade_rays_kernel), rays through the dense array
#endif
constexpr uint32_t kRefillIdle = WFPT_REFILL_IDLE, kRefillIdleFirst = WFPT_REFILL_IDLE_FIRST, kRefillIdlePreshaded = WFPT_REFILL_IDLE_PRESHADED,
                   kRefillIdleFirstPre = WFPT_REFILL_IDLE_FIRST_PRE;

struct RefillArgs {
    Batch batch;
    unsigned long long *stamps; // diagnostic builds (-DWFPT_STAMPS=1): 16 counters per launch kind, see wfpt_debug_read_stamps_ex
    const float4 *rec_in;  // compact hit records of the previous wavefront
    float4 *dense_out;     // [batch][capacity][2]: (p | pixel), (d | prim or kDenseMiss / kDenseInactive), indexed by ray
    const uint32_t *in_hits, *in_hit_base, *in_first_seg;
    float *image;
    Control *ctl;
    const CameraDev *camera;
    uint32_t gx, gy, capacity, rng_mode, image_width;
    Tiling tile;
    SceneDev scene;
};

struct CompactArgs {
    Batch batch;
    const float4 *dense_in;
    float4 *rec_out;
    MissQueue mq_out;
    uint32_t *out_hits, *out_miss;
    const Control *ctl; // n_in = rays of this wavefront
    uint32_t capacity;
};

struct ShadeArgs {
    Batch batch;
    RayQueue q, ext;
    HitQueue hq;
    const uint32_t *chunk_hits, *chunk_hit_base;
    const float4 *rec_in;   // extend's path records of these hits (ExtendArgs::rec_out); null = gather the ray through hq.ridx() as shade.wgsl:76-78 does
    float *image;
    Control *ctl;
    const uint32_t *n_hits; // hits to shade = min(*n_hits, limit)
    uint32_t limit;
    uint32_t gx;            // stage API: the host's dispatch x extent; 0: use ctl->shade_gx
    uint32_t rng_mode;
    uint32_t material;      // split != 0: 0/1/2 = walk only that material's lists, 0xffffffff = blockIdx.z picks
    uint32_t split;         // walk the per-material lists instead of the whole hit queue
    const uint16_t *mat_list;
    const uint32_t *chunk_mat;
    size_t mat_list_mstride, chunk_mat_mstride;
    uint32_t count_out;     // stage API: counters[2] += rays emitted (sh:155)
    uint32_t image_width;   // for the tile mapping
    SceneDev scene;
    Tiling tile;
};

struct MissArgs {
    Batch batch;
    RayQueue q;
    MissQueue mq;
    const uint32_t *chunk_miss, *chunk_miss_base;
    float *image;
    const Control *ctl;
    const uint32_t *n_miss;
    uint32_t limit;
    uint32_t image_width;
    Tiling tile;
};

struct AccumulateArgs {
    Batch batch;
    const float *image;
    float *accumulated;
    Control *ctl;
    uint32_t n_pixels;
    uint32_t bookkeeping; // fused loop: fold the bounce table into totals, frame += 1
};

hipError_t launch_generate(const GenerateArgs &a, hipStream_t s);
hipError_t launch_extend(const ExtendArgs &a, uint32_t grid, hipStream_t s);
hipError_t launch_scan(const ScanArgs &a, hipStream_t s); // one workgroup per sample
hipError_t launch_bounce(const BounceArgs &a, int mode, uint32_t grid, hipStream_t s);
hipError_t launch_bounce_binned(const BounceArgs &a, int mode, uint32_t grid, hipStream_t s); // LDS-resident scenes only
hipError_t launch_scan_binned(const ScanBinnedArgs &a, hipStream_t s);
hipError_t launch_plan(const PlanArgs &a, hipStream_t s);
hipError_t bounce_binned_blocks_per_cu(const SceneDev &scene, int *blocks);
hipError_t launch_refill(const RefillArgs &a, int mode, uint32_t grid, hipStream_t s, bool preshaded = false);
hipError_t launch_shade_rays(const RefillArgs &a, uint32_t n_chunks, hipStream_t s);
hipError_t launch_generate_dense(const RefillArgs &a, hipStream_t s); // the first wavefront's primary rays into the dense array (WFPT_PRESHADE)
hipError_t launch_compact(const CompactArgs &a, uint32_t n_chunks, hipStream_t s);
hipError_t bounce_blocks_per_cu(const SceneDev &scene, int *blocks);
uint32_t bounce_lds_bytes(uint32_t n_nodes, uint32_t n_prims, uint32_t prim_kind, bool lds_scene);
hipError_t launch_shade(const ShadeArgs &a, uint32_t grid, hipStream_t s);
hipError_t launch_miss(const MissArgs &a, uint32_t grid, hipStream_t s);
hipError_t launch_accumulate(const AccumulateArgs &a, uint32_t grid, hipStream_t s);
hipError_t launch_fill(float *p, float v, size_t n, hipStream_t s);
hipError_t launch_set_frame(Control *ctl, const wfpt_frame_buffer &f, hipStream_t s); // ctl->frame = f, ordered on the stream
// frame band (j * world + rank) <- slab band j for the first n_valid floats of a slab: the root of the multi-GPU gather
hipError_t launch_band_scatter(float *frame, const float *slab, size_t n_valid, size_t band_floats, uint32_t world, uint32_t rank, hipStream_t s);
// AoS <-> SoA converters for the read-back / injection paths
hipError_t launch_rays_to_aos(const RayQueue &q, wfpt_ray *out, uint32_t n, hipStream_t s);
hipError_t launch_rays_from_aos(const RayQueue &q, const wfpt_ray *in, uint32_t n, hipStream_t s);
hipError_t launch_selftest_math(int op, const float *a, const float *b, float *out, size_t n, hipStream_t s);
// Occupancy of the extend kernel for a given dynamic LDS size (workgroups per CU); also raises the
// kernel's dynamic-LDS limit when the scene needs more than the default 64 KiB.
hipError_t extend_blocks_per_cu(const SceneDev &scene, int *blocks);
uint32_t extend_lds_bytes(uint32_t n_nodes, uint32_t n_prims, uint32_t prim_kind, bool lds_scene);

} // namespace wfpt
