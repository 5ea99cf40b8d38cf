// wfpt_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the wavefront chain
//   generate_rays -> extend -> (scan) -> shade -> miss_kernel -> accumulate
// Hand-written for MI355X: SoA ray queues in HBM with coalesced loads, the whole sphere BVH staged in
// LDS by persistent extend workgroups, stack-free traversal (pending-level bit trail + parent table, so
// occupancy is not limited by a per-lane stack), wave64 ballot + mbcnt compaction into queue segments.
// No MFMA: this is branchy fp32 traversal, not a contraction.
//
// Arithmetic follows wfpt_device_math.h exactly (compiled -ffp-contract=off) so results are bit-equal to
// the oracle's. Citations: gr/ex/sh/mk/ac = gpu_wavefront_pt/shaders/{generate_rays,extend,shade,
// miss_kernel,accumulate}.wgsl, pt = gpu_wavefront_pt/src/path_tracer.rs.
#include "wfpt_kernels.h"
#include "wfpt_device_math.h"

namespace wfpt {

namespace {

constexpr float kPi = 3.1415927f; // gr:2, sh:2

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// number of set bits of `mask` below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ uint32_t mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// a value every lane of the wave holds alike (read through LDS or with a uniform address): as a scalar, so that what is computed
// from it (offsets, bounds) runs on the scalar unit instead of 64 lanes
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Pointers with their address space in the TYPE, for code that picks per lane between an LDS copy and global memory: hipcc
// otherwise folds `if (in_lds) x = lds[i]; else x = global[j];` into a select of two generic pointers and ONE flat load, which
// occupies both the LDS and the vector-memory path every time. Loads through these types cannot be merged.
#define WFPT_AS_LDS __attribute__((address_space(3)))
#define WFPT_AS_GLOBAL __attribute__((address_space(1)))
typedef float v4f_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4_lds(const float4 *p) {
    const v4f_ v = *(const WFPT_AS_LDS v4f_ *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load4_global(const float4 *p) {
    const v4f_ v = *(const WFPT_AS_GLOBAL v4f_ *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// pixel index -> slot in this context's image slab (identity unless the image is sharded by bands)
__device__ __forceinline__ uint32_t local_pixel(uint32_t pixel, uint32_t width, Tiling tile) {
    if (tile.world <= 1) return pixel;
    const uint32_t y = pixel / width, x = pixel - y * width;
    const uint32_t band = y >> 3;
    return ((band / tile.world) * 8u + (y & 7u)) * width + x;
}

__device__ __forceinline__ RayQueue slice(RayQueue q, size_t off) { return {q.base + off, q.cap}; }

// The per-sample throughput image (image_buffer, sh:6-10,47) keeps one float4 per pixel on the device, (r, g, b, unused): the
// read-modify-write of shade / miss_kernel (sh:84-87, mk:35-37) is then ONE 16-byte load and ONE 16-byte store inside one
// 64-byte line, where the reference's stride-12 layout costs three dword accesses that straddle lines for a quarter of the
// pixels. The stride-12 layout is what crosses the C ABI (wfpt_read_image strips the pad; `accumulated` stays stride 12).
__device__ __forceinline__ float4 *pixel_of(float *image, uint32_t local_px) { return reinterpret_cast<float4 *>(image) + local_px; }

// WGSL mat4x4f * vec4f, m column-major: ((c0*x + c1*y) + c2*z) + c3*w per component
struct float4_ { float x, y, z, w; };
__device__ __forceinline__ float4_ mat_mul(const float *m, float4_ v) {
    float4_ r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}

// gr:107-116
__device__ __forceinline__ void rng_next_in_unit_disk(uint32_t &state, float &x, float &y) {
    const float r = sqrt_(rng_next_float(state));
    const float alpha = 2.0f * kPi * rng_next_float(state);
    float s, c;
    sincos_(alpha, s, c);
    x = r * c;
    y = r * s;
}

// sh:203-216
__device__ __forceinline__ float3_ rng_next_in_unit_sphere(uint32_t &state) {
    const float r = pow_(rng_next_float(state), 0.33333f);
    const float cos_theta = 1.0f - 2.0f * rng_next_float(state);
    const float sin_theta = sqrt_(1.0f - cos_theta * cos_theta);
    const float phi = 2.0f * kPi * rng_next_float(state);
    float s, c;
    sincos_(phi, s, c);
    return {r * sin_theta * c, r * sin_theta * s, r * cos_theta};
}

// gr:60-88 for one pixel: jittered NDC -> far-plane point -> thin-lens origin -> normalised world direction.
// Shared by generate_rays_kernel and the fused first bounce so that both produce the same bits.
struct PrimaryRay { float ox, oy, oz, dx, dy, dz; };
__device__ __forceinline__ PrimaryRay primary_ray(const CameraDev &cam, uint32_t id_x, uint32_t id_y, uint32_t width,
                                                  uint32_t height, wfpt_frame_buffer fb) {
    uint32_t rng = init_rng(id_x, id_y, width, fb.frame); // gr:60
    rng = advance(rng, fb.sample_number * 10u);           // gr:61
    float off_x, off_y;
    rng_next_in_unit_disk(rng, off_x, off_y);             // gr:63

    float ndc_x = (static_cast<float>(id_x) + off_x) / static_cast<float>(width); // gr:66
    float ndc_y = 1.0f - (static_cast<float>(id_y) + off_y) / static_cast<float>(height);
    ndc_x = 2.0f * ndc_x - 1.0f; // gr:67
    ndc_y = 2.0f * ndc_y - 1.0f;
    float4_ pp = mat_mul(cam.inv_proj, {ndc_x, ndc_y, 1.0f, 1.0f}); // gr:68
    const float pw = pp.w;
    pp = {pp.x / pw, pp.y / pw, pp.z / pw, pp.w / pw}; // gr:69

    float4_ origin = {cam.cam.position[0], cam.cam.position[1], cam.cam.position[2], cam.cam.position[3]};
    if (cam.cam.defocus_radius > 0.0f) { // gr:73-82
        rng_next_in_unit_disk(rng, off_x, off_y);
        const float R = cam.cam.defocus_radius;
        const float4_ p_lens = {R * off_x, R * off_y, R * 0.0f, 1.0f};
        float4_ lo = mat_mul(cam.view, p_lens);
        const float lw = lo.w;
        origin = {lo.x / lw, lo.y / lw, lo.z / lw, lo.w / lw};
        const float tf = cam.cam.focus_distance / pp.z;
        pp = {tf * pp.x - p_lens.x, tf * pp.y - p_lens.y, tf * pp.z - p_lens.z, tf * pp.w - p_lens.w};
    }
    const float4_ rd = mat_mul(cam.view, {pp.x, pp.y, pp.z, 0.0f}); // gr:84
    // normalize(vec4) (gr:86): v * (1 / length), length = sqrt(((x*x + y*y) + z*z) + w*w) (wfpt_device_math.h: normalize3)
    const float inv_len = 1.0f / sqrt_(((rd.x * rd.x + rd.y * rd.y) + rd.z * rd.z) + rd.w * rd.w);
    return {origin.x, origin.y, origin.z, rd.x * inv_len, rd.y * inv_len, rd.z * inv_len};
}

// ================================================================================================
// generate_rays (gr:42-91): one thread per queue slot, slot = tile*64 + local (8x8-tile order), so a
// wave is one 8x8 pixel tile and the SoA stores are fully coalesced.
// ================================================================================================
__global__ __launch_bounds__(256) void generate_rays_kernel(GenerateArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_threads = a.gx * a.gy * 64u;
    if (idx >= n_threads || idx >= a.capacity) return;
    const uint32_t sample = blockIdx.y; // batch slice; renders frame (base frame + sample)
    a.q = slice(a.q, sample * a.batch.ray_stride);
    a.image += sample * a.batch.image_stride;
    const uint32_t workgroup_index = idx >> 6, local_index = idx & 63u;
    const uint32_t wx = workgroup_index % a.gx, wy = workgroup_index / a.gx;
    const uint32_t id_x = wx * 8u + (local_index & 7u);
    const uint32_t id_y = (wy * a.tile.world + a.tile.rank) * 8u + (local_index >> 3);
    wfpt_frame_buffer fb = a.ctl->frame;
    fb.frame += sample;
    if (a.set_n_in && idx == 0) a.ctl[sample].n_in = n_threads; // pt:313-316: counter[2] = rays for the first extend
    const uint32_t width = a.true_size ? fb.width : a.gx * 8u;   // gr:55-56
    const uint32_t height = a.true_size ? fb.height : a.gy * 8u;

    if (a.true_size && (id_x >= width || id_y >= height)) { // padding lane of a partial tile
        a.q.ox()[idx] = 0.0f; a.q.oy()[idx] = 0.0f; a.q.oz()[idx] = 0.0f;
        a.q.dx()[idx] = 0.0f; a.q.dy()[idx] = 0.0f; a.q.dz()[idx] = 0.0f;
        a.q.pixel()[idx] = WFPT_INACTIVE_PIXEL;
        return;
    }
    const uint32_t pixel_idx = id_x + id_y * width; // gr:57
    const PrimaryRay pr = primary_ray(*a.camera, id_x, id_y, width, height, fb);

    a.q.ox()[idx] = pr.ox; a.q.oy()[idx] = pr.oy; a.q.oz()[idx] = pr.oz;
    a.q.dx()[idx] = pr.dx; a.q.dy()[idx] = pr.dy; a.q.dz()[idx] = pr.dz;
    a.q.pixel()[idx] = pixel_idx;
    if (a.reset_image) { // pt:305-306 folded in: throughput starts at 1
        *pixel_of(a.image, local_pixel(pixel_idx, width, a.tile)) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    }
}

// ================================================================================================
// extend (ex:47-210)
// ================================================================================================
// ex:164-183 with the node held as two float4 (min.xyz|left_first, max.xyz|prim_count).
// EXACT (WFPT_FLAG_EXACT_TRAVERSAL): a missed box reports the reference's 1e30 (ex:181). Otherwise it reports kBoxMiss,
// a value ABOVE 1e30, which is also the reference's "nothing hit yet" value of `nearest`: while nothing is hit the
// reference's test `t_near > nearest` (ex:124) reads 1e30 > 1e30 = false for a pair of boxes the ray misses both of,
// and it walks down into that pair, left child first, down to a leaf whose primitive it then tests in vain (the
// primitive lies inside a box the ray misses). Those visits decide nothing; with the larger miss value the same
// comparison leaves such a pair alone: 11 % fewer visits for primary rays, 7 % for the others (oracle model,
// tools/model_schedule.py), the hits unchanged.
constexpr float kBoxMiss = 3.0e38f;
// the slab distances of ex:165-178, in the reference's operation order
__device__ __forceinline__ void slab_range(float4 bmin, float4 bmax, float ox, float oy, float oz, float ix, float iy, float iz,
                                           float &tmin_out, float &tmax_out) {
    const float t_x_min = (bmin.x - ox) * ix;
    const float t_x_max = (bmax.x - ox) * ix;
    float tmin = min_(t_x_min, t_x_max);
    float tmax = max_(t_x_min, t_x_max);
    const float t_y_min = (bmin.y - oy) * iy;
    const float t_y_max = (bmax.y - oy) * iy;
    tmin = max_(min_(t_y_min, t_y_max), tmin);
    tmax = min_(max_(t_y_min, t_y_max), tmax);
    const float t_z_min = (bmin.z - oz) * iz;
    const float t_z_max = (bmax.z - oz) * iz;
    tmin_out = max_(min_(t_z_min, t_z_max), tmin);
    tmax_out = min_(max_(t_z_min, t_z_max), tmax);
}
template <bool EXACT>
__device__ __forceinline__ float hit_bvh_node(float4 bmin, float4 bmax, float ox, float oy, float oz, float ix,
                                              float iy, float iz, float nearest) {
    const float t_x_min = (bmin.x - ox) * ix;
    const float t_x_max = (bmax.x - ox) * ix;
    float tmin = min_(t_x_min, t_x_max);
    float tmax = max_(t_x_min, t_x_max);
    const float t_y_min = (bmin.y - oy) * iy;
    const float t_y_max = (bmax.y - oy) * iy;
    tmin = max_(min_(t_y_min, t_y_max), tmin);
    tmax = min_(max_(t_y_min, t_y_max), tmax);
    const float t_z_min = (bmin.z - oz) * iz;
    const float t_z_max = (bmax.z - oz) * iz;
    tmin = max_(min_(t_z_min, t_z_max), tmin);
    tmax = min_(max_(t_z_min, t_z_max), tmax);
    return (tmin > tmax || tmax <= 0.0f || tmin > nearest) ? (EXACT ? 1e30f : kBoxMiss) : tmin;
}

// measurement-only switches (tools/build_variant.sh): what each safety layer of the free walks costs. NOT for use: without them
// the walks are no longer equivalent to the reference's (tests/test_traversal_model.py holds the counter-examples).
#ifndef WFPT_EXP_NO_TIE
#define WFPT_EXP_NO_TIE 0     // 1: no near-tie watch, no probe of failed leaves, no far-origin hand-over
#endif
#ifndef WFPT_EXP_NO_LEAFBOX
#define WFPT_EXP_NO_LEAFBOX 0 // 1: leaf boxes are not re-tested with the reference's arithmetic
#endif
#ifndef WFPT_VISIT4_PK
#define WFPT_VISIT4_PK 0 // 1: visit4's plane distances two children at a time with v_pk_fma_f32
#endif
#ifndef WFPT_VISIT4_PARTIAL_SORT
#define WFPT_VISIT4_PARTIAL_SORT 0 // 1: visit4 brings only the nearest child to the front (measured: profiles/r04_rejected_experiments.txt)
#endif
#ifndef WFPT_LEAF_BOX_FUSED
#define WFPT_LEAF_BOX_FUSED 1 // visit_leaf_in_place gathers the leaf's box in the loop that tests its primitives (one fetch of each)
#endif
#ifndef WFPT_LEAF_LANES
#define WFPT_LEAF_LANES 8 // refill_kernel: lanes that have to wait at a leaf before the wave runs the leaf code
#endif
#ifndef WFPT_STAMPS
#define WFPT_STAMPS 0 // 1: diagnostic build; the middle bounce launches add up, per wave, the shader cycles (s_memtime) spent in each phase
#endif
#if WFPT_STAMPS
// (volatile asm with a memory clobber: the compiler may neither move nor merge these reads)
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define WFPT_STAMP(var) const unsigned long long var = stamp_now()
// per-lane step counters of the free walk, in registers: [0] loop trips of the inner-visit loop as this WAVE ran them (a uniform
// count carried through readfirstlane), [1] leaf rounds of the wave, [2] this lane's own inner visits
#define WFPT_DBG_PARAM , uint32_t (&dbg)[3]
#define WFPT_DBG_ARG , dbg
#else
#define WFPT_STAMP(var)
#define WFPT_DBG_PARAM
#define WFPT_DBG_ARG
#endif
#ifndef WFPT_BUDGET_INNER
#define WFPT_BUDGET_INNER 0 // 1: count the step budget down on every inner visit as well (costs 3 instructions per visit)
#endif

// Marks all four components of loaded float4s as used, so each load stays one ds_read_b128 (hipcc otherwise
// narrows a load whose .w is consumed elsewhere to ds_read_b96 + ds_read_b32: 2.5x the LDS cycles). Input-only
// and placed after ALL loads of a step, so the loads issue back to back and are waited for once.
__device__ __forceinline__ void keep4(const float4 &a, const float4 &b, const float4 &c, const float4 &d) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y),
                 "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
}

// trace_ray (ex:72-162) without a stack. The reference pushes the far child when t_far < nearest and
// pops LIFO; pushes along one descent have strictly increasing depth, so "the stack" is exactly the set
// of tree levels with a pending far sibling: one bit per level (`trail`, kept as a shift register). Siblings sit at (2k, 2k+1)
// (bvh.rs:160, 191-206), so the pending node at a level is (path node at that level) ^ 1, and the path
// node is found by walking `pair_parent` up from the current node. Same visit order, same comparisons,
// same results as the stack version, but no per-lane stack memory.
//
// Scheduling is "while-while": all lanes first run inner-node steps until each sits on a leaf (or is
// done), then all leaf lanes test their spheres together. Each lane still performs exactly the
// reference's sequence of operations; only the interleaving across lanes changes, which keeps more lanes
// active per instruction than alternating leaf/inner work every iteration.
// STACK_DEPTH > 0 (HBM-resident scenes, where a parent-table step is an L2 round trip): the last STACK_DEPTH pushed
// nodes are also kept in a per-lane LDS column, so a pop is two ds_reads. The trail stays authoritative; pushes
// beyond the column's depth are only counted (`lost`) and popped by the parent walk, which keeps LIFO order because
// the lost entries are always the most recent ones. An entry is (node, its left_first / prim_count packed): the far
// child's fields are in registers when it is pushed, and re-reading them at the pop would be one more dependent
// trip to L2 / Infinity Cache before the popped node's own children can be requested. 8 entries cover 99.9 % of the
// inner visits of the 1 M-triangle scene (oracle model).
constexpr uint32_t kStackDepth = 8;
static_assert(kStack4Lds <= 2u * kStackDepth, "the four-wide traversal's LDS column reuses the binary variant's stack area");
constexpr uint32_t kMetaCountBits = 6; // packed fields: left_first << 6 | prim_count; 0xffffffff = does not fit, re-read

template <typename Trail, typename ParentT, uint32_t STACK_DEPTH = 0>
struct Traversal {
    uint32_t node, left_first, prim_count;
    Trail trail; // shift register: bit i = "the far sibling is still pending" at the path node i levels above the current one
    uint32_t sp = 0, lost = 0;
    uint32_t *stack = nullptr; // this lane's column: entry k at stack[k * kExtendThreads]

    __device__ __forceinline__ void push(uint32_t far_node, uint32_t far_left_first, uint32_t far_prim_count) {
        if (STACK_DEPTH == 0) return;
        if (sp < STACK_DEPTH) {
            const bool fits = far_left_first < (1u << (32u - kMetaCountBits)) && far_prim_count < (1u << kMetaCountBits) &&
                              ((far_left_first << kMetaCountBits) | far_prim_count) != 0xffffffffu;
            stack[(2u * sp) * kExtendThreads] = far_node;
            stack[(2u * sp + 1u) * kExtendThreads] = fits ? ((far_left_first << kMetaCountBits) | far_prim_count) : 0xffffffffu;
            sp += 1;
        } else {
            lost += 1;
        }
    }

    // LIFO pop: deepest pending level. Returns false when nothing is pending (ex:95-97, 125-127: break).
    __device__ __forceinline__ bool pop(const float4 *nodes, const ParentT *pair_parent) {
        if (trail == 0) return false;
        // levels to climb to the deepest pending sibling = trailing zeros
        const uint32_t up = (sizeof(Trail) == 8) ? static_cast<uint32_t>(__ffsll(static_cast<long long>(trail)) - 1)
                                                 : static_cast<uint32_t>(__ffs(static_cast<int>(trail)) - 1);
        trail = (trail >> up) & ~static_cast<Trail>(1); // now at that level, its pending flag consumed
        uint32_t fields = 0xffffffffu;
        if (STACK_DEPTH > 0 && lost == 0) {
            sp -= 1;
            node = stack[(2u * sp) * kExtendThreads];
            fields = stack[(2u * sp + 1u) * kExtendThreads];
        } else {
            for (uint32_t k = 0; k < up; ++k) node = pair_parent[node >> 1];
            node ^= 1u;
            if (STACK_DEPTH > 0) lost -= 1;
        }
        if (STACK_DEPTH > 0 && fields != 0xffffffffu) {
            left_first = fields >> kMetaCountBits;
            prim_count = fields & ((1u << kMetaCountBits) - 1u);
        } else {
            left_first = __float_as_uint(nodes[2u * node].w);
            prim_count = __float_as_uint(nodes[2u * node + 1u].w);
        }
        return true;
    }
};

// Primitive tests. PRIM 0: hit(), ex:185-210 (sphere, half-b quadratic, both roots against the window).
// PRIM 1 (build extension, no reference code): Moeller-Trumbore in a fixed evaluation order; a degenerate
// triangle gives inf/NaN barycentrics and fails the negated range tests.
// Two candidate distances within 2^-18 of each other (relative): which of them the reference reports can hinge on the ORDER in
// which its walk meets them (the first of two bit-equal hits wins, ex:190-207's strict `<`; a box whose entry distance lies a
// rounding error beyond a hit inside it is skipped or not depending on what was found before). The walks that are free in
// their visit order (trace_ray_conservative, trace_ray4) watch for this and hand such a ray to the reference's own walk.
__device__ __forceinline__ bool near_tie(float t, float nearest) { return !WFPT_EXP_NO_TIE && __builtin_fabsf(t - nearest) <= nearest * 3.8146973e-6f; }
// How a free walk marks "hand this ray over": no flag of its own (a lane mask kept alive across the whole loop nest cost the
// kernel ~40 scalar registers' worth of spills) but a poisoned result: nearest = -1 makes every later box and primitive test
// fail, so the walk runs out by itself, and best = kHandOver tells the caller why.
constexpr uint32_t kHandOver = 0xfffffffeu;
__device__ __forceinline__ void hand_over(float &nearest, uint32_t &best) {
    nearest = -1.0f;
    best = kHandOver;
}

template <int PRIM, bool TRACK = false>
__device__ __forceinline__ void hit_prim(const float4 *geom, uint32_t idx, float ox, float oy, float oz, float dx, float dy,
                                         float dz, float a, float &nearest, uint32_t &best) {
    if (PRIM == 0) {
        const float4 s = geom[idx];
        const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
        const float b = (dx * ocx + dy * ocy) + dz * ocz;
        const float c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - s.w * s.w;
        const float discrim = b * b - a * c;
        if (discrim >= 0.0f) {
            const float sq = sqrt_(discrim);
            float t = (-b - sq) / a;
            if (TRACK) { // as selects, not branches: a poisoned window (nearest = -1) accepts nothing below
                const bool tie = t > 0.001f && near_tie(t, nearest);
                nearest = tie ? -1.0f : nearest;
                best = tie ? kHandOver : best;
            }
            if (t > 0.001f && t < nearest) {
                nearest = t;
                best = idx;
            } else {
                t = (-b + sq) / a;
                if (TRACK) {
                    const bool tie = t > 0.001f && near_tie(t, nearest);
                    nearest = tie ? -1.0f : nearest;
                    best = tie ? kHandOver : best;
                }
                if (t > 0.001f && t < nearest) {
                    nearest = t;
                    best = idx;
                }
            }
        }
    } else {
        const float4 v0 = geom[3u * idx], e1 = geom[3u * idx + 1u], e2 = geom[3u * idx + 2u];
        const float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x; // cross(d, e2)
        const float det = (e1.x * px + e1.y * py) + e1.z * pz;
        const float inv_det = 1.0f / det;
        const float tx = ox - v0.x, ty = oy - v0.y, tz = oz - v0.z;
        const float u = ((tx * px + ty * py) + tz * pz) * inv_det;
        if (u >= 0.0f && u <= 1.0f) {
            const float qx = ty * e1.z - tz * e1.y, qy = tz * e1.x - tx * e1.z, qz = tx * e1.y - ty * e1.x; // cross(tvec, e1)
            const float v = ((dx * qx + dy * qy) + dz * qz) * inv_det;
            if (v >= 0.0f && u + v <= 1.0f) {
                const float t = ((e2.x * qx + e2.y * qy) + e2.z * qz) * inv_det;
                if (TRACK && t > 0.001f && near_tie(t, nearest)) {
                    hand_over(nearest, best);
                } else if (t > 0.001f && t < nearest) {
                    nearest = t;
                    best = idx;
                }
            }
        }
    }
}

// The box of primitive `idx` exactly as the builder computes it (sphere.rs:22-26: centre -+ radius; triangles, build extension:
// min / max over v0, v0 + e1, v0 + e2), grown into [lo, hi] like update_node_bounds does (bvh.rs:58-70). wfpt_create checks that
// the caller's leaf boxes are bit for bit what this gives (leaf_boxes_recomputable, wfpt_api.hip).
template <int PRIM>
__device__ __forceinline__ void grow_prim_box(const float4 *geom, uint32_t idx, float3_ &lo, float3_ &hi) {
    if (PRIM == 0) {
        const float4 s = geom[idx];
        lo = {min_(lo.x, s.x - s.w), min_(lo.y, s.y - s.w), min_(lo.z, s.z - s.w)};
        hi = {max_(hi.x, s.x + s.w), max_(hi.y, s.y + s.w), max_(hi.z, s.z + s.w)};
    } else {
        const float4 v0 = geom[3u * idx], e1 = geom[3u * idx + 1u], e2 = geom[3u * idx + 2u];
        const float bx = v0.x + e1.x, by = v0.y + e1.y, bz = v0.z + e1.z, cx = v0.x + e2.x, cy = v0.y + e2.y, cz = v0.z + e2.z;
        lo = {min_(lo.x, min_(min_(v0.x, bx), cx)), min_(lo.y, min_(min_(v0.y, by), cy)), min_(lo.z, min_(min_(v0.z, bz), cz))};
        hi = {max_(hi.x, max_(max_(v0.x, bx), cx)), max_(hi.y, max_(max_(v0.y, by), cy)), max_(hi.z, max_(max_(v0.z, bz), cz))};
    }
}
// The free walks' guarantees hold for rays that start within SceneDev::safe_r of safe_c (see build_nodes_ch: there the primitive
// test's own rounding slack stays inside the boxes' margin). A ray from farther away -- a bounce off the ground sphere hundreds of
// units out -- is traced by the reference's own walk.
__device__ __forceinline__ bool far_origin(const SceneDev &sc, float ox, float oy, float oz) {
    if (WFPT_EXP_NO_TIE) return false;
    const float fx = ox - sc.safe_c[0], fy = oy - sc.safe_c[1], fz = oz - sc.safe_c[2];
    return (fx * fx + fy * fy) + fz * fz > sc.safe_r2;
}

// A leaf of a free walk, and the ONE box test a free walk owes the reference (rewritten in round 4).
//
// The reference tests the primitives of exactly those leaves whose own box passes its test (ex:164-183), plus those its blind
// descent reaches (ex:124, `1e30 > 1e30`); a primitive test rounds, so a free walk -- which reaches every leaf a ray passes within
// the boxes' margin -- also meets "false positives": primitives the exact test accepts although the ray misses their leaf's box
// (DESIGN.md section 2). Round 3 asked the leaf's box for a verdict every time a primitive test changed the tentative result,
// inside the walk's most divergent code. One verdict per ray, after the walk, is enough:
//   * U = the primitives hit_prim accepts among all the walk tests, T = those the reference tests (T is a subset of U wherever the
//     walk's own pruning by `nearest` did not cut it short; a pruned box holds nothing nearer than the value that pruned it).
//     The walk reports b = min U, the reference r = min T (first visited on exactly equal distances).
//   * If b's leaf passes the reference's box test with entry distance tmin <= t(b), then b is in T: every ancestor's box passes too
//     (nested boxes, monotone arithmetic) with an entry distance <= tmin <= t(b) <= whatever `nearest` the reference holds when it
//     gets there (that value is the distance of some member of T, hence >= t(b)), so no `tmin > nearest` prunes the way to b, and a
//     far child is dropped by `t_far < nearest` only on an exact tie of distances, which the watch below hands over. min U in T
//     gives b = r.
//   * Otherwise (the box fails, or the hit computes nearer than its own box's entry distance by rounding, the corner in which the
//     reference's answer depends on what it found before) the ray is handed to the reference's own walk. An intermediate
//     false positive that a nearer hit replaced needs no verdict: it can only have pruned what could not beat it.
//   * Exactly equal distances of two different primitives are the one case left in which the ORDER of visits shows (ex:190-207's
//     strict `<`): hit_prim<TRACK> watches for candidates within 2^-18 (relative) of the nearest so far and poisons the result.
// A CPU model of exactly this walk (the checker's "mode 3") is compared with the reference's walk on the rays of real wavefronts:
// tests/hunt_conservative.py, tests/test_traversal_model.py.
// `leaf_word` names the leaf for the verdict (LDS-resident scenes: left_first | prim_count << 16; four-wide walk: the child word).
template <int PRIM>
__device__ __forceinline__ void visit_leaf(const float4 *geom, uint32_t first, uint32_t count, uint32_t leaf_word, float ox, float oy, float oz,
                                           float dx, float dy, float dz, float a, float &nearest, uint32_t &best, uint32_t &best_leaf) {
    const uint32_t before = best;
    for (uint32_t i = 0; i < count; ++i) hit_prim<PRIM, true>(geom, first + i, ox, oy, oz, dx, dy, dz, a, nearest, best);
    best_leaf = best != before ? leaf_word : best_leaf;
}
// The verdict (see above): the box of leaf [first, first + count), recomputed exactly as the builder computes it, tested with the
// reference's arithmetic (ex:164-183) against the hit's own distance. box_untested: the root is a leaf (its box is never tested, ex:84).
template <int PRIM>
__device__ __forceinline__ void leaf_box_verdict(const float4 *geom, uint32_t first, uint32_t count, bool box_untested, float ox, float oy, float oz,
                                                 float dx, float dy, float dz, float &nearest, uint32_t &best) {
    if (best >= kHandOver || box_untested || WFPT_EXP_NO_LEAFBOX) return; // no hit, or handed over already
    float3_ lo = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t i = 0; i < count; ++i) grow_prim_box<PRIM>(geom, first + i, lo, hi);
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz; // invDirection (gr:87, sh:153)
    float tmin, tmax;
    slab_range(make_float4(lo.x, lo.y, lo.z, 0.0f), make_float4(hi.x, hi.y, hi.z, 0.0f), ox, oy, oz, ix, iy, iz, tmin, tmax);
    if (tmin > tmax || tmax <= 0.0f || tmin > nearest) { // ex:179 with nearest = the hit's own distance
        if (!WFPT_EXP_NO_TIE) hand_over(nearest, best);
    }
}

// The same debt paid inside the walk, as round 3 did everywhere: the primitive tests run into a TENTATIVE result and, if that changed
// anything, the leaf's own box -- tested against the `nearest` the walk held when it arrived -- decides between keeping it and, if a
// primitive would have been accepted although the box fails, handing the ray over (the model's mode 2). Kept for refill_kernel: its
// per-lane state fills the 64 vector registers of 8 waves per SIMD exactly, and the leaf word plus a "result not written yet" flag that
// the deferred verdict needs pushed its hot loop into scratch (measured: 1.98 -> 1.27 Grays/s on the 1 M-triangle soup with the
// verdicts batched in front of a refill, 1.86 with one per finished ray; profiles/r04_rejected_experiments.txt).
// LAZY_INV: the caller does not keep the exact inverse direction (refill_kernel); it is computed where the box is.
template <int PRIM, bool LAZY_INV = false>
__device__ __forceinline__ void visit_leaf_in_place(const float4 *geom, uint32_t first, uint32_t count, bool box_untested, float ox, float oy, float oz,
                                           float dx, float dy, float dz, float ix, float iy, float iz, float a, float &nearest, uint32_t &best) {
    float n2 = nearest;
    uint32_t b2 = best;
#if WFPT_LEAF_BOX_FUSED
    // (round 5) the leaf's box is gathered while its primitives are in registers for their tests: with ~9 lanes in a leaf round some lane
    // accepts a primitive in nearly every round, so the wave ran the second loop -- the primitives fetched again, a dependent round trip --
    // nearly always anyway
    float3_ lo = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t i = 0; i < count; ++i) {
        hit_prim<PRIM, true>(geom, first + i, ox, oy, oz, dx, dy, dz, a, n2, b2);
        grow_prim_box<PRIM>(geom, first + i, lo, hi);
    }
#else
    for (uint32_t i = 0; i < count; ++i) hit_prim<PRIM, true>(geom, first + i, ox, oy, oz, dx, dy, dz, a, n2, b2);
#endif
    // The box's verdict only matters if something changed (a primitive accepted, or a near-tie poisoned the window: n2 = -1):
    // most leaf visits end here, without the box ever being tested.
    if (n2 < nearest) {
        bool enter = box_untested || WFPT_EXP_NO_LEAFBOX;
        if (!enter) {
#if !WFPT_LEAF_BOX_FUSED
            float3_ lo = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            for (uint32_t i = 0; i < count; ++i) grow_prim_box<PRIM>(geom, first + i, lo, hi);
#endif
            float tmin, tmax;
            if (LAZY_INV) { ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz; } // invDirection (gr:87, sh:153)
            slab_range(make_float4(lo.x, lo.y, lo.z, 0.0f), make_float4(hi.x, hi.y, hi.z, 0.0f), ox, oy, oz, ix, iy, iz, tmin, tmax);
            enter = !(tmin > tmax || tmax <= 0.0f || tmin > nearest); // ex:179: the box is entered unless one of the three holds
        }
        if (enter) {
            nearest = n2;
            best = b2;
        } else if (!WFPT_EXP_NO_TIE) {
            hand_over(nearest, best);
        }
    }
}

template <typename Trail, int PRIM, typename ParentT, uint32_t STACK_DEPTH, bool EXACT>
__device__ __forceinline__ bool trace_ray(const float4 *nodes, const float4 *prim_geom, const ParentT *pair_parent,
                                          uint32_t *stack_column, float ox, float oy, float oz, float dx, float dy,
                                          float dz, uint32_t max_steps, float &t_out, uint32_t &prim_out) {
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz; // invDirection (gr:87, sh:153)
    const float a = (dx * dx + dy * dy) + dz * dz;              // dot(direction, direction), ex:190
    float nearest = 1e30f;
    uint32_t best = 0xffffffffu;
    Traversal<Trail, ParentT, STACK_DEPTH> tr;
    tr.stack = stack_column;
    tr.node = 0; // ex:84: the root's box is never tested
    tr.left_first = __float_as_uint(nodes[0].w);
    tr.prim_count = __float_as_uint(nodes[1].w);
    tr.trail = 0;
    bool alive = true;
    // A traversal visits every node at most once, so `max_steps` (= node count) is never reached on a valid tree (the
    // tree is validated at wfpt_create: sibling layout, ranges, no cycles); the budget only guarantees that every wave
    // terminates if the node data were corrupt. It is counted per leaf / pop round; a descent between two leaves is at
    // most `depth` inner visits long on a validated tree.
    uint32_t budget = max_steps;
    while (alive) {
        // ---- inner nodes (ex:105-138)
        while (alive && tr.prim_count == 0) {
            if (WFPT_BUDGET_INNER && budget-- == 0) { alive = false; break; }
            const float4 *pair = nodes + 2u * tr.left_first;
            const float4 lmin = pair[0], lmax = pair[1], rmin = pair[2], rmax = pair[3];
            keep4(lmin, lmax, rmin, rmax);
            const float t_left = hit_bvh_node<EXACT>(lmin, lmax, ox, oy, oz, ix, iy, iz, nearest);
            const float t_right = hit_bvh_node<EXACT>(rmin, rmax, ox, oy, oz, ix, iy, iz, nearest);
            const bool swap = t_left > t_right; // strict: ties keep the left child first
            const float t_near = swap ? t_right : t_left;
            const float t_far = swap ? t_left : t_right;
            if (t_near > nearest) { // ex:124-131
                alive = tr.pop(nodes, pair_parent);
            } else { // ex:132-137: descend into the near child, remember the far one
                tr.node = tr.left_first + (swap ? 1u : 0u);
                const bool pending = t_far < nearest;
                tr.trail = (tr.trail << 1) | static_cast<Trail>(pending ? 1u : 0u);
                if (pending) tr.push(tr.left_first + (swap ? 0u : 1u), __float_as_uint(swap ? lmin.w : rmin.w), __float_as_uint(swap ? lmax.w : rmax.w));
                tr.left_first = __float_as_uint(swap ? rmin.w : lmin.w);
                tr.prim_count = __float_as_uint(swap ? rmax.w : lmax.w);
            }
        }
        // ---- leaf (ex:86-103)
        if (alive && budget-- == 0) alive = false;
        if (alive) {
            for (uint32_t i = 0; i < tr.prim_count; ++i)
                hit_prim<PRIM>(prim_geom, tr.left_first + i, ox, oy, oz, dx, dy, dz, a, nearest, best);
            alive = tr.pop(nodes, pair_parent);
        }
    }
    t_out = nearest;
    prim_out = best;
    return nearest < 1e30f; // ex:157
}

// The reference's walk once more, for the hand-over of the free walks: same visits, same arithmetic and same results as
// trace_ray<..., EXACT = true>, written as ONE flat loop (a step is an inner visit, a leaf, or one level of the climb to the
// pending sibling). It runs for a handful of rays per million, so its speed is irrelevant; what matters is that it asks little
// of the register allocator: the nested loops of trace_ray inlined beside the fast walk cost the fast walk scalar registers
// (every saved exec mask of a loop nest is a pair), which came back as spill traffic around every work item.
template <typename Trail, int PRIM, typename ParentT>
__device__ __forceinline__ bool retrace_reference(const float4 *nodes, const float4 *prim_geom, const ParentT *pair_parent, float ox, float oy,
                                                  float oz, float dx, float dy, float dz, uint32_t max_steps, float &t_out, uint32_t &prim_out) {
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    const float a = (dx * dx + dy * dy) + dz * dz;
    float nearest = 1e30f;
    uint32_t best = 0xffffffffu, node = 0, left_first = __float_as_uint(nodes[0].w), prim_count = __float_as_uint(nodes[1].w), climb = 0;
    Trail trail = 0;
    uint32_t budget = 4u * max_steps + 64u; // every node is visited once and climbed through at most once per pop
    bool alive = true, popping = false;
    while (alive && budget-- != 0u) {
        if (popping) { // one step of Traversal::pop
            if (climb != 0u) {
                node = pair_parent[node >> 1];
                climb -= 1u;
            } else {
                node ^= 1u;
                left_first = __float_as_uint(nodes[2u * node].w);
                prim_count = __float_as_uint(nodes[2u * node + 1u].w);
                popping = false;
            }
            continue;
        }
        bool pop = false;
        if (prim_count == 0u) { // ex:105-138
            const float4 *pair = nodes + 2u * left_first;
            const float4 lmin = pair[0], lmax = pair[1], rmin = pair[2], rmax = pair[3];
            const float t_left = hit_bvh_node<true>(lmin, lmax, ox, oy, oz, ix, iy, iz, nearest);
            const float t_right = hit_bvh_node<true>(rmin, rmax, ox, oy, oz, ix, iy, iz, nearest);
            const bool swap = t_left > t_right;
            const float t_near = swap ? t_right : t_left, t_far = swap ? t_left : t_right;
            if (t_near > nearest) {
                pop = true;
            } else {
                node = left_first + (swap ? 1u : 0u);
                trail = (trail << 1) | static_cast<Trail>(t_far < nearest ? 1u : 0u);
                left_first = __float_as_uint(swap ? rmin.w : lmin.w);
                prim_count = __float_as_uint(swap ? rmax.w : lmax.w);
            }
        } else { // ex:86-103
            for (uint32_t i = 0; i < prim_count; ++i) hit_prim<PRIM>(prim_geom, left_first + i, ox, oy, oz, dx, dy, dz, a, nearest, best);
            pop = true;
        }
        if (pop) {
            if (trail == 0) {
                alive = false;
            } else {
                climb = (sizeof(Trail) == 8) ? static_cast<uint32_t>(__ffsll(static_cast<long long>(trail)) - 1)
                                             : static_cast<uint32_t>(__ffs(static_cast<int>(trail)) - 1);
                trail = (trail >> climb) & ~static_cast<Trail>(1);
                popping = true;
            }
        }
    }
    t_out = nearest;
    prim_out = best;
    return nearest < 1e30f;
}

// ---- the LDS-resident traversal as it runs by default: same walk, CONSERVATIVE test of inner boxes, EXACT test of leaf boxes
// What must not change is the set of primitives whose exact test (hit_prim, ex:185-210) runs: that test rounds too (its
// discriminant cancels ~1e-7 of b^2), so a sphere "hit" can be reported for a ray that passes ~1e-4 outside the sphere -- and
// the reference never sees it, because the ray misses the sphere's BOX and the leaf is never entered. (Found by
// tools/hunt_conservative.py: with every box grown, 2 rays in 1.6e8 reported such a hit; in the dispatch-keyed RNG mode one
// changed hit re-keys the rest of the sample.) So boxes are filters the result depends on, and they are treated in two classes:
//   * LEAF boxes are tested with the reference's own arithmetic -- the box of the final hit's leaf, once, after the walk (leaf_box_verdict);
//   * INNER boxes only have to say "maybe" whenever the reference's test would enter them.
// That is enough, because the reference's test is MONOTONE in the box: IEEE subtraction and multiplication by a fixed inverse
// are monotone, so for nested boxes L inside A (a BVH's boxes nest in float coordinates; checked at wfpt_create) every plane
// distance of A lies outside the corresponding one of L and "the reference enters L" implies "the reference enters A" (its
// geometric part: tmin <= tmax, tmax > 0). Hence the reference tests exactly the primitives of the leaves whose own box passes
// its test -- and so does any walk that reaches every such leaf and applies that test there. (The reference's third condition,
// tmin > nearest, only skips boxes that cannot hold a nearer hit; which of two primitives with bit-equal t is reported, and a
// hit that is nearer than its own box's entry distance by rounding while another hit lies in between, are the two corners in
// which the ORDER of visits can show; the order below follows the reference's rule on conservative distances.)
// Inner boxes: a box is kept as centre c and half-extent h (SceneDev::nodes_ch, built at wfpt_create), h grown on the host by
// MORE than this test's own rounding error can reach (build_nodes_ch, wfpt_api.hip): for every ray whose origin lies within
// four scene extents of the origin, computed entry distance <= the exact box's and computed exit distance >= its exit
// distance. (wfpt_create falls back to the exact test when a camera or an injected ray lies outside that range, when a box
// is not finite, or when the caller's leaf boxes are not what leaf_box_verdict recomputes: tree_is_recomputable, wfpt_api.hip.)
//   per axis: tc = c * inv - o * inv (one fma against the per-ray constant -(o * inv)),
//             t_entry = tc - h * |inv|, t_exit = tc + h * |inv| (one fma each; the sign of inv needs no min / max),
//   entered  <=> max(t_entry over axes, 0) <= min(t_exit over axes, nearest)
// 9 fma + max3 + min3 + max + min + compare per box where the reference form costs 6 sub/mul pairs + 12 min / max + 3
// compares. The walk is the reference's (root box never tested; both children entered: nearer entry first, ties keep
// the left child; the far one pending). Infinite inverses (a direction component that is exactly zero) are clamped to
// +-1e30 for the inner boxes: the two planes of such an axis then read "always" or "never" like the reference's +-inf,
// without inf - inf; the leaf test uses the unclamped inverse, as the reference does.
#ifndef WFPT_WALK_ASM
#define WFPT_WALK_ASM 1 // 1: the inner-node loop of the LDS walk (32-bit trail) as hand-written gfx950 assembly (descend_asm); 0: the compiler's loop
#endif
// LDS byte offset of a pointer into the workgroup's dynamic LDS (the operand of a ds_read)
__device__ __forceinline__ uint32_t lds_offset(const void *p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((WFPT_AS_LDS const char *)p));
}
// ... of a pointer `p` derived from the kernel's `extern __shared__` array `base`: as a distance from that array, whose own address the
// compiler knows (a cast of a generic pointer back to LDS costs a null check and an aperture compare wherever it is rematerialised)
#define WFPT_LDS_BYTES(base, p) (lds_offset(base) + static_cast<uint32_t>(reinterpret_cast<const char *>(p) - reinterpret_cast<const char *>(base)))
// The inner-node loop of trace_ray_conservative -- every lane that sits on an inner node (prim_count == 0) visits node pairs until it sits
// on a leaf or its walk is over (prim_count = kWalkDone) -- written by hand, because what bounds the launch is instruction ISSUE, of both
// kinds: hipcc's structurizer turns the loop nest (visit | pop with its parent-table walk) into 34-36 scalar mask instructions and
// 46-47 vector instructions per visit; this loop has 11 scalar and 35 vector ones on the descending path. Same operations on the same
// values as the C++ statement of the visit (the #else branch below; WFPT_WALK_ASM=0 builds it), in the instruction selection hipcc itself
// chose for them (v_max3 / v_min / v_min3 / source modifiers), so both give the same bits:
//   per box: tc = c * b + (-o * b); t_in = max3(tc - h |b|); t_out = min(min(tcx + hx |bx|, tcy + hy |by|), tcz + hz |bz|, nearest)
//            entered <=> max(t_in, 0) <= t_out
//   go_right = hit_r & (!hit_l | l_in > r_in); both -> the far child stays pending (trail bit); none -> pop
//   pop: nothing pending -> done; else climb `ffbl(trail)` levels through the u16 parent table, take the sibling, read its fields.
// exec is narrowed to the lanes still in the loop and restored at the end. The node pair lands in v[32:47], named in the clobber list (an
// inline-asm operand cannot name the components of a 128-bit register tuple; of the ranges tried, [48:63] left 8-14 dwords of scratch spill in
// the bounce kernels, [32:47] and [16:31] none in the middle launches).
#if WFPT_STAMPS // diagnostic builds count the wave's trips through the loop (a scalar) and every lane's own visits
#define WFPT_ASM_COUNT "s_add_u32 %[dbgw], %[dbgw], 1\n\tv_add_u32_e32 %[dbgl], 1, %[dbgl]\n\t"
#define WFPT_ASM_COUNT_OPS , [dbgw] "+s"(dbg_wave), [dbgl] "+v"(dbg_lane)
#else
#define WFPT_ASM_COUNT
#define WFPT_ASM_COUNT_OPS
#endif
__device__ __forceinline__ void descend_asm(uint32_t nodes_lds, uint32_t parent_lds, float bx, float by, float bz, float nox, float noy, float noz,
                                            float nearest, uint32_t &node, uint32_t &left_first, uint32_t &prim_count, uint32_t &trail WFPT_DBG_PARAM) {
#if WFPT_STAMPS
    uint32_t dbg_wave = __builtin_amdgcn_readfirstlane(dbg[0]), dbg_lane = dbg[2];
#endif
    unsigned long long m_save, m_cur, m_l, m_r, m_go, m_t;
    float t1, t2; // every other temporary is a register of the node pair whose value has been consumed (the kernel has 64 vector registers)
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[pc]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lwfpt_end%=\n"
        ".Lwfpt_loop%=:\n\t" WFPT_ASM_COUNT
        "v_lshl_add_u32 %[t2], %[lf], 5, %[nodes]\n\t"
        "ds_read_b128 v[32:35], %[t2]\n\t"            // left:  centre.xyz | left_first
        "ds_read_b128 v[36:39], %[t2] offset:16\n\t"  //        half.xyz   | prim_count
        "ds_read_b128 v[40:43], %[t2] offset:32\n\t"  // right: centre.xyz | left_first
        "ds_read_b128 v[44:47], %[t2] offset:48\n\t"  //        half.xyz   | prim_count
        "s_mov_b64 %[cur], exec\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_fma_f32 v32, v32, %[bx], %[nox]\n\t"       // tc = c * b - o * b
        "v_fma_f32 v33, v33, %[by], %[noy]\n\t"
        "v_fma_f32 v34, v34, %[bz], %[noz]\n\t"
        "v_fma_f32 %[t1], v36, -|%[bx]|, v32\n\t"     // entry distances tc - h |b| ...
        "v_fma_f32 v32, v36, |%[bx]|, v32\n\t"        // ... exit distances tc + h |b|
        "v_fma_f32 v36, v37, -|%[by]|, v33\n\t"
        "v_fma_f32 v33, v37, |%[by]|, v33\n\t"
        "v_fma_f32 v37, v38, -|%[bz]|, v34\n\t"
        "v_fma_f32 v34, v38, |%[bz]|, v34\n\t"
        "v_max3_f32 %[t1], %[t1], v36, v37\n\t"       // l_in
        "v_min_f32_e32 v32, v32, v33\n\t"
        "v_min3_f32 v32, v32, v34, %[nearest]\n\t"
        "v_max_f32_e32 v33, 0, %[t1]\n\t"
        "v_cmp_le_f32_e64 %[ml], v33, v32\n\t"        // hit_l
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fma_f32 v40, v40, %[bx], %[nox]\n\t"
        "v_fma_f32 v41, v41, %[by], %[noy]\n\t"
        "v_fma_f32 v42, v42, %[bz], %[noz]\n\t"
        "v_fma_f32 %[t2], v44, -|%[bx]|, v40\n\t"
        "v_fma_f32 v40, v44, |%[bx]|, v40\n\t"
        "v_fma_f32 v44, v45, -|%[by]|, v41\n\t"
        "v_fma_f32 v41, v45, |%[by]|, v41\n\t"
        "v_fma_f32 v45, v46, -|%[bz]|, v42\n\t"
        "v_fma_f32 v42, v46, |%[bz]|, v42\n\t"
        "v_max3_f32 %[t2], %[t2], v44, v45\n\t"       // r_in
        "v_min_f32_e32 v40, v40, v41\n\t"
        "v_min3_f32 v40, v40, v42, %[nearest]\n\t"
        "v_max_f32_e32 v41, 0, %[t2]\n\t"
        "v_cmp_le_f32_e64 %[mr], v41, v40\n\t"        // hit_r
        "v_cmp_gt_f32_e32 vcc, %[t1], %[t2]\n\t"      // r_nearer = l_in > r_in
        "s_orn2_b64 %[mt], vcc, %[ml]\n\t"
        "s_and_b64 %[mgo], %[mt], %[mr]\n\t"          // go_right = hit_r & (r_nearer | !hit_l)
        "s_or_b64 %[mt], %[ml], %[mr]\n\t"            // entered at all
        "s_and_b64 %[ml], %[ml], %[mr]\n\t"           // both
        // ---- descend (lanes that entered a child)
        "s_and_b64 exec, %[cur], %[mt]\n\t"
        "v_addc_co_u32_e64 %[node], %[mr], 0, %[lf], %[mgo]\n\t" // node = left_first + go_right (the carry out is not used)
        "v_addc_co_u32_e64 %[trail], %[mr], %[trail], %[trail], %[ml]\n\t" // trail = trail << 1 | both (trail + trail + carry-in; a trail has at most 31 bits)
        "v_cndmask_b32_e64 %[lf], v35, v43, %[mgo]\n\t"
        "v_cndmask_b32_e64 %[pc], v39, v47, %[mgo]\n\t"
        // ---- pop (lanes that entered neither)
        "s_andn2_b64 exec, %[cur], %[mt]\n\t"
        "s_cbranch_execz .Lwfpt_next%=\n\t"
        "v_mov_b32_e32 %[pc], -1\n\t"                 // nothing pending: the walk is over (kWalkDone)
        "v_cmp_ne_u32_e32 vcc, 0, %[trail]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lwfpt_next%=\n\t"
        "v_ffbl_b32_e32 %[t1], %[trail]\n\t"          // levels to climb to the deepest pending sibling
        "s_mov_b64 %[mr], exec\n\t"
        "v_lshrrev_b32_e32 %[trail], %[t1], %[trail]\n\t"
        "v_and_b32_e32 %[trail], -2, %[trail]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lwfpt_climbed%=\n"
        ".Lwfpt_climb%=:\n\t"
        "v_and_b32_e32 %[t2], -2, %[node]\n\t"        // pair_parent[node >> 1], u16 entries
        "v_add_u32_e32 %[t2], %[parent], %[t2]\n\t"
        "ds_read_u16 %[node], %[t2]\n\t"
        "v_add_u32_e32 %[t1], -1, %[t1]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz .Lwfpt_climb%=\n"
        ".Lwfpt_climbed%=:\n\t"
        "s_mov_b64 exec, %[mr]\n\t"
        "v_xor_b32_e32 %[node], 1, %[node]\n\t"
        "v_lshl_add_u32 %[t2], %[node], 5, %[nodes]\n\t"
        "ds_read_b32 %[lf], %[t2] offset:12\n\t"
        "ds_read_b32 %[pc], %[t2] offset:28\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        ".Lwfpt_next%=:\n\t"
        "s_mov_b64 exec, %[cur]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[pc]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz .Lwfpt_loop%=\n"
        ".Lwfpt_end%=:\n\t"
        "s_mov_b64 exec, %[save]"
        : [node] "+v"(node), [lf] "+v"(left_first), [pc] "+v"(prim_count), [trail] "+v"(trail), [save] "=&s"(m_save), [cur] "=&s"(m_cur),
          [ml] "=&s"(m_l), [mr] "=&s"(m_r), [mgo] "=&s"(m_go), [mt] "=&s"(m_t), [t1] "=&v"(t1), [t2] "=&v"(t2) WFPT_ASM_COUNT_OPS
        : [nodes] "s"(nodes_lds), [parent] "s"(parent_lds), [bx] "v"(bx), [by] "v"(by), [bz] "v"(bz), [nox] "v"(nox), [noy] "v"(noy), [noz] "v"(noz),
          [nearest] "v"(nearest)
        : "vcc", "scc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
#if WFPT_STAMPS
    dbg[0] = dbg_wave;
    dbg[2] = dbg_lane;
#endif
}

// Round 5: the walk's state is four vector registers and NO boolean. The scalar unit of a CU serves its four SIMDs with about one
// instruction per cycle (tools/microbench_valu.hip salu: 0.55 per ns per SIMD), and the loop nest of rounds 3-4 -- an `alive` flag carried
// through two nested loops and an if / else per visit -- compiled to 36 scalar instructions (exec-mask bookkeeping) per 46 vector
// instructions of a visit: the static mix ran at 0.61 vector instructions per ns per SIMD where the vector instructions alone would reach
// 0.75 (profiles/r05_microbench_salu.txt), and the launch's scalar unit was 58 % busy. Here a finished walk is a value of prim_count
// (kWalkDone) instead of a flag, so both loop conditions are one vector compare each, the descent is computed for every lane and
// overwritten by the (one-sided) pop, and nothing but the exec mask itself lives in scalar registers across an iteration.
constexpr uint32_t kWalkDone = 0xffffffffu; // prim_count of a lane whose walk has ended (a scene in LDS holds fewer than 2^16 primitives)
template <typename Trail, int PRIM, typename ParentT>
__device__ __forceinline__ bool trace_ray_conservative(const float4 *nodes_ch, const float4 *prim_geom, const ParentT *pair_parent, float ox,
                                                       float oy, float oz, float dx, float dy, float dz, uint32_t max_steps, float &t_out,
                                                       uint32_t &prim_out WFPT_DBG_PARAM) {
    const float a = (dx * dx + dy * dy) + dz * dz; // dot(direction, direction), ex:190
    const float bx = min_(max_(1.0f / dx, -1e30f), 1e30f), by = min_(max_(1.0f / dy, -1e30f), 1e30f), bz = min_(max_(1.0f / dz, -1e30f), 1e30f);
    const float nox = -(ox * bx), noy = -(oy * by), noz = -(oz * bz);
    const float ax = __builtin_fabsf(bx), ay = __builtin_fabsf(by), az = __builtin_fabsf(bz);
    float nearest = 1e30f;
    uint32_t best = 0xffffffffu;
    uint32_t node = 0; // ex:84: the root's box is never tested
    uint32_t left_first = __float_as_uint(nodes_ch[0].w), prim_count = __float_as_uint(nodes_ch[1].w);
    Trail trail = 0; // bit i: the far sibling is still pending at the path node i levels above the current one (see Traversal)
    const bool root_leaf = prim_count != 0u; // the root's own box is never tested (ex:84)
    uint32_t best_leaf = 0; // the leaf of `best`: left_first | prim_count << 16
    uint32_t budget = max_steps; // see trace_ray
    const uint32_t nodes_lds = uniform(lds_offset(nodes_ch)), parent_lds = uniform(lds_offset(pair_parent));
    // LIFO pop (Traversal::pop): the deepest pending sibling, found by walking the parent table up from the current node; nothing
    // pending ends the walk (ex:95-97, 125-127: break)
#define WFPT_POP()                                                                                                                     \
    do {                                                                                                                               \
        if (trail == 0) {                                                                                                              \
            prim_count = kWalkDone;                                                                                                    \
        } else {                                                                                                                       \
            const uint32_t up = (sizeof(Trail) == 8) ? static_cast<uint32_t>(__ffsll(static_cast<long long>(trail)) - 1)              \
                                                     : static_cast<uint32_t>(__ffs(static_cast<int>(trail)) - 1);                      \
            trail = (trail >> up) & ~static_cast<Trail>(1);                                                                            \
            for (uint32_t k = 0; k < up; ++k) node = pair_parent[node >> 1];                                                           \
            node ^= 1u;                                                                                                                \
            left_first = __float_as_uint(nodes_ch[2u * node].w);                                                                       \
            prim_count = __float_as_uint(nodes_ch[2u * node + 1u].w);                                                                  \
        }                                                                                                                              \
    } while (0)
    while (prim_count != kWalkDone) {
        if (WFPT_WALK_ASM && sizeof(Trail) == 4 && sizeof(ParentT) == 2) { // inner nodes (ex:105-138), hand-written loop
            uint32_t trail32 = static_cast<uint32_t>(trail);
            descend_asm(nodes_lds, parent_lds, bx, by, bz, nox, noy, noz, nearest, node, left_first, prim_count, trail32 WFPT_DBG_ARG);
            trail = static_cast<Trail>(trail32);
        } else
        while (prim_count == 0u) { // inner nodes (ex:105-138)
#if WFPT_BUDGET_INNER
            if (budget-- == 0) { prim_count = kWalkDone; break; }
#endif
#if WFPT_STAMPS
            dbg[0] = __builtin_amdgcn_readfirstlane(dbg[0]) + 1u;
            dbg[2] += 1u;
#endif
            const float4 *pair = nodes_ch + 2u * left_first;
            const float4 lc = pair[0], lh = pair[1], rc = pair[2], rh = pair[3];
            keep4(lc, lh, rc, rh);
            const float lcx = fma_(lc.x, bx, nox), lcy = fma_(lc.y, by, noy), lcz = fma_(lc.z, bz, noz);
            const float l_in = max_(max_(fma_(lh.x, -ax, lcx), fma_(lh.y, -ay, lcy)), fma_(lh.z, -az, lcz));
            const float l_out = min_(min_(fma_(lh.x, ax, lcx), fma_(lh.y, ay, lcy)), fma_(lh.z, az, lcz));
            const float rcx = fma_(rc.x, bx, nox), rcy = fma_(rc.y, by, noy), rcz = fma_(rc.z, bz, noz);
            const float r_in = max_(max_(fma_(rh.x, -ax, rcx), fma_(rh.y, -ay, rcy)), fma_(rh.z, -az, rcz));
            const float r_out = min_(min_(fma_(rh.x, ax, rcx), fma_(rh.y, ay, rcy)), fma_(rh.z, az, rcz));
            const bool hit_l = max_(l_in, 0.0f) <= min_(l_out, nearest);
            const bool hit_r = max_(r_in, 0.0f) <= min_(r_out, nearest);
            const bool r_nearer = l_in > r_in;
            const bool go_right = hit_r && (!hit_l || r_nearer); // nearer entry first; ties keep the left child (ex:119)
            if (hit_l || hit_r) { // descend; both entered: the far one stays pending
                node = left_first + (go_right ? 1u : 0u);
                trail = (trail << 1) | static_cast<Trail>((hit_l && hit_r) ? 1u : 0u);
                left_first = __float_as_uint(go_right ? rc.w : lc.w);
                prim_count = __float_as_uint(go_right ? rh.w : lh.w);
            } else {
                WFPT_POP();
            }
        }
        if (prim_count != kWalkDone) { // leaf (ex:86-103)
            if (budget-- == 0) {
                prim_count = kWalkDone;
            } else {
#if WFPT_STAMPS
                dbg[1] = __builtin_amdgcn_readfirstlane(dbg[1]) + 1u;
#endif
                visit_leaf<PRIM>(prim_geom, left_first, prim_count, left_first | (prim_count << 16), ox, oy, oz, dx, dy, dz, a, nearest, best, best_leaf);
                WFPT_POP();
            }
        }
    }
#undef WFPT_POP
    leaf_box_verdict<PRIM>(prim_geom, best_leaf & 0xffffu, best_leaf >> 16, root_leaf, ox, oy, oz, dx, dy, dz, nearest, best);
    t_out = nearest;
    prim_out = best; // kHandOver: the caller re-traces with the reference's walk
    return nearest < 1e30f; // ex:157
}

// ---- four-wide traversal for HBM-resident scenes (build extension) ----------------------------
// The closest hit does not depend on the order in which nodes are visited (only on which primitives pass the exact
// test; equal-t ties between different primitives aside), so for scenes whose tree lives in HBM / Infinity Cache the
// binary tree is collapsed into four-wide nodes at wfpt_create: one 64-byte fetch tests four boxes (quantised, each
// enclosing the true box: struct Node4). The box test is the reference's arithmetic (hit_bvh_node: (b - o) * inv) on
// boxes that are never smaller than the reference's, so every primitive the reference would test is still tested. Children are visited nearest first; the others go on a per-lane stack (LDS
// column, spilling to global memory past kStack4Lds entries).
// Entry k of a lane's stack: in LDS (s_stack[k * kExtendThreads + thread]) for k < kStack4Lds, beyond that in the global spill area
// (spill[(k - kStack4Lds) * stride + block * kExtendThreads + thread]; at most a few percent of the pushes). The two bases and
// the stride are uniform and the index is 32-bit, so neither access needs a per-lane 64-bit pointer -- a version that kept
// "this lane's column" as pointers had them spilled to scratch and re-loaded at every pop -- and the LDS and the global access
// stay two instructions of their own address space (a select between the two pointers would make every access a flat one).
// Round 5: the stack pointer IS the LDS byte address of the lane's next free entry (`top` = area + sp * kRow + 4 * thread): a push that
// stays in LDS is one ds_write and one add, a pop one subtract and one ds_read, and "empty" / "in LDS" / "room for three more" are
// compares of `top` with uniform bounds -- where rounds 3-4 kept the index sp and rebuilt the address (two v_mbcnt, two shifts, an add3)
// behind an LDS / spill branch at every one of a visit's three push sites and two pop sites (~9 vector + 7 scalar instructions per site).
// push_lds / pop_lds are for the caller that has checked, once per visit and for the whole wave, that no lane leaves the LDS column
// (room3 / all_in_lds); push / pop keep the general form (spill to global memory beyond kStack4Lds entries).
struct Stack4 {
    static constexpr uint32_t kRow = 4u * kExtendThreads; // bytes between two entries of a lane
    uint32_t area;    // LDS byte offset of the stack area, [kStack4Lds][kExtendThreads] words (uniform)
    uint32_t *spill;  // the spill area (uniform)
    uint32_t stride;  // (uniform)
    uint32_t top;     // this lane's next free entry, as an LDS byte address (entries beyond the column: the address they would have)
    __device__ __forceinline__ void init(uint32_t lds_area_bytes, uint32_t *spill_area, uint32_t spill_stride) {
        area = uniform(lds_area_bytes);
        spill = spill_area;
        stride = spill_stride;
        top = area + 4u * threadIdx.x;
    }
    __device__ __forceinline__ void reset() { top = area + ((top - area) & (kRow - 1u)); }
    __device__ __forceinline__ bool empty() const { return top < area + kRow; }
    __device__ __forceinline__ bool room3() const { return top < area + (kStack4Lds - 2u) * kRow; } // three more pushes stay in LDS
    __device__ __forceinline__ bool in_lds() const { return top < area + (kStack4Lds + 1u) * kRow; } // the top entry (if any) is in LDS
    __device__ __forceinline__ void push_lds(uint32_t w) {
        *(WFPT_AS_LDS uint32_t *)(uintptr_t)top = w;
        top += kRow;
    }
    __device__ __forceinline__ uint32_t pop_lds() {
        top -= kRow;
        return *(const WFPT_AS_LDS uint32_t *)(uintptr_t)top;
    }
    __device__ __forceinline__ size_t spill_slot(uint32_t at) const { // of the entry whose LDS address would be `at`
        const uint32_t sp = (at - area) / kRow, thread = ((at - area) & (kRow - 1u)) >> 2;
        return static_cast<size_t>(sp - kStack4Lds) * stride + blockIdx.x * kExtendThreads + thread;
    }
    __device__ __forceinline__ void push(uint32_t w) {
        if (top < area + kStack4Lds * kRow) {
            *(WFPT_AS_LDS uint32_t *)(uintptr_t)top = w;
        } else {
            ((WFPT_AS_GLOBAL uint32_t *)spill)[spill_slot(top)] = w;
        }
        top += kRow;
    }
    __device__ __forceinline__ uint32_t pop() {
        top -= kRow;
        uint32_t w;
        if (top < area + kStack4Lds * kRow) {
            w = *(const WFPT_AS_LDS uint32_t *)(uintptr_t)top;
        } else {
            w = ((const WFPT_AS_GLOBAL uint32_t *)spill)[spill_slot(top)];
        }
        return w;
    }
};

__device__ __forceinline__ void order2(float &ta, uint32_t &wa, float &tb, uint32_t &wb) { // afterwards ta <= tb
    const bool swap = ta > tb;
    const float t0 = swap ? tb : ta, t1 = swap ? ta : tb;
    const uint32_t w0 = swap ? wb : wa, w1 = swap ? wa : wb;
    ta = t0; tb = t1; wa = w0; wb = w1;
}

// One visit of a (quantised) four-wide node. The four child boxes only have to say "maybe" whenever the reference would enter
// the box they stand for (inner boxes are free, leaf boxes are re-tested exactly: trace_ray_conservative), so a plane distance is
// ONE fma, (origin + q * scale - o) * inv = q * (scale * inv) + (origin - o) * inv, against two per-visit constants per
// axis, and the sign of inv picks the near / far plane WORD (four children at once) instead of a min / max per plane. The host
// grew every child box by more than the rounding error of this form before quantising it (collapse_bvh4's margin; same bound
// as build_nodes_ch). Returns the children to enter sorted by entry distance (t = 2e30: not entered).
struct Visit4 {
    float t0, t1, t2, t3;
    uint32_t w0, w1, w2, w3;
};
struct Ray4 { // per ray: the origin and the inverse direction clamped to +-1e30 (no inf - inf for axis-parallel rays)
    float ox, oy, oz, bx, by, bz;
};
__device__ __forceinline__ Ray4 make_ray4(float ox, float oy, float oz, float dx, float dy, float dz) {
    Ray4 r;
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.bx = min_(max_(1.0f / dx, -1e30f), 1e30f); r.by = min_(max_(1.0f / dy, -1e30f), 1e30f); r.bz = min_(max_(1.0f / dz, -1e30f), 1e30f);
    return r;
}
__device__ __forceinline__ float ubyte(uint32_t word, int k) { return static_cast<float>((word >> (8 * k)) & 0xffu); } // v_cvt_f32_ubyteK
__device__ __forceinline__ Visit4 visit4(const float4 a, const float4 b, const float4 c, const float4 d, const Ray4 &r, float nearest) {
    // scales: powers of two kept as the upper halves of their floats (Node4::scale_hi)
    const uint32_t sxy = __float_as_uint(d.z), szw = __float_as_uint(d.w);
    const float spx = __uint_as_float(sxy << 16) * r.bx, spy = __uint_as_float(sxy & 0xffff0000u) * r.by, spz = __uint_as_float(szw << 16) * r.bz;
    const float opx = (a.x - r.ox) * r.bx, opy = (a.y - r.oy) * r.by, opz = (a.z - r.oz) * r.bz; // (keeps three registers fewer per lane than a pre-multiplied origin)
    const uint32_t qlx = __float_as_uint(b.x), qly = __float_as_uint(b.y), qlz = __float_as_uint(b.z), qhx = __float_as_uint(b.w),
                   qhy = __float_as_uint(c.x), qhz = __float_as_uint(c.y);
    const bool ngx = r.bx < 0.0f, ngy = r.by < 0.0f, ngz = r.bz < 0.0f;
    const uint32_t nx = ngx ? qhx : qlx, fx = ngx ? qlx : qhx, ny = ngy ? qhy : qly, fy = ngy ? qly : qhy, nz = ngz ? qhz : qlz, fz = ngz ? qlz : qhz;
    Visit4 v;
    v.w0 = __float_as_uint(c.z); v.w1 = __float_as_uint(c.w); v.w2 = __float_as_uint(d.x); v.w3 = __float_as_uint(d.y);
    float t[4];
#if WFPT_VISIT4_PK
    // two children per instruction: v_pk_fma_f32 issues two fp32 FMAs in the slot of one (tools/microbench_valu.hip)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 sx = {spx, spx}, sy = {spy, spy}, sz = {spz, spz}, bx = {opx, opx}, by = {opy, opy}, bz = {opz, opz};
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        const f2 inx = __builtin_elementwise_fma(f2{ubyte(nx, k), ubyte(nx, k + 1)}, sx, bx), iny = __builtin_elementwise_fma(f2{ubyte(ny, k), ubyte(ny, k + 1)}, sy, by),
                 inz = __builtin_elementwise_fma(f2{ubyte(nz, k), ubyte(nz, k + 1)}, sz, bz);
        const f2 outx = __builtin_elementwise_fma(f2{ubyte(fx, k), ubyte(fx, k + 1)}, sx, bx), outy = __builtin_elementwise_fma(f2{ubyte(fy, k), ubyte(fy, k + 1)}, sy, by),
                 outz = __builtin_elementwise_fma(f2{ubyte(fz, k), ubyte(fz, k + 1)}, sz, bz);
        const float t_in0 = max_(max_(inx.x, iny.x), inz.x), t_in1 = max_(max_(inx.y, iny.y), inz.y);
        const float t_out0 = min_(min_(outx.x, outy.x), outz.x), t_out1 = min_(min_(outx.y, outy.y), outz.y);
        t[k] = (max_(t_in0, 0.0f) <= min_(t_out0, nearest)) ? t_in0 : 2e30f;
        t[k + 1] = (max_(t_in1, 0.0f) <= min_(t_out1, nearest)) ? t_in1 : 2e30f;
    }
#else
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t_in = max_(max_(fma_(ubyte(nx, k), spx, opx), fma_(ubyte(ny, k), spy, opy)), fma_(ubyte(nz, k), spz, opz));
        const float t_out = min_(min_(fma_(ubyte(fx, k), spx, opx), fma_(ubyte(fy, k), spy, opy)), fma_(ubyte(fz, k), spz, opz));
        t[k] = (max_(t_in, 0.0f) <= min_(t_out, nearest)) ? t_in : 2e30f;
    }
#endif
    // (an absent child has an inverted box, qlo = 255 > qhi = 0 on every axis, so it is not entered; should rounding ever make its
    // two plane distances meet, its word is a leaf of no primitives: nothing to guard here)
    v.t0 = t[0]; v.t1 = t[1]; v.t2 = t[2]; v.t3 = t[3];
#if WFPT_VISIT4_PARTIAL_SORT
    // only the nearest child is brought to the front (three compare-exchanges instead of five); the others are pushed in slot order
    order2(v.t0, v.w0, v.t1, v.w1); order2(v.t2, v.w2, v.t3, v.w3); order2(v.t0, v.w0, v.t2, v.w2);
#else
    order2(v.t0, v.w0, v.t1, v.w1); order2(v.t2, v.w2, v.t3, v.w3); order2(v.t0, v.w0, v.t2, v.w2); order2(v.t1, v.w1, v.t3, v.w3);
    order2(v.t1, v.w1, v.t2, v.w2);
#endif
    return v;
}
// the node's four 16-byte quarters: from the LDS copy of the top of the tree (tile, nodes [0, tile_n)) or from global memory
__device__ __forceinline__ Visit4 visit4_at(const float4 *nodes4, const float4 *tile, uint32_t tile_n, uint32_t cur, const Ray4 &r, float nearest) {
    float4 a, b, c, d;
    if (cur < tile_n) {
        const float4 *nd = tile + 4u * cur;
        a = load4_lds(nd); b = load4_lds(nd + 1); c = load4_lds(nd + 2); d = load4_lds(nd + 3);
    } else {
        const float4 *nd = nodes4 + 4u * static_cast<size_t>(cur);
        a = load4_global(nd); b = load4_global(nd + 1); c = load4_global(nd + 2); d = load4_global(nd + 3);
    }
    return visit4(a, b, c, d, r, nearest);
}

template <int PRIM>
__device__ __forceinline__ bool trace_ray4(const float4 *nodes4, const float4 *prim_geom, Stack4 st, float ox, float oy, float oz, float dx,
                                           float dy, float dz, uint32_t max_steps, bool root_leaf, float &t_out, uint32_t &prim_out) {
    const float a = (dx * dx + dy * dy) + dz * dz;
    const Ray4 r4 = make_ray4(ox, oy, oz, dx, dy, dz);
    float nearest = 1e30f;
    uint32_t best = 0xffffffffu, best_leaf = 0;
    uint32_t cur = 0; // node 0 is the root's four-wide node (the root's own box is never tested, ex:84)
    bool alive = true;
    uint32_t budget = max_steps; // every node is visited at most once on a valid tree
#ifdef WFPT_DEBUG_BUDGET
    budget = WFPT_DEBUG_BUDGET;
#endif
    while (alive) {
        while (alive && !(cur & kLeafFlag)) {
            if (budget-- == 0) { alive = false; break; }
            const Visit4 v = visit4_at(nodes4, nullptr, 0u, cur, r4, nearest);
            if (v.t0 >= 2e30f) { // nothing to enter
                if (st.empty()) alive = false; else cur = st.pop();
            } else {
                if (v.t3 < 2e30f) st.push(v.w3); // farthest first, so the nearer ones pop first
                if (v.t2 < 2e30f) st.push(v.w2);
                if (v.t1 < 2e30f) st.push(v.w1);
                cur = v.w0;
            }
        }
        if (alive && budget-- == 0) alive = false;
        if (alive) { // leaf child: kLeafFlag | count << 28 | first
            // the quantised boxes above are LARGER than the caller's: the box of the final hit's leaf gets the reference's verdict
            // after the walk (leaf_box_verdict)
            const uint32_t first = cur & kLeafFirstMask, count = (cur >> kLeafCountShift) & 7u;
            visit_leaf<PRIM>(prim_geom, first, count, cur, ox, oy, oz, dx, dy, dz, a, nearest, best, best_leaf);
            if (st.empty()) alive = false; else cur = st.pop();
        }
    }
    leaf_box_verdict<PRIM>(prim_geom, best_leaf & kLeafFirstMask, (best_leaf >> kLeafCountShift) & 7u, root_leaf, ox, oy, oz, dx, dy, dz, nearest, best);
    t_out = nearest;
    prim_out = best;
    return nearest < 1e30f;
}

// The eight waves' hit and miss counts of a work item -> this wave's offsets. `cnt` = [16] words in LDS: hits of waves 0-7, misses of waves
// 0-7 (written by lane 0 of every wave before the item's barrier). One ds_read per lane of the first row, an inclusive scan over the row with
// four DPP adds, four v_readlane: where the obvious loop over the waves -- (w < wave) ? count : 0 -- compiled to 16 v_readfirstlane and, because
// hipcc keeps the eight `w < wave` booleans as lane masks in scalar registers it then has to spill, ~32 v_readlane and ~40 scalar instructions.
#ifndef WFPT_DPP_PREFIX
#define WFPT_DPP_PREFIX 1
#endif
__device__ __forceinline__ void wave_offsets(const uint32_t *cnt, uint32_t wave, uint32_t lane, uint32_t &hit_before, uint32_t &hit_total,
                                             uint32_t &miss_before, uint32_t &miss_total) {
#if WFPT_DPP_PREFIX
    static_assert(kExtendWaves == 8, "the sixteen counts are one DPP row");
    uint32_t v = lane < 16u ? cnt[lane] : 0u;
    // row_shr:n within the row of 16 lanes, zero shifted in (bound_ctrl): inclusive prefix sums of the row
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, true));
    hit_total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 7));
    const uint32_t all = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 15));
    // (lane wave + 7 holds hits of all waves + misses of waves < wave; lane wave - 1 the hits of waves < wave; wave 0: lane 7 / nothing)
    const uint32_t m_incl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(wave + 7u)));
    const uint32_t h_incl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>((wave + 15u) & 15u)));
    hit_before = wave ? h_incl : 0u;
    miss_before = m_incl - hit_total;
    miss_total = all - hit_total;
#else
    hit_before = miss_before = hit_total = miss_total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kExtendWaves; ++w) {
        const uint32_t hc = uniform(cnt[w]), mc = uniform(cnt[kExtendWaves + w]);
        hit_before += (w < wave) ? hc : 0u;
        miss_before += (w < wave) ? mc : 0u;
        hit_total += hc;
        miss_total += mc;
    }
#endif
}

// Persistent workgroups of 512 threads (8 waves): stage the scene in LDS once, then trace queue
// segments of 512 rays handed out by an atomic ticket. A segment's hits / misses are compacted in
// thread order into the matching segment of the hit / miss queues with wave64 ballots + mbcnt and one
// LDS exchange of the eight per-wave counts; no global atomics on queue slots (ex:59,61 use one per ray).
// LDS_SCENE = false (build extension for scenes larger than a CU's LDS, e.g. BASELINE config 5's 1M-triangle
// BVH): nodes, primitives and a 32-bit parent table are read from HBM / Infinity Cache through L2 instead.
template <bool HAS_INACTIVE, typename Trail, int PRIM, bool LDS_SCENE, bool EXACT>
#ifndef WFPT_EXTEND_MIN_WAVES
#define WFPT_EXTEND_MIN_WAVES 8
#endif
// min 8 waves per SIMD = 4 workgroups per CU: caps the SGPR count (without it hipcc uses 106 SGPRs and only 3 fit)
__global__ __launch_bounds__(kExtendThreads, WFPT_EXTEND_MIN_WAVES) void extend_kernel(ExtendArgs a) {
    extern __shared__ float4 lds[];
    constexpr uint32_t kGeomWords = PRIM == 0 ? 1u : 3u; // float4 per primitive
    float4 *s_nodes = lds;
    float4 *s_sphere = lds + (LDS_SCENE ? 2u * a.scene.n_nodes : 0u);
    const uint32_t parent_words = LDS_SCENE ? ((a.scene.n_nodes / 2u + 1u) + 7u) / 8u : 0u; // uint4 words of 8 u16 entries
    const uint32_t geom_words = LDS_SCENE ? kGeomWords * a.scene.n_spheres : 0u;
    uint16_t *s_parent = reinterpret_cast<uint16_t *>(s_sphere + geom_words);
    uint32_t *s_misc = reinterpret_cast<uint32_t *>(s_sphere + geom_words + parent_words);
    // s_misc: [2][2][kExtendWaves] wave counts, [2] next work item, [kMaxBatchClassic] rays per sample,
    //         [kMaxBatchClassic + 1] first work item of each sample
    uint32_t *s_next = s_misc + 4 * kExtendWaves;
    uint32_t *s_rays = s_next + 2;
    uint32_t *s_first = s_rays + kMaxBatchClassic;
    uint32_t *s_mat = s_first + kMaxBatchClassic + 1; // [2][3][kExtendWaves] per-material wave counts
    uint32_t *s_stack = s_mat + 6 * kExtendWaves; // HBM-resident scenes: [2 * kStackDepth][kExtendThreads] node stack (node, packed fields)

    // Work items are (sample, segment) pairs, numbered sample-major. The samples' counters are read side by side (one
    // thread each); the prefix over them then runs out of LDS.
    if (threadIdx.x < a.batch.n) s_rays[threadIdx.x] = umin(a.n_in[static_cast<size_t>(threadIdx.x) * a.batch.ctl_stride], a.limit);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t smp = 0; smp < a.batch.n; ++smp) {
            s_first[smp] = total;
            total += (s_rays[smp] + kChunk - 1) / kChunk;
        }
        s_first[a.batch.n] = total;
    }
    __syncthreads();
    const uint32_t n_items = s_first[a.batch.n];
    uint32_t item = blockIdx.x;
    if (item >= n_items) return; // nothing to do: skip the LDS staging too
    const float4 *g_nodes = reinterpret_cast<const float4 *>(a.scene.nodes);
    if (LDS_SCENE) {
        const float4 *g_staged = EXACT ? g_nodes : a.scene.nodes_ch; // reference boxes, or conservative centre / half-extent boxes
        for (uint32_t i = threadIdx.x; i < 2u * a.scene.n_nodes; i += kExtendThreads) s_nodes[i] = g_staged[i];
        for (uint32_t i = threadIdx.x; i < geom_words; i += kExtendThreads) s_sphere[i] = a.scene.prim_geom[i];
        const uint4 *g_par = reinterpret_cast<const uint4 *>(a.scene.pair_parent);
        uint4 *s_par4 = reinterpret_cast<uint4 *>(s_parent);
        for (uint32_t i = threadIdx.x; i < parent_words; i += kExtendThreads) s_par4[i] = g_par[i];
        __syncthreads();
    }

    const uint32_t lane = lane_id(), wave = uniform(threadIdx.x >> 6); // (a scalar: what is selected or summed per wave below runs on the scalar unit)
    uint32_t iter = 0;
    while (item < n_items) {
        const uint32_t buf = iter & 1u;
        if (threadIdx.x == 0) s_next[buf] = gridDim.x + atomicAdd(&a.ctl->ticket, 1u);
        uint32_t smp = 0;
        while (item >= uniform(s_first[smp + 1])) ++smp; // at most batch.n - 1 steps, block-uniform (scalars: uniform())
        const uint32_t chunk = item - uniform(s_first[smp]);
        const uint32_t n = uniform(s_rays[smp]);
        const RayQueue q = slice(a.q, smp * a.batch.ray_stride);
        const size_t qo = smp * a.batch.queue_stride, co = smp * a.batch.chunk_stride;

        const uint32_t idx = chunk * kChunk + threadIdx.x; // ex:51
        bool live = idx < n;                               // ex:53
        float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
        if (live) {
            ox = q.ox()[idx]; oy = q.oy()[idx]; oz = q.oz()[idx];
            dx = q.dx()[idx]; dy = q.dy()[idx]; dz = q.dz()[idx];
            if (HAS_INACTIVE) live = q.pixel()[idx] != WFPT_INACTIVE_PIXEL;
        }
        float t = 0.0f;
        uint32_t prim = 0;
        bool hit = false;
        if (live) {
            if (LDS_SCENE && EXACT)
                hit = trace_ray<Trail, PRIM, uint16_t, 0, true>(s_nodes, s_sphere, s_parent, nullptr, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t,
                                                                prim);
            else if (LDS_SCENE) {
                prim = kHandOver;
                if (!far_origin(a.scene, ox, oy, oz))
                {
#if WFPT_STAMPS
                    uint32_t dbg[3] = {0, 0, 0};
#endif
                    hit = trace_ray_conservative<Trail, PRIM, uint16_t>(s_nodes, s_sphere, s_parent, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim WFPT_DBG_ARG);
                }
                if (prim == kHandOver) // near-tie, probe or far origin: the reference's own walk decides (its boxes are read from global memory: this is rare)
                    hit = retrace_reference<Trail, PRIM, uint16_t>(g_nodes, s_sphere, s_parent, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim);
            } else if (!EXACT && a.scene.nodes4) {
                Stack4 st;
                st.init(WFPT_LDS_BYTES(lds, s_stack), a.scene.stack_spill, a.scene.spill_stride);
                prim = kHandOver;
                if (!far_origin(a.scene, ox, oy, oz))
                    hit = trace_ray4<PRIM>(a.scene.nodes4, a.scene.prim_geom, st, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, a.scene.root_leaf != 0, t, prim);
                if (prim == kHandOver)
                    hit = retrace_reference<Trail, PRIM, uint32_t>(g_nodes, a.scene.prim_geom, a.scene.pair_parent32, ox, oy, oz, dx, dy, dz,
                                                                    a.scene.n_nodes, t, prim);
            } else
                hit = trace_ray<Trail, PRIM, uint32_t, kStackDepth, EXACT>(g_nodes, a.scene.prim_geom, a.scene.pair_parent32,
                                                                            s_stack + threadIdx.x, ox, oy, oz, dx, dy, dz,
                                                                            a.scene.n_nodes, t, prim);
        }
        const bool miss = live && !hit;
        const unsigned long long hit_mask = __ballot(hit), miss_mask = __ballot(miss);
        if (lane == 0) {
            s_misc[(buf * 2 + 0) * kExtendWaves + wave] = static_cast<uint32_t>(__popcll(hit_mask));
            s_misc[(buf * 2 + 1) * kExtendWaves + wave] = static_cast<uint32_t>(__popcll(miss_mask));
        }
        // material class of each hit (sphere.material_type, ex:199; `case 0u, default` of sh:102 folds > 2 into 0)
        uint32_t mclass = 3u;
        unsigned long long mat_mask[3] = {0, 0, 0};
        if (a.partition) {
            if (hit) {
                mclass = PRIM == 0 ? a.scene.spheres[prim].material_type : __float_as_uint(a.scene.prim_geom[3u * prim + 1u].w);
                if (mclass > 2u) mclass = 0u;
            }
#pragma unroll
            for (uint32_t m = 0; m < 3; ++m) {
                mat_mask[m] = __ballot(mclass == m);
                if (lane == 0) s_mat[(buf * 3 + m) * kExtendWaves + wave] = static_cast<uint32_t>(__popcll(mat_mask[m]));
            }
        }
        __syncthreads();
        uint32_t hit_before, miss_before, hit_total, miss_total;
        wave_offsets(s_misc + buf * 2u * kExtendWaves, wave, lane, hit_before, hit_total, miss_before, miss_total);
        const size_t seg = qo + static_cast<size_t>(chunk) * kChunk;
        // (the wave's first slot is a scalar address; a lane adds its 32-bit rank: no 64-bit vector arithmetic per store)
        if (hit) { // ex:57-59: payload (t, ray_idx, sphere_idx), slot = rank in thread order
            const size_t first = seg + hit_before;
            const uint32_t rank = mbcnt(hit_mask);
            (a.hq.t() + first)[rank] = t;
            (a.hq.prim() + first)[rank] = prim;
            (a.hq.ridx() + first)[rank] = idx;
            if (a.rec_out) { // what shade will need of this hit, in one place: p = origin + t * direction (sh:91), the direction, pixel, primitive
                float4 *rec = a.rec_out + 2u * first;
                rec[2u * rank] = make_float4(ox + t * dx, oy + t * dy, oz + t * dz, __uint_as_float(q.pixel()[idx]));
                rec[2u * rank + 1u] = make_float4(dx, dy, dz, __uint_as_float(prim));
            }
        }
        if (miss) { // ex:61, plus what miss_kernel reads of the ray (mk:29-32)
            const size_t first = seg + miss_before;
            const uint32_t rank = mbcnt(miss_mask);
            (a.mq.ridx() + first)[rank] = idx;
            (a.mq.dy() + first)[rank] = dy;
            (a.mq.pixel() + first)[rank] = q.pixel()[idx];
        }
        if (threadIdx.x == 0) {
            a.chunk_hits[co + chunk] = hit_total;
            a.chunk_miss[co + chunk] = miss_total;
        }
        if (a.partition) {
#pragma unroll
            for (uint32_t m = 0; m < 3; ++m) {
                uint32_t before, total;
                {
                    uint32_t v = lane < kExtendWaves ? s_mat[(buf * 3 + m) * kExtendWaves + lane] : 0u; // inclusive scan over the row (see wave_offsets)
                    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, true));
                    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, true));
                    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, true));
                    total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 7));
                    const uint32_t incl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>((wave + 15u) & 15u)));
                    before = wave ? incl : 0u;
                }
                if (mclass == m) // entry = the hit's rank within the segment's hit queue; lists stay ascending
                    a.mat_list[m * a.mat_list_mstride + seg + before + mbcnt(mat_mask[m])] =
                        static_cast<uint16_t>(hit_before + mbcnt(hit_mask));
                if (threadIdx.x == 0) a.chunk_mat[m * a.chunk_mat_mstride + co + chunk] = total;
            }
        }
        item = uniform(s_next[buf]);
        iter += 1;
    }
}

// ================================================================================================
// scan: one workgroup. Exclusive prefix of the per-segment hit / miss counts (=> queue positions in
// ascending thread order, i.e. the order the oracle resolves ex:59,61's atomicAdd in), the counter
// protocol of pt:327-353, and for the fused loop the `misses < miss_floor` exit (pt:332) on the device.
// ================================================================================================
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d, 64);
        if (static_cast<int>(lane_id()) >= d) v += up;
    }
    return v;
}

__global__ __launch_bounds__(kScanThreads) void scan_kernel(ScanArgs a) {
    __shared__ uint32_t s_wave[2][kScanThreads / 64];
    __shared__ uint32_t s_fac;
    const uint32_t sample = blockIdx.x;
    a.chunk_hits += sample * a.batch.chunk_stride;
    a.chunk_miss += sample * a.batch.chunk_stride;
    a.chunk_hit_base += sample * a.batch.chunk_stride;
    a.chunk_miss_base += sample * a.batch.chunk_stride;
    if (a.first_seg) a.first_seg += sample * a.batch.chunk_stride;
    const uint32_t n = umin(a.n_in[static_cast<size_t>(sample) * a.batch.ctl_stride], a.limit);
    const uint32_t n_chunks = (n + kChunk - 1) / kChunk;
    const uint32_t lane = lane_id(), wave = uniform(threadIdx.x >> 6); // (a scalar: what is selected or summed per wave below runs on the scalar unit)
    uint32_t carry_h = 0, carry_m = 0;
    for (uint32_t base = 0; base < n_chunks; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t vh = i < n_chunks ? a.chunk_hits[i] : 0u;
        const uint32_t vm = i < n_chunks ? a.chunk_miss[i] : 0u;
        const uint32_t ih = wave_inclusive_scan(vh), im = wave_inclusive_scan(vm);
        if (lane == 63) { s_wave[0][wave] = ih; s_wave[1][wave] = im; }
        __syncthreads();
        uint32_t before_h = 0, before_m = 0, tile_h = 0, tile_m = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
            const uint32_t h = s_wave[0][w], m = s_wave[1][w];
            before_h += (w < wave) ? h : 0u;
            before_m += (w < wave) ? m : 0u;
            tile_h += h;
            tile_m += m;
        }
        if (i < n_chunks) {
            const uint32_t base_h = carry_h + before_h + ih - vh;
            a.chunk_hit_base[i] = base_h;
            a.chunk_miss_base[i] = carry_m + before_m + im - vm;
            // the fused bounce kernel cuts the hit queue into runs of kChunk consecutive hits: remember which segment
            // holds the first hit of each run (a segment holds at most kChunk hits, so at most one multiple of kChunk)
            if (a.first_seg && vh > 0) {
                const uint32_t m = (base_h + kChunk - 1) / kChunk;
                if (m * kChunk < base_h + vh) a.first_seg[m] = i;
            }
        }
        carry_h += tile_h;
        carry_m += tile_m;
        __syncthreads();
    }
    const uint32_t hits = carry_h, misses = carry_m;

    // x extent of workgroup_size_64(hits) (pt:282-289), the dispatch shape shade.wgsl:72 keys its RNG on
    const uint32_t q = (hits + 63u) / 64u;
    uint32_t gx = 1;
    if (q > 1) {
        const uint32_t y = static_cast<uint32_t>(__builtin_ceilf(sqrt_(static_cast<float>(q))));
        if (threadIdx.x == 0) s_fac = 1u;
        __syncthreads();
        for (int z = static_cast<int>(y) - 1 - static_cast<int>(threadIdx.x); z >= 1; z -= kScanThreads) {
            if (q % static_cast<uint32_t>(z) == 0u) {
                atomicMax(&s_fac, static_cast<uint32_t>(z));
                break;
            }
        }
        __syncthreads();
        const uint32_t fac = s_fac;
        gx = (q / fac >= (1u << 16)) ? y : fac;
    }

    if (threadIdx.x == 0) {
        Control *c = a.ctl + sample;
        c->seg_n = n;
        c->hits = hits;
        c->misses = misses;
        c->shade_gx = gx;
        if (sample == 0) c->ticket = 0; // extend's work-item ticket lives in the first Control block
        if (a.fused) {
            uint32_t done = (a.bounce == 0) ? 0u : c->done;
            const uint32_t ran = done ? 0u : 1u;       // this wavefront's extend really ran
            if (!done && misses < a.miss_floor) done = 1; // pt:332: exit before shading
            c->done = done;
            c->shade_n = done ? 0u : hits;
            c->miss_n = done ? 0u : misses;
            c->n_in = done ? 0u : hits; // every shaded hit emits exactly one extension ray (sh:155)
            c->counters[0] = misses;
            c->counters[1] = hits;
            c->counters[2] = done ? n : 0u; // pt:335-336
            if (a.bounce < kMaxRows) {
                c->rows[a.bounce][0] = ran ? n : 0u;
                c->rows[a.bounce][1] = hits;
                c->rows[a.bounce][2] = misses;
                c->rows[a.bounce][3] = (ran && !done) ? 1u : 0u;
            }
            c->bounce = a.bounce + 1;
        } else {
            c->counters[1] += hits;   // ex:59
            c->counters[0] += misses; // ex:61
        }
    }
}

// ================================================================================================
// shade (sh:56-176)
// ================================================================================================
__device__ __forceinline__ float schlick(float cosine, float refraction_index) { // sh:158-162
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow_(1.0f - cosine, 5.0f);
}
__device__ __forceinline__ float3_ reflect(float3_ r, float3_ n) { // sh:164-166
    const float k = 2.0f * dot3(r, n);
    return {r.x - k * n.x, r.y - k * n.y, r.z - k * n.z};
}

// sh:71-73: shade's RNG state for hit `h` of a dispatch whose x extent is `gx` workgroups (keyed by the dispatch's
// global_invocation_id), or keyed by the ray's own pixel (WFPT_RNG_PIXEL).
// WAVE_H: the wave's lanes hold 64 consecutive hits h of one 64-aligned group (the fused bounce kernel), so the dispatch's workgroup
// index and its division by gx are scalars.
template <bool WAVE_H = false>
__device__ __forceinline__ uint32_t shade_rng(uint32_t rng_mode, uint32_t h, uint32_t gx, uint32_t pixel_idx, wfpt_frame_buffer fb) {
    uint32_t id_x, id_y;
    if (rng_mode == WFPT_RNG_PIXEL) {
        id_y = pixel_idx / fb.width;
        id_x = pixel_idx - id_y * fb.width;
    } else {
        const uint32_t wg = WAVE_H ? uniform(h >> 6) : h >> 6, li = h & 63u;
        const uint32_t wgy = wg / gx;
        id_x = (wg - wgy * gx) * 8u + (li & 7u);
        id_y = wgy * 8u + (li >> 3);
    }
    const uint32_t rng = init_rng(id_x, id_y, fb.width, fb.frame);
    return advance(rng, fb.sample_number * 10u);
}

// sh:93-151: the extension direction of one hit at point p of primitive record (rec0 = centre | normal, fuzz;
// rec1 = albedo, refraction index) for the incoming direction rdir (not normalised after bounce 0). Shared by
// shade_kernel and the fused bounce kernel so that both produce the same bits.
__device__ __forceinline__ float3_ scatter(uint32_t rng, float3_ p, float3_ rdir, float4 rec0, float4 rec1, uint32_t mat_type,
                                           uint32_t prim_kind) {
    const float fuzz = rec0.w, refract_index = rec1.w;
    // spheres: always-outward normal (sh:93); triangles: normalize(cross(e1, e2)), never flipped
    const float3_ nrm = prim_kind == 0 ? normalize3({p.x - rec0.x, p.y - rec0.y, p.z - rec0.z}) : float3_{rec0.x, rec0.y, rec0.z};
    float3_ ext;
    // metal and lambertian both start with normalize(rng_next_in_unit_sphere) (sh:104, 111): drawn once for the lanes of either
    // kind, so a wave that holds both materials runs the sampler (pow, sqrt, sin / cos, three divisions) once, not twice
    float3_ rb = {0.0f, 0.0f, 0.0f};
    if (mat_type != 2u) rb = normalize3(rng_next_in_unit_sphere(rng));
    if (mat_type == 1u) { // sh:110-114 metal
        const float3_ rf = reflect(rdir, nrm);
        ext = {rf.x + fuzz * rb.x, rf.y + fuzz * rb.y, rf.z + fuzz * rb.z};
    } else if (mat_type == 2u) { // sh:115-151 dielectric
        float3_ norm = nrm;
        const float3_ uv = normalize3(rdir);
        float cos_theta = min_(dot3(norm, {-uv.x, -uv.y, -uv.z}), 1.0f);
        float eta;
        if (cos_theta >= 0.0f) {
            eta = 1.0f / refract_index;
        } else {
            eta = refract_index;
            norm = {norm.x * -1.0f, norm.y * -1.0f, norm.z * -1.0f};
            cos_theta = cos_theta * -1.0f;
        }
        const float reflectance = schlick(cos_theta, eta);
        // refract(), sh:168-176
        const float ct = dot3(uv, norm);
        const float k = 1.0f - eta * eta * (1.0f - ct * ct);
        if (k >= 0.0f) {
            if (reflectance > rng_next_float(rng)) {
                ext = reflect(uv, norm);
            } else {
                const float m = eta * ct + sqrt_(k);
                ext = {eta * uv.x - m * norm.x, eta * uv.y - m * norm.y, eta * uv.z - m * norm.z};
            }
        } else {
            ext = reflect(uv, norm);
        }
    } else { // sh:102-109 lambertian (case 0u, default)
        ext = {nrm.x + rb.x, nrm.y + rb.y, nrm.z + rb.z};
        if (sqrt_(dot3(ext, ext)) < 0.001f) ext = nrm;
    }
    return ext;
}

// Walks the hit-queue segments: segment c holds chunk_hits[c] hits compacted at its front, and
// chunk_hit_base[c] is the queue position of its first hit, so the logical hit index (the thread index
// of the reference's shade dispatch) is base + rank. Whole waves beyond a segment's count skip, so lanes
// stay packed without a global compaction pass. Extension rays go to slot = logical hit index, which is
// where ascending-order resolution of sh:155's atomicAdd puts them: the next ray queue is compact and
// keeps the previous order (neighbouring pixels stay neighbours).
// (capping the SGPR count with a min-waves launch bound, as extend does, was measured 2 % SLOWER here: the spills cost
// more than the seventh and eighth wave per SIMD bring)
__global__ __launch_bounds__(kConsumerThreads) void shade_kernel(ShadeArgs a) {
    const uint32_t sample = blockIdx.y;
    wfpt_frame_buffer fb = a.ctl->frame;
    fb.frame += sample;
    a.ctl += sample;
    a.q = slice(a.q, sample * a.batch.ray_stride);
    a.ext = slice(a.ext, sample * a.batch.ray_stride);
    a.hq.base += sample * a.batch.queue_stride;
    a.chunk_hits += sample * a.batch.chunk_stride;
    a.chunk_hit_base += sample * a.batch.chunk_stride;
    a.image += sample * a.batch.image_stride;
    const uint32_t n_hits = umin(a.n_hits[static_cast<size_t>(sample) * a.batch.ctl_stride], a.limit); // sh:66
    const uint32_t n_chunks = (a.ctl->seg_n + kChunk - 1) / kChunk;
    const uint32_t gx = a.gx ? a.gx : a.ctl->shade_gx;
    // split: this workgroup shades one material class, walking that class's per-segment lists (lanes stay
    // packed and every lane takes the same branch of the material switch)
    const bool split = a.split != 0;
    const uint32_t mclass = a.material != 0xffffffffu ? a.material : blockIdx.z;
    const uint16_t *mat_list = a.mat_list + mclass * a.mat_list_mstride + sample * a.batch.queue_stride;
    const uint32_t *chunk_mat = a.chunk_mat + mclass * a.chunk_mat_mstride + sample * a.batch.chunk_stride;
    if (a.count_out && !split && blockIdx.x == 0 && threadIdx.x == 0) a.ctl->counters[2] += n_hits; // sh:155
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t count = split ? chunk_mat[chunk] : a.chunk_hits[chunk];
        const uint32_t base = a.chunk_hit_base[chunk];
        if (base >= n_hits) break; // bases ascend with the segment index
        if (a.rec_in) {
            // ---- streaming form: extend left (hit point | pixel), (direction | primitive) at the hit's slot; the next hit's record is
            // loaded before this hit's dependent gathers (shade record, throughput)
            const float4 *rec = a.rec_in + 2u * (sample * static_cast<size_t>(a.batch.queue_stride) + static_cast<size_t>(chunk) * kChunk);
            uint32_t r0 = threadIdx.x, nr = 0;
            float4 nra = make_float4(0, 0, 0, 0), nrb = nra;
            if (r0 < count) {
                nr = split ? mat_list[chunk * kChunk + r0] : r0;
                nra = rec[2u * nr]; nrb = rec[2u * nr + 1u];
            }
            for (; r0 < count; r0 += kConsumerThreads) {
                const uint32_t r = nr;
                const float4 ra = nra, rb = nrb;
                const uint32_t h = base + r; // the reference's shade thread index
                if (h >= n_hits) break;
                if (r0 + kConsumerThreads < count) {
                    nr = split ? mat_list[chunk * kChunk + r0 + kConsumerThreads] : r0 + kConsumerThreads;
                    nra = rec[2u * nr]; nrb = rec[2u * nr + 1u];
                }
                if (split && a.count_out) { // per-material stage of the stage API: counters[2] += rays this stage emits
                    const unsigned long long m = __ballot(true);
                    if (lane_id() == static_cast<uint32_t>(__ffsll(static_cast<long long>(m)) - 1))
                        atomicAdd(&a.ctl->counters[2], static_cast<uint32_t>(__popcll(m)));
                }
                const uint32_t prim = __float_as_uint(rb.w), pixel_idx = __float_as_uint(ra.w);
                const float4 rec0 = a.scene.shade_rec[3u * prim], rec1 = a.scene.shade_rec[3u * prim + 1u], rec2 = a.scene.shade_rec[3u * prim + 2u];
                float4 *px = pixel_of(a.image, local_pixel(pixel_idx, a.image_width, a.tile));
                const float4 thr = *px; // sh:84-87: throughput *= albedo, for every material type (load now, store after the scatter math)
                const uint32_t rng = shade_rng(a.rng_mode, h, gx, pixel_idx, fb);
                const float3_ ext = scatter(rng, {ra.x, ra.y, ra.z}, {rb.x, rb.y, rb.z}, rec0, rec1, __float_as_uint(rec2.x), a.scene.prim_kind);
                a.ext.ox()[h] = ra.x; a.ext.oy()[h] = ra.y; a.ext.oz()[h] = ra.z; // sh:153-155
                a.ext.dx()[h] = ext.x; a.ext.dy()[h] = ext.y; a.ext.dz()[h] = ext.z;
                a.ext.pixel()[h] = pixel_idx;
                *px = make_float4(thr.x * rec1.x, thr.y * rec1.y, thr.z * rec1.z, thr.w); // albedo
            }
            continue;
        }
        // Software-pipelined walk: the queue entry of the NEXT iteration is loaded before this iteration's
        // dependent gathers, so each hit costs two dependent memory levels instead of three.
        uint32_t r0 = threadIdx.x;
        uint32_t nr = 0, nprim = 0, nridx = 0;
        float nt = 0.0f;
        if (r0 < count) {
            nr = split ? mat_list[chunk * kChunk + r0] : r0; // rank within the segment's hit queue
            nt = a.hq.t()[chunk * kChunk + nr];
            nprim = a.hq.prim()[chunk * kChunk + nr];
            nridx = a.hq.ridx()[chunk * kChunk + nr];
        }
        for (; r0 < count; r0 += kConsumerThreads) {
            const uint32_t r = nr, prim = nprim, ridx = nridx;
            const float t = nt;
            const uint32_t h = base + r; // the reference's shade thread index
            if (h >= n_hits) break;
            if (r0 + kConsumerThreads < count) {
                nr = split ? mat_list[chunk * kChunk + r0 + kConsumerThreads] : r0 + kConsumerThreads;
                nt = a.hq.t()[chunk * kChunk + nr];
                nprim = a.hq.prim()[chunk * kChunk + nr];
                nridx = a.hq.ridx()[chunk * kChunk + nr];
            }
            if (split && a.count_out) { // per-material stage of the stage API: counters[2] += rays this stage emits
                const unsigned long long m = __ballot(true);
                if (lane_id() == static_cast<uint32_t>(__ffsll(static_cast<long long>(m)) - 1))
                    atomicAdd(&a.ctl->counters[2], static_cast<uint32_t>(__popcll(m)));
            }
            // one gather: primitive centre / normal, albedo, fuzz, refraction index, material type (== payload.mat_type, ex:199)
            const float4 rec0 = a.scene.shade_rec[3u * prim], rec1 = a.scene.shade_rec[3u * prim + 1u],
                         rec2 = a.scene.shade_rec[3u * prim + 2u];
            const uint32_t mat_type = __float_as_uint(rec2.x);
            const float ox = a.q.ox()[ridx], oy = a.q.oy()[ridx], oz = a.q.oz()[ridx];
            const float dx = a.q.dx()[ridx], dy = a.q.dy()[ridx], dz = a.q.dz()[ridx];
            const uint32_t pixel_idx = a.q.pixel()[ridx];
            // sh:84-87: throughput *= albedo, for every material type (load now, store after the scatter math)
            float4 *px = pixel_of(a.image, local_pixel(pixel_idx, a.image_width, a.tile));
            const float4 thr = *px;

            const uint32_t rng = shade_rng(a.rng_mode, h, gx, pixel_idx, fb);
            // sh:91-93
            const float p_x = ox + t * dx, p_y = oy + t * dy, p_z = oz + t * dz;
            const float3_ ext = scatter(rng, {p_x, p_y, p_z}, {dx, dy, dz}, rec0, rec1, mat_type, a.scene.prim_kind);
            // sh:153-155: direction is NOT normalised; invDirection is recomputed by extend
            a.ext.ox()[h] = p_x; a.ext.oy()[h] = p_y; a.ext.oz()[h] = p_z;
            a.ext.dx()[h] = ext.x; a.ext.dy()[h] = ext.y; a.ext.dz()[h] = ext.z;
            a.ext.pixel()[h] = pixel_idx;
            *px = make_float4(thr.x * rec1.x, thr.y * rec1.y, thr.z * rec1.z, thr.w); // albedo
        }
    }
}

// ================================================================================================
// miss_kernel (mk:13-38)
// ================================================================================================
__global__ __launch_bounds__(kConsumerThreads) void miss_kernel(MissArgs a) {
    const uint32_t sample = blockIdx.y;
    a.ctl += sample;
    a.q = slice(a.q, sample * a.batch.ray_stride);
    a.mq.base += sample * a.batch.queue_stride;
    a.chunk_miss += sample * a.batch.chunk_stride;
    a.chunk_miss_base += sample * a.batch.chunk_stride;
    a.image += sample * a.batch.image_stride;
    const uint32_t n_miss = umin(a.n_miss[static_cast<size_t>(sample) * a.batch.ctl_stride], a.limit); // mk:24
    const uint32_t n_chunks = (a.ctl->seg_n + kChunk - 1) / kChunk;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t count = a.chunk_miss[chunk];
        const uint32_t base = a.chunk_miss_base[chunk];
        if (base >= n_miss) break;
        // software-pipelined like shade: the next entry is loaded before this entry's image read-modify-write
        float ndy = 0.0f;
        uint32_t npix = 0;
        if (threadIdx.x < count) {
            ndy = a.mq.dy()[chunk * kChunk + threadIdx.x]; // ray_buffer[miss_buffer[idx]].direction.y, mk:28-32
            npix = a.mq.pixel()[chunk * kChunk + threadIdx.x];
        }
        for (uint32_t r = threadIdx.x; r < count; r += kConsumerThreads) {
            if (base + r >= n_miss) break;
            const float dy = ndy;
            const uint32_t pixel_idx = npix;
            if (r + kConsumerThreads < count) {
                ndy = a.mq.dy()[chunk * kChunk + r + kConsumerThreads];
                npix = a.mq.pixel()[chunk * kChunk + r + kConsumerThreads];
            }
            const float t = 0.5f * (dy + 1.0f); // mk:32: the direction is not normalised after bounce 0
            const float om = 1.0f - t;
            const float cr = om * 1.0f + t * 0.5f; // mk:33
            const float cg = om * 1.0f + t * 0.7f;
            const float cb = om * 1.0f + t * 1.0f;
            float4 *px = pixel_of(a.image, local_pixel(pixel_idx, a.image_width, a.tile));
            const float4 thr = *px;
            *px = make_float4(thr.x * cr, thr.y * cg, thr.z * cb, thr.w); // mk:35-37
        }
    }
}

// shade (sh:56-156) of hit h of the previous wavefront, as the fused loop runs it: find the hit's path record (its
// segment is the last one whose first hit is not after h, searched between the segments that hold the first hit of
// this run of kChunk hits and of the next run: scan's first_seg table), multiply the pixel's throughput by the albedo
// (sh:84-87) and, if SCATTER, produce the extension ray (origin = hit point, direction NOT normalised, sh:153-155).
// Shared by the fused bounce kernel and the refill traversal.
struct HitSource {
    const float4 *rec_in;
    const uint32_t *in_hit_base, *in_first_seg;
    const Control *ctl;
    float *image; // this sample's slice
    const float4 *shade_rec;
    size_t qo, co; // this sample's offsets into the record queue and the per-segment tables
    uint32_t capacity, rng_mode, image_width, prim_kind;
    Tiling tile;
};
// shade of one path record (ra = hit point | pixel, rb = incoming direction | -), primitive `prim`, as the reference's shade thread h.
// WAVE_SMP: the wave's lanes belong to one sample (its counters are scalars); WAVE_H: they hold 64 consecutive, 64-aligned h (shade_rng).
template <bool SCATTER, bool WAVE_SMP, bool WAVE_H>
__device__ __forceinline__ void shade_record(const HitSource &s, float4 ra, float4 rb, uint32_t prim, uint32_t h, wfpt_frame_buffer fb, float &ox,
                                             float &oy, float &oz, float &dx, float &dy, float &dz, uint32_t &pixel_idx) {
    pixel_idx = __float_as_uint(ra.w);
    const float4 rec1 = s.shade_rec[3u * prim + 1u];
    float4 *px = pixel_of(s.image, local_pixel(pixel_idx, s.image_width, s.tile));
    const float4 thr = *px;
    if (SCATTER) {
        const float4 rec0 = s.shade_rec[3u * prim], rec2 = s.shade_rec[3u * prim + 2u];
        const uint32_t gx = WAVE_SMP ? uniform(s.ctl->shade_gx) : s.ctl->shade_gx;
        const uint32_t rng = shade_rng<WAVE_H>(s.rng_mode, h, gx, pixel_idx, fb);
        const float3_ ext = scatter(rng, {ra.x, ra.y, ra.z}, {rb.x, rb.y, rb.z}, rec0, rec1, __float_as_uint(rec2.x), s.prim_kind);
        ox = ra.x; oy = ra.y; oz = ra.z;
        dx = ext.x; dy = ext.y; dz = ext.z;
    }
    *px = make_float4(thr.x * rec1.x, thr.y * rec1.y, thr.z * rec1.z, thr.w); // sh:84-87: throughput *= albedo, for every material type
}
// WAVE_RUN: every lane of the wave shades a hit of the same run of kChunk hits and the same sample (the fused bounce kernel), so the
// run's segment bounds and the sample's counters are scalars.
#if WFPT_STAMPS
#define WFPT_SHADE_STAMPS_PARAM , unsigned long long *shade_stamps = nullptr
#else
#define WFPT_SHADE_STAMPS_PARAM
#endif
template <bool SCATTER, bool WAVE_RUN = false>
__device__ __forceinline__ void shade_hit(const HitSource &s, uint32_t h, uint32_t n_hits, wfpt_frame_buffer fb, float &ox, float &oy,
                                          float &oz, float &dx, float &dy, float &dz, uint32_t &pixel_idx WFPT_SHADE_STAMPS_PARAM) {
    const uint32_t run = WAVE_RUN ? uniform(h / kChunk) : h / kChunk, n_runs = (n_hits + kChunk - 1) / kChunk;
    uint32_t lo = s.in_first_seg[s.co + run];
    uint32_t hi = run + 1 < n_runs ? s.in_first_seg[s.co + run + 1] : (umin(s.ctl->seg_n, s.capacity) + kChunk - 1) / kChunk - 1u;
    if (WAVE_RUN) { lo = uniform(lo); hi = uniform(hi); }
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (s.in_hit_base[s.co + mid] <= h) lo = mid; else hi = mid - 1u;
    }
    const size_t slot = s.qo + static_cast<size_t>(lo) * kChunk + (h - s.in_hit_base[s.co + lo]);
#if WFPT_STAMPS
    if (shade_stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); shade_stamps[0] = stamp_now(); } // the record's slot is known
#endif
    const float4 ra = s.rec_in[2u * slot], rb = s.rec_in[2u * slot + 1u];
#if WFPT_STAMPS
    if (shade_stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); shade_stamps[1] = stamp_now(); } // the record has arrived
#endif
    shade_record<SCATTER, WAVE_RUN, WAVE_RUN>(s, ra, rb, __float_as_uint(rb.w), h, fb, ox, oy, oz, dx, dy, dz, pixel_idx);
}

// ================================================================================================
// Fused bounce kernel of the device-resident loop. The reference's loop runs, per wavefront, extend -> (host reads
// the counters) -> shade -> miss_kernel -> copy extension rays back (pt:323-353). shade of wavefront b-1 and extend
// of wavefront b touch the same path one after the other, so here ONE launch per wavefront takes a hit of
// wavefront b-1, shades it (sh:56-156) and traces the extension ray straight from registers (ex:47-210): the
// extension-ray queue, its 28 B/ray write and 28 B/ray read, and two launches per wavefront disappear, and the
// latency-bound gathers of shade overlap with the VALU-bound traversal of the other waves. The queue between
// wavefronts is the hit queue, carried as 32-byte path records (hit point | pixel, incoming direction | primitive):
// shade streams them instead of gathering the ray through its index. miss_kernel (mk:13-38) of wavefront b-1 runs
// as extra work items of the same launch (the loop-exit test of pt:332 sits between the two wavefronts, so a
// wavefront's misses may only be applied once `scan` has decided that the loop goes on).
//
// Everything the reference's stages would compute is computed, in the same order per path, by the same device
// functions (primary_ray, shade_rng, scatter, trace_ray): the images are bit-identical to the stage-by-stage chain.
// shade's thread index h (its RNG key under WFPT_RNG_DISPATCH, sh:72) is the hit's position in the hit queue under
// ascending-order atomics, exactly as in shade_kernel: base[segment] + rank.
//
// MODE kBounceFirst : work item = 512 ray slots of generate_rays' numbering (8 tiles): generate -> trace.
//      kBounceMiddle: hit items = 512 consecutive hits h of the previous wavefront: shade -> trace; then miss items.
//      kBounceLast  : after the last extend: shade only multiplies the throughput (sh:84-87); miss items.
// ================================================================================================
// Tickets -> work items of a fused bounce launch. Hit items (shade + walk: vector-ALU work) come first in the numbering, miss items (one
// 16-byte read-modify-write per miss: memory latency, idle ALUs) after them; drawn in that order, every launch ended on a tail of nothing
// but miss items -- a tenth of its time (round 5: +9 % on the frame). A ticket is therefore mapped so that one miss item follows every
// `every - 1` hit items, with `every` chosen per launch so that the miss items spread over the whole of it (WFPT_MISS_EVERY fixes it). Both
// kinds keep their own ascending order (the kernels find an item's sample by walking on from the previous item's).
struct TicketMap {
    uint32_t n_hit, every, n_mixed; // (uniform)
    __device__ __forceinline__ TicketMap(uint32_t n_hit_items, uint32_t n_items, bool mix) : n_hit(n_hit_items) {
        const uint32_t n_miss = n_items - n_hit_items;
        every = kMissEvery ? kMissEvery : uniform(umin(umax(n_hit_items / umax(n_miss, 1u), 1u), 62u)) + 1u;
        n_mixed = mix ? umin(n_miss, n_hit_items / (every - 1u)) : 0u; // miss items placed among the hit items
    }
    __device__ __forceinline__ uint32_t item_of(uint32_t t) const {
        if (t < n_mixed * every) {
            const uint32_t k = t / every, r = t - k * every;
            return r == every - 1u ? n_hit + k : k * (every - 1u) + r;
        }
        const uint32_t u = t - n_mixed * every, rest_h = n_hit - n_mixed * (every - 1u);
        return u < rest_h ? n_mixed * (every - 1u) + u : n_hit + n_mixed + (u - rest_h);
    }
};

struct BounceLds {
    uint32_t *cnt;     // [2][2][kExtendWaves] per-wave hit / miss counts, double-buffered by iteration parity
    uint32_t *next;    // [2] next work item
    uint32_t *total;   // [2] hit items, all items of this launch
    uint16_t *items_h; // [kMaxBatch] hit work items of each sample (u16: the four LDS-resident scene copies of a CU leave ~1 KB)
    uint16_t *items_m; // [kMaxBatch] miss work items of each sample
    uint32_t *stack;   // HBM-resident scenes: [2 * kStackDepth][kExtendThreads]
};
constexpr uint32_t kBounceMiscWords = 4u * kExtendWaves + 2u + 2u + kMaxBatch; // the two u16 tables take kMaxBatch words
static_assert(kBounceMiscWords % 4u == 0, "the stack column area stays 16-byte aligned");

#ifndef WFPT_BOUNCE_ATTR
#define WFPT_BOUNCE_ATTR __launch_bounds__(kExtendThreads, WFPT_EXTEND_MIN_WAVES)
#endif
template <int MODE, typename Trail, int PRIM, bool LDS_SCENE, bool EXACT>
__global__ WFPT_BOUNCE_ATTR void bounce_kernel(BounceArgs a) {
    extern __shared__ float4 lds[];
    constexpr bool TRACE = MODE != kBounceLast;
    constexpr uint32_t kGeomWords = PRIM == 0 ? 1u : 3u;
    const bool stage_scene = TRACE && LDS_SCENE;
    float4 *s_nodes = lds;
    float4 *s_geom = lds + (stage_scene ? 2u * a.scene.n_nodes : 0u);
    const uint32_t parent_words = stage_scene ? ((a.scene.n_nodes / 2u + 1u) + 7u) / 8u : 0u;
    const uint32_t geom_words = stage_scene ? kGeomWords * a.scene.n_spheres : 0u;
    uint16_t *s_parent = reinterpret_cast<uint16_t *>(s_geom + geom_words);
    uint32_t *s_misc = reinterpret_cast<uint32_t *>(s_geom + geom_words + parent_words);
    BounceLds L;
    L.cnt = s_misc;
    L.next = L.cnt + 4 * kExtendWaves;
    L.total = L.next + 2;
    L.items_h = reinterpret_cast<uint16_t *>(L.total + 2);
    L.items_m = L.items_h + kMaxBatch;
    L.stack = s_misc + kBounceMiscWords;

    const uint32_t n_slots = a.gx * a.gy * 64u; // first wavefront: ray slots of this context's tiles
    // Work-item tables, one thread per sample: the counters of the samples are read side by side (a loop in one thread paid one
    // global-memory round trip per sample and launch: ~45 us at 32 samples in flight, on launches of 0.1-1 ms).
    // (a context has at most 65535 segments: wfpt_create keeps larger images on the stage-by-stage loop)
    if (threadIdx.x < 2) L.total[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < a.batch.n) {
        const uint32_t smp = threadIdx.x;
        const uint32_t n = MODE == kBounceFirst ? umin(n_slots, a.capacity) : umin(a.ctl[smp].shade_n, a.capacity);
        const uint32_t items_h = (n + kChunk - 1) / kChunk;
        uint32_t segs = 0;
        if (MODE != kBounceFirst && a.ctl[smp].miss_n > 0) segs = (umin(a.ctl[smp].seg_n, a.capacity) + kChunk - 1) / kChunk;
        const uint32_t items_m = (segs + kMissSegsPerItem - 1) / kMissSegsPerItem;
        L.items_h[smp] = static_cast<uint16_t>(items_h);
        L.items_m[smp] = static_cast<uint16_t>(items_m);
        atomicAdd(&L.total[0], items_h);           // hit items
        atomicAdd(&L.total[1], items_h + items_m); // all items of this launch
    }
    __syncthreads();
    const uint32_t n_hit_items = uniform(L.total[0]), n_items = uniform(L.total[1]);
    // a workgroup's tickets only grow, so the sample of an item is found by walking on from where the previous item was
    uint32_t smp_h = 0, first_h = 0, smp_m = 0, first_m = n_hit_items;
    if (MODE == kBounceFirst && blockIdx.x == 0 && threadIdx.x < a.batch.n)
        a.ctl[threadIdx.x].n_in = umin(n_slots, a.capacity); // pt:313-316: counter[2] = rays of the first wavefront (read by scan)
    const TicketMap tickets(n_hit_items, n_items, MODE != kBounceFirst);
    uint32_t ticket = blockIdx.x;
    if (ticket >= n_items) return;
    uint32_t item = tickets.item_of(ticket);
    const float4 *g_nodes = reinterpret_cast<const float4 *>(a.scene.nodes);
    if (stage_scene) {
        const float4 *g_staged = EXACT ? g_nodes : a.scene.nodes_ch; // reference boxes, or conservative centre / half-extent boxes
        for (uint32_t i = threadIdx.x; i < 2u * a.scene.n_nodes; i += kExtendThreads) s_nodes[i] = g_staged[i];
        for (uint32_t i = threadIdx.x; i < geom_words; i += kExtendThreads) s_geom[i] = a.scene.prim_geom[i];
        const uint4 *g_par = reinterpret_cast<const uint4 *>(a.scene.pair_parent);
        uint4 *s_par4 = reinterpret_cast<uint4 *>(s_parent);
        for (uint32_t i = threadIdx.x; i < parent_words; i += kExtendThreads) s_par4[i] = g_par[i];
        __syncthreads();
    }
    wfpt_frame_buffer fb0 = a.ctl->frame; // the same for every lane: kept in scalar registers
    fb0.width = uniform(fb0.width); fb0.height = uniform(fb0.height); fb0.frame = uniform(fb0.frame); fb0.sample_number = uniform(fb0.sample_number);
    const uint32_t lane = lane_id(), wave = uniform(threadIdx.x >> 6); // (a scalar: what is selected or summed per wave below runs on the scalar unit)
    uint32_t iter = 0;
#if WFPT_STAMPS
    unsigned long long acc_cyc[4] = {0, 0, 0, 0}, acc_search = 0, acc_record = 0; // (the last two: lane 0's view of shade's first two memory levels)
    uint32_t acc_cnt[5] = {0, 0, 0, 0, 0};
#endif
    while (ticket < n_items) {
        const uint32_t buf = iter & 1u;
        if (threadIdx.x == 0) L.next[buf] = gridDim.x + atomicAdd(&a.ctl->ticket, 1u);
        if (item >= n_hit_items) {
            // ---------------- miss_kernel (mk:13-38) for kMissSegsPerItem segments of the previous wavefront's miss queue
            while (item >= first_m + uniform(L.items_m[smp_m])) first_m += uniform(L.items_m[smp_m++]);
            const uint32_t smp = smp_m;
            const uint32_t first_seg = (item - first_m) * kMissSegsPerItem;
            const uint32_t n_segs = (uniform(umin(a.ctl[smp].seg_n, a.capacity)) + kChunk - 1) / kChunk;
            const size_t qo = smp * a.batch.queue_stride, co = smp * a.batch.chunk_stride;
            float *image = a.image + smp * a.batch.image_stride;
            // one segment per wave at a time: the eight waves keep eight independent load -> read-modify-write chains in flight
            for (uint32_t k = wave; k < kMissSegsPerItem; k += kExtendWaves) {
                const uint32_t seg = first_seg + k;
                if (seg >= n_segs) break;
                const uint32_t count = uniform(a.in_miss[co + seg]);
                for (uint32_t r = lane; r < count; r += 64u) {
                    const size_t slot = qo + static_cast<size_t>(seg) * kChunk + r;
                    const float dy = a.mq_in.dy()[slot]; // ray_buffer[miss_buffer[idx]].direction.y, mk:28-32
                    const uint32_t pixel_idx = a.mq_in.pixel()[slot];
                    const float t = 0.5f * (dy + 1.0f); // mk:32: the direction is not normalised after bounce 0
                    const float om = 1.0f - t;
                    const float cr = om * 1.0f + t * 0.5f; // mk:33
                    const float cg = om * 1.0f + t * 0.7f;
                    const float cb = om * 1.0f + t * 1.0f;
                    float4 *px = pixel_of(image, local_pixel(pixel_idx, a.image_width, a.tile));
                    const float4 thr = *px;
                    *px = make_float4(thr.x * cr, thr.y * cg, thr.z * cb, thr.w); // mk:35-37
                }
            }
            __syncthreads(); // L.next[buf] is visible
            ticket = uniform(L.next[buf]);
            item = tickets.item_of(ticket);
            iter += 1;
            continue;
        }
        WFPT_STAMP(t_item);
        while (item >= first_h + uniform(L.items_h[smp_h])) first_h += uniform(L.items_h[smp_h++]); // block-uniform
        const uint32_t smp = smp_h;
        const uint32_t seg_out = item - first_h;
        const uint32_t n = MODE == kBounceFirst ? umin(n_slots, a.capacity) : uniform(umin(a.ctl[smp].shade_n, a.capacity));
        const size_t qo = smp * a.batch.queue_stride, co = smp * a.batch.chunk_stride;
        float *image = a.image + smp * a.batch.image_stride;
        wfpt_frame_buffer fb = fb0;
        fb.frame += smp;

        const uint32_t h = seg_out * kChunk + threadIdx.x; // first wavefront: generate_rays' slot index; else shade's thread index
        bool live = h < n;
        float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
        uint32_t pixel_idx = 0;
        if (MODE == kBounceFirst) {
            // ---------------- generate_rays (gr:42-91), true-size semantics: lanes outside the image emit nothing
            const uint32_t workgroup_index = uniform(h >> 6), local_index = h & 63u; // one wave = one 8x8 tile of generate_rays
            const uint32_t wx = workgroup_index % a.gx, wy = workgroup_index / a.gx;
            const uint32_t id_x = wx * 8u + (local_index & 7u);
            const uint32_t id_y = (wy * a.tile.world + a.tile.rank) * 8u + (local_index >> 3);
            live = live && id_x < fb.width && id_y < fb.height;
            if (live) {
                pixel_idx = id_x + id_y * fb.width; // gr:57
                const PrimaryRay pr = primary_ray(*a.camera, id_x, id_y, fb.width, fb.height, fb);
                ox = pr.ox; oy = pr.oy; oz = pr.oz; dx = pr.dx; dy = pr.dy; dz = pr.dz;
                *pixel_of(image, local_pixel(pixel_idx, fb.width, a.tile)) = make_float4(1.0f, 1.0f, 1.0f, 1.0f); // pt:305-306 folded in: throughput starts at 1
            }
        } else if (live) {
            // ---------------- shade (sh:56-156) of hit h of the previous wavefront
            const HitSource src{a.rec_in, a.in_hit_base, a.in_first_seg, a.ctl + smp, image, a.scene.shade_rec, qo, co,
                                a.capacity, a.rng_mode, a.image_width, a.scene.prim_kind, a.tile};
#if WFPT_STAMPS
            unsigned long long sst[2] = {t_item, t_item};
            shade_hit<TRACE, true>(src, h, n, fb, ox, oy, oz, dx, dy, dz, pixel_idx, sst);
            if (MODE == kBounceMiddle) { acc_search += sst[0] - t_item; acc_record += sst[1] - sst[0]; }
#else
            shade_hit<TRACE, true>(src, h, n, fb, ox, oy, oz, dx, dy, dz, pixel_idx);
#endif
        }
        if (!TRACE) {
            __syncthreads();
            ticket = uniform(L.next[buf]);
            item = tickets.item_of(ticket);
            iter += 1;
            continue;
        }
        // ---------------- extend (ex:47-70) of the ray in registers
        WFPT_STAMP(t_trace);
#if WFPT_STAMPS
        uint32_t dbg[3] = {0, 0, 0};
#endif
        float t = 0.0f;
        uint32_t prim = 0;
        bool hit = false;
        if (live) {
            if (LDS_SCENE && EXACT)
                hit = trace_ray<Trail, PRIM, uint16_t, 0, true>(s_nodes, s_geom, s_parent, nullptr, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim);
            else if (LDS_SCENE) {
                prim = kHandOver;
                if (!far_origin(a.scene, ox, oy, oz))
                    hit = trace_ray_conservative<Trail, PRIM, uint16_t>(s_nodes, s_geom, s_parent, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim WFPT_DBG_ARG);
                if (prim == kHandOver) // near-tie, probe or far origin: the reference's own walk decides (its boxes are read from global memory: this is rare)
                    hit = retrace_reference<Trail, PRIM, uint16_t>(g_nodes, s_geom, s_parent, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim);
            } else if (!EXACT && a.scene.nodes4) {
                Stack4 st;
                st.init(WFPT_LDS_BYTES(lds, L.stack), a.scene.stack_spill, a.scene.spill_stride);
                prim = kHandOver;
                if (!far_origin(a.scene, ox, oy, oz))
                    hit = trace_ray4<PRIM>(a.scene.nodes4, a.scene.prim_geom, st, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, a.scene.root_leaf != 0, t, prim);
                if (prim == kHandOver)
                    hit = retrace_reference<Trail, PRIM, uint32_t>(g_nodes, a.scene.prim_geom, a.scene.pair_parent32, ox, oy, oz, dx, dy, dz,
                                                                    a.scene.n_nodes, t, prim);
            } else
                hit = trace_ray<Trail, PRIM, uint32_t, kStackDepth, EXACT>(g_nodes, a.scene.prim_geom, a.scene.pair_parent32, L.stack + threadIdx.x,
                                                                            ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim);
        }
        WFPT_STAMP(t_traced);
        const bool miss = live && !hit;
        const unsigned long long hit_mask = __ballot(hit), miss_mask = __ballot(miss);
        if (lane == 0) {
            L.cnt[(buf * 2 + 0) * kExtendWaves + wave] = static_cast<uint32_t>(__popcll(hit_mask));
            L.cnt[(buf * 2 + 1) * kExtendWaves + wave] = static_cast<uint32_t>(__popcll(miss_mask));
        }
        __syncthreads();
        WFPT_STAMP(t_synced);
        uint32_t hit_before, miss_before, hit_total, miss_total;
        wave_offsets(L.cnt + buf * 2u * kExtendWaves, wave, lane, hit_before, hit_total, miss_before, miss_total);
        const size_t seg = qo + static_cast<size_t>(seg_out) * kChunk;
        // (the wave's first slot is a scalar address; a lane adds its 32-bit rank: no 64-bit vector arithmetic per store)
        if (hit) { // the path record shade will read: p = origin + t * direction (sh:91), incoming direction, primitive, pixel
            float4 *rec = a.rec_out + 2u * (seg + hit_before);
            const uint32_t rank = mbcnt(hit_mask);
            rec[2u * rank] = make_float4(ox + t * dx, oy + t * dy, oz + t * dz, __uint_as_float(pixel_idx));
            rec[2u * rank + 1u] = make_float4(dx, dy, dz, __uint_as_float(prim));
        }
        if (miss) { // what miss_kernel reads of the ray (mk:29-32)
            const size_t first = seg + miss_before;
            const uint32_t rank = mbcnt(miss_mask);
            (a.mq_out.dy() + first)[rank] = dy;
            (a.mq_out.pixel() + first)[rank] = pixel_idx;
        }
        if (threadIdx.x == 0) {
            a.out_hits[co + seg_out] = hit_total;
            a.out_miss[co + seg_out] = miss_total;
        }
#if WFPT_STAMPS
        if (MODE == kBounceMiddle) { // per wave, summed over its items in registers (flushed once, at the end of the kernel)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WFPT_STAMP(t_done);
            acc_cyc[0] += t_trace - t_item;
            acc_cyc[1] += t_traced - t_trace;
            acc_cyc[2] += t_synced - t_traced;
            acc_cyc[3] += t_done - t_synced;
            acc_cnt[0] += 1u;
            acc_cnt[1] += static_cast<uint32_t>(__popcll(__ballot(live)));
            uint32_t w_visits = dbg[0], w_leaves = dbg[1], l_visits = dbg[2];
            for (int d = 1; d < 64; d <<= 1) { // the wave's trip counts = the maximum over its lanes; lane visits add up
                w_visits = max(w_visits, static_cast<uint32_t>(__shfl_xor(w_visits, d, 64)));
                w_leaves = max(w_leaves, static_cast<uint32_t>(__shfl_xor(w_leaves, d, 64)));
                l_visits += static_cast<uint32_t>(__shfl_xor(l_visits, d, 64));
            }
            acc_cnt[2] += w_visits; acc_cnt[3] += w_leaves; acc_cnt[4] += l_visits;
        }
#endif
        ticket = uniform(L.next[buf]);
        item = tickets.item_of(ticket);
        iter += 1;
    }
#if WFPT_STAMPS
    if (MODE == kBounceMiddle && a.stamps && lane == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&a.stamps[k], acc_cyc[k]);
        atomicAdd(&a.stamps[4], static_cast<unsigned long long>(acc_cnt[0]));
        atomicAdd(&a.stamps[5], static_cast<unsigned long long>(acc_cnt[1]));
        atomicAdd(&a.stamps[8], static_cast<unsigned long long>(acc_cnt[2]));
        atomicAdd(&a.stamps[9], static_cast<unsigned long long>(acc_cnt[3]));
        atomicAdd(&a.stamps[10], static_cast<unsigned long long>(acc_cnt[4]));
        atomicAdd(&a.stamps[11], acc_search);
        atomicAdd(&a.stamps[12], acc_record);
    }
#endif
}

// ================================================================================================
// Class-binned fused loop (round 4; LDS-resident scenes, WFPT_RNG_PIXEL, where the order of the hit queue is free).
// bounce_kernel takes 512 CONSECUTIVE hits per work item, so a wave holds rays that leave the ground next to rays that leave a marble, and
// lambertian next to dielectric hits: its while-while traversal runs at 0.49 / 0.40 / 0.36 lane utilisation at bounces 1-3 and every wave
// runs every material branch of shade. Here the hits of a segment are stored sorted by the COST CLASS of the primitive hit
// (ShadeRec::cost_class: 0 = the scene's dominant primitive, 1 + material type otherwise; tests/model_binning.py scores the keys) -- K
// ballots instead of one, counts exchanged as packed 10-bit fields, no atomics on the queues and nothing exchanged inside a work item; the
// scan leaves, per class, the running count in front of every segment, and a work item of the next launch is 512 hits of ONE class of one
// sample.
// Round 4 also carried the reference's ORDER through the binning for WFPT_RNG_DISPATCH (thread indices in the records, a hit flag per ray, a
// rank table per wavefront): 5.8 % slower than the thread-ordered loop, removed in round 5. Round 5 built the key that the gate model scores
// best -- (dominant | rest) x the direction.y the extension ray WILL have, computed at write time from the path's rb.y (constant per path in
// this RNG mode: it rode above the pixel index as an 8-bit snorm) with rcp / rsq arithmetic, 8 classes -- and measured it: lanes per VALU
// instruction 32.8 -> 34.8 and 8 % fewer visits, as tests/model_dirkey.py predicts, but the launch 6 % SLOWER (23 % more record loads of
// sparser classes, the key's own instructions, the first launch's rb.y): not kept; profiles/r05_rejected_experiments.txt.
// ================================================================================================
template <int K> __device__ __forceinline__ uint32_t cls_field(const uint32_t *w, int f) { return (w[f / 3] >> (10 * (f % 3))) & 1023u; }

template <int MODE, typename Trail, int PRIM, bool EXACT, int K>
__global__ __launch_bounds__(kExtendThreads, WFPT_EXTEND_MIN_WAVES) void bounce_binned_kernel(BounceArgs a) {
    extern __shared__ float4 lds[];
    constexpr bool TRACE = MODE != kBounceLast;
    constexpr int NW = ClsPack<K>::kWords;
    constexpr uint32_t kGeomWords = PRIM == 0 ? 1u : 3u;
    float4 *s_nodes = lds;
    float4 *s_geom = lds + (TRACE ? 2u * a.scene.n_nodes : 0u);
    const uint32_t parent_words = TRACE ? ((a.scene.n_nodes / 2u + 1u) + 7u) / 8u : 0u;
    const uint32_t geom_words = TRACE ? kGeomWords * a.scene.n_spheres : 0u;
    uint16_t *s_parent = reinterpret_cast<uint16_t *>(s_geom + geom_words);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_geom + geom_words + parent_words); // [2][kExtendWaves][NW] packed per-wave counts
    uint32_t *s_next = s_cnt + 2 * kExtendWaves * NW;                                    // [2] next work item
    uint8_t *s_cls = reinterpret_cast<uint8_t *>(s_next + 2);                            // [n_prims] ShadeRec::cost_class of each primitive

    const uint32_t nb = a.batch.n;
    const uint32_t n_slots = a.gx * a.gy * 64u; // first wavefront: ray slots of this context's tiles
    const uint32_t n_first = umin(n_slots, a.capacity), items_first = (n_first + kChunk - 1) / kChunk;
    // work items: [0, n_hit_items) hit items -- first: (sample, 512 ray slots); middle: (sample, class, run of 512 hits of that class);
    // last: (sample, segment) -- then the miss items (sample, kMissSegsPerItem segments), all numbered by the plan the scan's launch left
    const uint32_t *plan_h = a.plan, *plan_m = a.plan + a.plan_miss_off;
    if (MODE == kBounceLast) plan_h = a.plan + a.plan_seg_off;
    const uint32_t n_hit_items = MODE == kBounceFirst ? items_first * nb : uniform(plan_h[MODE == kBounceLast ? nb : nb * K]);
    const uint32_t n_items = MODE == kBounceFirst ? n_hit_items : uniform(plan_m[nb]);
    if (MODE == kBounceFirst && blockIdx.x == 0 && threadIdx.x < nb)
        a.ctl[threadIdx.x].n_in = n_first; // pt:313-316: counter[2] = rays of the first wavefront (read by scan)
    const TicketMap tickets(n_hit_items, n_items, MODE != kBounceFirst);
    uint32_t ticket = blockIdx.x;
    if (ticket >= n_items) return;
    uint32_t item = tickets.item_of(ticket);
    const float4 *g_nodes = reinterpret_cast<const float4 *>(a.scene.nodes);
    if (TRACE) {
        const float4 *g_staged = EXACT ? g_nodes : a.scene.nodes_ch; // reference boxes, or conservative centre / half-extent boxes
        for (uint32_t i = threadIdx.x; i < 2u * a.scene.n_nodes; i += kExtendThreads) s_nodes[i] = g_staged[i];
        for (uint32_t i = threadIdx.x; i < geom_words; i += kExtendThreads) s_geom[i] = a.scene.prim_geom[i];
        const uint4 *g_par = reinterpret_cast<const uint4 *>(a.scene.pair_parent);
        uint4 *s_par4 = reinterpret_cast<uint4 *>(s_parent);
        for (uint32_t i = threadIdx.x; i < parent_words; i += kExtendThreads) s_par4[i] = g_par[i];
        for (uint32_t i = threadIdx.x; i < a.scene.n_spheres; i += kExtendThreads)
            s_cls[i] = static_cast<uint8_t>(umin(__float_as_uint(a.scene.shade_rec[3u * i + 2u].y), static_cast<uint32_t>(K - 1)));
        __syncthreads();
    }
    wfpt_frame_buffer fb0 = a.ctl->frame; // the same for every lane: kept in scalar registers
    fb0.width = uniform(fb0.width); fb0.height = uniform(fb0.height); fb0.frame = uniform(fb0.frame); fb0.sample_number = uniform(fb0.sample_number);
    const uint32_t lane = lane_id(), wave = uniform(threadIdx.x >> 6);
    uint32_t idx_h = 0, idx_m = 0; // a workgroup's tickets only grow: the plan is searched on from where the previous item was found
    uint32_t iter = 0;
    while (ticket < n_items) {
        const uint32_t buf = iter & 1u;
        if (threadIdx.x == 0) s_next[buf] = gridDim.x + atomicAdd(&a.ctl->ticket, 1u);
        if (item >= n_hit_items) {
            // ---------------- miss_kernel (mk:13-38) for kMissSegsPerItem segments of the previous wavefront's miss queue
            while (item >= uniform(plan_m[idx_m + 1])) ++idx_m;
            const uint32_t smp = idx_m;
            const uint32_t first_seg = (item - uniform(plan_m[idx_m])) * kMissSegsPerItem;
            const uint32_t n_segs = uniform(a.ctl[smp].n_segs);
            const size_t qo = smp * a.batch.queue_stride, co = smp * a.batch.chunk_stride;
            float *image = a.image + smp * a.batch.image_stride;
            for (uint32_t k = wave; k < kMissSegsPerItem; k += kExtendWaves) { // one segment per wave at a time
                const uint32_t seg = first_seg + k;
                if (seg >= n_segs) break;
                const uint32_t count = uniform(a.in_miss[co + seg]);
                for (uint32_t r = lane; r < count; r += 64u) {
                    const size_t slot = qo + static_cast<size_t>(seg) * kChunk + r;
                    const float dy = a.mq_in.dy()[slot]; // ray_buffer[miss_buffer[idx]].direction.y, mk:28-32
                    const uint32_t pixel_idx = a.mq_in.pixel()[slot];
                    const float t = 0.5f * (dy + 1.0f); // mk:32: the direction is not normalised after bounce 0
                    const float om = 1.0f - t;
                    const float cr = om * 1.0f + t * 0.5f; // mk:33
                    const float cg = om * 1.0f + t * 0.7f;
                    const float cb = om * 1.0f + t * 1.0f;
                    float4 *px = pixel_of(image, local_pixel(pixel_idx, a.image_width, a.tile));
                    const float4 thr = *px;
                    *px = make_float4(thr.x * cr, thr.y * cg, thr.z * cb, thr.w); // mk:35-37
                }
            }
            __syncthreads(); // s_next[buf] is visible
            ticket = uniform(s_next[buf]);
            item = tickets.item_of(ticket);
            iter += 1;
            continue;
        }
        // ---------------- which rays: sample, output segment, and per lane the record (middle / last) or the ray slot (first)
        uint32_t smp, seg_out;
        bool live;
        float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
        uint32_t pixel_idx = 0;
        if (MODE == kBounceFirst) {
            smp = item / items_first;
            seg_out = item - smp * items_first;
        } else if (MODE == kBounceMiddle) {
            while (item >= uniform(plan_h[idx_h + 1])) ++idx_h;
            smp = idx_h / K;
            seg_out = item - uniform(plan_h[smp * K]);
        } else {
            while (item >= uniform(plan_h[idx_h + 1])) ++idx_h;
            smp = idx_h;
            seg_out = item - uniform(plan_h[idx_h]);
        }
        const size_t qo = smp * a.batch.queue_stride, co = smp * a.batch.chunk_stride;
        float *image = a.image + smp * a.batch.image_stride;
        wfpt_frame_buffer fb = fb0;
        fb.frame += smp;
        if (MODE == kBounceFirst) {
            // ---------------- generate_rays (gr:42-91), true-size semantics: lanes outside the image emit nothing
            const uint32_t h = seg_out * kChunk + threadIdx.x;
            const uint32_t workgroup_index = uniform(h >> 6), local_index = h & 63u; // one wave = one 8x8 tile of generate_rays
            const uint32_t wx = workgroup_index % a.gx, wy = workgroup_index / a.gx;
            const uint32_t id_x = wx * 8u + (local_index & 7u);
            const uint32_t id_y = (wy * a.tile.world + a.tile.rank) * 8u + (local_index >> 3);
            live = h < n_first && id_x < fb.width && id_y < fb.height;
            if (live) {
                pixel_idx = id_x + id_y * fb.width; // gr:57
                const PrimaryRay pr = primary_ray(*a.camera, id_x, id_y, fb.width, fb.height, fb);
                ox = pr.ox; oy = pr.oy; oz = pr.oz; dx = pr.dx; dy = pr.dy; dz = pr.dz;
                *pixel_of(image, local_pixel(pixel_idx, fb.width, a.tile)) = make_float4(1.0f, 1.0f, 1.0f, 1.0f); // pt:305-306 folded in: throughput starts at 1
            }
        } else {
            // ---------------- the hit record of the previous wavefront this lane shades
            const HitSource src{a.rec_in, nullptr, nullptr, a.ctl + smp, image, a.scene.shade_rec, qo, co,
                                a.capacity, a.rng_mode, a.image_width, a.scene.prim_kind, a.tile};
            size_t slot = 0;
            if (MODE == kBounceMiddle) {
                const uint32_t k = idx_h - smp * K, run = item - uniform(plan_h[idx_h]);
                const uint32_t n_k = uniform(a.ctl[smp].cls_n[k]);
                const uint32_t q = run * kChunk + threadIdx.x; // rank of this lane's hit among the sample's class-k hits
                live = q < n_k;
                if (live) {
                    // its segment: the last one whose class-k count in front of it is not above q, between the segments that hold the
                    // first hit of this run and of the next (the scan's first_seg table, per class)
                    const uint32_t *fs = a.first_seg_cls + (static_cast<size_t>(smp) * K + k) * a.batch.chunk_stride;
                    uint32_t lo = uniform(fs[run]);
                    uint32_t hi = (run + 1u) * kChunk < n_k ? uniform(fs[run + 1u]) : uniform(a.ctl[smp].n_segs) - 1u;
                    const uint2 *tab = a.cls_table + co * K + k; // the entry of segment s: tab[s * K]
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1u) >> 1;
                        if (tab[static_cast<size_t>(mid) * K].x <= q) lo = mid; else hi = mid - 1u;
                    }
                    const uint2 e = tab[static_cast<size_t>(lo) * K];
                    slot = qo + e.y + (q - e.x);
                }
            } else { // last: whole segments, in storage order (shade only multiplies the throughput: no key, no order)
                const uint32_t count = uniform(a.in_hits[co + seg_out]);
                live = threadIdx.x < count;
                slot = qo + static_cast<size_t>(seg_out) * kChunk + threadIdx.x;
            }
            if (live) {
                const float4 ra = a.rec_in[2u * slot], rb = a.rec_in[2u * slot + 1u];
                shade_record<TRACE, true, false>(src, ra, rb, __float_as_uint(rb.w), 0u, fb, ox, oy, oz, dx, dy, dz, pixel_idx);
            }
        }
        if (!TRACE) {
            __syncthreads();
            ticket = uniform(s_next[buf]);
            item = tickets.item_of(ticket);
            iter += 1;
            continue;
        }
        // ---------------- extend (ex:47-70) of the ray in registers
#if WFPT_STAMPS
        uint32_t dbg[3] = {0, 0, 0};
#endif
        float t = 0.0f;
        uint32_t prim = 0;
        bool hit = false;
        if (live) {
            if (EXACT)
                hit = trace_ray<Trail, PRIM, uint16_t, 0, true>(s_nodes, s_geom, s_parent, nullptr, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim);
            else {
                prim = kHandOver;
                if (!far_origin(a.scene, ox, oy, oz))
                    hit = trace_ray_conservative<Trail, PRIM, uint16_t>(s_nodes, s_geom, s_parent, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim WFPT_DBG_ARG);
                if (prim == kHandOver) // near-tie, failed verdict or far origin: the reference's own walk decides (its boxes are read from global memory: this is rare)
                    hit = retrace_reference<Trail, PRIM, uint16_t>(g_nodes, s_geom, s_parent, ox, oy, oz, dx, dy, dz, a.scene.n_nodes, t, prim);
            }
        }
        const bool miss = live && !hit;
        // ---------------- compaction, class by class: K ballots, per-wave counts as packed fields, one LDS exchange
        const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz; // the hit point shade will read: p = origin + t * direction (sh:91)
        uint32_t cls = K; // no hit
        if (hit) cls = s_cls[prim]; // ShadeRec::cost_class
        const unsigned long long hit_mask = __ballot(hit), miss_mask = __ballot(miss);
        uint32_t wcnt[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) wcnt[j] = 0;
        uint32_t rank = 0; // this lane's rank among the wave's lanes of its class
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(cls == static_cast<uint32_t>(k));
            wcnt[k / 3] |= static_cast<uint32_t>(__popcll(m)) << (10 * (k % 3));
            rank = cls == static_cast<uint32_t>(k) ? mbcnt(m) : rank;
        }
        wcnt[K / 3] |= static_cast<uint32_t>(__popcll(miss_mask)) << (10 * (K % 3));
        wcnt[(K + 1) / 3] |= static_cast<uint32_t>(__popcll(hit_mask)) << (10 * ((K + 1) % 3));
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NW; ++j) s_cnt[(buf * kExtendWaves + wave) * NW + j] = wcnt[j];
        }
        __syncthreads();
        uint32_t before[NW], total[NW];
#if WFPT_DPP_PREFIX
#pragma unroll
        for (int j = 0; j < NW; ++j) { // lane w < 8 holds wave w's packed counts; inclusive scan over the row (see wave_offsets)
            uint32_t v = lane < kExtendWaves ? s_cnt[(buf * kExtendWaves + lane) * NW + j] : 0u; // fields of at most 512 each: no carries between them
            v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, true));
            v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, true));
            v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, true));
            total[j] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 7));
            const uint32_t incl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>((wave + 15u) & 15u)));
            before[j] = wave ? incl : 0u;
        }
#else
#pragma unroll
        for (int j = 0; j < NW; ++j) { before[j] = 0; total[j] = 0; }
#pragma unroll
        for (uint32_t w = 0; w < kExtendWaves; ++w) {
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const uint32_t v = uniform(s_cnt[(buf * kExtendWaves + w) * NW + j]);
                before[j] += (w < wave) ? v : 0u; // fields of at most 512 each: no carries between them
                total[j] += v;
            }
        }
#endif
        // where class k starts inside the segment: the totals of the classes before it (packed like the counts)
        uint32_t coff[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) coff[j] = 0;
        uint32_t run_off = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            coff[k / 3] |= run_off << (10 * (k % 3));
            run_off += cls_field<K>(total, k);
        }
        const size_t seg = qo + static_cast<size_t>(seg_out) * kChunk;
        if (hit) { // the path record shade will read: hit point, incoming direction, primitive, pixel
            const uint32_t wsel = cls / 3u, sh = 10u * (cls - 3u * wsel);
            uint32_t wb = before[0], wc = coff[0];
#pragma unroll
            for (int j = 1; j < NW; ++j) { wb = wsel == static_cast<uint32_t>(j) ? before[j] : wb; wc = wsel == static_cast<uint32_t>(j) ? coff[j] : wc; }
            const size_t slot = seg + ((wc >> sh) & 1023u) + ((wb >> sh) & 1023u) + rank;
            a.rec_out[2u * slot] = make_float4(hx, hy, hz, __uint_as_float(pixel_idx));
            a.rec_out[2u * slot + 1u] = make_float4(dx, dy, dz, __uint_as_float(prim));
        }
        if (miss) { // what miss_kernel reads of the ray (mk:29-32)
            const size_t slot = seg + cls_field<K>(before, K) + mbcnt(miss_mask);
            a.mq_out.dy()[slot] = dy;
            a.mq_out.pixel()[slot] = pixel_idx;
        }
        if (threadIdx.x == 0) {
            a.out_hits[co + seg_out] = cls_field<K>(total, K + 1);
            a.out_miss[co + seg_out] = cls_field<K>(total, K);
#pragma unroll
            for (int j = 0; j < NW; ++j) a.out_cls[(co + seg_out) * NW + j] = total[j];
        }
        ticket = uniform(s_next[buf]);
        item = tickets.item_of(ticket);
        iter += 1;
    }
}

// ---- scan of the class-binned loop: one workgroup per sample. Per class the running count in front of every segment (and where the
// class's run starts inside the segment), the segment that holds the first hit of every run of kChunk hits of a class, and the counter
// protocol and loop exit of scan_kernel (pt:327-353).
template <int K>
__global__ __launch_bounds__(kScanThreads) void scan_binned_kernel(ScanBinnedArgs a) {
    constexpr int NW = ClsPack<K>::kWords, NQ = K + 2; // scanned quantities: K classes, all hits, misses
    __shared__ uint32_t s_wave[NQ][kScanThreads / 64];
    __shared__ uint32_t s_fac;
    const uint32_t sample = blockIdx.x;
    Control *c = a.ctl + sample;
    const size_t co = static_cast<size_t>(sample) * a.batch.chunk_stride;
    const uint32_t *chunk_hits = a.chunk_hits + co, *chunk_miss = a.chunk_miss + co, *chunk_cls = a.chunk_cls + co * NW;
    uint2 *cls_table = a.cls_table + co * K;
    uint32_t *first_seg_cls = a.first_seg_cls + co * K;
    const uint32_t n = umin(a.n_in[static_cast<size_t>(sample) * a.batch.ctl_stride], a.limit);
    // segments this wavefront's extend wrote: the first one 512 ray slots each, later ones one per hit work item (plan_kernel's count)
    const uint32_t n_chunks = a.bounce == 0 ? (n + kChunk - 1) / kChunk : (n == 0 ? 0u : c->next_segs);
    const uint32_t lane = lane_id(), wave = uniform(threadIdx.x >> 6);
    // ---- per-class prefixes over the segments
    uint32_t carry[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) carry[q] = 0;
    for (uint32_t base = 0; base < n_chunks; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v[NQ], inc[NQ];
        uint32_t words[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) words[j] = i < n_chunks ? chunk_cls[static_cast<size_t>(i) * NW + j] : 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = cls_field<K>(words, k);
        v[K] = i < n_chunks ? chunk_hits[i] : 0u;
        v[K + 1] = i < n_chunks ? chunk_miss[i] : 0u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            inc[q] = wave_inclusive_scan(v[q]);
            if (lane == 63) s_wave[q][wave] = inc[q];
        }
        __syncthreads();
        uint32_t before[NQ], tile[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { before[q] = 0; tile[q] = 0; }
#pragma unroll
        for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t x = s_wave[q][w];
                before[q] += (w < wave) ? x : 0u;
                tile[q] += x;
            }
        }
        if (i < n_chunks) {
            uint32_t off = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t bk = carry[k] + before[k] + inc[k] - v[k];
                cls_table[static_cast<size_t>(i) * K + k] = make_uint2(bk, i * kChunk + off);
                if (v[k] > 0) { // a segment holds at most kChunk hits of a class, so at most one multiple of kChunk
                    const uint32_t m = (bk + kChunk - 1) / kChunk;
                    if (m * kChunk < bk + v[k]) first_seg_cls[static_cast<size_t>(k) * a.batch.chunk_stride + m] = i;
                }
                off += v[k];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) carry[q] += tile[q];
        __syncthreads();
    }
    const uint32_t hits = carry[K], misses = carry[K + 1];
    // x extent of workgroup_size_64(hits) (pt:282-289), the dispatch shape shade.wgsl:72 keys its RNG on
    const uint32_t q = (hits + 63u) / 64u;
    uint32_t gx = 1;
    if (q > 1) {
        const uint32_t y = static_cast<uint32_t>(__builtin_ceilf(sqrt_(static_cast<float>(q))));
        if (threadIdx.x == 0) s_fac = 1u;
        __syncthreads();
        for (int z = static_cast<int>(y) - 1 - static_cast<int>(threadIdx.x); z >= 1; z -= kScanThreads) {
            if (q % static_cast<uint32_t>(z) == 0u) {
                atomicMax(&s_fac, static_cast<uint32_t>(z));
                break;
            }
        }
        __syncthreads();
        const uint32_t fac = s_fac;
        gx = (q / fac >= (1u << 16)) ? y : fac;
    }
    __syncthreads(); // every thread has read c->next_segs before thread 0 replaces it
    if (threadIdx.x == 0) {
        c->seg_n = n;
        c->n_segs = n_chunks;
        c->hits = hits;
        c->misses = misses;
        c->shade_gx = gx;
        if (sample == 0) c->ticket = 0; // the work-item ticket lives in the first Control block
        uint32_t done = (a.bounce == 0) ? 0u : c->done;
        const uint32_t ran = done ? 0u : 1u;       // this wavefront's extend really ran
        if (!done && misses < a.miss_floor) done = 1; // pt:332: exit before shading
        c->done = done;
        c->shade_n = done ? 0u : hits;
        c->miss_n = done ? 0u : misses;
        c->n_in = done ? 0u : hits; // every shaded hit emits exactly one extension ray (sh:155)
        uint32_t next = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c->cls_n[k] = done ? 0u : carry[k];
            next += done ? 0u : (carry[k] + kChunk - 1) / kChunk;
        }
        c->next_segs = next; // the next extend writes one segment per hit work item
        c->counters[0] = misses;
        c->counters[1] = hits;
        c->counters[2] = done ? n : 0u; // pt:335-336
        if (a.bounce < kMaxRows) {
            c->rows[a.bounce][0] = ran ? n : 0u;
            c->rows[a.bounce][1] = hits;
            c->rows[a.bounce][2] = misses;
            c->rows[a.bounce][3] = (ran && !done) ? 1u : 0u;
        }
        c->bounce = a.bounce + 1;
    }
}

// ---- the work-item plan of the next launch of the class-binned loop: exclusive prefixes over (sample, class) hit items, over the
// samples' segments (the last launch shades whole segments) and over the samples' miss items, which follow the hit items
constexpr uint32_t kPlanThreads = 1024; // >= kMaxBatch * kBinClasses
static_assert(kMaxBatch * kBinClasses <= static_cast<int>(kPlanThreads), "plan_kernel: one thread per (sample, class)");
__device__ __forceinline__ uint32_t block_exclusive_scan_plan(uint32_t v, uint32_t *s_part, uint32_t &total) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    uint32_t before = inc - v;
    total = 0;
    for (uint32_t w = 0; w < kPlanThreads / 64u; ++w) {
        const uint32_t x = s_part[w];
        before += (w < wave) ? x : 0u;
        total += x;
    }
    return before;
}
template <int K>
__global__ __launch_bounds__(kPlanThreads) void plan_kernel(PlanArgs a) {
    __shared__ uint32_t s_part[kPlanThreads / 64u];
    const uint32_t t = threadIdx.x, nb = a.batch.n; // nb * K <= kPlanThreads (kMaxBatch samples, kBinClasses classes)
    uint32_t total_h = 0, total_s = 0, total_m = 0;
    uint32_t v = 0;
    if (t < nb * K) v = (a.ctl[t / K].cls_n[t % K] + kChunk - 1) / kChunk;
    const uint32_t ex_h = block_exclusive_scan_plan(v, s_part, total_h);
    if (t < nb * K) a.plan[t] = ex_h;
    if (t == 0) a.plan[nb * K] = total_h;
    v = (t < nb && a.ctl[t].shade_n > 0) ? a.ctl[t].n_segs : 0u;
    const uint32_t ex_s = block_exclusive_scan_plan(v, s_part, total_s);
    if (t < nb) a.plan[a.plan_seg_off + t] = ex_s;
    if (t == 0) a.plan[a.plan_seg_off + nb] = total_s;
    v = (t < nb && a.ctl[t].miss_n > 0) ? (a.ctl[t].n_segs + kMissSegsPerItem - 1) / kMissSegsPerItem : 0u;
    const uint32_t ex_m = block_exclusive_scan_plan(v, s_part, total_m);
    const uint32_t first = a.last ? total_s : total_h;
    if (t < nb) a.plan[a.plan_miss_off + t] = first + ex_m;
    if (t == 0) a.plan[a.plan_miss_off + nb] = first + total_m;
}

// ================================================================================================
// HBM-resident scenes: four-wide traversal with dynamic lane refill + dense results + compaction (see RefillArgs).
// MODE kBounceFirst: ray G is slot G of generate_rays' numbering; kBounceMiddle: ray G is the extension ray of hit G of
// the previous wavefront (shade runs at refill time, by the lane that takes the ray).
// ================================================================================================
#ifndef WFPT_REFILL_MIN_WAVES
#define WFPT_REFILL_MIN_WAVES 8
#endif
// PRESHADED (middle wavefronts): the extension rays were produced by shade_rays_kernel -- shade at 64 lanes per wave instead of at
// the 24-40 idle lanes of a refill -- and wait in the dense array at their ray's slot, (o | pixel), (d | -); a lane that takes ray g
// reads its 32 bytes, traces, and leaves the result in the same slot. A refill then costs next to nothing, so waves refill early.
template <int MODE, int PRIM, bool PRESHADED = false>
__global__ __launch_bounds__(kExtendThreads, WFPT_REFILL_MIN_WAVES) void refill_kernel(RefillArgs a) {
    extern __shared__ float4 lds[];
    uint32_t *s_n = reinterpret_cast<uint32_t *>(lds);   // [kMaxBatch] rays of this wavefront per sample
    uint32_t *s_first = s_n + kMaxBatch;                  // [kMaxBatch + 1] first global ray index of each sample
    uint32_t *s_stack = s_first + kMaxBatch + 4;          // [kStack4Lds][kExtendThreads]
    const float4 *s_tile = reinterpret_cast<const float4 *>(s_stack + kStack4Lds * kExtendThreads); // [tile_n] nodes: the top of the tree
    const uint32_t n_slots = a.gx * a.gy * 64u;
    if (threadIdx.x < a.batch.n) // the samples' counters, read side by side
        s_n[threadIdx.x] = MODE == kBounceFirst ? umin(n_slots, a.capacity) : umin(a.ctl[threadIdx.x].shade_n, a.capacity);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t smp = 0; smp < a.batch.n; ++smp) {
            s_first[smp] = total;
            total += s_n[smp];
        }
        s_first[a.batch.n] = total;
    }
    __syncthreads();
    const uint32_t total = uniform(s_first[a.batch.n]);
    if (MODE == kBounceFirst && blockIdx.x == 0 && threadIdx.x < a.batch.n) a.ctl[threadIdx.x].n_in = umin(n_slots, a.capacity); // pt:313-316
    if (total == 0) return;
    // N1: the first tile_n nodes (breadth-first numbering: the top levels of the four-wide tree, where every ray passes) are
    // staged once per persistent workgroup and read with ds_read_b128; only visits below them go through the L1 / TA path
    const uint32_t tile_n = a.scene.tile_n;
    {
        float4 *tile_w = reinterpret_cast<float4 *>(s_stack + kStack4Lds * kExtendThreads);
        for (uint32_t i = threadIdx.x; i < 4u * tile_n; i += kExtendThreads) tile_w[i] = a.scene.nodes4[i];
        __syncthreads();
    }
    wfpt_frame_buffer fb0 = a.ctl->frame; // the same for every lane: kept in scalar registers
    fb0.width = uniform(fb0.width); fb0.height = uniform(fb0.height); fb0.frame = uniform(fb0.frame); fb0.sample_number = uniform(fb0.sample_number);
    const uint32_t lane = lane_id();
    uint32_t smp_cur = 0; // sample of the last group's first ray: a wave's tickets only grow, so the search goes on from there
    Stack4 st;
    st.init(WFPT_LDS_BYTES(lds, s_stack), a.scene.stack_spill, a.scene.spill_stride);
    const float4 *nodes4 = a.scene.nodes4;

    // per-lane ray and traversal state
    bool alive = false;
    uint32_t smp = 0, ray = 0, pixel_idx = 0, cur = 0, best = 0xffffffffu, budget = 0;
    // (the exact inverse direction and d.d are needed at leaves only, ~2 per ray against ~60 node visits: recomputed there, which
    // keeps the hot loop's registers free of them; more state here meant scratch spills inside the loop)
    float dx = 0, dy = 0, dz = 0, nearest = 1e30f;
    Ray4 r4 = {0, 0, 0, 0, 0, 0};
    bool more = true; // rays left at the global cursor (wave-uniform)
    // The wave reserves ray indices a block at a time ([cur, end), scalars) and hands them to its idle lanes from there: the global
    // cursor's atomic -- a round trip of microseconds that the whole wave waits for -- is paid once per kTicketBlock rays, not at
    // every refill (which made early refills, i.e. fuller waves, cost more than they brought).
    uint32_t cur_ray = 0, end_ray = 0;
#if WFPT_STAMPS
    // per wave, in registers: [0] iterations, [1] lanes holding a ray, [2] visit steps, [3] lanes visiting, [4] leaf rounds, [5] lanes in them,
    // [6] refill passes, [7] lanes refilled, [8] lanes that waited at a leaf through an iteration; cycles: [0] refill, [1] visit step, [2] leaf round
    unsigned long long st_n[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st_c[3] = {0, 0, 0};
    const unsigned long long st_t0 = stamp_now();
#endif
    for (;;) {
        WFPT_STAMP(st_a);
        // ---------------- refill: idle lanes take the next rays
        const unsigned long long idle = __ballot(!alive);
        const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
        if ((more || cur_ray < end_ray) && (n_idle >= (PRESHADED ? (MODE == kBounceFirst ? kRefillIdleFirstPre : kRefillIdlePreshaded) : (MODE == kBounceFirst ? kRefillIdleFirst : kRefillIdle)))) {
            // idle lane number `rank` takes ray `g`: first what is left of the wave's block, then (one atomic) the head of the next one,
            // so that every idle lane is served in this pass
            const uint32_t rank = mbcnt(idle);
            const uint32_t left = end_ray - cur_ray;
            uint32_t g = cur_ray + rank;
            bool take = rank < left;
            const uint32_t base = left > 0 ? cur_ray : 0xffffffffu; // first ray handed out (for the sample search)
            cur_ray += umin(n_idle, left);
            uint32_t first_g = base;
            if (left < n_idle && more) {
                uint32_t nb = 0;
                if (lane == 0) nb = atomicAdd(&a.ctl->ticket, kTicketBlock);
                nb = __builtin_amdgcn_readfirstlane(nb);
                const uint32_t b0 = umin(nb, total);
                end_ray = umin(nb + kTicketBlock, total);
                more = nb + kTicketBlock < total;
                const uint32_t want = n_idle - left, got = umin(want, end_ray - b0);
                if (rank >= left && rank - left < got) {
                    g = b0 + (rank - left);
                    take = true;
                }
                cur_ray = b0 + got;
                if (left == 0) first_g = b0;
            }
            while (first_g < total && first_g >= uniform(s_first[smp_cur + 1])) ++smp_cur; // scalar
#if WFPT_STAMPS
            st_n[6] += 1; st_n[7] += static_cast<unsigned long long>(__popcll(__ballot(!alive && take)));
#endif
            if (!alive && take) {
                smp = smp_cur;
                while (g >= s_first[smp + 1]) ++smp; // a group rarely straddles samples
                ray = g - s_first[smp];
                float *image = a.image + smp * a.batch.image_stride;
                wfpt_frame_buffer fb = fb0;
                fb.frame += smp;
                bool ok = true;
                float ox = 0, oy = 0, oz = 0;
                if (PRESHADED) { // the ray waits in the dense array: the extension ray of hit `ray` (shade_rays_kernel), or primary ray `ray` (generate_dense_kernel)
                    const size_t slot = smp * a.batch.queue_stride + ray;
                    const float4 ra = a.dense_out[2u * slot], rb = a.dense_out[2u * slot + 1u];
                    ox = ra.x; oy = ra.y; oz = ra.z; pixel_idx = __float_as_uint(ra.w);
                    dx = rb.x; dy = rb.y; dz = rb.z;
                    if (MODE == kBounceFirst) ok = __float_as_uint(rb.w) != kDenseInactive; // a lane outside the image: neither hit nor miss, the marker stays
                } else if (MODE == kBounceFirst) { // generate_rays (gr:42-91), true-size semantics
                    const uint32_t workgroup_index = ray >> 6, local_index = ray & 63u;
                    const uint32_t wx = workgroup_index % a.gx, wy = workgroup_index / a.gx;
                    const uint32_t id_x = wx * 8u + (local_index & 7u);
                    const uint32_t id_y = (wy * a.tile.world + a.tile.rank) * 8u + (local_index >> 3);
                    ok = id_x < fb.width && id_y < fb.height;
                    if (ok) {
                        pixel_idx = id_x + id_y * fb.width;
                        const PrimaryRay pr = primary_ray(*a.camera, id_x, id_y, fb.width, fb.height, fb);
                        ox = pr.ox; oy = pr.oy; oz = pr.oz; dx = pr.dx; dy = pr.dy; dz = pr.dz;
                        *pixel_of(image, local_pixel(pixel_idx, fb.width, a.tile)) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    } else { // a lane outside the image: neither hit nor miss
                        const size_t slot = smp * a.batch.queue_stride + ray;
                        a.dense_out[2u * slot + 1u] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kDenseInactive));
                    }
                } else { // shade (sh:56-156) of hit `ray` of the previous wavefront
                    const HitSource src{a.rec_in, a.in_hit_base, a.in_first_seg, a.ctl + smp, image, a.scene.shade_rec,
                                        smp * a.batch.queue_stride, smp * a.batch.chunk_stride, a.capacity, a.rng_mode, a.image_width,
                                        a.scene.prim_kind, a.tile};
                    shade_hit<true>(src, ray, s_n[smp], fb, ox, oy, oz, dx, dy, dz, pixel_idx);
                }
                if (ok) {
                    r4 = make_ray4(ox, oy, oz, dx, dy, dz);
                    nearest = 1e30f; best = 0xffffffffu; cur = 0; st.reset(); budget = a.scene.n_nodes;
                    alive = true;
                    if (far_origin(a.scene, ox, oy, oz)) hand_over(nearest, best); // (the walk then runs out at once; the re-trace at its end decides)
                }
            }
        }
        if (__ballot(alive) == 0) {
            if (!more && cur_ray == end_ray) break;
            continue; // everything fetched so far was outside the image: fetch again
        }
        // ---------------- one round of the four-wide traversal (trace_ray4's loop body) for the lanes that hold a ray
        // "if-if" scheduling: every iteration a lane does ONE step, a four-wide node visit or a leaf, and the wave runs both
        // pieces of code whenever some lane needs them. Rays of such scenes reach a leaf every ~4 node visits at
        // unrelated times; a "while-while" round (all lanes descend until the last one sits on a leaf) left ~3/4 of the
        // lanes waiting in the inner loop (691 Mrays/s on the 1M-triangle soup), while the leaf code (~80 instructions per
        // triangle) is cheap next to a four-box visit (~180): running it every iteration costs less than the waiting did (1024).
        bool fin = false;
        WFPT_STAMP(st_b);
#if WFPT_STAMPS
        st_c[0] += st_b - st_a;
        st_n[0] += 1; st_n[1] += static_cast<unsigned long long>(__popcll(__ballot(alive)));
        { const unsigned long long vm = __ballot(alive && (cur & kLeafFlag) == 0); if (vm) { st_n[2] += 1; st_n[3] += static_cast<unsigned long long>(__popcll(vm)); } }
#endif
        if (alive && (cur & kLeafFlag) == 0) {
#if WFPT_BUDGET_INNER
            if (budget-- == 0) {
                fin = true;
            } else
#endif
            {
            // (no step budget on node visits, as in the LDS walk's inner loop: wfpt_create / wfpt_update_scene reject a tree with a cycle
            // (validate_bvh), the four-wide nodes are collapsed from it, and a walk over a tree visits a node once; leaves keep theirs)
            const Visit4 v = visit4_at(nodes4, s_tile, tile_n, cur, r4, nearest);
            // (the whole wave takes the LDS-only form of the stack operations unless one of its lanes is within three entries of the
            // column's end / holds spilled entries: 0.3 % of the pushes go deeper than kStack4Lds)
            if (v.t0 >= 2e30f) {
                if (st.empty()) fin = true;
                else if (__ballot(!st.in_lds()) == 0) cur = st.pop_lds();
                else cur = st.pop();
            } else {
                if (__ballot(!st.room3()) == 0) {
                    if (v.t3 < 2e30f) st.push_lds(v.w3);
                    if (v.t2 < 2e30f) st.push_lds(v.w2);
                    if (v.t1 < 2e30f) st.push_lds(v.w1);
                } else {
                    if (v.t3 < 2e30f) st.push(v.w3);
                    if (v.t2 < 2e30f) st.push(v.w2);
                    if (v.t1 < 2e30f) st.push(v.w1);
                }
                cur = v.w0;
            }
            }
        }
        // A lane that has just arrived at a leaf goes on into the leaf code of the same iteration. The leaf code runs when enough
        // lanes wait at a leaf to be worth the wave's time, or when no lane has a node to visit.
        const bool at_leaf = alive && !fin && (cur & kLeafFlag) != 0;
        const bool leaf_round = __popcll(__ballot(at_leaf)) >= WFPT_LEAF_LANES || __ballot(alive && !fin && !at_leaf) == 0;
        WFPT_STAMP(st_c0);
#if WFPT_STAMPS
        st_c[1] += st_c0 - st_b;
        { const unsigned long long lm = __ballot(at_leaf); if (lm && leaf_round) { st_n[4] += 1; st_n[5] += static_cast<unsigned long long>(__popcll(lm)); } else st_n[8] += static_cast<unsigned long long>(__popcll(lm)); }
#endif
        if (at_leaf && leaf_round) { // at a leaf child
            if (budget-- == 0) {
                fin = true;
            } else {
                const uint32_t first = cur & kLeafFirstMask, count = (cur >> kLeafCountShift) & 7u;
                const float aa = (dx * dx + dy * dy) + dz * dz;              // dot(direction, direction), ex:190
                visit_leaf_in_place<PRIM, true>(a.scene.prim_geom, first, count, a.scene.root_leaf != 0, r4.ox, r4.oy, r4.oz, dx, dy, dz, 0.f, 0.f, 0.f, aa, nearest, best); // see trace_ray4
                if (st.empty()) fin = true;
                else if (__ballot(!st.in_lds()) == 0) cur = st.pop_lds();
                else cur = st.pop();
            }
        }
#if WFPT_STAMPS
        { const unsigned long long st_d = stamp_now(); st_c[2] += st_d - st_c0; }
#endif
        if (alive && fin) { // the ray is done: dense record in ray order (p = o + t d as shade reads it, sh:91)
            if (best == kHandOver) // rare: the reference's own walk over the caller's binary tree decides
                (void)retrace_reference<unsigned long long, PRIM, uint32_t>(reinterpret_cast<const float4 *>(a.scene.nodes), a.scene.prim_geom,
                                                                             a.scene.pair_parent32, r4.ox, r4.oy, r4.oz, dx, dy, dz, a.scene.n_nodes,
                                                                             nearest, best);
            const size_t slot = smp * a.batch.queue_stride + ray;
            const bool hit = nearest < 1e30f; // ex:157
            a.dense_out[2u * slot] = make_float4(r4.ox + nearest * dx, r4.oy + nearest * dy, r4.oz + nearest * dz, __uint_as_float(pixel_idx));
            a.dense_out[2u * slot + 1u] = make_float4(dx, dy, dz, __uint_as_float(hit ? best : kDenseMiss));
            alive = false;
        }
    }
#if WFPT_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long *out = a.stamps + (MODE == kBounceFirst ? 16 : 32);
        for (int k = 0; k < 9; ++k) atomicAdd(&out[k], st_n[k]);
        for (int k = 0; k < 3; ++k) atomicAdd(&out[9 + k], st_c[k]);
        atomicAdd(&out[12], stamp_now() - st_t0);
        atomicAdd(&out[13], 1ull);
    }
#endif
}

// shade (sh:56-156) of every hit of the previous wavefront at full waves, for the refill traversal: one workgroup = one run of
// kChunk consecutive hits of one sample (blockIdx.y); the extension ray of hit h goes to slot h of the dense array.
__global__ __launch_bounds__(kExtendThreads) void shade_rays_kernel(RefillArgs a) {
    const uint32_t smp = blockIdx.y;
    const uint32_t n = uniform(umin(a.ctl[smp].shade_n, a.capacity));
    if (blockIdx.x * kChunk >= n) return;
    wfpt_frame_buffer fb = a.ctl->frame;
    fb.width = uniform(fb.width); fb.height = uniform(fb.height); fb.frame = uniform(fb.frame) + smp; fb.sample_number = uniform(fb.sample_number);
    const uint32_t h = blockIdx.x * kChunk + threadIdx.x;
    if (h >= n) return;
    const HitSource src{a.rec_in, a.in_hit_base, a.in_first_seg, a.ctl + smp, a.image + smp * a.batch.image_stride, a.scene.shade_rec,
                        smp * a.batch.queue_stride, smp * a.batch.chunk_stride, a.capacity, a.rng_mode, a.image_width,
                        a.scene.prim_kind, a.tile};
    float ox, oy, oz, dx, dy, dz;
    uint32_t pixel_idx;
    shade_hit<true, true>(src, h, n, fb, ox, oy, oz, dx, dy, dz, pixel_idx);
    const size_t slot = smp * a.batch.queue_stride + h;
    a.dense_out[2u * slot] = make_float4(ox, oy, oz, __uint_as_float(pixel_idx));
    a.dense_out[2u * slot + 1u] = make_float4(dx, dy, dz, 0.0f);
}

// generate_rays (gr:42-91, true-size semantics) of the first wavefront at full waves, for the refill traversal: primary ray g of sample
// blockIdx.y goes to slot g of the dense array, (o | pixel), (d | 0); a lane outside the image leaves the inactive marker. Generating at
// refill time ran ~250 instructions for the 16-24 idle lanes of a wave and kept the first launch's kernel 12 vector registers over budget.
__global__ __launch_bounds__(256) void generate_dense_kernel(RefillArgs a) {
    const uint32_t smp = blockIdx.y;
    const uint32_t n = umin(a.gx * a.gy * 64u, a.capacity);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    wfpt_frame_buffer fb = a.ctl->frame;
    fb.width = uniform(fb.width); fb.height = uniform(fb.height); fb.frame = uniform(fb.frame) + smp; fb.sample_number = uniform(fb.sample_number);
    const uint32_t workgroup_index = g >> 6, local_index = g & 63u; // one wave = one 8x8 tile of generate_rays
    const uint32_t wx = workgroup_index % a.gx, wy = workgroup_index / a.gx;
    const uint32_t id_x = wx * 8u + (local_index & 7u);
    const uint32_t id_y = (wy * a.tile.world + a.tile.rank) * 8u + (local_index >> 3);
    const size_t slot = smp * a.batch.queue_stride + g;
    if (id_x < fb.width && id_y < fb.height) {
        const uint32_t pixel_idx = id_x + id_y * fb.width; // gr:57
        const PrimaryRay pr = primary_ray(*a.camera, id_x, id_y, fb.width, fb.height, fb);
        a.dense_out[2u * slot] = make_float4(pr.ox, pr.oy, pr.oz, __uint_as_float(pixel_idx));
        a.dense_out[2u * slot + 1u] = make_float4(pr.dx, pr.dy, pr.dz, 0.0f);
        *pixel_of(a.image + smp * a.batch.image_stride, local_pixel(pixel_idx, fb.width, a.tile)) = make_float4(1.0f, 1.0f, 1.0f, 1.0f); // pt:305-306 folded in
    } else {
        a.dense_out[2u * slot + 1u] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kDenseInactive));
    }
}

// dense per-ray results -> the segment-compacted path-record and miss queues of the fused loop, in ray order
__global__ __launch_bounds__(kExtendThreads) void compact_kernel(CompactArgs a) {
    __shared__ uint32_t s_cnt[2][kExtendWaves];
    const uint32_t smp = blockIdx.y, chunk = blockIdx.x;
    const uint32_t n = umin(a.ctl[smp].n_in, a.capacity);
    if (chunk * kChunk >= n) return;
    const size_t qo = smp * a.batch.queue_stride, co = smp * a.batch.chunk_stride;
    const uint32_t idx = chunk * kChunk + threadIdx.x;
    float4 ra = make_float4(0, 0, 0, 0), rb = make_float4(0, 0, 0, __uint_as_float(kDenseInactive));
    if (idx < n) {
        rb = a.dense_in[2u * (qo + idx) + 1u];
        if (__float_as_uint(rb.w) != kDenseInactive) ra = a.dense_in[2u * (qo + idx)];
    }
    const uint32_t prim = __float_as_uint(rb.w);
    const bool hit = prim < kDenseInactive, miss = prim == kDenseMiss;
    const unsigned long long hit_mask = __ballot(hit), miss_mask = __ballot(miss);
    const uint32_t lane = lane_id(), wave = uniform(threadIdx.x >> 6); // (a scalar: what is selected or summed per wave below runs on the scalar unit)
    if (lane == 0) {
        s_cnt[0][wave] = static_cast<uint32_t>(__popcll(hit_mask));
        s_cnt[1][wave] = static_cast<uint32_t>(__popcll(miss_mask));
    }
    __syncthreads();
    uint32_t hit_before, miss_before, hit_total, miss_total;
    wave_offsets(&s_cnt[0][0], wave, lane, hit_before, hit_total, miss_before, miss_total);
    const size_t seg = qo + static_cast<size_t>(chunk) * kChunk;
    if (hit) {
        const size_t slot = seg + hit_before + mbcnt(hit_mask);
        a.rec_out[2u * slot] = ra;
        a.rec_out[2u * slot + 1u] = rb;
    }
    if (miss) {
        const size_t slot = seg + miss_before + mbcnt(miss_mask);
        a.mq_out.dy()[slot] = rb.y;
        a.mq_out.pixel()[slot] = __float_as_uint(ra.w);
    }
    if (threadIdx.x == 0) {
        a.out_hits[co + chunk] = hit_total;
        a.out_miss[co + chunk] = miss_total;
    }
}

// ================================================================================================
// accumulate (ac:4-17): pure streaming, 16 B per lane and sample
// ================================================================================================
__global__ __launch_bounds__(256) void accumulate_kernel(AccumulateArgs a) {
    // accumulated += image_0; += image_1; ... in sample order, so a batch gives exactly the sums that sequential samples (one
    // accumulate dispatch each, pt:362) would. One thread = one pixel: 16 B of each image slice (float4 per pixel) in, 12 B of
    // `accumulated` (the reference's stride-12 layout) read and written. Sixteen samples' loads are in flight before the first add:
    // a band-sharded slab gives a SIMD only a few waves, and with one sample per trip each had one load in flight (1.2 TB/s
    // on 1/8 of the frame); the adds keep the sample order.
    const size_t stride4 = a.batch.image_stride / 4u; // a slice is a whole number of float4 pixels
    const float4 *image4 = reinterpret_cast<const float4 *>(a.image);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_pixels; i += gridDim.x * blockDim.x) {
        float r = a.accumulated[3u * i], g = a.accumulated[3u * i + 1u], b = a.accumulated[3u * i + 2u];
        const float4 *im0 = image4 + i;
        uint32_t smp = 0;
        for (; smp + 16u <= a.batch.n; smp += 16u) {
            float4 p[16];
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) p[k] = im0[(smp + k) * stride4];
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) { r += p[k].x; g += p[k].y; b += p[k].z; }
        }
        for (; smp + 4u <= a.batch.n; smp += 4u) {
            float4 p[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) p[k] = im0[(smp + k) * stride4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) { r += p[k].x; g += p[k].y; b += p[k].z; }
        }
        for (; smp < a.batch.n; ++smp) {
            const float4 p = im0[smp * stride4];
            r += p.x; g += p.y; b += p.z;
        }
        a.accumulated[3u * i] = r; a.accumulated[3u * i + 1u] = g; a.accumulated[3u * i + 2u] = b;
    }
    if (blockIdx.x == 0 && a.bookkeeping) { // end of a fused batch: the samples' per-wavefront rows -> the context's totals
        // one thread per sample and integer sums through LDS (one thread walking samples x rows of global memory took ~0.1 ms,
        // a fixed cost per frame that a band-sharded rank pays in full)
        __shared__ unsigned long long s_rows[kMaxRows][2]; // hits, misses per wavefront over the batch
        for (uint32_t k = threadIdx.x; k < 2u * kMaxRows; k += blockDim.x) (&s_rows[0][0])[k] = 0ull;
        __syncthreads();
        for (uint32_t smp = threadIdx.x; smp < a.batch.n; smp += blockDim.x) {
            const Control *c = a.ctl + smp;
            const uint32_t rows = c->bounce < kMaxRows ? c->bounce : kMaxRows;
            for (uint32_t b = 0; b < rows; ++b) {
                if (c->rows[b][0] == 0) continue;
                atomicAdd(&s_rows[b][0], static_cast<unsigned long long>(c->rows[b][1]));
                atomicAdd(&s_rows[b][1], static_cast<unsigned long long>(c->rows[b][2]));
            }
        }
        __syncthreads();
        Control *c0 = a.ctl;
        if (threadIdx.x < kMaxRows) {
            const uint32_t b = threadIdx.x;
            const unsigned long long h = s_rows[b][0], m = s_rows[b][1];
            if (h + m) {
                c0->wave_totals[b][0] += h + m;
                c0->wave_totals[b][1] += h;
                c0->wave_totals[b][2] += m;
            }
        }
        if (threadIdx.x == 0) {
            unsigned long long h = 0, m = 0;
            for (uint32_t b = 0; b < kMaxRows; ++b) { h += s_rows[b][0]; m += s_rows[b][1]; }
            c0->totals[0] += h + m;
            c0->totals[1] += h;
            c0->totals[2] += m;
            c0->totals[3] += a.batch.n;
            c0->samples += a.batch.n;
            c0->ticket = 0; // the fused loop's last bounce launch drew tickets after the last scan
            c0->frame.frame += a.batch.n; // RenderProgress::get_next_frame (parameters.rs:78-83) for the next samples
        }
    }
}

// ================================================================================================
// helpers
// ================================================================================================
// the frame uniform of the next sample (pt:296-297), written on the stream: no host synchronisation between frames
__global__ void set_frame_kernel(Control *ctl, wfpt_frame_buffer f) { ctl->frame = f; }

__global__ void fill_kernel(float *p, float v, size_t n) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        p[i] = v;
}

// Multi-GPU gather, root side: band j of rank `rank`'s slab is band j * world + rank of the frame. `n_valid` floats of
// the slab exist (an unsharded context holds width * height pixels, not whole bands).
__global__ void band_scatter_kernel(float *frame, const float *slab, size_t n_valid, uint32_t band_floats, uint32_t world, uint32_t rank) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n_valid; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t band = i / band_floats, k = i - band * band_floats;
        frame[(band * world + rank) * band_floats + k] = slab[i];
    }
}

__global__ void rays_to_aos_kernel(RayQueue q, wfpt_ray *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    wfpt_ray r;
    const uint32_t pixel = q.pixel()[i];
    r.origin[0] = q.ox()[i]; r.origin[1] = q.oy()[i]; r.origin[2] = q.oz()[i];
    r.direction[0] = q.dx()[i]; r.direction[1] = q.dy()[i]; r.direction[2] = q.dz()[i];
    r.direction[3] = 0.0f;
    if (pixel == WFPT_INACTIVE_PIXEL) { // padding rays are all-zero in the reference layout
        r.origin[3] = 0.0f;
        r.inv_direction[0] = r.inv_direction[1] = r.inv_direction[2] = 0.0f;
    } else {
        r.origin[3] = 1.0f;
        r.inv_direction[0] = 1.0f / r.direction[0]; // gr:87, sh:153
        r.inv_direction[1] = 1.0f / r.direction[1];
        r.inv_direction[2] = 1.0f / r.direction[2];
    }
    r.pixel_idx = pixel;
    out[i] = r;
}

__global__ void rays_from_aos_kernel(RayQueue q, const wfpt_ray *in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const wfpt_ray r = in[i];
    q.ox()[i] = r.origin[0]; q.oy()[i] = r.origin[1]; q.oz()[i] = r.origin[2];
    q.dx()[i] = r.direction[0]; q.dy()[i] = r.direction[1]; q.dz()[i] = r.direction[2];
    q.pixel()[i] = r.pixel_idx;
}

__global__ void selftest_math_kernel(int op, const float *a, const float *b, float *out, size_t n) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 0.0f;
    float r = 0.0f, tmp;
    switch (op) {
    case 0: r = sqrt_(x); break;
    case 1: r = x / y; break;
    case 2: sincos_(x, r, tmp); break;
    case 3: sincos_(x, tmp, r); break;
    case 4: r = pow_(x, y); break;
    case 5: r = u32_to_unit_float(__float_as_uint(x)); break;
    case 6: r = min_(x, y); break;
    case 7: r = max_(x, y); break;
    default: break;
    }
    out[i] = r;
}

} // namespace

// ================================================================================================
// launchers
// ================================================================================================
uint32_t extend_lds_bytes(uint32_t n_nodes, uint32_t n_prims, uint32_t prim_kind, bool lds_scene) {
    const uint32_t misc = 4u * (4u * kExtendWaves + 2u + 2u * kMaxBatchClassic + 1u + 6u * kExtendWaves) + 16u;
    if (!lds_scene) return misc + 4u * 2u * kStackDepth * kExtendThreads; // a stack entry is (node, packed fields)
    const uint32_t parent_words = ((n_nodes / 2u + 1u) + 7u) / 8u;
    return 32u * n_nodes + 16u * (prim_kind == 0 ? 1u : 3u) * n_prims + 16u * parent_words + misc;
}

namespace {
using ExtendFn = void (*)(ExtendArgs);
template <bool INACT, int PRIM, bool EXACT> ExtendFn extend_pick(bool lds_scene, bool deep) {
    if (!lds_scene) return extend_kernel<INACT, unsigned long long, PRIM, false, EXACT>;
    return deep ? extend_kernel<INACT, unsigned long long, PRIM, true, EXACT> : extend_kernel<INACT, uint32_t, PRIM, true, EXACT>;
}
template <bool EXACT> ExtendFn extend_variant_of(const SceneDev &sc, bool has_inactive) {
    const bool lds = sc.lds_scene != 0, deep = sc.depth > 31u; // a 32-bit trail covers trees up to 31 levels deep
    if (sc.prim_kind == 0) return has_inactive ? extend_pick<true, 0, EXACT>(lds, deep) : extend_pick<false, 0, EXACT>(lds, deep);
    return has_inactive ? extend_pick<true, 1, EXACT>(lds, deep) : extend_pick<false, 1, EXACT>(lds, deep);
}
ExtendFn extend_variant(const SceneDev &sc, bool has_inactive) {
    return sc.exact ? extend_variant_of<true>(sc, has_inactive) : extend_variant_of<false>(sc, has_inactive);
}
} // namespace

hipError_t extend_blocks_per_cu(const SceneDev &scene, int *blocks) {
    hipError_t e = hipSuccess;
    if (scene.lds_bytes > 64u * 1024u) {
        for (int v = 0; v < 4; ++v) { // both box tests: the context may switch between them later (decide_exact)
            SceneDev sc = scene;
            sc.exact = (v >> 1) & 1;
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(extend_variant(sc, (v & 1) != 0)),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(scene.lds_bytes));
            if (e != hipSuccess) return e;
        }
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, extend_variant(scene, false), kExtendThreads, scene.lds_bytes);
}

uint32_t bounce_lds_bytes(uint32_t n_nodes, uint32_t n_prims, uint32_t prim_kind, bool lds_scene) {
    const uint32_t misc = 4u * kBounceMiscWords;
    if (!lds_scene) return misc + 4u * 2u * kStackDepth * kExtendThreads;
    const uint32_t parent_words = ((n_nodes / 2u + 1u) + 7u) / 8u;
    return 32u * n_nodes + 16u * (prim_kind == 0 ? 1u : 3u) * n_prims + 16u * parent_words + misc;
}

namespace {
using BounceFn = void (*)(BounceArgs);
template <int MODE, int PRIM, bool EXACT> BounceFn bounce_pick(bool lds_scene, bool deep) {
    if (!lds_scene) return bounce_kernel<MODE, unsigned long long, PRIM, false, EXACT>;
    return deep ? bounce_kernel<MODE, unsigned long long, PRIM, true, EXACT> : bounce_kernel<MODE, uint32_t, PRIM, true, EXACT>;
}
template <bool EXACT> BounceFn bounce_variant_of(const SceneDev &sc, int mode) {
    const bool lds = sc.lds_scene != 0, deep = sc.depth > 31u;
    if (sc.prim_kind == 0) return mode == kBounceFirst ? bounce_pick<kBounceFirst, 0, EXACT>(lds, deep) : bounce_pick<kBounceMiddle, 0, EXACT>(lds, deep);
    return mode == kBounceFirst ? bounce_pick<kBounceFirst, 1, EXACT>(lds, deep) : bounce_pick<kBounceMiddle, 1, EXACT>(lds, deep);
}
BounceFn bounce_variant(const SceneDev &sc, int mode) {
    if (mode == kBounceLast) return bounce_kernel<kBounceLast, uint32_t, 0, false, false>; // no traversal: one variant
    return sc.exact ? bounce_variant_of<true>(sc, mode) : bounce_variant_of<false>(sc, mode);
}
uint32_t bounce_dynamic_lds(const SceneDev &sc, int mode) {
    return mode == kBounceLast ? 4u * kBounceMiscWords : bounce_lds_bytes(sc.n_nodes, sc.n_spheres, sc.prim_kind, sc.lds_scene != 0);
}
} // namespace

hipError_t bounce_blocks_per_cu(const SceneDev &scene, int *blocks) {
    hipError_t e = hipSuccess;
    const uint32_t bytes = bounce_dynamic_lds(scene, kBounceMiddle);
    if (bytes > 64u * 1024u) {
        for (int exact = 0; exact < 2; ++exact)
            for (int mode : {kBounceFirst, kBounceMiddle}) {
                SceneDev sc = scene;
                sc.exact = static_cast<uint32_t>(exact);
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(bounce_variant(sc, mode)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
                if (e != hipSuccess) return e;
            }
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, bounce_variant(scene, kBounceMiddle), kExtendThreads, bytes);
}

// ---- class-binned loop
namespace {
uint32_t bounce_binned_lds(const SceneDev &sc, int mode) {
    const uint32_t misc = 4u * (2u * kExtendWaves * ClsPack<kBinClasses>::kWords + 2u + 2u) + ((sc.n_spheres + 15u) & ~15u);
    if (mode == kBounceLast) return misc;
    const uint32_t parent_words = ((sc.n_nodes / 2u + 1u) + 7u) / 8u;
    return 32u * sc.n_nodes + 16u * (sc.prim_kind == 0 ? 1u : 3u) * sc.n_spheres + 16u * parent_words + misc;
}
template <int MODE, int PRIM, bool EXACT> BounceFn bounce_binned_pick(bool deep) {
    return deep ? bounce_binned_kernel<MODE, unsigned long long, PRIM, EXACT, kBinClasses> : bounce_binned_kernel<MODE, uint32_t, PRIM, EXACT, kBinClasses>;
}
template <bool EXACT> BounceFn bounce_binned_variant_of(const SceneDev &sc, int mode) {
    const bool deep = sc.depth > 31u;
    if (sc.prim_kind == 0) return mode == kBounceFirst ? bounce_binned_pick<kBounceFirst, 0, EXACT>(deep) : bounce_binned_pick<kBounceMiddle, 0, EXACT>(deep);
    return mode == kBounceFirst ? bounce_binned_pick<kBounceFirst, 1, EXACT>(deep) : bounce_binned_pick<kBounceMiddle, 1, EXACT>(deep);
}
BounceFn bounce_binned_variant(const SceneDev &sc, int mode) {
    if (mode == kBounceLast) return bounce_binned_kernel<kBounceLast, uint32_t, 0, false, kBinClasses>; // no traversal: one variant
    return sc.exact ? bounce_binned_variant_of<true>(sc, mode) : bounce_binned_variant_of<false>(sc, mode);
}
} // namespace

hipError_t bounce_binned_blocks_per_cu(const SceneDev &scene, int *blocks) {
    const uint32_t bytes = bounce_binned_lds(scene, kBounceMiddle);
    if (bytes > 64u * 1024u) {
        for (int exact = 0; exact < 2; ++exact)
            for (int mode : {kBounceFirst, kBounceMiddle}) {
                SceneDev sc = scene;
                sc.exact = static_cast<uint32_t>(exact);
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bounce_binned_variant(sc, mode)),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
                if (e != hipSuccess) return e;
            }
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, bounce_binned_variant(scene, kBounceMiddle), kExtendThreads, bytes);
}

hipError_t launch_bounce_binned(const BounceArgs &a, int mode, uint32_t grid, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(bounce_binned_variant(a.scene, mode), dim3(grid), dim3(kExtendThreads), bounce_binned_lds(a.scene, mode), s, a);
    return hipGetLastError();
}

hipError_t launch_scan_binned(const ScanBinnedArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(scan_binned_kernel<kBinClasses>, dim3(a.batch.n), dim3(kScanThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_plan(const PlanArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(plan_kernel<kBinClasses>, dim3(1), dim3(kPlanThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_bounce(const BounceArgs &a, int mode, uint32_t grid, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(bounce_variant(a.scene, mode), dim3(grid), dim3(kExtendThreads), bounce_dynamic_lds(a.scene, mode), s, a);
    return hipGetLastError();
}

namespace {
constexpr uint32_t kRefillLdsFixed = 4u * (2u * kMaxBatch + 4u + kStack4Lds * kExtendThreads); // + 64 B per staged node
}

hipError_t launch_refill(const RefillArgs &a, int mode, uint32_t grid, hipStream_t s, bool preshaded) {
    if (grid == 0) return hipSuccess;
    using Fn = void (*)(RefillArgs);
    Fn fn;
    if (mode == kBounceFirst && preshaded) fn = a.scene.prim_kind == 0 ? refill_kernel<kBounceFirst, 0, true> : refill_kernel<kBounceFirst, 1, true>;
    else if (mode == kBounceFirst) fn = a.scene.prim_kind == 0 ? refill_kernel<kBounceFirst, 0> : refill_kernel<kBounceFirst, 1>;
    else if (preshaded) fn = a.scene.prim_kind == 0 ? refill_kernel<kBounceMiddle, 0, true> : refill_kernel<kBounceMiddle, 1, true>;
    else fn = a.scene.prim_kind == 0 ? refill_kernel<kBounceMiddle, 0> : refill_kernel<kBounceMiddle, 1>;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kExtendThreads), kRefillLdsFixed + 64u * a.scene.tile_n, s, a);
    return hipGetLastError();
}

hipError_t launch_shade_rays(const RefillArgs &a, uint32_t n_chunks, hipStream_t s) {
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(shade_rays_kernel, dim3(n_chunks, a.batch.n), dim3(kExtendThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_generate_dense(const RefillArgs &a, hipStream_t s) {
    const uint32_t n = a.gx * a.gy * 64u < a.capacity ? a.gx * a.gy * 64u : a.capacity;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(generate_dense_kernel, dim3((n + 255u) / 256u, a.batch.n), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_compact(const CompactArgs &a, uint32_t n_chunks, hipStream_t s) {
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_kernel, dim3(n_chunks, a.batch.n), dim3(kExtendThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_generate(const GenerateArgs &a, hipStream_t s) {
    const uint32_t n = a.gx * a.gy * 64u;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(generate_rays_kernel, dim3((n + 255u) / 256u, a.batch.n), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_extend(const ExtendArgs &a, uint32_t grid, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(extend_variant(a.scene, a.has_inactive != 0), dim3(grid), dim3(kExtendThreads), a.scene.lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(scan_kernel, dim3(a.batch.n), dim3(kScanThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_shade(const ShadeArgs &a, uint32_t grid, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(shade_kernel, dim3(grid, a.batch.n, (a.split && a.material == 0xffffffffu) ? 3u : 1u), dim3(kConsumerThreads), 0,
                       s, a);
    return hipGetLastError();
}

hipError_t launch_miss(const MissArgs &a, uint32_t grid, hipStream_t s) {
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(miss_kernel, dim3(grid, a.batch.n), dim3(kConsumerThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_accumulate(const AccumulateArgs &a, uint32_t grid, hipStream_t s) {
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid ? grid : 1u), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_set_frame(Control *ctl, const wfpt_frame_buffer &f, hipStream_t s) {
    hipLaunchKernelGGL(set_frame_kernel, dim3(1), dim3(1), 0, s, ctl, f);
    return hipGetLastError();
}

hipError_t launch_fill(float *p, float v, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(fill_kernel, dim3(static_cast<uint32_t>(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, p, v, n);
    return hipGetLastError();
}

hipError_t launch_band_scatter(float *frame, const float *slab, size_t n_valid, size_t band_floats, uint32_t world, uint32_t rank, hipStream_t s) {
    if (n_valid == 0) return hipSuccess;
    const size_t blocks = (n_valid + 255) / 256;
    hipLaunchKernelGGL(band_scatter_kernel, dim3(static_cast<uint32_t>(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, frame, slab, n_valid,
                       static_cast<uint32_t>(band_floats), world, rank);
    return hipGetLastError();
}

hipError_t launch_rays_to_aos(const RayQueue &q, wfpt_ray *out, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rays_to_aos_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, q, out, n);
    return hipGetLastError();
}

hipError_t launch_rays_from_aos(const RayQueue &q, const wfpt_ray *in, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rays_from_aos_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, q, in, n);
    return hipGetLastError();
}

hipError_t launch_selftest_math(int op, const float *a, const float *b, float *out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(selftest_math_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, s, op, a, b, out, n);
    return hipGetLastError();
}

} // namespace wfpt
