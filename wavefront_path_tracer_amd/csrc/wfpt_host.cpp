// wfpt_host.cpp -- host-side data model of the path: scene generators, BVH builder, camera and
// projection set-up, dispatch sizing. C++ restatement of wavefront_common (Rust) so that a host
// without the crate can produce byte-identical inputs for the kernel chain. No GPU code here.
//
// Build with -ffp-contract=off: every expression below is evaluated in plain IEEE binary32, in the
// association order the Rust/glam source uses, so results are reproducible across compilers.
// Citations are relative to the reference root (wc = wavefront_common/src).
#include "wfpt.h"
#include "wfpt_bvh4.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// Box growth (glam Vec3::min / max). For (+0, -0) f32::min / max may return either zero, so the reference leaves the sign
// of a zero bound open; it is fixed here order-independently -- the minimum prefers -0, the maximum +0 -- which is also
// what the device builder's integer atomics give. NaN operands: the other one (fmin / fmax semantics).
inline float zmin(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : (b < a ? b : (std::signbit(a) ? a : b)))); }
inline float zmax(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : (b > a ? b : (std::signbit(a) ? b : a)))); }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {zmin(a.x, b.x), zmin(a.y, b.y), zmin(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {zmax(a.x, b.x), zmax(a.y, b.y), zmax(a.z, b.z)}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; } // glam Vec3::dot
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
// glam Vec3::cross
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
constexpr float kInf = std::numeric_limits<float>::infinity();

// ---- seeded stand-in for rand::thread_rng (wc/util_funcs.rs:6-30): PCG32 XSH-RR, stream 54 ----
class SceneRng {
  public:
    explicit SceneRng(uint64_t seed) : state_(0), inc_((54ULL << 1) | 1ULL) {
        next_u32();
        state_ += seed;
        next_u32();
    }
    uint32_t next_u32() {
        const uint64_t old = state_;
        state_ = old * 6364136223846793005ULL + inc_;
        const uint32_t xs = static_cast<uint32_t>(((old >> 18) ^ old) >> 27);
        const uint32_t rot = static_cast<uint32_t>(old >> 59);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    float f32() { return static_cast<float>(next_u32() >> 8) * (1.0f / 16777216.0f); } // random_f32
    float range(float lo, float hi) { return lo + (hi - lo) * f32(); }                  // random_range_f32
  private:
    uint64_t state_, inc_;
};

wfpt_sphere make_sphere(Vec3 c, float r, uint32_t mat_idx, uint32_t mat_type) { // wc/sphere.rs:18-20
    wfpt_sphere s;
    s.center[0] = c.x; s.center[1] = c.y; s.center[2] = c.z; s.center[3] = 1.0f;
    s.radius = r; s.material_idx = mat_idx; s.material_type = mat_type; s._buffer = 0;
    return s;
}
wfpt_material make_material(Vec3 albedo, float fuzz, float ri, uint32_t type) {
    wfpt_material m;
    m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z; m.albedo[3] = 1.0f;
    m.fuzz = fuzz; m.refract_index = ri; m.material_type = type; m._buffer = 0;
    return m;
}
wfpt_material lambertian(Vec3 a) { return make_material(a, 0.0f, 0.0f, 0); }            // material.rs:26-28
wfpt_material metal(Vec3 a, float fuzz) {                                                 // material.rs:30-32
    return make_material(a, fuzz < 0.0f ? 0.0f : (fuzz > 1.0f ? 1.0f : fuzz), 0.0f, 1);
}
wfpt_material dielectric(float ri) { return make_material({1.0f, 1.0f, 1.0f}, 0.0f, ri, 2); } // material.rs:34-36

// ---- BVH builder (wc/bvh.rs) ----
constexpr int kReferenceBins = 4096; // bvh.rs:4

struct Aabb {
    Vec3 lo{kInf, kInf, kInf}, hi{-kInf, -kInf, -kInf};
    void grow(Vec3 a, Vec3 b) { lo = vmin(lo, a); hi = vmax(hi, b); }
    float half_area() const { // Bin::get_area, bvh.rs:29-35
        if (!(std::isfinite(hi.x) && std::isfinite(hi.y) && std::isfinite(hi.z))) return 0.0f;
        const Vec3 e = hi - lo;
        return (e.x * e.y + e.y * e.z) + e.z * e.x;
    }
};

inline void sphere_bounds(const wfpt_sphere &s, Vec3 &lo, Vec3 &hi) { // sphere.rs:22-26
    lo = {s.center[0] - s.radius, s.center[1] - s.radius, s.center[2] - s.radius};
    hi = {s.center[0] + s.radius, s.center[1] + s.radius, s.center[2] + s.radius};
}

// What bvh.rs needs from a primitive: its box, the coordinate it is binned and partitioned by, a swap.
struct SpherePrims {
    wfpt_sphere *p;
    void bounds(uint32_t i, Vec3 &lo, Vec3 &hi) const { sphere_bounds(p[i], lo, hi); }
    float key(uint32_t i, int axis) const { return p[i].center[axis]; } // bvh.rs:97,179: the sphere's centre
    void swap(long long i, long long j) const { std::swap(p[i], p[j]); }
};
struct TrianglePrims { // build extension
    wfpt_triangle *p;
    void bounds(uint32_t i, Vec3 &lo, Vec3 &hi) const {
        const wfpt_triangle &t = p[i];
        const Vec3 a{t.v0[0], t.v0[1], t.v0[2]};
        const Vec3 b{t.v0[0] + t.e1[0], t.v0[1] + t.e1[1], t.v0[2] + t.e1[2]};
        const Vec3 c{t.v0[0] + t.e2[0], t.v0[1] + t.e2[1], t.v0[2] + t.e2[2]};
        lo = vmin(vmin(a, b), c);
        hi = vmax(vmax(a, b), c);
    }
    float key(uint32_t i, int axis) const { return p[i].v0[axis] + (p[i].e1[axis] + p[i].e2[axis]) * 0.33333334f; }
    void swap(long long i, long long j) const { std::swap(p[i], p[j]); }
};

template <typename Prims> class BvhBuilder {
  public:
    BvhBuilder(Prims prims, wfpt_bvh_node *nodes, int n_bins) : spheres_(prims), nodes_(nodes), kBins(n_bins) {
        bins_.resize(kBins);
        bin_count_.resize(kBins);
        left_count_.resize(kBins - 1);
        right_count_.resize(kBins - 1);
        left_area_.resize(kBins - 1);
        right_area_.resize(kBins - 1);
    }

    uint32_t build(uint32_t n) { // bvh.rs:152-164
        wfpt_bvh_node root{};
        root.left_first = 0;
        root.prim_count = n;
        fit(root);
        nodes_[count_++] = root;
        nodes_[count_++] = wfpt_bvh_node{}; // bvh.rs:160-161 placeholder so siblings sit at (2k, 2k+1)
        subdivide(0);
        return count_;
    }

  private:
    void fit(wfpt_bvh_node &nd) const { // update_node_bounds, bvh.rs:58-70
        Aabb box;
        for (uint32_t i = 0; i < nd.prim_count; ++i) {
            Vec3 lo, hi;
            spheres_.bounds(nd.left_first + i, lo, hi);
            box.grow(lo, hi);
        }
        nd.aabb_min[0] = box.lo.x; nd.aabb_min[1] = box.lo.y; nd.aabb_min[2] = box.lo.z;
        nd.aabb_max[0] = box.hi.x; nd.aabb_max[1] = box.hi.y; nd.aabb_max[2] = box.hi.z;
    }

    struct Split { float cost; int axis; float plane; };

    Split best_split(const wfpt_bvh_node &nd) { // find_best_split_plane, bvh.rs:73-139
        const float extent[3] = {nd.aabb_max[0] - nd.aabb_min[0], nd.aabb_max[1] - nd.aabb_min[1],
                                 nd.aabb_max[2] - nd.aabb_min[2]};
        Split best{kInf, 0, 0.0f};
        for (int axis = 0; axis < 3; ++axis) {
            if (extent[axis] < 0.00001f) continue;
            for (int b = 0; b < kBins; ++b) { bins_[b] = Aabb{}; bin_count_[b] = 0; }
            const float scale = static_cast<float>(kBins) / extent[axis];
            const float lo_bound = nd.aabb_min[axis];
            for (uint32_t i = 0; i < nd.prim_count; ++i) {
                const float pos = (spheres_.key(nd.left_first + i, axis) - lo_bound) * scale;
                // Rust `as usize` saturates (negative/NaN -> 0), then .min(BINS - 1)
                long long b = pos > 0.0f ? (pos >= 2147483648.0f ? 2147483647LL : static_cast<long long>(pos)) : 0;
                if (b > kBins - 1) b = kBins - 1;
                Vec3 lo, hi;
                spheres_.bounds(nd.left_first + i, lo, hi);
                bins_[b].grow(lo, hi);
                bin_count_[b] += 1;
            }
            // prefix sweeps from both ends; N bins -> N-1 candidate planes
            Aabb left, right;
            uint32_t nl = 0, nr = 0;
            for (int i = 0; i < kBins - 1; ++i) {
                nl += bin_count_[i];
                left_count_[i] = nl;
                left.grow(bins_[i].lo, bins_[i].hi);
                left_area_[i] = left.half_area();
                nr += bin_count_[kBins - 1 - i];
                right_count_[kBins - 2 - i] = nr;
                right.grow(bins_[kBins - 1 - i].lo, bins_[kBins - 1 - i].hi);
                right_area_[kBins - 2 - i] = right.half_area();
            }
            const float step = 1.0f / static_cast<float>(kBins);
            for (int i = 0; i < kBins - 1; ++i) {
                const float cost = static_cast<float>(left_count_[i]) * left_area_[i] +
                                   static_cast<float>(right_count_[i]) * right_area_[i];
                if (cost < best.cost) {
                    best.cost = cost;
                    best.axis = axis;
                    best.plane = lo_bound + extent[axis] * step * (1.0f + static_cast<float>(i));
                }
            }
        }
        return best;
    }

    void subdivide(uint32_t index) { // bvh.rs:166-210
        const Split split = best_split(nodes_[index]);
        const float ex = nodes_[index].aabb_max[0] - nodes_[index].aabb_min[0];
        const float ey = nodes_[index].aabb_max[1] - nodes_[index].aabb_min[1];
        const float ez = nodes_[index].aabb_max[2] - nodes_[index].aabb_min[2];
        const float leaf_cost = static_cast<float>(nodes_[index].prim_count) * ((ex * ey + ey * ez) + ez * ex);
        if (leaf_cost <= split.cost) return;

        // in-place partition; signed cursors (the reference's usize `j -= 1` can underflow)
        long long i = nodes_[index].left_first;
        long long j = i + static_cast<long long>(nodes_[index].prim_count) - 1;
        while (i <= j) {
            if (spheres_.key(static_cast<uint32_t>(i), split.axis) < split.plane) {
                ++i;
            } else {
                spheres_.swap(i, j);
                --j;
            }
        }
        const uint32_t n_left = static_cast<uint32_t>(i) - nodes_[index].left_first;
        if (n_left == 0 || n_left == nodes_[index].prim_count) return;

        const uint32_t child = count_;
        wfpt_bvh_node l{}, r{};
        l.left_first = nodes_[index].left_first;
        l.prim_count = n_left;
        fit(l);
        r.left_first = static_cast<uint32_t>(i);
        r.prim_count = nodes_[index].prim_count - n_left;
        fit(r);
        nodes_[index].left_first = child;
        nodes_[index].prim_count = 0;
        nodes_[count_++] = l;
        nodes_[count_++] = r;
        subdivide(child);
        subdivide(child + 1);
    }

    Prims spheres_; // named after bvh.rs's `spheres` argument
    wfpt_bvh_node *nodes_;
    const int kBins;
    uint32_t count_ = 0;
    std::vector<Aabb> bins_;
    std::vector<uint32_t> bin_count_, left_count_, right_count_;
    std::vector<float> left_area_, right_area_;
};

const char *const kStageNames[WFPT_STAGE_COUNT] = {"generate_rays",    "extend",      "shade",           "miss_kernel",
                                                   "accumulate",       "shade_lambertian", "shade_metal", "shade_dielectric",
                                                   "scan",             "bounce_first", "bounce",          "bounce_last",
                                                   "compact"};

} // namespace

// Collapses the binary tree (reference numbering, siblings at (2k, 2k+1)) into four-wide nodes: the children of node N are
// its grandchildren where a child is an inner node, and the child itself where it is a leaf; the child boxes, each grown by
// `margin` per axis (the rounding allowance of the device's one-fma plane distances, see visit4), are quantised to 8 bits per
// plane in the frame of their union, rounded OUTWARDS under the dequantisation arithmetic (plane = fmaf(q, 2^e, origin)), so
// a quantised box always encloses the caller's box plus the margin. Nodes are numbered BREADTH FIRST: any prefix of the array
// is a top of the tree (the traversal stages such a prefix in LDS). Returns false when a leaf cannot be written as a child
// word (more than wfpt::kLeafMaxCount primitives or an index beyond 28 bits) or a box is not finite: the caller then keeps
// the binary traversal. `depth4` = levels of four-wide nodes below the root node.
bool wfpt::collapse_bvh4(const wfpt_bvh_node *nodes, uint32_t n_nodes, const float margin[3], std::vector<wfpt::Node4> &out, uint32_t &depth4) {
    out.clear();
    depth4 = 0;
    auto leaf_word = [&](const wfpt_bvh_node &nd, uint32_t &w) {
        if (nd.prim_count > wfpt::kLeafMaxCount || nd.left_first > wfpt::kLeafFirstMask) return false;
        w = wfpt::kLeafFlag | (nd.prim_count << wfpt::kLeafCountShift) | nd.left_first;
        return true;
    };
    // quantises the boxes of `kids` (binary node indices) into nd4; false if a bound is not finite
    auto quantise = [&](wfpt::Node4 &nd4, const uint32_t *kids, uint32_t n_kids) {
        std::memset(&nd4, 0, sizeof nd4);
        for (int ax = 0; ax < 3; ++ax) {
            float lo = INFINITY, hi = -INFINITY;
            for (uint32_t k = 0; k < n_kids; ++k) {
                lo = std::min(lo, nodes[kids[k]].aabb_min[ax] - margin[ax]);
                hi = std::max(hi, nodes[kids[k]].aabb_max[ax] + margin[ax]);
            }
            if (!std::isfinite(lo) || !std::isfinite(hi) || hi < lo) return false;
            nd4.origin[ax] = lo;
            int e = 127;
            (void)std::frexp((hi - lo) / 255.0f, &e); // (hi - lo) / 255 = m * 2^e, 0.5 <= m < 1  =>  255 * 2^e >= hi - lo
            int biased = std::min(std::max(e + 127, 1), 254);
            for (;; ++biased) { // grow the scale until every upper plane fits 8 bits under the device's arithmetic
                if (biased > 254) return false;
                float scale;
                const uint32_t bits = static_cast<uint32_t>(biased) << 23;
                std::memcpy(&scale, &bits, 4);
                bool fits = true;
                for (uint32_t k = 0; k < n_kids && fits; ++k) {
                    const float cmin = nodes[kids[k]].aabb_min[ax] - margin[ax], cmax = nodes[kids[k]].aabb_max[ax] + margin[ax];
                    int ql = static_cast<int>(std::floor((cmin - lo) / scale));
                    ql = std::min(std::max(ql, 0), 255);
                    while (ql > 0 && std::fmaf(static_cast<float>(ql), scale, lo) > cmin) --ql; // q = 0 gives lo <= cmin exactly
                    int qh = static_cast<int>(std::ceil((cmax - lo) / scale));
                    qh = std::min(std::max(qh, 0), 255);
                    while (qh < 255 && std::fmaf(static_cast<float>(qh), scale, lo) < cmax) ++qh;
                    if (std::fmaf(static_cast<float>(qh), scale, lo) < cmax) { fits = false; break; }
                    nd4.qlo[ax][k] = static_cast<uint8_t>(ql);
                    nd4.qhi[ax][k] = static_cast<uint8_t>(qh);
                }
                if (fits) { nd4.exp[ax] = static_cast<uint8_t>(biased); nd4.scale_hi[ax] = static_cast<uint16_t>(biased << 7); break; }
            }
            for (uint32_t k = n_kids; k < 4; ++k) { nd4.qlo[ax][k] = 255; nd4.qhi[ax][k] = 0; } // absent children: inverted, never entered
        }
        return true;
    };
    struct Item { uint32_t bin, slot, depth; }; // binary inner node -> its four-wide node `slot`
    std::vector<Item> todo;
    out.emplace_back();
    if (nodes[0].prim_count > 0) { // a single leaf: a root node with one leaf child (its box is never tested by the reference either)
        const uint32_t kid = 0;
        if (!quantise(out[0], &kid, 1)) return false;
        for (int k = 0; k < 4; ++k) out[0].child[k] = wfpt::kEmptyChild;
        return leaf_word(nodes[0], out[0].child[0]);
    }
    todo.push_back({0u, 0u, 0u});
    for (size_t head = 0; head < todo.size(); ++head) { // first in, first out: slots are handed out level by level
        const Item it = todo[head];
        depth4 = std::max(depth4, it.depth);
        uint32_t kids[4], n_kids = 0;
        const uint32_t l = nodes[it.bin].left_first;
        for (uint32_t c = l; c <= l + 1u; ++c) {
            if (nodes[c].prim_count > 0) kids[n_kids++] = c;
            else { kids[n_kids++] = nodes[c].left_first; kids[n_kids++] = nodes[c].left_first + 1u; }
        }
        wfpt::Node4 nd4;
        if (!quantise(nd4, kids, n_kids)) return false;
        for (uint32_t k = 0; k < 4; ++k) {
            if (k >= n_kids) { nd4.child[k] = wfpt::kEmptyChild; continue; }
            const wfpt_bvh_node &ch = nodes[kids[k]];
            if (ch.prim_count > 0) {
                if (!leaf_word(ch, nd4.child[k])) return false;
            } else {
                if (out.size() >= wfpt::kLeafFlag) return false;
                nd4.child[k] = static_cast<uint32_t>(out.size());
                todo.push_back({kids[k], static_cast<uint32_t>(out.size()), it.depth + 1u});
                out.emplace_back();
            }
        }
        out[it.slot] = nd4;
    }
    (void)n_nodes;
    return true;
}


extern "C" int wfpt_debug_bvh4(const wfpt_bvh_node *nodes, uint32_t n_nodes, uint32_t counts[4]) {
    // Test hook (no GPU needed): collapses a binary tree into the quantised four-wide nodes the device walks and checks,
    // with the device's dequantisation arithmetic, that every child box encloses the binary node's box it stands for and
    // that every leaf / inner node of the binary tree is reachable exactly once. counts = {four-wide nodes, depth,
    // leaf children, inner children}. Returns WFPT_OK, WFPT_ERR_UNSUPPORTED if the tree cannot be collapsed, or
    // WFPT_ERR_INVALID_ARGUMENT on a violated property.
    if (!nodes || n_nodes == 0 || !counts) return WFPT_ERR_INVALID_ARGUMENT;
    std::vector<wfpt::Node4> n4;
    uint32_t depth4 = 0;
    const float no_margin[3] = {0.0f, 0.0f, 0.0f};
    if (!wfpt::collapse_bvh4(nodes, n_nodes, no_margin, n4, depth4)) return WFPT_ERR_UNSUPPORTED;
    // walk both trees together: four-wide node `slot` stands for binary inner node `bin`
    struct Item { uint32_t bin, slot; };
    std::vector<Item> todo;
    uint32_t leaves = 0, inners = 0;
    std::vector<uint8_t> seen(n4.size(), 0);
    if (nodes[0].prim_count == 0) todo.push_back({0u, 0u});
    else leaves = 1;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.slot >= n4.size() || seen[it.slot]++) return WFPT_ERR_INVALID_ARGUMENT;
        const wfpt::Node4 &nd = n4[it.slot];
        uint32_t kids[4], n_kids = 0;
        const uint32_t l = nodes[it.bin].left_first;
        for (uint32_t c = l; c <= l + 1u; ++c) {
            if (nodes[c].prim_count > 0) kids[n_kids++] = c;
            else { kids[n_kids++] = nodes[c].left_first; kids[n_kids++] = nodes[c].left_first + 1u; }
        }
        for (uint32_t k = 0; k < 4; ++k) {
            if (k >= n_kids) { if (nd.child[k] != wfpt::kEmptyChild) return WFPT_ERR_INVALID_ARGUMENT; continue; }
            const wfpt_bvh_node &ch = nodes[kids[k]];
            for (int ax = 0; ax < 3; ++ax) {
                float scale;
                const uint32_t bits = static_cast<uint32_t>(nd.exp[ax]) << 23;
                std::memcpy(&scale, &bits, 4);
                const float lo = std::fmaf(static_cast<float>(nd.qlo[ax][k]), scale, nd.origin[ax]);
                const float hi = std::fmaf(static_cast<float>(nd.qhi[ax][k]), scale, nd.origin[ax]);
                if (!(lo <= ch.aabb_min[ax]) || !(hi >= ch.aabb_max[ax])) return WFPT_ERR_INVALID_ARGUMENT;
            }
            if (ch.prim_count > 0) {
                if (nd.child[k] != (wfpt::kLeafFlag | (ch.prim_count << wfpt::kLeafCountShift) | ch.left_first)) return WFPT_ERR_INVALID_ARGUMENT;
                ++leaves;
            } else {
                if (nd.child[k] & wfpt::kLeafFlag) return WFPT_ERR_INVALID_ARGUMENT;
                ++inners;
                todo.push_back({kids[k], nd.child[k]});
            }
        }
    }
    counts[0] = static_cast<uint32_t>(n4.size()); counts[1] = depth4; counts[2] = leaves; counts[3] = inners;
    return WFPT_OK;
}

extern "C" {

uint32_t wfpt_scene_new(wfpt_sphere *sp, wfpt_material *mt) { // scene.rs:12-46
    mt[0] = lambertian({0.8f, 0.8f, 0.0f});
    mt[1] = lambertian({0.1f, 0.2f, 0.5f});
    mt[2] = dielectric(1.50f);
    mt[3] = metal({0.8f, 0.6f, 0.2f}, 1.0f);
    mt[4] = dielectric(1.00f / 1.50f);
    // vec![ground, center, right, left, bubble]
    sp[0] = make_sphere({0.0f, -100.5f, -1.0f}, 100.0f, 0, 0);
    sp[1] = make_sphere({0.0f, 0.0f, -1.2f}, 0.5f, 1, 0);
    sp[2] = make_sphere({1.0f, 0.0f, -1.0f}, 0.5f, 3, 1);
    sp[3] = make_sphere({-1.0f, 0.0f, -1.0f}, 0.5f, 2, 2);
    sp[4] = make_sphere({-1.0f, 0.0f, -1.0f}, 0.4f, 4, 2);
    return 5;
}

uint32_t wfpt_scene_book_one_final(uint64_t seed, wfpt_sphere *sp, wfpt_material *mt, uint32_t capacity) {
    if (capacity < 488) return 0; // 1 ground + 22*22 marbles + 3 big spheres
    SceneRng rng(seed);
    uint32_t n = 0;
    auto push = [&](const wfpt_material &m, Vec3 c, float r) {
        mt[n] = m;
        sp[n] = make_sphere(c, r, n, m.material_type); // (materials.len() - 1) as u32
        ++n;
    };
    push(lambertian({0.5f, 0.5f, 0.5f}), {0.0f, -1000.0f, 0.0f}, 1000.0f); // scene.rs:52-59
    for (int a = -11; a < 11; ++a) {                                          // scene.rs:62-91
        for (int b = -11; b < 11; ++b) {
            const float choose_mat = rng.f32();
            const float cx = static_cast<float>(a) + 0.9f * rng.f32();
            const float cz = static_cast<float>(b) + 0.9f * rng.f32();
            const Vec3 center{cx, 0.2f, cz};
            if (length(center - Vec3{4.0f, 0.2f, 0.0f}) > 0.9f) {
                if (choose_mat < 0.8f) {
                    const float r1 = rng.f32(), g1 = rng.f32(), b1 = rng.f32(); // random_vec3() * random_vec3()
                    const float r2 = rng.f32(), g2 = rng.f32(), b2 = rng.f32();
                    push(lambertian({r1 * r2, g1 * g2, b1 * b2}), center, 0.2f);
                } else if (choose_mat < 0.95f) {
                    const float r = rng.range(0.5f, 1.0f), g = rng.range(0.5f, 1.0f), bl = rng.range(0.5f, 1.0f);
                    const float fuzz = rng.range(0.0f, 0.5f);
                    push(metal({r, g, bl}, fuzz), center, 0.2f);
                } else {
                    push(dielectric(1.5f), center, 0.2f);
                }
            }
        }
    }
    push(dielectric(1.50f), {0.0f, 1.0f, 0.0f}, 1.0f);              // scene.rs:94-96
    push(lambertian({0.4f, 0.2f, 0.1f}), {-4.0f, 1.0f, 0.0f}, 1.0f); // scene.rs:98-100
    push(metal({0.7f, 0.6f, 0.5f}, 0.0f), {4.0f, 1.0f, 0.0f}, 1.0f); // scene.rs:102-104
    return n;
}

int wfpt_build_bvh(wfpt_sphere *spheres, uint32_t n, wfpt_bvh_node *nodes, uint32_t cap, uint32_t *n_nodes) {
    if (!spheres || !nodes || !n_nodes || n == 0) return WFPT_ERR_INVALID_ARGUMENT;
    if (cap < 2 * n) return WFPT_ERR_INVALID_ARGUMENT; // BVHTree::new reserves 2*n (bvh.rs:148-150)
    BvhBuilder<SpherePrims> builder(SpherePrims{spheres}, nodes, kReferenceBins);
    *n_nodes = builder.build(n);
    return WFPT_OK;
}

int wfpt_build_bvh_triangles(wfpt_triangle *tris, uint32_t n, wfpt_bvh_node *nodes, uint32_t cap, uint32_t *n_nodes,
                             uint32_t n_bins) {
    if (!tris || !nodes || !n_nodes || n == 0) return WFPT_ERR_INVALID_ARGUMENT;
    if (cap < 2 * n || n_bins > (1u << 20)) return WFPT_ERR_INVALID_ARGUMENT;
    BvhBuilder<TrianglePrims> builder(TrianglePrims{tris}, nodes, static_cast<int>(n_bins < 2 ? 2 : n_bins));
    *n_nodes = builder.build(n);
    return WFPT_OK;
}

int wfpt_load_obj(const char *path, wfpt_triangle *tris, uint32_t capacity, uint32_t *n_tris, uint32_t material_idx,
                  uint32_t material_type) {
    if (!path || !n_tris) return WFPT_ERR_INVALID_ARGUMENT;
    std::FILE *f = std::fopen(path, "r");
    if (!f) return WFPT_ERR_INVALID_ARGUMENT;
    std::vector<Vec3> verts;
    std::vector<long long> face;
    uint32_t count = 0;
    int status = WFPT_OK;
    char line[4096];
    while (status == WFPT_OK && std::fgets(line, sizeof line, f)) {
        const char *p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            Vec3 v{0.0f, 0.0f, 0.0f};
            if (std::sscanf(p + 1, "%f %f %f", &v.x, &v.y, &v.z) != 3) status = WFPT_ERR_INVALID_ARGUMENT;
            verts.push_back(v);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            face.clear();
            char *q = const_cast<char *>(p + 1);
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == 0 || *q == '\n' || *q == '\r' || *q == '#') break;
                char *end = nullptr;
                long long idx = std::strtoll(q, &end, 10); // the vertex index; "/t/n" parts are skipped
                if (end == q) { status = WFPT_ERR_INVALID_ARGUMENT; break; }
                while (*end && *end != ' ' && *end != '\t' && *end != '\n' && *end != '\r') ++end;
                q = end;
                if (idx < 0) idx += static_cast<long long>(verts.size()); // relative to the vertices read so far
                else idx -= 1;
                if (idx < 0 || idx >= static_cast<long long>(verts.size())) { status = WFPT_ERR_INVALID_ARGUMENT; break; }
                face.push_back(idx);
            }
            for (size_t k = 2; status == WFPT_OK && k < face.size(); ++k) { // fan around the first vertex
                if (tris) {
                    if (count >= capacity) { status = WFPT_ERR_INVALID_ARGUMENT; break; }
                    const Vec3 a = verts[face[0]], b = verts[face[k - 1]], c = verts[face[k]];
                    wfpt_triangle &t = tris[count];
                    t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z;
                    t.e1[0] = b.x - a.x; t.e1[1] = b.y - a.y; t.e1[2] = b.z - a.z;
                    t.e2[0] = c.x - a.x; t.e2[1] = c.y - a.y; t.e2[2] = c.z - a.z;
                    t.material_idx = material_idx;
                    t.material_type = material_type;
                    t._pad = 0;
                }
                count += 1;
            }
        }
    }
    std::fclose(f);
    *n_tris = count;
    return status;
}

uint32_t wfpt_scene_random_mesh(uint64_t seed, uint32_t n, wfpt_triangle *tris, wfpt_material *mt) {
    SceneRng rng(seed);
    mt[0] = lambertian({0.7f, 0.7f, 0.7f});
    mt[1] = metal({0.8f, 0.8f, 0.8f}, 0.1f);
    mt[2] = dielectric(1.5f);
    for (uint32_t i = 0; i < n; ++i) {
        float c[3], e1[3], e2[3];
        for (float &v : c) v = rng.range(-10.0f, 10.0f);
        for (float &v : e1) v = rng.range(-0.05f, 0.05f);
        for (float &v : e2) v = rng.range(-0.05f, 0.05f);
        wfpt_triangle &t = tris[i];
        for (int k = 0; k < 3; ++k) {
            t.v0[k] = c[k] - (e1[k] + e2[k]) * 0.33333334f; // the centre is the centroid
            t.e1[k] = e1[k];
            t.e2[k] = e2[k];
        }
        t.material_idx = t.material_type = i % 3u;
        t._pad = 0;
    }
    return 3;
}

void wfpt_camera_new(const float from[3], const float at[3], float *pitch, float *yaw) { // camera.rs:11-24
    Vec3 f{at[0] - from[0], at[1] - from[1], at[2] - from[2]};
    const float inv_len = 1.0f / length(f); // glam normalize(): self * length_recip()
    f = {f.x * inv_len, f.y * inv_len, f.z * inv_len};
    *pitch = std::acos(f.y);
    *yaw = std::atan2(f.x, f.z);
}

void wfpt_view_transform(const float pos[3], float pitch, float yaw, float view[16]) { // camera.rs:41-69
    const float sp = std::sin(pitch), cp = std::cos(pitch), sy = std::sin(yaw), cy = std::cos(yaw);
    const Vec3 dir{sp * sy, cp, sp * cy};
    const Vec3 right = cross(dir, Vec3{0.0f, 1.0f, 0.0f});
    const Vec3 up = cross(right, dir);
    const float cols[16] = {right.x, right.y, right.z, 0.0f, up.x,   up.y,   up.z,   0.0f,
                            dir.x,   dir.y,   dir.z,   0.0f, pos[0], pos[1], pos[2], 1.0f};
    std::memcpy(view, cols, sizeof cols);
}

void wfpt_p_inv(float vfov_rad, float aspect, float z_near, float z_far, float out[16]) { // projection_matrix.rs:21-37
    const float h = std::tan(vfov_rad / 2.0f);
    const float w = h * aspect;
    const float r = z_far / (z_far - z_near);
    const float cols[16] = {w,    0.0f, 0.0f, 0.0f,                    0.0f, h,    0.0f, 0.0f,
                            0.0f, 0.0f, 0.0f, -1.0f / (r * z_near),   0.0f, 0.0f, 1.0f, 1.0f / z_near};
    std::memcpy(out, cols, sizeof cols);
}

void wfpt_gpu_camera_new(const float pos[3], float pitch, float yaw, float defocus_angle_rad, float focus_distance,
                         wfpt_gpu_camera *out) { // camera_controller.rs:173-185
    out->position[0] = pos[0]; out->position[1] = pos[1]; out->position[2] = pos[2]; out->position[3] = 1.0f;
    out->pitch = pitch;
    out->yaw = yaw;
    out->defocus_radius = focus_distance * std::tan(0.5f * defocus_angle_rad);
    out->focus_distance = focus_distance;
}

float wfpt_to_radians(float deg) { return deg * 0.017453292519943295769236907684886f; }

void wfpt_camera_controller_update(float position[3], float *pitch, float *yaw, const float amounts[6], float rotate[2],
                                   float speed, float sensitivity, float dt) { // camera_controller.rs:125-158
    const float sin_yaw = std::sin(*yaw), cos_yaw = std::cos(*yaw);
    const float forward[3] = {sin_yaw, 0.0f, cos_yaw}, right[3] = {-cos_yaw, 0.0f, sin_yaw};
    const float fb = amounts[0] - amounts[1], rl = amounts[2] - amounts[3];
    for (int k = 0; k < 3; ++k) position[k] += forward[k] * fb * speed * dt; // :131
    for (int k = 0; k < 3; ++k) position[k] += right[k] * rl * speed * dt;   // :132
    position[1] += (amounts[4] - amounts[5]) * speed * dt;                   // :137: no roll, so y moves directly
    *yaw -= rotate[0] * sensitivity * dt;                                    // :140-141
    *pitch -= rotate[1] * sensitivity * dt;
    rotate[0] = rotate[1] = 0.0f;                                            // :146-147
    const float safe = 3.14159265358979323846f - 0.001f;                     // SAFE_FRAC_PI, :29
    if (*pitch < -safe) *pitch = -safe;
    else if (*pitch > safe) *pitch = safe;
}

void wfpt_workgroup_size_64(uint32_t x, uint32_t *gx, uint32_t *gy) { // path_tracer.rs:282-289
    const uint32_t groups = x / 64u + (x % 64u ? 1u : 0u); // div_ceil
    if (groups <= 1) { *gx = 1; *gy = 1; return; }         // the reference unwraps None and panics here
    const uint32_t y = static_cast<uint32_t>(std::ceil(std::sqrt(static_cast<float>(groups))));
    uint32_t fac = 1;
    for (uint32_t z = y - 1; z >= 1; --z) { // (1..y).rev().find(|z| groups % z == 0)
        if (groups % z == 0) { fac = z; break; }
    }
    if (groups / fac >= (1u << 16)) { *gx = y; *gy = y; } else { *gx = fac; *gy = groups / fac; }
}

int wfpt_stage_from_name(const char *name) {
    if (!name) return -1;
    for (int s = 0; s < WFPT_STAGE_COUNT; ++s)
        if (std::strcmp(name, kStageNames[s]) == 0) return s;
    return -1;
}
const char *wfpt_stage_name(int stage) { return (stage >= 0 && stage < WFPT_STAGE_COUNT) ? kStageNames[stage] : ""; }

void wfpt_tonemap_rgb8(const float *acc, uint32_t n_pixels, uint32_t n_samples, uint8_t *rgb) {
    // display_shader.wgsl:50-52: color = sqrt(invN * color); the 8-bit target clamps to [0,1]
    const float inv_n = 1.0f / static_cast<float>(n_samples);
    for (size_t i = 0; i < 3 * static_cast<size_t>(n_pixels); ++i) {
        float v = std::sqrt(inv_n * acc[i]);
        if (!(v > 0.0f)) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        rgb[i] = static_cast<uint8_t>(v * 255.0f + 0.5f);
    }
}

// PNG (8-bit RGB, no interlace) with the pixel rows in deflate "stored" blocks: no compression library, any viewer reads it.
static uint32_t png_crc(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
static bool png_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data) {
    const uint32_t n = static_cast<uint32_t>(data.size());
    const uint8_t len[4] = {uint8_t(n >> 24), uint8_t(n >> 16), uint8_t(n >> 8), uint8_t(n)};
    uint32_t crc = png_crc(0xffffffffu, reinterpret_cast<const uint8_t *>(type), 4);
    crc = png_crc(crc, data.data(), data.size()) ^ 0xffffffffu;
    const uint8_t tail[4] = {uint8_t(crc >> 24), uint8_t(crc >> 16), uint8_t(crc >> 8), uint8_t(crc)};
    return std::fwrite(len, 1, 4, f) == 4 && std::fwrite(type, 1, 4, f) == 4 &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) && std::fwrite(tail, 1, 4, f) == 4;
}
int wfpt_write_png_rgb8(const char *path, const uint8_t *rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || width == 0 || height == 0) return WFPT_ERR_INVALID_ARGUMENT;
    const size_t row = 3 * static_cast<size_t>(width);
    std::vector<uint8_t> raw((row + 1) * height); // every row behind its filter byte (0: none)
    for (uint32_t y = 0; y < height; ++y) {
        raw[y * (row + 1)] = 0;
        std::memcpy(&raw[y * (row + 1) + 1], rgb + y * row, row);
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01); // zlib header: deflate, 32 K window, no preset dictionary
    uint32_t a = 1, b = 0; // adler32 of the raw data
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0); // BFINAL, BTYPE = 00 (stored)
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = off; i < off + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    const uint32_t adler = (b << 16) | a;
    z.push_back(uint8_t(adler >> 24)); z.push_back(uint8_t(adler >> 16)); z.push_back(uint8_t(adler >> 8)); z.push_back(uint8_t(adler));
    FILE *f = std::fopen(path, "wb");
    if (!f) return WFPT_ERR_INVALID_ARGUMENT;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr = {uint8_t(width >> 24), uint8_t(width >> 16), uint8_t(width >> 8), uint8_t(width),
                                 uint8_t(height >> 24), uint8_t(height >> 16), uint8_t(height >> 8), uint8_t(height),
                                 8, 2, 0, 0, 0}; // 8 bits per channel, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    const bool ok = std::fwrite(sig, 1, 8, f) == 8 && png_chunk(f, "IHDR", ihdr) && png_chunk(f, "IDAT", z) && png_chunk(f, "IEND", {});
    std::fclose(f);
    return ok ? WFPT_OK : WFPT_ERR_INVALID_ARGUMENT;
}

} // extern "C"
