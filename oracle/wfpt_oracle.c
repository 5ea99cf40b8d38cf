/*
 * wfpt_oracle.c -- ORACLE: scalar CPU restatement of the reference's wavefront kernel chain
 * (generate_rays -> extend -> shade -> miss_kernel -> accumulate) and of its host loop.
 *
 * TEST INFRASTRUCTURE ONLY (see wfpt_oracle.h). "Parity unpinned" for floating point: the reference
 * holds no fixtures and cannot be built here; integer pieces are pinned by known-answer tests.
 *
 * Determinism rule (the reference is racy): every atomicAdd is resolved in
 * ascending global thread index, i.e. queues are stable compactions. That is one legal execution of
 * the WGSL. Loops marked `omp parallel for` touch disjoint outputs, so the OpenMP build is
 * bit-identical to the serial one.
 *
 * All paths cited are relative to the reference root; "gr" = gpu_wavefront_pt/shaders/generate_rays.wgsl,
 * "ex" = .../extend.wgsl, "sh" = .../shade.wgsl, "mk" = .../miss_kernel.wgsl, "ac" = .../accumulate.wgsl,
 * "pt" = gpu_wavefront_pt/src/path_tracer.rs, "wc" = wavefront_common/src.
 */
#include "wfpt_oracle.h"
#include "orc_math.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.1415927f /* gr:2, sh:2 */
#define ORC_MAX_STACK 128  /* ex:38 has STACKSIZE = 10 and no overflow check; see orc_trace_bvh */

/* ------------------------------------------------------------------------------------------------
 * small vector helpers: fixed association order, no fused operations (build with -ffp-contract=off)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v3_length(v3 a) { return orc_sqrt(v3_dot(a, a)); }
/* WGSL normalize(v): the built-in's accuracy is "inherited from v / length(v)", i.e. 2.5 ULP per component on top of length's. The
 * definition CHOSEN here (round 4; the device states the same one independently, wfpt_device_math.h: normalize3) is v * (1 / length(v)): one
 * correctly rounded division and three products per vector, at most 1.5 ULP away from v / length(v) per component
 * (tests/test_oracle_math.py::test_normalize_accuracy), inside WGSL's bound. The reference holds no fixture that pins the built-in's
 * rounding, so neither this form nor the per-component division of rounds 1-3 is "the reference's": parity of normalize is unpinned. */
static inline v3 v3_normalize(v3 a) { float inv = 1.0f / v3_length(a); return v3_make(a.x * inv, a.y * inv, a.z * inv); }
static inline float v4_dot(v4 a, v4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline v4 v4_normalize(v4 a) {
    float inv = 1.0f / orc_sqrt(v4_dot(a, a));
    v4 r = {a.x * inv, a.y * inv, a.z * inv, a.w * inv};
    return r;
}
/* WGSL mat4x4f * vec4f with m stored column-major: sum of columns scaled by the vector's components */
static inline v4 m4_mul(const float m[16], v4 v) {
    v4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * scene generation. The reference draws from an UNSEEDED rand::thread_rng (wc/util_funcs.rs:6-30), so
 * it is not reproducible; the build substitutes PCG32 (XSH-RR 64/32) with the same distributions and
 * the same draw order. f32 = top 24 bits * 2^-24 (what rand's Standard f32 does); range = a + (b-a)*u.
 * ---------------------------------------------------------------------------------------------- */
static uint32_t pcg32_next(uint64_t s[2]) {
    uint64_t old = s[0];
    s[0] = old * 6364136223846793005ULL + s[1];
    uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
void orc_scene_rng_seed(uint64_t s[2], uint64_t seed) {
    s[0] = 0;
    s[1] = (54ULL << 1) | 1ULL;
    pcg32_next(s);
    s[0] += seed;
    pcg32_next(s);
}
float orc_scene_rng_f32(uint64_t s[2]) { return (float)(pcg32_next(s) >> 8) * (1.0f / 16777216.0f); }
static float scene_range(uint64_t s[2], float a, float b) { return a + (b - a) * orc_scene_rng_f32(s); }

/* wc/sphere.rs:18-20 */
static orc_sphere sphere_new(float cx, float cy, float cz, float r, uint32_t mat_idx, uint32_t mat_type) {
    orc_sphere s = {{cx, cy, cz, 1.0f}, r, mat_idx, mat_type, 0};
    return s;
}
/* wc/material.rs:26-36 */
static orc_material mat_lambertian(float r, float g, float b) {
    orc_material m = {{r, g, b, 1.0f}, 0.0f, 0.0f, 0, 0};
    return m;
}
static orc_material mat_metal(float r, float g, float b, float fuzz) {
    if (fuzz < 0.0f) fuzz = 0.0f;
    if (fuzz > 1.0f) fuzz = 1.0f;
    orc_material m = {{r, g, b, 1.0f}, fuzz, 0.0f, 1, 0};
    return m;
}
static orc_material mat_dielectric(float ri) {
    orc_material m = {{1.0f, 1.0f, 1.0f, 1.0f}, 0.0f, ri, 2, 0};
    return m;
}

/* wc/scene.rs:12-46 */
uint32_t orc_scene_new(orc_sphere *sp, orc_material *mt) {
    mt[0] = mat_lambertian(0.8f, 0.8f, 0.0f);  /* ground */
    mt[1] = mat_lambertian(0.1f, 0.2f, 0.5f);  /* center */
    mt[2] = mat_dielectric(1.50f);             /* left */
    mt[3] = mat_metal(0.8f, 0.6f, 0.2f, 1.0f); /* right */
    mt[4] = mat_dielectric(1.00f / 1.50f);     /* bubble */
    sp[0] = sphere_new(0.0f, -100.5f, -1.0f, 100.0f, 0, 0);
    sp[1] = sphere_new(0.0f, 0.0f, -1.2f, 0.5f, 1, 0);
    sp[2] = sphere_new(1.0f, 0.0f, -1.0f, 0.5f, 3, 1);  /* right */
    sp[3] = sphere_new(-1.0f, 0.0f, -1.0f, 0.5f, 2, 2); /* left */
    sp[4] = sphere_new(-1.0f, 0.0f, -1.0f, 0.4f, 4, 2); /* bubble */
    return 5;
}

/* wc/scene.rs:48-107; draw order per cell: choose_mat, x, z, then the material's own draws */
uint32_t orc_scene_book_one_final(uint64_t seed, orc_sphere *sp, orc_material *mt) {
    uint64_t rng[2];
    orc_scene_rng_seed(rng, seed);
    uint32_t n = 0;
    mt[n] = mat_lambertian(0.5f, 0.5f, 0.5f);
    sp[n] = sphere_new(0.0f, -1000.0f, 0.0f, 1000.0f, 0, 0);
    n++;
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose_mat = orc_scene_rng_f32(rng);
            float cx = (float)a + 0.9f * orc_scene_rng_f32(rng);
            float cy = 0.2f;
            float cz = (float)b + 0.9f * orc_scene_rng_f32(rng);
            v3 d = v3_make(cx - 4.0f, cy - 0.2f, cz - 0.0f);
            if (v3_length(d) > 0.9f) {
                if (choose_mat < 0.8f) {
                    float r1 = orc_scene_rng_f32(rng), g1 = orc_scene_rng_f32(rng), b1 = orc_scene_rng_f32(rng);
                    float r2 = orc_scene_rng_f32(rng), g2 = orc_scene_rng_f32(rng), b2 = orc_scene_rng_f32(rng);
                    mt[n] = mat_lambertian(r1 * r2, g1 * g2, b1 * b2);
                    sp[n] = sphere_new(cx, cy, cz, 0.2f, n, 0);
                } else if (choose_mat < 0.95f) {
                    float r = scene_range(rng, 0.5f, 1.0f), g = scene_range(rng, 0.5f, 1.0f),
                          bl = scene_range(rng, 0.5f, 1.0f);
                    float fuzz = scene_range(rng, 0.0f, 0.5f);
                    mt[n] = mat_metal(r, g, bl, fuzz);
                    sp[n] = sphere_new(cx, cy, cz, 0.2f, n, 1);
                } else {
                    mt[n] = mat_dielectric(1.5f);
                    sp[n] = sphere_new(cx, cy, cz, 0.2f, n, 2);
                }
                n++;
            }
        }
    }
    mt[n] = mat_dielectric(1.50f);
    sp[n] = sphere_new(0.0f, 1.0f, 0.0f, 1.0f, n, 2);
    n++;
    mt[n] = mat_lambertian(0.4f, 0.2f, 0.1f);
    sp[n] = sphere_new(-4.0f, 1.0f, 0.0f, 1.0f, n, 0);
    n++;
    mt[n] = mat_metal(0.7f, 0.6f, 0.5f, 0.0f);
    sp[n] = sphere_new(4.0f, 1.0f, 0.0f, 1.0f, n, 1);
    n++;
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * BVH builder: wc/bvh.rs
 * ---------------------------------------------------------------------------------------------- */
#define ORC_BINS 4096 /* bvh.rs:4 */

typedef struct { v3 mn, mx; uint32_t prim_count; } orc_bin;

/* Box growth. glam's Vec3::min / max (f32::min / max) may return either zero for (+0, -0), so the reference leaves
 * the sign of a zero bound open and a sequential fold would make it depend on the primitive order. Fixed here (and in
 * the product's host and device builders) order-independently: the minimum prefers -0, the maximum +0. Operands are
 * never NaN here. */
static inline float fmin_(float a, float b) { return a < b ? a : (b < a ? b : (signbit(a) ? a : b)); }
static inline float fmax_(float a, float b) { return a > b ? a : (b > a ? b : (signbit(a) ? b : a)); }
static void bin_default(orc_bin *b) { /* bvh.rs:12-20 */
    b->mn = v3_make(INFINITY, INFINITY, INFINITY);
    b->mx = v3_make(-INFINITY, -INFINITY, -INFINITY);
    b->prim_count = 0;
}
static void bin_expand(orc_bin *b, v3 mn, v3 mx) { /* bvh.rs:23-27 */
    b->mn = v3_make(fmin_(b->mn.x, mn.x), fmin_(b->mn.y, mn.y), fmin_(b->mn.z, mn.z));
    b->mx = v3_make(fmax_(b->mx.x, mx.x), fmax_(b->mx.y, mx.y), fmax_(b->mx.z, mx.z));
    b->prim_count += 1;
}
static float bin_area(const orc_bin *b) { /* bvh.rs:29-35 */
    if (!(isfinite(b->mx.x) && isfinite(b->mx.y) && isfinite(b->mx.z))) return 0.0f;
    v3 e = v3_sub(b->mx, b->mn);
    return (e.x * e.y + e.y * e.z) + e.z * e.x;
}
static void sphere_aabb(const orc_sphere *s, v3 *mn, v3 *mx) { /* sphere.rs:22-26 */
    *mn = v3_make(s->center[0] - s->radius, s->center[1] - s->radius, s->center[2] - s->radius);
    *mx = v3_make(s->center[0] + s->radius, s->center[1] + s->radius, s->center[2] + s->radius);
}
/* The builder of bvh.rs only needs a box, a binning key (the sphere's centre) and a swap per primitive.
 * Triangles (build extension, no reference code) plug in here: box of the three vertices, key = centroid
 * v0 + (e1 + e2) * (1/3). */
typedef struct { void *base; int kind; } prim_view; /* kind 0: orc_sphere, 1: orc_triangle */
static void prim_aabb(const prim_view *pv, uint32_t i, v3 *mn, v3 *mx) {
    if (pv->kind == 0) { sphere_aabb(&((const orc_sphere *)pv->base)[i], mn, mx); return; }
    const orc_triangle *t = &((const orc_triangle *)pv->base)[i];
    v3 p0 = v3_make(t->v0[0], t->v0[1], t->v0[2]);
    v3 p1 = v3_make(t->v0[0] + t->e1[0], t->v0[1] + t->e1[1], t->v0[2] + t->e1[2]);
    v3 p2 = v3_make(t->v0[0] + t->e2[0], t->v0[1] + t->e2[1], t->v0[2] + t->e2[2]);
    *mn = v3_make(fmin_(fmin_(p0.x, p1.x), p2.x), fmin_(fmin_(p0.y, p1.y), p2.y), fmin_(fmin_(p0.z, p1.z), p2.z));
    *mx = v3_make(fmax_(fmax_(p0.x, p1.x), p2.x), fmax_(fmax_(p0.y, p1.y), p2.y), fmax_(fmax_(p0.z, p1.z), p2.z));
}
static float prim_key(const prim_view *pv, uint32_t i, int axis) {
    if (pv->kind == 0) return ((const orc_sphere *)pv->base)[i].center[axis];
    const orc_triangle *t = &((const orc_triangle *)pv->base)[i];
    return t->v0[axis] + (t->e1[axis] + t->e2[axis]) * 0.33333334f;
}
static void prim_swap(const prim_view *pv, int64_t i, int64_t j) {
    if (pv->kind == 0) {
        orc_sphere *a = (orc_sphere *)pv->base, t = a[i]; a[i] = a[j]; a[j] = t;
    } else {
        orc_triangle *a = (orc_triangle *)pv->base, t = a[i]; a[i] = a[j]; a[j] = t;
    }
}
static void node_update_bounds(orc_bvh_node *nd, const prim_view *sp) { /* bvh.rs:58-70 */
    v3 mn = v3_make(INFINITY, INFINITY, INFINITY), mx = v3_make(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t i = 0; i < nd->prim_count; i++) {
        v3 a, b;
        prim_aabb(sp, nd->left_first + i, &a, &b);
        mn = v3_make(fmin_(mn.x, a.x), fmin_(mn.y, a.y), fmin_(mn.z, a.z));
        mx = v3_make(fmax_(mx.x, b.x), fmax_(mx.y, b.y), fmax_(mx.z, b.z));
    }
    nd->aabb_min[0] = mn.x; nd->aabb_min[1] = mn.y; nd->aabb_min[2] = mn.z;
    nd->aabb_max[0] = mx.x; nd->aabb_max[1] = mx.y; nd->aabb_max[2] = mx.z;
}
static float node_cost(const orc_bvh_node *nd) { /* bvh.rs:51-56 */
    float ex = nd->aabb_max[0] - nd->aabb_min[0], ey = nd->aabb_max[1] - nd->aabb_min[1],
          ez = nd->aabb_max[2] - nd->aabb_min[2];
    float area = (ex * ey + ey * ez) + ez * ex;
    return (float)nd->prim_count * area;
}

typedef struct {
    int n_bins; /* bvh.rs:4 fixes 4096; the mesh builder may use fewer */
    orc_bin *bins;
    uint32_t *left_count, *right_count;
    float *left_area, *right_area;
} split_scratch;

/* bvh.rs:73-139 */
static void find_best_split_plane(const orc_bvh_node *nd, const prim_view *sp, split_scratch *w,
                                  float *out_cost, int *out_axis, float *out_plane) {
    float extent[3] = {nd->aabb_max[0] - nd->aabb_min[0], nd->aabb_max[1] - nd->aabb_min[1],
                       nd->aabb_max[2] - nd->aabb_min[2]};
    uint32_t start = nd->left_first;
    const int nb = w->n_bins;
    float low_cost = INFINITY;
    int best_axis = 0;
    float best_plane = 0.0f;
    for (int axis = 0; axis < 3; axis++) {
        if (extent[axis] < 0.00001f) continue;
        for (int i = 0; i < nb; i++) bin_default(&w->bins[i]);
        float scale = (float)nb / extent[axis];
        float min_bound = nd->aabb_min[axis];
        for (uint32_t i = 0; i < nd->prim_count; i++) {
            float f = (prim_key(sp, i + start, axis) - min_bound) * scale;
            /* Rust `as usize` saturates: NaN/negative -> 0 */
            size_t bi = (f > 0.0f) ? ((f >= 4294967040.0f) ? (size_t)0xffffffffu : (size_t)f) : 0;
            if (bi > (size_t)(nb - 1)) bi = (size_t)(nb - 1);
            v3 a, b;
            prim_aabb(sp, i + start, &a, &b);
            bin_expand(&w->bins[bi], a, b);
        }
        orc_bin left_sum, right_sum;
        bin_default(&left_sum);
        bin_default(&right_sum);
        for (int idx = 0; idx < nb - 1; idx++) {
            left_sum.prim_count += w->bins[idx].prim_count;
            w->left_count[idx] = left_sum.prim_count;
            right_sum.prim_count += w->bins[nb - 1 - idx].prim_count;
            w->right_count[nb - 2 - idx] = right_sum.prim_count;
            bin_expand(&left_sum, w->bins[idx].mn, w->bins[idx].mx);
            left_sum.prim_count -= 1;
            w->left_area[idx] = bin_area(&left_sum);
            bin_expand(&right_sum, w->bins[nb - 1 - idx].mn, w->bins[nb - 1 - idx].mx);
            right_sum.prim_count -= 1;
            w->right_area[nb - 2 - idx] = bin_area(&right_sum);
        }
        float inv_bins = 1.0f / (float)nb;
        for (int idx = 0; idx < nb - 1; idx++) {
            float cost = (float)w->left_count[idx] * w->left_area[idx] +
                         (float)w->right_count[idx] * w->right_area[idx];
            if (cost < low_cost) {
                best_axis = axis;
                best_plane = min_bound + extent[axis] * inv_bins * (1.0f + (float)idx);
                low_cost = cost;
            }
        }
    }
    *out_cost = low_cost;
    *out_axis = best_axis;
    *out_plane = best_plane;
}

typedef struct { orc_bvh_node *nodes; uint32_t n_nodes; split_scratch *scratch; } bvh_builder;

/* bvh.rs:166-210 (signed indices: the reference's `j -= 1` on usize can underflow) */
static void subdivide(bvh_builder *bb, uint32_t index, const prim_view *sp) {
    float split_cost, plane;
    int axis;
    find_best_split_plane(&bb->nodes[index], sp, bb->scratch, &split_cost, &axis, &plane);
    float cost = node_cost(&bb->nodes[index]);
    if (cost <= split_cost) return;
    int64_t i = bb->nodes[index].left_first;
    int64_t j = i + (int64_t)bb->nodes[index].prim_count - 1;
    while (i <= j) {
        if (prim_key(sp, (uint32_t)i, axis) < plane) {
            i += 1;
        } else {
            prim_swap(sp, i, j);
            j -= 1;
        }
    }
    uint32_t left_count = (uint32_t)i - bb->nodes[index].left_first;
    if (left_count == 0 || left_count == bb->nodes[index].prim_count) return;
    uint32_t node_idx = bb->n_nodes;
    orc_bvh_node l = {{0, 0, 0}, bb->nodes[index].left_first, {0, 0, 0}, left_count};
    node_update_bounds(&l, sp);
    orc_bvh_node r = {{0, 0, 0}, (uint32_t)i, {0, 0, 0}, bb->nodes[index].prim_count - left_count};
    node_update_bounds(&r, sp);
    bb->nodes[index].left_first = node_idx;
    bb->nodes[index].prim_count = 0;
    bb->nodes[bb->n_nodes++] = l;
    bb->nodes[bb->n_nodes++] = r;
    subdivide(bb, node_idx, sp);
    subdivide(bb, node_idx + 1, sp);
}

/* bvh.rs:152-164 */
static uint32_t build_bvh_generic(const prim_view *pv, uint32_t n, orc_bvh_node *nodes, int n_bins) {
    bvh_builder bb;
    split_scratch sc;
    sc.n_bins = n_bins;
    sc.bins = (orc_bin *)malloc(sizeof(orc_bin) * n_bins);
    sc.left_count = (uint32_t *)malloc(sizeof(uint32_t) * n_bins);
    sc.right_count = (uint32_t *)malloc(sizeof(uint32_t) * n_bins);
    sc.left_area = (float *)malloc(sizeof(float) * n_bins);
    sc.right_area = (float *)malloc(sizeof(float) * n_bins);
    bb.nodes = nodes;
    bb.n_nodes = 0;
    bb.scratch = &sc;
    orc_bvh_node root = {{0, 0, 0}, 0, {0, 0, 0}, n};
    node_update_bounds(&root, pv);
    nodes[bb.n_nodes++] = root;
    orc_bvh_node pad = {{0, 0, 0}, 0, {0, 0, 0}, 0}; /* bvh.rs:160-161: index 1 is never used */
    nodes[bb.n_nodes++] = pad;
    subdivide(&bb, 0, pv);
    free(sc.bins); free(sc.left_count); free(sc.right_count); free(sc.left_area); free(sc.right_area);
    return bb.n_nodes;
}
uint32_t orc_build_bvh(orc_sphere *sp, uint32_t n, orc_bvh_node *nodes) {
    prim_view pv = {sp, 0};
    return build_bvh_generic(&pv, n, nodes, ORC_BINS);
}
/* Build extension: the same builder over triangles, with a caller-chosen bin count (4096 bins per axis per
 * node, bvh.rs:4, is O(10^10) work for a million primitives). */
uint32_t orc_build_bvh_triangles(orc_triangle *tris, uint32_t n, orc_bvh_node *nodes, uint32_t n_bins) {
    prim_view pv = {tris, 1};
    return build_bvh_generic(&pv, n, nodes, (int)(n_bins < 2 ? 2 : n_bins));
}

/* BASELINE config 5: n triangles, centres U[-10,10]^3, edge vectors U[-0.05,0.05]^3, material
 * type i % 3 over three shared materials: Lambertian 0.7 grey, Metal (0.8,0.8,0.8) fuzz 0.1, Dielectric 1.5.
 * Draw order per triangle: centre x,y,z, e1 x,y,z, e2 x,y,z; v0 = centre - (e1 + e2) * (1/3). */
uint32_t orc_scene_random_mesh(uint64_t seed, uint32_t n, orc_triangle *tris, orc_material *mt) {
    uint64_t rng[2];
    orc_scene_rng_seed(rng, seed);
    mt[0] = mat_lambertian(0.7f, 0.7f, 0.7f);
    mt[1] = mat_metal(0.8f, 0.8f, 0.8f, 0.1f);
    mt[2] = mat_dielectric(1.5f);
    for (uint32_t i = 0; i < n; i++) {
        float c[3], e1[3], e2[3];
        for (int k = 0; k < 3; k++) c[k] = scene_range(rng, -10.0f, 10.0f);
        for (int k = 0; k < 3; k++) e1[k] = scene_range(rng, -0.05f, 0.05f);
        for (int k = 0; k < 3; k++) e2[k] = scene_range(rng, -0.05f, 0.05f);
        orc_triangle t;
        for (int k = 0; k < 3; k++) {
            t.v0[k] = c[k] - (e1[k] + e2[k]) * 0.33333334f;
            t.e1[k] = e1[k];
            t.e2[k] = e2[k];
        }
        t.material_idx = i % 3u;
        t.material_type = i % 3u;
        t._pad = 0;
        tris[i] = t;
    }
    return 3;
}

/* ------------------------------------------------------------------------------------------------
 * camera / projection (host side; libm trig exactly as the Rust std calls would)
 * ---------------------------------------------------------------------------------------------- */
float orc_to_radians(float deg) { return deg * 0.017453292519943295769236907684886f; }

/* camera.rs:11-24; glam normalize = v * (1 / length) */
void orc_camera_new(const float from[3], const float at[3], float *pitch, float *yaw) {
    v3 f = v3_make(at[0] - from[0], at[1] - from[1], at[2] - from[2]);
    float rl = 1.0f / v3_length(f);
    f = v3_make(f.x * rl, f.y * rl, f.z * rl);
    *pitch = acosf(f.y);
    *yaw = atan2f(f.x, f.z);
}

static v3 glam_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

/* camera.rs:41-69 */
void orc_view_transform(const float pos[3], float pitch, float yaw, float view[16]) {
    float sp = sinf(pitch), cp = cosf(pitch), sy = sinf(yaw), cy = cosf(yaw);
    v3 dir = v3_make(sp * sy, cp, sp * cy);
    v3 right = glam_cross(dir, v3_make(0.0f, 1.0f, 0.0f));
    v3 up = glam_cross(right, dir);
    float m[16] = {right.x, right.y, right.z, 0.0f, up.x, up.y, up.z, 0.0f,
                   dir.x,   dir.y,   dir.z,   0.0f, pos[0], pos[1], pos[2], 1.0f};
    memcpy(view, m, sizeof m);
}

/* projection_matrix.rs:21-37 */
void orc_p_inv(float vfov_rad, float aspect, float zn, float zf, float out[16]) {
    float h = tanf(vfov_rad / 2.0f);
    float w = h * aspect;
    float r = zf / (zf - zn);
    float m[16] = {w, 0, 0, 0, 0, h, 0, 0, 0, 0, 0, -1.0f / (r * zn), 0, 0, 1.0f, 1.0f / zn};
    memcpy(out, m, sizeof m);
}

/* camera_controller.rs:173-185 */
void orc_gpu_camera_new(const float pos[3], float pitch, float yaw, float defocus_angle_rad,
                        float focus_distance, orc_gpu_camera *out) {
    out->position[0] = pos[0]; out->position[1] = pos[1]; out->position[2] = pos[2]; out->position[3] = 1.0f;
    out->pitch = pitch;
    out->yaw = yaw;
    out->defocus_radius = focus_distance * tanf(0.5f * defocus_angle_rad);
    out->focus_distance = focus_distance;
}

/* pt:282-289. The reference panics (unwrap on None) for x <= 64; the build returns (1,1) there. */
void orc_workgroup_size_64(uint32_t x, uint32_t *gx, uint32_t *gy) {
    uint32_t q = x / 64u + ((x % 64u) ? 1u : 0u); /* u32::div_ceil: no overflow near 2^32 */
    if (q <= 1) { *gx = 1; *gy = 1; return; }
    uint32_t y = (uint32_t)ceilf(sqrtf((float)q));
    uint32_t fac = 1;
    for (uint32_t z = y - 1; z >= 1; z--) {
        if (q % z == 0) { fac = z; break; }
    }
    if (q / fac >= (1u << 16)) { *gx = y; *gy = y; } else { *gx = fac; *gy = q / fac; }
}

/* ------------------------------------------------------------------------------------------------
 * probes
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_probe_jenkins(uint32_t x) { return orc_jenkins_hash(x); }
uint32_t orc_probe_init_rng(uint32_t px, uint32_t py, uint32_t rx, uint32_t fr) { return orc_init_rng(px, py, rx, fr); }
uint32_t orc_probe_next_int(uint32_t *s) { return orc_rng_next_int(s); }
float    orc_probe_next_float(uint32_t *s) { return orc_rng_next_float(s); }
uint32_t orc_probe_advance(uint32_t s, uint32_t n) { orc_advance(&s, n); return s; }
void orc_probe_sincos(const float *x, float *s, float *c, size_t n) {
    for (size_t i = 0; i < n; i++) orc_sincos(x[i], &s[i], &c[i]);
}
void orc_probe_pow(const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = orc_pow(x[i], y[i]);
}

/* ------------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------------- */
struct orc_ctx {
    orc_params p;
    uint32_t n_pixels;   /* local pixels (padded to whole 8-row bands when sharded) */
    uint32_t n_slots;    /* ray-queue capacity */
    uint32_t tiles_x, tiles_y_local;
    orc_sphere *spheres; uint32_t n_spheres;
    orc_triangle *triangles; uint32_t n_triangles; /* build extension: when set, the primitives are triangles */
    orc_material *materials; uint32_t n_materials;
    orc_bvh_node *nodes; uint32_t n_nodes;
    orc_gpu_camera camera;
    float inv_proj[16], view[16];
    orc_frame_buffer frame;
    uint32_t counters[16];
    orc_ray *rays, *ext_rays;
    orc_hit_payload *hits;
    uint32_t *misses;
    float *image, *accumulated;
    /* scratch for the two-phase (parallel trace, ordered append) extend */
    orc_hit_payload *trace_out; uint8_t *trace_flag;
    /* host-loop state (wc/parameters.rs:61-101) */
    uint32_t progress_frame, accumulated_samples;
    uint32_t table[64][4]; uint32_t table_rows;
    uint64_t totals[3];
    uint64_t stat_max_depth, stat_nodes, stat_tests, stat_rays;
};

static uint32_t local_pixel(const orc_ctx *c, uint32_t pixel_idx) {
    if (c->p.tile_world <= 1) return pixel_idx;
    uint32_t W = c->p.width;
    uint32_t y = pixel_idx / W, x = pixel_idx % W;
    uint32_t band = y / 8u;
    return ((band / c->p.tile_world) * 8u + (y % 8u)) * W + x;
}

orc_ctx *orc_create(const orc_params *p, const orc_sphere *sp, uint32_t ns, const orc_material *mt, uint32_t nm,
                    const orc_bvh_node *nd, uint32_t nn, const orc_gpu_camera *cam, const float ip[16],
                    const float view[16]) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->p = *p;
    if (c->p.tile_world == 0) { c->p.tile_world = 1; c->p.tile_rank = 0; }
    uint32_t W = p->width, H = p->height;
    c->tiles_x = (W + 7) / 8;
    uint32_t tiles_y = (H + 7) / 8;
    uint32_t rank = c->p.tile_rank, world = c->p.tile_world;
    c->tiles_y_local = (tiles_y > rank) ? (tiles_y - rank + world - 1) / world : 0;
    c->n_pixels = (world <= 1) ? W * H : c->tiles_y_local * 8u * W;
    c->n_slots = c->tiles_x * c->tiles_y_local * 64u;
    if (c->n_slots < c->n_pixels) c->n_slots = c->n_pixels;
    c->spheres = (orc_sphere *)malloc(sizeof(orc_sphere) * (ns ? ns : 1));
    memcpy(c->spheres, sp, sizeof(orc_sphere) * ns);
    c->n_spheres = ns;
    c->materials = (orc_material *)malloc(sizeof(orc_material) * (nm ? nm : 1));
    memcpy(c->materials, mt, sizeof(orc_material) * nm);
    c->n_materials = nm;
    c->nodes = (orc_bvh_node *)malloc(sizeof(orc_bvh_node) * (nn ? nn : 1));
    memcpy(c->nodes, nd, sizeof(orc_bvh_node) * nn);
    c->n_nodes = nn;
    c->camera = *cam;
    memcpy(c->inv_proj, ip, sizeof c->inv_proj);
    memcpy(c->view, view, sizeof c->view);
    size_t ns_ = c->n_slots ? c->n_slots : 1;
    c->rays = (orc_ray *)calloc(ns_, sizeof(orc_ray));
    c->ext_rays = (orc_ray *)calloc(ns_, sizeof(orc_ray));
    c->hits = (orc_hit_payload *)calloc(ns_, sizeof(orc_hit_payload));
    c->misses = (uint32_t *)calloc(ns_, sizeof(uint32_t));
    c->trace_out = (orc_hit_payload *)calloc(ns_, sizeof(orc_hit_payload));
    c->trace_flag = (uint8_t *)calloc(ns_, 1);
    size_t np_ = c->n_pixels ? c->n_pixels : 1;
    c->image = (float *)malloc(sizeof(float) * 3 * np_);
    c->accumulated = (float *)calloc(3 * np_, sizeof(float));
    for (size_t i = 0; i < 3 * (size_t)c->n_pixels; i++) c->image[i] = 1.0f; /* pt:53 */
    c->frame.width = W;
    c->frame.height = H;
    return c;
}

orc_ctx *orc_create_mesh(const orc_params *p, const orc_triangle *tris, uint32_t nt, const orc_material *mt, uint32_t nm,
                         const orc_bvh_node *nd, uint32_t nn, const orc_gpu_camera *cam, const float ip[16],
                         const float view[16]) {
    orc_sphere dummy = {{0, 0, 0, 1}, 0, 0, 0, 0};
    orc_ctx *c = orc_create(p, &dummy, 1, mt, nm, nd, nn, cam, ip, view);
    c->triangles = (orc_triangle *)malloc(sizeof(orc_triangle) * (nt ? nt : 1));
    memcpy(c->triangles, tris, sizeof(orc_triangle) * nt);
    c->n_triangles = nt;
    return c;
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    free(c->spheres); free(c->triangles); free(c->materials); free(c->nodes); free(c->rays); free(c->ext_rays); free(c->hits);
    free(c->misses); free(c->trace_out); free(c->trace_flag); free(c->image); free(c->accumulated);
    free(c);
}

void orc_set_frame(orc_ctx *c, const orc_frame_buffer *f) { c->frame = *f; }
void orc_set_counters(orc_ctx *c, const uint32_t v[16]) { memcpy(c->counters, v, sizeof c->counters); }
void orc_get_counters(const orc_ctx *c, uint32_t v[16]) { memcpy(v, c->counters, sizeof c->counters); }
void orc_reset_image(orc_ctx *c) { for (size_t i = 0; i < 3 * (size_t)c->n_pixels; i++) c->image[i] = 1.0f; }
void orc_reset_accumulated(orc_ctx *c) { memset(c->accumulated, 0, sizeof(float) * 3 * c->n_pixels); }
/* wgpu_state.rs:115-130: clear dst, copy src -> dst, clear src. Queues are count-guarded, so a swap is
 * equivalent; the clear of the source is kept so stale slots read as zero rays like in the reference. */
/* Test hook (the product's wfpt_write_rays): overwrite the first n entries of the ray queue with caller-made rays,
 * e.g. axis-parallel, zero-length or non-finite ones that generate_rays never produces. */
void orc_write_rays(orc_ctx *c, const orc_ray *rays, uint32_t n) {
    if (n > c->n_slots) n = c->n_slots;
    memcpy(c->rays, rays, sizeof(orc_ray) * n);
}

void orc_swap_ray_queues(orc_ctx *c) {
    orc_ray *t = c->rays;
    c->rays = c->ext_rays;
    c->ext_rays = t;
    memset(c->ext_rays, 0, sizeof(orc_ray) * c->n_slots);
}

uint32_t orc_n_pixels(const orc_ctx *c) { return c->n_pixels; }
const orc_ray *orc_rays(const orc_ctx *c) { return c->rays; }
const orc_ray *orc_extension_rays(const orc_ctx *c) { return c->ext_rays; }
const orc_hit_payload *orc_hits(const orc_ctx *c) { return c->hits; }
const uint32_t *orc_misses(const orc_ctx *c) { return c->misses; }
const float *orc_image(const orc_ctx *c) { return c->image; }
const float *orc_accumulated(const orc_ctx *c) { return c->accumulated; }
uint32_t orc_frame(const orc_ctx *c) { return c->progress_frame; }
uint32_t orc_accumulated_samples(const orc_ctx *c) { return c->accumulated_samples; }

/* ------------------------------------------------------------------------------------------------
 * generate_rays  (gr:42-91)
 * ---------------------------------------------------------------------------------------------- */
/* gr:107-116 */
static v3 rng_next_vec3in_unit_disk(uint32_t *state) {
    float r = orc_sqrt(orc_rng_next_float(state));
    float alpha = 2.0f * ORC_PI * orc_rng_next_float(state);
    float s, co;
    orc_sincos(alpha, &s, &co);
    return v3_make(r * co, r * s, 0.0f);
}

/* gx, gy: the dispatch. true_size == 0: literal reference semantics, width/height derived from the
 * dispatch (gr:55-56). true_size == 1 (build rule for sizes that are not multiples of 8):
 * width/height from the frame uniform, out-of-image lanes write an inactive ray. Identical results
 * whenever width and height are multiples of 8. With tile sharding, workgroup row wy is the rank's
 * wy-th band and pixel coordinates stay global. */
void orc_generate_rays(orc_ctx *c, uint32_t gx, uint32_t gy, int true_size) {
    uint32_t width = true_size ? c->frame.width : gx * 8u;
    uint32_t height = true_size ? c->frame.height : gy * 8u;
    uint32_t world = c->p.tile_world, rank = c->p.tile_rank;
    int64_t n_threads = (int64_t)gx * gy * 64;
#pragma omp parallel for schedule(static, 4096)
    for (int64_t tid = 0; tid < n_threads; tid++) {
        uint32_t workgroup_index = (uint32_t)(tid / 64), local_index = (uint32_t)(tid % 64);
        uint32_t wx = workgroup_index % gx, wy = workgroup_index / gx;
        uint32_t idx = workgroup_index * 64u + local_index; /* gr:48-51 */
        if (idx >= c->n_slots) continue;
        uint32_t idx_x = wx * 8u + local_index % 8u;
        uint32_t idx_y = (wy * world + rank) * 8u + local_index / 8u;
        orc_ray ray;
        memset(&ray, 0, sizeof ray);
        if (true_size && (idx_x >= width || idx_y >= height)) {
            ray.pixel_idx = ORC_INACTIVE_PIXEL;
            c->rays[idx] = ray;
            continue;
        }
        uint32_t pixel_idx = idx_x + idx_y * width; /* gr:57 */
        uint32_t rng_state = orc_init_rng(idx_x, idx_y, width, c->frame.frame); /* gr:60 */
        orc_advance(&rng_state, c->frame.sample_number * 10u);                  /* gr:61 */
        v3 offset = rng_next_vec3in_unit_disk(&rng_state);                      /* gr:63 */
        float ndc_x = ((float)idx_x + offset.x) / (float)width;                 /* gr:66 */
        float ndc_y = 1.0f - ((float)idx_y + offset.y) / (float)height;
        ndc_x = 2.0f * ndc_x - 1.0f; /* gr:67 */
        ndc_y = 2.0f * ndc_y - 1.0f;
        v4 ndc = {ndc_x, ndc_y, 1.0f, 1.0f};
        v4 pp = m4_mul(c->inv_proj, ndc); /* gr:68 */
        float pw = pp.w;
        pp.x = pp.x / pw; pp.y = pp.y / pw; pp.z = pp.z / pw; pp.w = pp.w / pw; /* gr:69 */
        v4 origin = {c->camera.position[0], c->camera.position[1], c->camera.position[2], c->camera.position[3]};
        if (c->camera.defocus_radius > 0.0f) { /* gr:73-82 */
            offset = rng_next_vec3in_unit_disk(&rng_state);
            float R = c->camera.defocus_radius;
            v4 p_lens = {R * offset.x, R * offset.y, R * offset.z, 1.0f};
            v4 lo = m4_mul(c->view, p_lens);
            float lw = lo.w;
            lo.x = lo.x / lw; lo.y = lo.y / lw; lo.z = lo.z / lw; lo.w = lo.w / lw;
            origin = lo;
            float tf = c->camera.focus_distance / pp.z;
            pp.x = tf * pp.x - p_lens.x; pp.y = tf * pp.y - p_lens.y;
            pp.z = tf * pp.z - p_lens.z; pp.w = tf * pp.w - p_lens.w;
        }
        v4 pd = {pp.x, pp.y, pp.z, 0.0f};
        v4 dir = v4_normalize(m4_mul(c->view, pd)); /* gr:84-86 */
        ray.origin[0] = origin.x; ray.origin[1] = origin.y; ray.origin[2] = origin.z; ray.origin[3] = origin.w;
        ray.direction[0] = dir.x; ray.direction[1] = dir.y; ray.direction[2] = dir.z; ray.direction[3] = dir.w;
        ray.inv_direction[0] = 1.0f / dir.x; ray.inv_direction[1] = 1.0f / dir.y; ray.inv_direction[2] = 1.0f / dir.z;
        ray.pixel_idx = pixel_idx;
        c->rays[idx] = ray; /* gr:90 */
    }
}

/* ------------------------------------------------------------------------------------------------
 * extend  (ex:47-210)
 * ---------------------------------------------------------------------------------------------- */
/* ex:185-210 */
static int hit_sphere(const orc_ctx *c, const orc_ray *ray, uint32_t sphere_idx, float t_min, float t_nearest,
                      orc_hit_payload *payload) {
    const orc_sphere *s = &c->spheres[sphere_idx];
    v4 d = {ray->direction[0], ray->direction[1], ray->direction[2], ray->direction[3]};
    v4 oc = {ray->origin[0] - s->center[0], ray->origin[1] - s->center[1], ray->origin[2] - s->center[2],
             ray->origin[3] - s->center[3]};
    float a = v4_dot(d, d);
    float b = v4_dot(d, oc);
    float cc = v4_dot(oc, oc) - s->radius * s->radius;
    float discrim = b * b - a * cc;
    if (discrim >= 0.0f) {
        float t = (-b - orc_sqrt(discrim)) / a;
        if (t > t_min && t < t_nearest) {
            payload->t = t; payload->ray_idx = 0; payload->sphere_idx = sphere_idx; payload->mat_type = s->material_type;
            return 1;
        }
        t = (-b + orc_sqrt(discrim)) / a;
        if (t > t_min && t < t_nearest) {
            payload->t = t; payload->ray_idx = 0; payload->sphere_idx = sphere_idx; payload->mat_type = s->material_type;
            return 1;
        }
    }
    return 0;
}

/* Build extension (no reference code): Moeller-Trumbore with the same (t_min, t_nearest) window as ex:185-210.
 * Fixed evaluation order, no fused operations; a degenerate triangle (det == 0) yields inf/NaN barycentrics
 * and is rejected by the negated range tests. payload.sphere_idx carries the triangle index. */
static v3 cross3(v3 a, v3 b) { return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static int hit_triangle(const orc_ctx *c, const orc_ray *ray, uint32_t tri_idx, float t_min, float t_nearest,
                        orc_hit_payload *payload) {
    const orc_triangle *tr = &c->triangles[tri_idx];
    v3 d = v3_make(ray->direction[0], ray->direction[1], ray->direction[2]);
    v3 e1 = v3_make(tr->e1[0], tr->e1[1], tr->e1[2]), e2 = v3_make(tr->e2[0], tr->e2[1], tr->e2[2]);
    v3 pvec = cross3(d, e2);
    float det = v3_dot(e1, pvec);
    float inv_det = 1.0f / det;
    v3 tvec = v3_make(ray->origin[0] - tr->v0[0], ray->origin[1] - tr->v0[1], ray->origin[2] - tr->v0[2]);
    float u = v3_dot(tvec, pvec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    v3 qvec = cross3(tvec, e1);
    float v = v3_dot(d, qvec) * inv_det;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    float t = v3_dot(e2, qvec) * inv_det;
    if (t > t_min && t < t_nearest) {
        payload->t = t; payload->ray_idx = 0; payload->sphere_idx = tri_idx; payload->mat_type = tr->material_type;
        return 1;
    }
    return 0;
}
static int hit_prim(const orc_ctx *c, const orc_ray *ray, uint32_t idx, float t_min, float t_nearest, orc_hit_payload *p) {
    return c->triangles ? hit_triangle(c, ray, idx, t_min, t_nearest, p) : hit_sphere(c, ray, idx, t_min, t_nearest, p);
}

/* ex:164-183 */
static float hit_bvh_node(const orc_bvh_node *node, const orc_ray *ray, float nearest_hit) {
    float t_x_min = (node->aabb_min[0] - ray->origin[0]) * ray->inv_direction[0];
    float t_x_max = (node->aabb_max[0] - ray->origin[0]) * ray->inv_direction[0];
    float tmin = orc_min(t_x_min, t_x_max);
    float tmax = orc_max(t_x_min, t_x_max);
    float t_y_min = (node->aabb_min[1] - ray->origin[1]) * ray->inv_direction[1];
    float t_y_max = (node->aabb_max[1] - ray->origin[1]) * ray->inv_direction[1];
    tmin = orc_max(orc_min(t_y_min, t_y_max), tmin);
    tmax = orc_min(orc_max(t_y_min, t_y_max), tmax);
    float t_z_min = (node->aabb_min[2] - ray->origin[2]) * ray->inv_direction[2];
    float t_z_max = (node->aabb_max[2] - ray->origin[2]) * ray->inv_direction[2];
    tmin = orc_max(orc_min(t_z_min, t_z_max), tmin);
    tmax = orc_min(orc_max(t_z_min, t_z_max), tmax);
    if (tmin > tmax || tmax <= 0.0f || tmin > nearest_hit) return 1e30f;
    return tmin;
}

typedef struct { uint64_t max_depth, nodes, tests; } trace_stat;

/* ex:72-162, USE_BVH branch. The reference's stack holds 10 whole nodes with no overflow check (ex:38,
 * 135-136: undefined behaviour beyond depth 10); the oracle's stack holds ORC_MAX_STACK (128) and aborts beyond that.
 * Storing node copies or node indices is equivalent. */
static int trace_ray_bvh(const orc_ctx *c, const orc_ray *ray, orc_hit_payload *hit, trace_stat *st) {
    float nearest_hit = 1e30f;
    orc_hit_payload temp;
    memset(&temp, 0, sizeof temp);
    orc_bvh_node stack[ORC_MAX_STACK];
    uint32_t sp = 0;
    orc_bvh_node node = c->nodes[0]; /* the root's box is never tested */
    for (;;) {
        st->nodes++;
        if (node.prim_count > 0) {
            for (uint32_t i = 0; i < node.prim_count; i++) {
                orc_hit_payload nh;
                st->tests++;
                if (hit_prim(c, ray, node.left_first + i, 0.001f, nearest_hit, &nh)) {
                    nearest_hit = nh.t;
                    temp = nh;
                }
            }
            if (sp == 0) break;
            sp--;
            node = stack[sp];
            continue;
        } else {
            orc_bvh_node left = c->nodes[node.left_first];
            orc_bvh_node right = c->nodes[node.left_first + 1];
            float t_left = hit_bvh_node(&left, ray, nearest_hit);
            float t_right = hit_bvh_node(&right, ray, nearest_hit);
            if (t_left > t_right) { /* strict: ties keep the left child first */
                float tt = t_left; t_left = t_right; t_right = tt;
                orc_bvh_node tn = left; left = right; right = tn;
            }
            if (t_left > nearest_hit) {
                if (sp == 0) break;
                sp--;
                node = stack[sp];
            } else {
                node = left;
                if (t_right < nearest_hit) {
                    if (sp >= ORC_MAX_STACK) { fprintf(stderr, "oracle: BVH stack overflow\n"); abort(); }
                    stack[sp++] = right;
                    if (sp > st->max_depth) st->max_depth = sp;
                }
            }
        }
    }
    if (nearest_hit < 1e30f) { *hit = temp; return 1; }
    return 0;
}

/* ex:141-153: the USE_BVH == false branch */
int orc_trace_brute(const orc_ctx *c, const orc_ray *ray, orc_hit_payload *out) {
    float nearest_hit = 1e30f;
    orc_hit_payload temp;
    memset(&temp, 0, sizeof temp);
    const uint32_t n_prims = c->triangles ? c->n_triangles : c->n_spheres;
    for (uint32_t i = 0; i < n_prims; i++) {
        orc_hit_payload nh;
        if (hit_prim(c, ray, i, 0.001f, nearest_hit, &nh)) { nearest_hit = nh.t; temp = nh; }
    }
    if (nearest_hit < 1e30f) { *out = temp; return 1; }
    return 0;
}

int orc_trace_bvh(orc_ctx *c, const orc_ray *ray, orc_hit_payload *out) {
    trace_stat st = {0, 0, 0};
    return trace_ray_bvh(c, ray, out, &st);
}

/* Diagnostics for kernel design (not part of the chain): what the reference's traversal does on the current ray
 * queue, summed over n rays. out[0] inner visits, [1] of them with both children entered (a push), [2] one child,
 * [3] none (pop from an inner node), [4] leaf visits, [5] pops that found the stack empty (ray done),
 * [6] sum over pops of the levels climbed from the current node to the popped node's parent,
 * [8 + k] pops taken with k entries on the stack (k = 0..7, deeper ones in [15]). */
void orc_traversal_profile(orc_ctx *c, uint32_t n, uint64_t out[16]) {
    uint64_t acc[16] = {0};
#pragma omp parallel
    {
        uint64_t loc[16] = {0};
#pragma omp for schedule(dynamic, 1024)
        for (int64_t idx = 0; idx < (int64_t)n; idx++) {
            const orc_ray *ray = &c->rays[idx];
            if (ray->pixel_idx == ORC_INACTIVE_PIXEL) continue;
            float nearest_hit = 1e30f;
            orc_bvh_node stack[ORC_MAX_STACK];
            uint32_t depth_of[ORC_MAX_STACK];
            uint32_t sp = 0, depth = 0;
            orc_bvh_node node = c->nodes[0];
            for (;;) {
                int pop = 0;
                if (node.prim_count > 0) {
                    loc[4]++;
                    for (uint32_t i = 0; i < node.prim_count; i++) {
                        orc_hit_payload nh;
                        if (hit_prim(c, ray, node.left_first + i, 0.001f, nearest_hit, &nh)) nearest_hit = nh.t;
                    }
                    pop = 1;
                } else {
                    loc[0]++;
                    orc_bvh_node left = c->nodes[node.left_first], right = c->nodes[node.left_first + 1];
                    float t_left = hit_bvh_node(&left, ray, nearest_hit), t_right = hit_bvh_node(&right, ray, nearest_hit);
                    if (t_left > t_right) {
                        float tt = t_left; t_left = t_right; t_right = tt;
                        orc_bvh_node tn = left; left = right; right = tn;
                    }
                    if (t_left > nearest_hit) { loc[3]++; pop = 1; }
                    else {
                        node = left;
                        depth++;
                        if (t_right < nearest_hit) { loc[1]++; depth_of[sp] = depth; stack[sp++] = right; }
                        else loc[2]++;
                    }
                }
                if (pop) {
                    loc[8 + (sp < 7 ? sp : 7)]++;
                    if (sp == 0) { loc[5]++; break; }
                    sp--;
                    loc[6] += depth - depth_of[sp] + 1; /* levels from the current node up to the popped node's parent */
                    node = stack[sp];
                    depth = depth_of[sp];
                }
            }
        }
#pragma omp critical
        for (int k = 0; k < 16; k++) acc[k] += loc[k];
    }
    for (int k = 0; k < 16; k++) out[k] = acc[k];
}

/* Diagnostics for kernel design: per ray, the inner visits between consecutive leaf visits ("rounds" of a
 * while-while schedule): segs[16 * idx + k] = inner visits before the k-th leaf (k < 15; the visits after the last
 * leaf are added to the last used slot + 1), n_leaves[idx] = leaf visits. */
void orc_ray_rounds(orc_ctx *c, uint32_t n, uint8_t *segs, uint8_t *n_leaves) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        const orc_ray *ray = &c->rays[idx];
        uint8_t *sg = segs + 16 * idx;
        memset(sg, 0, 16);
        n_leaves[idx] = 0;
        if (ray->pixel_idx == ORC_INACTIVE_PIXEL) continue;
        float nearest_hit = 1e30f;
        orc_bvh_node stack[ORC_MAX_STACK];
        uint32_t sp = 0, k = 0;
        orc_bvh_node node = c->nodes[0];
        for (;;) {
            int pop = 0;
            if (node.prim_count > 0) {
                for (uint32_t i = 0; i < node.prim_count; i++) {
                    orc_hit_payload nh;
                    if (hit_prim(c, ray, node.left_first + i, 0.001f, nearest_hit, &nh)) nearest_hit = nh.t;
                }
                if (n_leaves[idx] < 255) n_leaves[idx]++;
                if (k < 15) k++;
                pop = 1;
            } else {
                if (sg[k] < 255) sg[k]++;
                orc_bvh_node left = c->nodes[node.left_first], right = c->nodes[node.left_first + 1];
                float t_left = hit_bvh_node(&left, ray, nearest_hit), t_right = hit_bvh_node(&right, ray, nearest_hit);
                if (t_left > t_right) {
                    float tt = t_left; t_left = t_right; t_right = tt;
                    orc_bvh_node tn = left; left = right; right = tn;
                }
                if (t_left > nearest_hit) pop = 1;
                else { node = left; if (t_right < nearest_hit) stack[sp++] = right; }
            }
            if (pop) {
                if (sp == 0) break;
                node = stack[--sp];
            }
        }
    }
}

/* Diagnostics for kernel design: t of primitive `prim` for each ray of the current queue (1e30 = not hit), and
 * orc_ray_rounds for a traversal that starts with `nearest` already set per ray (a primitive tested ahead of the tree) and
 * optionally tests the root's own box first (the reference never does, ex:84). */
void orc_prim_hit_t(orc_ctx *c, uint32_t n, uint32_t prim, float *t_out) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        const orc_ray *ray = &c->rays[idx];
        orc_hit_payload nh;
        t_out[idx] = 1e30f;
        if (ray->pixel_idx == ORC_INACTIVE_PIXEL) continue;
        if (hit_prim(c, ray, prim, 0.001f, 1e30f, &nh)) t_out[idx] = nh.t;
    }
}
void orc_ray_rounds_init(orc_ctx *c, uint32_t n, const float *init_nearest, int test_root, uint8_t *segs, uint8_t *n_leaves) {
    const int no_blind_descent = test_root & 2; /* bit 1: a pair missed by the ray is never entered (the reference enters it while nothing is hit yet, ex:126) */
    test_root &= 1;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        const orc_ray *ray = &c->rays[idx];
        uint8_t *sg = segs + 16 * idx;
        memset(sg, 0, 16);
        n_leaves[idx] = 0;
        if (ray->pixel_idx == ORC_INACTIVE_PIXEL) continue;
        float nearest_hit = init_nearest[idx];
        orc_bvh_node stack[ORC_MAX_STACK];
        uint32_t sp = 0, k = 0;
        orc_bvh_node node = c->nodes[0];
        if (test_root && hit_bvh_node(&node, ray, nearest_hit) >= 1e30f) continue;
        for (;;) {
            int pop = 0;
            if (node.prim_count > 0) {
                for (uint32_t i = 0; i < node.prim_count; i++) {
                    orc_hit_payload nh;
                    if (hit_prim(c, ray, node.left_first + i, 0.001f, nearest_hit, &nh)) nearest_hit = nh.t;
                }
                if (n_leaves[idx] < 255) n_leaves[idx]++;
                if (k < 15) k++;
                pop = 1;
            } else {
                if (sg[k] < 255) sg[k]++;
                orc_bvh_node left = c->nodes[node.left_first], right = c->nodes[node.left_first + 1];
                float t_left = hit_bvh_node(&left, ray, nearest_hit), t_right = hit_bvh_node(&right, ray, nearest_hit);
                if (t_left > t_right) {
                    float tt = t_left; t_left = t_right; t_right = tt;
                    orc_bvh_node tn = left; left = right; right = tn;
                }
                if (t_left > nearest_hit || (no_blind_descent && t_left >= 1e30f)) pop = 1;
                else { node = left; if (t_right < nearest_hit) stack[sp++] = right; }
            }
            if (pop) {
                if (sp == 0) break;
                node = stack[--sp];
            }
        }
    }
}

/* Diagnostics for kernel design: cost model of a wave64 "while-while" schedule with up to Q postponed leaves per lane
 * (Q = 0: a lane that reaches a leaf waits for the leaf phase, the schedule extend_kernel uses). A lane with a free
 * slot notes the leaf, pops and keeps traversing with its `nearest` not yet updated (speculative: visits more nodes,
 * results unchanged); the leaf phase runs when no lane can make an inner step and tests one noted leaf per lane and
 * pass; Q's bits 8.. optionally hold a threshold T: the inner phase also ends once T lanes wait at a leaf.
 * out[0] = inner-loop iterations summed over waves, [1] = leaf passes, [2] = inner visits (lane-steps),
 * [3] = leaf tests, [4] = waves. */
typedef struct { orc_bvh_node node; orc_bvh_node stack[ORC_MAX_STACK]; uint32_t sp; float nearest; int done; int at_leaf;
                 uint32_t q[8]; uint32_t nq; } sim_lane;
static void sim_pop(sim_lane *l) { if (l->sp == 0) l->done = 1; else l->node = l->stack[--l->sp]; }
void orc_sim_postpone(orc_ctx *c, uint32_t n, uint32_t Q, uint64_t out[8]) {
    uint64_t iters = 0, passes = 0, visits = 0, tests = 0, waves = 0;
    const uint32_t T = Q >> 8; /* optional: the inner phase also ends once T lanes wait at a leaf (0 = never) */
    Q &= 0xffu;
    if (Q > 8) Q = 8;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : iters, passes, visits, tests, waves)
    for (int64_t w0 = 0; w0 < (int64_t)n; w0 += 64) {
        sim_lane *L = (sim_lane *)malloc(sizeof(sim_lane) * 64);
        int nl = (int)((n - w0) < 64 ? (n - w0) : 64);
        for (int i = 0; i < nl; i++) {
            L[i].node = c->nodes[0]; L[i].sp = 0; L[i].nearest = 1e30f; L[i].nq = 0; L[i].at_leaf = 0;
            L[i].done = c->rays[w0 + i].pixel_idx == ORC_INACTIVE_PIXEL;
        }
        waves++;
        for (;;) {
            /* inner phase */
            for (;;) {
                int any = 0;
                for (int i = 0; i < nl; i++) {
                    sim_lane *l = &L[i];
                    if (l->done || l->at_leaf) continue;
                    const orc_ray *ray = &c->rays[w0 + i];
                    if (l->node.prim_count > 0) { /* arrived at a leaf by a pop */
                        if (l->nq < Q) { l->q[l->nq++] = l->node.left_first | (l->node.prim_count << 24); sim_pop(l); if (l->done && l->nq) { l->done = 0; l->at_leaf = 2; } }
                        else l->at_leaf = 1;
                        continue;
                    }
                    any = 1;
                    visits++;
                    orc_bvh_node left = c->nodes[l->node.left_first], right = c->nodes[l->node.left_first + 1];
                    float t_left = hit_bvh_node(&left, ray, l->nearest), t_right = hit_bvh_node(&right, ray, l->nearest);
                    if (t_left > t_right) { float tt = t_left; t_left = t_right; t_right = tt; orc_bvh_node tn = left; left = right; right = tn; }
                    if (t_left > l->nearest) { sim_pop(l); if (l->done && l->nq) { l->done = 0; l->at_leaf = 2; } }
                    else { l->node = left; if (t_right < l->nearest) l->stack[l->sp++] = right; }
                    /* a leaf reached by descent is handled at the top of the next iteration (costs no inner step) */
                    if (!l->done && !l->at_leaf && l->node.prim_count > 0) {
                        if (l->nq < Q) { l->q[l->nq++] = l->node.left_first | (l->node.prim_count << 24); sim_pop(l); if (l->done && l->nq) { l->done = 0; l->at_leaf = 2; } }
                        else l->at_leaf = 1;
                    }
                }
                if (!any) break;
                iters++;
                if (T) {
                    uint32_t waiting = 0;
                    for (int i = 0; i < nl; i++) waiting += (!L[i].done && L[i].at_leaf == 1);
                    if (waiting >= T) break;
                }
            }
            /* leaf phase: every lane tests its noted leaves (one per pass) and the leaf it waits at */
            int alive = 0;
            uint32_t max_pass = 0;
            for (int i = 0; i < nl; i++) {
                sim_lane *l = &L[i];
                if (l->done) continue;
                const orc_ray *ray = &c->rays[w0 + i];
                uint32_t np = 0;
                for (uint32_t k = 0; k < l->nq; k++) {
                    uint32_t first = l->q[k] & 0xffffffu, cnt = l->q[k] >> 24;
                    for (uint32_t j = 0; j < cnt; j++) { orc_hit_payload nh; tests++; if (hit_prim(c, ray, first + j, 0.001f, l->nearest, &nh)) l->nearest = nh.t; }
                    np++;
                }
                l->nq = 0;
                if (l->at_leaf == 1) {
                    for (uint32_t j = 0; j < l->node.prim_count; j++) { orc_hit_payload nh; tests++; if (hit_prim(c, ray, l->node.left_first + j, 0.001f, l->nearest, &nh)) l->nearest = nh.t; }
                    np++;
                    l->at_leaf = 0;
                    sim_pop(l);
                } else if (l->at_leaf == 2) { l->at_leaf = 0; l->done = 1; }
                if (np > max_pass) max_pass = np;
                if (!l->done) alive = 1;
            }
            passes += max_pass;
            if (!alive) break;
        }
        free(L);
    }
    out[0] = iters; out[1] = passes; out[2] = visits; out[3] = tests; out[4] = waves;
}

/* Diagnostics for kernel design (not part of the chain): per-ray traversal step counts of the current ray
 * queue, so lane-utilisation models of the GPU schedule can be evaluated offline. */
/* ---- MODEL of the device's default traversal (wfpt_kernels.hip: trace_ray_conservative; wfpt_api.hip: build_nodes_ch), kept here
 * so that the hunt for rays on which a conservative box test could change the reported hit runs on the CPU, over millions of
 * rays, without a GPU. Not part of the oracle proper: the reference's traversal is trace_ray_bvh above. Same arithmetic as the
 * device (fmaf = one fused multiply-add, -ffp-contract=off elsewhere): node boxes as centre / half-extent, the half-extent grown
 * by 2^-17 * extent per axis (extent[0..2]; extent[3..5] = centre and extent[6] = squared radius of the ball of origins the
 * free walk is used for, wfpt_api.hip: safe_region); per axis tc = fma(c, b, -(o b)), entry = fma(h, -|b|, tc), exit = fma(h, |b|, tc);
 * entered <=> max(entry, 0) <= min(exit, nearest). */
#include <math.h>
static float model_up(double v) { float f = (float)v; if ((double)f < v) f = nextafterf(f, INFINITY); return f; }
static void model_box(const orc_bvh_node *nd, const float extent[7], float c3[3], float h3[3]) {
    for (int ax = 0; ax < 3; ax++) {
        double lo = nd->aabb_min[ax], hi = nd->aabb_max[ax];
        float c = (float)(0.5 * (lo + hi));
        double h = fmax((double)c - lo, hi - (double)c);
        c3[ax] = c;
        h3[ax] = model_up((double)model_up(h) + ldexp((double)extent[ax], -17));
    }
}
static int orc_leaf_rejected(const orc_bvh_node *node, const orc_ray *ray, float nearest_hit) { /* ex:165-179 as a boolean */
    float tmin = -INFINITY, tmax = INFINITY;
    for (int ax = 0; ax < 3; ax++) {
        float t1 = (node->aabb_min[ax] - ray->origin[ax]) * ray->inv_direction[ax];
        float t2 = (node->aabb_max[ax] - ray->origin[ax]) * ray->inv_direction[ax];
        if (ax == 0) { tmin = orc_min(t1, t2); tmax = orc_max(t1, t2); }
        else { tmin = orc_max(orc_min(t1, t2), tmin); tmax = orc_min(orc_max(t1, t2), tmax); }
    }
    return tmin > tmax || tmax <= 0.0f || tmin > nearest_hit;
}
typedef struct { float b[3], no[3], ab[3]; } model_ray;
static uint64_t model_handovers = 0;
static int model_enter(const float c3[3], const float h3[3], const model_ray *r, float nearest, float *t_in) {
    float in = -INFINITY, out = INFINITY;
    for (int ax = 0; ax < 3; ax++) {
        float tc = fmaf(c3[ax], r->b[ax], r->no[ax]);
        in = orc_max(in, fmaf(h3[ax], -r->ab[ax], tc));
        out = orc_min(out, fmaf(h3[ax], r->ab[ax], tc));
    }
    *t_in = in;
    return orc_max(in, 0.0f) <= orc_min(out, nearest);
}
static int model_near_tie(float t, float nearest) { return fabsf(t - nearest) <= nearest * 3.8146973e-6f; }
/* the candidate distances hit_prim compares with the (0.001, nearest) window, for the model's near-tie watch (spheres only) */
static int model_sphere_risk(const orc_ctx *c, const orc_ray *ray, uint32_t idx, float nearest) {
    const orc_sphere *s = &c->spheres[idx];
    float ocx = ray->origin[0] - s->center[0], ocy = ray->origin[1] - s->center[1], ocz = ray->origin[2] - s->center[2];
    float dx = ray->direction[0], dy = ray->direction[1], dz = ray->direction[2];
    float a = (dx * dx + dy * dy) + dz * dz;
    float b = (dx * ocx + dy * ocy) + dz * ocz;
    float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - s->radius * s->radius;
    float disc = b * b - a * cc;
    if (!(disc >= 0.0f)) return 0;
    float sq = sqrtf(disc);
    float t = (-b - sq) / a;
    if (t > 0.001f && model_near_tie(t, nearest)) return 1;
    if (t > 0.001f && t < nearest) return 0;
    t = (-b + sq) / a;
    return t > 0.001f && model_near_tie(t, nearest);
}
static int trace_ray_model(const orc_ctx *c, const orc_ray *ray, const float extent[7], int leaf_exact, orc_hit_payload *hit) {
    model_ray r;
    for (int ax = 0; ax < 3; ax++) {
        float inv = 1.0f / ray->direction[ax];
        r.b[ax] = orc_min(orc_max(inv, -1e30f), 1e30f);
        r.no[ax] = -(ray->origin[ax] * r.b[ax]);
        r.ab[ax] = fabsf(r.b[ax]);
    }
    float nearest = 1e30f;
    orc_hit_payload temp;
    memset(&temp, 0, sizeof temp);
    uint32_t stack[ORC_MAX_STACK], sp = 0, node = 0, best_leaf = 0;
    int risk = 0; /* leaf_exact >= 2: near-ties, probes and far origins hand the ray to the reference's walk, as the device does */
    if (leaf_exact >= 2) {
        float fx = ray->origin[0] - extent[3], fy = ray->origin[1] - extent[4], fz = ray->origin[2] - extent[5];
        if ((fx * fx + fy * fy) + fz * fz > extent[6]) { trace_stat st0 = {0, 0, 0}; return trace_ray_bvh(c, ray, hit, &st0); }
    }
    for (;;) {
        const orc_bvh_node *nd = &c->nodes[node];
        if (nd->prim_count > 0) {
            /* the leaf's own box with the reference's arithmetic (the device recomputes it from the primitives; wfpt_create
             * checks that this equals the node's box); the root's box is never tested (ex:84). Mode 2 = visit_leaf of the device:
             * the primitive tests run into a tentative result whatever the box says (with the near-tie watch); the box then
             * decides between keeping it and, if something changed, handing the ray over. */
            int enter = node == 0 || !leaf_exact || !orc_leaf_rejected(nd, ray, nearest);
            if (leaf_exact == 3) {
                /* Mode 3 = the device's walk since round 4 (visit_leaf + leaf_box_verdict): candidates are accepted as they come
                 * (with the near-tie watch) and ONE box is tested, after the walk: the leaf of the final hit, against the hit's own
                 * distance. See the argument at leaf_box_verdict in wfpt_kernels.hip. */
                for (uint32_t i = 0; i < nd->prim_count; i++) {
                    orc_hit_payload nh;
                    if (!c->triangles && model_sphere_risk(c, ray, nd->left_first + i, nearest)) risk = 1;
                    if (hit_prim(c, ray, nd->left_first + i, 0.001f, nearest, &nh)) { nearest = nh.t; temp = nh; best_leaf = node; }
                }
            } else if (leaf_exact == 2) {
                float n2 = nearest;
                orc_hit_payload t2 = temp;
                int tie = 0;
                for (uint32_t i = 0; i < nd->prim_count; i++) {
                    orc_hit_payload nh;
                    if (!c->triangles && model_sphere_risk(c, ray, nd->left_first + i, n2)) tie = 1;
                    if (hit_prim(c, ray, nd->left_first + i, 0.001f, n2, &nh)) { n2 = nh.t; t2 = nh; }
                }
                if (tie || (!enter && n2 < nearest)) risk = 1;
                if (enter) { nearest = n2; temp = t2; }
            } else if (enter) {
                for (uint32_t i = 0; i < nd->prim_count; i++) {
                    orc_hit_payload nh;
                    if (hit_prim(c, ray, nd->left_first + i, 0.001f, nearest, &nh)) { nearest = nh.t; temp = nh; }
                }
            }
            if (sp == 0) break;
            node = stack[--sp];
            continue;
        }
        float cl[3], hl[3], cr[3], hr[3], l_in, r_in;
        model_box(&c->nodes[nd->left_first], extent, cl, hl);
        model_box(&c->nodes[nd->left_first + 1], extent, cr, hr);
        int hit_l = model_enter(cl, hl, &r, nearest, &l_in), hit_r = model_enter(cr, hr, &r, nearest, &r_in);
        if (!(hit_l || hit_r)) {
            if (sp == 0) break;
            node = stack[--sp];
        } else {
            int go_right = hit_r && (!hit_l || l_in > r_in);
            if (hit_l && hit_r) {
                if (sp >= ORC_MAX_STACK) { fprintf(stderr, "oracle: model stack overflow\n"); abort(); }
                stack[sp++] = nd->left_first + (go_right ? 0u : 1u);
            }
            node = nd->left_first + (go_right ? 1u : 0u);
        }
    }
    if (leaf_exact == 3 && !risk && nearest < 1e30f && best_leaf != 0 && orc_leaf_rejected(&c->nodes[best_leaf], ray, nearest)) risk = 1;
    if (risk) {
        trace_stat st = {0, 0, 0};
#pragma omp atomic
        model_handovers++;
        return trace_ray_bvh(c, ray, hit, &st);
    }
    if (nearest < 1e30f) { *hit = temp; return 1; }
    return 0;
}
/* test probe: normalize of n 3-vectors (xyz interleaved) */
void orc_probe_normalize3(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        v3 r = v3_normalize(v3_make(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
/* rays the model handed to the reference's walk since the process started (near-tie, failed leaf box of the hit); far origins not counted */
uint64_t orc_model_handovers(void) { return model_handovers; }
/* Traces rays 0..n-1 of the ray queue with the reference's traversal and with the model; writes up to max_out records
 * (ray index, kind) of rays whose result differs (leaf_exact = 2: the device's walk; 1: without the near-tie hand-over; 0: also WITHOUT the exact test of leaf boxes, i.e. every
 * box merely conservative -- the variant that is NOT equivalent to the reference, kept to show the counter-example): kind 1 = the model reports a hit the reference does not, 2 = the reference
 * reports a hit the model does not, 3 = both hit, different t or primitive. Returns the number of differing rays. */
uint32_t orc_model_mismatches(orc_ctx *c, uint32_t n, const float extent[7], int leaf_exact, uint32_t *out, uint32_t max_out) {
    uint32_t count = 0;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        const orc_ray *ray = &c->rays[idx];
        if (ray->pixel_idx == ORC_INACTIVE_PIXEL) continue;
        trace_stat st = {0, 0, 0};
        orc_hit_payload a, b;
        memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
        int ha = trace_ray_bvh(c, ray, &a, &st), hb = trace_ray_model(c, ray, extent, leaf_exact, &b);
        int kind = 0;
        if (hb && !ha) kind = 1;
        else if (ha && !hb) kind = 2;
        else if (ha && hb && (memcmp(&a.t, &b.t, 4) != 0 || a.sphere_idx != b.sphere_idx)) kind = 3;
        if (kind) {
            uint32_t k;
#pragma omp atomic capture
            k = count++;
            if (k < max_out) { out[2 * k] = (uint32_t)idx; out[2 * k + 1] = (uint32_t)kind; }
        }
    }
    return count;
}

void orc_ray_steps(orc_ctx *c, uint32_t n, uint16_t *inner_steps, uint16_t *leaf_steps) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        trace_stat st = {0, 0, 0};
        orc_hit_payload payload;
        if (c->rays[idx].pixel_idx == ORC_INACTIVE_PIXEL) { inner_steps[idx] = 0; leaf_steps[idx] = 0; continue; }
        trace_ray_bvh(c, &c->rays[idx], &payload, &st);
        /* every leaf of the seeded scenes holds one sphere, so sphere tests == leaf visits */
        leaf_steps[idx] = (uint16_t)st.tests;
        inner_steps[idx] = (uint16_t)(st.nodes - st.tests);
    }
}

/* ex:47-70. Phase 1 traces every live thread (parallel, disjoint outputs); phase 2 resolves the two
 * atomicAdd streams in ascending thread index. Rays with pixel_idx == ORC_INACTIVE_PIXEL (true-size
 * padding, never produced with reference-legal sizes) are dropped: neither hit nor miss. */
void orc_extend(orc_ctx *c, uint32_t gx, uint32_t gy) {
    uint64_t n_threads = (uint64_t)gx * gy * 64u;
    uint32_t n = c->counters[2];
    if (n > n_threads) n = (uint32_t)n_threads;
    if (n > c->n_slots) n = c->n_slots;
    uint64_t max_depth = 0, nodes = 0, tests = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(max : max_depth) reduction(+ : nodes, tests)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        const orc_ray *ray = &c->rays[idx];
        if (ray->pixel_idx == ORC_INACTIVE_PIXEL) { c->trace_flag[idx] = 2; continue; }
        trace_stat st = {0, 0, 0};
        orc_hit_payload payload;
        memset(&payload, 0, sizeof payload);
        int h = trace_ray_bvh(c, ray, &payload, &st);
        payload.ray_idx = (uint32_t)idx; /* ex:58 */
        c->trace_out[idx] = payload;
        c->trace_flag[idx] = (uint8_t)h;
        if (st.max_depth > max_depth) max_depth = st.max_depth;
        nodes += st.nodes;
        tests += st.tests;
    }
    if (max_depth > c->stat_max_depth) c->stat_max_depth = max_depth;
    c->stat_nodes += nodes;
    c->stat_tests += tests;
    c->stat_rays += n;
    for (uint32_t idx = 0; idx < n; idx++) {
        if (c->trace_flag[idx] == 1) {
            uint32_t slot = c->counters[1]++; /* ex:59 */
            if (slot < c->n_slots) c->hits[slot] = c->trace_out[idx];
        } else if (c->trace_flag[idx] == 0) {
            uint32_t slot = c->counters[0]++; /* ex:61 */
            if (slot < c->n_slots) c->misses[slot] = idx;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * shade  (sh:56-176, 203-216)
 * ---------------------------------------------------------------------------------------------- */
/* sh:203-216 */
static v3 rng_next_vec3in_unit_sphere(uint32_t *state) {
    float r = orc_pow(orc_rng_next_float(state), 0.33333f);
    float cos_theta = 1.0f - 2.0f * orc_rng_next_float(state);
    float sin_theta = orc_sqrt(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * ORC_PI * orc_rng_next_float(state);
    float sphi, cphi;
    orc_sincos(phi, &sphi, &cphi);
    float x = r * sin_theta * cphi;
    float y = r * sin_theta * sphi;
    float z = r * cos_theta;
    return v3_make(x, y, z);
}
/* sh:158-162 */
static float schlick(float cosine, float refraction_index) {
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * orc_pow(1.0f - cosine, 5.0f);
}
/* sh:164-166: r - 2.0 * dot(r,n) * n, evaluated left to right */
static v3 reflect_(v3 r, v3 n) {
    float k = 2.0f * v3_dot(r, n);
    return v3_sub(r, v3_scale(k, n));
}
/* sh:168-176 */
static int refract_(v3 uv, v3 n, float ri, v3 *dir) {
    float cos_theta = v3_dot(uv, n);
    float k = 1.0f - ri * ri * (1.0f - cos_theta * cos_theta);
    if (k >= 0.0f) {
        float m = ri * cos_theta + orc_sqrt(k);
        *dir = v3_sub(v3_scale(ri, uv), v3_scale(m, n));
        return 1;
    }
    return 0;
}

void orc_shade(orc_ctx *c, uint32_t gx, uint32_t gy) {
    uint64_t n_threads = (uint64_t)gx * gy * 64u;
    uint32_t n = c->counters[1]; /* sh:66 */
    if (n > n_threads) n = (uint32_t)n_threads;
    if (n > c->n_slots) n = c->n_slots;
    uint32_t ext_base = c->counters[2];
    uint32_t W = c->frame.width;
#pragma omp parallel for schedule(static, 1024)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        /* sh:62-65, global_invocation_id of a gx x gy dispatch of 8x8 workgroups */
        uint32_t workgroup_index = (uint32_t)idx / 64u, local_index = (uint32_t)idx % 64u;
        uint32_t id_x = (workgroup_index % gx) * 8u + local_index % 8u;
        uint32_t id_y = (workgroup_index / gx) * 8u + local_index / 8u;
        orc_hit_payload payload = c->hits[idx];              /* sh:76 */
        const orc_ray *ray = &c->rays[payload.ray_idx];      /* sh:78 */
        uint32_t pixel_idx = ray->pixel_idx;
        if (c->p.rng_mode == ORC_RNG_PIXEL) { /* build-side order-independent keying (deviates from sh:72) */
            id_x = pixel_idx % W;
            id_y = pixel_idx / W;
        }
        uint32_t rng_state = orc_init_rng(id_x, id_y, c->frame.width, c->frame.frame); /* sh:71-72 */
        orc_advance(&rng_state, c->frame.sample_number * 10u);                         /* sh:73 */
        const orc_sphere *sphere = c->triangles ? NULL : &c->spheres[payload.sphere_idx];
        const orc_triangle *tri = c->triangles ? &c->triangles[payload.sphere_idx] : NULL;
        uint32_t mat_idx = tri ? tri->material_idx : sphere->material_idx;
        const orc_material *mat = &c->materials[mat_idx];
        float *px = &c->image[3 * (size_t)local_pixel(c, pixel_idx)];
        px[0] = px[0] * mat->albedo[0]; /* sh:84-87 */
        px[1] = px[1] * mat->albedo[1];
        px[2] = px[2] * mat->albedo[2];
        uint32_t mat_type = payload.mat_type;
        /* sh:91-93: p = origin + t * direction (vec4), n = normalize(p - center).xyz */
        v4 p = {ray->origin[0] + payload.t * ray->direction[0], ray->origin[1] + payload.t * ray->direction[1],
                ray->origin[2] + payload.t * ray->direction[2], ray->origin[3] + payload.t * ray->direction[3]};
        v3 nrm;
        if (tri) { /* build extension: geometric normal of the winding, normalize(cross(e1, e2)), never flipped */
            nrm = v3_normalize(cross3(v3_make(tri->e1[0], tri->e1[1], tri->e1[2]), v3_make(tri->e2[0], tri->e2[1], tri->e2[2])));
        } else {
            v4 pc = {p.x - sphere->center[0], p.y - sphere->center[1], p.z - sphere->center[2], p.w - sphere->center[3]};
            v4 n4 = v4_normalize(pc);
            nrm = v3_make(n4.x, n4.y, n4.z);
        }
        v3 rdir = v3_make(ray->direction[0], ray->direction[1], ray->direction[2]);
        v3 ext_dir = v3_make(0.0f, 0.0f, 0.0f);
        if (mat_type == 1u) { /* sh:110-114 */
            v3 rb = v3_normalize(rng_next_vec3in_unit_sphere(&rng_state));
            float fuzz = mat->fuzz;
            ext_dir = v3_add(reflect_(rdir, nrm), v3_scale(fuzz, rb));
        } else if (mat_type == 2u) { /* sh:115-151 */
            float refract_idx = mat->refract_index;
            v3 norm = nrm;
            v3 uv = v3_normalize(rdir);
            v3 neg_uv = v3_make(-uv.x, -uv.y, -uv.z);
            float cos_theta = orc_min(v3_dot(norm, neg_uv), 1.0f);
            float eta;
            if (cos_theta >= 0.0f) {
                eta = 1.0f / refract_idx;
            } else {
                eta = refract_idx;
                norm = v3_make(norm.x * -1.0f, norm.y * -1.0f, norm.z * -1.0f);
                cos_theta = cos_theta * -1.0f;
            }
            float reflectance = schlick(cos_theta, eta);
            v3 refr = v3_make(0.0f, 0.0f, 0.0f);
            if (refract_(uv, norm, eta, &refr)) {
                if (reflectance > orc_rng_next_float(&rng_state)) ext_dir = reflect_(uv, norm);
                else ext_dir = refr;
            } else {
                ext_dir = reflect_(uv, norm);
            }
        } else { /* sh:102-109: case 0u, default */
            v3 rb = v3_normalize(rng_next_vec3in_unit_sphere(&rng_state));
            ext_dir = v3_add(nrm, rb);
            if (v3_length(ext_dir) < 0.001f) ext_dir = nrm;
        }
        orc_ray er;
        er.origin[0] = p.x; er.origin[1] = p.y; er.origin[2] = p.z; er.origin[3] = p.w;
        er.direction[0] = ext_dir.x; er.direction[1] = ext_dir.y; er.direction[2] = ext_dir.z; er.direction[3] = 0.0f;
        er.inv_direction[0] = 1.0f / ext_dir.x; er.inv_direction[1] = 1.0f / ext_dir.y; er.inv_direction[2] = 1.0f / ext_dir.z;
        er.pixel_idx = pixel_idx;
        /* sh:155: atomicAdd resolved in ascending thread index => slot = base + idx */
        uint64_t slot = (uint64_t)ext_base + (uint64_t)idx;
        if (slot < c->n_slots) c->ext_rays[slot] = er;
    }
    c->counters[2] = ext_base + n;
}

/* ------------------------------------------------------------------------------------------------
 * miss_kernel (mk:13-38) and accumulate (ac:4-17)
 * ---------------------------------------------------------------------------------------------- */
void orc_miss(orc_ctx *c, uint32_t gx, uint32_t gy) {
    uint64_t n_threads = (uint64_t)gx * gy * 64u;
    uint32_t n = c->counters[0]; /* mk:24 */
    if (n > n_threads) n = (uint32_t)n_threads;
    if (n > c->n_slots) n = c->n_slots;
#pragma omp parallel for schedule(static, 4096)
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        uint32_t ray_idx = c->misses[idx];
        const orc_ray *ray = &c->rays[ray_idx];
        uint32_t pixel_idx = ray->pixel_idx;
        float a = 0.5f * (ray->direction[1] + 1.0f); /* mk:32: direction is NOT normalised after bounce 0 */
        float om = 1.0f - a;
        float cr = om * 1.0f + a * 0.5f; /* mk:33 */
        float cg = om * 1.0f + a * 0.7f;
        float cb = om * 1.0f + a * 1.0f;
        float *px = &c->image[3 * (size_t)local_pixel(c, pixel_idx)];
        px[0] *= cr; /* mk:35-37 */
        px[1] *= cg;
        px[2] *= cb;
    }
}

/* ac:4-17 has no bounds guard (its buffers are monitor-sized); the build guards at n_pixels. */
void orc_accumulate(orc_ctx *c, uint32_t gx, uint32_t gy) {
    uint64_t n = (uint64_t)gx * gy * 64u;
    if (n > c->n_pixels) n = c->n_pixels;
#pragma omp parallel for schedule(static, 16384)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        c->accumulated[3 * i + 0] += c->image[3 * i + 0];
        c->accumulated[3 * i + 1] += c->image[3 * i + 1];
        c->accumulated[3 * i + 2] += c->image[3 * i + 2];
    }
}

/* ------------------------------------------------------------------------------------------------
 * host loop: pt:291-368 for ONE sample (SPF = 1, wc/parameters.rs:5)
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_render_sample(orc_ctx *c) {
    uint32_t W = c->p.width, H = c->p.height;
    c->progress_frame += 1; /* parameters.rs:78-83: first frame is 1 */
    orc_frame_buffer f = {W, H, c->progress_frame, 0};
    orc_set_frame(c, &f);                                  /* pt:296-297 */
    orc_reset_image(c);                                    /* pt:305-306 */
    memset(c->rays, 0, sizeof(orc_ray) * c->n_slots);      /* pt:309-310 */
    memset(c->ext_rays, 0, sizeof(orc_ray) * c->n_slots);
    /* pt:313-318 with the true-size rule: ceil-div tiles, padded ray count (== W*H for multiples of 8) */
    uint32_t gx = c->tiles_x, gy = c->tiles_y_local;
    uint32_t counters[16] = {0};
    counters[2] = gx * gy * 64u;
    orc_set_counters(c, counters);
    orc_generate_rays(c, gx, gy, 1);
    uint32_t wavefront = 0;
    uint32_t ex, ey;
    orc_workgroup_size_64(counters[2], &ex, &ey); /* pt:322 */
    c->table_rows = 0;
    while (wavefront < c->p.max_wavefronts) {     /* pt:323 */
        uint32_t rays_in = c->counters[2];
        orc_extend(c, ex, ey);                    /* pt:325 */
        uint32_t num_misses = c->counters[0], num_hits = c->counters[1];
        /* rays actually traced: true-size padding rays are in rays_in but are neither hits nor misses */
        c->totals[0] += (uint64_t)num_hits + num_misses; c->totals[1] += num_hits; c->totals[2] += num_misses;
        uint32_t row = c->table_rows < 64 ? c->table_rows++ : 63;
        c->table[row][0] = rays_in; c->table[row][1] = num_hits; c->table[row][2] = num_misses; c->table[row][3] = 0;
        if (num_misses < c->p.miss_floor) break;  /* pt:332 */
        c->counters[2] = 0;                       /* pt:335-336 */
        uint32_t sx, sy, mx, my;
        orc_workgroup_size_64(num_hits, &sx, &sy);
        orc_workgroup_size_64(num_misses, &mx, &my);
        orc_shade(c, sx, sy);                     /* pt:339 */
        orc_miss(c, mx, my);                      /* pt:340 */
        c->table[row][3] = 1;
        uint32_t num_extension = c->counters[2];  /* pt:343-345 */
        orc_swap_ray_queues(c);                   /* pt:348 */
        orc_workgroup_size_64(num_extension, &ex, &ey); /* pt:350 */
        uint32_t nc[16] = {0};
        nc[2] = num_extension;                    /* pt:352 */
        orc_set_counters(c, nc);
        wavefront += 1;
    }
    uint32_t ax, ay;
    orc_workgroup_size_64(c->n_pixels, &ax, &ay);
    orc_accumulate(c, ax, ay);                    /* pt:362 */
    c->accumulated_samples += 1;                  /* pt:363 */
    return wavefront;
}

uint32_t orc_bounce_table(const orc_ctx *c, uint32_t *rows4, uint32_t max_rows) {
    uint32_t n = c->table_rows < max_rows ? c->table_rows : max_rows;
    memcpy(rows4, c->table, sizeof(uint32_t) * 4 * n);
    return n;
}
void orc_totals(const orc_ctx *c, uint64_t out[3]) { memcpy(out, c->totals, sizeof c->totals); }
void orc_trace_stats(const orc_ctx *c, uint64_t out[4]) {
    out[0] = c->stat_max_depth; out[1] = c->stat_nodes; out[2] = c->stat_tests; out[3] = c->stat_rays;
}

/* wavefront_common/shaders/display_shader.wgsl:50-52: sqrt(invN * color); BGRA8 swapchain write clamps */
void orc_tonemap_rgb8(const float *acc, uint32_t n_pixels, uint32_t n_samples, uint8_t *rgb) {
    float inv_n = 1.0f / (float)n_samples;
    for (size_t i = 0; i < 3 * (size_t)n_pixels; i++) {
        float v = orc_sqrt(inv_n * acc[i]);
        if (!(v > 0.0f)) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        rgb[i] = (uint8_t)(v * 255.0f + 0.5f);
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the following parallel regions (the CPU baseline is timed at the box's CPU share and, where that is not a quota, at every
 * core of the affinity mask as well); results do not depend on the thread count (stable compactions, disjoint outputs) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
