/*
 * wfpt_oracle.h -- ORACLE: CPU restatement of the reference's wavefront kernel chain.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under wavefront_path_tracer_amd/ may include, link or call this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" for floating point. The reference (rchiaramo/wavefront_path_tracer
 * @ 2024_10_08) ships no golden images, fixtures or assertions (its single #[test], camera.rs:72-86,
 * only prints), its CPU renderer cpu_wavefront_pt has no source in the snapshot, and no Rust/WGSL
 * toolchain exists here, so no output of the reference can be produced. What IS pinned: the integer RNG
 * (jenkins / PCG-RXS-M-XS / the as-written `advance`), `workgroup_size_64` and the camera constants,
 * against known-answer values derived from the reference's formulas -- see
 * tests/test_oracle_known_answers.py.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference root).
 */
#ifndef WFPT_ORACLE_H
#define WFPT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* wavefront_common/src/sphere.rs:3-11, extend.wgsl:10-15 -- 32 B */
typedef struct { float center[4]; float radius; uint32_t material_idx, material_type, _buffer; } orc_sphere;
/* wavefront_common/src/material.rs:12-20, shade.wgsl:19-24 -- 32 B */
typedef struct { float albedo[4]; float fuzz, refract_index; uint32_t material_type, _buffer; } orc_material;
/* wavefront_common/src/bvh.rs:38-45, extend.wgsl:3-8 -- 32 B */
typedef struct { float aabb_min[3]; uint32_t left_first; float aabb_max[3]; uint32_t prim_count; } orc_bvh_node;
/* wavefront_common/src/camera_controller.rs:161-185, generate_rays.wgsl:13-19 -- 32 B */
typedef struct { float position[4]; float pitch, yaw, defocus_radius, focus_distance; } orc_gpu_camera;
/* wavefront_common/src/gpu_structs.rs:5-12, generate_rays.wgsl:21-26 -- 16 B */
typedef struct { uint32_t width, height, frame, sample_number; } orc_frame_buffer;
/* Build extension (the reference has spheres only, README.md:22-26 lists other shapes as future work):
 * triangle as vertex + two edges -- 48 B */
typedef struct { float v0[3]; uint32_t material_idx; float e1[3]; uint32_t material_type; float e2[3]; uint32_t _pad; } orc_triangle;
/* extend.wgsl:17-22 (device layout) -- 48 B */
typedef struct { float origin[4]; float direction[4]; float inv_direction[3]; uint32_t pixel_idx; } orc_ray;
/* extend.wgsl:24-29 -- 16 B */
typedef struct { float t; uint32_t ray_idx, sphere_idx, mat_type; } orc_hit_payload;

enum { ORC_RNG_DISPATCH = 0, ORC_RNG_PIXEL = 1 };
#define ORC_INACTIVE_PIXEL 0xffffffffu

typedef struct {
    uint32_t width, height;
    uint32_t max_wavefronts; /* path_tracer.rs:323 (50) */
    uint32_t miss_floor;     /* path_tracer.rs:332 (128) */
    uint32_t rng_mode;       /* ORC_RNG_* : how shade keys its RNG (shade.wgsl:72) */
    uint32_t tile_rank, tile_world; /* build-side pixel-tile sharding: rank owns 8-row bands k % world == rank */
} orc_params;

typedef struct orc_ctx orc_ctx;

/* ---- host-side inputs ---- */
void     orc_scene_rng_seed(uint64_t state[2], uint64_t seed);
float    orc_scene_rng_f32(uint64_t state[2]);
/* scene.rs:12-46: returns sphere count (5); materials count equals it */
uint32_t orc_scene_new(orc_sphere *spheres, orc_material *materials);
/* scene.rs:48-107 with a seeded generator: capacity >= 488 each; returns sphere count (== material count) */
uint32_t orc_scene_book_one_final(uint64_t seed, orc_sphere *spheres, orc_material *materials);
/* bvh.rs:147-210: reorders spheres in place, writes <= 2*n nodes (n >= 1), returns node count */
uint32_t orc_build_bvh(orc_sphere *spheres, uint32_t n, orc_bvh_node *nodes);
/* build extension: the bvh.rs builder over triangles with n_bins bins per axis */
uint32_t orc_build_bvh_triangles(orc_triangle *tris, uint32_t n, orc_bvh_node *nodes, uint32_t n_bins);
/* BASELINE config 5: seeded random triangle soup; writes n triangles and 3 materials, returns 3 */
uint32_t orc_scene_random_mesh(uint64_t seed, uint32_t n, orc_triangle *tris, orc_material *materials);
/* camera.rs:11-24 */
void     orc_camera_new(const float look_from[3], const float look_at[3], float *pitch, float *yaw);
/* camera.rs:41-69: 16 floats, column-major (4 columns of 4) */
void     orc_view_transform(const float position[3], float pitch, float yaw, float view[16]);
/* projection_matrix.rs:21-37 */
void     orc_p_inv(float vfov_rad, float aspect, float z_near, float z_far, float p_inv[16]);
/* camera_controller.rs:173-185 */
void     orc_gpu_camera_new(const float position[3], float pitch, float yaw,
                            float defocus_angle_rad, float focus_distance, orc_gpu_camera *out);
float    orc_to_radians(float deg);
/* path_tracer.rs:282-289 */
void     orc_workgroup_size_64(uint32_t x, uint32_t *gx, uint32_t *gy);

/* ---- math / rng probes for known-answer tests ---- */
uint32_t orc_probe_jenkins(uint32_t x);
uint32_t orc_probe_init_rng(uint32_t px, uint32_t py, uint32_t res_x, uint32_t frame);
uint32_t orc_probe_next_int(uint32_t *state);
float    orc_probe_next_float(uint32_t *state);
uint32_t orc_probe_advance(uint32_t state, uint32_t n);
void     orc_probe_sincos(const float *x, float *s, float *c, size_t n);
void     orc_probe_pow(const float *x, const float *y, float *out, size_t n);

/* ---- the kernel chain ---- */
orc_ctx *orc_create(const orc_params *p,
                    const orc_sphere *spheres, uint32_t n_spheres,
                    const orc_material *materials, uint32_t n_materials,
                    const orc_bvh_node *nodes, uint32_t n_nodes,
                    const orc_gpu_camera *camera, const float inv_proj[16], const float view[16]);
orc_ctx *orc_create_mesh(const orc_params *p, const orc_triangle *tris, uint32_t n_tris,
                         const orc_material *materials, uint32_t n_materials,
                         const orc_bvh_node *nodes, uint32_t n_nodes,
                         const orc_gpu_camera *camera, const float inv_proj[16], const float view[16]);
void     orc_destroy(orc_ctx *);
void     orc_set_frame(orc_ctx *, const orc_frame_buffer *);
void     orc_set_counters(orc_ctx *, const uint32_t c[16]);
void     orc_get_counters(const orc_ctx *, uint32_t c[16]);
void     orc_reset_image(orc_ctx *);       /* path_tracer.rs:305-306 */
void     orc_reset_accumulated(orc_ctx *); /* path_tracer.rs:248-250 */
void     orc_swap_ray_queues(orc_ctx *);   /* path_tracer.rs:348 (wgpu_state.rs:115-130) */
void     orc_write_rays(orc_ctx *, const orc_ray *rays, uint32_t n); /* test hook: caller-made rays into the ray queue */

/* Stage kernels with the reference's dispatch semantics (gx*gy workgroups of 8x8). */
void orc_generate_rays(orc_ctx *, uint32_t gx, uint32_t gy, int true_size);
void orc_extend(orc_ctx *, uint32_t gx, uint32_t gy);
void orc_shade(orc_ctx *, uint32_t gx, uint32_t gy);
void orc_miss(orc_ctx *, uint32_t gx, uint32_t gy);
void orc_accumulate(orc_ctx *, uint32_t gx, uint32_t gy);

/* path_tracer.rs:291-368: one sample (frame += 1) of the wavefront loop. Returns wavefronts run. */
uint32_t orc_render_sample(orc_ctx *);
uint32_t orc_frame(const orc_ctx *);
uint32_t orc_accumulated_samples(const orc_ctx *);

/* ---- read-back (all in the reference's own layouts) ---- */
uint32_t orc_n_pixels(const orc_ctx *);
const orc_ray         *orc_rays(const orc_ctx *);
const orc_ray         *orc_extension_rays(const orc_ctx *);
const orc_hit_payload *orc_hits(const orc_ctx *);
const uint32_t        *orc_misses(const orc_ctx *);
const float           *orc_image(const orc_ctx *);       /* 3*n_pixels */
const float           *orc_accumulated(const orc_ctx *); /* 3*n_pixels */
/* per-bounce table of the LAST sample: rows of (rays_in, hits, misses, shaded) */
uint32_t orc_bounce_table(const orc_ctx *, uint32_t *rows4, uint32_t max_rows);
/* totals over all samples so far: rays into extend, hits, misses */
void     orc_totals(const orc_ctx *, uint64_t out[3]);
/* traversal statistics since creation: [0] max stack depth, [1] node visits, [2] sphere tests, [3] rays traced */
void     orc_trace_stats(const orc_ctx *, uint64_t out[4]);
/* extend.wgsl:141-153 (USE_BVH == false branch): brute-force closest hit, for cross-checks */
int      orc_trace_brute(const orc_ctx *, const orc_ray *ray, orc_hit_payload *out);
int      orc_trace_bvh(orc_ctx *, const orc_ray *ray, orc_hit_payload *out);
/* diagnostics: inner-node and leaf steps of each of the first n rays of the current ray queue */
void     orc_ray_steps(orc_ctx *, uint32_t n, uint16_t *inner_steps, uint16_t *leaf_steps);
void     orc_ray_rounds(orc_ctx *, uint32_t n, uint8_t *segs, uint8_t *n_leaves); /* diagnostics: inner visits per while-while round */
void     orc_prim_hit_t(orc_ctx *, uint32_t n, uint32_t prim, float *t_out); /* diagnostics: t of one primitive per ray */
void     orc_ray_rounds_init(orc_ctx *, uint32_t n, const float *init_nearest, int test_root, uint8_t *segs, uint8_t *n_leaves); /* diagnostics */
void     orc_sim_postpone(orc_ctx *, uint32_t n, uint32_t Q, uint64_t out[8]); /* diagnostics: wave64 schedule model with postponed leaves */
void     orc_traversal_profile(orc_ctx *, uint32_t n, uint64_t out[16]); /* diagnostics: push / pop / climb counts of the reference traversal */
/* display_shader.wgsl:50-52: sqrt(acc / n) -> 8-bit RGB */
void     orc_tonemap_rgb8(const float *acc, uint32_t n_pixels, uint32_t n_samples, uint8_t *rgb);
int      orc_num_threads(void);
void     orc_set_num_threads(int n);

/* model of the DEVICE's conservative traversal vs the reference's, over the ray queue (see wfpt_oracle.c) */
void     orc_probe_normalize3(const float *in, float *out, size_t n); /* test probe */
uint64_t orc_model_handovers(void); /* diagnostics */
uint32_t orc_model_mismatches(orc_ctx *c, uint32_t n, const float extent[7], int leaf_exact, uint32_t *out, uint32_t max_out);

#ifdef __cplusplus
}
#endif
#endif
