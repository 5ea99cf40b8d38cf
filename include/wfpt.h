/*
 * wfpt.h -- C ABI of the MI355X-native wavefront path-tracing kernel chain
 *           generate_rays -> extend -> shade -> miss_kernel -> accumulate.
 *
 * Drop-in boundary for ONE path of rchiaramo/wavefront_path_tracer @ 2024_10_08: the kernel-stage API
 * `Kernel::new / Kernel::run / Kernel::get_timing` (gpu_wavefront_pt/src/kernel.rs:26-146) together with the
 * buffers and the wavefront loop `PathTracer::new / run` owns (gpu_wavefront_pt/src/path_tracer.rs:43-371).
 * Everything here is `extern "C"`, plain pointers and sizes; struct layouts are byte-identical to the
 * `#[repr(C)]` structs of `wavefront_common` that the reference uploads with bytemuck::cast_slice
 * (path_tracer.rs:120-156), so a Rust `-sys` shim can pass them through unchanged.
 *
 * Conventions: functions returning `int` return WFPT_OK (0) or a negative wfpt_status and never throw;
 * `wfpt_last_error` gives the message. A context is bound to one HIP device and one stream; it is not
 * thread-safe, independent contexts may be used from different threads/processes (one per GPU).
 * Calls are asynchronous on the context's stream unless they read data back.
 *
 * All citations are relative to the reference root.
 */
#ifndef WFPT_H
#define WFPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ data model (wavefront_common) */

/* wavefront_common/src/sphere.rs:3-11 == extend.wgsl:10-15. center.w is 1.0 (sphere.rs:18-20). */
typedef struct wfpt_sphere {
    float center[4];
    float radius;
    uint32_t material_idx;
    uint32_t material_type;
    uint32_t _buffer;
} wfpt_sphere;

/* wavefront_common/src/material.rs:12-20 == shade.wgsl:19-24. 0 Lambertian, 1 Metal, 2 Dielectric. */
typedef struct wfpt_material {
    float albedo[4];
    float fuzz;
    float refract_index;
    uint32_t material_type;
    uint32_t _buffer;
} wfpt_material;

/* wavefront_common/src/bvh.rs:38-45 == extend.wgsl:3-8. prim_count > 0: leaf, left_first = first
 * sphere; else left_first = left child and right child = left_first + 1. nodes[1] is a pad (bvh.rs:160). */
typedef struct wfpt_bvh_node {
    float aabb_min[3];
    uint32_t left_first;
    float aabb_max[3];
    uint32_t prim_count;
} wfpt_bvh_node;

/* BUILD EXTENSION -- the reference intersects spheres only (extend.wgsl:185-210; "other geometric shapes" and
 * OBJ loading are future items, README.md:22-26). A triangle is a vertex and two edge vectors; the hit test
 * is Moeller-Trumbore with extend.wgsl's (t_min, t_nearest) window, the shading normal is
 * normalize(cross(e1, e2)) (never flipped), HitPayload.sphere_idx carries the triangle index. 48 B. */
typedef struct wfpt_triangle {
    float v0[3];
    uint32_t material_idx;
    float e1[3];
    uint32_t material_type;
    float e2[3];
    uint32_t _pad;
} wfpt_triangle;

/* wavefront_common/src/camera_controller.rs:161-185 == generate_rays.wgsl:13-19 */
typedef struct wfpt_gpu_camera {
    float position[4];
    float pitch;
    float yaw;
    float defocus_radius;
    float focus_distance;
} wfpt_gpu_camera;

/* wavefront_common/src/gpu_structs.rs:5-12 == generate_rays.wgsl:21-26 */
typedef struct wfpt_frame_buffer {
    uint32_t width;
    uint32_t height;
    uint32_t frame;
    uint32_t sample_number;
} wfpt_frame_buffer;

/* extend.wgsl:17-22: the reference's device Ray. Used only by the read-back functions; on the device
 * rays live in SoA planes. */
typedef struct wfpt_ray {
    float origin[4];
    float direction[4];
    float inv_direction[3];
    uint32_t pixel_idx;
} wfpt_ray;

/* extend.wgsl:24-29 */
typedef struct wfpt_hit_payload {
    float t;
    uint32_t ray_idx;
    uint32_t sphere_idx;
    uint32_t mat_type;
} wfpt_hit_payload;

/* Compile-time layout checks: these are the byte layouts the reference uploads with bytemuck::cast_slice
 * (path_tracer.rs:120-156) and declares in WGSL (extend.wgsl:3-29, shade.wgsl:19-24, generate_rays.wgsl:13-26). */
#if defined(__cplusplus)
#define WFPT_LAYOUT_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define WFPT_LAYOUT_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
WFPT_LAYOUT_ASSERT(sizeof(wfpt_sphere) == 32 && offsetof(wfpt_sphere, radius) == 16 && offsetof(wfpt_sphere, material_idx) == 20 &&
                       offsetof(wfpt_sphere, material_type) == 24,
                   "Sphere: sphere.rs:3-11, 32 bytes");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_material) == 32 && offsetof(wfpt_material, fuzz) == 16 && offsetof(wfpt_material, refract_index) == 20 &&
                       offsetof(wfpt_material, material_type) == 24,
                   "Material: material.rs:12-20, 32 bytes");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_bvh_node) == 32 && offsetof(wfpt_bvh_node, left_first) == 12 && offsetof(wfpt_bvh_node, aabb_max) == 16 &&
                       offsetof(wfpt_bvh_node, prim_count) == 28,
                   "BVHNode: bvh.rs:38-45, 32 bytes");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_triangle) == 48 && offsetof(wfpt_triangle, e1) == 16 && offsetof(wfpt_triangle, e2) == 32,
                   "wfpt_triangle: three 16-byte rows");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_gpu_camera) == 32 && offsetof(wfpt_gpu_camera, pitch) == 16 && offsetof(wfpt_gpu_camera, yaw) == 20 &&
                       offsetof(wfpt_gpu_camera, defocus_radius) == 24 && offsetof(wfpt_gpu_camera, focus_distance) == 28,
                   "GPUCamera: camera_controller.rs:161-185, 32 bytes");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_frame_buffer) == 16 && offsetof(wfpt_frame_buffer, frame) == 8 && offsetof(wfpt_frame_buffer, sample_number) == 12,
                   "GPUFrameBuffer: gpu_structs.rs:5-12, 16 bytes");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_ray) == 48 && offsetof(wfpt_ray, direction) == 16 && offsetof(wfpt_ray, inv_direction) == 32 &&
                       offsetof(wfpt_ray, pixel_idx) == 44,
                   "Ray: extend.wgsl:17-22, 48 bytes");
WFPT_LAYOUT_ASSERT(sizeof(wfpt_hit_payload) == 16 && offsetof(wfpt_hit_payload, ray_idx) == 4 && offsetof(wfpt_hit_payload, sphere_idx) == 8 &&
                       offsetof(wfpt_hit_payload, mat_type) == 12,
                   "HitPayload: extend.wgsl:24-29, 16 bytes");

/* ------------------------------------------------------------------ enums */

typedef enum wfpt_status {
    WFPT_OK = 0,
    WFPT_ERR_INVALID_ARGUMENT = -1,
    WFPT_ERR_HIP = -2,          /* a HIP runtime call failed; message holds hipGetErrorString */
    WFPT_ERR_OUT_OF_MEMORY = -3,
    WFPT_ERR_UNSUPPORTED = -4,  /* e.g. BVH deeper than the traversal supports */
    WFPT_ERR_NO_DEVICE = -5     /* no gfx950 device visible: there is NO CPU fallback */
} wfpt_status;

/* Stage names are the reference's shader basenames (kernel.rs:32; call sites path_tracer.rs:162,167,175,
 * 180,185). wfpt_stage_from_name maps the strings. The three per-material stages implement the
 * reference's README to-do "split shade into by-material shade kernels" (README.md:19). */
typedef enum wfpt_stage {
    WFPT_STAGE_GENERATE_RAYS = 0, /* "generate_rays" */
    WFPT_STAGE_EXTEND = 1,        /* "extend"        */
    WFPT_STAGE_SHADE = 2,         /* "shade"         */
    WFPT_STAGE_MISS = 3,          /* "miss_kernel"   */
    WFPT_STAGE_ACCUMULATE = 4,    /* "accumulate"    */
    WFPT_STAGE_SHADE_LAMBERTIAN = 5,
    WFPT_STAGE_SHADE_METAL = 6,
    WFPT_STAGE_SHADE_DIELECTRIC = 7,
    WFPT_STAGE_SCAN = 8, /* internal helper launched with extend (queue positions + loop control); only
                            appears in wfpt_render_sample_timed's per-stage times */
    /* The device-resident loop's fused launches (not dispatchable through wfpt_kernel_run; they appear in the
     * per-stage times of wfpt_render_timed). One launch per wavefront: */
    WFPT_STAGE_BOUNCE_FIRST = 9,  /* "bounce_first": generate_rays + extend of wavefront 0 */
    WFPT_STAGE_BOUNCE = 10,       /* "bounce": shade of wavefront b-1 + extend of wavefront b + miss_kernel of b-1 */
    WFPT_STAGE_BOUNCE_LAST = 11,  /* "bounce_last": shade + miss_kernel of the last wavefront */
    WFPT_STAGE_COMPACT = 12,      /* "compact": scenes beyond LDS only -- dense per-ray results into the queues */
    WFPT_STAGE_COUNT = 13
} wfpt_stage;

/* How shade keys its RNG (shade.wgsl:72 uses the dispatch's global_invocation_id):
 *  DISPATCH: exactly that, with every atomicAdd resolved in ascending thread index (stable queues).
 *            One legal execution of the reference; the default.
 *  PIXEL:    keyed by the ray's own pixel (x = pixel_idx % W, y = pixel_idx / W). Deviates from
 *            shade.wgsl:72, but makes the image independent of queue order, which tile sharding across
 *            GPUs and the per-material split need to be bit-identical to a single-queue render. */
typedef enum wfpt_rng_mode { WFPT_RNG_DISPATCH = 0, WFPT_RNG_PIXEL = 1 } wfpt_rng_mode;

enum {
    WFPT_FLAG_SPLIT_SHADE = 1u << 0, /* fused loop runs the three per-material shade stages */
    WFPT_FLAG_NO_GRAPH = 1u << 1,    /* fused loop launches kernels directly instead of replaying a hipGraph */
    WFPT_FLAG_UNFUSED = 1u << 2,     /* device-resident loop runs the stage kernels one by one (extend, scan, shade,
                                        miss_kernel per wavefront) instead of one fused bounce launch per wavefront.
                                        Same images bit for bit; WFPT_FLAG_SPLIT_SHADE implies it. */
    WFPT_FLAG_BINARY_BVH = 1u << 3,  /* scenes too large for LDS: walk the caller's binary tree as it is instead of the
                                        four-wide collapse built at wfpt_create (same hits; for comparisons) */
    WFPT_FLAG_NO_REFILL = 1u << 4,   /* scenes too large for LDS: lanes keep their ray until the whole 512-ray segment is
                                        done (the fused bounce kernel) instead of taking new rays as they finish */
    WFPT_FLAG_NO_LDS_SCENE = 1u << 5, /* treat the scene as too large for LDS even if it fits (experiments, tests of the
                                        HBM-resident traversal on small scenes) */
    WFPT_FLAG_EXACT_TRAVERSAL = 1u << 6, /* trace_ray / hit_bvh_node exactly as extend.wgsl:72-183 writes them: slab planes
                                        (b - o) * inv with min / max per axis, a missed box reports 1e30 (so the reference's
                                        `1e30 > 1e30` descent into doubly-missed pairs happens), the binary tree walked as it
                                        is. Default (flag clear): the same walk with a CONSERVATIVE box test (boxes grown by
                                        more than the test's rounding error) -- same hits, fewer instructions; the library
                                        falls back to the exact test by itself when a camera or an injected ray lies outside
                                        the range that bound covers. Same images bit for bit; for bisecting and proofs. */
    WFPT_FLAG_NO_BINNING = 1u << 7,  /* keep the hit queue in thread order (a work item of the fused loop = 512 consecutive hits). This
                                        is the default in both RNG modes since the end of round 5 (rounds 4-5 binned contexts of at
                                        least 3/4 Mpixel in WFPT_RNG_PIXEL by default); the flag is accepted and wins over
                                        WFPT_FLAG_BINNING. */
    WFPT_FLAG_BINNING = 1u << 8      /* WFPT_RNG_PIXEL, scenes in LDS: the class-binned loop -- every segment's hits stored sorted by cost
                                        class (the dominant primitive | lambertian | metal | dielectric), a work item = 512 hits of ONE
                                        class: 32.8 instead of 29.0 of 64 lanes per vector instruction, and level with the thread-ordered
                                        loop end to end at 1920x1080 (behind it on smaller slabs), hence opt-in. Same images bit for bit. With
                                        WFPT_RNG_DISPATCH wfpt_create refuses the flag (WFPT_ERR_INVALID_ARGUMENT): shade.wgsl:72 keys
                                        its RNG on the dispatch's thread index, i.e. on the order of the hit queue, which this loop
                                        gives up. (Round 4 carried that order through the binning -- thread indices in the records, a hit
                                        flag per ray, a rank table per wavefront -- measured it 5.8 % slower than the thread-ordered loop
                                        and round 5 removed it; so was the two-chain experiment, WFPT_FLAG_TWO_CHAINS, bit 9:
                                        profiles/r04_rejected_experiments.txt.) */
};

#define WFPT_INACTIVE_PIXEL 0xffffffffu

typedef struct wfpt_params {
    uint32_t width;          /* viewport (RenderParameters::viewport_size, parameters.rs:43-45) */
    uint32_t height;
    uint32_t max_pixels;     /* buffer capacity, the reference's max_window_size (path_tracer.rs:44); 0 = width*height */
    uint32_t max_wavefronts; /* path_tracer.rs:323: 50 */
    uint32_t miss_floor;     /* path_tracer.rs:332: loop exits before shading when misses < 128 */
    uint32_t rng_mode;       /* wfpt_rng_mode */
    uint32_t flags;          /* WFPT_FLAG_* */
    uint32_t tile_rank;      /* pixel-tile sharding: this context owns the 8-pixel-high bands k with */
    uint32_t tile_world;     /*   k % tile_world == tile_rank; 0 or 1 = whole image */
    int32_t device;          /* HIP device ordinal */
    uint32_t batch;          /* samples kept in flight per launch by wfpt_render (1..128, 1..64 for the stage-by-stage
                                loop; 0 = 16; larger values are clamped). Results are bit-identical for every value:
                                samples are independent and accumulate in order. */
} wfpt_params;

typedef struct wfpt_ctx wfpt_ctx;

/* ------------------------------------------------------------------ host-side data model helpers
 * (no GPU needed). They restate wavefront_common so a host without the Rust crate can build inputs. */

/* scene.rs:12-46: 5 spheres, 5 materials; arrays need capacity 5. Returns the count. */
uint32_t wfpt_scene_new(wfpt_sphere *spheres, wfpt_material *materials);
/* scene.rs:48-107 with a SEEDED generator (the reference's thread_rng is unseeded): PCG32 stream 54,
 * f32 = top 24 bits * 2^-24, same distributions and draw order. Returns the count (<= 488), or 0 if
 * capacity is too small. */
uint32_t wfpt_scene_book_one_final(uint64_t seed, wfpt_sphere *spheres, wfpt_material *materials, uint32_t capacity);
/* bvh.rs:147-210 (4096-bin SAH): reorders `spheres` in place, writes at most 2*n nodes. */
int wfpt_build_bvh(wfpt_sphere *spheres, uint32_t n_spheres, wfpt_bvh_node *nodes, uint32_t node_capacity,
                   uint32_t *n_nodes);
/* Build extension: the same builder over triangles (bin key = centroid v0 + (e1 + e2)/3) with a caller-chosen
 * number of bins per axis (bvh.rs:4's 4096 is O(10^10) work for a million primitives). Reorders in place. */
int wfpt_build_bvh_triangles(wfpt_triangle *triangles, uint32_t n_triangles, wfpt_bvh_node *nodes,
                             uint32_t node_capacity, uint32_t *n_nodes, uint32_t n_bins);
/* Build extension: the same two builders ON THE DEVICE (csrc/wfpt_bvh_build.hip). Same
 * arguments, same results byte for byte -- node array in bvh.rs's depth-first numbering, primitives reordered in
 * place -- so either can feed wfpt_create / wfpt_create_mesh. `device` is the HIP device ordinal; `device_ms`
 * (may be NULL) receives the time between the first and the last build kernel (host<->device copies of the
 * inputs / outputs excluded). WFPT_ERR_NO_DEVICE without a GPU: there is no silent fall back to the host builder. */
int wfpt_build_bvh_device(wfpt_sphere *spheres, uint32_t n_spheres, wfpt_bvh_node *nodes, uint32_t node_capacity,
                          uint32_t *n_nodes, int device, float *device_ms);
int wfpt_build_bvh_triangles_device(wfpt_triangle *triangles, uint32_t n_triangles, wfpt_bvh_node *nodes,
                                    uint32_t node_capacity, uint32_t *n_nodes, uint32_t n_bins, int device,
                                    float *device_ms);
/* Build extension (README.md:25 "start loading in obj files"): reads the `v` and `f` records of a
 * Wavefront OBJ file into wfpt_triangle (v0, e1 = v1 - v0, e2 = v2 - v0), fan-triangulating polygons; `f` entries may
 * be `i`, `i/t`, `i//n` or `i/t/n`, 1-based or negative (relative). Every triangle gets `material_idx` /
 * `material_type`. With triangles == NULL it only counts. Returns WFPT_OK and the count in *n_triangles;
 * WFPT_ERR_INVALID_ARGUMENT for an unreadable file, a bad index or too small a capacity. */
int wfpt_load_obj(const char *path, wfpt_triangle *triangles, uint32_t capacity, uint32_t *n_triangles,
                  uint32_t material_idx, uint32_t material_type);
/* BASELINE config 5: seeded triangle soup -- centres U[-10,10]^3, edges U[-0.05,0.05]^3, material i % 3 over
 * {Lambertian 0.7, Metal 0.8 fuzz 0.1, Dielectric 1.5}. Writes n triangles and 3 materials; returns 3. */
uint32_t wfpt_scene_random_mesh(uint64_t seed, uint32_t n_triangles, wfpt_triangle *triangles, wfpt_material *materials);
/* camera.rs:11-24 */
void wfpt_camera_new(const float look_from[3], const float look_at[3], float *pitch, float *yaw);
/* camera.rs:41-69: world-from-camera, 16 floats column-major (columns right, up, dir, position) */
void wfpt_view_transform(const float position[3], float pitch, float yaw, float view[16]);
/* projection_matrix.rs:21-37: inverse projection, 16 floats column-major */
void wfpt_p_inv(float vfov_rad, float aspect_ratio, float z_near, float z_far, float p_inv[16]);
/* camera_controller.rs:173-185 */
void wfpt_gpu_camera_new(const float position[3], float pitch, float yaw, float defocus_angle_rad,
                         float focus_distance, wfpt_gpu_camera *out);
/* f32::to_radians */
float wfpt_to_radians(float degrees);
/* camera_controller.rs:125-158 (CameraController::update_camera): moves the camera by the pressed-key amounts
 * {forward, backward, right, left, up, down} and the pending mouse rotation {horizontal, vertical} over dt seconds,
 * clamps the pitch to +-(pi - 0.001), and zeroes `rotate` like the reference does. */
void wfpt_camera_controller_update(float position[3], float *pitch, float *yaw, const float amounts[6],
                                   float rotate[2], float speed, float sensitivity, float dt);
/* path_tracer.rs:282-289. The reference panics for x <= 64; this returns (1,1) there. */
void wfpt_workgroup_size_64(uint32_t x, uint32_t *gx, uint32_t *gy);
/* kernel.rs:32: shader basename -> stage; -1 if unknown */
int wfpt_stage_from_name(const char *name);
const char *wfpt_stage_name(int stage);

/* ------------------------------------------------------------------ context: PathTracer::new (path_tracer.rs:43-217) */

int wfpt_device_count(void);
/* Copies every input; no pointer is retained. Returns NULL on failure (wfpt_last_error(NULL)). */
wfpt_ctx *wfpt_create(const wfpt_params *params,
                      const wfpt_sphere *spheres, uint32_t n_spheres,
                      const wfpt_material *materials, uint32_t n_materials,
                      const wfpt_bvh_node *nodes, uint32_t n_nodes,
                      const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16]);
/* Same context over a triangle mesh (build extension). Scenes whose BVH does not fit a CU's LDS are traversed
 * from HBM / Infinity Cache by a second extend variant. */
wfpt_ctx *wfpt_create_mesh(const wfpt_params *params,
                           const wfpt_triangle *triangles, uint32_t n_triangles,
                           const wfpt_material *materials, uint32_t n_materials,
                           const wfpt_bvh_node *nodes, uint32_t n_nodes,
                           const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16]);
void wfpt_destroy(wfpt_ctx *ctx);
const char *wfpt_last_error(const wfpt_ctx *ctx);

/* Dynamic scenes (the reference builds its scene once, in PathTracer::new, path_tracer.rs:117-128):
 * replaces the scene of a live context. `spheres` / `triangles` are reordered in place by the BVH build, as
 * BVHTree::build_bvh_tree does (bvh.rs:182); the BVH is rebuilt on the context's device by wfpt_build_bvh_device /
 * wfpt_build_bvh_triangles_device (byte-identical to bvh.rs:147-210; n_bins = 0 means 32), the traversal's derived data
 * re-derived, and the accumulation and the frame counter reset like update_buffers does for a parameter change
 * (path_tracer.rs:240-277). The primitive count and kind may change; queues and images keep their size. The context is
 * unchanged if the arguments are refused; a HIP failure half way leaves it without a scene (only wfpt_destroy is safe). */
int wfpt_update_scene(wfpt_ctx *ctx, wfpt_sphere *spheres, uint32_t n_spheres, const wfpt_material *materials, uint32_t n_materials);
int wfpt_update_scene_mesh(wfpt_ctx *ctx, wfpt_triangle *triangles, uint32_t n_triangles, const wfpt_material *materials,
                           uint32_t n_materials, uint32_t n_bins);

/* Image chunking on one GPU (the reference's to-do "split rendering of image into chunks so that the buffers aren't so
 * big", README.md:20): renders n_samples of the whole frame as `chunks` band-interleaved slabs, one context after the
 * other (chunk k holds the 8-row bands j with j % chunks == k, so every device buffer is sized for 1/chunks of the
 * frame), and writes the accumulated frame (3 floats per pixel, row-major, width * height pixels) to host memory at
 * `rgb`. params->tile_rank / tile_world / max_pixels are ignored. With chunks > 1 params->rng_mode must be
 * WFPT_RNG_PIXEL (the dispatch-keyed RNG of shade.wgsl:72 depends on a ray's queue position, hence on the cut); the
 * frame is then bit-identical to the unchunked WFPT_RNG_PIXEL render as long as the loop-exit test of path_tracer.rs:332
 * (`misses < miss_floor`), which every chunk applies to its OWN miss count, fires in no chunk before it would for the
 * whole frame -- with miss_floor = 0 (wavefront limit only) it never does and the image is independent of the cut.
 * The same holds for contexts sharded with tile_rank / tile_world across GPUs. Blocking. */
int wfpt_render_chunked(const wfpt_params *params,
                        const wfpt_sphere *spheres, uint32_t n_spheres,
                        const wfpt_material *materials, uint32_t n_materials,
                        const wfpt_bvh_node *nodes, uint32_t n_nodes,
                        const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16],
                        uint32_t n_samples, uint32_t chunks, float *rgb);
int wfpt_render_chunked_mesh(const wfpt_params *params,
                             const wfpt_triangle *triangles, uint32_t n_triangles,
                             const wfpt_material *materials, uint32_t n_materials,
                             const wfpt_bvh_node *nodes, uint32_t n_nodes,
                             const wfpt_gpu_camera *camera, const float inv_proj[16], const float view[16],
                             uint32_t n_samples, uint32_t chunks, float *rgb);

/* frame_buffer.queue_for_gpu (path_tracer.rs:296-297, 366-367). The uniform set here is what the STAGE API
 * (wfpt_kernel_run) reads. The device-resident loop (wfpt_render_sample / wfpt_render) owns the frame uniform like
 * PathTracer::run does (path_tracer.rs:293-297): its next call overwrites it with {width, height,
 * wfpt_frame() + 1, sample_number 0}, whatever was set here. */
int wfpt_set_frame(wfpt_ctx *ctx, const wfpt_frame_buffer *frame);
/* update_buffers (path_tracer.rs:240-277): new camera / matrices / viewport; zeroes the accumulated image
 * and resets progress exactly as the reference does on any parameter change. */
int wfpt_update_render_parameters(wfpt_ctx *ctx, uint32_t width, uint32_t height, const wfpt_gpu_camera *camera,
                                  const float inv_proj[16], const float view[16]);
/* counter_buffer protocol (path_tracer.rs:313-316, 335-336, 352; extend.wgsl:41):
 * [0] miss count, [1] hit count, [2] rays in for extend / extension rays out of shade, [3..15] unused. */
int wfpt_set_counters(wfpt_ctx *ctx, const uint32_t counters[16]);
int wfpt_read_counters(wfpt_ctx *ctx, uint32_t counters[16]); /* blocking, like wgpu_state.rs:132-147 */
int wfpt_reset_image(wfpt_ctx *ctx);       /* image <- 1.0 (path_tracer.rs:305-306) */
int wfpt_reset_accumulated(wfpt_ctx *ctx); /* accumulated <- 0 (path_tracer.rs:248-250) */
/* RenderProgress::reset (parameters.rs:92-95) + the accumulation clear above: the next sample is frame 1 again. What
 * update_buffers does after a change (path_tracer.rs:248-250, 276) without re-uploading camera or matrices. */
int wfpt_reset_progress(wfpt_ctx *ctx);
int wfpt_clear_ray_queues(wfpt_ctx *ctx);  /* path_tracer.rs:309-310 */
/* copy_buffer_to_buffer(extension_ray_buffer -> ray_buffer) (path_tracer.rs:348, wgpu_state.rs:115-130)
 * done as a pointer swap. */
int wfpt_swap_ray_queues(wfpt_ctx *ctx);

/* Kernel::run((gx, gy)) (kernel.rs:107-140): gx*gy workgroups of 64 threads, thread index linearised as
 * in the shaders (workgroup_index*64 + local_index). Semantics per stage follow the WGSL entry points:
 *   generate_rays: width = 8*gx, height = 8*gy, one ray per thread (generate_rays.wgsl:42-91)
 *   extend:        threads idx < counters[2] trace; counters[1] += hits, counters[0] += misses (extend.wgsl:47-70)
 *   shade:         threads idx < counters[1]; counters[2] += rays emitted (shade.wgsl:56-156)
 *   miss_kernel:   threads idx < counters[0] (miss_kernel.wgsl:13-38)
 *   accumulate:    threads idx < pixel count (accumulate.wgsl:4-17; the reference has no guard)
 * counters[0] and [1] must be zero when extend runs and counters[2] zero when shade runs, as the
 * reference's host guarantees (path_tracer.rs:335-336, 352). */
int wfpt_kernel_run(wfpt_ctx *ctx, int stage, uint32_t gx, uint32_t gy);
/* Kernel::get_timing (kernel.rs:142-146, query_gpu.rs:26-43): blocks, returns the running mean (us) of
 * the last <= 10 timed dispatches of that stage; 0 if it never ran. */
float wfpt_kernel_timing_us(wfpt_ctx *ctx, int stage);

/* wfpt_render_sample: PathTracer::run for one sample (path_tracer.rs:291-368) with the whole wavefront loop resident on the
 * device: frame += 1, image <- 1, generate, up to max_wavefronts x (extend, shade, miss) with the
 * `misses < miss_floor` exit evaluated on the device, accumulate. No host synchronisation. Sizes that
 * are not multiples of 8 use true-size semantics: out-of-image lanes emit inactive rays. */
/* Which loop wfpt_render enqueues for this context, as decided at wfpt_create / wfpt_update_scene from the flags, the RNG mode, the size
 * of the slab and the scene (hosts and benchmarks label their figures with this instead of re-deriving the library's gates):
 *   WFPT_LOOP_STAGES        the stage kernels one by one (WFPT_FLAG_UNFUSED / WFPT_FLAG_SPLIT_SHADE, or more than 65535 queue segments)
 *   WFPT_LOOP_FUSED         one fused bounce launch per wavefront, hit queue in the reference's thread order (bounce_kernel)
 *   WFPT_LOOP_FUSED_BINNED  the same with the hit queue binned by cost class (WFPT_RNG_PIXEL, scenes in LDS; bounce_binned_kernel)
 *   WFPT_LOOP_REFILL        scenes beyond LDS: four-wide traversal with dynamic lane refill (refill_kernel)
 * Returns the kind, or a negative status. */
typedef enum wfpt_loop_kind { WFPT_LOOP_STAGES = 0, WFPT_LOOP_FUSED = 1, WFPT_LOOP_FUSED_BINNED = 2, WFPT_LOOP_REFILL = 3 } wfpt_loop_kind;
int wfpt_loop_kind_of(const wfpt_ctx *ctx);
int wfpt_render_sample(wfpt_ctx *ctx);
int wfpt_render(wfpt_ctx *ctx, uint32_t n_samples);
int wfpt_synchronize(wfpt_ctx *ctx);
uint32_t wfpt_frame(const wfpt_ctx *ctx);               /* RenderProgress.frame */
uint32_t wfpt_accumulated_samples(const wfpt_ctx *ctx); /* RenderProgress.accumulated_samples */
float wfpt_progress(const wfpt_ctx *ctx, uint32_t spp); /* PathTracer::progress (path_tracer.rs:219-221) */
/* Same loop with hipEvent pairs around every stage launch; adds the elapsed milliseconds per stage
 * into stage_ms[WFPT_STAGE_COUNT] (+= , caller zeroes) and counts launches in stage_launches (may be
 * NULL). Blocks until the sample is done. */
int wfpt_render_sample_timed(wfpt_ctx *ctx, float *stage_ms, uint32_t *stage_launches);
/* n_samples with the same batching as wfpt_render, timed per launch the same way. */
int wfpt_render_timed(wfpt_ctx *ctx, uint32_t n_samples, float *stage_ms, uint32_t *stage_launches);

/* ------------------------------------------------------------------ read-back (blocking) */

uint32_t wfpt_n_pixels(const wfpt_ctx *ctx);  /* pixels held by this context (whole 8-row bands when sharded) */
uint32_t wfpt_ray_capacity(const wfpt_ctx *ctx);
/* accumulated_image_buffer / image_buffer: 3 floats per pixel, stride 12, row-major (accumulate.wgsl:1-2) */
int wfpt_read_accumulated(wfpt_ctx *ctx, float *rgb, size_t n_floats);
int wfpt_read_image(wfpt_ctx *ctx, float *rgb, size_t n_floats);
/* device-to-device copy of the accumulated slab on the context's stream (for a RCCL gather) */
int wfpt_copy_accumulated_to_device(wfpt_ctx *ctx, void *device_ptr, size_t n_bytes);
/* ------------------------------------------------------------------ multi-GPU: RCCL gather of the band-sharded frame
 * BUILD-SIDE ADDITION (the reference is single-GPU). One process (or thread) per GPU creates a context
 * with tile_rank = r, tile_world = R: it renders the 8-row bands k with k % R == r, with no exchange inside the
 * bounce loop. The only collective is this gather of the accumulated slabs to rank 0 over xGMI:
 *   rank 0:    wfpt_comm_unique_id(id);  ... hand the 128 bytes to every rank by any means (MPI, a file, a socket) ...
 *   all ranks: wfpt_comm_init(ctx, id, r, R);                      (collective: ncclCommInitRank on the context's device)
 *   all ranks: wfpt_render(ctx, spp); wfpt_gather_accumulated(ctx); (asynchronous, ordered on the context's stream)
 *   rank 0:    wfpt_read_gathered(ctx, rgb, 3 * width * height);    (blocking; the assembled frame, row-major)
 * RCCL (librccl.so) is opened at run time; without it these calls return WFPT_ERR_UNSUPPORTED and nothing else in the
 * library depends on it. In the pixel-keyed RNG mode the gathered frame is bit-identical to a single-GPU render. */
#define WFPT_COMM_UNIQUE_ID_BYTES 128
int wfpt_comm_unique_id(void *id128);
int wfpt_comm_init(wfpt_ctx *ctx, const void *id128, int rank, int world);
int wfpt_gather_accumulated(wfpt_ctx *ctx);
/* The same gather bracketed by two events on the context's stream; blocks until it is done and returns its duration on this rank
 * (BASELINE.md section 3's "gather time"). What the stream did before -- renders still in flight -- is waited for first, so the figure
 * is the gather's own time when every rank calls it idle, after a barrier. */
int wfpt_gather_accumulated_timed(wfpt_ctx *ctx, float *ms);
int wfpt_read_gathered(wfpt_ctx *ctx, float *rgb, size_t n_floats);
int wfpt_comm_destroy(wfpt_ctx *ctx); /* also done by wfpt_destroy */

/* Queues in the reference's own layouts and in its (ascending-thread-index) order. */
int wfpt_read_rays(wfpt_ctx *ctx, wfpt_ray *rays, uint32_t n);
int wfpt_read_extension_rays(wfpt_ctx *ctx, wfpt_ray *rays, uint32_t n);
int wfpt_read_hits(wfpt_ctx *ctx, wfpt_hit_payload *hits, uint32_t n);
int wfpt_read_misses(wfpt_ctx *ctx, uint32_t *ray_indices, uint32_t n);
/* test/bench input injection: overwrite the first n rays of the current ray queue */
int wfpt_write_rays(wfpt_ctx *ctx, const wfpt_ray *rays, uint32_t n);
/* Per-bounce table of the last fused sample: rows of (rays_in, hits, misses, shaded). */
int wfpt_read_bounce_table(wfpt_ctx *ctx, uint32_t *rows4, uint32_t max_rows, uint32_t *n_rows);
/* Totals since creation over fused samples: [0] rays traced by extend, [1] hits, [2] misses. */
int wfpt_read_totals(wfpt_ctx *ctx, uint64_t totals[3]);
/* The same per wavefront: rows of (rays traced, hits, misses) summed over every fused sample since creation, one row
 * per wavefront 0 .. max_wavefronts-1 (what a bench needs to price each launch of the loop in bytes). */
int wfpt_read_wavefront_totals(wfpt_ctx *ctx, uint64_t *rows3, uint32_t max_rows, uint32_t *n_rows);
/* hipDeviceProp_t facts a report needs: CU count, memory clock (kHz) and bus width (bits), memory size. Any output
 * pointer may be NULL. Double-data-rate peak bandwidth = 2 * clock * width / 8. */
int wfpt_device_info(int device, uint32_t *compute_units, uint32_t *memory_clock_khz, uint32_t *memory_bus_width_bits,
                     uint64_t *total_memory_bytes);
/* display_shader.wgsl:50-52 tone map, sqrt(acc / n_samples) -> 8-bit RGB (host side, for image dumps) */
void wfpt_tonemap_rgb8(const float *accumulated, uint32_t n_pixels, uint32_t n_samples, uint8_t *rgb);

/* Image output, the step right after the path (the reference presents through a fullscreen pass, display.rs:112-150,
 * display_shader.wgsl:44-55; here the frame goes to a file). Both read the accumulated image of this context and
 * divide by wfpt_accumulated_samples. Rows are written top to bottom as stored (pixel_idx = x + y*width).
 *   wfpt_save_ppm: binary P6, 8-bit, sqrt(acc / n) exactly like display_shader.wgsl:50-52.
 *   wfpt_save_pfm: binary PF, linear f32 acc / n (PFM stores rows bottom-up, so rows are flipped on write).
 *   (wfpt_save_png below.) Contexts created with tile sharding hold only their own bands and are refused. */
int wfpt_save_ppm(wfpt_ctx *ctx, const char *path);
int wfpt_save_pfm(wfpt_ctx *ctx, const char *path);
/*   wfpt_save_png: the same 8-bit image as wfpt_save_ppm as a PNG (rows in deflate "stored" blocks: no compression library).
 *   wfpt_write_png_rgb8: the writer itself, for any width x height RGB8 buffer (e.g. wfpt_tonemap_rgb8 of a gathered frame). */
int wfpt_save_png(wfpt_ctx *ctx, const char *path);
int wfpt_write_png_rgb8(const char *path, const uint8_t *rgb, uint32_t width, uint32_t height);

/* ------------------------------------------------------------------ diagnostics */
/* Runs the device math primitives over arrays (op: 0 sqrt(a), 1 a/b, 2 sin(a), 3 cos(a), 4 pow(a,b),
 * 5 f32(u32 bits of a)*2^-32, 6 min(a,b), 7 max(a,b)); used by the parity tests to prove the device
 * arithmetic matches the oracle's bit for bit. */
int wfpt_selftest_math(int device, int op, const float *a, const float *b, float *out, size_t n);
/* Workgroups of the extend kernel that fit one CU when each declares `lds_bytes` of dynamic LDS (occupancy query;
 * negative status on error). Diagnostic for sizing the LDS-resident scene. */
int wfpt_debug_extend_blocks_per_cu(int device, uint32_t lds_bytes);
/* Diagnostic builds only (-DWFPT_STAMPS=1): per-phase shader-cycle sums of the middle bounce launches; zeros otherwise. */
int wfpt_debug_read_stamps(wfpt_ctx *ctx, uint64_t out[16], int reset);
/* ... and of the refill traversal's launches (scenes beyond LDS): which = 1 the first launch, 2 the middle launches (0: the call above) */
int wfpt_debug_read_stamps_ex(wfpt_ctx *ctx, int which, uint64_t out[16], int reset);
/* Host-side check of the four-wide quantised tree the device walks for scenes beyond LDS (no GPU needed): collapses
 * `nodes` and verifies that every quantised child box encloses the binary node's box and that the two trees have the
 * same leaves. counts = {four-wide nodes, depth, leaf children, inner children}. */
int wfpt_debug_bvh4(const wfpt_bvh_node *nodes, uint32_t n_nodes, uint32_t counts[4]);
/* Static facts about the built library, e.g. "gfx950;chunk=512;..." */
const char *wfpt_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* WFPT_H */
