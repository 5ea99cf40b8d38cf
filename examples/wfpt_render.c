/* examples/wfpt_render.c -- the whole path through the C ABI (include/wfpt.h) from plain C99, the way the reference's
 * main.rs + PathTracer::new + PathTracer::run use it (gpu_wavefront_pt/src/main.rs:17-36, path_tracer.rs:43-371).
 *
 *   gcc -std=c99 -O1 -Iinclude examples/wfpt_render.c -Lwavefront_path_tracer_amd -lwfpt \
 *       -Wl,-rpath,$PWD/wavefront_path_tracer_amd -lm -o wfpt_render
 *   ./wfpt_render out.ppm [width height spp bounces [device_bvh]]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wfpt.h"

#define CHECK(call)                                                             \
    do {                                                                        \
        int st_ = (call);                                                       \
        if (st_ != WFPT_OK) {                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #call, st_, wfpt_last_error(ctx)); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char **argv) {
    wfpt_ctx *ctx = NULL;
    if (argc < 2) {
        fprintf(stderr, "usage: %s out.ppm [width height spp bounces [device_bvh]]\n", argv[0]);
        return 2;
    }
    const uint32_t width = argc > 3 ? (uint32_t)atoi(argv[2]) : 400, height = argc > 3 ? (uint32_t)atoi(argv[3]) : 225;
    const uint32_t spp = argc > 4 ? (uint32_t)atoi(argv[4]) : 8, bounces = argc > 5 ? (uint32_t)atoi(argv[5]) : 8;
    const int device_bvh = argc > 6 ? atoi(argv[6]) : 0;

    /* main.rs:20: the book's final scene (seeded here), path_tracer.rs:117-118: its BVH */
    wfpt_sphere spheres[512];
    wfpt_material materials[512];
    const uint32_t n = wfpt_scene_book_one_final(1, spheres, materials, 512);
    wfpt_bvh_node *nodes = (wfpt_bvh_node *)calloc(2u * n, sizeof *nodes);
    uint32_t n_nodes = 0;
    if (device_bvh) CHECK(wfpt_build_bvh_device(spheres, n, nodes, 2u * n, &n_nodes, 0, NULL)); /* build extension */
    else CHECK(wfpt_build_bvh(spheres, n, nodes, 2u * n, &n_nodes));

    /* main.rs:23-32: camera (13,2,3) -> origin, vfov 20, defocus angle 0.6, focus distance 10 */
    const float look_from[3] = {13.0f, 2.0f, 3.0f}, look_at[3] = {0.0f, 0.0f, 0.0f};
    float pitch, yaw, view[16], inv_proj[16];
    wfpt_gpu_camera camera;
    wfpt_camera_new(look_from, look_at, &pitch, &yaw);
    wfpt_view_transform(look_from, pitch, yaw, view);
    wfpt_p_inv(wfpt_to_radians(20.0f), (float)width / (float)height, 0.1f, 100.0f, inv_proj); /* path_tracer.rs:135-138 */
    wfpt_gpu_camera_new(look_from, pitch, yaw, wfpt_to_radians(0.6f), 10.0f, &camera);

    wfpt_params params;
    memset(&params, 0, sizeof params);
    params.width = width;
    params.height = height;
    params.max_wavefronts = bounces; /* path_tracer.rs:323 uses 50 */
    params.miss_floor = 128;         /* path_tracer.rs:332 */
    ctx = wfpt_create(&params, spheres, n, materials, n, nodes, n_nodes, &camera, inv_proj, view);
    if (!ctx) {
        fprintf(stderr, "wfpt_create: %s\n", wfpt_last_error(NULL));
        return 1;
    }
    CHECK(wfpt_render(ctx, spp)); /* the loop of path_tracer.rs:291-368, resident on the device */
    CHECK(wfpt_save_ppm(ctx, argv[1]));
    uint64_t totals[4];
    CHECK(wfpt_read_totals(ctx, totals));
    printf("%ux%u, %u spp, %u bounces: %llu rays, %u BVH nodes%s -> %s\n", width, height, spp, bounces,
           (unsigned long long)totals[0], n_nodes, device_bvh ? " (built on the device)" : "", argv[1]);
    wfpt_destroy(ctx);
    free(nodes);
    return 0;
}
