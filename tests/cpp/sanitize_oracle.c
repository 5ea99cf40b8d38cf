/* Drives the oracle (small renders of both scenes, both RNG modes, a size with padding rays, a mesh) under
 * AddressSanitizer + UBSan (tests/test_sanitizers.py). */
#include "wfpt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void) {
    static orc_sphere sp[512]; static orc_material mt[512];
    uint32_t n = orc_scene_book_one_final(1, sp, mt);
    orc_bvh_node *nodes = calloc(2 * n, sizeof *nodes);
    uint32_t nn = orc_build_bvh(sp, n, nodes);
    float from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, pitch, yaw, view[16], ip[16];
    orc_gpu_camera cam;
    orc_camera_new(from, at, &pitch, &yaw); orc_view_transform(from, pitch, yaw, view);
    for (int mode = 0; mode < 2; ++mode) for (int odd = 0; odd < 2; ++odd) {
        uint32_t w = 72, h = odd ? 45 : 40;
        orc_p_inv(orc_to_radians(20.0f), (float)w / (float)h, 0.1f, 100.0f, ip);
        orc_gpu_camera_new(from, pitch, yaw, orc_to_radians(0.6f), 10.0f, &cam);
        orc_params p; memset(&p, 0, sizeof p);
        p.width = w; p.height = h; p.max_wavefronts = 50; p.miss_floor = 128; p.rng_mode = mode; p.tile_rank = 0; p.tile_world = 1;
        orc_ctx *c = orc_create(&p, sp, n, mt, n, nodes, nn, &cam, ip, view);
        for (int s = 0; s < 3; ++s) orc_render_sample(c);
        uint64_t t[3]; orc_totals(c, t);
        printf("mode %d %ux%u rays %llu\n", mode, w, h, (unsigned long long)t[0]);
        orc_destroy(c);
    }
    /* mesh */
    uint32_t nt = 3000; orc_triangle *tr = calloc(nt, sizeof *tr); orc_material m3[3];
    orc_scene_random_mesh(1, nt, tr, m3);
    for (uint32_t i = 0; i < nt; ++i) for (int k = 0; k < 3; ++k) { tr[i].e1[k] *= 8.0f; tr[i].e2[k] *= 8.0f; }
    orc_bvh_node *mn = calloc(2 * nt, sizeof *mn);
    uint32_t mnn = orc_build_bvh_triangles(tr, nt, mn, 32);
    float mf[3] = {0, 0, 30};
    orc_camera_new(mf, at, &pitch, &yaw); orc_view_transform(mf, pitch, yaw, view);
    orc_p_inv(orc_to_radians(40.0f), 80.0f / 48.0f, 0.1f, 100.0f, ip);
    orc_gpu_camera_new(mf, pitch, yaw, 0.0f, 10.0f, &cam);
    orc_params p; memset(&p, 0, sizeof p); p.width = 80; p.height = 48; p.max_wavefronts = 6; p.miss_floor = 128; p.tile_world = 1;
    orc_ctx *c = orc_create_mesh(&p, tr, nt, m3, 3, mn, mnn, &cam, ip, view);
    orc_render_sample(c); orc_render_sample(c);
    uint64_t t[3]; orc_totals(c, t); printf("mesh rays %llu\n", (unsigned long long)t[0]);
    orc_destroy(c);
    free(nodes); free(tr); free(mn);
    puts("ok");
    return 0;
}
