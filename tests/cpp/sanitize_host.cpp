// Drives the product's host code (wfpt_host.cpp: scenes, BVH builders, camera, OBJ reader, dispatch sizing) under
// AddressSanitizer + UBSan (tests/test_sanitizers.py). No GPU code is linked.
#include "wfpt.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    // scenes + sphere BVH
    std::vector<wfpt_sphere> sp(512); std::vector<wfpt_material> mt(512);
    for (uint64_t seed = 1; seed < 4; ++seed) {
        uint32_t n = wfpt_scene_book_one_final(seed, sp.data(), mt.data(), 512);
        std::vector<wfpt_bvh_node> nodes(2 * n); uint32_t nn = 0;
        if (wfpt_build_bvh(sp.data(), n, nodes.data(), 2 * n, &nn) != 0) return 1;
        printf("spheres %u nodes %u\n", n, nn);
    }
    // meshes with various bins, incl. tiny and degenerate
    for (uint32_t n : {1u, 2u, 3u, 65u, 1000u, 20000u}) for (uint32_t bins : {2u, 32u, 100u}) {
        std::vector<wfpt_triangle> t(n); wfpt_material m3[3];
        wfpt_scene_random_mesh(n, n, t.data(), m3);
        if (n == 1000) for (auto& x : t) x = t[0];
        std::vector<wfpt_bvh_node> nodes(2 * n); uint32_t nn = 0;
        if (wfpt_build_bvh_triangles(t.data(), n, nodes.data(), 2 * n, &nn, bins) != 0) return 2;
    }
    // camera, projection, misc
    float from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, pitch, yaw, view[16], ip[16]; wfpt_gpu_camera cam;
    wfpt_camera_new(from, at, &pitch, &yaw); wfpt_view_transform(from, pitch, yaw, view); wfpt_p_inv(0.3f, 1.7f, 0.1f, 100.f, ip);
    wfpt_gpu_camera_new(from, pitch, yaw, 0.01f, 10.f, &cam);
    float amounts[6] = {1, 0, 0, 1, 1, 0}, rot[2] = {3, -2};
    wfpt_camera_controller_update(from, &pitch, &yaw, amounts, rot, 4.f, 0.1f, 0.25f);
    for (uint32_t x : {0u, 1u, 64u, 65u, 4096u, 2073600u, 0xffffffffu}) { uint32_t gx, gy; wfpt_workgroup_size_64(x, &gx, &gy); }
    if (argc > 1) { uint32_t cnt = 0; int st = wfpt_load_obj(argv[1], nullptr, 0, &cnt, 0, 0); std::vector<wfpt_triangle> t(cnt ? cnt : 1); st = wfpt_load_obj(argv[1], t.data(), cnt, &cnt, 0, 0); printf("obj %d %u\n", st, cnt); }
    uint8_t rgb[6]; float acc[6] = {0, 4, 16, 1, 100, 2.25f}; wfpt_tonemap_rgb8(acc, 2, 4, rgb);
    puts("ok");
    return 0;
}
