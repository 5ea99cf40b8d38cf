// Exercises the C++ host mirror (wavefront_path_tracer_amd/host/wfpt.hpp) the way the reference's main.rs +
// PathTracer are used (gpu_wavefront_pt/src/main.rs:17-36, path_tracer.rs:279-371).
//   host_mirror model <out.bin>                         scene + BVH + camera bytes (no GPU needed)
//   host_mirror controller -                            interactive camera controller + change flags
//   host_mirror run <w> <h> <spp> <bounces> <out.bin>   host-driven PathTracer::run() x spp
//   host_mirror render <w> <h> <spp> <bounces> <out.bin> device-resident loop
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "wfpt.hpp"

static void put(FILE *f, const void *p, size_t n) { fwrite(p, 1, n, f); }

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    try {
        wfpt::Scene scene = wfpt::Scene::book_one_final(1);                     // main.rs:20 (seeded)
        wfpt::Camera camera = wfpt::Camera::book_one_final_camera();            // main.rs:23
        wfpt::CameraController cc(camera, 20.0f, 0.6f, 10.0f, 0.1f, 100.0f, 4.0f, 0.1f); // main.rs:24-32
        if (mode == "model") {
            wfpt::BVHTree tree(scene.spheres.size());
            tree.build_bvh_tree(scene.spheres);
            const auto view = cc.get_view_matrix();
            const auto proj = wfpt::ProjectionMatrix(cc.vfov_rad(), 1920.0f / 1080.0f, 0.1f, 100.0f).p_inv();
            const wfpt_gpu_camera cam = cc.get_GPU_camera();
            FILE *f = fopen(argv[2], "wb");
            put(f, scene.spheres.data(), scene.spheres.size() * sizeof(wfpt_sphere));
            put(f, scene.materials.data(), scene.materials.size() * sizeof(wfpt_material));
            put(f, tree.nodes.data(), tree.nodes.size() * sizeof(wfpt_bvh_node));
            put(f, &cam, sizeof cam);
            put(f, proj.data(), 64);
            put(f, view.data(), 64);
            fclose(f);
            const auto g = wfpt::workgroup_size_64(2073600);
            printf("%zu %zu %u %u\n", scene.spheres.size(), tree.nodes.size(), g.first, g.second);
            return 0;
        }
        if (mode == "controller") { // camera_controller.rs:74-158 + parameters.rs:51-57, printed as raw f32 bits
            wfpt::RenderParameters rp(cc, {640u, 360u});
            wfpt::CameraController moved = rp.camera_controller();
            moved.move_forward(1); moved.move_left(1); moved.move_up(1);
            moved.process_mouse({3.0f, -2.0f});
            moved.update_camera(0.25f);
            moved.set_vfov(40.0f);
            rp.update_camera_controller(moved);
            const wfpt::Camera &c = rp.camera_controller().camera();
            const float v[6] = {c.position[0], c.position[1], c.position[2], c.pitch, c.yaw, rp.camera_controller().vfov_rad()};
            uint32_t u[6];
            memcpy(u, v, sizeof v);
            printf("%08x %08x %08x %08x %08x %08x %d %d\n", u[0], u[1], u[2], u[3], u[4], u[5], rp.camera_changed() ? 1 : 0,
                   rp.resized() ? 1 : 0);
            return 0;
        }
        if (argc < 7) return 2;
        const uint32_t w = atoi(argv[2]), h = atoi(argv[3]), spp = atoi(argv[4]), bounces = atoi(argv[5]);
        wfpt::RenderParameters rp(cc, {w, h});
        wfpt::PathTracer::Options opt;
        opt.max_wavefronts = bounces;
        opt.spp = spp;
        wfpt::PathTracer pt(scene, rp, opt);
        if (mode == "run") {
            for (uint32_t s = 0; s < spp; ++s) pt.run();
            printf("progress %.3f timing_us %.2f\n", pt.progress(), pt.extend_kernel().get_timing());
        } else {
            pt.render(spp);
        }
        const auto acc = pt.accumulated();
        FILE *f = fopen(argv[6], "wb");
        put(f, acc.data(), acc.size() * sizeof(float));
        fclose(f);
        return 0;
    } catch (const wfpt::Error &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
