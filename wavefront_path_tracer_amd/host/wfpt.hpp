// wfpt.hpp -- header-only C++ host mirror of the reference's API for the wavefront path, over the C ABI
// (include/wfpt.h). The reference's host is Rust; this image has no Rust toolchain, so the compiled-language
// host side is C++ with the reference's type and method names:
//
//   Scene, BVHTree                 wavefront_common/src/{scene,bvh}.rs
//   Camera, CameraController       wavefront_common/src/{camera,camera_controller}.rs
//   ProjectionMatrix               wavefront_common/src/projection_matrix.rs
//   RenderParameters, RenderProgress  wavefront_common/src/parameters.rs
//   Kernel                         gpu_wavefront_pt/src/kernel.rs
//   PathTracer                     gpu_wavefront_pt/src/path_tracer.rs
//
// Error behaviour: the reference unwraps/panics; here every negative wfpt_status becomes a wfpt::Error.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "wfpt.h"

namespace wfpt {

constexpr uint32_t SPP = 10; // wavefront_common/src/parameters.rs:4
constexpr uint32_t SPF = 1;  // wavefront_common/src/parameters.rs:5

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error("wfpt status " + std::to_string(st) + ": " + msg), status(st) {}
};

// wavefront_common/src/scene.rs
struct Scene {
    std::vector<wfpt_sphere> spheres;
    std::vector<wfpt_material> materials;

    static Scene new_scene() { // Scene::new, scene.rs:12-46
        Scene s;
        s.spheres.resize(5);
        s.materials.resize(5);
        wfpt_scene_new(s.spheres.data(), s.materials.data());
        return s;
    }
    static Scene book_one_final(uint64_t seed = 1) { // scene.rs:48-107, seeded
        Scene s;
        s.spheres.resize(512);
        s.materials.resize(512);
        const uint32_t n = wfpt_scene_book_one_final(seed, s.spheres.data(), s.materials.data(), 512);
        if (n == 0) throw Error(WFPT_ERR_INVALID_ARGUMENT, "wfpt_scene_book_one_final");
        s.spheres.resize(n);
        s.materials.resize(n);
        return s;
    }
};

// wavefront_common/src/bvh.rs:143-210
struct BVHTree {
    std::vector<wfpt_bvh_node> nodes;
    explicit BVHTree(size_t num_primitives) { nodes.reserve(2 * num_primitives); }
    float device_ms = 0.0f; // device builds: time between the first and the last build kernel
    // Reorders spheres in place (bvh.rs:182). device < 0: the host builder; otherwise the same builder on that HIP
    // device (build extension, wfpt_build_bvh_device) -- same nodes, same order, byte for byte.
    void build_bvh_tree(std::vector<wfpt_sphere> &spheres, int device = -1) {
        nodes.assign(2 * (spheres.empty() ? 1 : spheres.size()), wfpt_bvh_node{});
        uint32_t n = 0;
        const uint32_t count = static_cast<uint32_t>(spheres.size()), cap = static_cast<uint32_t>(nodes.size());
        const int st = device < 0 ? wfpt_build_bvh(spheres.data(), count, nodes.data(), cap, &n)
                                  : wfpt_build_bvh_device(spheres.data(), count, nodes.data(), cap, &n, device, &device_ms);
        if (st != WFPT_OK) throw Error(st, std::string("wfpt_build_bvh: ") + wfpt_last_error(nullptr));
        nodes.resize(n);
    }
    void build_bvh_tree_triangles(std::vector<wfpt_triangle> &triangles, uint32_t n_bins = 32, int device = -1) {
        nodes.assign(2 * (triangles.empty() ? 1 : triangles.size()), wfpt_bvh_node{});
        uint32_t n = 0;
        const uint32_t count = static_cast<uint32_t>(triangles.size()), cap = static_cast<uint32_t>(nodes.size());
        const int st = device < 0 ? wfpt_build_bvh_triangles(triangles.data(), count, nodes.data(), cap, &n, n_bins)
                                  : wfpt_build_bvh_triangles_device(triangles.data(), count, nodes.data(), cap, &n, n_bins, device, &device_ms);
        if (st != WFPT_OK) throw Error(st, std::string("wfpt_build_bvh_triangles: ") + wfpt_last_error(nullptr));
        nodes.resize(n);
    }
};

// wavefront_common/src/camera.rs
struct Camera {
    std::array<float, 3> position{};
    float pitch = 0.0f, yaw = 0.0f;
    Camera(std::array<float, 3> look_from, std::array<float, 3> look_at) : position(look_from) {
        wfpt_camera_new(look_from.data(), look_at.data(), &pitch, &yaw);
    }
    static Camera book_one_final_camera() { return Camera({13.0f, 2.0f, 3.0f}, {0.0f, 0.0f, 0.0f}); } // camera.rs:26-30
    std::array<float, 16> view_transform() const { // camera.rs:41-69
        std::array<float, 16> m{};
        wfpt_view_transform(position.data(), pitch, yaw, m.data());
        return m;
    }
};

// wavefront_common/src/camera_controller.rs:8-158
class CameraController {
  public:
    CameraController(Camera camera, float vfov, float defocus_angle, float focus_distance, float z_near, float z_far,
                     float speed = 4.0f, float sensitivity = 0.1f)
        : camera_(camera), vfov_rad_(wfpt_to_radians(vfov)), defocus_angle_rad_(wfpt_to_radians(defocus_angle)),
          focus_distance_(focus_distance), z_near_(z_near), z_far_(z_far), speed_(speed), sensitivity_(sensitivity) {}
    float vfov_rad() const { return vfov_rad_; }
    void set_vfov(float vfov) { vfov_rad_ = wfpt_to_radians(vfov); }
    std::pair<float, float> dof() const { return {defocus_angle_rad_, focus_distance_}; }
    void set_defocus_angle(float da) { defocus_angle_rad_ = wfpt_to_radians(da); }
    void set_focus_distance(float fd) { focus_distance_ = fd; }
    // camera_controller.rs:74-125: key state and pending mouse rotation
    void process_mouse(std::array<float, 2> delta) { rotate_ = delta; }
    void move_forward(uint32_t dir) { amounts_[0] = dir == 1 ? 1.0f : 0.0f; }
    void move_backwards(uint32_t dir) { amounts_[1] = dir == 1 ? 1.0f : 0.0f; }
    void move_right(uint32_t dir) { amounts_[2] = dir == 1 ? 1.0f : 0.0f; }
    void move_left(uint32_t dir) { amounts_[3] = dir == 1 ? 1.0f : 0.0f; }
    void move_up(uint32_t dir) { amounts_[4] = dir == 1 ? 1.0f : 0.0f; }
    void move_down(uint32_t dir) { amounts_[5] = dir == 1 ? 1.0f : 0.0f; }
    void update_camera(float dt) { // camera_controller.rs:125-158
        wfpt_camera_controller_update(camera_.position.data(), &camera_.pitch, &camera_.yaw, amounts_.data(), rotate_.data(),
                                      speed_, sensitivity_, dt);
    }
    std::pair<float, float> get_clip_planes() const { return {z_near_, z_far_}; }
    wfpt_gpu_camera get_GPU_camera() const { // camera_controller.rs:66-68
        wfpt_gpu_camera c{};
        wfpt_gpu_camera_new(camera_.position.data(), camera_.pitch, camera_.yaw, defocus_angle_rad_, focus_distance_, &c);
        return c;
    }
    std::array<float, 16> get_view_matrix() const { return camera_.view_transform(); }
    const Camera &camera() const { return camera_; }

  private:
    Camera camera_;
    float vfov_rad_, defocus_angle_rad_, focus_distance_, z_near_, z_far_, speed_, sensitivity_;
    std::array<float, 6> amounts_{}; // forward, backward, right, left, up, down
    std::array<float, 2> rotate_{};  // horizontal, vertical
};

// wavefront_common/src/projection_matrix.rs
struct ProjectionMatrix {
    float vfov_rad, aspect_ratio, z_near, z_far;
    ProjectionMatrix(float vfov, float ar, float zn, float zf) : vfov_rad(vfov), aspect_ratio(ar), z_near(zn), z_far(zf) {}
    std::array<float, 16> p_inv() const {
        std::array<float, 16> m{};
        wfpt_p_inv(vfov_rad, aspect_ratio, z_near, z_far, m.data());
        return m;
    }
};

// wavefront_common/src/parameters.rs:7-58
class RenderParameters {
  public:
    RenderParameters(CameraController cc, std::pair<uint32_t, uint32_t> viewport) : cc_(cc), viewport_(viewport) {}
    bool changed() const { return resized_ || camera_changed_; }
    bool resized() const { return resized_; }
    bool camera_changed() const { return camera_changed_; }
    void set_viewport(std::pair<uint32_t, uint32_t> size) { viewport_ = size; resized_ = true; }
    std::pair<uint32_t, uint32_t> viewport_size() const { return viewport_; }
    void reset() { resized_ = camera_changed_ = false; }
    const CameraController &camera_controller() const { return cc_; }
    void update_camera_controller(CameraController cc) { cc_ = cc; camera_changed_ = true; }

  private:
    CameraController cc_;
    std::pair<uint32_t, uint32_t> viewport_;
    bool resized_ = false, camera_changed_ = false;
};

// wavefront_common/src/parameters.rs:61-101
class RenderProgress {
  public:
    wfpt_frame_buffer get_next_frame(const RenderParameters &rp) {
        frame_ += 1;
        return {rp.viewport_size().first, rp.viewport_size().second, frame_, 0};
    }
    void incr_accumulated_samples(uint32_t d) { accumulated_samples_ += d; }
    void reset() { accumulated_samples_ = 0; frame_ = 0; }
    float progress() const { return static_cast<float>(accumulated_samples_) / static_cast<float>(SPP); }
    uint32_t accumulated_samples() const { return accumulated_samples_; }
    uint32_t frame() const { return frame_; }

  private:
    uint32_t frame_ = 0, accumulated_samples_ = 0;
};

inline std::pair<uint32_t, uint32_t> workgroup_size_64(uint32_t x) { // path_tracer.rs:282-289
    uint32_t gx = 0, gy = 0;
    wfpt_workgroup_size_64(x, &gx, &gy);
    return {gx, gy};
}

class PathTracer;

// gpu_wavefront_pt/src/kernel.rs
class Kernel {
  public:
    Kernel() = default;
    Kernel(const char *name, wfpt_ctx *ctx) : ctx_(ctx), stage_(wfpt_stage_from_name(name)) {
        if (stage_ < 0 || stage_ >= WFPT_STAGE_SCAN) throw Error(WFPT_ERR_INVALID_ARGUMENT, std::string("no such kernel stage: ") + name);
    }
    void run(std::pair<uint32_t, uint32_t> workgroup_size) { // kernel.rs:107-140
        const int st = wfpt_kernel_run(ctx_, stage_, workgroup_size.first, workgroup_size.second);
        if (st != WFPT_OK) throw Error(st, wfpt_last_error(ctx_));
    }
    float get_timing() { return wfpt_kernel_timing_us(ctx_, stage_); } // kernel.rs:142-146, microseconds

  private:
    wfpt_ctx *ctx_ = nullptr;
    int stage_ = -1;
};

// gpu_wavefront_pt/src/path_tracer.rs
class PathTracer {
  public:
    struct Options {
        uint32_t max_window_size = 0; // path_tracer.rs:44
        uint32_t max_wavefronts = 50; // path_tracer.rs:323
        uint32_t miss_floor = 128;    // path_tracer.rs:332
        uint32_t rng_mode = WFPT_RNG_DISPATCH;
        uint32_t flags = 0, tile_rank = 0, tile_world = 1;
        int32_t device = 0;
        uint32_t spp = SPP;
        uint32_t batch = 0; // samples in flight per launch of render(); 0 = library default
    };

    // PathTracer::new (path_tracer.rs:43-217): builds the BVH (reordering scene.spheres), uploads everything.
    PathTracer(Scene &scene, const RenderParameters &rp, const Options &opt)
        : render_parameters_(rp), opt_(opt), bvh_tree_(scene.spheres.size()) {
        bvh_tree_.build_bvh_tree(scene.spheres); // :117-118
        const CameraController &cc = rp.camera_controller();
        const auto [w, h] = rp.viewport_size();
        const float ar = static_cast<float>(w) / static_cast<float>(h);
        const auto [zn, zf] = cc.get_clip_planes();
        const auto proj = ProjectionMatrix(cc.vfov_rad(), ar, zn, zf).p_inv(); // :137-138
        const auto view = cc.get_view_matrix();
        const wfpt_gpu_camera cam = cc.get_GPU_camera();
        wfpt_params p{};
        p.width = w; p.height = h; p.max_pixels = opt.max_window_size;
        p.max_wavefronts = opt.max_wavefronts; p.miss_floor = opt.miss_floor;
        p.rng_mode = opt.rng_mode; p.flags = opt.flags;
        p.tile_rank = opt.tile_rank; p.tile_world = opt.tile_world; p.device = opt.device; p.batch = opt.batch;
        ctx_ = wfpt_create(&p, scene.spheres.data(), static_cast<uint32_t>(scene.spheres.size()), scene.materials.data(),
                           static_cast<uint32_t>(scene.materials.size()), bvh_tree_.nodes.data(),
                           static_cast<uint32_t>(bvh_tree_.nodes.size()), &cam, proj.data(), view.data());
        if (!ctx_) throw Error(WFPT_ERR_HIP, wfpt_last_error(nullptr));
        generate_ray_kernel_ = Kernel("generate_rays", ctx_); // :162
        extend_kernel_ = Kernel("extend", ctx_);              // :167
        shade_kernel_ = Kernel("shade", ctx_);                // :175
        miss_kernel_ = Kernel("miss_kernel", ctx_);           // :180
        accumulate_kernel_ = Kernel("accumulate", ctx_);      // :185
    }
    ~PathTracer() { wfpt_destroy(ctx_); }
    PathTracer(const PathTracer &) = delete;
    PathTracer &operator=(const PathTracer &) = delete;

    float progress() const { return render_progress_.progress(); }                    // :219-221
    RenderParameters get_render_parameters() const { return render_parameters_; }     // :227-229
    void update_render_parameters(const RenderParameters &rp) { render_parameters_ = rp; } // :236-238
    void resize(const RenderParameters &rp) { update_render_parameters(rp); }         // :231-235

    void update_buffers() { // :240-277
        if (!render_parameters_.changed()) return;
        const CameraController &cc = render_parameters_.camera_controller();
        const auto [w, h] = render_parameters_.viewport_size();
        const auto [zn, zf] = cc.get_clip_planes();
        const auto proj = ProjectionMatrix(cc.vfov_rad(), static_cast<float>(w) / static_cast<float>(h), zn, zf).p_inv();
        const auto view = cc.get_view_matrix();
        const wfpt_gpu_camera cam = cc.get_GPU_camera();
        check(wfpt_update_render_parameters(ctx_, w, h, &cam, proj.data(), view.data()));
        render_parameters_.reset();
        render_progress_.reset();
    }

    // PathTracer::run (path_tracer.rs:279-371): the host-driven loop over the five Kernels.
    void run() {
        update_buffers();
        if (render_progress_.accumulated_samples() < opt_.spp) {
            wfpt_frame_buffer frame = render_progress_.get_next_frame(render_parameters_);
            for (uint32_t sample_number = 0; sample_number < SPF; ++sample_number) {
                frame.sample_number = sample_number;
                check(wfpt_set_frame(ctx_, &frame));                           // :296-297
                check(wfpt_reset_image(ctx_));                                 // :305-306
                check(wfpt_clear_ray_queues(ctx_));                            // :309-310
                const auto [width, height] = render_parameters_.viewport_size();
                uint32_t counter[16] = {0};
                counter[2] = width * height;                                   // :313-316
                check(wfpt_set_counters(ctx_, counter));
                generate_ray_kernel_.run({width / 8, height / 8});             // :318
                uint32_t wavefront = 0;
                auto extend_size = workgroup_size_64(width * height);          // :322
                while (wavefront < opt_.max_wavefronts) {                      // :323
                    extend_kernel_.run(extend_size);                           // :325
                    check(wfpt_read_counters(ctx_, counter));                  // :327-328
                    const uint32_t num_misses = counter[0], num_hits = counter[1];
                    if (num_misses < opt_.miss_floor) break;                   // :332
                    counter[2] = 0;                                            // :335-336
                    check(wfpt_set_counters(ctx_, counter));
                    shade_kernel_.run(workgroup_size_64(num_hits));            // :339
                    miss_kernel_.run(workgroup_size_64(num_misses));           // :340
                    check(wfpt_read_counters(ctx_, counter));                  // :343-345
                    const uint32_t num_extension = counter[2];
                    check(wfpt_swap_ray_queues(ctx_));                         // :348
                    extend_size = workgroup_size_64(num_extension);            // :350
                    const uint32_t next[16] = {0, 0, num_extension, 0};        // :352
                    check(wfpt_set_counters(ctx_, next));
                    wavefront += 1;
                }
                last_wavefronts_ = wavefront;
                accumulate_kernel_.run(workgroup_size_64(width * height));     // :362
                render_progress_.incr_accumulated_samples(1);                  // :363
                frame.sample_number = render_progress_.accumulated_samples();  // :366-367
                check(wfpt_set_frame(ctx_, &frame));
            }
        }
    }

    // The same loop resident on the device (no host synchronisation).
    void render(uint32_t spp) { check(wfpt_render(ctx_, spp)); }

    std::vector<float> accumulated() {
        std::vector<float> a(3 * static_cast<size_t>(wfpt_n_pixels(ctx_)));
        check(wfpt_read_accumulated(ctx_, a.data(), a.size()));
        return a;
    }
    // Multi-GPU (build-side addition, include/wfpt.h): this context was created with Options::tile_rank / tile_world;
    // rank 0 makes the 128-byte id with wfpt::comm_unique_id() and hands it to every rank.
    void comm_init(const std::array<uint8_t, WFPT_COMM_UNIQUE_ID_BYTES> &id, int rank, int world) { check(wfpt_comm_init(ctx_, id.data(), rank, world)); }
    void gather_accumulated() { check(wfpt_gather_accumulated(ctx_)); } // peers -> rank 0 over xGMI, asynchronous
    std::vector<float> gathered(uint32_t width, uint32_t height) {     // rank 0: the assembled frame
        std::vector<float> a(3 * static_cast<size_t>(width) * height);
        check(wfpt_read_gathered(ctx_, a.data(), a.size()));
        return a;
    }
    uint32_t last_wavefronts() const { return last_wavefronts_; }
    wfpt_ctx *handle() { return ctx_; }
    Kernel &generate_ray_kernel() { return generate_ray_kernel_; }
    Kernel &extend_kernel() { return extend_kernel_; }
    Kernel &shade_kernel() { return shade_kernel_; }
    Kernel &miss_kernel() { return miss_kernel_; }
    Kernel &accumulate_kernel() { return accumulate_kernel_; }

  private:
    void check(int st) const {
        if (st != WFPT_OK) throw Error(st, wfpt_last_error(ctx_));
    }
    wfpt_ctx *ctx_ = nullptr;
    RenderParameters render_parameters_;
    RenderProgress render_progress_;
    Options opt_;
    BVHTree bvh_tree_;
    Kernel generate_ray_kernel_, extend_kernel_, shade_kernel_, miss_kernel_, accumulate_kernel_;
    uint32_t last_wavefronts_ = 0;
};

inline std::array<uint8_t, WFPT_COMM_UNIQUE_ID_BYTES> comm_unique_id() {
    std::array<uint8_t, WFPT_COMM_UNIQUE_ID_BYTES> id{};
    const int st = wfpt_comm_unique_id(id.data());
    if (st != WFPT_OK) throw Error(st, wfpt_last_error(nullptr));
    return id;
}

} // namespace wfpt
