"""ctypes binding of the ORACLE (oracle/wfpt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (wavefront_path_tracer_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

SPHERE = np.dtype([("center", "<f4", 4), ("radius", "<f4"), ("material_idx", "<u4"),
                   ("material_type", "<u4"), ("_buffer", "<u4")])
MATERIAL = np.dtype([("albedo", "<f4", 4), ("fuzz", "<f4"), ("refract_index", "<f4"),
                     ("material_type", "<u4"), ("_buffer", "<u4")])
BVH_NODE = np.dtype([("aabb_min", "<f4", 3), ("left_first", "<u4"), ("aabb_max", "<f4", 3),
                     ("prim_count", "<u4")])
GPU_CAMERA = np.dtype([("position", "<f4", 4), ("pitch", "<f4"), ("yaw", "<f4"),
                       ("defocus_radius", "<f4"), ("focus_distance", "<f4")])
RAY = np.dtype([("origin", "<f4", 4), ("direction", "<f4", 4), ("inv_direction", "<f4", 3),
                ("pixel_idx", "<u4")])
HIT = np.dtype([("t", "<f4"), ("ray_idx", "<u4"), ("sphere_idx", "<u4"), ("mat_type", "<u4")])
TRIANGLE = np.dtype([("v0", "<f4", 3), ("material_idx", "<u4"), ("e1", "<f4", 3), ("material_type", "<u4"),
                     ("e2", "<f4", 3), ("_pad", "<u4")])
assert SPHERE.itemsize == 32 and MATERIAL.itemsize == 32 and BVH_NODE.itemsize == 32 and TRIANGLE.itemsize == 48
assert GPU_CAMERA.itemsize == 32 and RAY.itemsize == 48 and HIT.itemsize == 16

RNG_DISPATCH, RNG_PIXEL = 0, 1
INACTIVE_PIXEL = 0xFFFFFFFF


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_wavefronts", C.c_uint32),
                ("miss_floor", C.c_uint32), ("rng_mode", C.c_uint32), ("tile_rank", C.c_uint32),
                ("tile_world", C.c_uint32)]


class FrameBuffer(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("frame", C.c_uint32),
                ("sample_number", C.c_uint32)]


def build(force=False):
    """Compile the oracle with its committed recipe (oracle/Makefile)."""
    so = os.path.join(_BUILD, "libwfpt_oracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("wfpt_oracle.c", "wfpt_oracle.h", "orc_math.h", "Makefile")):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return so


_libs = {}


def _limit_openmp():
    """A GPU box shows every host CPU but the job owns a share of them (16 per GPU): an OpenMP team of 128
    spin-waiting threads on 16 cores makes small parallel regions pathologically slow. Must run before libgomp loads."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(avail, 16))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def cpu_share():
    """The CPUs this job may really use: the affinity mask and, where the job runs under one, the cgroup CPU quota (v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`). `granted` = min(quota, affinity) threads; without a quota the affinity mask is all there is
    to go by (bench.py then times the CPU baseline at 16 threads AND at the whole mask)."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota, source = None, "no cgroup CPU quota found"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        source = f"/sys/fs/cgroup/cpu.max = '{q} {per}'"
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            source = f"cpu.cfs_quota_us / cpu.cfs_period_us = {q} / {per}"
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    granted = affinity if quota is None else max(1, min(affinity, int(quota + 0.999)))
    return {"affinity": affinity, "host_cpus": os.cpu_count() or affinity, "quota_cores": quota, "quota_source": source, "granted": granted}


def set_num_threads(n):
    """OpenMP threads of the following oracle calls (the results do not depend on it)."""
    lib().orc_set_num_threads(int(n))


def lib(serial=False):
    key = "serial" if serial else "omp"
    if key in _libs:
        return _libs[key]
    so = os.path.join(_BUILD, "libwfpt_oracle_serial.so" if serial else "libwfpt_oracle.so")
    if not os.path.exists(so):
        build()
    _limit_openmp()
    L = C.CDLL(so)
    vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
    L.orc_create.restype = vp
    L.orc_create.argtypes = [C.POINTER(Params), vp, u32, vp, u32, vp, u32, vp, vp, vp]
    L.orc_destroy.argtypes = [vp]
    L.orc_scene_new.restype = u32
    L.orc_scene_new.argtypes = [vp, vp]
    L.orc_scene_book_one_final.restype = u32
    L.orc_scene_book_one_final.argtypes = [C.c_uint64, vp, vp]
    L.orc_build_bvh.restype = u32
    L.orc_build_bvh.argtypes = [vp, u32, vp]
    L.orc_build_bvh_triangles.restype = u32
    L.orc_build_bvh_triangles.argtypes = [vp, u32, vp, u32]
    L.orc_scene_random_mesh.restype = u32
    L.orc_scene_random_mesh.argtypes = [C.c_uint64, u32, vp, vp]
    L.orc_create_mesh.restype = vp
    L.orc_create_mesh.argtypes = [C.POINTER(Params), vp, u32, vp, u32, vp, u32, vp, vp, vp]
    L.orc_ray_steps.argtypes = [vp, u32, vp, vp]
    L.orc_traversal_profile.argtypes = [vp, u32, vp]
    L.orc_ray_rounds.argtypes = [vp, u32, vp, vp]
    L.orc_sim_postpone.argtypes = [vp, u32, u32, vp]
    L.orc_prim_hit_t.argtypes = [vp, u32, u32, vp]
    L.orc_ray_rounds_init.argtypes = [vp, u32, vp, C.c_int, vp, vp]
    L.orc_write_rays.argtypes = [vp, vp, u32]
    L.orc_camera_new.argtypes = [vp, vp, C.POINTER(f32), C.POINTER(f32)]
    L.orc_view_transform.argtypes = [vp, f32, f32, vp]
    L.orc_p_inv.argtypes = [f32, f32, f32, f32, vp]
    L.orc_gpu_camera_new.argtypes = [vp, f32, f32, f32, f32, vp]
    L.orc_to_radians.restype = f32
    L.orc_to_radians.argtypes = [f32]
    L.orc_workgroup_size_64.argtypes = [u32, C.POINTER(u32), C.POINTER(u32)]
    L.orc_probe_jenkins.restype = u32
    L.orc_probe_jenkins.argtypes = [u32]
    L.orc_probe_init_rng.restype = u32
    L.orc_probe_init_rng.argtypes = [u32, u32, u32, u32]
    L.orc_probe_next_int.restype = u32
    L.orc_probe_next_int.argtypes = [C.POINTER(u32)]
    L.orc_probe_next_float.restype = f32
    L.orc_probe_next_float.argtypes = [C.POINTER(u32)]
    L.orc_probe_advance.restype = u32
    L.orc_probe_advance.argtypes = [u32, u32]
    L.orc_probe_sincos.argtypes = [vp, vp, vp, C.c_size_t]
    L.orc_probe_pow.argtypes = [vp, vp, vp, C.c_size_t]
    L.orc_set_frame.argtypes = [vp, C.POINTER(FrameBuffer)]
    L.orc_set_counters.argtypes = [vp, vp]
    L.orc_get_counters.argtypes = [vp, vp]
    for name in ("orc_reset_image", "orc_reset_accumulated", "orc_swap_ray_queues"):
        getattr(L, name).argtypes = [vp]
    L.orc_generate_rays.argtypes = [vp, u32, u32, C.c_int]
    for name in ("orc_extend", "orc_shade", "orc_miss", "orc_accumulate"):
        getattr(L, name).argtypes = [vp, u32, u32]
    L.orc_render_sample.restype = u32
    L.orc_render_sample.argtypes = [vp]
    for name in ("orc_frame", "orc_accumulated_samples", "orc_n_pixels"):
        getattr(L, name).restype = u32
        getattr(L, name).argtypes = [vp]
    for name in ("orc_rays", "orc_extension_rays", "orc_hits", "orc_misses", "orc_image", "orc_accumulated"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    L.orc_bounce_table.restype = u32
    L.orc_bounce_table.argtypes = [vp, vp, u32]
    L.orc_totals.argtypes = [vp, vp]
    L.orc_trace_stats.argtypes = [vp, vp]
    L.orc_trace_brute.restype = C.c_int
    L.orc_trace_brute.argtypes = [vp, vp, vp]
    L.orc_trace_bvh.restype = C.c_int
    L.orc_trace_bvh.argtypes = [vp, vp, vp]
    L.orc_tonemap_rgb8.argtypes = [vp, u32, u32, vp]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    _libs[key] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- host-side inputs
def scene_new(serial=False):
    sp, mt = np.zeros(5, SPHERE), np.zeros(5, MATERIAL)
    n = lib(serial).orc_scene_new(_p(sp), _p(mt))
    return sp[:n].copy(), mt[:n].copy()


def scene_book_one_final(seed=1, serial=False):
    sp, mt = np.zeros(512, SPHERE), np.zeros(512, MATERIAL)
    n = lib(serial).orc_scene_book_one_final(seed, _p(sp), _p(mt))
    return sp[:n].copy(), mt[:n].copy()


def build_bvh(spheres, serial=False):
    """Returns (reordered spheres, nodes) -- bvh.rs reorders the sphere array in place."""
    sp = np.ascontiguousarray(spheres).copy()
    nodes = np.zeros(2 * max(len(sp), 1) + 2, BVH_NODE)
    n = lib(serial).orc_build_bvh(_p(sp), len(sp), _p(nodes))
    return sp, nodes[:n].copy()


def scene_random_mesh(n, seed=1):
    """BASELINE config 5's triangle soup: (triangles, materials)."""
    tris, mt = np.zeros(n, TRIANGLE), np.zeros(3, MATERIAL)
    lib().orc_scene_random_mesh(seed, n, _p(tris), _p(mt))
    return tris, mt


def build_bvh_triangles(triangles, n_bins=32):
    """Returns (reordered triangles, nodes)."""
    tr = np.ascontiguousarray(triangles).copy()
    nodes = np.zeros(2 * max(len(tr), 1) + 2, BVH_NODE)
    n = lib().orc_build_bvh_triangles(_p(tr), len(tr), _p(nodes), n_bins)
    return tr, nodes[:n].copy()


def mesh_camera(width, height):
    """camera at (0,0,30) looking at the origin, vfov 40, no defocus."""
    return camera((0.0, 0.0, 30.0), (0.0, 0.0, 0.0), 40.0, 0.0, 10.0, 0.1, 100.0, width, height)


def camera(look_from, look_at, vfov_deg, defocus_deg, focus_dist, z_near, z_far, width, height):
    """main.rs:23-32 + path_tracer.rs:132-156: returns (gpu_camera, inv_proj[16], view[16])."""
    L = lib()
    lf = np.asarray(look_from, "<f4")
    la = np.asarray(look_at, "<f4")
    pitch, yaw = C.c_float(), C.c_float()
    L.orc_camera_new(_p(lf), _p(la), C.byref(pitch), C.byref(yaw))
    view = np.zeros(16, "<f4")
    L.orc_view_transform(_p(lf), pitch, yaw, _p(view))
    inv_proj = np.zeros(16, "<f4")
    aspect = np.float32(width) / np.float32(height)
    L.orc_p_inv(L.orc_to_radians(vfov_deg), aspect, z_near, z_far, _p(inv_proj))
    cam = np.zeros(1, GPU_CAMERA)
    L.orc_gpu_camera_new(_p(lf), pitch, yaw, L.orc_to_radians(defocus_deg), focus_dist, _p(cam))
    return cam, inv_proj, view


def shirley_camera(width, height):
    """main.rs:23-32: look_from (13,2,3) -> origin, vfov 20, defocus 0.6, focus 10, z 0.1..100."""
    return camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, 0.6, 10.0, 0.1, 100.0, width, height)


def workgroup_size_64(x):
    gx, gy = C.c_uint32(), C.c_uint32()
    lib().orc_workgroup_size_64(x, C.byref(gx), C.byref(gy))
    return gx.value, gy.value


# ---------------------------------------------------------------- the chain
class Oracle:
    def __init__(self, width, height, spheres, materials, nodes, cam, inv_proj, view,
                 max_wavefronts=50, miss_floor=128, rng_mode=RNG_DISPATCH, tile_rank=0, tile_world=1,
                 serial=False, triangles=None):
        self.L = lib(serial)
        self.width, self.height = width, height
        self.params = Params(width, height, max_wavefronts, miss_floor, rng_mode, tile_rank, tile_world)
        self._keep = [np.ascontiguousarray(a) for a in (spheres, materials, nodes, cam, inv_proj, view)]
        sp, mt, nd, cm, ip, vw = self._keep
        if triangles is not None:
            tr = np.ascontiguousarray(triangles, TRIANGLE)
            self.h = self.L.orc_create_mesh(C.byref(self.params), _p(tr), len(tr), _p(mt), len(mt), _p(nd), len(nd),
                                            _p(cm), _p(ip), _p(vw))
        else:
            self.h = self.L.orc_create(C.byref(self.params), _p(sp), len(sp), _p(mt), len(mt), _p(nd), len(nd),
                                       _p(cm), _p(ip), _p(vw))
        self.n_pixels = self.L.orc_n_pixels(self.h)
        self.n_slots = max(self.n_pixels, ((width + 7) // 8) * 64 *
                           (((height + 7) // 8 - tile_rank + tile_world - 1) // tile_world))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_frame(self, frame, sample_number=0):
        f = FrameBuffer(self.width, self.height, frame, sample_number)
        self.L.orc_set_frame(self.h, C.byref(f))

    def set_counters(self, c):
        a = np.zeros(16, "<u4")
        a[:len(c)] = c
        self.L.orc_set_counters(self.h, _p(a))

    def counters(self):
        a = np.zeros(16, "<u4")
        self.L.orc_get_counters(self.h, _p(a))
        return a

    def reset_image(self):
        self.L.orc_reset_image(self.h)

    def reset_accumulated(self):
        self.L.orc_reset_accumulated(self.h)

    def swap_ray_queues(self):
        self.L.orc_swap_ray_queues(self.h)

    def write_rays(self, rays):
        a = np.ascontiguousarray(rays, RAY)
        self.L.orc_write_rays(self.h, _p(a), len(a))

    def generate_rays(self, gx, gy, true_size=False):
        self.L.orc_generate_rays(self.h, gx, gy, int(true_size))

    def extend(self, gx, gy):
        self.L.orc_extend(self.h, gx, gy)

    def shade(self, gx, gy):
        self.L.orc_shade(self.h, gx, gy)

    def miss(self, gx, gy):
        self.L.orc_miss(self.h, gx, gy)

    def accumulate(self, gx, gy):
        self.L.orc_accumulate(self.h, gx, gy)

    def render_sample(self):
        return self.L.orc_render_sample(self.h)

    def render(self, spp):
        for _ in range(spp):
            self.render_sample()
        return self.accumulated()

    def _view(self, ptr, dtype, n):
        buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    def rays(self, n=None):
        return self._view(self.L.orc_rays(self.h), RAY, self.n_slots if n is None else n)

    def extension_rays(self, n=None):
        return self._view(self.L.orc_extension_rays(self.h), RAY, self.n_slots if n is None else n)

    def hits(self, n):
        return self._view(self.L.orc_hits(self.h), HIT, n)

    def misses(self, n):
        return self._view(self.L.orc_misses(self.h), "<u4", n)

    def image(self):
        return self._view(self.L.orc_image(self.h), "<f4", 3 * self.n_pixels).reshape(-1, 3)

    def accumulated(self):
        return self._view(self.L.orc_accumulated(self.h), "<f4", 3 * self.n_pixels).reshape(-1, 3)

    def bounce_table(self):
        t = np.zeros((64, 4), "<u4")
        n = self.L.orc_bounce_table(self.h, _p(t), 64)
        return t[:n].copy()

    def totals(self):
        t = np.zeros(3, "<u8")
        self.L.orc_totals(self.h, _p(t))
        return t

    def trace_stats(self):
        t = np.zeros(4, "<u8")
        self.L.orc_trace_stats(self.h, _p(t))
        return dict(max_stack_depth=int(t[0]), node_visits=int(t[1]), sphere_tests=int(t[2]), rays=int(t[3]))

    def trace_brute(self, ray):
        r = np.ascontiguousarray(ray, RAY).reshape(1)
        out = np.zeros(1, HIT)
        h = self.L.orc_trace_brute(self.h, _p(r), _p(out))
        return bool(h), out[0]

    def trace_bvh(self, ray):
        r = np.ascontiguousarray(ray, RAY).reshape(1)
        out = np.zeros(1, HIT)
        h = self.L.orc_trace_bvh(self.h, _p(r), _p(out))
        return bool(h), out[0]


def tonemap_rgb8(acc, n_samples):
    a = np.ascontiguousarray(acc, "<f4").reshape(-1)
    out = np.zeros(a.size, np.uint8)
    lib().orc_tonemap_rgb8(_p(a), a.size // 3, n_samples, _p(out))
    return out.reshape(-1, 3)


def mesh_oracle(width, height, n_triangles, seed=1, n_bins=32, **kw):
    """Config helper: seeded triangle soup + BVH + the C5 camera."""
    tris, mt = scene_random_mesh(n_triangles, seed)
    tris, nodes = build_bvh_triangles(tris, n_bins)
    cam, ip, vw = mesh_camera(width, height)
    return Oracle(width, height, np.zeros(1, SPHERE), mt, nodes, cam, ip, vw, triangles=tris, **kw)


def shirley_oracle(width, height, seed=1, **kw):
    """Config helper: seeded Shirley scene + BVH + Shirley camera."""
    sp, mt = scene_book_one_final(seed)
    sp, nodes = build_bvh(sp)
    cam, ip, vw = shirley_camera(width, height)
    return Oracle(width, height, sp, mt, nodes, cam, ip, vw, **kw)
