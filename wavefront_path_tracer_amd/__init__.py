"""wavefront_path_tracer_amd -- host-side mirror of the reference's API for ONE path:
the wavefront kernel chain generate_rays -> extend -> shade -> miss_kernel -> accumulate.

Everything here is a thin layer over the C ABI in include/wfpt.h (libwfpt.so: hand-written gfx950 HIP
kernels). Class and method names follow the reference (rchiaramo/wavefront_path_tracer @ 2024_10_08):

    Scene, Material, Sphere        wavefront_common/src/{scene,material,sphere}.rs
    BVHTree                        wavefront_common/src/bvh.rs
    Camera, CameraController       wavefront_common/src/{camera,camera_controller}.rs
    ProjectionMatrix               wavefront_common/src/projection_matrix.rs
    GPUFrameBuffer                 wavefront_common/src/gpu_structs.rs
    RenderParameters/RenderProgress wavefront_common/src/parameters.rs
    Kernel                         gpu_wavefront_pt/src/kernel.rs
    PathTracer                     gpu_wavefront_pt/src/path_tracer.rs

There is no CPU fallback: if libwfpt.so is missing or no MI355X is visible, device calls raise WfptError.
"""
import ctypes as C
import os

import numpy as np

from . import _build

__all__ = ["SPP", "SPF", "Scene", "BVHTree", "Camera", "CameraController", "ProjectionMatrix", "GPUFrameBuffer",
           "RenderParameters", "RenderProgress", "Kernel", "PathTracer", "WfptError", "workgroup_size_64",
           "RNG_DISPATCH", "RNG_PIXEL", "FLAG_SPLIT_SHADE", "FLAG_NO_GRAPH", "FLAG_UNFUSED", "FLAG_BINARY_BVH", "FLAG_NO_REFILL", "FLAG_NO_LDS_SCENE", "FLAG_EXACT_TRAVERSAL", "FLAG_NO_BINNING", "FLAG_BINNING", "STAGES", "lib", "build",
           "tonemap_rgb8", "selftest_math", "device_count"]

SPP = 10  # wavefront_common/src/parameters.rs:4
SPF = 1   # wavefront_common/src/parameters.rs:5

RNG_DISPATCH, RNG_PIXEL = 0, 1
LOOP_KINDS = ("stages", "fused", "fused_binned", "refill")  # wfpt_loop_kind
FLAG_SPLIT_SHADE, FLAG_NO_GRAPH, FLAG_UNFUSED, FLAG_BINARY_BVH, FLAG_NO_REFILL, FLAG_NO_LDS_SCENE, FLAG_EXACT_TRAVERSAL, FLAG_NO_BINNING, FLAG_BINNING = 1, 2, 4, 8, 16, 32, 64, 128, 256
INACTIVE_PIXEL = 0xFFFFFFFF
# kernel.rs:32 loads shaders/{name}.wgsl; these are the stage names (path_tracer.rs:162,167,175,180,185)
STAGES = {"generate_rays": 0, "extend": 1, "shade": 2, "miss_kernel": 3, "accumulate": 4,
          "shade_lambertian": 5, "shade_metal": 6, "shade_dielectric": 7, "scan": 8,
          # fused launches of the device-resident loop (timing only; not dispatchable through Kernel)
          "bounce_first": 9, "bounce": 10, "bounce_last": 11, "compact": 12}
STAGE_COUNT = 13

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
# build extension (the reference has spheres only): vertex + two edges, 48 B
TRIANGLE = np.dtype([("v0", "<f4", 3), ("material_idx", "<u4"), ("e1", "<f4", 3), ("material_type", "<u4"),
                     ("e2", "<f4", 3), ("_pad", "<u4")])


OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_UNSUPPORTED, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5  # wfpt_status


class WfptError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"wfpt status {status}: {message}")
        self.status = status


class _Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_pixels", C.c_uint32),
                ("max_wavefronts", C.c_uint32), ("miss_floor", C.c_uint32), ("rng_mode", C.c_uint32),
                ("flags", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32),
                ("device", C.c_int32), ("batch", C.c_uint32)]


class GPUFrameBuffer(C.Structure):
    """wavefront_common/src/gpu_structs.rs:5-28"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("frame", C.c_uint32),
                ("sample_number", C.c_uint32)]

    @classmethod
    def new(cls, width, height, frame):
        return cls(width, height, frame, 0)

    def into_array(self):
        return [self.width, self.height, self.frame, self.sample_number]

    def set_sample_number(self, sample_number):
        self.sample_number = sample_number


def build(force=False, verbose=False):
    return _build.build(force=force, verbose=verbose)


_lib = None


def _agree_on_hip_runtime():
    """libwfpt.so needs `libamdhip64.so.7`; a PyTorch ROCm wheel ships its own copy of that runtime (with its own HSA
    runtime beside it) under the same SONAME. One process can initialise the GPU through one HIP/HSA pair only: with the
    system copy mapped first, a later `import torch` would bring a second HSA runtime along and find "No HIP GPUs". So when a
    PyTorch ROCm wheel is installed (whether or not it has been, or ever will be, imported) the copy it ships is the one
    mapped first, by its file name: whichever of torch and this package loads first, both end up on the same runtime. Nothing
    of torch is imported or executed here; hosts without PyTorch (C, C++, Rust) are not concerned."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # its runtime is already mapped; libwfpt.so's NEEDED entry resolves to it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return
    # Only a copy that will actually SATISFY libwfpt.so's NEEDED entry may be mapped first: the dynamic loader matches by
    # DT_SONAME, so a wheel built against another HIP major version (another SONAME) would sit beside the system runtime
    # that libwfpt.so then pulls in -- two HIP/HSA pairs in one process, the very failure this function exists to avoid.
    try:
        want = [n for n in _elf_dynamic(os.environ.get("WFPT_LIB", _build.LIB_PATH))[1] if n.startswith("libamdhip64.so")]
        soname = _elf_dynamic(cand)[0]
    except (OSError, ValueError):
        return
    if not want or soname != want[0]:
        return  # a different runtime generation: leave the choice to the load order, as before
    try:
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # fall back to the system runtime; `import torch` before this package still works as before


def _elf_dynamic(path):
    """(DT_SONAME, [DT_NEEDED ...]) of a 64-bit little-endian ELF shared object, read from the file (nothing is loaded)."""
    import struct
    with open(path, "rb") as f:
        head = f.read(64)
        if head[:4] != b"\x7fELF" or head[4] != 2 or head[5] != 1:
            raise ValueError("not a 64-bit little-endian ELF file")
        shoff, = struct.unpack_from("<Q", head, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", head, 0x3A)
        f.seek(shoff)
        sections = [struct.unpack_from("<IIQQQQIIQQ", f.read(shentsize)) for _ in range(shnum)]
        dyn = next((sec for sec in sections if sec[1] == 6), None)  # SHT_DYNAMIC
        if dyn is None:
            raise ValueError("no dynamic section")
        strtab = sections[dyn[6]]  # sh_link: the string table of the dynamic section
        f.seek(strtab[4])
        strings = f.read(strtab[5])
        f.seek(dyn[4])
        raw = f.read(dyn[5])
    soname, needed = None, []
    for off in range(0, len(raw) - 15, 16):
        tag, val = struct.unpack_from("<qQ", raw, off)
        if tag == 0:
            break
        if tag in (1, 14):  # DT_NEEDED, DT_SONAME
            name = strings[val:strings.index(b"\0", val)].decode()
            if tag == 1:
                needed.append(name)
            else:
                soname = name
    return soname, needed


def _one_hip_runtime_mapped():
    """After libwfpt.so is loaded: the process must hold exactly one libamdhip64 (see _agree_on_hip_runtime)."""
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    real = {os.path.realpath(x) for x in paths}
    if len(real) > 1:
        raise WfptError(-5, "two HIP runtimes are mapped into this process (" + ", ".join(sorted(real)) + "): the GPU can be "
                            "initialised through one of them only. Import torch before this package, or set LD_LIBRARY_PATH so "
                            "that libwfpt.so resolves libamdhip64 to the copy PyTorch ships")


def _lib_loaded():
    return _lib is not None


def lib():
    """Load libwfpt.so. Fails loudly when the HIP extension has not been built: nothing here can run without it."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("WFPT_LIB", _build.LIB_PATH)  # WFPT_LIB: tuning builds made with WFPT_EXTRA_FLAGS / WFPT_LIB_OUT
    if not os.path.exists(path):
        raise WfptError(-5, f"{path} is missing: run wavefront_path_tracer_amd.build() "
                            "(python -m wavefront_path_tracer_amd._build); there is no CPU fallback")
    _agree_on_hip_runtime()
    L = C.CDLL(path)
    _one_hip_runtime_mapped()
    vp, u32, i32, f32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_size_t
    sig = {
        "wfpt_scene_new": (u32, [vp, vp]),
        "wfpt_scene_book_one_final": (u32, [C.c_uint64, vp, vp, u32]),
        "wfpt_build_bvh": (i32, [vp, u32, vp, u32, C.POINTER(u32)]),
        "wfpt_build_bvh_triangles": (i32, [vp, u32, vp, u32, C.POINTER(u32), u32]),
        "wfpt_build_bvh_device": (i32, [vp, u32, vp, u32, C.POINTER(u32), i32, C.POINTER(f32)]),
        "wfpt_build_bvh_triangles_device": (i32, [vp, u32, vp, u32, C.POINTER(u32), u32, i32, C.POINTER(f32)]),
        "wfpt_load_obj": (i32, [C.c_char_p, vp, u32, C.POINTER(u32), u32, u32]),
        "wfpt_scene_random_mesh": (u32, [C.c_uint64, u32, vp, vp]),
        "wfpt_create_mesh": (vp, [C.POINTER(_Params), vp, u32, vp, u32, vp, u32, vp, vp, vp]),
        "wfpt_render_chunked": (i32, [C.POINTER(_Params), vp, u32, vp, u32, vp, u32, vp, vp, vp, u32, u32, vp]),
        "wfpt_render_chunked_mesh": (i32, [C.POINTER(_Params), vp, u32, vp, u32, vp, u32, vp, vp, vp, u32, u32, vp]),
        "wfpt_camera_new": (None, [vp, vp, C.POINTER(f32), C.POINTER(f32)]),
        "wfpt_view_transform": (None, [vp, f32, f32, vp]),
        "wfpt_p_inv": (None, [f32, f32, f32, f32, vp]),
        "wfpt_gpu_camera_new": (None, [vp, f32, f32, f32, f32, vp]),
        "wfpt_to_radians": (f32, [f32]),
        "wfpt_camera_controller_update": (None, [vp, C.POINTER(f32), C.POINTER(f32), vp, vp, f32, f32, f32]),
        "wfpt_workgroup_size_64": (None, [u32, C.POINTER(u32), C.POINTER(u32)]),
        "wfpt_stage_from_name": (i32, [C.c_char_p]),
        "wfpt_stage_name": (C.c_char_p, [i32]),
        "wfpt_device_count": (i32, []),
        "wfpt_create": (vp, [C.POINTER(_Params), vp, u32, vp, u32, vp, u32, vp, vp, vp]),
        "wfpt_destroy": (None, [vp]),
        "wfpt_update_scene": (i32, [vp, vp, u32, vp, u32]),
        "wfpt_update_scene_mesh": (i32, [vp, vp, u32, vp, u32, u32]),
        "wfpt_last_error": (C.c_char_p, [vp]),
        "wfpt_set_frame": (i32, [vp, C.POINTER(GPUFrameBuffer)]),
        "wfpt_update_render_parameters": (i32, [vp, u32, u32, vp, vp, vp]),
        "wfpt_set_counters": (i32, [vp, vp]),
        "wfpt_read_counters": (i32, [vp, vp]),
        "wfpt_reset_image": (i32, [vp]),
        "wfpt_reset_accumulated": (i32, [vp]),
        "wfpt_reset_progress": (i32, [vp]),
        "wfpt_clear_ray_queues": (i32, [vp]),
        "wfpt_swap_ray_queues": (i32, [vp]),
        "wfpt_kernel_run": (i32, [vp, i32, u32, u32]),
        "wfpt_kernel_timing_us": (f32, [vp, i32]),
        "wfpt_render_sample": (i32, [vp]),
        "wfpt_render": (i32, [vp, u32]),
        "wfpt_render_sample_timed": (i32, [vp, vp, vp]),
        "wfpt_render_timed": (i32, [vp, u32, vp, vp]),
        "wfpt_synchronize": (i32, [vp]),
        "wfpt_frame": (u32, [vp]),
        "wfpt_accumulated_samples": (u32, [vp]),
        "wfpt_progress": (f32, [vp, u32]),
        "wfpt_n_pixels": (u32, [vp]),
        "wfpt_ray_capacity": (u32, [vp]),
        "wfpt_read_accumulated": (i32, [vp, vp, sz]),
        "wfpt_read_image": (i32, [vp, vp, sz]),
        "wfpt_copy_accumulated_to_device": (i32, [vp, vp, sz]),
        "wfpt_comm_unique_id": (i32, [vp]),
        "wfpt_comm_init": (i32, [vp, vp, i32, i32]),
        "wfpt_gather_accumulated": (i32, [vp]),
        "wfpt_gather_accumulated_timed": (i32, [vp, C.POINTER(f32)]),
        "wfpt_loop_kind_of": (i32, [vp]),
        "wfpt_read_gathered": (i32, [vp, vp, sz]),
        "wfpt_comm_destroy": (i32, [vp]),
        "wfpt_read_rays": (i32, [vp, vp, u32]),
        "wfpt_read_extension_rays": (i32, [vp, vp, u32]),
        "wfpt_read_hits": (i32, [vp, vp, u32]),
        "wfpt_read_misses": (i32, [vp, vp, u32]),
        "wfpt_write_rays": (i32, [vp, vp, u32]),
        "wfpt_read_bounce_table": (i32, [vp, vp, u32, C.POINTER(u32)]),
        "wfpt_read_totals": (i32, [vp, vp]),
        "wfpt_read_wavefront_totals": (i32, [vp, vp, u32, C.POINTER(u32)]),
        "wfpt_device_info": (i32, [i32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64)]),
        "wfpt_tonemap_rgb8": (None, [vp, u32, u32, vp]),
        "wfpt_selftest_math": (i32, [i32, i32, vp, vp, vp, sz]),
        "wfpt_build_info": (C.c_char_p, []),
        "wfpt_save_ppm": (i32, [vp, C.c_char_p]),
        "wfpt_save_pfm": (i32, [vp, C.c_char_p]),
        "wfpt_save_png": (i32, [vp, C.c_char_p]),
        "wfpt_write_png_rgb8": (i32, [C.c_char_p, vp, u32, u32]),
        "wfpt_debug_extend_blocks_per_cu": (i32, [i32, u32]),
        "wfpt_debug_read_stamps": (i32, [vp, vp, i32]),
        "wfpt_debug_read_stamps_ex": (i32, [vp, i32, vp, i32]),
        "wfpt_debug_bvh4": (i32, [vp, u32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what wfpt.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


ABI_SYMBOLS = None  # filled lazily by abi_symbols()


def abi_symbols():
    """Every function include/wfpt.h declares (parsed from the header)."""
    import re
    hdr = os.path.join(_build.ROOT, "include", "wfpt.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wfpt_[a-z0-9_]+)\s*\(", text)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().wfpt_device_count()


def comm_unique_id():
    """Rank 0: a fresh RCCL unique id (128 bytes) to hand to every rank's PathTracer.comm_init."""
    buf = np.zeros(128, np.uint8)
    st = lib().wfpt_comm_unique_id(_p(buf))
    if st != 0:
        raise WfptError(st, lib().wfpt_last_error(None).decode())
    return buf.tobytes()


def device_info(device=0):
    """CU count, memory clock (kHz), memory bus width (bits), memory bytes of HIP device `device`."""
    cu, clk, width, mem = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    st = lib().wfpt_device_info(device, C.byref(cu), C.byref(clk), C.byref(width), C.byref(mem))
    if st != 0:
        raise WfptError(st, lib().wfpt_last_error(None).decode())
    return {"compute_units": cu.value, "memory_clock_khz": clk.value, "memory_bus_width_bits": width.value,
            "total_memory_bytes": mem.value}


def workgroup_size_64(x):
    """path_tracer.rs:282-289"""
    gx, gy = C.c_uint32(), C.c_uint32()
    lib().wfpt_workgroup_size_64(x, C.byref(gx), C.byref(gy))
    return gx.value, gy.value


def tonemap_rgb8(accumulated, n_samples):
    a = np.ascontiguousarray(accumulated, "<f4").reshape(-1)
    out = np.zeros(a.size, np.uint8)
    lib().wfpt_tonemap_rgb8(_p(a), a.size // 3, n_samples, _p(out))
    return out.reshape(-1, 3)


def write_png(path, rgb8, width, height):
    """An (height, width, 3) uint8 image as a PNG file (wfpt_write_png_rgb8; no compression library)."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    if a.size != 3 * width * height:
        raise ValueError("rgb8 must hold width * height RGB pixels")
    st = lib().wfpt_write_png_rgb8(os.fsencode(path), _p(a), width, height)
    if st != 0:
        raise WfptError(st, f"cannot write {path}")


def selftest_math(op, a, b=None, device=0):
    a = np.ascontiguousarray(a, "<f4")
    out = np.zeros_like(a)
    bb = None if b is None else np.ascontiguousarray(b, "<f4")
    st = lib().wfpt_selftest_math(device, op, _p(a), None if bb is None else _p(bb), _p(out), a.size)
    if st != 0:
        raise WfptError(st, lib().wfpt_last_error(None).decode())
    return out


# ------------------------------------------------------------------------------------------------
# wavefront_common data model
# ------------------------------------------------------------------------------------------------
class Scene:
    """wavefront_common/src/scene.rs: `spheres` and `materials` in the reference's 32-byte layouts."""

    def __init__(self, spheres, materials, triangles=None):
        self.spheres = np.ascontiguousarray(spheres, SPHERE)
        self.materials = np.ascontiguousarray(materials, MATERIAL)
        self.triangles = None if triangles is None else np.ascontiguousarray(triangles, TRIANGLE)  # build extension

    @classmethod
    def from_obj(cls, path, materials=None, material_idx=0):
        """Build extension (README.md:25): the triangles of a Wavefront OBJ file, all with one material
        (default: Lambertian 0.7 grey, like the mesh scene's first material)."""
        if materials is None:
            materials = np.zeros(1, MATERIAL)
            materials["albedo"][0] = (0.7, 0.7, 0.7, 1.0)
        materials = np.ascontiguousarray(materials, MATERIAL)
        n = C.c_uint32()
        st = lib().wfpt_load_obj(os.fsencode(path), None, 0, C.byref(n), 0, 0)
        if st != 0 or n.value == 0:
            raise WfptError(st or ERR_INVALID_ARGUMENT, f"cannot read triangles from {path}")
        tris = np.zeros(n.value, TRIANGLE)
        mtype = int(materials["material_type"][material_idx])
        st = lib().wfpt_load_obj(os.fsencode(path), _p(tris), len(tris), C.byref(n), material_idx, mtype)
        if st != 0:
            raise WfptError(st, f"cannot read triangles from {path}")
        return cls(np.zeros(0, SPHERE), materials, triangles=tris)

    @classmethod
    def random_mesh(cls, n_triangles, seed=1):
        """BASELINE config 5's seeded triangle soup (build extension: the reference has no triangle type)."""
        tr, mt = np.zeros(n_triangles, TRIANGLE), np.zeros(3, MATERIAL)
        lib().wfpt_scene_random_mesh(seed, n_triangles, _p(tr), _p(mt))
        return cls(np.zeros(0, SPHERE), mt, tr)

    @classmethod
    def new(cls):
        """scene.rs:12-46"""
        sp, mt = np.zeros(5, SPHERE), np.zeros(5, MATERIAL)
        n = lib().wfpt_scene_new(_p(sp), _p(mt))
        return cls(sp[:n], mt[:n])

    @classmethod
    def book_one_final(cls, seed=1):
        """scene.rs:48-107, seeded (the reference draws from an unseeded thread_rng)."""
        sp, mt = np.zeros(512, SPHERE), np.zeros(512, MATERIAL)
        n = lib().wfpt_scene_book_one_final(seed, _p(sp), _p(mt), 512)
        if n == 0:
            raise WfptError(-1, "wfpt_scene_book_one_final failed")
        return cls(sp[:n].copy(), mt[:n].copy())


class BVHTree:
    """wavefront_common/src/bvh.rs:143-210"""

    def __init__(self, num_primitives):
        self.capacity = 2 * max(int(num_primitives), 1)
        self.nodes = np.zeros(0, BVH_NODE)

    def build_bvh_tree(self, spheres, device=None):
        """Reorders `spheres` (a SPHERE array) in place, like the reference."""
        if not (isinstance(spheres, np.ndarray) and spheres.dtype == SPHERE and spheres.flags.c_contiguous):
            raise TypeError("spheres must be a contiguous SPHERE array (it is reordered in place)")
        nodes = np.zeros(self.capacity, BVH_NODE)
        n = C.c_uint32()
        if device is None:
            st = lib().wfpt_build_bvh(_p(spheres), len(spheres), _p(nodes), len(nodes), C.byref(n))
        else:  # build extension: the same builder on HIP device `device`, same bytes out
            ms = C.c_float()
            st = lib().wfpt_build_bvh_device(_p(spheres), len(spheres), _p(nodes), len(nodes), C.byref(n), device, C.byref(ms))
            self.device_ms = ms.value
        if st != 0:
            raise WfptError(st, "wfpt_build_bvh failed: " + lib().wfpt_last_error(None).decode())
        self.nodes = nodes[:n.value].copy()

    def build_bvh_tree_triangles(self, triangles, n_bins=32, device=None):
        """Build extension: the same builder over a TRIANGLE array (reordered in place), n_bins bins per axis."""
        if not (isinstance(triangles, np.ndarray) and triangles.dtype == TRIANGLE and triangles.flags.c_contiguous):
            raise TypeError("triangles must be a contiguous TRIANGLE array (it is reordered in place)")
        nodes = np.zeros(self.capacity, BVH_NODE)
        n = C.c_uint32()
        if device is None:
            st = lib().wfpt_build_bvh_triangles(_p(triangles), len(triangles), _p(nodes), len(nodes), C.byref(n), n_bins)
        else:
            ms = C.c_float()
            st = lib().wfpt_build_bvh_triangles_device(_p(triangles), len(triangles), _p(nodes), len(nodes), C.byref(n), n_bins,
                                                       device, C.byref(ms))
            self.device_ms = ms.value
        if st != 0:
            raise WfptError(st, "wfpt_build_bvh_triangles failed: " + lib().wfpt_last_error(None).decode())
        self.nodes = nodes[:n.value].copy()


class Camera:
    """wavefront_common/src/camera.rs"""

    def __init__(self, look_from, look_at):
        self.position = np.asarray(look_from, "<f4").copy()
        la = np.asarray(look_at, "<f4")
        pitch, yaw = C.c_float(), C.c_float()
        lib().wfpt_camera_new(_p(self.position), _p(la), C.byref(pitch), C.byref(yaw))
        self.pitch, self.yaw = pitch.value, yaw.value

    @classmethod
    def book_one_final_camera(cls):
        """camera.rs:26-30"""
        return cls((13.0, 2.0, 3.0), (0.0, 0.0, 0.0))

    def get_camera(self):
        return self.position, self.pitch, self.yaw

    def view_transform(self):
        """camera.rs:41-69: 16 floats, column-major."""
        view = np.zeros(16, "<f4")
        lib().wfpt_view_transform(_p(self.position), self.pitch, self.yaw, _p(view))
        return view


class CameraController:
    """wavefront_common/src/camera_controller.rs:8-158"""

    def __init__(self, camera, vfov, defocus_angle, focus_distance, z_near, z_far, speed=4.0, sensitivity=0.1):
        L = lib()
        self.camera = camera
        self._vfov_rad = L.wfpt_to_radians(vfov)
        self.defocus_angle_rad = L.wfpt_to_radians(defocus_angle)
        self.focus_distance = focus_distance
        self.z_near, self.z_far = z_near, z_far
        self.speed, self.sensitivity = speed, sensitivity
        self._amounts = np.zeros(6, "<f4")  # forward, backward, right, left, up, down
        self._rotate = np.zeros(2, "<f4")   # horizontal, vertical

    def copy(self):
        """The reference's controller is `Copy`; hosts edit a copy and hand it to update_camera_controller."""
        other = CameraController.__new__(CameraController)
        other.__dict__.update(self.__dict__)
        cam = Camera.__new__(Camera)
        cam.position, cam.pitch, cam.yaw = self.camera.position.copy(), self.camera.pitch, self.camera.yaw
        other.camera, other._amounts, other._rotate = cam, self._amounts.copy(), self._rotate.copy()
        return other

    def vfov_rad(self):
        return self._vfov_rad

    def set_vfov(self, vfov):
        self._vfov_rad = lib().wfpt_to_radians(vfov)

    def dof(self):
        return self.defocus_angle_rad, self.focus_distance

    def set_defocus_angle(self, defocus_angle):
        self.defocus_angle_rad = lib().wfpt_to_radians(defocus_angle)

    def set_focus_distance(self, focus_distance):
        self.focus_distance = focus_distance

    def process_mouse(self, delta):
        """camera_controller.rs:74-77"""
        self._rotate[:] = delta

    def _press(self, slot, direction):
        self._amounts[slot] = 1.0 if direction == 1 else 0.0

    def move_forward(self, direction):  # camera_controller.rs:95-101
        self._press(0, direction)

    def move_backwards(self, direction):  # :103-109
        self._press(1, direction)

    def move_right(self, direction):  # :111-117
        self._press(2, direction)

    def move_left(self, direction):  # :119-125
        self._press(3, direction)

    def move_up(self, direction):  # :79-85
        self._press(4, direction)

    def move_down(self, direction):  # :87-93
        self._press(5, direction)

    def update_camera(self, dt):
        """camera_controller.rs:125-158"""
        pitch, yaw = C.c_float(self.camera.pitch), C.c_float(self.camera.yaw)
        lib().wfpt_camera_controller_update(_p(self.camera.position), C.byref(pitch), C.byref(yaw), _p(self._amounts),
                                            _p(self._rotate), self.speed, self.sensitivity, dt)
        self.camera.pitch, self.camera.yaw = pitch.value, yaw.value

    def get_clip_planes(self):
        return self.z_near, self.z_far

    def get_GPU_camera(self):
        """camera_controller.rs:66-68, 173-185"""
        cam = np.zeros(1, GPU_CAMERA)
        lib().wfpt_gpu_camera_new(_p(self.camera.position), self.camera.pitch, self.camera.yaw,
                                  self.defocus_angle_rad, self.focus_distance, _p(cam))
        return cam

    def get_view_matrix(self):
        return self.camera.view_transform()


class ProjectionMatrix:
    """wavefront_common/src/projection_matrix.rs"""

    def __init__(self, vfov_rad, aspect_ratio, z_near, z_far):
        self.vfov_rad, self.aspect_ratio, self.z_near, self.z_far = vfov_rad, aspect_ratio, z_near, z_far

    def p_inv(self):
        out = np.zeros(16, "<f4")
        lib().wfpt_p_inv(self.vfov_rad, self.aspect_ratio, self.z_near, self.z_far, _p(out))
        return out


class RenderParameters:
    """wavefront_common/src/parameters.rs:7-58"""

    def __init__(self, camera_controller, viewport_size):
        self._camera_controller = camera_controller
        self._viewport_size = tuple(viewport_size)
        self._resized = False
        self._camera_changed = False

    def changed(self):
        return self._resized or self._camera_changed

    def resized(self):
        return self._resized

    def camera_changed(self):
        return self._camera_changed

    def set_viewport(self, size):
        self._viewport_size = tuple(size)
        self._resized = True

    def viewport_size(self):
        return self._viewport_size

    def reset(self):
        self._resized = False
        self._camera_changed = False

    def camera_controller(self):
        return self._camera_controller

    def update_camera_controller(self, camera_controller):
        self._camera_controller = camera_controller
        self._camera_changed = True


class RenderProgress:
    """wavefront_common/src/parameters.rs:61-101"""

    def __init__(self):
        self.frame = 0
        self._accumulated_samples = 0

    def get_next_frame(self, rp):
        w, h = rp.viewport_size()
        self.frame += 1
        return GPUFrameBuffer.new(w, h, self.frame)

    def incr_accumulated_samples(self, delta):
        self._accumulated_samples += delta

    def reset(self):
        self._accumulated_samples = 0
        self.frame = 0

    def progress(self):
        return self._accumulated_samples / SPP

    def accumulated_samples(self):
        return self._accumulated_samples


# ------------------------------------------------------------------------------------------------
# kernel-stage API and the wavefront loop
# ------------------------------------------------------------------------------------------------
class Kernel:
    """gpu_wavefront_pt/src/kernel.rs: `Kernel::new(name, ...)`, `run((gx, gy))`, `get_timing()`.

    The reference binds buffers explicitly; here the context owns them (path_tracer.rs:53-128) and the
    stage name selects the binding table of SURVEY.md section 2.2."""

    def __init__(self, name, path_tracer):
        stage = lib().wfpt_stage_from_name(name.encode())
        if stage < 0 or stage >= STAGES["scan"]:
            # kernel.rs:36 unwraps the shader read and panics on an unknown name
            raise WfptError(-1, f"no such kernel stage: {name!r}")
        self.name, self.stage, self._pt = name, stage, path_tracer

    def run(self, workgroup_size):
        gx, gy = workgroup_size
        self._pt._check(lib().wfpt_kernel_run(self._pt.handle, self.stage, gx, gy))

    def get_timing(self):
        """Running mean (microseconds) of the last <= 10 dispatches (query_gpu.rs:26-43)."""
        return lib().wfpt_kernel_timing_us(self._pt.handle, self.stage)


class PathTracer:
    """gpu_wavefront_pt/src/path_tracer.rs. `new` builds the BVH (reordering scene.spheres, path_tracer.rs:117-118)
    and uploads everything; `run()` is the reference's host-driven loop over the five Kernels with blocking
    counter read-backs; `render(spp)` is the same loop resident on the device (no host synchronisation)."""

    def __init__(self, scene, rp, max_window_size=0, max_wavefronts=50, miss_floor=128, rng_mode=RNG_DISPATCH,
                 flags=0, tile_rank=0, tile_world=1, device=0, spp=SPP, batch=0, mesh_bins=32, device_bvh=False, bvh=None):
        """`bvh`: a BVHTree the caller built itself over scene.spheres / scene.triangles AS THEY ARE (the C ABI takes any tree in bvh.rs's
        layout: siblings at (2k, 2k + 1), at most 63 levels); default: built here like path_tracer.rs:117-118 does."""
        L = lib()
        self.handle = None
        self.scene = scene
        self.render_parameters = rp
        self.render_progress = RenderProgress()
        self.spp = spp
        self.max_wavefronts, self.miss_floor = max_wavefronts, miss_floor
        if bvh is not None:
            pass
        elif scene.triangles is not None:
            bvh = BVHTree(len(scene.triangles))
            bvh.build_bvh_tree_triangles(scene.triangles, mesh_bins, device=device if device_bvh else None)
        else:
            bvh = BVHTree(len(scene.spheres))
            bvh.build_bvh_tree(scene.spheres, device=device if device_bvh else None)  # path_tracer.rs:117-118
        self.bvh_tree = bvh
        cc = rp.camera_controller()
        w, h = rp.viewport_size()
        self.width, self.height = w, h
        z_near, z_far = cc.get_clip_planes()
        ar = np.float32(w) / np.float32(h)
        proj = ProjectionMatrix(cc.vfov_rad(), ar, z_near, z_far).p_inv()  # path_tracer.rs:135-138
        view = cc.get_view_matrix()
        cam = cc.get_GPU_camera()
        self._params = _Params(w, h, max_window_size, max_wavefronts, miss_floor, rng_mode, flags,
                               tile_rank, tile_world, device, batch)
        if scene.triangles is not None:
            self.handle = L.wfpt_create_mesh(C.byref(self._params), _p(scene.triangles), len(scene.triangles),
                                             _p(scene.materials), len(scene.materials), _p(bvh.nodes), len(bvh.nodes),
                                             _p(cam), _p(proj), _p(view))
        else:
            self.handle = L.wfpt_create(C.byref(self._params), _p(scene.spheres), len(scene.spheres),
                                        _p(scene.materials), len(scene.materials), _p(bvh.nodes), len(bvh.nodes),
                                        _p(cam), _p(proj), _p(view))
        if not self.handle:
            raise WfptError(-2, L.wfpt_last_error(None).decode())
        self.n_pixels = L.wfpt_n_pixels(self.handle)
        self.ray_capacity = L.wfpt_ray_capacity(self.handle)
        # path_tracer.rs:158-186
        self.generate_ray_kernel = Kernel("generate_rays", self)
        self.extend_kernel = Kernel("extend", self)
        self.shade_kernel = Kernel("shade", self)
        self.miss_kernel = Kernel("miss_kernel", self)
        self.accumulate_kernel = Kernel("accumulate", self)
        self.last_wavefronts = 0

    # ---- plumbing
    def _check(self, status):
        if status != 0:
            raise WfptError(status, lib().wfpt_last_error(self.handle).decode())

    def close(self):
        if self.handle:
            lib().wfpt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API
    def progress(self):
        return self.render_progress.progress()

    def get_render_parameters(self):
        return self.render_parameters

    def update_render_parameters(self, render_parameters):
        self.render_parameters = render_parameters

    def resize(self, rp):
        self.update_render_parameters(rp)

    def update_buffers(self):
        """path_tracer.rs:240-277"""
        rp = self.render_parameters
        if not rp.changed():
            return
        cc = rp.camera_controller()
        w, h = rp.viewport_size()
        z_near, z_far = cc.get_clip_planes()
        proj = ProjectionMatrix(cc.vfov_rad(), np.float32(w) / np.float32(h), z_near, z_far).p_inv()
        view = cc.get_view_matrix()
        cam = cc.get_GPU_camera()
        self._check(lib().wfpt_update_render_parameters(self.handle, w, h, _p(cam), _p(proj), _p(view)))
        self.width, self.height = w, h
        self.n_pixels = lib().wfpt_n_pixels(self.handle)
        rp.reset()
        self.render_progress.reset()

    def update_scene(self, scene, mesh_bins=32):
        """Build extension (dynamic scenes): replaces the scene of this live context. The BVH is rebuilt on the device
        (reordering scene.spheres / scene.triangles in place, like PathTracer::new does, path_tracer.rs:117-118) and the
        accumulation restarts at frame 1 (update_buffers, path_tracer.rs:240-277)."""
        if scene.triangles is not None:
            self._check(lib().wfpt_update_scene_mesh(self.handle, _p(scene.triangles), len(scene.triangles), _p(scene.materials),
                                                     len(scene.materials), mesh_bins))
        else:
            self._check(lib().wfpt_update_scene(self.handle, _p(scene.spheres), len(scene.spheres), _p(scene.materials),
                                                len(scene.materials)))
        self.scene = scene
        self.bvh_tree = None  # the new tree exists on the device only (the primitives above were reordered to match it)
        self.render_progress.reset()

    def set_frame(self, frame):
        self._check(lib().wfpt_set_frame(self.handle, C.byref(frame)))

    def set_counters(self, values):
        a = np.zeros(16, "<u4")
        a[:len(values)] = values
        self._check(lib().wfpt_set_counters(self.handle, _p(a)))

    def read_counters(self):
        a = np.zeros(16, "<u4")
        self._check(lib().wfpt_read_counters(self.handle, _p(a)))
        return a

    def reset_image(self):
        self._check(lib().wfpt_reset_image(self.handle))

    def reset_accumulated(self):
        self._check(lib().wfpt_reset_accumulated(self.handle))

    def reset_progress(self):
        """RenderProgress::reset + accumulation clear: the next sample is frame 1 again."""
        self._check(lib().wfpt_reset_progress(self.handle))
        self.render_progress.reset()

    def clear_ray_queues(self):
        self._check(lib().wfpt_clear_ray_queues(self.handle))

    def swap_ray_queues(self):
        self._check(lib().wfpt_swap_ray_queues(self.handle))

    def run(self):
        """path_tracer.rs:279-371, host-driven, one sample (SPF = 1) per call until `spp` are accumulated."""
        self.update_buffers()
        if self.render_progress.accumulated_samples() < self.spp:
            frame = self.render_progress.get_next_frame(self.render_parameters)
            for sample_number in range(SPF):
                frame.set_sample_number(sample_number)
                self.set_frame(frame)                                    # :296-297
                self.reset_image()                                       # :305-306
                self.clear_ray_queues()                                  # :309-310
                width, height = self.render_parameters.viewport_size()
                self.set_counters([0, 0, width * height])                # :313-316
                self.generate_ray_kernel.run((width // 8, height // 8))  # :318
                wavefront = 0
                extend_size = workgroup_size_64(width * height)          # :322
                while wavefront < self.max_wavefronts:                   # :323
                    self.extend_kernel.run(extend_size)                  # :325
                    counter = self.read_counters()                       # :327-328
                    num_misses, num_hits = int(counter[0]), int(counter[1])
                    if num_misses < self.miss_floor:                     # :332
                        break
                    counter[2] = 0                                       # :335-336
                    self.set_counters(counter)
                    self.shade_kernel.run(workgroup_size_64(num_hits))   # :339
                    self.miss_kernel.run(workgroup_size_64(num_misses))  # :340
                    num_extension = int(self.read_counters()[2])         # :343-345
                    self.swap_ray_queues()                               # :348
                    extend_size = workgroup_size_64(num_extension)       # :350
                    self.set_counters([0, 0, num_extension, 0])          # :352
                    wavefront += 1
                self.last_wavefronts = wavefront
                self.accumulate_kernel.run(workgroup_size_64(width * height))  # :362
                self.render_progress.incr_accumulated_samples(1)         # :363
                frame.set_sample_number(self.render_progress.accumulated_samples())
                self.set_frame(frame)                                    # :366-367

    # ---- device-resident loop
    def render_sample(self):
        self._check(lib().wfpt_render_sample(self.handle))

    def render(self, spp):
        self._check(lib().wfpt_render(self.handle, spp))

    def render_sample_timed(self):
        ms = np.zeros(STAGE_COUNT, "<f4")
        launches = np.zeros(STAGE_COUNT, "<u4")
        self._check(lib().wfpt_render_sample_timed(self.handle, _p(ms), _p(launches)))
        return ms, launches

    def render_timed(self, n_samples):
        """Like render(n_samples) (same batching) with hipEvent pairs around every launch: (ms, launches) per stage."""
        ms = np.zeros(STAGE_COUNT, "<f4")
        launches = np.zeros(STAGE_COUNT, "<u4")
        self._check(lib().wfpt_render_timed(self.handle, n_samples, _p(ms), _p(launches)))
        return ms, launches

    def synchronize(self):
        self._check(lib().wfpt_synchronize(self.handle))

    # ---- read-back
    def accumulated(self):
        a = np.zeros((self.n_pixels, 3), "<f4")
        self._check(lib().wfpt_read_accumulated(self.handle, _p(a), a.size))
        return a

    def image(self):
        a = np.zeros((self.n_pixels, 3), "<f4")
        self._check(lib().wfpt_read_image(self.handle, _p(a), a.size))
        return a

    def copy_accumulated_to_device(self, device_ptr, n_bytes):
        self._check(lib().wfpt_copy_accumulated_to_device(self.handle, C.c_void_p(device_ptr), n_bytes))

    # ---- multi-GPU gather (RCCL behind the C ABI)
    def comm_init(self, unique_id, rank, world):
        """Collective over all ranks: joins the RCCL communicator named by `unique_id` (128 bytes from comm_unique_id())."""
        buf = np.frombuffer(bytes(unique_id), np.uint8).copy()
        self._check(lib().wfpt_comm_init(self.handle, _p(buf), rank, world))

    def gather_accumulated(self):
        """Every rank: its slab goes to rank 0 over xGMI (asynchronous on the context's stream)."""
        self._check(lib().wfpt_gather_accumulated(self.handle))

    def gather_accumulated_timed(self):
        """The same gather, blocking; returns its duration on this rank in milliseconds (hipEvents on the context's stream)."""
        ms = C.c_float(0.0)
        self._check(lib().wfpt_gather_accumulated_timed(self.handle, C.byref(ms)))
        return float(ms.value)

    @property
    def loop_kind(self):
        """Which loop render() enqueues: "stages", "fused", "fused_binned" or "refill" (wfpt_loop_kind_of)."""
        k = lib().wfpt_loop_kind_of(self.handle)
        if k < 0:
            self._check(k)
        return LOOP_KINDS[k]

    def gathered(self):
        """Rank 0: the assembled (width * height, 3) accumulated frame."""
        a = np.zeros((self.width * self.height, 3), "<f4")
        self._check(lib().wfpt_read_gathered(self.handle, _p(a), a.size))
        return a

    def rays(self, n):
        a = np.zeros(n, RAY)
        self._check(lib().wfpt_read_rays(self.handle, _p(a), n))
        return a

    def extension_rays(self, n):
        a = np.zeros(n, RAY)
        self._check(lib().wfpt_read_extension_rays(self.handle, _p(a), n))
        return a

    def write_rays(self, rays):
        a = np.ascontiguousarray(rays, RAY)
        self._check(lib().wfpt_write_rays(self.handle, _p(a), len(a)))

    def hits(self, n):
        a = np.zeros(n, HIT)
        self._check(lib().wfpt_read_hits(self.handle, _p(a), n))
        return a

    def misses(self, n):
        a = np.zeros(n, "<u4")
        self._check(lib().wfpt_read_misses(self.handle, _p(a), n))
        return a

    def save_ppm(self, path):
        """8-bit P6 of sqrt(accumulated / samples), the display shader's tone map (display_shader.wgsl:50-52)."""
        self._check(lib().wfpt_save_ppm(self.handle, os.fsencode(path)))

    def save_png(self, path):
        """The tone-mapped frame (display_shader.wgsl:50-52) as an 8-bit RGB PNG."""
        self._check(lib().wfpt_save_png(self.handle, os.fsencode(path)))

    def save_pfm(self, path):
        """Linear float32 PFM of accumulated / samples."""
        self._check(lib().wfpt_save_pfm(self.handle, os.fsencode(path)))

    def bounce_table(self):
        t = np.zeros((64, 4), "<u4")
        n = C.c_uint32()
        self._check(lib().wfpt_read_bounce_table(self.handle, _p(t), 64, C.byref(n)))
        return t[:n.value].copy()

    def totals(self):
        t = np.zeros(3, "<u8")
        self._check(lib().wfpt_read_totals(self.handle, _p(t)))
        return t

    def wavefront_totals(self):
        """(rays traced, hits, misses) per wavefront, summed over every sample rendered by the device-resident loop."""
        t = np.zeros((64, 3), "<u8")
        n = C.c_uint32()
        self._check(lib().wfpt_read_wavefront_totals(self.handle, _p(t), 64, C.byref(n)))
        return t[:n.value].copy()


def render_chunked(scene, rp, spp, chunks, max_wavefronts=50, miss_floor=128, rng_mode=RNG_PIXEL, flags=0, device=0, batch=0,
                   mesh_bins=32):
    """wfpt_render_chunked: the frame as `chunks` band-interleaved slabs rendered one after the other on one GPU (README.md:20),
    assembled on the host; returns the accumulated (width * height, 3) image. Builds the BVH like PathTracer does."""
    L = lib()
    cc = rp.camera_controller()
    w, h = rp.viewport_size()
    z_near, z_far = cc.get_clip_planes()
    proj = ProjectionMatrix(cc.vfov_rad(), np.float32(w) / np.float32(h), z_near, z_far).p_inv()
    view, cam = cc.get_view_matrix(), cc.get_GPU_camera()
    params = _Params(w, h, 0, max_wavefronts, miss_floor, rng_mode, flags, 0, 1, device, batch)
    out = np.zeros((w * h, 3), "<f4")
    if scene.triangles is not None:
        bvh = BVHTree(len(scene.triangles))
        bvh.build_bvh_tree_triangles(scene.triangles, mesh_bins)
        st = L.wfpt_render_chunked_mesh(C.byref(params), _p(scene.triangles), len(scene.triangles), _p(scene.materials), len(scene.materials),
                                        _p(bvh.nodes), len(bvh.nodes), _p(cam), _p(proj), _p(view), spp, chunks, _p(out))
    else:
        bvh = BVHTree(len(scene.spheres))
        bvh.build_bvh_tree(scene.spheres)
        st = L.wfpt_render_chunked(C.byref(params), _p(scene.spheres), len(scene.spheres), _p(scene.materials), len(scene.materials),
                                   _p(bvh.nodes), len(bvh.nodes), _p(cam), _p(proj), _p(view), spp, chunks, _p(out))
    if st != 0:
        raise WfptError(st, L.wfpt_last_error(None).decode())
    return out


def shirley_path_tracer(width, height, seed=1, **kw):
    """main.rs:17-36: seeded Shirley scene, book camera (13,2,3)->origin, vfov 20, defocus 0.6, focus 10."""
    scene = Scene.book_one_final(seed)
    cc = CameraController(Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    return PathTracer(scene, RenderParameters(cc, (width, height)), **kw)


def mesh_path_tracer(width, height, n_triangles, seed=1, **kw):
    """BASELINE config 5: seeded triangle soup, camera (0,0,30) -> origin, vfov 40, no defocus."""
    scene = Scene.random_mesh(n_triangles, seed)
    cc = CameraController(Camera((0.0, 0.0, 30.0), (0.0, 0.0, 0.0)), 40.0, 0.0, 10.0, 0.1, 100.0)
    return PathTracer(scene, RenderParameters(cc, (width, height)), **kw)
