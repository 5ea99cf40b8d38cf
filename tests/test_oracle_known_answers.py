"""Pins the oracle's integer pieces and host constants against known-answer values derived from the
reference's own formulas -- the only vectors the reference can provide: it ships no
fixtures, its single #[test] (wavefront_common/src/camera.rs:72-86) asserts nothing."""
import ctypes as C

import numpy as np


def test_jenkins_hash(orc):  # generate_rays.wgsl:173-181
    L = orc.lib()
    assert L.orc_probe_jenkins(0) == 0x0
    assert L.orc_probe_jenkins(1) == 0x124EA49D
    assert L.orc_probe_jenkins(2) == 0x249DC93B


def test_init_rng_and_pcg_stream(orc):  # generate_rays.wgsl:138-153
    L = orc.lib()
    cases = [((0, 0, 1920, 1), 0xC0738807, [0x3A1AF3C9, 0x0DF48F23, 0x13A1C877, 0xC40A3AEB]),
             ((1, 0, 1920, 1), 0x0A5E9BDC, [0x85FCA401, 0x56D442B7]),
             ((0, 1, 1920, 1), 0xB59D70C3, [0x3356AA3E, 0xDE87A63D]),
             ((959, 539, 1920, 1), 0xC66E6241, [0xB6BB0AE7, 0x9A768E5D]),
             ((7, 3, 400, 2), 0xC0AFB202, [0xB641D8FC, 0xE254E680])]
    for args, state, ints in cases:
        assert L.orc_probe_init_rng(*args) == state
        s = C.c_uint32(state)
        assert [L.orc_probe_next_int(C.byref(s)) for _ in ints] == ints


def test_rng_next_float(orc):  # generate_rays.wgsl:133-136
    L = orc.lib()
    s = C.c_uint32(0xC0738807)
    assert np.float32(L.orc_probe_next_float(C.byref(s))) == np.float32(0.22697376)
    assert np.float32(L.orc_probe_next_float(C.byref(s))) == np.float32(0.054512925)
    assert np.float32(2.3283064365387e-10) == np.float32(2.0 ** -32)  # the WGSL literal is exactly 2^-32
    # u32 -> f32 rounds to nearest even, so the range is [0, 1] INCLUSIVE
    assert np.float32(np.uint32(0xFFFFFFFF)) * np.float32(2.0 ** -32) == np.float32(1.0)


def test_advance_reproduces_the_reference_bug(orc):  # generate_rays.wgsl:155-171
    L = orc.lib()
    s = 0x12345678
    want = {0: 0x12345678, 1: 0xCFF935DD, 2: 0x439D1B46, 3: 0x439D1B46, 10: 0x7647AD10, 20: 0x304714A8, 30: 0x304714A8}
    for n, v in want.items():
        assert L.orc_probe_advance(s, n) == v
    # a true LCG skip-ahead would give these instead; the reference (and so the oracle) does not
    assert L.orc_probe_advance(s, 3) != 0x18041D83 and L.orc_probe_advance(s, 10) != 0xCAC74D1E


def test_workgroup_size_64(orc):  # path_tracer.rs:282-289
    table = {2073600: (162, 200), 90000: (21, 67), 89600: (35, 40), 4665600: (243, 300), 8294400: (324, 400),
             1000003: (26, 601), 65: (1, 2), 129: (1, 3), 1500000: (2, 11719)}
    for x, want in table.items():
        assert orc.workgroup_size_64(x) == want
    for x in (0, 1, 64):  # the reference panics here; the build defines (1, 1)
        assert orc.workgroup_size_64(x) == (1, 1)


def test_camera_constants(orc):  # camera.rs:11-30, camera_controller.rs:173-185, projection_matrix.rs:21-37
    cam, inv_proj, view = orc.shirley_camera(1920, 1080)
    assert abs(float(cam["pitch"][0]) - 1.7195947) < 2e-7
    assert abs(float(cam["yaw"][0]) - (-1.7975952)) < 2e-7
    assert abs(float(cam["defocus_radius"][0]) - 0.0523604) < 1e-7
    assert list(cam["position"][0]) == [13.0, 2.0, 3.0, 1.0]
    h = inv_proj[5]
    assert abs(float(h) - 0.17632698) < 1e-8  # tan(10 deg)
    assert inv_proj[0] == np.float32(h * (np.float32(1920) / np.float32(1080)))
    assert inv_proj[14] == 1.0 and inv_proj[15] == np.float32(10.0) and inv_proj[10] == 0.0
    # view's columns: right, up, dir, position; dir points from the camera to the origin
    d = view[8:11]
    want = -np.array([13.0, 2.0, 3.0]) / np.linalg.norm([13.0, 2.0, 3.0])
    assert np.allclose(d, want, atol=1e-6)
    assert list(view[12:16]) == [13.0, 2.0, 3.0, 1.0]


def test_struct_layouts(orc):
    assert orc.SPHERE.itemsize == 32 and orc.MATERIAL.itemsize == 32 and orc.BVH_NODE.itemsize == 32
    assert orc.GPU_CAMERA.itemsize == 32 and orc.RAY.itemsize == 48 and orc.HIT.itemsize == 16
    assert orc.RAY.fields["direction"][1] == 16 and orc.RAY.fields["inv_direction"][1] == 32
    assert orc.RAY.fields["pixel_idx"][1] == 44 and orc.BVH_NODE.fields["aabb_max"][1] == 16
