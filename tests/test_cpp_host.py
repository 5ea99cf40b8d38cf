"""The C++ host mirror (wavefront_path_tracer_amd/host/wfpt.hpp) compiled with g++ against the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from conftest import assert_bit_equal
from helpers import inputs_for, make_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(wf, tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "host_mirror"
    pkg = os.path.join(ROOT, "wavefront_path_tracer_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "host"),
           os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"), "-o", str(out), "-L", pkg, "-lwfpt", f"-Wl,-rpath,{pkg}",
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return str(out)


def test_cpp_data_model_matches_oracle(exe, orc, tmp_path):
    out = tmp_path / "model.bin"
    r = subprocess.run([exe, "model", str(out)], capture_output=True, text=True, check=True)
    assert r.stdout.split() == ["485", "970", "162", "200"]
    sp, mt = orc.scene_book_one_final(1)
    sp, nodes = orc.build_bvh(sp)
    cam, ip, vw = orc.shirley_camera(1920, 1080)
    want = sp.tobytes() + mt.tobytes() + nodes.tobytes() + cam.tobytes() + ip.tobytes() + vw.tobytes()
    assert out.read_bytes() == want


def test_cpp_camera_controller(exe, wf):
    """The C++ and Python mirrors of CameraController (camera_controller.rs:74-158) agree to the bit."""
    r = subprocess.run([exe, "controller", "-"], capture_output=True, text=True, check=True)
    cc = wf.CameraController(wf.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    cc.move_forward(1); cc.move_left(1); cc.move_up(1)
    cc.process_mouse((3.0, -2.0))
    cc.update_camera(0.25)
    cc.set_vfov(40.0)
    vals = np.array(list(cc.camera.position) + [cc.camera.pitch, cc.camera.yaw, cc.vfov_rad()], "<f4")
    want = [f"{u:08x}" for u in vals.view("<u4")] + ["1", "0"]
    assert r.stdout.split() == want


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["run", "render"])
def test_cpp_path_tracer(gpu, exe, orc, tmp_path, mode):
    """PathTracer::run() written like path_tracer.rs:279-371, in C++, over libwfpt.so: bit-equal to the oracle."""
    w, h, spp, bounces = 200, 120, 2, 5
    out = tmp_path / "acc.bin"
    r = subprocess.run([exe, mode, str(w), str(h), str(spp), str(bounces), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, "<f4").reshape(-1, 3)
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
    assert_bit_equal(got, o.render(spp), f"C++ PathTracer ({mode})")
    o.close()


@pytest.fixture(scope="module")
def c_example(wf, tmp_path_factory):
    """examples/wfpt_render.c: the C ABI from plain C99 (-Wall -Wextra -Werror), nothing but include/wfpt.h and libwfpt.so."""
    out = tmp_path_factory.mktemp("c") / "wfpt_render"
    pkg = os.path.join(ROOT, "wavefront_path_tracer_amd")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "wfpt_render.c"), "-o", str(out), "-L", pkg, "-lwfpt", f"-Wl,-rpath,{pkg}",
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True)
    return str(out)


def test_c_example_fails_loudly_without_a_gpu(c_example, wf, tmp_path):
    if wf.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([c_example, str(tmp_path / "x.ppm"), "64", "64", "1", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "x.ppm").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("device_bvh", [0, 1])
def test_c_example_renders_the_same_image(gpu, c_example, tmp_path, device_bvh):
    W = gpu
    w, h, spp, bounces = 200, 120, 3, 6
    out = tmp_path / "c.ppm"
    r = subprocess.run([c_example, str(out), str(w), str(h), str(spp), str(bounces), str(device_bvh)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    pt = W.shirley_path_tracer(w, h, max_wavefronts=bounces)
    pt.render(spp)
    ref = tmp_path / "py.ppm"
    pt.save_ppm(str(ref))
    pt.close()
    assert out.read_bytes() == ref.read_bytes()
