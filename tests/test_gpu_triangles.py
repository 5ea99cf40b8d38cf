"""Build extension: triangle meshes (north-star's "ray-triangle intersect", BASELINE config 5). The reference
has no triangle code (spheres only, extend.wgsl:185-210), so these results are pinned by the oracle alone --
"parity unpinned" by the reference -- and the oracle's BVH traversal is itself cross-checked against a brute-
force loop over all triangles (tests/test_oracle_triangles.py)."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from helpers import make_mesh_oracle, make_mesh_tracer, mesh_inputs

pytestmark = pytest.mark.gpu

# (triangles, edge scale): 300 fits a CU's LDS (LDS-resident variant of extend); 30 000 does not
# (nodes + triangles = 3.4 MB: the HBM / Infinity-Cache variant, the one the 1M-triangle config runs)
MESHES = [(300, 40.0), (30000, 6.0)]


@pytest.mark.parametrize("n_tris,scale", MESHES)
@pytest.mark.parametrize("rng_mode", [0, 1])
def test_mesh_stage_by_stage(gpu, orc, n_tris, scale, rng_mode):
    W = gpu
    w, h = 160, 96
    n = w * h
    o = make_mesh_oracle(orc, mesh_inputs(orc, w, h, n_tris, scale), w, h, rng_mode=rng_mode)
    pt = make_mesh_tracer(W, w, h, n_tris, scale, rng_mode=rng_mode)
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 2)); o.set_frame(2, 0)
    pt.reset_image(); o.reset_image()
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    pt.generate_ray_kernel.run((w // 8, h // 8)); o.generate_rays(w // 8, h // 8, False)
    assert_bit_equal(pt.rays(n), o.rays(n).view(W.RAY), "rays")
    ext = W.workgroup_size_64(n)
    for wavefront in range(3):
        pt.extend_kernel.run(ext); o.extend(*ext)
        c = o.counters()
        assert np.array_equal(pt.read_counters()[:3], c[:3])
        misses, hits = int(c[0]), int(c[1])
        assert wavefront > 0 or hits > 200, "the test mesh must actually be hit"
        assert_bit_equal(pt.hits(hits), o.hits(hits).view(W.HIT), f"hit queue {wavefront}")
        assert_bit_equal(pt.misses(misses), o.misses(misses), f"miss queue {wavefront}")
        c[2] = 0
        pt.set_counters(c); o.set_counters(c)
        sh, ms = W.workgroup_size_64(hits), W.workgroup_size_64(misses)
        pt.shade_kernel.run(sh); o.shade(*sh)
        assert_bit_equal(pt.extension_rays(hits), o.extension_rays(hits).view(W.RAY), f"extension rays {wavefront}")
        pt.miss_kernel.run(ms); o.miss(*ms)
        assert_bit_equal(pt.image(), o.image(), f"image {wavefront}")
        pt.swap_ray_queues(); o.swap_ray_queues()
        ext = W.workgroup_size_64(hits)
        pt.set_counters([0, 0, hits, 0]); o.set_counters([0, 0, hits, 0])
    pt.close(); o.close()


@pytest.mark.parametrize("n_tris,scale", MESHES)
@pytest.mark.parametrize("w,h", [(200, 120), (203, 125)])  # the second size leaves inactive padding rays in the queue
def test_mesh_device_loop(gpu, orc, n_tris, scale, w, h):
    W = gpu
    spp, bounces = 5, 6
    o = make_mesh_oracle(orc, mesh_inputs(orc, w, h, n_tris, scale), w, h, max_wavefronts=bounces)
    want = o.render(spp)
    # fused bounce launches / stage kernels one by one; four-wide collapsed tree (default for scenes beyond LDS) / the caller's
    # binary tree as it is: all four must give the oracle's image
    # (scenes beyond LDS additionally trace with dynamic lane refill by default; WFPT_FLAG_NO_REFILL = the fused bounce kernel)
    for batch, flags in ((1, 0), (4, 0), (4, W.FLAG_NO_REFILL), (4, W.FLAG_BINARY_BVH), (4, W.FLAG_UNFUSED),
                         (2, W.FLAG_UNFUSED | W.FLAG_BINARY_BVH), (5, W.FLAG_NO_REFILL)):
        pt = make_mesh_tracer(W, w, h, n_tris, scale, max_wavefronts=bounces, batch=batch, flags=flags)
        pt.render(spp)
        assert_bit_equal(pt.accumulated(), want, f"mesh image, batch {batch}, flags {flags}")
        assert np.array_equal(pt.totals(), o.totals())
        assert np.array_equal(pt.bounce_table(), o.bounce_table())
        pt.close()
    o.close()


def test_golden_mesh_on_gpu(gpu):
    import os
    W = gpu
    for mode in (0, 1):
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"mesh5000_200x120_mode{mode}.npz"))
        pt = make_mesh_tracer(W, int(g["width"]), int(g["height"]), 5000, 8.0, max_wavefronts=int(g["bounces"]), rng_mode=mode)
        pt.render(int(g["spp"]))
        assert_bit_equal(pt.accumulated(), g["acc"], "golden mesh image")
        assert np.array_equal(pt.bounce_table(), g["table"])
        pt.close()


def test_hbm_mesh_per_material_split(gpu):
    """Per-material hit lists (WFPT_FLAG_SPLIT_SHADE) written by the HBM-scene extend give the unsplit image."""
    W = gpu
    a = make_mesh_tracer(W, 320, 200, 30000, 6.0, max_wavefronts=5)
    b = make_mesh_tracer(W, 320, 200, 30000, 6.0, max_wavefronts=5, flags=W.FLAG_SPLIT_SHADE)
    a.render(3); b.render(3)
    assert np.array_equal(a.bounce_table(), b.bounce_table())
    assert_bit_equal(a.accumulated(), b.accumulated(), "split vs unified shade on an HBM-resident mesh")
    a.close(); b.close()


def test_obj_scene_end_to_end(gpu, orc, tmp_path):
    """README.md:25: an OBJ file -> Scene.from_obj -> BVH built on the device -> the kernel chain; against the oracle
    given the same triangles (its own BVH builder, its own traversal)."""
    W = gpu
    # a wavy height field of 48 x 48 quads facing the camera of the mesh scene (0, 0, 30) -> origin
    n = 48
    xs = np.linspace(-9, 9, n + 1)
    lines = []
    for j in range(n + 1):
        for i in range(n + 1):
            lines.append(f"v {xs[i]:.6f} {xs[j]:.6f} {1.5 * np.sin(0.7 * xs[i]) * np.cos(0.5 * xs[j]):.6f}")
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i + 1
            lines.append(f"f {a} {a + 1} {a + n + 2} {a + n + 1}")
    path = tmp_path / "wave.obj"
    path.write_text("\n".join(lines) + "\n")
    mt = np.zeros(1, W.MATERIAL)
    mt["albedo"][0] = (0.8, 0.6, 0.3, 1.0)
    mt["fuzz"][0] = 0.05
    mt["material_type"][0] = 1  # metal: bounces between the waves
    scene = W.Scene.from_obj(str(path), mt)
    assert len(scene.triangles) == 2 * n * n and np.all(scene.triangles["material_type"] == 1)
    w, h, spp, bounces = 240, 160, 3, 5
    cc = W.CameraController(W.Camera((0.0, 0.0, 30.0), (0.0, 0.0, 0.0)), 40.0, 0.0, 10.0, 0.1, 100.0)
    tris_for_oracle = scene.triangles.copy()  # PathTracer reorders scene.triangles while building the BVH
    pt = W.PathTracer(scene, W.RenderParameters(cc, (w, h)), max_wavefronts=bounces, device_bvh=True)
    pt.render(spp)
    sorted_tris, nodes = orc.build_bvh_triangles(tris_for_oracle.view(orc.TRIANGLE), 32)
    assert nodes.tobytes() == pt.bvh_tree.nodes.tobytes()
    cam, ip, vw = orc.mesh_camera(w, h)
    o = orc.Oracle(w, h, np.zeros(1, orc.SPHERE), mt.view(orc.MATERIAL), nodes, cam, ip, vw, triangles=sorted_tris, max_wavefronts=bounces)
    assert_bit_equal(pt.accumulated(), o.render(spp), "OBJ scene")
    assert pt.bounce_table()[0, 1] > w * h // 4  # the sheet fills a good part of the frame
    pt.close(); o.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_config5_full_size_against_golden(gpu, mode):
    """BASELINE config 5 at its stated size: the seeded 1 000 000-triangle soup, 1920x1080, 8 bounces, BVH built on the
    device. One sample against the oracle's committed golden (tests/golden/mesh1m_1920x1080_mode*.npz, made by
    tests/golden/make_golden.py config5): per-bounce (rays, hits, misses) table, totals, SHA-256 of the accumulated
    image, its 16x down-sampled copy; then the chain's size-independent properties on further samples. Both the fused
    bounce launches and the stage kernels one by one must give the golden image."""
    import hashlib
    import os
    W = gpu
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"mesh1m_1920x1080_mode{mode}.npz"))
    w, h, bounces, n_tri = int(g["width"]), int(g["height"]), int(g["bounces"]), int(g["n_triangles"])
    # (WFPT_FLAG_EXACT_TRAVERSAL: the reference's own walk over the caller's binary tree for every ray of the frame -- the four-wide
    # free walk with its leaf-box verdicts and hand-overs, and the walk it hands over to, give the same golden)
    for flags in (0, W.FLAG_UNFUSED, W.FLAG_EXACT_TRAVERSAL):
        pt = W.mesh_path_tracer(w, h, n_tri, seed=1, max_wavefronts=bounces, rng_mode=mode, device_bvh=True, batch=2, flags=flags)
        assert len(pt.bvh_tree.nodes) == int(g["n_nodes"])
        pt.render_sample()
        t = pt.bounce_table().astype(np.int64)
        assert np.array_equal(t, g["table"]), f"per-bounce table, flags={flags}"
        acc = pt.accumulated()
        assert np.array_equal(pt.totals(), g["totals"])
        assert hashlib.sha256(acc.tobytes()).hexdigest() == str(g["acc_sha256"]), f"image hash, flags={flags}"
        f = int(g["downsample"])
        small = acc.reshape(h, w, 3)[:(h // f) * f, :(w // f) * f].reshape(h // f, f, w // f, f, 3).mean(axis=(1, 3))
        assert np.allclose(small, g["acc_small"], rtol=1e-5, atol=1e-6)
        # size-independent properties of the chain
        assert t[0, 0] == w * h and (t[:, 1] + t[:, 2] == t[:, 0]).all() and (t[1:, 0] == t[:-1, 1]).all()
        assert np.isfinite(acc).all() and acc.min() >= 0.0
        if flags == 0:  # two more samples in one batch: accumulation is additive and frame-ordered
            pt.render(2)
            acc3 = pt.accumulated()
            assert (acc3 >= acc).all() and np.isfinite(acc3).all()
            assert int(pt.totals()[0]) > 2 * int(g["totals"][0])
            wt = pt.wavefront_totals()
            assert np.array_equal(wt.sum(axis=0), pt.totals()) and (wt[1:, 0] == wt[:-1, 1]).all()
        pt.close()
