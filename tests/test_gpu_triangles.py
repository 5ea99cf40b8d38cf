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
    for batch in (1, 4):
        pt = make_mesh_tracer(W, w, h, n_tris, scale, max_wavefronts=bounces, batch=batch)
        pt.render(spp)
        assert_bit_equal(pt.accumulated(), want, f"mesh image, batch {batch}")
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
