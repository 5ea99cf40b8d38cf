"""GPU tests added in round 3: the run-time switch back to the reference's traversal arithmetic (WFPT_FLAG_EXACT_TRAVERSAL),
rays built to sit on the edges of the conservative box test's argument, dynamic scenes (wfpt_update_scene), the 1920x1080
frame as 8 band shards against the committed golden, gather buffers across a viewport change, wfpt_render_chunked's edges.
Everything through the C ABI, bit for bit against the oracle (tolerance: 0 ULP)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import assert_bit_equal
from helpers import (adversarial_mesh as _adversarial_mesh, adversarial_rays as _adversarial_rays, adversarial_rays_mesh as _adversarial_rays_mesh,
                     close_pairs_scene as _close_pairs_scene, inputs_for, make_mesh_oracle, make_mesh_tracer, make_oracle, make_tracer,
                     mesh_inputs)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ WFPT_FLAG_EXACT_TRAVERSAL
@pytest.mark.parametrize("kind,w,h,spp,bounces", [("simple", 128, 72, 3, 5), ("shirley", 400, 224, 3, 6), ("shirley", 200, 123, 2, 50)])
@pytest.mark.parametrize("rng_mode", [0, 1])
@pytest.mark.parametrize("loop", ["fused", "unfused", "no_lds"])
def test_exact_traversal_equals_conservative_and_oracle(gpu, orc, kind, w, h, spp, bounces, rng_mode, loop):
    """The reference's box test operation for operation (extend.wgsl:124,164-183: (b - o) * inv slabs, 1e30 for a missed
    box, hence its descent into doubly-missed pairs) and the default conservative test give the same image, tables and
    totals -- the oracle's."""
    W = gpu
    base = {"fused": 0, "unfused": W.FLAG_UNFUSED, "no_lds": W.FLAG_NO_LDS_SCENE}[loop]
    o = make_oracle(orc, inputs_for(orc, kind, w, h), w, h, rng_mode=rng_mode, max_wavefronts=bounces)
    want = o.render(spp)
    for flags in (base, base | W.FLAG_EXACT_TRAVERSAL):
        pt = make_tracer(W, kind, w, h, rng_mode=rng_mode, max_wavefronts=bounces, flags=flags, batch=2)
        pt.render(spp)
        assert np.array_equal(pt.bounce_table(), o.bounce_table()), f"flags={flags}"
        assert np.array_equal(pt.totals(), o.totals())
        assert_bit_equal(pt.accumulated(), want, f"{kind} {loop} flags={flags}")
        pt.close()
    o.close()


@pytest.mark.parametrize("exact", [False, True])
def test_exact_traversal_on_an_lds_resident_mesh(gpu, orc, exact):
    """Triangles reach the corners of their boxes: the conservative test (boxes grown by more than its rounding error) and
    the exact one must both give the oracle's image on a mesh that lives in LDS."""
    W = gpu
    w, h, n_tri = 200, 120, 1500
    inputs = mesh_inputs(orc, w, h, n_tri, edge_scale=20.0)
    o = make_mesh_oracle(orc, inputs, w, h, max_wavefronts=5)
    want = o.render(3)
    pt = make_mesh_tracer(W, w, h, n_tri, edge_scale=20.0, max_wavefronts=5, flags=W.FLAG_EXACT_TRAVERSAL if exact else 0)
    pt.render(3)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), want, "LDS-resident mesh")
    pt.close(); o.close()


# ------------------------------------------------------------------ rays aimed at the edges of the argument (builders: helpers.py)
@pytest.mark.parametrize("scene", ["shirley", "pairs"])
@pytest.mark.parametrize("exact", [False, True, "hbm-binary"])
def test_adversarial_rays_against_the_oracle(gpu, orc, scene, exact):
    """exact = "hbm-binary" (ADVICE r3): the scene kept out of LDS and walked as the caller's binary tree, WITHOUT
    WFPT_FLAG_EXACT_TRAVERSAL -- a context must pick the reference's own walk for it by itself (decide_exact), because the binary
    walk without the reference's blind descent loses hits on exactly these rays."""
    W = gpu
    w, h = 128, 64
    walk_flags = (W.FLAG_NO_LDS_SCENE | W.FLAG_BINARY_BVH) if exact == "hbm-binary" else (W.FLAG_EXACT_TRAVERSAL if exact else 0)
    if scene == "shirley":
        inputs = inputs_for(orc, "shirley", w, h)
        pt = make_tracer(W, "shirley", w, h, flags=walk_flags)
    else:
        sp, mt = _close_pairs_scene(orc)
        sp_o, nodes = orc.build_bvh(sp.copy())
        cam, ip, vw = orc.shirley_camera(w, h)
        inputs = (sp_o, mt, nodes, cam, ip, vw)
        cc = W.CameraController(W.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
        pt = W.PathTracer(W.Scene(sp.view(W.SPHERE).copy(), mt.view(W.MATERIAL)), W.RenderParameters(cc, (w, h)), flags=walk_flags)
        assert_bit_equal(pt.bvh_tree.nodes, nodes.view(W.BVH_NODE), "host BVH of the pairs scene")
    o = make_oracle(orc, inputs, w, h)
    n_max = w * h
    rays = _adversarial_rays(W, inputs[0], inputs[2], n_max)
    n = len(rays)
    assert n > 2000
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 1)); o.set_frame(1, 0)
    pt.write_rays(rays); o.write_rays(rays.view(orc.RAY))
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    ext = W.workgroup_size_64(n)
    pt.extend_kernel.run(ext); o.extend(*ext)
    c = o.counters()
    assert np.array_equal(pt.read_counters()[:3], c[:3]), f"{pt.read_counters()[:3]} vs {c[:3]}"
    misses, hits = int(c[0]), int(c[1])
    assert hits > n // 10 and misses > n // 50
    assert_bit_equal(pt.hits(hits), o.hits(hits).view(W.HIT), f"hit queue of the adversarial rays ({scene}, exact={exact})")
    assert_bit_equal(pt.misses(misses), o.misses(misses), "miss queue of the adversarial rays")
    pt.close(); o.close()


@pytest.mark.parametrize("where", ["lds", "hbm", "hbm-binary"])
@pytest.mark.parametrize("exact", [False, True])
def test_adversarial_rays_on_a_mesh(gpu, orc, where, exact):
    """The same for triangles (a build extension; the oracle is the only checker): a mesh with exact duplicates (bit-equal
    distances: the visit order decides), shared edges and triangles whose boxes have no thickness; rays at vertices, edge
    midpoints and centroids, axis-parallel, in the planes of triangle boxes and along node-box edges. Default walk (conservative
    inner boxes / quantised four-wide nodes + exact leaf test + hand-over) and the reference's walk, LDS-resident and from HBM."""
    W = gpu
    w, h, n_tri = 128, 64, 1200
    tris, mt = orc.scene_random_mesh(n_tri, 1)
    tris["e1"] *= np.float32(12.0); tris["e2"] *= np.float32(12.0)
    _adversarial_mesh(tris)
    tris_o, nodes = orc.build_bvh_triangles(tris, 32)
    cam, ip, vw = orc.mesh_camera(w, h)
    o = orc.Oracle(w, h, np.zeros(1, orc.SPHERE), mt, nodes, cam, ip, vw, triangles=tris_o)
    scene = W.Scene.random_mesh(n_tri, 1)
    scene.triangles["e1"] *= np.float32(12.0); scene.triangles["e2"] *= np.float32(12.0)
    _adversarial_mesh(scene.triangles)
    cc = W.CameraController(W.Camera((0.0, 0.0, 30.0), (0.0, 0.0, 0.0)), 40.0, 0.0, 10.0, 0.1, 100.0)
    flags = (W.FLAG_EXACT_TRAVERSAL if exact else 0) | (W.FLAG_NO_LDS_SCENE if where != "lds" else 0) | \
            (W.FLAG_BINARY_BVH if where == "hbm-binary" else 0)  # hbm-binary without EXACT: decide_exact picks the reference's walk
    pt = W.PathTracer(scene, W.RenderParameters(cc, (w, h)), mesh_bins=32, flags=flags)
    assert_bit_equal(pt.bvh_tree.nodes, nodes.view(W.BVH_NODE), "host BVH of the adversarial mesh")
    rays = _adversarial_rays_mesh(W, tris_o, nodes, w * h)
    n = len(rays)
    assert n > 3000
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 1)); o.set_frame(1, 0)
    pt.write_rays(rays); o.write_rays(rays.view(orc.RAY))
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    ext = W.workgroup_size_64(n)
    pt.extend_kernel.run(ext); o.extend(*ext)
    c = o.counters()
    assert np.array_equal(pt.read_counters()[:3], c[:3]), f"{pt.read_counters()[:3]} vs {c[:3]}"
    misses, hits = int(c[0]), int(c[1])
    assert hits > n // 10 and misses > n // 50
    assert_bit_equal(pt.hits(hits), o.hits(hits).view(W.HIT), f"hit queue of the adversarial mesh rays ({where}, exact={exact})")
    assert_bit_equal(pt.misses(misses), o.misses(misses), "miss queue of the adversarial mesh rays")
    # and a short render of the same mesh through the device-resident loop
    want = o.render(2)
    pt.reset_progress()
    pt.render(2)
    assert_bit_equal(pt.accumulated(), want, f"adversarial mesh image ({where}, exact={exact})")
    pt.close(); o.close()


@pytest.mark.parametrize("where,n_tri,scale", [("lds", 1200, 12.0), ("hbm", 1200, 12.0), ("hbm", 30000, 3.0)])
def test_grazing_rays_on_a_mesh(gpu, orc, where, n_tri, scale):
    """Where Moeller-Trumbore is worst conditioned (helpers.grazing_rays_mesh: within 1e-7 .. 1e-2 rad of a triangle's
    plane, passing an edge at +-1e-7 .. 1e-3 edge lengths): the device's free walks (LDS-resident binary walk; four-wide walk from HBM)
    against the oracle's hit and miss queues, bit for bit. The CPU models of the same walks: tests/test_traversal_model.py."""
    from helpers import grazing_rays_mesh
    W = gpu
    w, h = 256, 128
    tris, mt = orc.scene_random_mesh(n_tri, 1)
    tris["e1"] *= np.float32(scale); tris["e2"] *= np.float32(scale)
    tris_o, nodes = orc.build_bvh_triangles(tris, 32)
    cam, ip, vw = orc.mesh_camera(w, h)
    o = orc.Oracle(w, h, np.zeros(1, orc.SPHERE), mt, nodes, cam, ip, vw, triangles=tris_o)
    scene = W.Scene.random_mesh(n_tri, 1)
    scene.triangles["e1"] *= np.float32(scale); scene.triangles["e2"] *= np.float32(scale)
    cc = W.CameraController(W.Camera((0.0, 0.0, 30.0), (0.0, 0.0, 0.0)), 40.0, 0.0, 10.0, 0.1, 100.0)
    pt = W.PathTracer(scene, W.RenderParameters(cc, (w, h)), mesh_bins=32, flags=W.FLAG_NO_LDS_SCENE if where == "hbm" else 0)
    assert_bit_equal(pt.bvh_tree.nodes, nodes.view(W.BVH_NODE), "host BVH")
    rays = grazing_rays_mesh(W, tris_o, w * h)
    n = len(rays)
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 1)); o.set_frame(1, 0)
    pt.write_rays(rays); o.write_rays(rays.view(orc.RAY))
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    ext = W.workgroup_size_64(n)
    pt.extend_kernel.run(ext); o.extend(*ext)
    c = o.counters()
    assert np.array_equal(pt.read_counters()[:3], c[:3]), f"{pt.read_counters()[:3]} vs {c[:3]}"
    misses, hits = int(c[0]), int(c[1])
    assert hits > n // 20 and misses > n // 20
    assert_bit_equal(pt.hits(hits), o.hits(hits).view(W.HIT), f"hit queue of the grazing rays ({where}, {n_tri} triangles)")
    assert_bit_equal(pt.misses(misses), o.misses(misses), "miss queue of the grazing rays")
    pt.close(); o.close()


def test_far_origins_switch_the_stage_extend_to_the_exact_test(gpu, orc):
    """Origins beyond four scene extents leave the range the conservative margin was sized for: wfpt_write_rays switches the
    context's extend to the reference's own box test; results stay the oracle's."""
    W = gpu
    w, h = 64, 64
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h)
    pt = make_tracer(W, "shirley", w, h)
    rng = np.random.default_rng(5)
    n = 4096
    rays = np.zeros(n, W.RAY)
    target = rng.uniform(-8, 8, (n, 3)).astype("<f4")
    target[:, 1] = np.abs(target[:, 1]) * 0.1
    origin = (target + rng.normal(size=(n, 3)).astype("<f4") * np.float32(5e4)).astype("<f4")  # tens of thousands of units away
    d = (target - origin).astype("<f4")
    rays["origin"][:, :3] = origin
    rays["origin"][:, 3] = 1.0
    rays["direction"][:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True).astype("<f4")
    rays["inv_direction"] = (np.float32(1.0) / rays["direction"][:, :3]).astype("<f4")
    rays["pixel_idx"] = np.arange(n) % 4096
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 1)); o.set_frame(1, 0)
    pt.write_rays(rays); o.write_rays(rays.view(orc.RAY))
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    ext = W.workgroup_size_64(n)
    pt.extend_kernel.run(ext); o.extend(*ext)
    c = o.counters()
    assert np.array_equal(pt.read_counters()[:3], c[:3])
    assert int(c[1]) > n // 4
    assert_bit_equal(pt.hits(int(c[1])), o.hits(int(c[1])).view(W.HIT), "hits of far-away origins")
    pt.close(); o.close()


@pytest.mark.parametrize("exact", [False, True])
def test_full_hd_seed_two(gpu, orc, exact):
    """One more scene than the golden's: seed 2 at 1920x1080, 2 spp, 8 bounces, both box tests, against the oracle."""
    W = gpu
    w, h, spp, bounces = 1920, 1080, 2, 8
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h, seed=2), w, h, max_wavefronts=bounces)
    want = o.render(spp)
    pt = make_tracer(W, "shirley", w, h, seed=2, max_wavefronts=bounces, flags=W.FLAG_EXACT_TRAVERSAL if exact else 0)
    pt.render(spp)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert np.array_equal(pt.totals(), o.totals())
    assert_bit_equal(pt.accumulated(), want, f"seed 2, exact={exact}")
    pt.close(); o.close()


# ------------------------------------------------------------------ 1920x1080 as 8 band shards (what 8 ranks render)
def test_full_hd_as_eight_band_shards_against_golden(gpu):
    """The frame of BASELINE configs 2 / 3 cut the way 8 ranks cut it (band k -> shard k % 8), rendered shard after shard
    behind the C ABI (wfpt_render_chunked, pixel-keyed RNG) and assembled: the committed golden of the UNSHARDED frame."""
    W = gpu
    g = np.load(os.path.join(GOLDEN, "shirley_1920x1080_mode1.npz"))
    w, h, spp, bounces = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["bounces"])
    cc = W.CameraController(W.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    acc = W.render_chunked(W.Scene.book_one_final(1), W.RenderParameters(cc, (w, h)), spp, 8, max_wavefronts=bounces, rng_mode=W.RNG_PIXEL)
    assert hashlib.sha256(acc.tobytes()).hexdigest() == str(g["acc_sha256"])


# ------------------------------------------------------------------ dynamic scenes
@pytest.mark.parametrize("flags", [0, "UNFUSED", "BINNING"])  # BINNING: the class-binned loop (cost classes are re-derived by upload_scene)
def test_update_scene_moves_fifty_spheres(gpu, orc, flags):
    """wfpt_update_scene: move 50 spheres of the live context's scene; the BVH is rebuilt on the device, the accumulation
    restarts at frame 1, and the image equals both a fresh context on the moved scene and the oracle."""
    W = gpu
    fl = getattr(W, "FLAG_" + flags) if flags else 0
    mode = W.RNG_PIXEL if flags == "BINNING" else W.RNG_DISPATCH  # the class-binned loop exists in the pixel-keyed mode only
    w, h, spp, bounces = 400, 224, 3, 6
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, flags=fl, rng_mode=mode)
    pt.render(2)
    before = pt.accumulated()
    moved = W.Scene.book_one_final(1)
    rng = np.random.default_rng(9)
    idx = 1 + rng.permutation(len(moved.spheres) - 1)[:50]  # not the ground sphere
    moved.spheres["center"][idx, 0] += rng.uniform(-0.4, 0.4, 50).astype("<f4")
    moved.spheres["center"][idx, 1] += rng.uniform(0.0, 0.8, 50).astype("<f4")
    moved.spheres["center"][idx, 2] += rng.uniform(-0.4, 0.4, 50).astype("<f4")
    for_update = W.Scene(moved.spheres.copy(), moved.materials.copy())
    for_fresh = W.Scene(moved.spheres.copy(), moved.materials.copy())
    sp_o, nodes_o = orc.build_bvh(moved.spheres.view(orc.SPHERE).copy())
    pt.update_scene(for_update)
    assert W.lib().wfpt_frame(pt.handle) == 0 and pt.accumulated().sum() == 0
    assert_bit_equal(for_update.spheres, sp_o.view(W.SPHERE), "spheres reordered by the device builder like bvh.rs does")
    pt.render(spp)
    cam, ip, vw = orc.shirley_camera(w, h)
    o = orc.Oracle(w, h, sp_o, moved.materials.view(orc.MATERIAL), nodes_o, cam, ip, vw, max_wavefronts=bounces, rng_mode=mode)
    want = o.render(spp)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), want, "after wfpt_update_scene vs the oracle")
    cc = W.CameraController(W.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    fresh = W.PathTracer(for_fresh, W.RenderParameters(cc, (w, h)), max_wavefronts=bounces, flags=fl, rng_mode=mode)
    fresh.render(spp)
    assert_bit_equal(pt.accumulated(), fresh.accumulated(), "after wfpt_update_scene vs a fresh context")
    assert not np.array_equal(before, pt.accumulated())
    # a different primitive count, and back
    small = W.Scene(for_fresh.spheres[:100].copy(), for_fresh.materials.copy())
    pt.update_scene(small)
    pt.render(1)
    assert np.isfinite(pt.accumulated()).all()
    with pytest.raises(W.WfptError):
        bad = W.Scene(for_fresh.spheres[:10].copy(), for_fresh.materials[:3].copy())  # material_idx out of range
        pt.update_scene(bad)
    pt.render(1)  # the refused update left the context usable
    fresh.close(); pt.close(); o.close()


def test_update_scene_mesh(gpu, orc):
    """The same for a triangle mesh: LDS-resident -> replaced by a perturbed mesh, against the oracle."""
    W = gpu
    w, h, n_tri = 200, 120, 1500
    pt = make_mesh_tracer(W, w, h, n_tri, edge_scale=20.0, max_wavefronts=5)
    pt.render(1)
    scene = W.Scene.random_mesh(n_tri, 2)
    scene.triangles["e1"] *= np.float32(15.0)
    scene.triangles["e2"] *= np.float32(15.0)
    tris_o, nodes_o = orc.build_bvh_triangles(scene.triangles.view(orc.TRIANGLE).copy(), 32)
    pt.update_scene(scene)
    pt.render(2)
    cam, ip, vw = orc.mesh_camera(w, h)
    o = orc.Oracle(w, h, np.zeros(1, orc.SPHERE), scene.materials.view(orc.MATERIAL), nodes_o, cam, ip, vw, triangles=tris_o, max_wavefronts=5)
    assert_bit_equal(pt.accumulated(), o.render(2), "mesh after wfpt_update_scene_mesh")
    pt.close(); o.close()


# ------------------------------------------------------------------ gather buffers across a viewport change
def test_gather_buffers_follow_the_viewport(gpu):
    """A wider, shorter viewport of about the same pixel count needs MORE whole-band floats than the gather buffers were
    allocated with at wfpt_comm_init (200x123 -> 2733x9: 2 bands of 2733 pixels x 8 rows): they are re-allocated by
    wfpt_update_render_parameters, and the gathered frame is the accumulated image."""
    W = gpu
    pt = make_tracer(W, "shirley", 200, 123, max_wavefronts=4, rng_mode=W.RNG_PIXEL, max_window_size=50000)
    pt.comm_init(W.comm_unique_id(), 0, 1)
    pt.render(2)
    pt.gather_accumulated()
    assert_bit_equal(pt.gathered(), pt.accumulated(), "before the resize")
    rp = pt.get_render_parameters()
    rp.set_viewport((2733, 9))
    pt.update_render_parameters(rp)
    pt.update_buffers()
    pt.render(2)
    pt.gather_accumulated()
    got = pt.gathered()
    assert got.shape == (2733 * 9, 3)
    assert_bit_equal(got, pt.accumulated(), "after the resize")
    pt.close()


# ------------------------------------------------------------------ wfpt_render_chunked's edges
def test_render_chunked_edges(gpu):
    W = gpu
    w, h, spp = 400, 225, 2  # 29 bands
    cc = W.CameraController(W.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    rp = W.RenderParameters(cc, (w, h))
    ref = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL, miss_floor=0)
    ref.render(spp)
    # more chunks than bands: the empty chunks are skipped
    got = W.render_chunked(W.Scene.book_one_final(1), rp, spp, 40, max_wavefronts=4, miss_floor=0)
    assert_bit_equal(got, ref.accumulated(), "40 chunks over 29 bands")
    ref.close()
    # a context that cannot be created reports ITS status, not a generic HIP error
    scene = W.Scene.book_one_final(1)
    bvh = W.BVHTree(len(scene.spheres))
    bvh.build_bvh_tree(scene.spheres)
    nodes = bvh.nodes.copy()
    nodes["left_first"][0] = 3  # children must sit at (2k, 2k+1)
    proj = W.ProjectionMatrix(cc.vfov_rad(), np.float32(w) / np.float32(h), *cc.get_clip_planes()).p_inv()
    view, cam = cc.get_view_matrix(), cc.get_GPU_camera()
    params = W._Params(w, h, 0, 4, 0, W.RNG_PIXEL, 0, 0, 1, 0, 0)
    out = np.zeros((w * h, 3), "<f4")
    st = W.lib().wfpt_render_chunked(C.byref(params), W._p(scene.spheres), len(scene.spheres), W._p(scene.materials), len(scene.materials),
                                     W._p(nodes), len(nodes), W._p(cam), W._p(proj), W._p(view), spp, 3, W._p(out))
    assert st == W.ERR_UNSUPPORTED and b"2k" in W.lib().wfpt_last_error(None)
