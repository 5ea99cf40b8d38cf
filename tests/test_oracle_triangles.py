"""Triangle extension, CPU side: the oracle's BVH traversal against brute force, the product's host builder and
mesh generator against the oracle's, and a golden vector."""
import numpy as np

from conftest import assert_bit_equal
from helpers import make_mesh_oracle, mesh_inputs


def test_triangle_bvh_equals_brute_force(orc):
    w = h = 64
    o = make_mesh_oracle(orc, mesh_inputs(orc, w, h, 1000, 30.0), w, h)
    rng = np.random.default_rng(5)
    n = 15000
    rays = np.zeros(n, orc.RAY)
    rays["origin"][:, :3] = rng.uniform(-12, 12, (n, 3))
    rays["origin"][:, 3] = 1
    rays["direction"][:, :3] = rng.normal(size=(n, 3)) * rng.uniform(0.3, 2.0, (n, 1))
    rays["inv_direction"] = np.float32(1) / rays["direction"][:, :3]
    hits = 0
    for r in rays:
        hb, pb = o.trace_bvh(r)
        hf, pf = o.trace_brute(r)
        assert hb == hf
        if hb:
            hits += 1
            assert pb["t"] == pf["t"] and pb["sphere_idx"] == pf["sphere_idx"] and pb["mat_type"] == pf["mat_type"]
            assert pb["t"] > np.float32(0.001)
    assert hits > 2000
    o.close()


def test_moeller_trumbore_known_answers(orc):
    """One axis-aligned triangle: hits inside, misses outside / behind / parallel, degenerate triangle never hit."""
    tris = np.zeros(2, orc.TRIANGLE)
    tris[0]["v0"], tris[0]["e1"], tris[0]["e2"] = (0, 0, 0), (2, 0, 0), (0, 2, 0)
    tris[1]["v0"], tris[1]["e1"], tris[1]["e2"] = (5, 5, 0), (1, 1, 0), (2, 2, 0)  # degenerate (collinear)
    tris["material_type"] = [1, 2]
    tr, nodes = orc.build_bvh_triangles(tris, 8)
    cam, ip, vw = orc.mesh_camera(8, 8)
    o = orc.Oracle(8, 8, np.zeros(1, orc.SPHERE), np.zeros(3, orc.MATERIAL), nodes, cam, ip, vw, triangles=tr)

    def ray(org, d):
        r = np.zeros(1, orc.RAY)
        r["origin"][0] = (*org, 1.0)
        r["direction"][0] = (*d, 0.0)
        with np.errstate(divide="ignore"):
            r["inv_direction"][0] = np.float32(1) / np.asarray(d, "<f4")
        return r[0]

    hit, p = o.trace_bvh(ray((0.5, 0.5, 3.0), (0, 0, -1)))
    assert hit and p["t"] == np.float32(3.0) and p["mat_type"] == 1
    hit, p = o.trace_bvh(ray((0.5, 0.5, 3.0), (0, 0, -2)))  # non-unit direction: t scales
    assert hit and p["t"] == np.float32(1.5)
    assert not o.trace_bvh(ray((1.5, 1.5, 3.0), (0, 0, -1)))[0]   # u + v > 1
    assert not o.trace_bvh(ray((0.5, 0.5, 3.0), (0, 0, 1)))[0]    # behind the origin
    assert not o.trace_bvh(ray((0.5, 0.5, 3.0), (1, 0, 0)))[0]    # parallel to the plane
    assert not o.trace_bvh(ray((0.5, 0.5, 0.0005), (0, 0, -1)))[0]  # t below t_min = 0.001 (extend.wgsl:90)
    assert not o.trace_brute(ray((5.5, 5.5, 3.0), (0, 0, -1)))[0]  # degenerate triangle
    o.close()


def test_host_mesh_and_builder_match_oracle(wf, orc):
    for n, bins in ((1, 8), (2, 2), (777, 16), (20000, 32)):
        s = wf.Scene.random_mesh(n, seed=3)
        tr, mt = orc.scene_random_mesh(n, seed=3)
        assert s.triangles.tobytes() == tr.tobytes() and s.materials.tobytes() == mt.tobytes()
        tree = wf.BVHTree(n)
        tree.build_bvh_tree_triangles(s.triangles, bins)
        tr2, nodes = orc.build_bvh_triangles(tr, bins)
        assert tree.nodes.tobytes() == nodes.tobytes() and s.triangles.tobytes() == tr2.tobytes()
    assert (tr["material_type"] == np.arange(n) % 3).all() and mt["material_type"].tolist() == [0, 1, 2]


def test_mesh_render_properties(orc):
    w, h = 96, 64
    o = make_mesh_oracle(orc, mesh_inputs(orc, w, h, 5000, 8.0), w, h, max_wavefronts=6)
    acc = o.render(2)
    t = o.bounce_table().astype(np.int64)
    assert (t[:, 1] + t[:, 2] == t[:, 0]).all() and (t[1:, 0] == t[:-1, 1]).all() and t[0, 1] > 500
    assert np.isfinite(acc).all() and acc.min() >= 0
    a = make_mesh_oracle(orc, mesh_inputs(orc, w, h, 5000, 8.0), w, h, max_wavefronts=6, serial=True)
    assert_bit_equal(a.render(2), acc, "serial vs OpenMP, mesh")
    a.close(); o.close()


def test_golden_mesh(orc):
    import hashlib
    import os
    for mode in (0, 1):
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"mesh5000_200x120_mode{mode}.npz"))
        w, h = int(g["width"]), int(g["height"])
        o = make_mesh_oracle(orc, mesh_inputs(orc, w, h, 5000, 8.0), w, h, max_wavefronts=int(g["bounces"]), rng_mode=mode)
        acc = o.render(int(g["spp"]))
        assert hashlib.sha256(acc.tobytes()).hexdigest() == str(g["acc_sha256"])
        assert np.array_equal(o.bounce_table(), g["table"]) and np.array_equal(o.totals(), g["totals"])
        o.close()
