"""A tree deeper than 31 levels, in LDS. The LDS walk keeps one pending bit per tree level: 32 bits for trees up to 31 levels deep -- the
variant whose inner-node loop is hand-written assembly (descend_asm) -- and 64 bits beyond, which runs the C++ statement of the same loop.
The reference's builder never makes such a tree out of float coordinates that fit (its binned SAH peels several primitives per split), but
the C ABI takes any tree in bvh.rs's layout: here a hand-built chain, every inner node = (one sphere | the rest), 40 levels. Both walks
(default and WFPT_FLAG_EXACT_TRAVERSAL), the fused loop, the class-binned loop and the stage kernels must give the oracle's image on it, and
the same chain cut to 20 levels (the 32-bit variant, i.e. the assembly) too."""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


def chain_scene(O, n):
    """n spheres in a row in front of the camera of main.rs:21-22 and the chain BVH over them, in bvh.rs's layout (root 0, slot 1 unused,
    siblings at (2k, 2k + 1)); leaf boxes exactly as sphere.rs:22-26 computes them, inner boxes their unions."""
    sp = np.zeros(n, O.SPHERE)
    mt = np.zeros(3, O.MATERIAL)
    mt["albedo"][0, :3] = (0.8, 0.3, 0.3); mt["albedo"][1, :3] = (0.8, 0.8, 0.8); mt["albedo"][2, :3] = (1.0, 1.0, 1.0)
    mt["fuzz"][1] = 0.1
    mt["refract_index"][2] = 1.5
    for i in range(n):
        sp["center"][i] = (np.float32(-0.45 * (n - 1) / 2 + 0.45 * i), np.float32(0.15 * ((i * 7) % 5 - 2)), np.float32(-3.0 - 0.1 * (i % 3)), 1.0)
        sp["radius"][i] = np.float32(0.2 + 0.01 * (i % 4))
        sp["material_idx"][i] = i % 3
        sp["material_type"][i] = i % 3
    lo = (sp["center"][:, :3] - sp["radius"][:, None]).astype(np.float32)
    hi = (sp["center"][:, :3] + sp["radius"][:, None]).astype(np.float32)
    nodes = np.zeros(2 * n, O.BVH_NODE)
    # inner node k (k = 0 .. n-2) holds spheres k .. n-1: children (leaf of sphere k | inner node k + 1, or the last leaf)
    inner = [0] + [2 * k + 1 for k in range(1, n - 1)]  # node index of inner node k
    for k in range(n - 1):
        me, left = inner[k], 2 * (k + 1)
        nodes["left_first"][me], nodes["prim_count"][me] = left, 0
        nodes["aabb_min"][me], nodes["aabb_max"][me] = lo[k:].min(axis=0), hi[k:].max(axis=0)
        nodes["left_first"][left], nodes["prim_count"][left] = k, 1
        nodes["aabb_min"][left], nodes["aabb_max"][left] = lo[k], hi[k]
    last = 2 * (n - 1) + 1
    nodes["left_first"][last], nodes["prim_count"][last] = n - 1, 1
    nodes["aabb_min"][last], nodes["aabb_max"][last] = lo[n - 1], hi[n - 1]
    return sp, mt, nodes


@pytest.mark.parametrize("n", [21, 41])  # 20 levels: the 32-bit trail (hand-written loop); 40 levels: the 64-bit trail (the C++ loop)
@pytest.mark.parametrize("rng_mode", [0, 1])
def test_chain_tree(gpu, orc, n, rng_mode):
    W, O = gpu, orc
    w, h, spp, bounces = 256, 144, 3, 6
    sp, mt, nodes = chain_scene(O, n)
    cam, ip, vw = O.camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), 90.0, 0.0, 10.0, 0.1, 100.0, w, h)
    o = O.Oracle(w, h, sp, mt, nodes, cam, ip, vw, max_wavefronts=bounces, rng_mode=rng_mode, miss_floor=0)
    want = o.render(spp)
    assert want.sum() > 0
    cc = W.CameraController(W.Camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0)), 90.0, 0.0, 10.0, 0.1, 100.0)
    flag_sets = [0, W.FLAG_EXACT_TRAVERSAL, W.FLAG_UNFUSED, W.FLAG_NO_GRAPH] + ([W.FLAG_BINNING] if rng_mode == W.RNG_PIXEL else [])
    for flags in flag_sets:
        tree = W.BVHTree(n)
        tree.nodes = nodes.view(W.BVH_NODE).copy()
        scene = W.Scene(sp.view(W.SPHERE).copy(), mt.view(W.MATERIAL).copy())
        pt = W.PathTracer(scene, W.RenderParameters(cc, (w, h)), max_wavefronts=bounces, rng_mode=rng_mode, miss_floor=0, flags=flags, bvh=tree, batch=2)
        pt.render(spp)
        assert np.array_equal(pt.bounce_table(), o.bounce_table()), f"{n} spheres, flags {flags}"
        assert_bit_equal(pt.accumulated(), want, f"chain of {n} spheres, flags {flags}, mode {rng_mode}")
        pt.close()
    o.close()
