"""Device BVH builder (csrc/wfpt_bvh_build.hip) against the host builder, which is itself
compared with the oracle's restatement of bvh.rs in test_host_model.py / test_oracle_triangles.py. Bar: the node
array and the reordered primitives are byte-identical."""
import time

import numpy as np
import pytest

from conftest import assert_bit_equal


def both_sphere_builds(W, spheres):
    a, b = spheres.copy(), spheres.copy()
    host, dev = W.BVHTree(len(a)), W.BVHTree(len(b))
    host.build_bvh_tree(a)
    dev.build_bvh_tree(b, device=0)
    return (a, host), (b, dev)


def both_mesh_builds(W, tris, n_bins):
    a, b = tris.copy(), tris.copy()
    host, dev = W.BVHTree(len(a)), W.BVHTree(len(b))
    t0 = time.time()
    host.build_bvh_tree_triangles(a, n_bins)
    host.host_s = time.time() - t0
    dev.build_bvh_tree_triangles(b, n_bins, device=0)
    return (a, host), (b, dev)


def check(host_pair, dev_pair, what):
    (hp, host), (dp, dev) = host_pair, dev_pair
    assert len(dev.nodes) == len(host.nodes), f"{what}: node count {len(dev.nodes)} vs {len(host.nodes)}"
    assert_bit_equal(dev.nodes.view("<u4").reshape(-1, 8), host.nodes.view("<u4").reshape(-1, 8), f"{what}: nodes")
    assert dp.tobytes() == hp.tobytes(), f"{what}: primitive order"


def test_no_device_is_an_error_not_a_fallback(wf):
    """Without a GPU the device builder must fail loudly (this test also runs on the CPU-only container)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sc = wf.Scene.new()
    with pytest.raises(wf.WfptError) as e:
        wf.BVHTree(len(sc.spheres)).build_bvh_tree(sc.spheres.copy(), device=0)
    assert e.value.status == wf.ERR_NO_DEVICE


@pytest.mark.gpu
def test_sphere_scenes(gpu):
    W = gpu
    check(*both_sphere_builds(W, W.Scene.new().spheres), "5-sphere scene")  # one wave-free path: 4096 bins
    for seed in (1, 2, 3):
        check(*both_sphere_builds(W, W.Scene.book_one_final(seed).spheres), f"book_one_final seed {seed}")
    one = W.Scene.new().spheres[:1].copy()
    check(*both_sphere_builds(W, one), "a single sphere")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 129, 1000, 5000, 100000])
def test_meshes_32_bins(gpu, n):
    W = gpu
    check(*both_mesh_builds(W, W.Scene.random_mesh(n, seed=n).triangles, 32), f"{n} triangles")


@pytest.mark.gpu
@pytest.mark.parametrize("n_bins", [2, 3, 8, 63, 64, 65, 100, 1000, 4096])
def test_bin_counts(gpu, n_bins):
    """<= 64 bins: level path + one wave per small subtree; more: the level path all the way down."""
    W = gpu
    check(*both_mesh_builds(W, W.Scene.random_mesh(700, seed=5).triangles, n_bins), f"{n_bins} bins")


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_bins", [(40, 32), (3000, 32), (3000, 100)])
def test_degenerate_inputs(gpu, n, n_bins):
    """Coincident primitives: every centre falls into one bin, the partition puts everything on one side
    (bvh.rs:188-190 returns AFTER the swap loop has rotated the primitives) -- in the wave path and in the level path."""
    W = gpu
    tris = W.Scene.random_mesh(n, seed=9).triangles
    same = tris.copy()
    same[:] = tris[0]
    same["material_idx"] = np.arange(n) % 3  # tell the copies apart
    check(*both_mesh_builds(W, same, n_bins), "identical triangles")
    two = tris.copy()
    two[: n // 3] = tris[0]
    two[n // 3:] = tris[1]
    two["material_idx"] = np.arange(n) % 3
    check(*both_mesh_builds(W, two, n_bins), "two clusters of identical triangles")
    flat = tris.copy()
    for f in ("v0", "e1", "e2"):
        flat[f][:, 1] = 0.0  # everything in the plane y = 0: one axis is narrower than 1e-5 and is skipped
    check(*both_mesh_builds(W, flat, n_bins), "flat mesh")
    zeros = tris.copy()
    rng = np.random.default_rng(3)
    for f in ("v0", "e1", "e2"):  # +0 / -0 mixes in every box: the builders fix min -> -0, max -> +0 (wfpt_host.cpp zmin / zmax)
        zeros[f][:, 2] = np.where(rng.random(n) < 0.5, np.float32(0.0), np.float32(-0.0))
    check(*both_mesh_builds(W, zeros, n_bins), "signed zeros")


@pytest.mark.gpu
def test_config5_mesh_and_render(gpu):
    """BASELINE config 5 (1 000 000 triangles): same tree as the host builder, and a context created from the
    device-built tree renders the same image."""
    W = gpu
    host_pair, dev_pair = both_mesh_builds(W, W.Scene.random_mesh(1000000, seed=1).triangles, 32)
    check(host_pair, dev_pair, "1M triangles")
    print(f"1M triangles: host builder {host_pair[1].host_s:.2f} s, device builder {dev_pair[1].device_ms:.1f} ms")
    a = W.mesh_path_tracer(320, 200, 20000, max_wavefronts=4)
    b = W.mesh_path_tracer(320, 200, 20000, max_wavefronts=4, device_bvh=True)
    a.render(2); b.render(2)
    assert_bit_equal(a.accumulated(), b.accumulated(), "render from a device-built BVH")
    a.close(); b.close()
    c = W.shirley_path_tracer(200, 120, max_wavefronts=4, device_bvh=True)
    d = W.shirley_path_tracer(200, 120, max_wavefronts=4)
    c.render(2); d.render(2)
    assert_bit_equal(c.accumulated(), d.accumulated(), "render from a device-built sphere BVH")
    c.close(); d.close()
