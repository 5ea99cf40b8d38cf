"""The class-binned fused loop (bounce_binned_kernel): every segment's hits stored sorted by cost class (the
dominant primitive | lambertian | metal | dielectric), a work item = 512 hits of ONE class. WFPT_RNG_PIXEL only (the order of the queue
is free there; round 4's dispatch-keyed variant was measured slower and removed in round 5), opt-in: WFPT_FLAG_BINNING asks for it (it was
the default for slabs of >= 3/4 Mpixel until the thread-ordered loop drew level at the end of round 5). It must give the oracle's image bit for bit, whatever the sizes, batches, tile shards and scene changes:
the class decides where a record is stored, never what is in it. The general parity tests (tests/test_gpu_parity.py) run the loop too (flags 128 / 256);
here are the cases that aim at its own machinery."""
import numpy as np
import pytest

from helpers import inputs_for, make_mesh_oracle, make_mesh_tracer, make_oracle, make_tracer, mesh_inputs

pytestmark = pytest.mark.gpu


def assert_bit_equal(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), what


@pytest.mark.parametrize("rng_mode", [1])
@pytest.mark.parametrize("w,h", [(8, 8), (72, 40), (520, 8), (1000, 600)])
def test_binned_loop_sizes(gpu, orc, rng_mode, w, h):
    """From one tile (every class's run shorter than a work item, most classes empty) to images whose classes span hundreds of
    segments; 1000 x 600 has partial tiles (inactive slots are thread indices without a hit)."""
    W = gpu
    bounces, spp = 6, 3
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, rng_mode=rng_mode, max_wavefronts=bounces, miss_floor=0)
    want = o.render(spp)
    pt = make_tracer(W, "shirley", w, h, rng_mode=rng_mode, max_wavefronts=bounces, miss_floor=0, flags=W.FLAG_BINNING, batch=2)
    pt.render(spp)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), want, f"{w}x{h} mode {rng_mode}")
    assert np.array_equal(pt.totals(), o.totals())
    pt.close(); o.close()


@pytest.mark.parametrize("rng_mode", [1])
def test_binned_loop_exits_like_the_reference(gpu, orc, rng_mode):
    """The reference's loop exit (misses < 128 => leave before shading, path_tracer.rs:332) inside the binned loop: the plan of the
    next launch must be empty for a sample that has left, while its neighbours in the batch go on."""
    W = gpu
    w, h, spp = 200, 120, 5
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, rng_mode=rng_mode, max_wavefronts=50)
    want = o.render(spp)
    pt = make_tracer(W, "shirley", w, h, rng_mode=rng_mode, max_wavefronts=50, flags=W.FLAG_BINNING, batch=4)
    pt.render(spp)
    assert_bit_equal(pt.accumulated(), want, "50 wavefronts, miss floor 128")
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    pt.close(); o.close()


@pytest.mark.parametrize("rng_mode", [1])
def test_binned_loop_on_the_five_sphere_scene_and_a_mesh_in_lds(gpu, orc, rng_mode):
    """scene.rs:12-46 has no dominant primitive by the quarter-of-the-root-box rule except its ground; a triangle mesh in LDS has none
    at all (classes = materials only) and three float4 per primitive in LDS beside the class table."""
    W = gpu
    o = make_oracle(orc, inputs_for(orc, "simple", 128, 72), 128, 72, rng_mode=rng_mode, max_wavefronts=6)
    want = o.render(4)
    pt = make_tracer(W, "simple", 128, 72, rng_mode=rng_mode, max_wavefronts=6, flags=W.FLAG_BINNING)
    pt.render(4)
    assert_bit_equal(pt.accumulated(), want, "five spheres")
    pt.close(); o.close()
    w, h, n_tri = 200, 120, 1500
    inputs = mesh_inputs(orc, w, h, n_tri, edge_scale=20.0)
    o = make_mesh_oracle(orc, inputs, w, h, max_wavefronts=5, rng_mode=rng_mode)
    want = o.render(3)
    pt = make_mesh_tracer(W, w, h, n_tri, edge_scale=20.0, max_wavefronts=5, rng_mode=rng_mode, flags=W.FLAG_BINNING)
    pt.render(3)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), want, "LDS-resident mesh")
    pt.close(); o.close()


def test_binned_and_thread_ordered_loops_agree_at_full_size(gpu):
    """1920x1080, 8 bounces, 3 samples: the binned loop against the thread-ordered one (the default, which the goldens pin; WFPT_FLAG_NO_BINNING
    names it and wins over WFPT_FLAG_BINNING)."""
    W = gpu
    for rng_mode, on, off in ((W.RNG_PIXEL, W.FLAG_BINNING, 0), (W.RNG_PIXEL, W.FLAG_BINNING, W.FLAG_BINNING | W.FLAG_NO_BINNING)):
        a = make_tracer(W, "shirley", 1920, 1080, rng_mode=rng_mode, max_wavefronts=8, flags=on, batch=3)
        b = make_tracer(W, "shirley", 1920, 1080, rng_mode=rng_mode, max_wavefronts=8, flags=off, batch=3)
        assert a.loop_kind == "fused_binned" and b.loop_kind == "fused"
        a.render(3); b.render(3)
        assert_bit_equal(a.accumulated(), b.accumulated(), f"mode {rng_mode}")
        assert np.array_equal(a.wavefront_totals(), b.wavefront_totals())
        a.close(); b.close()


def test_binned_loop_in_band_shards(gpu, orc):
    """Pixel-keyed mode, the frame cut into 3 band shards (what 3 ranks render): every shard runs the binned loop on its own bands."""
    W = gpu
    w, h, spp, bounces = 400, 225, 3, 5
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, rng_mode=1, max_wavefronts=bounces, miss_floor=0)
    want = o.render(spp)
    cc = W.CameraController(W.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    acc = W.render_chunked(W.Scene.book_one_final(1), W.RenderParameters(cc, (w, h)), spp, 3, max_wavefronts=bounces, rng_mode=W.RNG_PIXEL, miss_floor=0,
                           flags=W.FLAG_BINNING)  # (slabs this small keep the thread-ordered queue by default)
    assert_bit_equal(acc, want, "3 band shards, binned")
    o.close()


@pytest.mark.parametrize("rng_mode", [1])
@pytest.mark.parametrize("bounces,miss_floor", [(1, 128), (2, 128), (3, 10 ** 9), (8, 3000)])
def test_binned_loop_short_chains_and_early_exits(gpu, orc, rng_mode, bounces, miss_floor):
    """One and two wavefronts (no middle launch at all / exactly one), a miss floor that stops the loop right after the first extend
    (path_tracer.rs:332: the hits are never shaded), and one that stops it in the middle of the chain for some samples of the batch."""
    W = gpu
    w, h, spp = 200, 120, 5
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, rng_mode=rng_mode, max_wavefronts=bounces, miss_floor=miss_floor)
    want = o.render(spp)
    pt = make_tracer(W, "shirley", w, h, rng_mode=rng_mode, max_wavefronts=bounces, miss_floor=miss_floor, flags=W.FLAG_BINNING, batch=3)
    pt.render(spp)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), want, f"{bounces} wavefronts, miss floor {miss_floor}, mode {rng_mode}")
    assert np.array_equal(pt.totals(), o.totals())
    pt.close(); o.close()


def test_binning_is_refused_with_the_dispatch_keyed_rng(gpu):
    """shade.wgsl:72 keys its RNG on the dispatch's thread index, i.e. on the order of the hit queue, which the class-binned loop gives
    up: WFPT_FLAG_BINNING with WFPT_RNG_DISPATCH is an error at wfpt_create, not a silently different loop."""
    W = gpu
    with pytest.raises(W.WfptError) as e:
        make_tracer(W, "shirley", 200, 120, rng_mode=W.RNG_DISPATCH, flags=W.FLAG_BINNING)
    assert "WFPT_RNG_PIXEL" in str(e.value)
    pt = make_tracer(W, "shirley", 200, 120, rng_mode=W.RNG_DISPATCH)
    assert pt.loop_kind == "fused"
    pt.close()
    pt = make_tracer(W, "shirley", 200, 120, rng_mode=W.RNG_PIXEL, flags=W.FLAG_BINNING)
    assert pt.loop_kind == "fused_binned"
    pt.close()
