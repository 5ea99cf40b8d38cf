"""GPU parity tests proper: the HIP path, called through the C ABI (libwfpt.so), against the oracle on the
same seeded inputs. Bit-exact for everything: ray payloads, queue order, counters, images.

Tolerance: none. North-star asks for "per-pixel L2 tolerance (RNG bit-exact)"; the build's arithmetic is
defined so that the whole image is bit-equal, so the tolerance written here is 0 ULP.
"""
import numpy as np
import pytest

from conftest import assert_bit_equal
from helpers import inputs_for, make_oracle, make_tracer

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ device arithmetic
def test_device_math_matches_oracle(gpu, orc):
    rng = np.random.default_rng(7)
    L = orc.lib()
    n = 1 << 20
    u = rng.random(n, dtype=np.float32)
    x = np.concatenate([u * np.float32(6.2831855), np.array([0.0, 6.2831855, 3.1415927, 1.5707964], "<f4")])
    s, c = np.zeros_like(x), np.zeros_like(x)
    L.orc_probe_sincos(orc._p(x), orc._p(s), orc._p(c), x.size)
    assert_bit_equal(gpu.selftest_math(2, x), s, "sin")
    assert_bit_equal(gpu.selftest_math(3, x), c, "cos")
    # pow as shade uses it: pow(u, 0.33333) and pow(1 - cos, 5) incl. 0, 1, tiny and negative bases
    base = np.concatenate([u, 2 * u, np.array([0.0, 1.0, 2.0, 1e-9, 1e-30, -1e-7, 2.3283064e-10], "<f4")])
    for y in (0.33333, 5.0):
        yy = np.full_like(base, y)
        ref = np.zeros_like(base)
        L.orc_probe_pow(orc._p(base), orc._p(yy), orc._p(ref), base.size)
        assert_bit_equal(gpu.selftest_math(4, base, yy), ref, f"pow(x,{y})")
    # IEEE sqrt and divide must be correctly rounded on the device (numpy float32 ops are)
    a = (rng.random(n, dtype=np.float32) * 1e4).astype("<f4")
    b = (rng.random(n, dtype=np.float32) * 1e-3 + 1e-9).astype("<f4")
    assert_bit_equal(gpu.selftest_math(0, a), np.sqrt(a), "sqrt")
    assert_bit_equal(gpu.selftest_math(1, a, b), a / b, "div")
    assert_bit_equal(gpu.selftest_math(1, np.ones_like(b), b - 5e-4), np.float32(1) / (b - np.float32(5e-4)), "rcp")
    # u32 -> f32 * 2^-32 (round to nearest even), incl. the values that round up to 1.0
    k = np.concatenate([rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4"),
                        np.array([0, 1, 0xFFFFFFFF, 0xFFFFFF7F, 0xFFFFFF80, 0xFFFFFF81, 0x80000000], "<u4")])
    ref = (k.astype(np.float32) * np.float32(2.3283064365387e-10)).astype("<f4")
    assert_bit_equal(gpu.selftest_math(5, k.view("<f4")), ref, "u32->unit float")


# ------------------------------------------------------------------ stage-wise parity (Kernel::run API)
@pytest.mark.parametrize("kind,w,h", [("simple", 64, 64), ("simple", 128, 72), ("shirley", 400, 224)])
@pytest.mark.parametrize("rng_mode", [0, 1])
def test_stage_by_stage(gpu, orc, kind, w, h, rng_mode):
    """Drive both sides exactly like PathTracer::run drives its Kernels (path_tracer.rs:296-367) and compare
    every observable buffer after every stage of the first three wavefronts."""
    W = gpu
    o = make_oracle(orc, inputs_for(orc, kind, w, h), w, h, rng_mode=rng_mode)
    pt = make_tracer(W, kind, w, h, rng_mode=rng_mode)
    n = w * h
    frame = W.GPUFrameBuffer.new(w, h, 3)
    pt.set_frame(frame)
    o.set_frame(3, 0)
    pt.reset_image(); o.reset_image()
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    pt.generate_ray_kernel.run((w // 8, h // 8)); o.generate_rays(w // 8, h // 8, False)
    assert_bit_equal(pt.rays(n), o.rays(n).view(W.RAY), "generate_rays: ray_buffer")
    ext = W.workgroup_size_64(n)
    for wavefront in range(3):
        pt.extend_kernel.run(ext); o.extend(*ext)
        c_gpu, c_orc = pt.read_counters(), o.counters()
        assert np.array_equal(c_gpu[:3], c_orc[:3]), f"counters after extend {wavefront}: {c_gpu[:3]} vs {c_orc[:3]}"
        misses, hits = int(c_orc[0]), int(c_orc[1])
        assert_bit_equal(pt.hits(hits), o.hits(hits).view(W.HIT), f"hit_buffer, wavefront {wavefront}")
        assert_bit_equal(pt.misses(misses), o.misses(misses), f"miss_buffer, wavefront {wavefront}")
        c_orc[2] = 0
        pt.set_counters(c_orc); o.set_counters(c_orc)
        sh, ms = W.workgroup_size_64(hits), W.workgroup_size_64(misses)
        pt.shade_kernel.run(sh); o.shade(*sh)
        assert_bit_equal(pt.extension_rays(hits), o.extension_rays(hits).view(W.RAY), f"extension rays, wavefront {wavefront}")
        assert_bit_equal(pt.image(), o.image(), f"image after shade, wavefront {wavefront}")
        pt.miss_kernel.run(ms); o.miss(*ms)
        assert_bit_equal(pt.image(), o.image(), f"image after miss, wavefront {wavefront}")
        n_ext = int(pt.read_counters()[2])
        assert n_ext == int(o.counters()[2]) == hits
        pt.swap_ray_queues(); o.swap_ray_queues()
        ext = W.workgroup_size_64(n_ext)
        pt.set_counters([0, 0, n_ext, 0]); o.set_counters([0, 0, n_ext, 0])
    acc = W.workgroup_size_64(n)
    pt.accumulate_kernel.run(acc); o.accumulate(*acc)
    assert_bit_equal(pt.accumulated(), o.accumulated(), "accumulated")
    for k in (pt.generate_ray_kernel, pt.extend_kernel, pt.shade_kernel, pt.miss_kernel, pt.accumulate_kernel):
        assert k.get_timing() > 0.0  # Kernel::get_timing: microseconds, running mean
    pt.close(); o.close()


# ------------------------------------------------------------------ whole loop
CASES = [
    # kind, w, h, spp, max_wavefronts
    ("simple", 64, 64, 4, 4),
    ("shirley", 400, 224, 4, 4),   # strict: multiples of 8, identical to the reference's dispatch
    ("shirley", 400, 225, 4, 4),   # BASELINE config 1; 225 % 8 != 0 -> true-size rule
    ("shirley", 200, 120, 2, 50),  # reference defaults: 50 wavefronts, exit when misses < 128
]


@pytest.mark.parametrize("kind,w,h,spp,bounces", CASES)
@pytest.mark.parametrize("rng_mode", [0, 1])
# fused bounce launches: hipGraph replay / direct; 4 = WFPT_FLAG_UNFUSED: stage kernels one by one; 128 = WFPT_FLAG_NO_BINNING (names the default,
# the thread-ordered queue), 256 = WFPT_FLAG_BINNING (the class-binned loop; pixel-keyed mode only)
@pytest.mark.parametrize("flags", [0, 2, 4, 6, 128, 256, 258])
def test_device_resident_loop(gpu, orc, kind, w, h, spp, bounces, rng_mode, flags):
    W = gpu
    if (flags & W.FLAG_BINNING) and rng_mode != W.RNG_PIXEL:
        with pytest.raises(W.WfptError):  # the class-binned loop gives up the queue's order, which shade.wgsl:72's RNG key needs
            make_tracer(W, kind, w, h, rng_mode=rng_mode, max_wavefronts=bounces, flags=flags)
        return
    o = make_oracle(orc, inputs_for(orc, kind, w, h), w, h, rng_mode=rng_mode, max_wavefronts=bounces)
    pt = make_tracer(W, kind, w, h, rng_mode=rng_mode, max_wavefronts=bounces, flags=flags)
    for s in range(spp):
        o.render_sample()
        pt.render_sample()
        assert np.array_equal(pt.bounce_table(), o.bounce_table()), f"per-bounce (rays,hits,misses) table, sample {s}"
    assert_bit_equal(pt.accumulated(), o.accumulated(), "accumulated image")
    assert np.array_equal(pt.totals(), o.totals())
    pt.close(); o.close()


@pytest.mark.parametrize("kind,w,h", [("simple", 64, 64), ("shirley", 400, 224)])
def test_host_driven_run_equals_device_loop(gpu, orc, kind, w, h):
    """PathTracer.run() (the reference's host loop over Kernel::run with counter read-backs) and the
    device-resident loop are the same computation."""
    W = gpu
    a = make_tracer(W, kind, w, h, max_wavefronts=6, spp=3)
    b = make_tracer(W, kind, w, h, max_wavefronts=6)
    o = make_oracle(orc, inputs_for(orc, kind, w, h), w, h, max_wavefronts=6)
    for _ in range(3):
        a.run()
    b.render(3)
    o.render(3)
    assert a.progress() == pytest.approx(3 / W.SPP)
    assert_bit_equal(a.accumulated(), b.accumulated(), "host-driven vs device-resident")
    assert_bit_equal(a.accumulated(), o.accumulated(), "host-driven vs oracle")
    a.close(); b.close(); o.close()


def test_per_material_split_is_identical(gpu, orc):
    """README.md:19 to-do: shade split by material. Same image bit for bit, in both RNG modes."""
    W = gpu
    for rng_mode in (0, 1):
        a = make_tracer(W, "shirley", 400, 224, max_wavefronts=5, rng_mode=rng_mode)
        b = make_tracer(W, "shirley", 400, 224, max_wavefronts=5, rng_mode=rng_mode, flags=W.FLAG_SPLIT_SHADE)
        a.render(2); b.render(2)
        assert np.array_equal(a.bounce_table(), b.bounce_table())
        assert_bit_equal(a.accumulated(), b.accumulated(), "split vs unified shade")
        a.close(); b.close()


def test_interactive_camera_resets_accumulation(gpu, orc):
    """path_tracer.rs:231-277: a moved camera (camera_controller.rs:125-158) or a resized viewport marks the render
    parameters changed; the next run() re-uploads camera / matrices, clears the accumulation and restarts at frame 1,
    so the image equals a fresh render from the new pose."""
    W = gpu
    w, h = 400, 224
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=4, spp=2)
    pt.run(); pt.run()
    assert pt.render_progress.accumulated_samples() == 2
    rp = pt.get_render_parameters()
    cc = rp.camera_controller().copy()
    cc.move_forward(1); cc.move_up(1); cc.process_mouse((5.0, 1.0))
    cc.update_camera(0.5)
    rp.update_camera_controller(cc)
    assert rp.changed() and rp.camera_changed() and not rp.resized()
    pt.update_render_parameters(rp)
    pt.run(); pt.run()
    assert pt.render_progress.accumulated_samples() == 2 and not rp.changed()
    # the oracle rendered from the moved pose
    view = cc.get_view_matrix()
    cam = cc.get_GPU_camera()
    sp, mt, nodes, cam0, _, _ = inputs_for(orc, "shirley", w, h)
    proj = W.ProjectionMatrix(cc.vfov_rad(), np.float32(w) / np.float32(h), *cc.get_clip_planes()).p_inv()
    o = make_oracle(orc, (sp, mt, nodes, cam.view(cam0.dtype), proj, view), w, h, max_wavefronts=4)
    o.render(2)
    assert_bit_equal(pt.accumulated(), o.accumulated(), "after the camera moved")
    # resize: new viewport, same rule
    rp.set_viewport((200, 120))
    pt.resize(rp)
    pt.run()
    assert pt.n_pixels == 200 * 120 and pt.render_progress.accumulated_samples() == 1
    pt.close(); o.close()


def test_reset_progress_restarts_at_frame_one(gpu, orc):
    """wfpt_reset_progress = RenderProgress::reset (parameters.rs:92-95) + the accumulation clear of path_tracer.rs:248-250."""
    W = gpu
    pt = make_tracer(W, "shirley", 200, 120, max_wavefronts=4)
    pt.render(5)
    assert W.lib().wfpt_frame(pt.handle) == 5
    pt.reset_progress()
    assert W.lib().wfpt_frame(pt.handle) == 0 and pt.render_progress.accumulated_samples() == 0
    pt.render(3)
    o = make_oracle(orc, inputs_for(orc, "shirley", 200, 120), 200, 120, max_wavefronts=4)
    assert_bit_equal(pt.accumulated(), o.render(3), "frames 1..3 again after reset_progress")
    pt.close(); o.close()


def test_tile_sharding_pixel_mode(gpu, orc):
    """Bands of 8 rows dealt round-robin to `world` contexts reproduce the unsharded image in PIXEL mode."""
    from wavefront_path_tracer_amd import tiles
    W = gpu
    w, h, spp = 400, 225, 2
    full = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL)
    full.render(spp)
    ref = full.accumulated()
    for world in (2, 3):
        slabs = []
        for rank in range(world):
            pt = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL, tile_rank=rank, tile_world=world)
            pt.render(spp)
            slabs.append(pt.accumulated())
            pt.close()
        assert_bit_equal(tiles.assemble(slabs, w, h), ref, f"assembled from {world} ranks")
    # image chunking on one GPU (README.md:20): the same cut, rendered slab after slab with queues 1/5 the size
    def chunk(rank, world):
        return make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL, tile_rank=rank, tile_world=world)
    probe = chunk(0, 5)
    assert probe.ray_capacity < full.ray_capacity / 4
    probe.close()
    assert_bit_equal(tiles.render_in_chunks(chunk, w, h, spp, 5), ref, "rendered in 5 chunks")
    # the same behind the C ABI (wfpt_render_chunked): one call, frame assembled into the caller's buffer; 225 % 8 != 0
    cc = W.CameraController(W.Camera.book_one_final_camera(), 20.0, 0.6, 10.0, 0.1, 100.0, 4.0, 0.1)
    rp = W.RenderParameters(cc, (w, h))
    for chunks in (1, 5):
        scene = W.Scene.book_one_final(1)  # fresh: building the BVH reorders the spheres in place
        assert_bit_equal(W.render_chunked(scene, rp, spp, chunks, max_wavefronts=4), ref, f"wfpt_render_chunked, {chunks} chunks")
    # The loop-exit test `misses < miss_floor` (path_tracer.rs:332) sees each chunk's own miss count: with one band per chunk
    # (29 chunks) some chunks fall below the reference's 128 although the whole frame does not. miss_floor = 0 leaves the
    # wavefront limit as the only bound and makes the image independent of the cut.
    nofloor = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL, miss_floor=0)
    nofloor.render(spp)
    assert_bit_equal(W.render_chunked(W.Scene.book_one_final(1), rp, spp, 29, max_wavefronts=4, miss_floor=0), nofloor.accumulated(),
                     "wfpt_render_chunked, one band per chunk, miss_floor 0")
    nofloor.close()
    with pytest.raises(W.WfptError):
        W.render_chunked(W.Scene.book_one_final(1), rp, spp, 3, max_wavefronts=4, rng_mode=W.RNG_DISPATCH)  # the dispatch-keyed RNG depends on the cut
    full.close()


# ------------------------------------------------------------------ BASELINE.json's full size, against golden vectors
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("config", ["config2", "config4-split-shade", "unfused", "binned", "not-binned"])
def test_full_hd_against_golden(gpu, mode, config):
    """1920x1080, 8 bounces (BASELINE config 2's frame) for 2 samples: per-bounce (rays, hits, misses) tables and
    the SHA-256 of the accumulated image must equal the oracle's committed golden vectors; plus the size-
    independent properties of the chain. config4-split-shade: BASELINE config 4 at its stated size -- the per-material shade
    stages (lambertian / metal / dielectric queues, README.md:19) give the same frame; unfused: the stage kernels one by one."""
    import hashlib
    import os
    W = gpu
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"shirley_1920x1080_mode{mode}.npz"))
    w, h, spp, bounces = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["bounces"])
    flags = {"config2": 0, "config4-split-shade": W.FLAG_SPLIT_SHADE, "unfused": W.FLAG_UNFUSED,
             "binned": W.FLAG_BINNING, "not-binned": W.FLAG_NO_BINNING}[config]
    if config == "binned" and mode != W.RNG_PIXEL:
        pytest.skip("the class-binned loop exists in the pixel-keyed RNG mode only")
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, rng_mode=mode, flags=flags)
    for s in range(spp):
        pt.render_sample()
        t = pt.bounce_table().astype(np.int64)
        assert np.array_equal(t, g["tables"][s][:len(t)]), f"sample {s}"
        assert t[0, 0] == w * h and (t[:, 1] + t[:, 2] == t[:, 0]).all() and (t[1:, 0] == t[:-1, 1]).all()
    acc = pt.accumulated()
    assert hashlib.sha256(acc.tobytes()).hexdigest() == str(g["acc_sha256"])
    assert np.array_equal(pt.totals(), g["totals"])
    assert np.isfinite(acc).all() and acc.min() >= 0.0
    f = int(g["downsample"])
    small = acc.reshape(h, w, 3)[:(h // f) * f, :(w // f) * f].reshape(h // f, f, w // f, f, 3).mean(axis=(1, 3))
    assert np.allclose(small, g["acc_small"], rtol=1e-5, atol=1e-6)
    pt.close()


def test_golden_small_cases_on_gpu(gpu):
    """The committed stage dumps of the 5-sphere scene (tests/golden/simple_64x64_mode*.npz) straight against the GPU."""
    import os
    W = gpu
    for mode in (0, 1):
        for (w, h) in ((64, 64), (128, 72)):
            g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"simple_{w}x{h}_mode{mode}.npz"))
            pt = make_tracer(W, "simple", w, h, max_wavefronts=4, rng_mode=mode)
            pt.render(4)
            assert_bit_equal(pt.accumulated(), g["acc"], f"golden {w}x{h} image")
            pt.close()


def test_error_paths(gpu):
    W = gpu
    pt = make_tracer(W, "simple", 64, 64)
    with pytest.raises(W.WfptError):
        pt.generate_ray_kernel.run((4096, 4096))  # dispatch larger than the ray buffer
    with pytest.raises(W.WfptError):
        W.Kernel("display", pt)  # not a stage of this path (kernel.rs:36 would panic on the missing shader)
    with pytest.raises(W.WfptError):
        pt.hits(10 ** 6)
    pt.extend_kernel.run((0, 0))  # an empty dispatch is a no-op
    # update_buffers (path_tracer.rs:240-277): a viewport change zeroes the accumulation and restarts progress
    pt.render(2)
    assert pt.accumulated().sum() > 0
    rp = pt.get_render_parameters()
    rp.set_viewport((48, 32))
    pt.update_render_parameters(rp)
    pt.update_buffers()
    assert pt.accumulated().sum() == 0 and pt.n_pixels == 48 * 32
    pt.render(1)
    assert W.lib().wfpt_frame(pt.handle) == 1
    with pytest.raises(W.WfptError):
        rp.set_viewport((640, 480))
        pt.update_buffers()  # larger than the capacity given at creation
    pt.close()


def test_many_samples_in_flight(gpu, orc):
    """More samples in flight than the stage-by-stage loop supports (the fused launches take up to 128; a request beyond
    that is clamped, the unfused loop clamps at 64): same image as the oracle's sequential samples, remainder included."""
    W = gpu
    w, h, bounces = 96, 64, 5
    spp = 128 + 37
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
    want = o.render(spp)
    for batch, flags in ((128, 0), (100, W.FLAG_NO_GRAPH), (500, 0), (128, W.FLAG_UNFUSED)):
        pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, batch=batch, flags=flags)
        pt.render(spp)
        assert_bit_equal(pt.accumulated(), want, f"batch={batch} flags={flags}")
        assert np.array_equal(pt.totals(), o.totals())
        pt.close()
    o.close()
    # the class-binned loop (pixel-keyed RNG) with 128 samples in flight: its plan holds 128 x kBinClasses (sample, class) entries
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces, rng_mode=W.RNG_PIXEL)
    want = o.render(spp)
    for batch, flags in ((128, W.FLAG_BINNING), (100, W.FLAG_BINNING | W.FLAG_NO_GRAPH)):
        pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, batch=batch, flags=flags, rng_mode=W.RNG_PIXEL)
        assert pt.loop_kind == "fused_binned"
        pt.render(spp)
        assert_bit_equal(pt.accumulated(), want, f"binned, batch={batch} flags={flags}")
        assert np.array_equal(pt.totals(), o.totals())
        pt.close()
    o.close()


@pytest.mark.parametrize("batch", [1, 4, 8, 16])
def test_batched_samples_equal_sequential(gpu, orc, batch):
    """wfpt_render keeps `batch` samples in flight per launch; samples are independent and accumulate in frame
    order, so the image must be bit-equal to one-sample-at-a-time rendering (and to the oracle)."""
    W = gpu
    w, h, spp, bounces = 200, 120, 19, 5  # 19 = full batches plus a remainder rendered one by one
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
    want = o.render(spp)
    for flags in (0, W.FLAG_NO_GRAPH, W.FLAG_UNFUSED):
        pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, batch=batch, flags=flags)
        pt.render(spp)
        assert_bit_equal(pt.accumulated(), want, f"batch={batch} flags={flags}")
        assert np.array_equal(pt.totals(), o.totals())
        assert np.array_equal(pt.bounce_table(), o.bounce_table())  # table of the last sample
        assert W.lib().wfpt_frame(pt.handle) == spp
        ms, launches = pt.render_timed(batch)  # one more batch, timed: same results as untimed
        o2 = o.render(batch)
        assert_bit_equal(pt.accumulated(), o2, "render_timed")
        if flags & W.FLAG_UNFUSED:  # stage kernels one by one
            assert launches[W.STAGES["extend"]] == bounces and ms[W.STAGES["extend"]] > 0
            assert launches[W.STAGES["shade"]] == bounces and launches[W.STAGES["bounce"]] == 0
        else:  # one fused launch per wavefront, plus the shade + miss of the last one
            assert launches[W.STAGES["bounce_first"]] == 1 and launches[W.STAGES["bounce_last"]] == 1
            assert launches[W.STAGES["bounce"]] == bounces - 1 and ms[W.STAGES["bounce"]] > 0
            assert launches[W.STAGES["extend"]] == 0 and launches[W.STAGES["scan"]] == bounces
        want = o2
        spp_done = spp + batch
        pt.close()
        # rebuild the oracle state for the second flags pass
        o.close()
        o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
        want = o.render(spp)
    o.close()


@pytest.mark.parametrize("rng_mode", [0, 1])
def test_per_material_stages_of_the_stage_api(gpu, orc, rng_mode):
    """`shade_lambertian`, `shade_metal`, `shade_dielectric` (README.md:19's by-material kernels) run one after the
    other over extend's material partition must equal the single `shade` stage: same extension rays in the same
    slots, same image, counters[2] == hits."""
    W = gpu
    w, h = 400, 224
    n = w * h
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, rng_mode=rng_mode)
    pt = make_tracer(W, "shirley", w, h, rng_mode=rng_mode)
    stages = [W.Kernel(name, pt) for name in ("shade_lambertian", "shade_metal", "shade_dielectric")]
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 9)); o.set_frame(9, 0)
    pt.reset_image(); o.reset_image()
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    pt.generate_ray_kernel.run((w // 8, h // 8)); o.generate_rays(w // 8, h // 8, False)
    ext = W.workgroup_size_64(n)
    for wavefront in range(3):
        pt.extend_kernel.run(ext); o.extend(*ext)
        c = o.counters()
        hits, misses = int(c[1]), int(c[0])
        c[2] = 0
        pt.set_counters(c); o.set_counters(c)
        sh = W.workgroup_size_64(hits)
        emitted = []
        for k in stages:  # any order gives the same result; each adds the rays it emits to counters[2]
            k.run(sh)
            emitted.append(int(pt.read_counters()[2]))
        o.shade(*sh)
        assert emitted[-1] == hits and emitted[0] > emitted[1] - emitted[0] > 0  # mostly lambertian, some metal
        assert_bit_equal(pt.extension_rays(hits), o.extension_rays(hits).view(W.RAY), f"extension rays {wavefront}")
        assert_bit_equal(pt.image(), o.image(), f"image {wavefront}")
        ms = W.workgroup_size_64(misses)
        pt.miss_kernel.run(ms); o.miss(*ms)
        pt.swap_ray_queues(); o.swap_ray_queues()
        ext = W.workgroup_size_64(hits)
        pt.set_counters([0, 0, hits, 0]); o.set_counters([0, 0, hits, 0])
    pt.close(); o.close()


# ------------------------------------------------------------------ edge cases
def _single_sphere_inputs(orc, w, h):
    sp = np.zeros(1, orc.SPHERE)
    sp["center"][0] = (0.0, 0.0, -1.0, 1.0)
    sp["radius"] = 0.5
    mt = np.zeros(1, orc.MATERIAL)
    mt["albedo"][0] = (0.8, 0.3, 0.3, 1.0)
    sp, nodes = orc.build_bvh(sp)  # the root is a leaf (extend.wgsl:84: its box is never tested)
    cam, ip, vw = orc.camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), 60.0, 0.0, 10.0, 0.1, 100.0, w, h)
    return sp, mt, nodes, cam, ip, vw


@pytest.mark.parametrize("w,h", [(8, 8), (16, 8), (24, 16), (100, 64), (64, 100), (33, 17)])
@pytest.mark.parametrize("rng_mode", [0, 1])
def test_small_and_ragged_sizes(gpu, orc, w, h, rng_mode):
    """8x8 is one workgroup (the reference's workgroup_size_64 panics for <= 64 threads; the build defines (1,1));
    ragged sizes exercise the true-size rule in both dimensions."""
    W = gpu
    o = make_oracle(orc, inputs_for(orc, "simple", w, h), w, h, rng_mode=rng_mode, max_wavefronts=5, miss_floor=0)
    pt = make_tracer(W, "simple", w, h, rng_mode=rng_mode, max_wavefronts=5, miss_floor=0)
    for _ in range(3):
        o.render_sample(); pt.render_sample()
        assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), o.accumulated(), f"{w}x{h}")
    pt.close(); o.close()


def test_single_sphere_root_leaf(gpu, orc):
    W = gpu
    w, h = 64, 48
    inputs = _single_sphere_inputs(orc, w, h)
    o = make_oracle(orc, inputs, w, h, max_wavefronts=4, miss_floor=0)
    scene = W.Scene(inputs[0].view(W.SPHERE), inputs[1].view(W.MATERIAL))
    cc = W.CameraController(W.Camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0)), 60.0, 0.0, 10.0, 0.1, 100.0)
    pt = W.PathTracer(scene, W.RenderParameters(cc, (w, h)), max_wavefronts=4, miss_floor=0)
    assert len(pt.bvh_tree.nodes) == 2 and pt.bvh_tree.nodes[0]["prim_count"] == 1
    o.render(2); pt.render(2)
    assert_bit_equal(pt.accumulated(), o.accumulated(), "single sphere")
    t = pt.bounce_table()
    assert t[0, 1] > 0 and t[0, 2] > 0  # some primary rays hit the sphere, some the sky
    pt.close(); o.close()


@pytest.mark.parametrize("sample_number", [1, 3, 7])
def test_nonzero_sample_number_takes_the_advance_path(gpu, orc, sample_number):
    """generate_rays.wgsl:155-171 / shade.wgsl: `advance(state, sample_number * 10)` with its as-written bug (it only
    accumulates while the remaining count equals 1). The shipped loop always passes 0 (SPF = 1, path_tracer.rs:301),
    so this drives the kernels through the stage API with the frame uniform's sample_number set."""
    W = gpu
    w, h = 128, 72
    n = w * h
    o = make_oracle(orc, inputs_for(orc, "simple", w, h), w, h)
    pt = make_tracer(W, "simple", w, h)
    frame = W.GPUFrameBuffer.new(w, h, 5)
    frame.set_sample_number(sample_number)
    pt.set_frame(frame); o.set_frame(5, sample_number)
    pt.reset_image(); o.reset_image()
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    pt.generate_ray_kernel.run((w // 8, h // 8)); o.generate_rays(w // 8, h // 8, False)
    assert_bit_equal(pt.rays(n), o.rays(n).view(W.RAY), "rays with a non-zero sample_number")
    ext = W.workgroup_size_64(n)
    pt.extend_kernel.run(ext); o.extend(*ext)
    c = o.counters()
    hits = int(c[1])
    c[2] = 0
    pt.set_counters(c); o.set_counters(c)
    sh = W.workgroup_size_64(hits)
    pt.shade_kernel.run(sh); o.shade(*sh)
    assert_bit_equal(pt.extension_rays(hits), o.extension_rays(hits).view(W.RAY), "extension rays with a non-zero sample_number")
    pt.close(); o.close()


def _odd_rays(W, n, seed):
    """Rays generate_rays never makes: axis-parallel (1/0 = inf in the slab test), zero-length, huge / tiny / denormal
    magnitudes, origins inside spheres and exactly on box planes, NaN and infinite components."""
    rng = np.random.default_rng(seed)
    r = np.zeros(n, W.RAY)
    r["origin"][:, :3] = rng.uniform(-12, 12, (n, 3)).astype("<f4")
    r["origin"][:, 1] = np.abs(r["origin"][:, 1]) * np.float32(0.2)
    r["origin"][:, 3] = 1.0
    d = rng.normal(size=(n, 3)).astype("<f4")
    kind = np.arange(n) % 16
    d[kind == 1, 0] = 0.0                                  # parallel to the yz plane
    d[kind == 2, 1:] = 0.0                                 # along x
    d[kind == 3] = 0.0                                     # no direction at all
    d[kind == 4] *= np.float32(1e-30)                      # inverse overflows for some
    d[kind == 5] *= np.float32(1e-42)                      # denormal direction
    d[kind == 6] *= np.float32(1e18)                       # a = d.d overflows
    d[kind == 7, 2] = np.float32(-0.0)
    r["origin"][kind == 8, :3] = np.float32([4.0, 1.0, 0.0])   # centre of a big sphere (scene.rs:97-99)
    r["origin"][kind == 9, :3] = np.float32([0.0, -1000.0, 0.0])  # centre of the ground sphere
    d[kind == 10, 0] = np.float32(np.nan)
    r["origin"][kind == 11, 2] = np.float32(np.inf)
    d[kind == 12, 1] = np.float32(-np.inf)
    r["origin"][kind == 13, 1] = np.float32(0.0)           # on the ground: t ~ 0 roots against the 0.001 window
    d[kind == 13, 1] = -np.abs(d[kind == 13, 1])
    r["origin"][kind == 14, :3] = np.float32(1e30)
    r["direction"][:, :3] = d
    with np.errstate(all="ignore"):
        r["inv_direction"] = (np.float32(1.0) / d).astype("<f4")  # carried by the reference's Ray, recomputed by extend
    r["pixel_idx"] = np.arange(n, dtype="<u4") % 1000
    return r


@pytest.mark.parametrize("seed", [1, 2])
def test_extend_on_degenerate_rays(gpu, orc, seed):
    """extend on hand-made rays full of zeros, infinities, NaNs and denormals: the slab test's min/max chain, the
    quadratic's sqrt and divide and the (0.001, nearest) window must treat them like the oracle does, bit for bit."""
    W = gpu
    w, h = 128, 64
    n = w * h
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h)
    pt = make_tracer(W, "shirley", w, h)
    rays = _odd_rays(W, n, seed)
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 1)); o.set_frame(1, 0)
    pt.write_rays(rays); o.write_rays(rays.view(orc.RAY))
    assert_bit_equal(pt.rays(n), o.rays(n).view(W.RAY), "injected rays")
    pt.set_counters([0, 0, n]); o.set_counters([0, 0, n])
    ext = W.workgroup_size_64(n)
    pt.extend_kernel.run(ext); o.extend(*ext)
    c = o.counters()
    assert np.array_equal(pt.read_counters()[:3], c[:3])
    misses, hits = int(c[0]), int(c[1])
    assert hits > n // 8 and misses > n // 8
    assert_bit_equal(pt.hits(hits), o.hits(hits).view(W.HIT), "hit queue of the degenerate rays")
    assert_bit_equal(pt.misses(misses), o.misses(misses), "miss queue of the degenerate rays")
    pt.close(); o.close()


@pytest.mark.parametrize("max_wavefronts,miss_floor", [(1, 128), (3, 0), (50, 10 ** 9), (50, 128), (64, 1)])
def test_loop_termination_policies(gpu, orc, max_wavefronts, miss_floor):
    """path_tracer.rs:323,332: `wavefront < max` and `misses < miss_floor -> break` evaluated on the device.
    miss_floor = 1e9 leaves after the first extend: nothing is shaded, every pixel accumulates its initial 1.0."""
    W = gpu
    w, h, spp = 120, 80, 2
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=max_wavefronts, miss_floor=miss_floor)
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=max_wavefronts, miss_floor=miss_floor)
    o.render(spp); pt.render(spp)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert_bit_equal(pt.accumulated(), o.accumulated(), f"max {max_wavefronts}, floor {miss_floor}")
    if miss_floor == 10 ** 9:
        assert (pt.accumulated() == spp).all() and len(pt.bounce_table()) == 1
    pt.close(); o.close()


def test_device_side_slab_assembly(gpu):
    """What bench.py does at N > 1 minus the collective: each rank's slab is copied device-to-device into a torch
    tensor (wfpt_copy_accumulated_to_device) and de-interleaved on the GPU; must equal the unsharded frame."""
    import torch
    from wavefront_path_tracer_amd import tiles
    W = gpu
    w, h, spp, world = 400, 225, 3, 4
    full = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL)
    full.render(spp)
    ref = full.accumulated()
    full.close()
    dev = torch.device("cuda", 0)
    pad = 3 * max(tiles.slab_pixels(r, world, w, h) for r in range(world))
    slabs = []
    for rank in range(world):
        pt = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL, tile_rank=rank, tile_world=world, batch=2)
        pt.render(spp)
        # the fill runs on torch's stream, the copy on the context's: drain torch's stream first
        t = torch.zeros(pad, dtype=torch.float32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        pt.copy_accumulated_to_device(t.data_ptr(), 12 * tiles.slab_pixels(rank, world, w, h))
        slabs.append(t)
        pt.close()
    frame = tiles.assemble_torch(slabs, w, h)
    assert frame.is_cuda and tuple(frame.shape) == (h, w, 3)
    assert_bit_equal(frame.cpu().numpy().reshape(-1, 3), ref, "device-side assembly")


def test_image_output(gpu, orc, tmp_path):
    """the frame right after the path, as PPM (display_shader.wgsl:50-52 tone map) and PFM."""
    W = gpu
    w, h, spp = 96, 40, 3
    pt = make_tracer(W, "simple", w, h, max_wavefronts=4)
    with pytest.raises(W.WfptError):
        pt.save_ppm(tmp_path / "early.ppm")  # nothing accumulated yet
    pt.render(spp)
    acc = pt.accumulated()
    pt.save_ppm(tmp_path / "f.ppm")
    pt.save_pfm(tmp_path / "f.pfm")
    raw = (tmp_path / "f.ppm").read_bytes()
    head = b"P6\n%d %d\n255\n" % (w, h)
    assert raw.startswith(head)
    rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(-1, 3)
    assert np.array_equal(rgb, orc.tonemap_rgb8(acc, spp))
    pt.save_png(tmp_path / "f.png")
    from test_host_model import _read_png
    w2, h2, png = _read_png(tmp_path / "f.png")
    assert (w2, h2) == (w, h) and np.array_equal(png.reshape(-1, 3), rgb)
    pfm = (tmp_path / "f.pfm").read_bytes()
    head = b"PF\n%d %d\n-1.0\n" % (w, h)
    assert pfm.startswith(head)
    lin = np.frombuffer(pfm[len(head):], "<f4").reshape(h, w, 3)[::-1]  # PFM rows are bottom-up
    assert_bit_equal(np.ascontiguousarray(lin).reshape(-1, 3), (np.float32(1.0) / np.float32(spp)) * acc, "PFM")
    tiled = make_tracer(W, "simple", w, h, tile_rank=0, tile_world=2)
    tiled.render(1)
    with pytest.raises(W.WfptError):
        tiled.save_ppm(tmp_path / "t.ppm")  # a sharded context holds only its bands
    tiled.close(); pt.close()


@pytest.mark.parametrize("w,h", [(2880, 1620), (3840, 2160)])  # main.rs:33: the shipped default and the commented 4K size
def test_large_frames(gpu, orc, w, h):
    """The reference's own frame sizes: 4.7 M and 8.3 M rays per wavefront, 16 samples in flight (multi-GB queues).
    One batch of 16 samples + 1 single sample against the oracle's bounce tables and image, bit for bit."""
    W = gpu
    bounces, spp = 4, 17
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces)
    pt.render(spp)
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
    o.render(spp)
    assert np.array_equal(pt.bounce_table(), o.bounce_table())
    assert np.array_equal(pt.totals(), o.totals())
    assert_bit_equal(pt.accumulated(), o.accumulated(), f"{w}x{h}")
    pt.close(); o.close()


def test_rccl_gather_behind_the_c_abi_single_rank(gpu):
    """wfpt_comm_unique_id / wfpt_comm_init / wfpt_gather_accumulated / wfpt_read_gathered with a one-rank communicator:
    opens RCCL at run time, joins the communicator on the context's device and assembles the frame on the device. The
    send / receive legs need one GPU per rank (RCCL refuses two ranks on one device), so on a one-GPU box this covers the
    loader, ncclCommInitRank and the root's de-interleave; the band arithmetic of N > 1 is covered by tests/test_tiles_gloo.py."""
    W = gpu
    w, h, spp = 200, 123, 3  # 123 % 8 != 0: the last band is partial
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL)
    uid = W.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    pt.comm_init(uid, 0, 1)
    with pytest.raises(W.WfptError):
        pt.comm_init(uid, 0, 1)  # already initialised
    pt.render(spp)
    pt.gather_accumulated()
    assert_bit_equal(pt.gathered(), pt.accumulated(), "gathered frame of a one-rank job")
    pt.close()
    other = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL, tile_rank=1, tile_world=2)
    with pytest.raises(W.WfptError):
        other.comm_init(uid, 0, 1)  # (rank, world) must match the context's tile cut
    with pytest.raises(W.WfptError):
        other.gather_accumulated()  # no communicator
    other.close()


def test_stage_api_refuses_pixels_outside_the_image(gpu):
    """a literal generate_rays dispatch of ceil(W/8) x ceil(H/8) tiles on a viewport that is not a multiple of 8
    would index the image past its end in shade / miss_kernel (the reference relies on monitor-sized buffers); so would an
    injected ray with a wild pixel_idx. Both are refused; the reference's own truncating dispatch (W/8, H/8) is accepted."""
    W = gpu
    w, h = 100, 100
    pt = make_tracer(W, "shirley", w, h, batch=1)
    pt.set_frame(W.GPUFrameBuffer.new(w, h, 1))
    with pytest.raises(W.WfptError):
        pt.generate_ray_kernel.run(((w + 7) // 8, (h + 7) // 8))  # 13 x 13 tiles = 10816 pixels > 10000
    pt.generate_ray_kernel.run((w // 8, h // 8))                  # path_tracer.rs:318
    rays = pt.rays(64).copy()
    rays["pixel_idx"][5] = w * h  # one past the end
    with pytest.raises(W.WfptError):
        pt.write_rays(rays)
    rays["pixel_idx"][5] = W.INACTIVE_PIXEL  # padding rays are fine
    pt.write_rays(rays)
    pt.close()
    big = make_tracer(W, "shirley", w, h, max_window_size=13 * 13 * 64, batch=1)  # monitor-sized buffers, like the reference
    big.set_frame(W.GPUFrameBuffer.new(w, h, 1))
    big.generate_ray_kernel.run(((w + 7) // 8, (h + 7) // 8))
    big.close()


def test_wavefront_totals(gpu, orc):
    """wfpt_read_wavefront_totals: per-wavefront (rays, hits, misses) summed over the samples of the device-resident loop
    equal the sum of the oracle's per-sample bounce tables."""
    W = gpu
    w, h, spp, bounces = 160, 96, 5, 6
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
    want = np.zeros((bounces, 3), np.uint64)
    for _ in range(spp):
        o.render_sample()
        t = o.bounce_table().astype(np.uint64)
        want[:len(t), 0] += t[:, 1] + t[:, 2]
        want[:len(t), 1] += t[:, 1]
        want[:len(t), 2] += t[:, 2]
    for flags in (0, W.FLAG_UNFUSED):
        pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, batch=4, flags=flags)
        pt.render(spp)
        assert np.array_equal(pt.wavefront_totals(), want), f"flags={flags}"
        assert np.array_equal(pt.wavefront_totals().sum(axis=0), pt.totals())
        pt.close()
    o.close()


@pytest.mark.parametrize("flags", ["NO_LDS_SCENE", "NO_LDS_SCENE|NO_REFILL", "NO_LDS_SCENE|BINARY_BVH", "NO_LDS_SCENE|UNFUSED"])
def test_sphere_scene_through_the_hbm_traversal(gpu, orc, flags):
    """The traversals built for scenes beyond LDS (four-wide collapsed tree with and without lane refill, binary tree from
    global memory) on the SPHERE scene, which normally lives in LDS: same image, tables and totals as the oracle."""
    W = gpu
    fl = 0
    for name in flags.split("|"):
        fl |= getattr(W, "FLAG_" + name)
    w, h, spp, bounces = 200, 123, 3, 6
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=bounces)
    want = o.render(spp)
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, flags=fl, batch=2)
    pt.render(spp)
    assert_bit_equal(pt.accumulated(), want, flags)
    assert np.array_equal(pt.totals(), o.totals()) and np.array_equal(pt.bounce_table(), o.bounce_table())
    pt.close(); o.close()


def test_gather_slabs_is_ordered_after_queued_torch_work(gpu):
    """tiles.gather_slabs fills its send buffer on torch's stream and the context copies into it on its
    own non-blocking stream. With a long kernel queued on torch's stream in front of the fill, a copy that did not wait for
    that stream would be overwritten by the late fill; the slab must arrive intact."""
    import torch
    from wavefront_path_tracer_amd import tiles
    W = gpu
    w, h, spp = 400, 224, 2  # whole bands: an unsharded context holds exactly the slab gather_slabs asks for
    dev = torch.device("cuda", 0)
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=4, rng_mode=W.RNG_PIXEL)
    pt.render(spp)
    ref = pt.accumulated()
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(20):  # ~ tens of milliseconds of queued work on torch's current stream
        a = a @ a
        a = a / a.abs().max()
    frame = tiles.gather_slabs(pt.copy_accumulated_to_device, 0, 1, w, h, device=dev, keep_on_device=True)
    assert_bit_equal(frame.cpu().numpy().reshape(-1, 3), ref, "slab copied behind queued torch work")
    pt.close()
