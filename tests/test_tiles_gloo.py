"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups run the exact sharding / gather / assemble
code bench.py uses (wavefront_path_tracer_amd.tiles), with the oracle standing in for the GPU renderer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch.distributed as dist
    from oracle import oracle as O
    from helpers import inputs_for, make_oracle
    from wavefront_path_tracer_amd import tiles
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    o = make_oracle(O, inputs_for(O, "shirley", w, h), w, h, rng_mode=O.RNG_PIXEL, max_wavefronts=4,
                    tile_rank=rank, tile_world=world)
    slab = o.render(spp)
    frame = tiles.gather_slabs(slab, rank, world, w, h)
    if rank == 0:
        np.save(out_path, frame)
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_gathers_to_the_unsharded_image(orc, tmp_path, world):
    from helpers import inputs_for, make_oracle
    w, h, spp = 160, 100, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    full = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, rng_mode=orc.RNG_PIXEL, max_wavefronts=4)
    want = full.render(spp)
    full.close()
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_band_ownership():
    from wavefront_path_tracer_amd import tiles
    h = 1080
    for world in (1, 2, 4, 8):
        owned = sorted(b for r in range(world) for b in tiles.bands_of(r, world, h))
        assert owned == list(range(135))
        sizes = [len(tiles.bands_of(r, world, h)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1  # balanced to within one band
    assert tiles.slab_rows(0, 8, 1080) == 8 * 17 and tiles.slab_rows(7, 8, 1080) == 8 * 16
    assert tiles.bands_of(1, 2, 225) == list(range(1, 29, 2))  # 225 rows -> 29 bands, the last one partial


def test_torch_assembly_equals_numpy():
    """The on-device de-interleave bench.py uses (torch strided copies) against the numpy reference, ragged height."""
    import torch
    from wavefront_path_tracer_amd import tiles
    w, h = 40, 100  # 13 bands, the last one partial
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        slabs = [rng.random((tiles.slab_pixels(r, world, w, h), 3), dtype=np.float32) for r in range(world)]
        want = tiles.assemble(slabs, w, h)
        pad = 3 * max(tiles.slab_pixels(r, world, w, h) for r in range(world))
        flat = []
        for s_ in slabs:
            t = torch.zeros(pad, dtype=torch.float32)
            t[:s_.size] = torch.from_numpy(s_.reshape(-1))
            flat.append(t)
        got = tiles.assemble_torch(flat, w, h).numpy().reshape(-1, 3)
        assert np.array_equal(got, want)
