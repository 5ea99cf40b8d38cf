"""Pixel-tile sharding across GPUs (build-side addition; the reference is single-GPU).

The image is cut into bands of 8 pixel rows (the height of the reference's 8x8 workgroup tile,
generate_rays.wgsl:42). Band k belongs to rank k % world, so sky, horizon and ground bands are dealt
round-robin and the ranks stay balanced. Every rank runs the whole kernel chain on its own bands with
GLOBAL pixel coordinates (so generate_rays is bit-identical to a single-GPU render) and keeps its pixels in
a compact slab of whole bands. There is no exchange inside the bounce loop; the only collective is one
gather of the accumulated slabs at the end (RCCL over xGMI when the backend is "nccl").

The same cut serves the reference's to-do "image chunking" (README.md:20) on ONE GPU:
`render_in_chunks` renders the slabs one after the other, so the ray queues are sized for 1/chunks of the frame.
"""
import numpy as np


def bands_of(rank, world, height):
    """Indices of the 8-row bands owned by `rank`."""
    n_bands = (height + 7) // 8
    return list(range(rank, n_bands, world))


def slab_rows(rank, world, height):
    """Rows in a rank's slab: whole bands (the last band of the image may be partly outside it)."""
    return 8 * len(bands_of(rank, world, height))


def slab_pixels(rank, world, width, height):
    return slab_rows(rank, world, height) * width


def assemble(slabs, width, height):
    """Interleave per-rank slabs [(slab_pixels, 3) float32] back into the (width*height, 3) image."""
    world = len(slabs)
    out = np.zeros((height, width, 3), np.float32)
    for rank, slab in enumerate(slabs):
        rows = np.asarray(slab, np.float32).reshape(-1, width, 3)
        for local, band in enumerate(bands_of(rank, world, height)):
            y0 = band * 8
            n = min(8, height - y0)
            out[y0:y0 + n] = rows[local * 8:local * 8 + n]
    return out.reshape(-1, 3)


def _torch():
    """PyTorch, for the collective and the device-side assembly only. Its ROCm wheel bundles its own HIP runtime under
    the SONAME libwfpt.so also links; the package maps the wheel's copy first (`_agree_on_hip_runtime`), so either load
    order works. Should a process have mapped another copy by other means, say so instead of torch's "No HIP GPUs"."""
    import sys
    loaded_before = "torch" in sys.modules
    import torch
    from . import _lib_loaded
    if not loaded_before and _lib_loaded() and not torch.cuda.is_available():
        raise RuntimeError("import torch before the first wavefront_path_tracer_amd call in this process: libwfpt.so was "
                           "loaded first and PyTorch's bundled HIP runtime now sees no GPU")
    return torch


def assemble_torch(slabs, width, height):
    """`assemble` on whatever device the slabs live on: one strided copy per rank, no host round trip.
    slabs[r] is a flat float32 tensor holding rank r's bands; returns a (height, width, 3) tensor."""
    torch = _torch()
    world = len(slabs)
    n_bands = (height + 7) // 8
    out = torch.empty((n_bands, 8, width, 3), dtype=torch.float32, device=slabs[0].device)
    for rank, slab in enumerate(slabs):
        mine = len(bands_of(rank, world, height))
        if mine:
            out[rank::world] = slab[:mine * 8 * width * 3].view(mine, 8, width, 3)
    return out.view(n_bands * 8, width, 3)[:height]


def gather_slabs(local_slab, rank, world, width, height, device=None, group=None, keep_on_device=False):
    """One gather of every rank's accumulated slab to rank 0 through torch.distributed.

    `local_slab` is either a numpy array (CPU / gloo) or a callable `fill(ptr, n_bytes)` that copies the
    slab into device memory at `ptr` (GPU / RCCL: wfpt_copy_accumulated_to_device). Returns the assembled
    (width*height, 3) image on rank 0 and None elsewhere. With keep_on_device the de-interleave runs on the
    device the slabs were gathered to and a (height, width, 3) torch tensor is returned (no host copy).
    """
    torch = _torch()
    import torch.distributed as dist

    max_floats = 3 * max(slab_pixels(r, world, width, height) for r in range(world))
    mine = 3 * slab_pixels(rank, world, width, height)
    dev = torch.device("cpu") if device is None else device
    # `send` is written twice: its tail by a torch fill (torch's current stream) and its head by the context's
    # device-to-device copy (the context's own non-blocking stream). The two streams are not ordered with each other, so
    # the regions are kept disjoint (torch never touches [0, mine)) and torch's stream is drained before the copy is
    # queued: a late fill can then neither zero the slab nor race the collective that follows.
    send = torch.empty(max_floats, dtype=torch.float32, device=dev)
    if mine < max_floats:
        send[mine:].zero_()
    if callable(local_slab):
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        if mine:
            local_slab(send.data_ptr(), 4 * mine)  # wfpt_copy_accumulated_to_device: returns after its stream has drained
    else:
        send[:mine] = torch.from_numpy(np.ascontiguousarray(local_slab, np.float32).reshape(-1)).to(dev)
    if world == 1:
        if keep_on_device:
            return assemble_torch([send], width, height)
        return assemble([send[:mine].cpu().numpy().reshape(-1, 3)], width, height)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gather_list=recv, dst=0, group=group)
    if rank != 0:
        return None
    if keep_on_device:
        return assemble_torch(recv, width, height)
    slabs = [recv[r][:3 * slab_pixels(r, world, width, height)].cpu().numpy().reshape(-1, 3) for r in range(world)]
    return assemble(slabs, width, height)


def render_in_chunks(make_tracer, width, height, spp, chunks):
    """Image chunking on one GPU: the frame is rendered as `chunks` band-interleaved slabs in sequence and
    assembled, so every queue and image buffer on the device is sized for ceil(bands / chunks) bands instead of
    the whole frame. `make_tracer(tile_rank, tile_world)` must return a PathTracer created with those two
    parameters and rng_mode = RNG_PIXEL (the dispatch-keyed RNG of shade.wgsl:72 depends on the queue a ray sits
    in, so only the pixel-keyed mode is independent of the cut). Returns the accumulated (width*height, 3) image,
    bit-identical to the unchunked RNG_PIXEL render."""
    slabs = []
    for k in range(chunks):
        pt = make_tracer(k, chunks)
        try:
            pt.render(spp)
            slabs.append(pt.accumulated())
        finally:
            pt.close()
    return assemble(slabs, width, height)
