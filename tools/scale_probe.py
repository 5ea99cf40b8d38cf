"""What one rank of an N-GPU run does, measured on ONE GPU: a context holding the bands of rank 0 of N, `batch` samples in
flight; prints ms per sample of the slab and the strong-scaling efficiency against the unsharded frame."""
import sys, time
sys.path.insert(0, '/root/repo')
import wavefront_path_tracer_amd as W
w, h, spp = 1920, 1080, 256
def run(world, batch):
    pt = W.shirley_path_tracer(w, h, max_wavefronts=8, rng_mode=W.RNG_PIXEL, tile_rank=0, tile_world=world, batch=batch)
    pt.render(batch); pt.synchronize()
    t0 = time.perf_counter(); pt.render(spp); pt.synchronize(); el = time.perf_counter() - t0
    pt.close()
    return el / spp * 1e3
base = run(1, 64)  # what bench.py runs at N = 1
print(f"N=1 batch 64: {base:.4f} ms/sample (batch 32: {run(1, 32):.4f})")
for world in (2, 4, 8):
    for batch in (32, 64, 128):
        ms = run(world, batch)
        print(f"N={world} batch {batch}: {ms:.4f} ms/sample of the slab -> speed-up {base / ms:.2f}x, efficiency {base / ms / world:.2f}", flush=True)
