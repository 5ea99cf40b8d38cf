import sys, time
sys.path.insert(0, '/root/repo')
import torch
import wavefront_path_tracer_amd as W
w, h = 1920, 1080
def mk(batch): return W.shirley_path_tracer(w, h, max_wavefronts=8, batch=batch)
def timeit(fn, reps=5):
    fn(); best=1e9
    for _ in range(reps):
        t0=time.perf_counter(); fn(); best=min(best,time.perf_counter()-t0)
    return best
for total in (20, 64):
    a = mk(total)
    def one():
        a.render(total); a.synchronize()
    t1 = timeit(one)
    a.close()
    b, c = mk(total//2), mk(total//2)
    def two():
        b.render(total//2); c.render(total//2); b.synchronize(); c.synchronize()
    t2 = timeit(two)
    b.close(); c.close()
    ctxs=[mk(total//4) for _ in range(4)]
    def four():
        for x in ctxs: x.render(total//4)
        for x in ctxs: x.synchronize()
    t4 = timeit(four)
    for x in ctxs: x.close()
    rays = 5326459*total
    print(f"{total} spp: one context {rays/t1/1e9:.2f} Grays/s; two contexts of {total//2} in flight each, concurrent streams: {rays/t2/1e9:.2f}; four of {total//4}: {rays/t4/1e9:.2f}", flush=True)
