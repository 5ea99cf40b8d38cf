"""diagnostic: where does the 3840x2160 image differ from the oracle's?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa
import wavefront_path_tracer_amd as W
from oracle import oracle as O
from helpers import inputs_for, make_oracle, make_tracer
w, h, bounces = 3840, 2160, 4
for spp, batch, flags in ((1, 1, 0), (2, 2, 0), (16, 16, 0), (17, 16, 0), (2, 2, W.FLAG_UNFUSED)):
    pt = make_tracer(W, "shirley", w, h, max_wavefronts=bounces, batch=batch, flags=flags)
    pt.render(spp)
    o = make_oracle(O, inputs_for(O, "shirley", w, h), w, h, max_wavefronts=bounces)
    o.render(spp)
    a, b = pt.accumulated(), o.accumulated()
    bad = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[:, 0]
    print(f"spp {spp} batch {batch} flags {flags}: {len(bad)} pixels differ", flush=True)
    if len(bad):
        print("   first", bad[:8], "last", bad[-8:], "rows", np.unique(bad // w)[:10], "x%4", np.bincount(bad % 4, minlength=4))
        k = bad[0]
        print("   gpu", a[k], "oracle", b[k], "diff", a[k] - b[k])
    pt.close(); o.close()
