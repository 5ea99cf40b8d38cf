"""tools/load_order_probe.py -- run on the GPU box: import the package BEFORE torch, render, then import torch and use the GPU
through it too. Before round 3 this order ended in torch's "No HIP GPUs are available" (two HIP/HSA runtimes in one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavefront_path_tracer_amd as W
assert "torch" not in sys.modules
assert W.device_count() >= 1, "no HIP device through libwfpt.so"
pt = W.shirley_path_tracer(200, 120, seed=1, max_wavefronts=4)
pt.render(2)
a = pt.accumulated()
import torch
assert torch.cuda.is_available(), "torch sees no GPU after libwfpt.so was loaded first"
x = torch.arange(1024, device="cuda", dtype=torch.float32)
assert float((x * 2).sum().item()) == 1023 * 1024.0
pt.render(2)  # and the context still works after torch initialised its side
b = pt.accumulated()
assert np.isfinite(a).all() and np.isfinite(b).all() and (b >= a).all()
maps = open("/proc/self/maps").read()
hips = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})
hsas = sorted({l.split()[-1] for l in maps.splitlines() if "libhsa-runtime64" in l})
print("load-order probe ok: package first, then torch; HIP runtimes mapped:", hips, "HSA runtimes mapped:", hsas)
assert len(hips) == 1 and len(hsas) == 1
pt.close()
