"""tools/stamps_probe.py -- run on the GPU box with WFPT_LIB=build/libwfpt_stamps.so: where does a wave of the middle bounce launches spend
its cycles? (shade | walk | barrier wait | compaction + stores), plus inner visits per wave-item."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import wavefront_path_tracer_amd as W
import datetime
from wavefront_path_tracer_amd import _build
info = _build.build_info()
print(f"# command: WFPT_LIB={os.environ.get('WFPT_LIB', '')} python tools/stamps_probe.py {' '.join(sys.argv[1:])}   commit {info.get('git_head')} "
      f"(dirty at build: {info.get('git_dirty')}; the -DWFPT_STAMPS=1 diagnostic build of the same sources)   device: {torch.cuda.get_device_name(0)}   "
      f"{datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%M:%SZ')}")
flags = W.FLAG_EXACT_TRAVERSAL if "--exact" in sys.argv else 0
pt = W.shirley_path_tracer(1920, 1080, seed=1, max_wavefronts=8, batch=64, flags=flags)
pt.render(64)
out = np.zeros(16, "<u8")
W.lib().wfpt_debug_read_stamps(pt.handle, W._p(out), 1)
pt.render(64)
W.lib().wfpt_debug_read_stamps(pt.handle, W._p(out), 0)
tot = out[:4].sum()
items = max(int(out[4]), 1)
print("wave-items", items, "live rays per wave-item %.1f" % (out[5] / items))
for name, v in zip(("shade (item start -> walk)", "walk", "barrier wait", "compaction + stores"), out[:4]):
    print(f"  {name:28s} {v / items:10.0f} cycles per wave-item  {100.0 * v / tot:5.1f} %")
if out[8]:
    print("  wave-level inner visits per wave-item %.1f, leaf rounds %.1f, lanes per inner visit %.1f" % (out[8] / items, out[9] / items, out[10] / max(out[8], 1)))
    print("  walk cycles per wave-level inner visit %.0f" % (out[1] / out[8]))
if out[11]:
    print("  of shade: %.0f cycles until the record's slot is known (first_seg table + search of the segment bases), %.0f more until the record has arrived" % (out[11] / items, out[12] / items))
pt.close()
