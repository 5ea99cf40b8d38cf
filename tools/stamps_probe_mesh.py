"""tools/stamps_probe_mesh.py -- run on the GPU box with WFPT_LIB=build/libwfpt_stamps.so: what do the lanes of the refill traversal (config 5,
1 M triangles) do? Per launch kind (first / middle): loop iterations, lanes holding a ray, lanes in the four-box visit steps, leaf rounds and the lanes
in them, refills, lanes that sat at a leaf waiting for a round, and the shader cycles of the three sections."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import wavefront_path_tracer_amd as W
import datetime
from wavefront_path_tracer_amd import _build
info = _build.build_info()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print(f"# command: WFPT_LIB={os.environ.get('WFPT_LIB', '')} python tools/stamps_probe_mesh.py {' '.join(sys.argv[1:])}   commit {info.get('git_head')} "
      f"(dirty at build: {info.get('git_dirty')}; the -DWFPT_STAMPS=1 diagnostic build)   device: {torch.cuda.get_device_name(0)}   "
      f"{datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%M:%SZ')}   {spp} samples in flight, 1920x1080, 1 M triangles, 8 bounces")
pt = W.mesh_path_tracer(1920, 1080, 1000000, seed=1, max_wavefronts=8, batch=spp)
pt.render(spp)
out = np.zeros(16, "<u8")
for which in (1, 2):
    W.lib().wfpt_debug_read_stamps_ex(pt.handle, which, W._p(out), 1)
pt.render(spp)
for which, name in ((1, "first launch (primary rays)"), (2, "middle launches (wavefronts 1-7)")):
    W.lib().wfpt_debug_read_stamps_ex(pt.handle, which, W._p(out), 0)
    o = out.astype(np.float64)
    it = max(o[0], 1.0)
    print(f"{name}: {int(o[13])} waves, {o[0] / max(o[13], 1):.0f} loop iterations per wave")
    print(f"  lanes holding a ray per iteration            {o[1] / it:5.1f} of 64")
    print(f"  visit steps: {100 * o[2] / it:5.1f} % of the iterations, {o[3] / max(o[2], 1):5.1f} lanes each")
    print(f"  leaf rounds: {100 * o[4] / it:5.1f} % of the iterations, {o[5] / max(o[4], 1):5.1f} lanes each")
    print(f"  lanes at a leaf waiting for a round, per iteration {o[8] / it:5.2f}")
    print(f"  refill passes: {100 * o[6] / it:5.2f} % of the iterations, {o[7] / max(o[6], 1):5.1f} lanes each")
    tot = max(o[12], 1.0)
    print(f"  shader cycles: refill {100 * o[9] / tot:5.1f} %, visit step {100 * o[10] / tot:5.1f} %, leaf round + result {100 * o[11] / tot:5.1f} %  "
          f"({o[10] / max(o[2], 1):.0f} cycles per visit step, {o[11] / max(o[4], 1):.0f} per leaf round incl. the iterations without one)")
pt.close()
