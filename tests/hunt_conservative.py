"""tests/hunt_conservative.py [width height frames bounces seed leaf_exact [mesh_triangles]] -- CPU only. Runs the oracle's stage chain on the seeded Shirley scene (or, with mesh_triangles > 0, on BASELINE config 5's triangle soup)
and, before every extend, traces the wavefront's rays twice: with the reference's traversal and with the oracle's MODEL of the
device's conservative traversal (oracle/wfpt_oracle.c: trace_ray_model). Prints every ray whose reported hit differs."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

w, h, frames, bounces = (int(a) for a in (sys.argv[1:5] + ["1920", "1080", "2", "8"][len(sys.argv) - 1:]))
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
MESH = int(sys.argv[7]) if len(sys.argv) > 7 else 0  # > 0: BASELINE config 5's triangle soup with this many triangles instead of the Shirley scene
LEAF_EXACT = int(sys.argv[6]) if len(sys.argv) > 6 else 3  # 3: the device's walk (one leaf-box verdict, after the walk); 2: round 3's walk (a verdict per changed leaf); 1: no near-tie hand-over; 0: every box merely conservative (1, 0: NOT equivalent)
O.build()
L = O.lib()
L.orc_model_mismatches.restype = C.c_uint32
L.orc_model_mismatches.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_traversal_model import _extent
if MESH:
    tris, mt = O.scene_random_mesh(MESH, seed)
    tris, nodes = O.build_bvh_triangles(tris, 32)
    cam, ip, vw = O.mesh_camera(w, h)
    o = O.Oracle(w, h, np.zeros(1, O.SPHERE), mt, nodes, cam, ip, vw, triangles=tris, max_wavefronts=bounces)
    extent = _extent(O, nodes, cam, None)  # (no safe region for triangles: every origin takes the free walk)
else:
    o = O.shirley_oracle(w, h, seed=seed, max_wavefronts=bounces)
    sp, mt = O.scene_book_one_final(seed)
    sp, nodes = O.build_bvh(sp)
    cam, _, _ = O.shirley_camera(w, h)
    extent = _extent(O, nodes, cam, sp)
print("extent", extent[:3], "margin", extent[:3] * 2.0 ** -17, "safe ball: centre", extent[3:6], "radius", float(np.sqrt(max(extent[6], 0))))
out = np.zeros((4096, 2), "<u4")
total = 0
for frame in range(1, frames + 1):
    n = w * h
    o.set_frame(frame, 0)
    o.reset_image()
    o.set_counters([0, 0, n])
    o.generate_rays((w + 7) // 8, (h + 7) // 8, True)
    n = int(o.counters()[2])
    ext = O.workgroup_size_64(max(n, 65))
    for b in range(bounces):
        cnt = L.orc_model_mismatches(o.h, n, O._p(extent), LEAF_EXACT, O._p(out), len(out))
        total += n
        if cnt:
            rays = o.rays(n)
            print(f"frame {frame} bounce {b}: {cnt} of {n} rays differ")
            for k in range(min(cnt, 6)):
                i, kind = int(out[k, 0]), int(out[k, 1])
                r = rays[i]
                print("   ray", i, "kind", {1: "model hits, reference misses", 2: "reference hits, model misses", 3: "different hit"}[kind],
                      "o", r["origin"][:3], "d", r["direction"][:3], "ref", o.trace_bvh(r), "brute", o.trace_brute(r))
        o.extend(*ext)
        c = o.counters()
        misses, hits = int(c[0]), int(c[1])
        c[2] = 0
        o.set_counters(c)
        o.shade(*O.workgroup_size_64(max(hits, 65)))
        o.miss(*O.workgroup_size_64(max(misses, 65)))
        o.swap_ray_queues()
        n = hits
        ext = O.workgroup_size_64(max(n, 65))
        o.set_counters([0, 0, n, 0])
        if n == 0:
            break
L.orc_model_handovers.restype = C.c_uint64
print("rays compared:", total, "handed over by the model (near-tie / leaf box):", int(L.orc_model_handovers()))
