#!/usr/bin/env python3
"""tools/model_node_format.py [N_TRIANGLES]: the CPU gate of VERDICT r4 item 3b (a four-wide node whose planes need no integer conversion): how much the
child boxes of config 5's four-wide tree GROW under (a) the shipped format, 8-bit offsets in the frame of the children's union (struct Node4), and (b)
absolute fp16 planes (24 x 2 bytes + 4 child words = the same 64-byte line; one v_fma_mix_f32 per plane instead of v_cvt_f32_ubyteN + v_fma_f32), both
rounded outwards and grown by the product's margin, measured by surface area: the expected visits of a ray that crosses the root box. No GPU needed.
Result and pricing: profiles/r05_rejected_experiments.txt."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavefront_path_tracer_amd as W
n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
scene = W.Scene.random_mesh(n_tri)
bvh = W.BVHTree(n_tri); bvh.build_bvh_tree_triangles(scene.triangles, 32)
nd = bvh.nodes
lo = nd["aabb_min"].astype(np.float64); hi = nd["aabb_max"].astype(np.float64)
lf = nd["left_first"].astype(np.int64); pc = nd["prim_count"].astype(np.int64)
extent = np.maximum(np.abs(lo[[0]]), np.abs(hi[[0]])).max(axis=0)
margin = extent * 2.0 ** -17
def area(l, h):
    e = np.maximum(h - l, 0); return e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]
def f16_down(x):
    h = x.astype(np.float16); bad = h.astype(np.float64) > x
    h[bad] = np.nextafter(h[bad], np.float16(-np.inf)); return h.astype(np.float64)
def f16_up(x):
    h = x.astype(np.float16); bad = h.astype(np.float64) < x
    h[bad] = np.nextafter(h[bad], np.float16(np.inf)); return h.astype(np.float64)
# four-wide nodes: BFS over binary inner nodes that head a four-wide node
heads = np.array([0]); tot = {"exact": 0.0, "q8": 0.0, "f16": 0.0}; leaf = {"exact": 0.0, "q8": 0.0, "f16": 0.0}; n4 = 0
while len(heads):
    n4 += len(heads)
    l = lf[heads]; kids = []
    for c in (l, l + 1):
        inner = pc[c] == 0
        # inner child: its two children; leaf child: itself (+ a hole)
        k0 = np.where(inner, lf[c], c); k1 = np.where(inner, lf[c] + 1, -1)
        kids += [k0, k1]
    K = np.stack(kids, 1)            # [n, 4], -1 = absent
    valid = K >= 0; Kc = np.where(valid, K, 0)
    clo = lo[Kc] - margin; chi = hi[Kc] + margin     # [n, 4, 3]
    big = 1e30
    ulo = np.where(valid[..., None], clo, big).min(1); uhi = np.where(valid[..., None], chi, -big).max(1)
    # q8 in the union's frame, power-of-two scale
    sc = 2.0 ** np.ceil(np.log2(np.maximum((uhi - ulo) / 255.0, 1e-30)))
    # (the host grows the scale until the upper planes fit 8 bits; ceil(log2) of extent / 255 already guarantees 255 * scale >= extent)
    qlo = ulo[:, None, :] + np.floor((clo - ulo[:, None, :]) / sc[:, None, :]) * sc[:, None, :]
    qhi = ulo[:, None, :] + np.ceil((chi - ulo[:, None, :]) / sc[:, None, :]) * sc[:, None, :]
    hlo = f16_down(clo); hhi = f16_up(chi)
    isleaf = pc[Kc] > 0
    for name, (a, b) in {"exact": (lo[Kc], hi[Kc]), "q8": (qlo, qhi), "f16": (hlo, hhi)}.items():
        ar = area(a.reshape(-1, 3), b.reshape(-1, 3)).reshape(K.shape)
        tot[name] += ar[valid & ~isleaf].sum(); leaf[name] += ar[valid & isleaf].sum()
    heads = K[valid & ~isleaf]
root = area(lo[[0]], hi[[0]])[0]
print(f"{n_tri} triangles: {n4} four-wide nodes; expected visits of a ray that crosses the root box (surface-area measure), nodes below the root | leaf children")
for name in ("exact", "q8", "f16"):
    print(f"  {name:6s} {tot[name] / root:9.3f} ({tot[name] / tot['exact']:.4f} x exact) | {leaf[name] / root:8.3f} ({leaf[name] / leaf['exact']:.4f} x exact)")
