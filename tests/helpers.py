"""Builders shared by the parity tests: the same scene / camera / parameters for oracle and product."""
import numpy as np


def shirley_inputs(O, width, height, seed=1):
    sp, mt = O.scene_book_one_final(seed)
    sp, nodes = O.build_bvh(sp)
    cam, ip, vw = O.shirley_camera(width, height)
    return sp, mt, nodes, cam, ip, vw


def simple_inputs(O, width, height):
    """scene.rs:12-46 (5 spheres, all three materials, nested dielectric) with the commented-out camera of
    main.rs:21-22: (0,0,1) looking at (0,0,-1); vfov 90, no defocus."""
    sp, mt = O.scene_new()
    sp, nodes = O.build_bvh(sp)
    cam, ip, vw = O.camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), 90.0, 0.0, 10.0, 0.1, 100.0, width, height)
    return sp, mt, nodes, cam, ip, vw


def make_oracle(O, inputs, width, height, **kw):
    sp, mt, nodes, cam, ip, vw = inputs
    return O.Oracle(width, height, sp, mt, nodes, cam, ip, vw, **kw)


def make_tracer(W, kind, width, height, seed=1, **kw):
    """Product-side PathTracer on the same inputs (its own host code builds scene, BVH and camera)."""
    if kind == "shirley":
        return W.shirley_path_tracer(width, height, seed=seed, **kw)
    scene = W.Scene.new()
    cc = W.CameraController(W.Camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0)), 90.0, 0.0, 10.0, 0.1, 100.0)
    return W.PathTracer(scene, W.RenderParameters(cc, (width, height)), **kw)


def inputs_for(O, kind, width, height, seed=1):
    return shirley_inputs(O, width, height, seed) if kind == "shirley" else simple_inputs(O, width, height)


def sorted_rays(r):
    return np.sort(r, order=["pixel_idx"])


def mesh_inputs(O, width, height, n_triangles, edge_scale=1.0, seed=1, n_bins=32):
    """BASELINE config 5's triangle soup (optionally with longer edges so that small meshes get hit)."""
    tris, mt = O.scene_random_mesh(n_triangles, seed)
    if edge_scale != 1.0:
        tris["e1"] *= np.float32(edge_scale)
        tris["e2"] *= np.float32(edge_scale)
    tris, nodes = O.build_bvh_triangles(tris, n_bins)
    cam, ip, vw = O.mesh_camera(width, height)
    return tris, mt, nodes, cam, ip, vw


def make_mesh_oracle(O, inputs, width, height, **kw):
    tris, mt, nodes, cam, ip, vw = inputs
    return O.Oracle(width, height, np.zeros(1, O.SPHERE), mt, nodes, cam, ip, vw, triangles=tris, **kw)


def make_mesh_tracer(W, width, height, n_triangles, edge_scale=1.0, seed=1, n_bins=32, **kw):
    scene = W.Scene.random_mesh(n_triangles, seed)
    if edge_scale != 1.0:
        scene.triangles["e1"] *= np.float32(edge_scale)
        scene.triangles["e2"] *= np.float32(edge_scale)
    cc = W.CameraController(W.Camera((0.0, 0.0, 30.0), (0.0, 0.0, 0.0)), 40.0, 0.0, 10.0, 0.1, 100.0)
    return W.PathTracer(scene, W.RenderParameters(cc, (width, height)), mesh_bins=n_bins, **kw)


# ---- rays and a scene built to sit on the edges of the free walks' equivalence argument (tests/test_gpu_round3.py,
# tests/test_traversal_model.py)
def next_ulp(x, k):
    """x moved by k ulps (float32)."""
    x = np.float32(x)
    for _ in range(abs(k)):
        x = np.nextafter(x, np.float32(np.inf if k > 0 else -np.inf), dtype=np.float32)
    return x


def adversarial_rays(W, spheres, nodes, n_max):
    """Rays the conservative-box argument has to survive: head-on at the face centres of leaf and inner
    boxes (two direction components exactly zero: infinite inverses), origins exactly on the box planes at the tops / sides
    of spheres travelling along the plane (grazing), rays passing the centre of a sphere at r (1 +- k ulp) (near-tangent,
    where the exact test decides by the last bit), the same tilted by a few ulp, and rays towards pairs of spheres whose
    hits differ by a few ulp."""
    rays = []

    def add(o, d):
        rays.append((np.asarray(o, np.float32), np.asarray(d, np.float32)))

    axes = np.eye(3, dtype=np.float32)
    boxes = [(nd["aabb_min"], nd["aabb_max"]) for i, nd in enumerate(nodes) if i != 1]
    rng = np.random.default_rng(3)
    for lo, hi in boxes[: n_max // 40]:
        c = (lo + hi) * np.float32(0.5)
        for ax in range(3):
            for sgn in (-1.0, 1.0):
                o = c.copy()
                o[ax] = (hi[ax] + np.float32(3.0)) if sgn < 0 else (lo[ax] - np.float32(3.0))
                add(o, axes[ax] * np.float32(sgn))                       # head-on at the face centre, axis-parallel
                d = axes[ax] * np.float32(sgn)
                d[(ax + 1) % 3] = np.float32(1e-7)
                add(o, d)                                                 # almost axis-parallel: huge inverse
                e = c.copy()                                              # along an EDGE of the box, in the face plane
                e[ax] = hi[ax] if sgn > 0 else lo[ax]
                e[(ax + 1) % 3] = lo[(ax + 1) % 3] - np.float32(2.0)
                add(e, axes[(ax + 1) % 3])
    pick = rng.permutation(len(spheres))[: n_max // 60]
    for i in pick:
        c = spheres["center"][i][:3].astype(np.float32)
        r = np.float32(spheres["radius"][i])
        for ax in range(3):
            t_ax = (ax + 1) % 3
            for k in (-4, -2, -1, 0, 1, 2, 4):
                # origin on (or k ulp off) the plane that touches the sphere at its extreme point on axis `ax`, travelling in the plane
                o = c.copy()
                o[ax] = next_ulp(c[ax] + r, k)
                o[t_ax] = c[t_ax] - np.float32(2.0) * r - np.float32(1.0)
                add(o, axes[t_ax])
                d = axes[t_ax].copy()
                d[ax] = np.float32(k) * np.float32(1e-8)                  # tilted by next to nothing towards / away from the sphere
                add(o, d)
            # near-tangent in a random direction: pass the centre at distance r (1 + k 2^-23)
            u = rng.normal(size=3).astype(np.float32)
            u /= np.float32(np.linalg.norm(u))
            v = np.cross(u, rng.normal(size=3)).astype(np.float32)
            v /= np.float32(np.linalg.norm(v))
            for k in (-3, -1, 0, 1, 3):
                p = c + v * (r * (np.float32(1.0) + np.float32(k) * np.float32(2.0 ** -23)))
                add(p - u * np.float32(5.0), u)
    # neighbours: rays through the midpoints between close sphere pairs (two hits within a few ulp of each other are most
    # likely where spheres nearly touch or overlap)
    cs = spheres["center"][:, :3].astype(np.float64)
    for i in pick[:40]:
        dist = np.linalg.norm(cs - cs[i], axis=1)
        dist[i] = np.inf
        j = int(np.argmin(dist))
        m = ((cs[i] + cs[j]) * 0.5).astype(np.float32)
        for _ in range(4):
            u = rng.normal(size=3).astype(np.float32)
            u /= np.float32(np.linalg.norm(u))
            add(m - u * np.float32(4.0), u)
    rays = rays[:n_max]
    out = np.zeros(len(rays), W.RAY)
    for k, (o, d) in enumerate(rays):
        out["origin"][k, :3] = o
        out["origin"][k, 3] = 1.0
        out["direction"][k, :3] = d
    with np.errstate(all="ignore"):
        out["inv_direction"] = (np.float32(1.0) / out["direction"][:, :3]).astype("<f4")
    out["pixel_idx"] = np.arange(len(out), dtype="<u4") % 1024
    return out


def close_pairs_scene(orc):
    """A scene made for ties: pairs of spheres whose centres differ by 1-4 ulp (equal radii), nested and overlapping
    spheres, plus a ground sphere; BVH built by the oracle's builder."""
    rng = np.random.default_rng(11)
    sp = np.zeros(81, orc.SPHERE)
    mt = np.zeros(3, orc.MATERIAL)
    mt["albedo"][:] = (0.7, 0.6, 0.5, 1.0)
    mt["material_type"] = (0, 1, 2)
    mt["refract_index"][2] = 1.5
    sp["center"][0] = (0.0, -1000.0, 0.0, 1.0)
    sp["radius"][0] = 1000.0
    for k in range(40):
        c = rng.uniform(-6, 6, 3).astype(np.float32)
        c[1] = np.float32(abs(c[1]) * 0.3 + 0.5)
        r = np.float32(rng.uniform(0.2, 0.6))
        c2 = c.copy()
        c2[k % 3] = next_ulp(c2[k % 3], 1 + k % 4)
        sp["center"][1 + 2 * k, :3], sp["center"][2 + 2 * k, :3] = c, c2
        sp["radius"][1 + 2 * k] = r
        sp["radius"][2 + 2 * k] = r if k % 2 == 0 else next_ulp(r, 1)
    sp["center"][:, 3] = 1.0
    sp["material_idx"] = np.arange(len(sp)) % 3
    sp["material_type"] = np.arange(len(sp)) % 3
    return sp, mt


def adversarial_mesh(tris):
    """Mutates a TRIANGLE array (the same way for oracle and product) into a mesh made for the edges of the traversal argument:
    exact duplicates (two hits at bit-equal distances: the visit order decides, extend.wgsl:190-207), triangles that share an
    edge, and axis-aligned triangles whose boxes have no thickness."""
    n = len(tris)
    for k in range(0, min(200, n - 1), 2):
        tris["v0"][k + 1], tris["e1"][k + 1], tris["e2"][k + 1] = tris["v0"][k], tris["e1"][k], tris["e2"][k]
    for k in range(200, min(300, n - 1), 2):  # the second triangle completes the parallelogram: the diagonal is shared
        tris["v0"][k + 1] = tris["v0"][k] + tris["e1"][k] + tris["e2"][k]
        tris["e1"][k + 1], tris["e2"][k + 1] = -tris["e1"][k], -tris["e2"][k]
    for k in range(300, min(380, n)):
        ax = k % 3
        tris["e1"][k, ax] = 0.0
        tris["e2"][k, ax] = 0.0
    return tris


def adversarial_rays_mesh(W, tris, nodes, n_max):
    """Rays for a triangle mesh: at vertices, edge midpoints and centroids head-on and axis-parallel, in the planes of the boxes
    of flat triangles, along box edges of leaves and inner nodes, and a few ulp around each of them."""
    rays = []
    axes = np.eye(3, dtype=np.float32)

    def add(o, d):
        rays.append((np.asarray(o, np.float32), np.asarray(d, np.float32)))

    rng = np.random.default_rng(7)
    for k in range(min(len(tris), n_max // 40)):
        a = tris["v0"][k].astype(np.float32)
        b = (a + tris["e1"][k]).astype(np.float32)
        c = (a + tris["e2"][k]).astype(np.float32)
        nrm = np.cross(tris["e1"][k], tris["e2"][k]).astype(np.float32)
        ln = np.float32(np.linalg.norm(nrm))
        nrm = nrm / ln if ln > 0 else axes[k % 3]
        for p in (a, b, c, (a + b) * np.float32(0.5), (b + c) * np.float32(0.5), (a + b + c) / np.float32(3.0)):
            add(p + nrm * np.float32(4.0), -nrm)                       # head-on along the normal
            ax = int(np.argmax(np.abs(nrm)))
            o = p.copy()
            o[ax] += np.float32(5.0)
            add(o, -axes[ax])                                            # axis-parallel: two infinite inverses
        lo, hi = np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)
        for ax in range(3):                                              # in the planes of the triangle's own box
            t_ax = (ax + 1) % 3
            for plane in (lo[ax], hi[ax]):
                o = (lo + hi) * np.float32(0.5)
                o[ax] = plane
                o[t_ax] = lo[t_ax] - np.float32(2.0)
                add(o, axes[t_ax])
                o2 = o.copy()
                o2[ax] = next_ulp(plane, int(rng.integers(-2, 3)))
                add(o2, axes[t_ax])
    for i, nd in enumerate(nodes[: n_max // 60]):
        if i == 1:
            continue
        lo, hi = nd["aabb_min"], nd["aabb_max"]
        c = (lo + hi) * np.float32(0.5)
        for ax in range(3):
            o = c.copy()
            o[ax] = hi[ax] + np.float32(3.0)
            add(o, -axes[ax])
            e = c.copy()
            e[ax] = hi[ax]
            e[(ax + 1) % 3] = lo[(ax + 1) % 3] - np.float32(2.0)
            add(e, axes[(ax + 1) % 3])
    rays = rays[:n_max]
    out = np.zeros(len(rays), W.RAY)
    for k, (o, d) in enumerate(rays):
        out["origin"][k, :3] = o
        out["origin"][k, 3] = 1.0
        out["direction"][k, :3] = d
    with np.errstate(all="ignore"):
        out["inv_direction"] = (np.float32(1.0) / out["direction"][:, :3]).astype("<f4")
    out["pixel_idx"] = np.arange(len(out), dtype="<u4") % 1024
    return out


def grazing_rays_mesh(W, tris, n_max, seed=11):
    """Rays for which Moeller-Trumbore is worst conditioned: within 1e-7 .. 1e-2 rad of a triangle's plane, passing
    an edge at -1e-3 .. +1e-3 (relative to the edge's length; negative = inside) as seen in the plane. The rounding slack of the
    primitive test grows like 1 / (cos(angle to the normal) * sin(angle between the edges)): such rays are where a hit could be reported
    for a ray that passes outside the triangle's box by more than the free walks' margin."""
    rng = np.random.default_rng(seed)
    rays = []
    per = max(n_max // 12, 1)
    for k in rng.permutation(len(tris))[:per]:
        a = tris["v0"][k].astype(np.float64)
        e1, e2 = tris["e1"][k].astype(np.float64), tris["e2"][k].astype(np.float64)
        nrm = np.cross(e1, e2)
        ln = np.linalg.norm(nrm)
        if ln == 0:
            continue
        nrm /= ln
        for (p0, ed) in ((a, e1), (a, e2), (a + e1, e2 - e1)):
            le = np.linalg.norm(ed)
            if le == 0:
                continue
            along = ed / le
            out = np.cross(along, nrm)                      # in the plane, perpendicular to the edge
            if np.dot(out, (a + (e1 + e2) / 3.0) - p0) > 0:  # make it point away from the triangle
                out = -out
            for _ in range(4):
                s = rng.uniform(0.05, 0.95)
                off = rng.choice([-1e-3, -1e-5, -1e-7, 0.0, 1e-7, 1e-5, 1e-3]) * le
                tilt = 10.0 ** rng.uniform(-7, -2) * rng.choice([-1.0, 1.0])
                phi = rng.uniform(0, 2 * np.pi)
                d_in_plane = np.cos(phi) * along + np.sin(phi) * out
                d = d_in_plane + tilt * nrm
                d /= np.linalg.norm(d)
                target = p0 + s * ed + off * out
                dist = rng.uniform(2.0, 40.0)
                rays.append(((target - d * dist).astype(np.float32), d.astype(np.float32)))
    rays = rays[:n_max]
    res = np.zeros(len(rays), W.RAY)
    for k, (o, d) in enumerate(rays):
        res["origin"][k, :3] = o
        res["origin"][k, 3] = 1.0
        res["direction"][k, :3] = d
    with np.errstate(all="ignore"):
        res["inv_direction"] = (np.float32(1.0) / res["direction"][:, :3]).astype("<f4")
    res["pixel_idx"] = np.arange(len(res), dtype="<u4") % 1024
    return res
