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
