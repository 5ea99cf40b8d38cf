"""The oracle's diagnostics behind tests/model_schedule.py (design tools, CPU only): they must agree with the chain they model."""
import ctypes as C

import numpy as np

from helpers import inputs_for, make_oracle


def _rounds(o, n, init=None, flags=0):
    segs = np.zeros((n, 16), np.uint8)
    nl = np.zeros(n, np.uint8)
    if init is None:
        o.L.orc_ray_rounds(o.h, n, segs.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.c_void_p))
    else:
        o.L.orc_ray_rounds_init(o.h, n, init.ctypes.data_as(C.c_void_p), flags, segs.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.c_void_p))
    return segs.astype(np.int64), nl.astype(np.int64)


def test_round_diagnostics_match_the_traversal_statistics(orc):
    w, h = 96, 64
    o = make_oracle(orc, inputs_for(orc, "shirley", w, h), w, h, max_wavefronts=4)
    n = w * h
    o.set_frame(1, 0); o.reset_image(); o.set_counters([0, 0, n]); o.generate_rays(w // 8, h // 8, False)
    segs, nl = _rounds(o, n)
    o.extend(*orc.workgroup_size_64(n))
    st = o.trace_stats()
    # every leaf of the seeded scene holds one sphere: leaf visits == sphere tests, inner visits == nodes - tests
    assert int(nl.sum()) == int(st["sphere_tests"]) and int(segs.sum()) == int(st["node_visits"]) - int(st["sphere_tests"])
    # the same traversal started from "nothing hit yet" without the root test is the plain diagnostic
    inf = np.full(n, 1e30, np.float32)
    segs0, nl0 = _rounds(o, n, inf, 0)
    assert np.array_equal(segs0, segs) and np.array_equal(nl0, nl)
    # leaving pairs of missed boxes alone (the device's kBoxMiss) only ever removes visits and sphere tests
    segs2, nl2 = _rounds(o, n, inf, 2)
    assert (segs2.sum(axis=1) <= segs.sum(axis=1)).all() and (nl2 <= nl).all() and segs2.sum() < segs.sum()
    # t of one primitive per ray: never below the nearest hit extend reports, equal to it where that primitive is the hit
    hits = int(o.counters()[1])
    hq = o.hits(hits)
    prim = int(np.bincount(hq["sphere_idx"]).argmax())
    t = np.zeros(n, np.float32)
    o.L.orc_prim_hit_t(o.h, n, prim, t.ctypes.data_as(C.c_void_p))
    sel = hq["sphere_idx"] == prim
    assert sel.any() and np.array_equal(t[hq["ray_idx"][sel]], hq["t"][sel])
    assert (t[hq["ray_idx"][~sel]] >= hq["t"][~sel]).all()
    o.close()


def test_schedule_model_runs(orc, monkeypatch, capsys):
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("model_schedule", os.path.join(root, "tests", "model_schedule.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    # wave cost of 64 identical lanes = the lane's own cost: utilisation 1
    segs = np.tile(np.array([[3, 2] + [0] * 14], np.int32), (64, 1))
    nl = np.full(64, 2, np.int32)
    assert m.model_base(segs, nl) == m.lane_work(segs, nl)[0]
    # one long lane sets the cost of the whole wave
    segs[0, 0] = 30
    assert m.model_base(segs, nl) == m.lane_work(segs, nl)[0]
    assert m.model_sort(np.concatenate([segs, segs]), np.concatenate([nl, nl]), np.arange(128, dtype=np.float64), 128) == 2 * m.model_base(segs, nl)


def test_binning_model_scores(orc):
    """tests/model_binning.py (global binning of the hit queue by key) and tests/model_leaf_threshold.py (if-if schedule with a leaf-lane
    threshold): the scoring functions on hand-made waves."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tests", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    mb = load("model_binning")
    # two kinds of rays, cheap and dear, alternating: in hit order every wave pays for the dear ones, binned by kind half of them do
    n = 1024
    segs = np.zeros((n, 16), np.int32)
    segs[:, 0] = np.where(np.arange(n) % 2 == 0, 2, 20)
    nl = np.ones(n, np.int32)
    mat = (np.arange(n) % 2).astype(np.int64)
    base = mb.score(segs, nl, mat)
    binned = mb.score(segs, nl, mat, mb.binned_order(np.arange(n) % 2))
    assert binned["rounds"] < 0.6 * base["rounds"] and binned["trace"] < 0.6 * base["trace"]
    assert binned["shade"] < base["shade"]  # a wave of one material runs one branch
    assert mb.score(segs, nl, mat, mb.binned_order(np.zeros(n, np.int64)))["trace"] == base["trace"]  # one class: the hit order
    ml = load("model_leaf_threshold")
    # 64 identical lanes: every schedule costs the lane's own work
    s = np.tile(np.array([[3, 2] + [0] * 14], np.int32), (64, 1))
    l = np.full(64, 2, np.int32)
    cost, it_v, it_l = ml.sim_ifif(s, l, 8)
    assert (it_v, it_l) == (5, 2) and cost == 5 * ml.C_VISIT + 2 * ml.C_LEAF
