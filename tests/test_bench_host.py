"""bench.py's host logic that needs no GPU: which committed profile a run may quote (a 4K run quoted the 1080p
profile's traffic as "of this command"), and the CPU share the CPU baseline is timed on (item 4)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_profile_of_another_workload_is_no_match():
    hd = bench.workload_key(bench.parse([]))
    assert hd["width"] == 1920 and hd["height"] == 1080 and hd["spp"] == 64 and hd["rng_mode"] == "dispatch" and hd["samples_in_flight"] == 64
    # the committed 1080p profile of the fused loop is the default run's, and nobody else's
    assert bench.load_pmc("shirley", "fused", hd) is not None
    for other in (["--width", "3840", "--height", "2160"], ["--spp", "256"], ["--bounces", "4"], ["--rng-mode", "pixel"], ["--seed", "2"],
                  ["--batch", "16"], ["--width", "400", "--height", "225", "--spp", "4", "--bounces", "4"]):
        assert bench.load_pmc("shirley", "fused", bench.workload_key(bench.parse(other))) is None, other
    assert bench.load_pmc("shirley", "fused", bench.workload_key(bench.parse([]), world=8)) is None
    # the mesh profile belongs to the 1 M-triangle soup only
    mesh = bench.workload_key(bench.parse(["--scene", "mesh"]))
    assert mesh["triangles"] == 1000000 and bench.load_pmc("mesh", "fused", mesh) is not None
    assert bench.load_pmc("mesh", "fused", bench.workload_key(bench.parse(["--scene", "mesh", "--triangles", "5000"]))) is None


def test_profile_workload_from_key_or_command(tmp_path):
    new = {"workload_key": bench.workload_key(bench.parse(["--width", "3840", "--height", "2160"]))}
    assert bench.profile_workload(new)["width"] == 3840
    old = {"bench_command": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pixel-anchor --rng-mode pixel"}
    k = bench.profile_workload(old)
    assert k["rng_mode"] == "pixel" and k["width"] == 1920 and k["ranks"] == 1
    assert bench.profile_workload({"bench_command": "python3 something_else.py"}) is None
    assert bench.profile_workload({}) is None


def test_committed_4k_line_does_not_quote_the_1080p_profile():
    import glob
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_bench_line_4k.json")))
    for p in lines:
        d = json.load(open(p))
        assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"] is None and "secondary" not in d["roofline"], p


def test_cpu_share_reads_quota_and_affinity(orc):
    s = orc.cpu_share()
    assert 1 <= s["granted"] <= s["affinity"] <= s["host_cpus"]
    if s["quota_cores"] is not None:
        assert s["granted"] <= int(s["quota_cores"] + 0.999)
    orc.set_num_threads(2)
    assert orc.lib().orc_num_threads() == 2
    orc.set_num_threads(min(s["granted"], 16))
