"""bench.py's N > 1 path on a one-GPU box: two ranks under torch.distributed.run share the GPU and exchange their
slabs over gloo (`--dist-backend gloo`, a rehearsal mode; the driver's multi-GPU runs use RCCL with one GPU per
rank). The gathered frame must be byte-identical to the single-rank frame in the pixel-keyed RNG mode, and rank 0 must
print exactly one JSON line with the contract's fields."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

COMMON = ["--steps", "2", "--warmup", "1", "--spp", "6", "--width", "400", "--height", "225", "--bounces", "5", "--no-stage-times", "--no-cpu-baseline"]


def last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_two_rank_rehearsal_equals_single_rank(gpu, tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    one, two = tmp_path / "n1.ppm", tmp_path / "n2.ppm"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, "--rng-mode", "pixel", "--dump", str(one)],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON, "--dist-backend", "gloo",
                         "--dump", str(two)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r2.returncode == 0, r2.stderr[-3000:]
    a, b = last_json_line(r1.stdout), last_json_line(r2.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["config"]["rng_mode"] == "pixel" and b["scaling"] == "strong"
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in b
    assert a["config"]["rays_traced"] == b["config"]["rays_traced"]  # the same rays, split over two ranks
    assert a["config"]["gather_check"] is None and b["config"]["gather_check"].startswith("every rank's slab equals")  # bench's own check of the gather
    assert one.read_bytes() == two.read_bytes()
    # where an N > 1 figure comes from: every rank's own time, the slab's render time, the gather timed alone
    assert "gather_ms" not in a and "rank_ms_per_step" not in a
    assert len(b["rank_ms_per_step"]) == 2 and len(b["rank_render_ms_per_step"]) == 2 and len(b["gather_ms_per_rank"]) == 2
    assert all(x > 0 for x in b["rank_ms_per_step"] + b["rank_render_ms_per_step"]) and b["gather_ms"] == max(b["gather_ms_per_rank"]) > 0
    assert "REHEARSAL" in b["gather_ms_note"]
    assert b["value_per_rank_ceiling"] == pytest.approx(b["config"]["rays_traced"] / b["steps"] / (max(b["rank_render_ms_per_step"]) * 1e-3) / 1e6, rel=1e-3)


def test_rccl_branch_of_bench_with_a_one_rank_communicator(gpu):
    """bench.py's N > 1 data path -- wfpt_comm_unique_id -> wfpt_comm_init -> wfpt_gather_accumulated per frame ->
    wfpt_read_gathered -- executed at world 1 (`--force-rccl`), so that it has run before an 8-GPU box sees it. bench.py itself
    compares the gathered frame with the accumulated image and fails on a difference."""
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, "--rng-mode", "pixel", "--force-rccl"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json_line(r.stdout)
    assert line["n_gpus"] == 1 and "RCCL" in line["config"]["gather"]
    assert line["config"]["samples_in_flight"] == [6] and "6 spp" in line["config"]["workload"]
    # the gather timed alone by a hipEvent pair behind the C ABI (wfpt_gather_accumulated_timed), the rank's own time, the ceiling
    assert line["gather_ms"] > 0 and line["gather_ms_per_rank"] == [line["gather_ms"]] and "hipEvent" in line["gather_ms_note"]
    assert len(line["rank_ms_per_step"]) == 1 and len(line["rank_render_ms_per_step"]) == 1 and line["value_per_rank_ceiling"] > 0
    assert line["config"]["loop_kind"] == "fused"  # 400x225 pixel-keyed: below the binned loop's size gate, and the line says what really ran


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (torch.distributed.run as a child). On a box with
    two GPUs the line comes back; on a one-GPU box the run must fail at RCCL's own refusal of two ranks on one device -- with
    a non-zero exit code and that reason on stderr -- not at an argument check."""
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=600)
    if gpu.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-3000:]
        line = last_json_line(r.stdout)
        assert line["n_gpus"] == 2 and "RCCL" in line["config"]["gather"]
        assert line["config"]["gather_check"].startswith("every rank's slab equals")
    else:
        assert r.returncode != 0
        assert "RCCL communicator could not be built" in r.stderr, r.stderr[-3000:]
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no number from a run that did not shard over GPUs


def test_two_gpus_over_rccl_equal_one_gpu(gpu, tmp_path):
    """Where the box has two GPUs: the frame of two ranks, one GPU each, gathered by ncclSend / ncclRecv behind the C ABI, is the
    single-GPU frame of the pixel-keyed RNG mode byte for byte (the legs a one-GPU box cannot run; skipped there)."""
    if gpu.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one, two = tmp_path / "n1.ppm", tmp_path / "n2.ppm"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, "--rng-mode", "pixel", "--dump", str(one)],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON, "--dump", str(two)],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    b = last_json_line(r2.stdout)
    assert b["n_gpus"] == 2 and "RCCL" in b["config"]["gather"] and b["config"]["rng_mode"] == "pixel"
    assert one.read_bytes() == two.read_bytes()
