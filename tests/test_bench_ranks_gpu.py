"""bench.py under the driver's launcher with two ranks.  One GPU is enough for the logic: --all-ranks-on-device puts both ranks on
device 0 and gloo carries the barrier / the reductions, so the rank-sharded problem ids, the max-over-ranks timing, the summed
counters and the distinct-device count of the JSON line are exercised exactly as with one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_device_report_the_aggregate_and_one_device():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256", "--dist-backend", "gloo",
           "--all-ranks-on-device", "0", "--no-cpu-baseline", "--no-extra-legs"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["config"]["ranks"] == 2 and d["config"]["problems_per_gpu"] == 256
    assert d["n_gpus"] == 1                                 # distinct devices behind the ranks, not the rank count
    assert d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    # the aggregate: both ranks' problems over the slower rank's wall clock
    frames = d["frames_per_s"] * d["ms_per_step"] * 1e-3
    assert abs(frames - 2 * 256) < 1e-6 * 512
    assert abs(d["value"] - d["iters_per_frame"] * d["frames_per_s"]) < 1e-6 * d["value"]
    assert 7.0 < d["iters_per_frame"] < 11.0


@pytest.mark.gpu
def test_asking_for_more_gpus_than_the_node_has_fails_loudly():
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--batch", "64"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


@pytest.mark.gpu
def test_dry_ranks_rehearses_the_n_rank_launch_on_the_devices_that_exist():
    """`--dry-ranks N`: the script spawns N ranks itself, rank r on device r mod #GPUs, RCCL when every rank has its own GPU and gloo
    otherwise; the line carries the ranks, the distinct devices, the per-rank step times and says that it was a rehearsal."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", "2", "--steps", "1", "--warmup", "1", "--batch", "128",
                        "--no-cpu-baseline", "--no-extra-legs"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    import torch
    ndev = torch.cuda.device_count()
    assert d["config"]["ranks"] == 2 and d["config"]["dry_ranks"] is True
    assert d["n_gpus"] == min(2, ndev)
    assert d["config"]["ranks_per_device"] == -(-2 // ndev)
    assert d["config"]["dist_backend"] == ("nccl" if ndev >= 2 else "gloo")
    assert len(d["rank_ms_per_step"]) == 2 and all(v > 0 for v in d["rank_ms_per_step"])
    assert abs(max(d["rank_ms_per_step"]) - d["ms_per_step"]) < 0.25 * d["ms_per_step"]


@pytest.mark.gpu
def test_dry_ranks_8_is_the_eight_rank_launch_on_the_devices_that_exist():
    """The launch the driver makes on an 8-GPU node, rehearsed: eight ranks, 64 problems each, one step.  One JSON line with eight per-rank
    step times; on a one-GPU box all ranks share device 0 (gloo carries the barrier), n_gpus stays the number of distinct devices."""
    import torch
    ndev = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-ranks", "8", "--steps", "1", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline", "--no-extra-legs"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["ranks"] == 8 and d["config"]["dry_ranks"] is True and d["config"]["problems_per_gpu"] == 64
    assert d["config"]["parallelism"] == "8 x independent problems (no collective)"
    assert d["n_gpus"] == min(8, ndev)
    assert len(d["rank_ms_per_step"]) == 8 and all(v > 0 for v in d["rank_ms_per_step"])
    frames = d["frames_per_s"] * d["ms_per_step"] * 1e-3
    assert abs(frames - 8 * 64) < 1e-6 * 512


@pytest.mark.gpu
def test_dry_ranks_8_of_the_stress_config_two_problems_per_rank():
    """BASELINE configs[4] as the driver would launch it on an 8-GPU node: 16 C5 problems (2000 nodes, 4000 matches) round-robin on eight ranks =
    two per rank, i.e. the latency shape (speculative lanes, two-sided factorisation with helper workgroups) on every rank.  One JSON line, eight
    per-rank step times, the optional collective leg off."""
    import torch
    ndev = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C5", "--dry-ranks", "8", "--steps", "1", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline", "--no-extra-legs"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["ranks"] == 8 and d["config"]["dry_ranks"] is True and d["config"]["problems_per_gpu"] == 2
    assert d["config"]["workload"].startswith("C5")
    assert d["n_gpus"] == min(8, ndev)
    assert len(d["rank_ms_per_step"]) == 8 and all(v > 0 for v in d["rank_ms_per_step"])
    assert "shared_camera" not in d
    frames = d["frames_per_s"] * d["ms_per_step"] * 1e-3
    assert abs(frames - 16) < 1e-6 * 16
    assert 8.0 < d["iters_per_frame"] < 16.0


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_plan_under_the_drivers_launcher_one_gpu_per_rank():
    """`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`: rank r computes on device LOCAL_RANK, RCCL carries the barrier, the
    line says "8 x independent problems"; a WORLD_SIZE that contradicts --gpus or a device that does not exist is refused."""
    b = _bench_module()
    for lr in range(8):
        env = {"RANK": str(lr), "LOCAL_RANK": str(lr), "WORLD_SIZE": "8"}
        p = b.rank_plan(8, 0, -1, "nccl", env, 8)
        assert p["device"] == lr and p["rank"] == lr and p["world"] == 8
        assert p["backend"] == "nccl" and p["ranks_per_device"] == 1
        assert p["parallelism"] == "8 x independent problems (no collective)"
    assert "error" in b.rank_plan(8, 0, -1, "nccl", {"RANK": "5", "LOCAL_RANK": "5", "WORLD_SIZE": "8"}, 4)      # device 5 of 4
    assert "error" in b.rank_plan(4, 0, -1, "nccl", {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "8"}, 8)      # --gpus 4, eight ranks
    # rehearsal on one device: every rank on device 0, gloo instead of RCCL (duplicate GPU in one communicator)
    p = b.rank_plan(8, 8, -1, "nccl", {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}, 1)
    assert p["device"] == 0 and p["ranks_per_device"] == 8 and p["backend"] == "gloo"
    # rehearsal on a full node: the real backend
    p = b.rank_plan(8, 8, -1, "nccl", {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}, 8)
    assert p["device"] == 3 and p["ranks_per_device"] == 1 and p["backend"] == "nccl"
    # single process
    p = b.rank_plan(1, 0, -1, "nccl", {}, 1)
    assert p["device"] == 0 and p["world"] == 1


def test_rank_plan_keeps_the_shared_camera_leg_off_with_several_ranks():
    """The optional joint-problem leg (a collective of the library's own RCCL communicator) runs by default with ONE rank only: the first
    multi-GPU launch must not depend on a collective that has never run on that node.  `--shared-camera on` asks for it explicitly; without
    RCCL (the gloo rehearsal) it cannot run at all."""
    b = _bench_module()
    assert b.rank_plan(1, 0, -1, "nccl", {}, 1)["shared_camera"] is True
    for n in (2, 4, 8):
        env = {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": str(n)}
        assert b.rank_plan(n, 0, -1, "nccl", env, 8)["shared_camera"] is False                 # auto
        assert b.rank_plan(n, 0, -1, "nccl", env, 8, "off")["shared_camera"] is False
        assert b.rank_plan(n, 0, -1, "nccl", env, 8, "on")["shared_camera"] is True            # asked for
        assert b.rank_plan(n, n, -1, "nccl", env, 1, "on")["shared_camera"] is False           # rehearsal on one device: gloo
    assert b.rank_plan(1, 0, -1, "nccl", {}, 1, "off")["shared_camera"] is False
