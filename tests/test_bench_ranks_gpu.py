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
