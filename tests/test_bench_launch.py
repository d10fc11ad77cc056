"""bench.py's own launcher: `python bench.py --gpus N` started plainly must run N ranks — the parent touches no
GPU and starts `python -m torch.distributed.run` on itself as a fresh child — and a WORLD_SIZE that disagrees with --gpus is an error.
Here (no GPU) `--rehearse` runs launcher + gloo rendezvous + the gather with the real shapes and no kernel; the GPU test runs the whole
N = 2 step (kernels on cuda:0 from both ranks, gloo collective), the configs[3] pass and the one-process leg on a one-GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _json_line(stdout):
    rows = [json.loads(l) for l in stdout.split("\n") if l.startswith("{")]
    assert len(rows) == 1, stdout
    return rows[0]


def test_plain_launch_starts_two_ranks_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--windows", "12", "--haps", "3", "--reads", "7"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    row = _json_line(r.stdout)
    assert row["n_gpus"] == 2 and row["ranks"] == 2 and row["backend"] == "gloo"
    assert row["value"] is None and "launch-only" in row["rehearsal"]


def test_plain_launch_starts_eight_ranks_cpu():
    """The rank count the driver's scaling run uses: launcher, rendezvous and the gather layout with 8 ranks (launch-only: no GPU here)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("launch-only rehearsal: needs a host without GPU")
    env = _env()
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse", "--windows", "6", "--haps", "2", "--reads", "5"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    row = _json_line(r.stdout)
    assert row["n_gpus"] == 8 and row["ranks"] == 8 and row["backend"] == "gloo" and row["value"] is None


def test_world_size_must_equal_gpus():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in r.stderr
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 8" in r.stderr


@pytest.mark.gpu
def test_plain_launch_two_ranks_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--windows", "300", "--steps", "2", "--warmup", "1",
                        "--configs3-windows", "1500", "--max-batch-windows", "400"],
                       capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    row = _json_line(r.stdout)
    assert row["n_gpus"] == 2 and row["ranks"] == 2 and row["scaling"] == "weak"
    assert row["value"] > 1e9 and row["config"]["windows_per_gpu"] == 300
    c3 = row["configs3"]
    assert c3["scaling"] == "strong" and c3["windows_per_gpu"] == 750 and c3["sub_batches_per_gpu"] == [[375, 2]] and c3["cells_per_s"] > 1e9
    ip = row["in_process"]
    assert "error" not in ip, ip
    assert ip["devices"] == [0, 0] and ip["equals_resident_launch"] is True and ip["cells_per_s"] > 1e9
