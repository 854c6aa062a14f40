"""bench.py's own launcher (`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the ranks itself): it
will first run for real on the driver's 8-GPU node, where nobody can debug it, so its control flow is exercised here --
on the GPU box with two gloo ranks on the one card (the rehearsal knobs), and its refusal to print an n_gpus=N line on a
box with fewer devices.  The parent must count devices without touching the GPU (sysfs, bench.visible_gpu_count)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FTR_BENCH_BACKEND", "FTR_BENCH_FORCE_DEVICE")}
    env.update(extra)
    return env


def test_visible_gpu_count_reads_sysfs_and_honours_the_visibility_variables(monkeypatch):
    import bench
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    n = bench.visible_gpu_count()
    assert isinstance(n, int) and n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count() == min(n, 1)


@pytest.mark.gpu
def test_visible_gpu_count_matches_the_runtime(dev):
    import torch
    import bench
    assert bench.visible_gpu_count() == torch.cuda.device_count()


@pytest.mark.gpu
def test_self_launch_two_ranks_prints_one_line(dev):
    """Two ranks on one card (gloo): one JSON line from rank 0 with n_gpus = 2, twice the per-rank batch, a roofline object."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-graph",
                          "--no-gemm-tuning"], env=_clean_env(FTR_BENCH_BACKEND="gloo", FTR_BENCH_FORCE_DEVICE="0"),
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 64 and d["scaling"] == "weak"
    assert d["roofline"] and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["value"] > 0 and d["cpu_baseline"] is None


@pytest.mark.gpu
def test_self_launch_refuses_more_gpus_than_the_box_has(dev):
    import torch
    have = torch.cuda.device_count()
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 2, (out.returncode, out.stderr[-1000:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_external_launcher_with_the_wrong_world_size_is_refused(dev):
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
