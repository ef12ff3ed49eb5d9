"""`python bench.py --gpus N` must start N ranks itself (it used to ignore N).  Driven here on CPU: BENCH_FAKE=1
swaps the HIP sampler for a per-cell stand-in and RCCL for gloo; the launcher, the rank/WORLD_SIZE handling, the barrier +
max-over-ranks timing, the all-gather and the JSON line are the code that runs on the GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_launches_two_ranks_and_gathers():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64"], {"BENCH_FAKE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2
    assert j["config"]["cells_per_gpu"] == 64 and j["config"]["global_cells"] == 128
    assert j["config"]["gathered_rows"] == 2 * 2 * 64          # both ranks' (2B) rows arrived in the all-gather
    assert j["scaling"] == "weak" and j["steps"] == 2 and j["value"] > 0
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1   # rank 0 only prints
    # self-validation of the sharded run: rank 0 recomputed 8 cells of rank 1's shard and compared bit for bit
    chk = j["cross_rank_check"]
    assert chk["checked_rank"] == 1 and chk["cells"] == 8 and chk["bit_equal"] is True
    assert len(j["per_rank_ms_per_step"]) == 2 and j["allgather_ms"] >= 0


def test_cross_rank_check_catches_a_rank_that_ran_something_else():
    """BENCH_FAKE_CORRUPT_RANK=1: rank 1's stand-in sampler is perturbed - the self-check must fail the job loudly."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "16"], {"BENCH_FAKE": "1", "BENCH_FAKE_CORRUPT_RANK": "1"})
    assert r.returncode != 0
    assert "cross-rank self-check FAILED" in (r.stderr + r.stdout)


def test_strong_scaling_leg_splits_the_global_batch():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"BENCH_FAKE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    s = j["strong_scaling"]
    assert s["global_cells"] == 8192 and s["cells_per_gpu"] == 4096 and s["scaling"] == "strong"


def test_single_rank_default_is_one_gpu():
    r = _run(["--steps", "1", "--warmup", "0", "--batch", "8", "--no-extra", "--no-cpu-baseline"], {"BENCH_FAKE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["config"]["gathered_rows"] == 16


def test_more_gpus_than_devices_fails_loudly():
    r = _run(["--gpus", "64"], {})
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4"], {"BENCH_FAKE": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)
