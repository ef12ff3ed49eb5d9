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


def _strong(extra):
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "16", *extra], {"BENCH_FAKE": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and len(j["per_rank_ms_per_step"]) == 8 and j["allgather_ms"] >= 0
    assert j["config"]["gathered_rows"] == 8 * 2 * 16 and j["cross_rank_check"]["bit_equal"] is True
    assert len(json.dumps(j)) <= 6144          # the driver keeps an 8 KB tail of stdout: the line must fit with room to spare
    return j["strong_scaling"]


def test_gpus_8_strong_scaling_splits_8192_cells_evenly():
    """The first real 8-GPU run (configs[3]: 8 192 cells over 8 ranks) must not die on plumbing: eight gloo ranks of the fake sampler."""
    s = _strong([])
    assert s["global_cells"] == 8192 and s["cells_per_gpu_by_rank"] == [1024] * 8 and s["gathered_rows"] == 8 * 2 * 1024
    assert s["cross_rank_check"]["checked_rank"] == 7 and s["cross_rank_check"]["bit_equal"] is True


def test_gpus_8_strong_scaling_pads_uneven_shards():
    """8 190 cells do not divide by 8: six ranks take 1 024 cells, two take 1 023, every rank's [unconditional | guided] block is
    padded to 1 024 for the all-gather, and rank 0 verifies the SHORT last shard bit for bit through the padded layout."""
    s = _strong(["--global-cells", "8190"])
    assert s["global_cells"] == 8190 and s["cells_per_gpu_by_rank"] == [1024] * 6 + [1023] * 2
    assert s["gathered_rows"] == 8 * 2 * 1024
    assert s["cross_rank_check"]["checked_rank"] == 7 and s["cross_rank_check"]["cells"] == 8 and s["cross_rank_check"]["bit_equal"] is True


def test_rank_binds_its_device_before_anything_creates_a_handle(monkeypatch):
    """One process per GPU: rank r must call torch.cuda.set_device(LOCAL_RANK) BEFORE a model (native handle, streams, workspaces)
    exists - a handle created on device 0 by every rank is the classic first-8-GPU-run failure.  torch.cuda is mocked: this runs on CPU."""
    import importlib
    import torch
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    order = []

    class Stop(Exception):
        pass

    def fake_make_model(*a, **k):
        order.append(("make_model", None))
        raise Stop()
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: order.append(("set_device", d)))
    monkeypatch.setattr(bench, "make_model", fake_make_model)
    for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_FAKE", "BENCH_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline"])
    try:
        bench.main()
    except Stop:
        pass
    assert order[0] == ("set_device", 5) and order[1] == ("make_model", None), order
    # and a LOCAL_RANK beyond the visible devices is refused before any device call
    monkeypatch.setenv("LOCAL_RANK", "9")
    order.clear()
    try:
        bench.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert "LOCAL_RANK 9" in str(e) and not order
