"""bench.py's own N > 1 launcher on a CPU-only box: `python bench.py --gpus 2` (no torchrun environment) must start two
rank processes, rendezvous on 127.0.0.1 (gloo here, RCCL on GPUs), exchange the far-end packet every step and print ONE
JSON line with n_gpus = 2 -- VERDICT r01 item 2.  The stub workload has no compute; the plumbing is what is tested."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_starts_that_many_ranks():
    r = _run(["--gpus", "2", "--workload", "stub_cpu", "--steps", "5", "--warmup", "2", "--prime", "3", "--no-cpu"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2
    assert d["dist_backend"] == "gloo" and d["launched_by"] == "bench.py"
    assert len(d["per_rank_ms_per_step"]) == 2
    # rank 0's far-end value reached every step: 1 + 2 + ... + (3 + 2 + 5 + 5 extra breakdown steps)
    assert d["config"]["far_sum"] == sum(range(1, 16))
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_gpus_flag_must_match_world_size():
    r = _run(["--gpus", "4", "--workload", "stub_cpu", "--steps", "1", "--warmup", "0", "--prime", "0", "--no-cpu"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_torchrun_launch_still_works():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--workload", "stub_cpu", "--steps", "3", "--warmup", "1", "--prime", "0", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["launched_by"] == "torchrun"


def test_a_dead_rank_ends_the_launch_with_its_code():
    """Round-2 ADVICE: a rank that dies must not leave the parent waiting for the survivors' collective timeout."""
    import time
    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--workload", "stub_cpu", "--steps", "5", "--warmup", "2", "--prime", "3", "--no-cpu"],
             {"WMIX_STUB_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert "ranks failed" in r.stderr and "(1, 7)" in r.stderr
    assert time.monotonic() - t0 < 90
