"""bench.py's own process launcher (`python bench.py --gpus N` without torch.distributed.run), proven on the CPU:
`--dry-run-launcher` makes the ranks rendezvous over gloo and all-reduce one number instead of running the GPU step,
so what is tested is exactly the launcher: child processes (no exec), RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
rank 0's JSON line relayed, a failing rank failing the whole run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_self_launch_two_ranks_relays_rank0_line():
    r = _run(["--gpus", "2", "--steps", "7", "--warmup", "2", "--dry-run-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]        # (gloo prints its own connection notes)
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    col = d.pop("collective")
    assert d == {"dry_run": True, "n_gpus": 2, "local_rank": 0, "rank_sum": 3.0, "steps": 7, "warmup": 2,
                 "batch_per_gpu": 2048, "global_batch": 4096, "config3_leg": None}
    # the collective facts a multi-GPU line carries: backend, world size as torch.distributed sees it, one entry per rank
    assert col["backend"] == "gloo" and col["world_size"] == 2 and col["rccl_version"] is None
    assert [(r["rank"], r["local_rank"]) for r in col["ranks"]] == [(0, 0), (1, 1)]


def test_global_batch_is_split_over_the_ranks_and_config3_runs_at_eight():
    """`--global-batch G`: G / N windows per GPU (strong scaling); BASELINE config 3 (8192 = 8 x 1024) is a second leg of the
    default --gpus 8 run, next to the 2048-per-GPU weak-scaling headline"""
    r = _run(["--gpus", "2", "--global-batch", "8192", "--dry-run-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert (d["batch_per_gpu"], d["global_batch"], d["config3_leg"]) == (4096, 8192, None)
    r = _run(["--gpus", "2", "--global-batch", "8191", "--dry-run-launcher"])
    assert r.returncode != 0
    r = _run(["--gpus", "8", "--dry-run-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert (d["n_gpus"], d["batch_per_gpu"], d["global_batch"], d["config3_leg"]) == (8, 2048, 16384, 1024)
    assert d["collective"]["world_size"] == 8 and len(d["collective"]["ranks"]) == 8


def test_self_launch_fails_when_a_rank_fails():
    r = _run(["--gpus", "2", "--dry-run-launcher", "--test-fail-rank", "1"])
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_self_launch_ends_the_other_ranks_when_one_fails():
    """rank 1 fails while rank 0 would run for an hour (a deadlocked collective looks like this): the launcher must
    terminate rank 0 and return promptly instead of hanging in wait()"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--dry-run-launcher", "--test-fail-rank", "1", "--test-hang-rank", "0"])
    assert r.returncode != 0 and "ranks failed" in r.stderr
    assert time.time() - t0 < 120


def test_no_launch_when_the_ranks_were_started_for_us():
    """WORLD_SIZE in the environment (torch.distributed.run started us): bench.py must not spawn anything."""
    r = _run(["--gpus", "1", "--dry-run-launcher"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
