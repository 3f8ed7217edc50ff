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
                 "batch_per_gpu": 2048, "global_batch": 4096, "config3_leg": None, "unet_dp": None}
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


def test_unet_line_names_both_data_parallel_modes():
    """`--config unet --gpus N`: per-rank BatchNorm statistics (1 collective per step) are the throughput headline, the exact
    mode (21 collectives) is timed next to it, and the measured deviation of the headline mode is quoted in the line"""
    r = _run(["--gpus", "2", "--config", "unet", "--dry-run-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["unet_dp"]
    assert d["headline_mode"] == "per_rank_bn"
    assert (d["per_rank_bn"]["collectives_per_step"], d["sync_bn"]["collectives_per_step"]) == (1, 21)
    dev = d["per_rank_bn"]["deviation_from_global_batch_step"]
    assert dev["loss_rel"] == 3.5e-5 and dev["grad_rel_of_norm"] == 3.6e-2 and "test_dp_gloo" in dev["source"]
    # ... and the test it quotes exists
    src = open(os.path.join(ROOT, "tests", "test_dp_gloo.py")).read()
    assert "def " + dev["source"].split("::")[1] in src


def test_roofline_object_carries_the_pipe_and_busy_fractions():
    """the roofline object says which pipe the kernel's contractions run on and what the counters measured (mfma_busy /
    valu_issue of the newest committed counter run), the whole step's useful-FLOP fraction and the attention block's
    (forward + backward) - next to `frac`, which stays useful FLOP over the fp32 peak"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.roofline_object("attn_bwd", "ralenet", 512, 2048, kind_ms=3.357 * 5, launches=90, rl_steps=5,
                              attn_ms={"attn_fwd": 1.746, "attn_bwd": 3.357}, n_timed=160, dt=160 * 12.865e-3)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "pipe", "mfma_busy", "valu_issue", "counters_source",
              "step_frac", "attn_block_frac", "attn_block"):
        assert k in r, k
    assert abs(r["frac"] - 0.374) < 2e-3 and abs(r["step_frac"] - 0.447) < 2e-3       # round-5 figures reproduce
    assert "f16 mfma" in r["pipe"] and 0 < r["mfma_busy"] < 1 and 0 < r["valu_issue"] <= 1
    assert r["counters_source"].startswith("profiles/r") and r["traffic"] > 1e8
    blk = r["attn_block"]
    want = (bench.kind_work("attn_fwd", 512, 2048) + bench.kind_work("attn_bwd", 512, 2048)) / ((1.746 + 3.357) * 1e-3) / 1e12 / 157.3
    assert abs(r["attn_block_frac"] - want) < 1e-3 and blk["fwd_frac"] < blk["bwd_frac"]
