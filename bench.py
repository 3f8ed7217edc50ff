"""Headline benchmark: RA-LENet training-step throughput (ECG windows/s) on MI355X.

    python bench.py [--gpus N --steps K --warmup W]

N > 1 runs one process per GPU over RCCL: either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or
bench.py starts them itself when WORLD_SIZE is not set (child processes, never exec; rank 0's JSON line is relayed).

One "step" = zero_grad -> forward -> mse/SNR/RMSE -> backward -> (all-reduce) -> Adam on one batch of
synthetic 512-sample windows already resident in HBM.  Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CH = [8, 16, 32, 64, 128]
BLOCKS_PER_LEVEL = [2, 4, 4, 4, 4]
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA peak (the dtype every contraction here runs in)
VALU_F32_PEAK_TF = 157.3


def kind_work(kind, L, B):
    """Algorithmic FLOPs of all launches of one kernel kind in a train step."""
    fl = 0.0
    for lvl, nb in enumerate(BLOCKS_PER_LEVEL):
        N, Cc = L >> lvl, CH[lvl]
        if kind == "attn_fwd":
            fl += nb * 4.0 * N * N * Cc * B                 # QK^T + PV, mul+add
        elif kind == "attn_bwd":
            fl += nb * 2.5 * 4.0 * N * N * Cc * B           # S, dP, dV, dK, dQ: 5 products (S, dP counted once)
        elif kind == "mlp_fwd":
            fl += nb * 2.0 * N * Cc * Cc * 9 * B            # proj C*C + fc1 4C*C + fc2 4C*C
        elif kind == "mlp_bwd":
            fl += nb * 2.0 * N * Cc * Cc * 9 * B
        elif kind == "qkv_fwd" or kind == "qkv_bwd":
            fl += nb * 2.0 * N * Cc * Cc * 3 * B
        elif kind == "dw":
            fl += nb * 2.0 * N * Cc * Cc * 12 * B
    return fl


def measured_traffic(kind):
    """HBM bytes per launch of this kernel kind from the committed PMC run (profiles/r02_hbm_traffic.json)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")))
        return d["per_launch_bytes"][kind]["total"]
    except Exception:
        return None


def conv_stage_roofline(dev, L, B=2048):
    """The HBM-bound conv stages of the path (north star: fraction of the HBM roofline on the U-Net stages at batch
    2048 x 512): U-Net eval forward, 2 leads, at the STATED batch.  Byte convention = SURVEY 8d's stage-granular count:
    every conv reads its input tensor and writes its output tensor once (11 x 2), the three decoder skips and the
    bottleneck residual are re-read (+ 4): 26 tensors of leads * L floats per window.  Two figures:
      `achieved` / `frac`: the product path - the whole forward fused into ONE kernel (all stage tensors stay in LDS), timed
        as hipGraph replays; its stage-granular-equivalent rate, with the HBM bytes it really moves next to it;
      `staged`: the stage-by-stage path (11 conv launches + the output BatchNorm pass: what training runs) at the same
        batch, 27 tensors with that extra pass, where every stage tensor does make the HBM round trip."""
    import torch
    from ecg_denoise_amd import UNet, _lib
    leads = 2
    m = UNet(leads=leads, L=L, max_batch=B, train=False, device=dev, seed=1)
    m.eval()
    x = torch.randn(B, leads, L, device=dev)
    w = leads * L * 4                       # bytes of one stage tensor of one window

    def graph_time(n=200):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                y = m(x)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = m(x)
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n
    out = {"bound": "hbm", "batch": B, "unit": "GB/s", "peak": HBM_PEAK_GBS}
    _lib.check(_lib.lib().ral_set_option(m.eng.h, b"unet_fused", 1))
    dt = graph_time()
    out.update({"kernel": "U-Net eval forward fused into one kernel (k_unet_pack + k_unet_infer), hipGraph replay",
                "achieved": round(B * 26 * w / dt / 1e9, 1), "frac": round(B * 26 * w / dt / 1e9 / HBM_PEAK_GBS, 4),
                "tensors_per_window": 26, "us_per_forward": round(dt * 1e6, 1), "windows_per_s": round(B / dt, 1),
                "hbm_bytes_moved_per_window": 2 * w, "hbm_GBps_moved": round(B * 2 * w / dt / 1e9, 1)})
    _lib.check(_lib.lib().ral_set_option(m.eng.h, b"unet_fused", 0))
    dt = graph_time()
    out["staged"] = {"kernel": "stage by stage: 11 conv launches + output BatchNorm pass, hipGraph replay",
                     "achieved": round(B * 27 * w / dt / 1e9, 1), "frac": round(B * 27 * w / dt / 1e9 / HBM_PEAK_GBS, 4),
                     "tensors_per_window": 27, "us_per_forward": round(dt * 1e6, 1), "windows_per_s": round(B / dt, 1)}
    del m, x
    torch.cuda.empty_cache()
    return out


def cpu_baseline(leads, L, variant):
    """The oracle (CPU restatement of the reference op graph, parity-pinned by tests/golden) timed on the
    host cores of this box on a bounded sample: batch 32 (the reference's own batch, BASELINE config 0)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from collections import OrderedDict
    import torch
    import ralenet_oracle as O
    # intra-op threads: the count with the highest measured throughput on the GPU box's host (256 logical cores;
    # tools/diag/cpu_threads.py: 1 thread 40, 2: 60, 4: 74, 8: 84, 16: 78, 32: 56, 64: 26 windows/s) - the op graph is
    # ~21k small ATen calls per step, and past 8 threads the fork/join cost of each call exceeds its work
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    B = 32
    p = O.init_params(O.ralenet_param_shapes(variant, leads), 1)
    g = torch.Generator().manual_seed(2023)
    x = torch.randn(B, leads, L, generator=g); tgt = torch.randn(B, leads, L, generator=g)
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
    v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    bn = O.new_bn_state()
    fwd = lambda pp, xx: O.ralenet_forward(pp, xx, variant, True, bn)
    O.train_step(p, x, tgt, fwd, m, v, 1)
    n, t0 = 0, time.time()
    while time.time() - t0 < 10.0 and n < 40:
        O.train_step(p, x, tgt, fwd, m, v, n + 2)
        n += 1
    dt = time.time() - t0
    return {"value": round(B * n / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of the CPU oracle at batch {B} x {leads} x {L} fp32 "
                      f"(torch-CPU op graph of the reference, {cores} threads)"}


def self_launch(n, argv):
    """Start the n ranks of a single-node job as CHILD processes of this one (which has not touched the GPU and never
    will), wait for all of them, relay rank 0's stdout.  Returns the exit code: non-zero if any rank failed."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def launcher_dry_run(a):
    """`--dry-run-launcher`: the ranks rendezvous over gloo on the CPU, all-reduce one number and rank 0 prints a line
    marked "dry_run".  It exercises the process launcher and the environment contract only (tests/test_bench_cpu.py);
    nothing of the measured path runs and the line is not a benchmark result."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if os.environ.get("RAL_BENCH_FAIL_RANK") == str(rank):      # (test hook: a failing rank must fail the launcher)
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "rank_sum": t.item(), "steps": a.steps, "warmup": a.warmup}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2048, help="windows per GPU")
    ap.add_argument("--leads", type=int, default=1)
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--variant", default="full")
    ap.add_argument("--kind", default="attn_bwd", help="kernel kind timed for the roofline object")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-infer", action="store_true", help="skip the inference-forward leg")
    ap.add_argument("--kinds", action="store_true", help="print a per-kernel-kind time table to stderr (3 steps each)")
    ap.add_argument("--dry-run-launcher", action="store_true", help="CPU/gloo rendezvous only: tests the process launcher")
    a = ap.parse_args()

    # N > 1 and nobody started the ranks for us: start them (before torch is imported or the GPU touched)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    if a.dry_run_launcher:
        return launcher_dry_run(a)

    import torch
    import torch.distributed as dist
    from ecg_denoise_amd import RALENet, _lib
    from ecg_denoise_amd.dp import DataParallelTrainer, HipEngineAdapter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs WORLD_SIZE={a.gpus} (launch with torch.distributed.run)")
    if os.environ.get("RAL_BENCH_SHARE_GPU"):    # test hook: every rank on device 0 (tests/test_gpu_dp_procs.py)
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RAL_BENCH_BACKEND", "nccl")   # "gloo": the same hook (RCCL needs one device per rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    B = a.batch
    model = RALENet(a.variant, leads=a.leads, L=a.L, max_batch=B, train=True, device=dev, seed=2023)
    g = torch.Generator().manual_seed(2023 + rank)
    x = torch.randn(B, a.leads, a.L, generator=g).to(dev)
    tgt = torch.randn(B, a.leads, a.L, generator=g).to(dev)
    trainer = DataParallelTrainer(HipEngineAdapter(model))
    model.train()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.train_step(x, tgt)
    sync()
    lib = _lib.lib()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = trainer.train_step(x, tgt)
    sync()
    dt = time.perf_counter() - t0
    # roofline leg: the same step with the kernels serialised (one lane, no side stream), so that the hipEvent
    # pair around each launch of the selected kernel measures that kernel alone, not its share of a busy GPU
    rl_steps = max(3, a.steps // 4)
    ms, cnt = C.c_double(), C.c_int64()
    if rank == 0 or world > 1:
        _lib.check(lib.ral_set_option(model.eng.h, b"lanes", 1))
        _lib.check(lib.ral_set_option(model.eng.h, b"side_stream", 0))
        trainer.train_step(x, tgt)
        sync()
        _lib.check(lib.ral_profile_select(model.eng.h, a.kind.encode()))
        for _ in range(rl_steps):
            trainer.train_step(x, tgt)
        sync()
        _lib.check(lib.ral_profile_read(model.eng.h, C.byref(ms), C.byref(cnt)))
        _lib.check(lib.ral_profile_select(model.eng.h, b""))
        if a.kinds and rank == 0:
            tot = 0.0
            for kind in ("qkv_fwd", "attn_fwd", "mlp_fwd", "resample_fwd", "mlp_bwd", "attn_bwd", "qkv_bwd", "dw", "resample_bwd"):
                _lib.check(lib.ral_profile_select(model.eng.h, kind.encode()))
                for _ in range(3):
                    trainer.train_step(x, tgt)
                sync()
                ms2, cnt2 = C.c_double(), C.c_int64()
                _lib.check(lib.ral_profile_read(model.eng.h, C.byref(ms2), C.byref(cnt2)))
                w = kind_work(kind, a.L, B)
                tot += ms2.value / 3
                print(f"  {kind:14s} {ms2.value/3:8.3f} ms/step  {cnt2.value//3:4d} launches  "
                      f"{(w / (ms2.value / 3 * 1e-3) / 1e12) if w else 0:7.2f} TF/s", file=sys.stderr)
            print(f"  sum of kinds   {tot:8.3f} ms/step (serialised)", file=sys.stderr)
            _lib.check(lib.ral_profile_select(model.eng.h, b""))
        _lib.check(lib.ral_set_option(model.eng.h, b"lanes", int(os.environ.get("RAL_LANES", "2"))))
        _lib.check(lib.ral_set_option(model.eng.h, b"side_stream", 0 if os.environ.get("RAL_NO_SIDE_STREAM") else 1))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    loss = out["loss"].item()

    infer = infer_graph = None
    if not a.no_infer:
        # BASELINE config 4: eval-mode forward (BatchNorm running statistics), eager and hipGraph-captured
        from ecg_denoise_amd.infer import GraphedForward
        model.eval()
        for _ in range(2):
            model(x)
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            model(x)
        sync()
        ti = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        gf = GraphedForward(model, B)
        gf(x); sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            gf.graph.replay()
        sync()
        tg = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:          # replicas, no collective: the job's rate is set by the slowest rank
            dist.all_reduce(ti, op=dist.ReduceOp.MAX); dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        infer = B * world * a.steps / ti.item()
        infer_graph = B * world * a.steps / tg.item()

    if rank == 0:
        ksec = ms.value * 1e-3
        flops = kind_work(a.kind, a.L, B) * rl_steps
        ach = flops / ksec / 1e12 if ksec > 0 else 0.0
        peak = VALU_F32_PEAK_TF if a.kind.startswith("attn") else MFMA_F32_PEAK_TF
        res = {
            "metric": f"ECG windows/sec ({a.L}-sample, bs{B}) train step; inference forward in infer_*_windows_per_s",
            "value": round(B * world * a.steps / dt, 1), "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RA-LENet '{a.variant}' train step (fwd+mse/SNR/RMSE+bwd+Adam), "
                                   f"{a.leads}-lead {a.L}-sample windows, batch {B}/GPU, "
                                   f"N(0,1) inputs seed 2023, random-init weights",
                       "global_batch": B * world, "parallelism": f"dp{world}", "sync_bn": True},
            "final_loss": round(loss, 6),
            "roofline": {"bound": "valu" if a.kind.startswith("attn") else "mfma", "kernel": a.kind, "achieved": round(ach, 3), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": measured_traffic(a.kind),
                         "launches": int(cnt.value), "avg_launch_ms": round(ms.value / max(cnt.value, 1), 4),
                         "measured": f"hipEvent pairs on the kernel's stream over {rl_steps} serialised steps "
                                     "(lanes=1, no side stream) run right after the timed region"},
        }
        if infer is not None:
            res["infer_windows_per_s"] = round(infer, 1)
            res["infer_hipgraph_windows_per_s"] = round(infer_graph, 1)
        if world == 1 and not a.no_cpu:
            try:
                res["conv_stage_roofline"] = conv_stage_roofline(dev, a.L)
            except Exception as exc:      # an extra, never at the expense of the headline line
                res["conv_stage_roofline"] = {"error": str(exc)[:200]}
            res["cpu_baseline"] = cpu_baseline(a.leads, a.L, a.variant)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
