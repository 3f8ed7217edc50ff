"""Headline benchmark: RA-LENet training-step throughput (ECG windows/s) on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--config ralenet|unet|newrale]

N > 1 runs one process per GPU over RCCL: either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or
bench.py starts them itself when WORLD_SIZE is not set (child processes, never exec; rank 0's JSON line is relayed).

One "step" = zero_grad -> forward -> mse/SNR/RMSE -> backward -> (all-reduce) -> Adam on one batch of synthetic windows
already resident in HBM.  Prints ONE JSON line (rank 0).  Workloads (`--config`):
  ralenet (default)  BASELINE config 2 / 3: RA-LENet "full", 1 lead x 512 samples, 2048 windows per GPU
  unet               the conv U-Net of the same path (UNet.py), 2 leads x 512 samples, 2048 windows per GPU
  newrale            BASELINE config 4: 12 leads x 1024 samples through the transfer-learning adapter, 256 windows per GPU
"""
import argparse
import ctypes as C
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CH = [8, 16, 32, 64, 128]
BLOCKS_PER_LEVEL = [2, 4, 4, 4, 4]
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA peak (the dtype every contraction here runs in)
VALU_F32_PEAK_TF = 157.3
KINDS = ("qkv_fwd", "attn_fwd", "mlp_fwd", "resample_fwd", "mlp_bwd", "attn_bwd", "qkv_bwd", "dw", "resample_bwd")


def kind_work(kind, L, B, executed=False):
    """Algorithmic FLOPs of all launches of one kernel kind in a train step.  Attention backward:
    SURVEY 8d's count is 2 x forward = 8 N^2 C per block (dV, dP, dK, dQ); the kernel also re-computes S (nothing N x N is
    stored), which `executed=True` adds (10 N^2 C) - a design cost, not work the reference asks for."""
    fl = 0.0
    for lvl, nb in enumerate(BLOCKS_PER_LEVEL):
        N, Cc = L >> lvl, CH[lvl]
        if kind == "attn_fwd":
            fl += nb * 4.0 * N * N * Cc * B                 # QK^T + PV, mul+add
        elif kind == "attn_bwd":
            fl += nb * (10.0 if executed else 8.0) * N * N * Cc * B
        elif kind == "mlp_fwd":
            fl += nb * 2.0 * N * Cc * Cc * 9 * B            # proj C*C + fc1 4C*C + fc2 4C*C
        elif kind == "mlp_bwd":
            fl += nb * 2.0 * N * Cc * Cc * 9 * B
        elif kind == "qkv_fwd" or kind == "qkv_bwd":
            fl += nb * 2.0 * N * Cc * Cc * 3 * B
        elif kind == "dw":
            fl += nb * 2.0 * N * Cc * Cc * 12 * B
    return fl


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _profile_rounds(suffix):
    """committed profiles/rNN_<suffix>, newest round first"""
    import glob
    return [os.path.basename(f) for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True)]


def measured_traffic(kind):
    """HBM bytes per launch of this kernel kind from the committed PMC run (newest profiles/rNN_hbm_traffic.json)."""
    for name in _profile_rounds("hbm_traffic.json"):
        d = _profile_json(name)
        if d and kind in d.get("per_launch_bytes", {}):
            return d["per_launch_bytes"][kind]["total"]
    return None


def measured_counters(kind):
    """(matrix-pipe busy fraction, vector-issue busy fraction, file) of this kernel kind from the newest committed SQ counter
    run (profiles/rNN_sq_counters.json, tools/sq_counters.py): what the pipes did, as opposed to useful FLOP over a peak"""
    for name in _profile_rounds("sq_counters.json"):
        d = _profile_json(name)
        k = (d or {}).get("per_kind", {}).get(kind)
        if k and "mfma_busy" in k:
            return k.get("mfma_busy"), k.get("valu_issue"), "profiles/" + name
    return None, None, None


# the pipe a kernel kind's contractions execute on in the default arithmetic; `peak` in the roofline
# object stays the fp32 figure the path's dtype is priced against - the f16 matrix pipe's own dense peak is ~2.5 PFLOP/s,
# which is why `mfma_busy` / `valu_issue` are reported next to `frac`
KIND_PIPE = {"attn_bwd": "f16 mfma (fp16-pair operands, fp32 accumulate) + valu exp/convert", "attn_fwd": "valu (exp2, packed fma) + f16 mfma score tiles",
             "mlp_fwd": "f16 mfma (fp16 pairs) + valu gelu", "mlp_bwd": "f16 mfma (fp16 pairs; C <= 16: f32 mfma) + valu gelu",
             "qkv_fwd": "f16 mfma (C >= 64) / f32 mfma", "qkv_bwd": "f16 mfma (C >= 64) / f32 mfma", "dw": "f16 mfma (C >= 64) / f32 mfma",
             "resample_fwd": "f32 mfma", "resample_bwd": "f32 mfma"}
STEP_FLOP_PER_WINDOW = 441.7e6      # fwd 147.2 MFLOP x 3 (SURVEY 8d / DESIGN.md section 3), 512-sample window


def roofline_object(kind, config, L, B, kind_ms, launches, rl_steps, attn_ms, n_timed, dt):
    """The `roofline` object of the JSON line for an RA-LENet workload.  kind_ms: summed hipEvent time of the `launches` launches
    of `kind` over `rl_steps` serialised steps; attn_ms: {"attn_fwd", "attn_bwd"} -> serialised ms per step; n_timed / dt:
    steps and seconds of the timed region (the whole step's useful FLOP rate).  `frac` is useful FLOP (SURVEY 8d's count)
    over the fp32 peak; what the pipes did is `mfma_busy` / `valu_issue` from the newest committed counter run."""
    ksec = kind_ms * 1e-3
    ach = kind_work(kind, L, B) * rl_steps / ksec / 1e12 if ksec > 0 else 0.0
    peak = VALU_F32_PEAK_TF if kind.startswith("attn") else MFMA_F32_PEAK_TF
    busy, vissue, csrc = measured_counters(kind) if config == "ralenet" else (None, None, None)
    step_tf = STEP_FLOP_PER_WINDOW * (L / 512.0) * B * n_timed / dt / 1e12 if config == "ralenet" and dt > 0 else None
    blk = None
    if attn_ms.get("attn_fwd") and attn_ms.get("attn_bwd"):
        fl = kind_work("attn_fwd", L, B) + kind_work("attn_bwd", L, B)
        blk = fl / ((attn_ms["attn_fwd"] + attn_ms["attn_bwd"]) * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": kind, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
         "traffic": measured_traffic(kind) if config == "ralenet" else None,
         "pipe": KIND_PIPE.get(kind),
         "peak_is": "fp32 vector = fp32 matrix peak (the path's dtype); the f16 matrix pipe the fp16-pair products execute on peaks at "
                    "~2.5 PFLOP/s dense, so `frac` is useful fp32-equivalent FLOP over the fp32 peak, not pipe occupancy: that is "
                    "mfma_busy / valu_issue",
         "mfma_busy": busy, "valu_issue": vissue, "counters_source": csrc,
         "step_frac": round(step_tf / MFMA_F32_PEAK_TF, 4) if step_tf else None,
         "step_TFLOPs": round(step_tf, 2) if step_tf else None,
         "attn_block_frac": round(blk / VALU_F32_PEAK_TF, 4) if blk else None,
         "attn_block": ({"fwd_ms": round(attn_ms["attn_fwd"], 3), "bwd_ms": round(attn_ms["attn_bwd"], 3),
                         "fwd_frac": round(kind_work("attn_fwd", L, B) / (attn_ms["attn_fwd"] * 1e-3) / 1e12 / VALU_F32_PEAK_TF, 4),
                         "bwd_frac": round(kind_work("attn_bwd", L, B) / (attn_ms["attn_bwd"] * 1e-3) / 1e12 / VALU_F32_PEAK_TF, 4),
                         "TFLOPs": round(blk, 2),
                         "what": "serialised per-step times of all attention forward / backward launches, SURVEY 8d FLOP "
                                 "(4 / 8 N^2 C per block)"} if blk else None),
         "launches": launches, "avg_launch_ms": round(kind_ms / max(launches, 1), 4),
         "flop_count": "algorithmic (SURVEY 8d)",
         "measured": f"hipEvent pairs on the kernel's stream over {rl_steps} serialised steps (lanes=1, no side stream) run right "
                     "after the timed region"}
    if kind == "attn_bwd":       # the S re-computation counted as well (what the kernel executes per visit once)
        ach10 = kind_work(kind, L, B, True) * rl_steps / ksec / 1e12 if ksec > 0 else 0.0
        r["flop_count"] = "8 N^2 C per block (SURVEY 8d: 2 x forward); with the S re-computation (10 N^2 C) in *_incl_recompute"
        r["achieved_incl_recompute"] = round(ach10, 3)
        r["frac_incl_recompute"] = round(ach10 / peak, 4)
    return r


def unet_fused_traffic():
    """(bytes per launch of the fused U-Net inference kernel at batch 2048 from the committed PMC run, file name)"""
    for name in _profile_rounds("unet_hbm_traffic.json"):
        d = _profile_json(name)
        if d:
            f = d.get("fused", {})
            tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in f.values())
            if tot:
                return tot, name
    return None, None


def conv_stage_roofline(dev, L, B=2048):
    """The HBM-bound conv stages of the path (north star: fraction of the HBM roofline on the U-Net stages at batch
    2048 x 512): U-Net eval forward, 2 leads, at the STATED batch.
      `achieved` / `frac` (top level) = the STAGED path, where every stage tensor really makes its HBM round trip: 11 conv
        launches + the output BatchNorm pass, SURVEY 8d's stage-granular count (every conv reads its input and writes its
        output once, the three decoder skips and the bottleneck residual are re-read: 26 tensors, + 1 for the extra pass)
        over the measured time.  This is the figure that is comparable with the 8 TB/s peak.
      `fused`: the product path for inference - the whole forward in ONE kernel, stage tensors in LDS.  It is NOT an
        HBM-bound kernel, so it carries no fraction of the HBM peak: its HBM rate uses the bytes it really moves (PMC
        FETCH + WRITE of the committed profile, scaled by batch), its bound is the fp32 MFMA / latency (`mfma_frac` =
        0.76 MFLOP per window over the time against 157.3 TF/s); `stage_equivalent_GBps` (26 tensors per window over the
        time) is kept only as the speed-up over a staged implementation."""
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
    _lib.check(_lib.lib().ral_set_option(m.eng.h, b"unet_fused", 0))
    dt = graph_time()
    out.update({"kernel": "U-Net eval forward stage by stage: 11 conv launches + output BatchNorm pass, hipGraph replay",
                "achieved": round(B * 27 * w / dt / 1e9, 1), "frac": round(B * 27 * w / dt / 1e9 / HBM_PEAK_GBS, 4),
                "tensors_per_window": 27, "us_per_forward": round(dt * 1e6, 1), "windows_per_s": round(B / dt, 1)})
    _lib.check(_lib.lib().ral_set_option(m.eng.h, b"unet_fused", 1))
    dt = graph_time()
    moved, src = unet_fused_traffic()       # bytes per launch at batch 2048 (PMC), or None
    moved_pw = moved / 2048.0 if moved else None
    flop_pw = 0.76e6 * (L / 512.0)
    out["fused"] = {"kernel": "whole forward in one kernel (k_unet_pack + k_unet_infer), hipGraph replay",
                    "bound": "mfma/latency", "us_per_forward": round(dt * 1e6, 1), "windows_per_s": round(B / dt, 1),
                    "mfma_TFLOPs": round(B * flop_pw / dt / 1e12, 2), "mfma_frac": round(B * flop_pw / dt / 1e12 / MFMA_F32_PEAK_TF, 4),
                    "hbm_bytes_moved_per_window": round(moved_pw) if moved_pw else None, "hbm_bytes_source": src,
                    "hbm_GBps_moved": round(B * moved_pw / dt / 1e9, 1) if moved_pw else None,
                    "stage_equivalent_GBps": round(B * 26 * w / dt / 1e9, 1)}
    del m, x
    torch.cuda.empty_cache()
    return out


def cpu_baseline(leads, L, variant, big_batch=256):
    """The oracle (CPU restatement of the reference op graph, parity-pinned by tests/golden) timed on the host cores of
    this box on bounded samples: batch 32 (the reference's own batch, BASELINE config 0) is `value`; `large_batch` is the
    same op graph at a batch that amortises the per-call overhead (SURVEY 8d asks for the bench batch 2048: one such
    step materialises ~50 GB of attention probabilities and takes ~25 s, so the default run times batch 256 and
    tools/cpu_baseline_big.py records the 2048 figure in profiles/)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from collections import OrderedDict
    import torch
    import ralenet_oracle as O
    # intra-op threads: the count with the highest measured throughput on the GPU box's host (256 logical cores;
    # tools/diag/cpu_threads.py: 1 thread 40, 2: 60, 4: 74, 8: 84, 16: 78, 32: 56, 64: 26 windows/s) - the op graph is
    # ~21k small ATen calls per step, and past 8 threads the fork/join cost of each call exceeds its work
    cores = min(os.cpu_count() or 1, 8)

    def run(B, threads, budget, max_steps):
        torch.set_num_threads(threads)
        p = O.init_params(O.ralenet_param_shapes(variant, leads), 1)
        g = torch.Generator().manual_seed(2023)
        x = torch.randn(B, leads, L, generator=g); tgt = torch.randn(B, leads, L, generator=g)
        m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
        v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
        bn = O.new_bn_state()
        fwd = lambda pp, xx: O.ralenet_forward(pp, xx, variant, True, bn)
        O.train_step(p, x, tgt, fwd, m, v, 1)
        n, t0 = 0, time.time()
        while time.time() - t0 < budget and n < max_steps:
            O.train_step(p, x, tgt, fwd, m, v, n + 2)
            n += 1
        return n, time.time() - t0
    B = 32
    n, dt = run(B, cores, 10.0, 40)
    res = {"value": round(B * n / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port",
           "sample": f"{n} train steps of the CPU oracle at batch {B} x {leads} x {L} fp32 "
                     f"(torch-CPU op graph of the reference, {cores} threads)"}
    if big_batch:
        th = min(os.cpu_count() or 1, 32)      # large tensors: more threads pay (the per-call fork/join is amortised)
        try:
            n2, dt2 = run(big_batch, th, 8.0, 3)
            res["large_batch"] = {"value": round(big_batch * n2 / dt2, 2), "batch": big_batch, "cores": th,
                                  "sample": f"{n2} train steps at batch {big_batch} x {leads} x {L}"}
        except Exception as exc:
            res["large_batch"] = {"error": str(exc)[:200]}
    # the figure at the bench batch itself, measured once per round by tools/cpu_baseline_big.py on the GPU box's host
    # (40 s per step and ~50 GB of autograd state: not re-measured inside the default run) and committed under profiles/
    for name in _profile_rounds("cpu_baseline_b2048.json"):
        d = _profile_json(name)
        if d and "value" in d and leads == 1 and L == 512:
            res["bench_batch"] = {"value": d["value"], "batch": 2048, "cores": d.get("cores"), "source": "profiles/" + name,
                                  "s_per_step": min(r["s_per_step"] for r in d.get("runs", [{"s_per_step": None}]))}
            break                                   # (the newest round's file)
    return res


def self_launch(n, argv):
    """Start the n ranks of a single-node job as CHILD processes of this one (which has not touched the GPU and never
    will), relay rank 0's stdout.  All children are polled: the first rank that exits non-zero (or a SIGTERM / Ctrl-C to
    this process) terminates the others, so a failed or deadlocked job ends instead of hanging.  Returns the exit code."""
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
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_all(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()
    old = signal.signal(signal.SIGTERM, lambda *_: (stop_all(), sys.exit(143)))
    bad = []
    try:
        while any(p.poll() is None for p in procs):
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad:
                stop_all()
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        raise
    finally:
        signal.signal(signal.SIGTERM, old)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    sys.stdout.write("".join(o or "" for o in out))
    sys.stdout.flush()
    bad = bad or [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
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
    if a.test_fail_rank == rank:          # a failing rank must fail the launcher
        sys.exit(3)
    if a.test_hang_rank == rank:          # a rank that never finishes: the launcher must end it when another one fails
        time.sleep(3600)
    facts = collective_facts(dist, world, rank, int(os.environ.get("LOCAL_RANK", "0")), "gloo" if world > 1 else None, None)
    if rank == 0:
        B, G = batch_plan(a, world)
        print(json.dumps({"dry_run": True, "n_gpus": world, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "rank_sum": t.item(), "steps": a.steps, "warmup": a.warmup,
                          "batch_per_gpu": B, "global_batch": G, "config3_leg": config3_batch(a, world), "collective": facts,
                          "unet_dp": unet_dp_modes(world) if a.config == "unet" else None}))
    if world > 1:
        dist.destroy_process_group()


def batch_plan(a, world):
    """-> (windows per GPU, global batch).  `--global-batch G` fixes the job's batch (G / world per GPU: strong scaling);
    otherwise `--batch` (per GPU) or the config's stated per-GPU batch (weak scaling: 2048 windows per GPU for RA-LENet and
    U-Net, 256 for newrale)."""
    default = 256 if a.config == "newrale" else 2048
    if a.global_batch:
        if a.global_batch % world:
            raise SystemExit(f"--global-batch {a.global_batch} is not a multiple of the {world} ranks")
        return a.global_batch // world, a.global_batch
    B = a.batch or default
    return B, B * world


def config3_batch(a, world):
    """BASELINE config 3 is RA-LENet at global batch 8192 on 8 GPUs (8 x 1024 windows; SURVEY 8d 'C2').  The headline line
    keeps 2048 windows per GPU at every N (weak scaling, so that the driver's per-N values are comparable); at --gpus 8 the
    same process ALSO times config 3's own shape and reports it as `config3` - per-GPU batch of that leg, or None."""
    if a.config == "ralenet" and world == 8 and not a.batch and not a.global_batch and not a.leads and not a.L:
        return 1024
    return None


UNET_DP_DEVIATION = {"loss_rel": 3.5e-5, "grad_rel_of_norm": 3.6e-2, "at": "2 ranks x 64 windows, random initialisation, fp64 stand-in",
                     "source": "tests/test_dp_gloo.py::test_unet_per_rank_batchnorm_statistics_stay_within_1e_3_of_the_global_batch_step"}


def unet_dp_modes(world):
    """U-Net under data parallelism has a BatchNorm after every conv (UNet.py:46-141).  Two modes, both timed at N > 1:
      per_rank_bn (HEADLINE): every rank normalises with its own shard's statistics - what DistributedDataParallel does without
        SyncBatchNorm - the fused single-GPU step plus ONE gradient all-reduce;
      sync_bn (exact): the global batch's statistics, 10 + 10 dependent all-reduces of 64 doubles inside a 0.4 ms step + the
        gradient all-reduce = 21 latency-bound collectives: it reproduces the single-process step, and cannot scale."""
    if world <= 1:
        return None
    return {"headline_mode": "per_rank_bn",
            "per_rank_bn": {"collectives_per_step": 1, "sync_bn": False, "deviation_from_global_batch_step": UNET_DP_DEVIATION},
            "sync_bn": {"collectives_per_step": 21, "sync_bn": True, "deviation_from_global_batch_step": None}}


def collective_facts(dist, world, rank, local, backend, trainer):
    """What proves that N ranks on N devices took part: backend, RCCL version, the world size as torch.distributed sees it,
    the device index of every rank (gathered) and the collectives one train step issues (counted on this rank)."""
    if world <= 1:
        return None
    import torch
    devs = [None] * world
    dist.all_gather_object(devs, {"rank": rank, "local_rank": local, "host": socket.gethostname(),
                                   "device": (torch.cuda.current_device() if torch.cuda.is_available() else None)})
    ver = None
    try:
        if backend == "nccl":
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return {"backend": dist.get_backend(), "rccl_version": ver, "world_size": dist.get_world_size(), "ranks": devs,
            "collectives_per_step": getattr(trainer, "collectives_last_step", None) if trainer is not None else None,
            "metric_collectives_per_step": getattr(trainer, "metric_collectives_last_step", None) if trainer is not None else None}


def build_workload(a, dev, rank):
    """-> dict(model, inner (the RA-LENet handle kernels are profiled on, or None), trainer, x, tgt, B, leads, L, text)"""
    import torch
    from ecg_denoise_amd import NewRALE, RALENet, UNet
    from ecg_denoise_amd.dp import DataParallelTrainer, HipEngineAdapter, NewRALEEngineAdapter, UNetEngineAdapter
    g = torch.Generator().manual_seed(2023 + rank)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    Bplan = a.batch_override or batch_plan(a, world)[0]
    if a.config == "newrale":
        B, leads, L = Bplan, 12, a.L or 1024
        inner = RALENet(a.variant, leads=2, L=L, max_batch=B, train=True, device=dev, seed=2023)
        model = NewRALE(inner, seed=2024)
        eng = NewRALEEngineAdapter(model)
        text = (f"newrale (12-lead adapter around a frozen RA-LENet '{a.variant}', ralenet_12leads.py:680-709) train step, "
                f"12-lead {L}-sample windows, batch {B}/GPU")
    elif a.config == "unet":
        B, leads, L = Bplan, a.leads or 2, a.L or 512
        model = inner_none = UNet(leads=leads, L=L, max_batch=B, train=True, device=dev, seed=2023)
        inner, eng = None, UNetEngineAdapter(model)
        text = f"U-Net (UNet.py:96-141) train step, {leads}-lead {L}-sample windows, batch {B}/GPU"
    else:
        B, leads, L = Bplan, a.leads or 1, a.L or 512
        model = inner = RALENet(a.variant, leads=leads, L=L, max_batch=B, train=True, device=dev, seed=2023)
        eng = HipEngineAdapter(model)
        text = f"RA-LENet '{a.variant}' train step, {leads}-lead {L}-sample windows, batch {B}/GPU"
    x = torch.randn(B, leads, L, generator=g).to(dev)
    tgt = torch.randn(B, leads, L, generator=g).to(dev)
    model.train()
    sync_bn = not (a.config == "unet" and world > 1 and not getattr(a, "unet_sync_bn_leg", False))   # U-Net at N > 1: per-rank statistics are the headline
    return {"model": model, "inner": inner, "trainer": DataParallelTrainer(eng, sync_bn=sync_bn), "sync_bn": sync_bn, "x": x, "tgt": tgt, "B": B, "leads": leads,
            "L": L, "text": text + " (fwd+mse/SNR/RMSE+bwd+Adam), N(0,1) inputs seed 2023, random-init weights"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="the timed region repeats the --steps block until it lasts at least this long (whole blocks; "
                         "`steps` in the line = steps actually timed, `steps_requested` = --steps; 0 = exactly --steps)")
    ap.add_argument("--config", default="ralenet", choices=("ralenet", "unet", "newrale"))
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU (default: the config's stated batch)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="windows of the whole job (G / N per GPU: strong scaling); default: the per-GPU batch at every N (weak)")
    ap.add_argument("--leads", type=int, default=0)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--variant", default="full")
    ap.add_argument("--kind", default="attn_bwd", help="kernel kind timed for the roofline object")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and the other N = 1 extras")
    ap.add_argument("--no-infer", action="store_true", help="skip the inference-forward leg")
    ap.add_argument("--no-fp32", action="store_true", help="skip the strict-fp32 leg (profile collection: one arithmetic per trace)")
    ap.add_argument("--kinds", action="store_true", help="print a per-kernel-kind time table to stderr (3 steps each)")
    ap.add_argument("--opt", action="append", default=[], help="library switch key=value (ral_global_option; diagnostics only, repeatable)")
    ap.add_argument("--dry-run-launcher", action="store_true", help="CPU/gloo rendezvous only: tests the process launcher")
    # test hooks (tests/test_gpu_dp_procs.py, tests/test_bench_cpu.py): never part of a measurement
    ap.add_argument("--test-share-gpu", action="store_true", help="every rank on device 0 (a box with one GPU)")
    ap.add_argument("--test-backend", default="nccl", help="collective backend; gloo when ranks share a device")
    ap.add_argument("--test-config3-batch", type=int, default=0, help="run the config-3 leg at this per-GPU batch whatever N is")
    ap.add_argument("--test-fail-rank", type=int, default=-1)
    ap.add_argument("--test-hang-rank", type=int, default=-1)
    a = ap.parse_args()
    a.batch_override = 0

    # N > 1 and nobody started the ranks for us: start them (before torch is imported or the GPU touched)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    if a.dry_run_launcher:
        return launcher_dry_run(a)

    # more hardware queues than the default 4, before the HIP runtime initialises: the data-parallel communication stream
    # must not share a queue with a compute chain (ecg_denoise_amd/__init__.py; no effect on the single-GPU step)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    from ecg_denoise_amd import _lib

    _lib.apply_options(",".join(a.opt))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs WORLD_SIZE={a.gpus} (launch with torch.distributed.run)")
    if a.test_share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.test_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(a.test_backend)

    W = build_workload(a, dev, rank)
    model, inner, trainer, x, tgt, B, L = W["model"], W["inner"], W["trainer"], W["x"], W["tgt"], W["B"], W["L"]
    Lk, Bk = L, B                         # shape the profiled RA-LENet kernels see (newrale: the inner (B, 2, L) model)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # world == 1 and U-Net: the fused entry points (ral_forward / ral_backward); the stage-by-stage adapter is the
    # data-parallel form (one reduction per BatchNorm layer)
    step = (lambda: model.train_step(x, tgt)) if (world == 1 and a.config == "unet") else (lambda: trainer.train_step(x, tgt))
    for _ in range(a.warmup):
        step()
    sync()
    lib = _lib.lib()
    # A timed region of --steps steps is 0.26 s at the default 20: too short for anything sampling the GPU from outside (the
    # driver's 5 s SMI samples never land in it).  One untimed calibration block of --steps steps sizes the region: it repeats
    # the block until >= --min-seconds, with no synchronisation between blocks; every rank uses the slowest rank's estimate.
    steps_req, blocks = a.steps, 1
    if a.min_seconds > 0 and a.steps > 0:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        tb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        blocks = max(1, int(-(-a.min_seconds // max(tb.item(), 1e-6))))
    a.steps = steps_req * blocks
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(a.steps):
        out = step()
        ev[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
    median_ms = per_step[len(per_step) // 2] if per_step else None
    n_timed, a.steps = a.steps, steps_req          # (the legs below size themselves from --steps)
    blocks_calibrated = a.min_seconds > 0 and steps_req > 0

    # roofline leg: the same step with the kernels serialised (one lane, no side stream), so that the hipEvent
    # pair around each launch of the selected kernel measures that kernel alone, not its share of a busy GPU
    rl_steps = max(3, a.steps // 4)
    ms, cnt = C.c_double(), C.c_int64()
    if inner is not None:
        h = inner.eng.h
        _lib.check(lib.ral_set_option(h, b"lanes", 1))
        _lib.check(lib.ral_set_option(h, b"side_stream", 0))
        step()
        sync()
        _lib.check(lib.ral_profile_select(h, a.kind.encode()))
        for _ in range(rl_steps):
            step()
        sync()
        _lib.check(lib.ral_profile_read(h, C.byref(ms), C.byref(cnt)))
        _lib.check(lib.ral_profile_select(h, b""))
        # the attention block as a whole (north star: "MFMA utilisation on the attention block"): forward + backward kernels
        attn_ms = {}
        for kind in ("attn_fwd", "attn_bwd"):
            if kind == a.kind:
                attn_ms[kind] = ms.value / rl_steps
                continue
            _lib.check(lib.ral_profile_select(h, kind.encode()))
            for _ in range(3):
                step()
            sync()
            ms2, cnt2 = C.c_double(), C.c_int64()
            _lib.check(lib.ral_profile_read(h, C.byref(ms2), C.byref(cnt2)))
            attn_ms[kind] = ms2.value / 3
        _lib.check(lib.ral_profile_select(h, b""))
        if a.kinds:                        # every rank runs the steps (they contain collectives); rank 0 prints
            tot = 0.0
            for kind in KINDS:
                _lib.check(lib.ral_profile_select(h, kind.encode()))
                for _ in range(3):
                    step()
                sync()
                ms2, cnt2 = C.c_double(), C.c_int64()
                _lib.check(lib.ral_profile_read(h, C.byref(ms2), C.byref(cnt2)))
                w = kind_work(kind, Lk, Bk)
                tot += ms2.value / 3
                if rank == 0:
                    print(f"  {kind:14s} {ms2.value/3:8.3f} ms/step  {cnt2.value//3:4d} launches  "
                          f"{(w / (ms2.value / 3 * 1e-3) / 1e12) if w and ms2.value else 0:7.2f} TF/s", file=sys.stderr)
            if rank == 0:
                print(f"  sum of kinds   {tot:8.3f} ms/step (serialised)", file=sys.stderr)
            _lib.check(lib.ral_profile_select(h, b""))
        _lib.check(lib.ral_set_option(h, b"lanes", int(os.environ.get("RAL_LANES", "2"))))
        _lib.check(lib.ral_set_option(h, b"side_stream", 0 if os.environ.get("RAL_NO_SIDE_STREAM") else 1))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    loss = out["loss"].item()

    # strict-fp32 leg: the same step with every product on the fp32 MFMA (f16_split = 0: no fp16-pair operand anywhere),
    # timed right after the headline region on the same box, so that the line carries both arithmetics
    fp32_leg = None
    if inner is not None and not a.no_fp32:
        h = inner.eng.h
        _lib.check(lib.ral_set_option(h, b"f16_split", 0))
        n32 = max(5, min(10, a.steps))
        for _ in range(2):
            step()
        sync()
        t1 = time.perf_counter()
        for _ in range(n32):
            step()
        sync()
        t32 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t32, op=dist.ReduceOp.MAX)
        fp32_leg = {"ms_per_step": round(t32.item() / n32 * 1e3, 3), "value": round(B * world * n32 / t32.item(), 1),
                    "unit": "windows/s", "steps": n32,
                    "what": "the same step with ral_set_option f16_split=0: every contraction on v_mfma_f32_16x16x4_f32 / the vector ALU"}
        _lib.check(lib.ral_set_option(h, b"f16_split", int(os.environ.get("RAL_F16_SPLIT", "64"))))
        step(); sync()

    infer = infer_graph = None
    if not a.no_infer and a.config != "newrale":
        # BASELINE config 5's kernel path: eval-mode forward (BatchNorm running statistics), eager and hipGraph-captured
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
        model.train()

    # what proves N ranks on N devices took part (every rank: it gathers)
    facts = collective_facts(dist, world, rank, local, a.test_backend if world > 1 else None, trainer)

    # BASELINE config 3's own shape (global batch 8192 = 8 x 1024) next to the weak-scaling headline, in the same job
    config3 = None
    b3 = config3_batch(a, world) or (a.test_config3_batch if world > 1 else 0)
    if b3:
        a.batch_override = b3
        W3 = build_workload(a, dev, rank)
        for _ in range(max(2, a.warmup)):
            W3["trainer"].train_step(W3["x"], W3["tgt"])
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            W3["trainer"].train_step(W3["x"], W3["tgt"])
        sync()
        t3 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        config3 = {"workload": W3["text"] + f"; global batch {b3 * world} = {world} x {b3} (BASELINE configs[2])",
                   "global_batch": b3 * world, "batch_per_gpu": b3, "value": round(b3 * world * a.steps / t3.item(), 1),
                   "unit": "windows/s", "ms_per_step": round(t3.item() / a.steps * 1e3, 3), "steps": a.steps}
        del W3
        a.batch_override = 0

    # U-Net at N > 1: the exact mode (global-batch statistics, 21 collectives per step) next to the per-rank headline
    unet_dp = None
    if a.config == "unet" and world > 1:
        unet_dp = unet_dp_modes(world)
        unet_dp["per_rank_bn"].update({"value": round(B * world * n_timed / dt, 1), "unit": "windows/s", "ms_per_step": round(dt / n_timed * 1e3, 3),
                                       "collectives_counted": trainer.collectives_last_step})
        a.unet_sync_bn_leg = True
        Ws = build_workload(a, dev, rank)
        a.unet_sync_bn_leg = False
        for _ in range(max(2, a.warmup)):
            Ws["trainer"].train_step(Ws["x"], Ws["tgt"])
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            Ws["trainer"].train_step(Ws["x"], Ws["tgt"])
        sync()
        ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        unet_dp["sync_bn"].update({"value": round(B * world * a.steps / ts.item(), 1), "unit": "windows/s",
                                   "ms_per_step": round(ts.item() / a.steps * 1e3, 3), "steps": a.steps,
                                   "collectives_counted": Ws["trainer"].collectives_last_step})
        del Ws

    if rank == 0:
        res = {
            "metric": f"ECG windows/sec ({L}-sample, bs{B}) train step; inference forward in infer_*_windows_per_s",
            "value": round(B * world * n_timed / dt, 1), "unit": "windows/s",
            "n_gpus": world, "steps": n_timed, "steps_requested": steps_req, "timed_s": round(dt, 3),
            "warmup": a.warmup + (steps_req if blocks_calibrated else 0), "ms_per_step": round(dt / n_timed * 1e3, 3),
            "median_ms_per_step_hipevent": round(median_ms, 3) if median_ms else None,
            "higher_is_better": True, "scaling": "strong" if a.global_batch else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 tensors and accumulators; the Linear layers of the two wide levels (C = 64, 128: forward, "
                          "data-gradient and weight-gradient products), the MLP forward of the narrow levels (C = 8, 16, 32), the channel products of "
                          "the C = 32 MLP backward and every contraction of the attention backward plus the forward's score tiles "
                          "(S = q k^T, dP = dO v^T, P and dS against v, dO, q, k; head_dim 4) are error-compensated fp16-pair products on the "
                          "f16 matrix cores (x = h1 + h2, ~2^-22 relative; operands brought into range by powers of two per "
                          "weight matrix / token / head, nothing clamped); everything else on the fp32 MFMA / vector ALU.  "
                          "fp32_mfma: the same step with ral_set_option f16_split=0 (every product on the fp32 MFMA)",
            "config": {"workload": W["text"], "global_batch": B * world, "batch_per_gpu": B, "parallelism": f"dp{world}", "sync_bn": W["sync_bn"]},
            "final_loss": round(loss, 6),
        }
        if facts is not None:
            res["collective"] = facts
        if config3 is not None:
            res["config3"] = config3
        if unet_dp is not None:
            res["unet_dp"] = unet_dp
        if inner is not None:
            res["roofline"] = roofline_object(a.kind, a.config, Lk, Bk, ms.value, int(cnt.value), rl_steps, attn_ms, n_timed, dt)
        else:
            # U-Net: every kernel of the step is an HBM-bound conv stage; stage-granular bytes of SURVEY 8d per window
            # (forward with batch statistics 147 KB + backward 200 KB at 2 leads x 512 samples) over the whole step
            by = 347e3 * (W["leads"] * L) / 1024.0
            ach = by * B * world * n_timed / dt / 1e9 / world
            res["roofline"] = {"bound": "hbm", "kernel": "whole U-Net train step (all conv stages)", "achieved": round(ach, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                               "measured": "stage-granular algorithmic bytes per window (SURVEY 8d) x windows / step time, per GPU"}
        if fp32_leg is not None:
            res["fp32_mfma"] = fp32_leg
        if infer is not None:
            res["infer_windows_per_s"] = round(infer, 1)
            res["infer_hipgraph_windows_per_s"] = round(infer_graph, 1)
        if world == 1 and not a.no_cpu and a.config == "ralenet":
            # the reference-parity shape next to the single-lead headline ("also report leads = 2")
            try:
                del model, trainer, W
                torch.cuda.empty_cache()
                from ecg_denoise_amd import RALENet
                m2 = RALENet(a.variant, leads=2, L=L, max_batch=B, train=True, device=dev, seed=2023)
                x2 = torch.randn(B, 2, L, device=dev); t2 = torch.randn(B, 2, L, device=dev)
                m2.train()
                for _ in range(3):
                    m2.train_step(x2, t2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    m2.train_step(x2, t2)
                torch.cuda.synchronize()
                d2 = (time.perf_counter() - t1) / 10
                res["leads2"] = {"value": round(B / d2, 1), "unit": "windows/s", "ms_per_step": round(d2 * 1e3, 3),
                                 "workload": f"the same step on 2-lead windows (the reference's own shape), batch {B}, 10 steps"}
                del m2, x2, t2
                torch.cuda.empty_cache()
            except Exception as exc:
                res["leads2"] = {"error": str(exc)[:200]}
            try:
                res["conv_stage_roofline"] = conv_stage_roofline(dev, L)
            except Exception as exc:      # an extra, never at the expense of the headline line
                res["conv_stage_roofline"] = {"error": str(exc)[:200]}
            res["cpu_baseline"] = cpu_baseline(a.leads or 1, L, a.variant)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
