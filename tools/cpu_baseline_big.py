"""the CPU baseline at the BENCH batch (2048 windows), next to the batch-32 figure bench.py reports.
One train step of the oracle (torch-CPU op graph of the reference, parity-pinned by tests/golden) at batch 2048
materialises ~50 GB of attention probabilities and takes tens of seconds, so it is not part of the default bench run; this
script measures it once per round on the GPU box's host (3 TB of memory, 256 logical cores) and its JSON line is committed
as profiles/rNN_cpu_baseline_b2048.json.    python tools/cpu_baseline_big.py [batch] [threads ...]"""
import json, os, sys, time
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ralenet_oracle as O


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


def run(B, leads, L, threads, steps):
    torch.set_num_threads(threads)
    p = O.init_params(O.ralenet_param_shapes("full", leads), 1)
    g = torch.Generator().manual_seed(2023)
    x = torch.randn(B, leads, L, generator=g); tgt = torch.randn(B, leads, L, generator=g)
    m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
    v = OrderedDict((k, torch.zeros_like(t)) for k, t in p.items())
    bn = O.new_bn_state()
    fwd = lambda pp, xx: O.ralenet_forward(pp, xx, "full", True, bn)
    O.train_step(p, x, tgt, fwd, m, v, 1)            # warm-up (allocator, thread pool)
    t0 = time.time()
    for i in range(steps):
        O.train_step(p, x, tgt, fwd, m, v, i + 2)
    return (time.time() - t0) / steps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    threads = [int(t) for t in sys.argv[2:]] or [8, 32, 64]
    need = 0.06 * B            # ~60 MB of autograd state per 512-sample window (measured: 12 MB of probabilities fwd, x ~5)
    avail = mem_available_gb()
    if avail < 1.5 * need:
        print(json.dumps({"error": f"batch {B} needs ~{need:.0f} GB, {avail:.0f} GB available"}))
        return
    out = {"workload": f"RA-LENet 'full' train step, 1-lead 512-sample windows, batch {B}, CPU oracle (kind: port)",
           "host_cores": os.cpu_count(), "mem_available_GB": round(avail), "runs": []}
    for th in threads:
        dt = run(B, 1, 512, th, 1)
        out["runs"].append({"threads": th, "s_per_step": round(dt, 2), "windows_per_s": round(B / dt, 1)})
        print(json.dumps(out["runs"][-1]), file=sys.stderr, flush=True)
    best = max(out["runs"], key=lambda r: r["windows_per_s"])
    out["value"], out["unit"], out["cores"] = best["windows_per_s"], "windows/s", best["threads"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
