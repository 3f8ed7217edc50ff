"""run: one DANet train loop (for rocprofv3 --kernel-trace); read: per-kernel timeline of one step (between k_adam launches)"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch
    from ecg_denoise_amd import DANet
    B = 2048
    m = DANet(L=512, max_batch=B, device="cuda:0", seed=1); m.train()
    x = torch.randn(B, 2, 512, device="cuda:0"); t = torch.randn(B, 2, 512, device="cuda:0")
    for _ in range(12): m.train_step(x, t)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "*", "*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if r["n"].startswith("k_adam")]
    step = rows[adam[8] + 1: adam[9] + 1]
    t0 = step[0]["s"]
    print(f"# {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us")
    agg = {}
    for r in step:
        d = (r["e"] - r["s"]) / 1e3
        a = agg.setdefault(r["n"][:60], [0, 0.0]); a[0] += 1; a[1] += d
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{d:8.1f} us {c:3d} x {d / c:6.1f}  {n}")
