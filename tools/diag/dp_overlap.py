"""Does the early gradient bucket's reduction overlap the rest of the backward pass?  Evidence from a rocprofv3 kernel +
memory-copy trace of `bench.py --gpus 2 --test-backend gloo --test-share-gpu` (two ranks on one GPU; gloo reduces a device
tensor by copying it to the host, so the collective shows up as device-to-host / host-to-device copies on the rank's
communication stream):
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03/dp2_trace -- python3 bench.py --gpus 2 ...
    python tools/diag/dp_overlap.py gpurun_out/r03/dp2_trace
Per rank and step (steps end with k_adam): when the first device-to-host copy after the start of the backward pass begins
(= bucket 1, the decoder half of the gradients, on its way to the all-reduce), when the backward pass ends, and how many
backward kernels start after that copy - i.e. run UNDER the collective."""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
ktr = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")))
print(f"# {len(ktr)} kernel traces under {d}")
for kf in ktr:
    mf = kf.replace("kernel_trace", "memory_copy_trace")
    ks = list(csv.DictReader(open(kf)))
    if not ks or not os.path.exists(mf):
        continue
    ms = list(csv.DictReader(open(mf)))
    for r in ks:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    ks.sort(key=lambda r: r["s"])
    cps = []
    for r in ms:
        direction = r.get("Direction", "")
        cps.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), direction))
    cps.sort()
    adam = [i for i, r in enumerate(ks) if r["n"].startswith("k_adam")]
    if len(adam) < 3:
        continue
    print(f"## {os.path.basename(kf)}: {len(ks)} kernels, {len(cps)} copies, {len(adam)} steps; copy directions: "
          f"{sorted(set(c[2] for c in cps))}")
    for si in range(max(1, len(adam) - 3), len(adam)):
        step = ks[adam[si - 1] + 1: adam[si] + 1]
        bwd = [r for r in step if "_bwd" in r["n"] or r["n"].startswith("k_dw") or "bn8_bwd" in r["n"]]
        if not bwd:
            continue
        b0, b1 = min(r["s"] for r in bwd), max(r["e"] for r in bwd)
        d2h = [c for c in cps if b0 <= c[0] <= step[-1]["s"] and "DEVICE_TO_HOST" in c[2].upper().replace(" ", "_")]
        if not d2h:
            print(f"step {si}: no device-to-host copy inside the backward pass"); continue
        c0 = d2h[0]
        under = [r for r in bwd if r["s"] >= c0[0]]
        after_cp_end = [r for r in bwd if r["s"] >= c0[1]]
        print(f"step {si}: backward {b0 and 0:.0f}..{(b1 - b0) / 1e6:.3f} ms; first device-to-host copy (bucket 1) starts at "
              f"{(c0[0] - b0) / 1e6:.3f} ms, takes {(c0[1] - c0[0]) / 1e3:.0f} us; {len(under)} backward kernels "
              f"({sum(r['e'] - r['s'] for r in under) / 1e6:.3f} ms of kernel time) START after it began, "
              f"{len(after_cp_end)} after it ended; {len(d2h)} device-to-host copies before Adam "
              f"(last at {(d2h[-1][0] - b0) / 1e6:.3f} ms)")
