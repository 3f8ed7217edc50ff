"""Kernel-by-kernel timeline of one U-Net training step from a rocprofv3 kernel trace: start offset, duration and the gap
to the previous kernel of every launch (what is kernel time, what is between kernels).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/unet_trace -- python3 tools/unet_bench.py
    python tools/diag/unet_timeline.py gpurun_out/r03/unet_trace [step]"""
import csv, glob, re, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = max(glob.glob(d + "/*/*kernel_trace.csv"), key=__import__("os").path.getmtime)   # (newest: re-collections merge into the same directory)
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "k_adam" in r["n"]]
step = rows[adam[which] + 1: adam[which + 1] + 1]
t0 = step[0]["s"]
print(f"# step {which}: {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us from first start to last end")
prev = None
tk = tg = 0.0
for r in step:
    gap = (r["s"] - prev) / 1e3 if prev else 0.0
    dur = (r["e"] - r["s"]) / 1e3
    tk += dur; tg += max(gap, 0.0)
    print(f"{(r['s'] - t0) / 1e3:8.1f} us  +{gap:6.1f} gap  {dur:7.1f} us  grid {r.get('Grid_Size_X', '?'):>7s} lds {r.get('LDS_Block_Size', '?'):>6s}  {r['n'][:60]}")
    prev = r["e"]
print(f"# kernel time {tk:.1f} us, gaps {tg:.1f} us")
