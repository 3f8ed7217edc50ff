"""Timeline of one training step in the DEFAULT schedule from a rocprofv3 kernel trace (tools/collect_profiles.sh writes
gpurun_out/<round>/ks_default/*/*_kernel_trace.csv): per hardware queue (two lane chains + two weight-gradient side
streams) the kernel count, first start, last end and summed kernel time; how long 0 / 1 / 2 / 3 / 4 kernels were in flight;
when the side streams finish relative to the chains (a tail there would mean the weight-gradient kernels are the
critical path), and the gaps between consecutive kernels of a queue.

    python tools/diag/step_timeline.py gpurun_out/r02/ks_default [step_index]
"""
import csv
import glob
import re
import sys
from collections import Counter


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02/ks_default"
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 4           # a step of the timed region (2 warm-up steps first)
    f = max(glob.glob(d + "/*/*kernel_trace.csv"), key=__import__("os").path.getmtime)   # (newest: re-collections merge into the same directory)
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if "adam" in r["n"].lower()]
    step = rows[adam[which] + 1: adam[which + 1] + 1]
    t0, t1 = step[0]["s"], step[-1]["e"]
    ms = lambda t: (t - t0) / 1e6
    print(f"# step {which} of {f}: {len(step)} kernels, {ms(t1):.3f} ms")
    queues = {}
    for r in step:
        queues.setdefault(r["Queue_Id"], []).append(r)
    for q, rs in sorted(queues.items()):
        busy = sum(r["e"] - r["s"] for r in rs) / 1e6
        dw = [r for r in rs if r["n"].startswith("k_dw")]
        role = "weight-gradient side stream" if len(dw) > len(rs) // 2 else "lane chain"
        print(f"queue {q} ({role}): {len(rs):3d} kernels, first start {ms(rs[0]['s']):7.3f}, last end "
              f"{ms(max(r['e'] for r in rs)):7.3f}, summed kernel time {busy:7.3f} ms; last kernel {rs[-1]['n'][:44]}")
    # gaps between consecutive kernels of one queue.  rocprofv3 makes back-to-back kernels of a queue abut (no gap), so what
    # shows up here are the long ones: a kernel whose first wave had to wait - for an event of another stream, or (the
    # common case at the wide levels) for LDS / registers that the other lane's workgroups still hold
    for q, rs in sorted(queues.items()):
        rs = sorted(rs, key=lambda r: r["s"])
        gaps = sorted((b["s"] - a["e"]) / 1e3 for a, b in zip(rs, rs[1:]) if b["s"] > a["e"])
        if gaps:
            print(f"queue {q}: {len(gaps)} gaps between consecutive kernels, {sum(gaps) / 1e3:.3f} ms in all, median "
                  f"{gaps[len(gaps) // 2]:.1f} us, {sum(1 for g in gaps if g > 20)} longer than 20 us (waits for another stream's event or for CU resources)")
    pts = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
    c, k, last = Counter(), 0, pts[0][0]
    for t, dlt in pts:
        c[k] += t - last
        last = t
        k += dlt
    print("kernels in flight -> ms: " + ", ".join(f"{k}: {v / 1e6:.2f}" for k, v in sorted(c.items())))
    fwd_end = max((r["e"] for r in step if "_fwd" in r["n"] or "k_loss" in r["n"] or "mse" in r["n"].lower()), default=t0)
    print(f"forward + loss end at {ms(fwd_end):.3f} ms; backward + Adam take {ms(t1) - ms(fwd_end):.3f} ms")


if __name__ == "__main__":
    main()
