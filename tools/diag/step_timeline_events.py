"""A profiler-free timeline of one training step: hipEvent pairs around every profiled launch
(`ral_profile_select(h, "*")` / `ral_profile_timeline`), so the host is NOT paced by rocprofv3.  Per stream: busy time,
idle time inside the stream's own span, the longest gaps and what ran on the other streams meanwhile; and how long 1, 2,
3, 4 streams were busy at once.

    python tools/diag/step_timeline_events.py [out.txt]        (RAL_LANES / RAL_NO_SIDE_STREAM select the schedule)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ecg_denoise_amd import RALENet, _lib

KINDS = ["qkv_fwd", "attn_fwd", "mlp_fwd", "mlp_bwd", "attn_bwd", "qkv_bwd", "dw", "resample_fwd", "resample_bwd", "stem"]


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    B, L = 2048, 512
    m = RALENet("full", leads=1, L=L, max_batch=B, train=True, device="cuda:0", seed=2023)
    x = torch.randn(B, 1, L, device="cuda:0"); t = torch.randn(B, 1, L, device="cuda:0")
    m.train()
    lib, h = _lib.lib(), m.eng.h
    for _ in range(5):
        m.train_step(x, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m.train_step(x, t)
    e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) / 10
    _lib.check(lib.ral_profile_select(h, b"*"))
    m.train_step(x, t)                       # warm: creates the events
    torch.cuda.synchronize()
    _lib.check(lib.ral_profile_select(h, b"*"))
    e0.record()
    m.train_step(x, t)
    e1.record(); torch.cuda.synchronize()
    traced = e0.elapsed_time(e1)
    cap = 4096
    rows = (C.c_double * (4 * cap))()
    n = C.c_int64()
    _lib.check(lib.ral_profile_timeline(h, rows, cap, C.byref(n)))
    _lib.check(lib.ral_profile_select(h, b""))
    ev = [(int(rows[4 * i]), int(rows[4 * i + 1]), rows[4 * i + 2], rows[4 * i + 3]) for i in range(n.value)]
    t_end = max(e[3] for e in ev)
    p = lambda *a: print(*a, file=out)
    p(f"step timeline from hipEvents (no profiler): RA-LENet 'full' 2048 x 1 x 512, lanes={os.environ.get('RAL_LANES', '2')}, "
      f"side streams={'off' if os.environ.get('RAL_NO_SIDE_STREAM') else 'on'}")
    p(f"step without events {plain:.3f} ms; the step that recorded {n.value} event pairs {traced:.3f} ms; "
      f"first launch -> last kernel end {t_end:.3f} ms (stem / loss / Adam kernels are outside the recorded kinds)")
    streams = sorted(set(e[1] for e in ev))
    for s in streams:
        mine = sorted([e for e in ev if e[1] == s], key=lambda e: e[2])
        busy = sum(e[3] - e[2] for e in mine)
        span = mine[-1][3] - mine[0][2]
        kinds = {}
        for e in mine:
            kinds[KINDS[e[0]]] = kinds.get(KINDS[e[0]], 0.0) + e[3] - e[2]
        p(f"\nstream {s}: {len(mine)} launches, first start {mine[0][2]:.3f} ms, last end {mine[-1][3]:.3f} ms, busy {busy:.3f} ms, "
          f"idle inside its span {span - busy:.3f} ms ({100 * (span - busy) / span:.1f} %)")
        p("   busy by kind: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])))
        gaps = sorted(((mine[i + 1][2] - mine[i][3], mine[i], mine[i + 1]) for i in range(len(mine) - 1)), key=lambda g: -g[0])
        for g, a, b in gaps[:6]:
            if g < 0.02:
                break
            others = [KINDS[e[0]] + f"@s{e[1]}" for e in ev if e[1] != s and e[2] < b[2] and e[3] > a[3]]
            p(f"   gap {g * 1e3:7.1f} us at {a[3]:.3f} ms between {KINDS[a[0]]} and {KINDS[b[0]]}; meanwhile on other streams: "
              + (", ".join(sorted(set(others))) or "nothing"))
    # concurrency histogram
    pts = sorted([(e[2], 1) for e in ev] + [(e[3], -1) for e in ev])
    hist, cur, last = {}, 0, pts[0][0]
    for tt, d in pts:
        hist[cur] = hist.get(cur, 0.0) + tt - last
        cur += d; last = tt
    p("\nstreams busy at once (ms): " + ", ".join(f"{k}: {v:.3f}" for k, v in sorted(hist.items())))
    p(f"sum of all kernel durations {sum(e[3] - e[2] for e in ev):.3f} ms in a {t_end:.3f} ms span: "
      f"{sum(e[3] - e[2] for e in ev) / t_end:.2f} kernels in flight on average")


if __name__ == "__main__":
    main()
