"""Overlap, not just ordering: a kernel enqueued on the data-parallel trainer's COMMUNICATION stream right after
`ral_grad_bucket_wait(1)` (the place of the early bucket's all-reduce) - when does it run relative to the backward pass?
One GPU, one process; the stand-in for the collective is a device-to-device copy kernel of the bucket (what an RCCL
all-reduce on one rank degenerates to).  Run under rocprofv3 --kernel-trace and read the trace with the same script:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/bucket_trace -- python3 tools/diag/bucket_overlap.py run
    python tools/diag/bucket_overlap.py read gpurun_out/r03/bucket_trace"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    import torch
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.dp import HipEngineAdapter
    B = int(os.environ.get("B", 2048))
    m = RALENet("full", leads=1, L=512, max_batch=B, device="cuda:0", seed=3)
    x = torch.randn(B, 1, 512, device="cuda:0"); t = torch.randn(B, 1, 512, device="cuda:0")
    m.train()
    e = HipEngineAdapter(m)
    (o0, n0), (o1, n1) = e.grad_buckets()
    comm = e.bucket_stream()
    sink = torch.empty(n1, device="cuda:0")
    for _ in range(8):
        e.forward_begin(x); pred = e.forward_end(B)
        e.loss(pred, t, B)
        e.backward_begin()
        e.bucket_wait(1, comm)
        with torch.cuda.stream(comm):
            if os.environ.get("STANDIN") == "kernel":
                sink.add_(1.0)                             # an elementwise kernel that reads nothing of the step
            else:
                sink.copy_(m.eng.grads[o1:o1 + n1])      # <- where the early all-reduce kernel would be
        e.backward_end(B)
        torch.cuda.current_stream().wait_stream(comm)
        e.adam(1e-3)
    torch.cuda.synchronize()


def events():
    """the same question without a profiler in the way (rocprofv3 slows every launch down so much that the HOST becomes the
    pacing side and reaches the communication-stream launch late): hipEvents on the streams involved"""
    import ecg_denoise_amd           # (first: sets GPU_MAX_HW_QUEUES before the HIP runtime initialises)
    import torch
    from ecg_denoise_amd import RALENet
    from ecg_denoise_amd.dp import HipEngineAdapter
    B = int(os.environ.get("B", 2048))
    m = RALENet("full", leads=1, L=512, max_batch=B, device="cuda:0", seed=3)
    x = torch.randn(B, 1, 512, device="cuda:0"); t = torch.randn(B, 1, 512, device="cuda:0")
    m.train()
    e = HipEngineAdapter(m)
    (o0, n0), (o1, n1) = e.grad_buckets()
    comm = e.bucket_stream()
    sink = torch.empty(n1, device="cuda:0")
    ev = lambda: torch.cuda.Event(enable_timing=True)
    res = []
    for it in range(12):
        e.forward_begin(x); pred = e.forward_end(B)
        e.loss(pred, t, B)
        e0, e1, ec0, ec1 = ev(), ev(), ev(), ev()
        e0.record()
        e.backward_begin()
        e.bucket_wait(1, comm)
        with torch.cuda.stream(comm):
            ec0.record()
            sink.copy_(m.eng.grads[o1:o1 + n1])
            ec1.record()
        e.backward_end(B)
        e1.record()
        torch.cuda.current_stream().wait_stream(comm)
        e.adam(1e-3)
        torch.cuda.synchronize()
        if it >= 4:
            res.append((e0.elapsed_time(e1), e0.elapsed_time(ec0), e0.elapsed_time(ec1)))
    for tot, c0, c1 in res:
        print(f"backward pass {tot:.3f} ms; the communication-stream copy of bucket 1 ran from {c0:.3f} to {c1:.3f} ms = "
              f"{100 * c0 / tot:.0f} % .. {100 * c1 / tot:.0f} % of it")


def read(d):
    f = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if r["n"].startswith("k_adam")]
    for si in range(max(1, len(adam) - 4), len(adam)):
        step = rows[adam[si - 1] + 1: adam[si] + 1]
        bwd = [r for r in step if "_bwd" in r["n"] or r["n"].startswith("k_dw") or "bn8_bwd" in r["n"]]
        b0, b1 = min(r["s"] for r in bwd), max(r["e"] for r in bwd)
        # the stand-in: the one elementwise copy kernel of the step that is not part of the library
        cand = [r for r in step if r["s"] >= b0 and ("elementwise" in r["n"] or "copy" in r["n"].lower()) and not r["n"].startswith("k_")]
        if not cand:
            print(f"step {si}: stand-in kernel not found"); continue
        c = cand[0]
        under = [r for r in bwd if r["s"] >= c["e"]]
        print(f"step {si}: backward pass {(b1 - b0) / 1e6:.3f} ms; the communication-stream kernel ({c['n'][:40]}, queue {c.get('Queue_Id')}) "
              f"starts at {(c['s'] - b0) / 1e6:.3f} ms = {100.0 * (c['s'] - b0) / (b1 - b0):.0f} % of it and ends at {(c['e'] - b0) / 1e6:.3f} ms; "
              f"{len(under)} backward kernels ({sum(r['e'] - r['s'] for r in under) / 1e6:.3f} ms of kernel time) start after it has ENDED")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "events":
        events()
    else:
        read(sys.argv[2])
