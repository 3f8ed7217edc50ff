"""Per-level timing of the attention operator (ral_attention_forward / ral_attention_backward) at the bench batch:
one line per (N, H, Len) with the algorithmic TFLOP/s (forward 4 N^2 C, backward 10 N^2 C per window, C = 4 H) and the
fraction of the fp32 vector/matrix peak.  Parity of each shape against an fp64 torch reference is checked first
(a fast wrong kernel is not a result).

    python tools/attn_bench.py [B] [reps]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ecg_denoise_amd import _lib
_lib.apply_options(os.environ.get("RAL_TOOL_OPTIONS", ""))   # library switches for this run (diagnostics), e.g. attn_f16=0

DEV = "cuda:0"
PEAK = 157.3
LEVELS = [(512, 2, 32, 2), (256, 4, 16, 4), (128, 8, 8, 4), (64, 16, 4, 4), (32, 32, 0, 4)]   # N, H, Len, blocks per step


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bias_full(table, Len, N):
    H = table.shape[1]
    b = torch.zeros(H, N, N, dtype=table.dtype)
    off = (N - Len) // 2
    i = torch.arange(Len)
    idx = i[:, None] - i[None, :] + Len - 1
    b[:, off:off + Len, off:off + Len] = table[idx].permute(2, 0, 1)
    return b


def check(N, H, Len):
    g = torch.Generator().manual_seed(N)
    B = 2
    qkv = torch.randn(B, 3 * H, N, 4, generator=g)
    qkv[:, :H] *= 0.5
    table = 0.5 * torch.randn(2 * Len - 1, H, generator=g) if Len else None
    do = torch.randn(B, H, N, 4, generator=g)
    q, k, v = (t.double().requires_grad_(True) for t in (qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]))
    tb = table.double().requires_grad_(True) if Len else None
    s = q @ k.transpose(-1, -2)
    if Len:
        s = s + bias_full(tb, Len, N)[None]
    o_ref = torch.softmax(s, -1) @ v
    lse_ref = torch.logsumexp(s, -1)
    gr = torch.autograd.grad((o_ref * do.double()).sum(), [q, k, v] + ([tb] if Len else []))
    qd, dod = qkv.to(DEV), do.to(DEV)
    td = table.to(DEV) if Len else None
    o = torch.empty(B, H, N, 4, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    dqkv = torch.empty_like(qd); gt = torch.zeros_like(td) if Len else None
    L = _lib.lib()
    _lib.check(L.ral_attention_forward(vp(qd), vp(o), vp(lse), vp(td), N, H, Len, B, stream()))
    ns = L.ral_attention_backward_scratch_floats(N, H, Len, int(bool(Len)), B)
    sc = torch.empty(max(ns, 1), device=DEV)
    _lib.check(L.ral_attention_backward(vp(qd), vp(o), vp(dod), vp(lse), vp(td), vp(gt), vp(dqkv), vp(sc), ns, N, H, Len, B, stream()))
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
    errs = {"o": rel(o, o_ref.detach()), "lse": rel(lse, lse_ref.detach()),
            "dq": rel(dqkv[:, :H], 0.5 * gr[0]), "dk": rel(dqkv[:, H:2 * H], gr[1]), "dv": rel(dqkv[:, 2 * H:], gr[2])}
    if Len:
        errs["dtable"] = rel(gt, gr[3])
    assert all(e < 2e-5 for e in errs.values()), (N, H, Len, errs)
    return max(errs.values())


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    L = _lib.lib()
    tot_f = tot_b = fl_f = fl_b = 0.0
    rows = []
    only = os.environ.get("ATTN_ONLY", "")                 # "fwd" / "bwd": time one direction only (PMC runs)
    levels = [LEVELS[int(i)] for i in os.environ["ATTN_LEVELS"].split(",")] if os.environ.get("ATTN_LEVELS") else LEVELS
    for N, H, Len, blocks in levels:
        if os.environ.get("ATTN_NOTABLE"):
            Len = 0
        err = 0.0 if os.environ.get("ATTN_NOCHECK") else check(N, H, Len)   # (ATTN_NOCHECK: diagnostic builds that are wrong on purpose)
        qkv = torch.randn(B, 3 * H, N, 4, device=DEV)
        do = torch.randn(B, H, N, 4, device=DEV)
        table = (0.1 * torch.randn(2 * Len - 1, H, device=DEV)) if Len else None
        gt = torch.zeros_like(table) if Len else None
        o = torch.empty(B, H, N, 4, device=DEV); lse = torch.empty(B, H, N, device=DEV); dqkv = torch.empty_like(qkv)
        fwd = lambda: _lib.check(L.ral_attention_forward(vp(qkv), vp(o), vp(lse), vp(table), N, H, Len, B, stream()))
        ns = L.ral_attention_backward_scratch_floats(N, H, Len, int(bool(Len)), B)
        sc = torch.empty(max(ns, 1), device=DEV)
        bwd = lambda: _lib.check(L.ral_attention_backward(vp(qkv), vp(o), vp(do), vp(lse), vp(table), vp(gt), vp(dqkv), vp(sc), ns, N, H, Len, B, stream()))
        out = {}
        for name, fn, mult in (("fwd", fwd, 4.0), ("bwd", bwd, 10.0)):
            if only and name != only:
                out[name] = (1e-9, 0.0)
                continue
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = mult * N * N * 4 * H * B
            out[name] = (ms, fl / ms / 1e9)
        tot_f += out["fwd"][0] * blocks; tot_b += out["bwd"][0] * blocks
        fl_f += 4.0 * N * N * 4 * H * B * blocks; fl_b += 10.0 * N * N * 4 * H * B * blocks
        rows.append({"N": N, "H": H, "Len": Len, "fwd_us": round(out["fwd"][0] * 1e3, 1), "fwd_TF": round(out["fwd"][1], 1),
                     "fwd_frac": round(out["fwd"][1] / PEAK, 3), "bwd_us": round(out["bwd"][0] * 1e3, 1),
                     "bwd_TF": round(out["bwd"][1], 1), "bwd_frac": round(out["bwd"][1] / PEAK, 3), "max_rel_err": float(f"{err:.1e}")})
        print(json.dumps(rows[-1]))
    print(json.dumps({"B": B, "step_fwd_ms": round(tot_f, 3), "step_bwd_ms": round(tot_b, 3),
                      "fwd_frac": round(fl_f / tot_f / 1e9 / PEAK, 4), "bwd_frac": round(fl_b / tot_b / 1e9 / PEAK, 4)}))


if __name__ == "__main__":
    main()
