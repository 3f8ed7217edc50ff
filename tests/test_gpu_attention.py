"""The attention operator on its own (ral_attention_forward / ral_attention_backward through the C-ABI) against an fp64
torch autograd restatement of `softmax(q k^T + R-wave bias) v` (raletransformer.py:299-316; bias table :534-558), one case
per kernel the launchers can pick (default switches; `test_every_kernel_choice_of_the_launchers` re-runs the file with the
others):
  * N = 512 / 256, with and without a table: k_attn_bwd_mh (workgroup per head group, ONE sweep, every contraction on the
    f16 matrix cores), forward k_attn_fwd<2, 0, true, true> (f16 S tile)
  * N = 128 / 64 / 32, with and without a table: k_attn_bwd_m (one wave per head, rotated tiles with a table), forward
    k_attn_fwd_v (N = 64, 128) and k_attn_fwd_w (N = 32)
  * N = 48 (L = 768 windows) and N = 1024: the generic kernels of ral_bwd.hip / ral_fwd.hip (QT = 1; one head per item)
  * a table as wide as the window (N = 32, Len = 8 ... N = 64, Len = 32): every tile takes the table path
Batches that are not a multiple of anything (5, 3) and B = 700 (persistent workgroups take several items)."""
import ctypes as C

import pytest
import torch

from ecg_denoise_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bias_full(table, Len, N):
    H = table.shape[1]
    b = torch.zeros(H, N, N, dtype=table.dtype)
    off = (N - Len) // 2
    i = torch.arange(Len)
    idx = i[:, None] - i[None, :] + Len - 1
    b[:, off:off + Len, off:off + Len] = table[idx].permute(2, 0, 1)
    return b


def _case(N, H, Len, B, seed, scales=(1.0, 1.0, 1.0, 1.0)):
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(B, 3 * H, N, 4, generator=g)
    qkv[:, :H] *= 0.5 * scales[0]
    qkv[:, H:2 * H] *= scales[1]
    qkv[:, 2 * H:] *= scales[2]
    table = 0.5 * torch.randn(2 * Len - 1, H, generator=g) if Len else None
    do = torch.randn(B, H, N, 4, generator=g) * scales[3]
    q, k, v = (t.double().requires_grad_(True) for t in (qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]))
    tb = table.double().requires_grad_(True) if Len else None
    s = q @ k.transpose(-1, -2)
    if Len:
        s = s + _bias_full(tb, Len, N)[None]
    o_ref = torch.softmax(s, -1) @ v
    lse_ref = torch.logsumexp(s, -1)
    gr = torch.autograd.grad((o_ref * do.double()).sum(), [q, k, v] + ([tb] if Len else []))
    qd, dod = qkv.to(DEV), do.to(DEV)
    td = table.to(DEV) if Len else None
    o = torch.empty(B, H, N, 4, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    dqkv = torch.full_like(qd, float("nan"))           # every element must be written
    gt = torch.zeros_like(td) if Len else None
    L = _lib.lib()
    _lib.check(L.ral_attention_forward(_vp(qd), _vp(o), _vp(lse), _vp(td), N, H, Len, B, _stream()))
    ns = L.ral_attention_backward_scratch_floats(N, H, Len, int(bool(Len)), B)
    assert ns >= 0, L.ral_last_error()
    scratch = torch.empty(max(ns, 1), device=DEV)      # caller-owned: the entry point keeps no state
    _lib.check(L.ral_attention_backward(_vp(qd), _vp(o), _vp(dod), _vp(lse), _vp(td), _vp(gt), _vp(dqkv), _vp(scratch), ns,
                                        N, H, Len, B, _stream()))
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
    errs = {"o": rel(o, o_ref.detach()), "lse": rel(lse, lse_ref.detach()),
            "dq": rel(dqkv[:, :H], 0.5 * gr[0]),       # q = 0.5 (h Wq^T + b): the operator's dq carries the 0.5
            "dk": rel(dqkv[:, H:2 * H], gr[1]), "dv": rel(dqkv[:, 2 * H:], gr[2])}
    if Len:
        errs["dtable"] = rel(gt, gr[3])
    return errs


@pytest.mark.parametrize("N,H,Len,B", [
    (512, 2, 32, 3), (256, 4, 16, 5), (128, 8, 8, 5), (64, 16, 4, 5), (32, 32, 0, 5),   # the five levels of a 512-sample window
    (128, 8, 0, 3), (64, 16, 0, 5),                                                      # the same lengths without a table
    (48, 8, 8, 3), (1024, 2, 64, 1),                                                     # L = 768 and L = 1024 top levels
    (256, 4, 0, 3), (512, 2, 0, 2), (32, 8, 8, 5), (64, 2, 32, 3), (32, 2, 24, 3),                       # long windows without a table; a table as wide as the short window's tiles
])
def test_attention_operator_against_fp64(N, H, Len, B):
    errs = _case(N, H, Len, B, seed=N + Len)
    assert all(e == e and e < 2e-5 for e in errs.values()), errs    # fp32 tolerance; NaN = an element left unwritten


@pytest.mark.parametrize("N,H,Len", [(512, 2, 32), (64, 16, 4), (32, 32, 0)])
@pytest.mark.parametrize("scales", [
    (1e4, 1e-4, 1.0, 1.0), (1e-4, 1e4, 1.0, 1.0),      # q and k twenty-six binades apart (scores still O(1))
    (1e-3, 1e-3, 1.0, 1.0),                            # scores ~1e-6: a uniform softmax
    (1.0, 1.0, 1e-6, 1.0), (1.0, 1.0, 1e6, 1.0),       # v far below / above fp16's range
    (1.0, 1.0, 1.0, 1e-12), (1.0, 1.0, 1.0, 1e8),      # dO: a converged loss, an exploding one
    (30.0, 1.0, 3e-5, 1e-9),
])
def test_attention_operator_operand_ranges(N, H, Len, scales):
    """The S and dP tiles multiply fp16 PAIRS.  Every operand reaches them through a power of two chosen per head (dO, v:
    largest magnitude into [2^13, 2^14); q, k: balanced against each other so that q.k is unchanged), so magnitudes far
    outside fp16's range - where a plain conversion would flush to zero or saturate at 65504 - keep the fp32 tolerance."""
    errs = _case(N, H, Len, 3, seed=11, scales=scales)
    assert all(e == e and e < 2e-5 for e in errs.values()), (scales, errs)


def test_attention_operator_propagates_non_finite_values():
    """no clamp anywhere: a NaN in q / an Inf in dO reaches the outputs of its head (as in the fp32 reference) instead of
    being replaced by a finite number"""
    L = _lib.lib()
    for N, H, Len in ((64, 16, 4), (512, 2, 32)):
        B = 2
        qkv = torch.randn(B, 3 * H, N, 4, device=DEV)
        do = torch.randn(B, H, N, 4, device=DEV)
        qkv[1, 0, 5, 2] = float("nan")
        do[0, 1, 7, 1] = float("inf")
        table = torch.zeros(2 * Len - 1, H, device=DEV); gt = torch.zeros_like(table)
        o = torch.empty(B, H, N, 4, device=DEV); lse = torch.empty(B, H, N, device=DEV); dqkv = torch.zeros_like(qkv)
        _lib.check(L.ral_attention_forward(_vp(qkv), _vp(o), _vp(lse), _vp(table), N, H, Len, B, _stream()))
        ns = L.ral_attention_backward_scratch_floats(N, H, Len, 1, B)
        sc = torch.empty(max(ns, 1), device=DEV)
        _lib.check(L.ral_attention_backward(_vp(qkv), _vp(o), _vp(do), _vp(lse), _vp(table), _vp(gt), _vp(dqkv), _vp(sc), ns,
                                            N, H, Len, B, _stream()))
        torch.cuda.synchronize()
        assert not torch.isfinite(o[1, 0, 5]).all()                  # the NaN query's own output row
        assert not torch.isfinite(dqkv[1, 0]).all()                  # ... and the gradients of its head
        assert not torch.isfinite(dqkv[0, H + 1]).all()              # the Inf output gradient reaches dk of its head
        assert torch.isfinite(dqkv[0, 0]).all() and torch.isfinite(o[0]).all()   # other heads are untouched


@pytest.mark.parametrize("N,H,Len", [(64, 16, 4), (32, 32, 0)])
def test_attention_operator_many_windows(N, H, Len):
    """More windows than workgroup slots at the short levels: workgroups loop over several items, the R-wave table gradient
    is accumulated across all of them."""
    errs = _case(N, H, Len, 700, seed=7)
    assert all(e == e and e < 2e-5 for e in errs.values()), errs


def test_attention_backward_rejects_missing_scratch():
    """a shape whose kernels hand partial results from one launch to the next (here the per-workgroup R-wave table
    gradient rows of the short-window kernel) needs caller scratch; without it the call fails instead of allocating"""
    L = _lib.lib()
    N, H, Len, B = 64, 16, 4, 4
    ns = L.ral_attention_backward_scratch_floats(N, H, Len, 1, B)
    assert ns > 0
    qkv = torch.randn(B, 3 * H, N, 4, device=DEV)
    table = torch.zeros(2 * Len - 1, H, device=DEV); gt = torch.zeros_like(table)
    o = torch.empty(B, H, N, 4, device=DEV); lse = torch.empty(B, H, N, device=DEV); dqkv = torch.empty_like(qkv)
    _lib.check(L.ral_attention_forward(_vp(qkv), _vp(o), _vp(lse), _vp(table), N, H, Len, B, _stream()))
    rc = L.ral_attention_backward(_vp(qkv), _vp(o), _vp(o), _vp(lse), _vp(table), _vp(gt), _vp(dqkv), None, 0, N, H, Len, B, _stream())
    assert rc != 0 and b"scratch" in L.ral_last_error()
    small = torch.empty(ns - 1, device=DEV)
    rc = L.ral_attention_backward(_vp(qkv), _vp(o), _vp(o), _vp(lse), _vp(table), _vp(gt), _vp(dqkv), _vp(small), ns - 1, N, H, Len, B, _stream())
    assert rc != 0


@pytest.mark.parametrize("opts", [
    "attn_f16=0",                                              # strict mode: the two-sweep kernels with fp32-MFMA tiles
    "attn_bwd_m=0,attn_bwd_mh=0",                              # the one-sweep matrix-core backward switched off (two-sweep kernels take f16 callers too)
    "attn_bwd_m=0,attn_bwd_mh=0,attn_bwd_w=0,attn_fwd_w=0,attn_fwd_h=0",   # the workgroup / scalar-path kernels of ral_fwd.hip, ral_bwd.hip
    "attn_fwd_w=2",                                            # wave-autonomous forward at N = 64 and 128 too
    "attn_fwd_w=2,attn_f16=0",
])
def test_every_kernel_choice_of_the_launchers(opts):
    """The launchers pick kernels by shape and by process-wide switches (ral_global_option, read once): each alternative
    choice runs the fp64 comparison of this file in a process of its own (tests/conftest.py applies RAL_TEST_OPTIONS)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "against_fp64 or many_windows or operand_ranges or non_finite"],
                       env=dict(os.environ, RAL_TEST_OPTIONS=opts), cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
