"""CPU-side checks of the C ABI: the library loads, exports every symbol that
include/ralenet.h declares, and its host-only entry points (layout, PE tables, config
validation) agree with the reference contract.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ralenet_oracle as O
from ecg_denoise_amd import _lib
from ecg_denoise_amd.model import pe_table_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    hdr = open(os.path.join(ROOT, "include", "ralenet.h")).read()
    declared = set(re.findall(r"\b(ral_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"ral_handle"}
    L = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in ralenet.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


@pytest.mark.parametrize("variant", ["nra", "full", "mlp"])
@pytest.mark.parametrize("leads", [1, 2])
def test_layout_matches_reference_state_dict(variant, leads, golden_dir):
    cfg = _lib.make_config(variant, leads, 512, 4, 1)
    ent = _lib.layout(cfg)
    shapes = O.ralenet_param_shapes(variant, leads)
    mine = [(e["name"], tuple(e["shape"])) for e in ent if e["kind"] == _lib.KIND_PARAM]
    assert mine == [(k, tuple(s)) for k, s in shapes.items()]
    # golden key list comes from the reference's own named_parameters()
    if leads == 2:
        g = np.load(os.path.join(golden_dir, f"g3_{variant}_l2_L256.npz"))
        assert [str(k) for k in g["keys"]] == [n for n, _ in mine]
    segs = sorted((e["offset"], e["offset"] + int(np.prod(e["shape"]))) for e in ent if e["kind"] == _lib.KIND_PARAM)
    assert all(a[1] <= b[0] for a, b in zip(segs, segs[1:]))
    assert all(s[0] % 4 == 0 for s in segs)  # 16-byte aligned tensors
    assert segs[-1][1] <= _lib.lib().ral_param_floats(C.byref(cfg))
    names = [e["name"] for e in ent]
    assert "conv1.2.running_mean" in names and "conv1.2.num_batches_tracked" in names
    n_expected = {"nra": 303, "full": 311, "mlp": 293}[variant]  # reference state_dict sizes
    assert len(ent) == n_expected
    total = sum(int(np.prod(s)) for _, s in mine)
    if leads == 2:
        assert total == {"nra": 1086800, "full": 1087282, "mlp": 1087228}[variant]


def test_danet_layout_matches_reference_state_dict(golden_dir):
    """every state_dict entry of the reference's Seq2Seq2, in its order (the fixture's gradient keys are the reference's
    named_parameters(), its after_* keys the buffers); fcn2.* alias fcn1.*; bad configurations are rejected"""
    import danet_oracle as D
    cfg = _lib.make_config("danet", 2, 512, 4, 1)
    ent = _lib.layout(cfg)
    shapes = D.danet_state_shapes()
    assert [(e["name"], tuple(e["shape"])) for e in ent] == [(k, tuple(s)) for k, s in shapes.items()]
    g = np.load(os.path.join(golden_dir, "g3_danet_L512.npz"))
    ref_params = [k[5:] for k in g.files if k.startswith("grad_")]
    assert ref_params == [e["name"] for e in ent if e["kind"] == _lib.KIND_PARAM and ".dam.fcn2." not in e["name"]]
    assert sorted(k[6:] for k in g.files if k.startswith("after_")) == sorted(e["name"] for e in ent if e["kind"] != _lib.KIND_PARAM)
    off = {e["name"]: (e["kind"], e["offset"]) for e in ent}
    for k in off:
        if ".dam.fcn2." in k:
            assert off[k] == off[k.replace(".dam.fcn2.", ".dam.fcn1.")]
    assert sum(int(np.prod(e["shape"])) for e in ent if e["kind"] == _lib.KIND_PARAM and ".dam.fcn2." not in e["name"]) == 18009
    for leads, L in ((1, 512), (2, 520), (2, 16)):
        bad = _lib.make_config("danet", leads, L, 4, 1)
        assert _lib.lib().ral_layout_count(C.byref(bad)) == -1


def test_pe_table_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pe_tables.npz"))
    for lvl, Cc in enumerate(O.CHANNELS):
        p = pe_table_host(1024, lvl)
        ref = O.pe_table(1024 >> lvl, Cc).numpy()
        n = min(64, p.shape[0])
        assert np.abs(p[:n] - g[f"C{Cc}"][:n]).max() < 5e-7
        # libm vs torch differ by an ulp in pow/sin arguments: bounded by ulp(pos) ~ 6e-5 at pos~1000
        assert np.abs(p - ref).max() < 2e-4


@pytest.mark.parametrize("kw,msg", [
    (dict(L=500), "multiple of 16"), (dict(L=2048), "multiple of 16"), (dict(L=16), "at least 32"), (dict(leads=3), "leads"),
    (dict(max_batch=0), "max_batch"),
])
def test_config_validation(kw, msg):
    args = dict(variant="full", leads=2, L=512, max_batch=4, train=1)
    args.update(kw)
    cfg = _lib.make_config(**args)
    assert _lib.lib().ral_param_floats(C.byref(cfg)) < 0
    assert msg in _lib.lib().ral_last_error().decode()


def test_workspace_scales_with_batch():
    a = _lib.lib().ral_workspace_bytes(C.byref(_lib.make_config("full", 2, 512, 32, 1)))
    b = _lib.lib().ral_workspace_bytes(C.byref(_lib.make_config("full", 2, 512, 64, 1)))
    c = _lib.lib().ral_workspace_bytes(C.byref(_lib.make_config("full", 2, 512, 64, 0)))
    assert 1.9 < b / a < 2.1 and c < b / 3


def test_product_path_has_no_cpu_fallback():
    import torch
    from ecg_denoise_amd import RALENet, RalError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises((RalError, RuntimeError, AssertionError)):
        RALENet("full", device="cpu")
    # and the package never imports the oracle
    import ecg_denoise_amd
    pkg = os.path.dirname(ecg_denoise_amd.__file__)
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("no CPU fallback", ""), f


def test_global_option_validates_its_key_and_value():
    """`ral_global_option` is the only way to change a library switch (no environment variable does): unknown names and negative
    values are rejected with a message, known names are case-insensitive and accept an optional RAL_ prefix"""
    L = _lib.lib()
    assert L.ral_global_option(b"attn_f16", 1) == 0
    assert L.ral_global_option(b"RAL_ATTN_F16", 1) == 0
    assert L.ral_global_option(b"Grid_QkvB", 192) == 0
    assert L.ral_global_option(b"no_such_switch", 1) != 0 and b"unknown switch" in L.ral_last_error()
    assert L.ral_global_option(b"attn_f16", -1) != 0 and b"negative" in L.ral_last_error()
    assert L.ral_global_option(None, 1) != 0
    # grid / thread / split counts: 0 would be a launch without a workgroup
    for k in (b"unet_fwd_grid", b"grid_fwd", b"danet_grid_a", b"dw_ksplit_64", b"grid_qkvb"):
        assert L.ral_global_option(k, 0) != 0 and b"out of range" in L.ral_last_error(), k
    assert L.ral_global_option(b"mlp_hthreads", 384) != 0
    assert L.ral_global_option(b"grid_attnw", 0) == 0          # 0 = automatic for this one


def test_global_option_after_the_first_read_is_refused_not_ignored():
    """switches are latched at first use: once the library has read one, a different value is an error (the same value
    again is accepted)"""
    L = _lib.lib()
    cfg = _lib.make_config("full", 2, 512, 8, 1)
    first = L.ral_global_option(b"dw_sets", 6)                 # (an earlier test of this process may have sized a workspace)
    assert first == 0 or b"already read" in L.ral_last_error()
    assert L.ral_workspace_bytes(C.byref(cfg)) > 0             # plan_workspace reads DW_SETS
    if first == 0:
        assert L.ral_global_option(b"dw_sets", 6) == 0
    assert L.ral_global_option(b"dw_sets", 4) != 0 and b"already read" in L.ral_last_error()


def test_library_reads_only_the_two_documented_environment_variables():
    """the product build of the library has exactly two getenv call sites outside `#ifdef RAL_DIAG`: ral_env_int (RAL_LANES,
    RAL_NO_SIDE_STREAM) - and no other source file calls getenv at all"""
    import glob
    n_outside, in_diag = 0, False
    for f in sorted(glob.glob(os.path.join(ROOT, "ecg_denoise_amd", "csrc", "*.h*"))):
        depth_diag = None
        for line in open(f):
            s = line.strip()
            if s.startswith("#ifdef RAL_DIAG"):
                in_diag = True
            elif s.startswith("#endif") and in_diag:
                in_diag = False
            elif "getenv(" in line and not s.startswith("//") and not in_diag:
                n_outside += 1
                assert f.endswith("ral_api.hip"), (f, line)
    assert n_outside == 1                      # the one inside ral_env_int
    src = open(os.path.join(ROOT, "ecg_denoise_amd", "csrc", "ral_api.hip")).read()
    assert sorted(set(re.findall(r'ral_env_int\("(RAL_[A-Z_]+)"', src))) == ["RAL_LANES", "RAL_NO_SIDE_STREAM"]
