"""ctypes binding of libralenet.so (C ABI declared in include/ralenet.h).

There is no CPU fallback: if the HIP library has not been built the import of the
product path fails loudly (`python __graft_entry__.py` or `make -C ecg_denoise_amd/csrc`
builds it for gfx950)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAL_LIB_PATH") or os.path.join(HERE, "libralenet.so")   # override: diagnostic builds only
CSRC = os.path.join(HERE, "csrc")

VARIANTS = {"nra": 0, "full": 1, "mlp": 2, "unet": 3, "acdae": 4, "danet": 5}
KIND_PARAM, KIND_STATE, KIND_COUNTER, KIND_INDEX = 0, 1, 2, 3


class RalConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("leads", C.c_int32), ("L", C.c_int32),
                ("max_batch", C.c_int32), ("train", C.c_int32)]


class RalError(RuntimeError):
    pass


def build(jobs=8, verbose=False):
    """Compile every HIP source for gfx950 into ecg_denoise_amd/libralenet.so (in-tree)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RalError("building libralenet.so failed")
    return LIB_PATH


_lib = None

_VP = C.c_void_p
_SIGS = {
    "ral_last_error": (C.c_char_p, []),
    "ral_layout_count": (C.c_int, [C.POINTER(RalConfig)]),
    "ral_layout_entry": (C.c_int, [C.POINTER(RalConfig), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ral_param_floats": (C.c_int64, [C.POINTER(RalConfig)]),
    "ral_state_floats": (C.c_int64, [C.POINTER(RalConfig)]),
    "ral_workspace_bytes": (C.c_int64, [C.POINTER(RalConfig)]),
    "ral_bn_sums_doubles": (C.c_int64, [C.POINTER(RalConfig)]),
    "ral_create": (C.c_int, [C.POINTER(RalConfig), C.POINTER(_VP)]),
    "ral_destroy": (C.c_int, [_VP]),
    "ral_bind": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ral_forward": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP]),
    "ral_forward_begin": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "ral_forward_end": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, _VP]),
    "ral_loss": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int64, _VP, _VP, _VP, _VP, _VP]),
    "ral_loss_flat": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int64, _VP, _VP, _VP, _VP, _VP]),
    "ral_loss_mean": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ral_loss_means": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ral_forward_loss_means": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ral_backward": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP]),
    "ral_backward_input": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP]),
    "ral_backward_begin": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "ral_backward_input_begin": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "ral_backward_input_end": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, _VP]),
    "ral_backward_end": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, _VP]),
    "ral_unet_stage_bn": (C.c_int, [C.c_int]),
    "ral_unet_forward_stage": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int64, _VP]),
    "ral_unet_forward_finish": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int64, _VP]),
    "ral_unet_backward_start": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, _VP]),
    "ral_unet_backward_stage": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int64, _VP]),
    "ral_unet_backward_finish": (C.c_int, [_VP, C.c_int, C.c_int64, _VP]),
    "ral_grad_bucket": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ral_grad_bucket_wait": (C.c_int, [_VP, C.c_int, _VP]),
    "ral_adam_step": (C.c_int, [_VP, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_float, _VP]),
    "ral_debug_tensor": (C.c_int, [_VP, C.c_char_p, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "ral_conv13_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "ral_conv13_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "ral_adam_flat": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                C.c_float, _VP]),
    "ral_prep_windows": (C.c_int, [_VP, _VP, C.c_int64, C.c_int, C.c_int, C.c_double, _VP, _VP, _VP, _VP]),
    "ral_stream_windows": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _VP, _VP, _VP]),
    "ral_stream_stitch": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "ral_attention_forward": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP]),
    "ral_attention_backward_scratch_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ral_attention_backward": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_int, C.c_int, C.c_int,
                                         C.c_int, _VP]),
    "ral_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "ral_global_option": (C.c_int, [C.c_char_p, C.c_longlong]),
    "ral_wavelet_denoise": (C.c_int, [_VP, _VP, C.c_int64, C.c_int, C.c_float, _VP]),
    "ral_profile_select": (C.c_int, [_VP, C.c_char_p]),
    "ral_profile_read": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ral_profile_timeline": (C.c_int, [_VP, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]),
    "ral_pe_table": (C.c_int, [C.POINTER(RalConfig), C.c_int, _VP, C.c_int64]),
}
EXPORTS = tuple(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RalError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                           "(run `python __graft_entry__.py` or `make -C ecg_denoise_amd/csrc`); "
                           "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def apply_options(spec):
    """`"attn_f16=0,mlp_fwd_w=2"` -> ral_global_option for each pair.  Process-wide library switches (kernel choices, grid
    caps); to be called before the library is first used.  Nothing calls this implicitly: the tests (tests/conftest.py,
    RAL_TEST_OPTIONS) and the diagnostic tools (bench.py --opt, RAL_TOOL_OPTIONS in tools/) do, explicitly."""
    for kv in (spec or "").split(","):
        if kv.strip():
            k, v = kv.split("=")
            check(lib().ral_global_option(k.strip().encode(), int(v)))


def check(rc):
    if rc != 0:
        raise RalError(lib().ral_last_error().decode())


def make_config(variant, leads, L, max_batch, train):
    return RalConfig(VARIANTS[variant], int(leads), int(L), int(max_batch), 1 if train else 0)


def layout(cfg):
    """-> list of dict(name, kind, offset, shape) in the reference state_dict order."""
    L = lib()
    n = L.ral_layout_count(C.byref(cfg))
    if n < 0:
        raise RalError(L.ral_last_error().decode())
    out = []
    name = C.create_string_buffer(256)
    kind, ndim, off = C.c_int32(), C.c_int32(), C.c_int64()
    shape = (C.c_int64 * 4)()
    for i in range(n):
        check(L.ral_layout_entry(C.byref(cfg), i, name, 256, C.byref(kind), C.byref(off), C.byref(ndim), shape))
        out.append({"name": name.value.decode(), "kind": kind.value, "offset": off.value,
                    "shape": tuple(shape[j] for j in range(ndim.value))})
    return out
