"""ecg_denoise_amd — MI355X-native RA-LENet / U-Net ECG-denoising hot path.

Python here is host glue only: device memory and streams come from PyTorch-ROCm, all
arithmetic runs in hand-written HIP kernels behind the C ABI of include/ralenet.h."""
from ._lib import RalError, build  # noqa: F401
from .baselines import wavelet_denoise  # noqa: F401
import os as _os

# The step uses four HIP streams of its own (two micro-batch chains and their weight-gradient side streams); a data-parallel
# rank adds a communication stream (and RCCL its own).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and a queue runs its packets in order, so with the default the communication stream SHARES a queue with a chain and the
# "early" gradient bucket's all-reduce starts only when that chain has drained - after the backward pass instead of
# half-way through it (measured, tools/diag/bucket_overlap.py: 100 % vs 51 % of the backward pass).  The variable is read
# when the HIP runtime initialises, i.e. it must be set before the first GPU call of the process: importing this package
# first does that; a caller that initialises the GPU earlier sets it itself.
_HWQ_SET_LATE = False
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    try:
        import torch as _torch
        _HWQ_SET_LATE = bool(_torch.cuda.is_initialized())   # the runtime is up already: the variable can no longer take effect
    except Exception:                                       # pragma: no cover
        pass
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def hw_queues_set_too_late():
    """True when this package was imported AFTER the process had initialised the GPU with GPU_MAX_HW_QUEUES unset: the
    data-parallel trainer's early gradient bucket then shares a hardware queue with a compute chain and its all-reduce
    runs after the backward pass instead of under it (dp.HipEngineAdapter warns once)."""
    return _HWQ_SET_LATE


from .model import ACDAE, DANet, NewRALE, RALENet, UNet, ralenet  # noqa: F401
