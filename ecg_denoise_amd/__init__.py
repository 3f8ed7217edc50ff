"""ecg_denoise_amd — MI355X-native RA-LENet / U-Net ECG-denoising hot path.

Python here is host glue only: device memory and streams come from PyTorch-ROCm, all
arithmetic runs in hand-written HIP kernels behind the C ABI of include/ralenet.h."""
from ._lib import RalError, build  # noqa: F401
from .baselines import wavelet_denoise  # noqa: F401
from .model import ACDAE, DANet, NewRALE, RALENet, UNet, ralenet  # noqa: F401
