"""unipose_amd — MI355X-native (gfx950) UniPose / UniPose-LSTM forward+backward path.

The compute lives in libunipose_hip.so (hand-written HIP kernels behind the C ABI of
include/unipose_hip.h); this package is the thin PyTorch-facing host layer that mirrors the
reference's nn.Module interface.
"""
import os as _os

# The weight-gradient side stream (unipose_amd.ops) needs its own hardware queue: ROCm multiplexes HIP streams
# onto GPU_MAX_HW_QUEUES (default 4) queues round-robin, and once RCCL has created its streams the side stream
# can end up sharing the main stream's queue, which serialises the two (measured +11 ms per step).  Only
# effective if the HIP runtime has not been initialised yet, i.e. import this package before the first CUDA call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__all__ = ["unipose", "unipose_lstm"]


def __getattr__(name):
    if name == "unipose":
        from .unipose import unipose
        return unipose
    if name == "unipose_lstm":
        from .uniposeLSTM import unipose_lstm
        return unipose_lstm
    raise AttributeError(name)
