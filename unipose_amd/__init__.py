"""unipose_amd — MI355X-native (gfx950) UniPose / UniPose-LSTM forward+backward path.

The compute lives in libunipose_hip.so (hand-written HIP kernels behind the C ABI of
include/unipose_hip.h); this package is the thin PyTorch-facing host layer that mirrors the
reference's nn.Module interface.
"""
__all__ = ["unipose", "unipose_lstm"]


def __getattr__(name):
    if name == "unipose":
        from .unipose import unipose
        return unipose
    if name == "unipose_lstm":
        from .uniposeLSTM import unipose_lstm
        return unipose_lstm
    raise AttributeError(name)
