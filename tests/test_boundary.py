"""Drop-in boundary and C-ABI checks that need no GPU."""
import ctypes
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keys(golden_dir):
    with open(os.path.join(golden_dir, "g0_state_dict_keys.json")) as f:
        return json.load(f)


def test_state_dict_contract_image(golden_dir):
    from model.unipose import unipose
    ref = _keys(golden_dir)
    m = unipose("LSP", num_classes=14)
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == ref["unipose_K14"]                      # names, shapes, dtypes AND order
    assert [k for k, _ in m.named_parameters()] == ref["unipose_K14_params"]   # optimizer param order


def test_state_dict_contract_lstm(golden_dir):
    from model.uniposeLSTM import unipose, unipose_lstm
    from model.video_unipose import unipose as via_driver_name      # uniposeLSTM.py:26
    assert unipose is unipose_lstm is via_driver_name
    ref = _keys(golden_dir)
    m = unipose_lstm(num_classes=13)
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == ref["unipose_lstm_K13"]
    assert [k for k, _ in m.named_parameters()] == ref["unipose_lstm_K13_params"]


def test_constructor_contract():
    from model.unipose import unipose
    from model.uniposeLSTM import unipose_lstm
    with pytest.raises(NotImplementedError):
        unipose("LSP", backbone="drn")                    # backbone/__init__.py:6-7
    with pytest.raises(NotImplementedError):
        unipose("LSP", output_stride=32)                  # resnet.py:57-58
    m = unipose("LSP", backbone="resnet", output_stride=16, num_classes=14, sync_bn=True, freeze_bn=True, stride=8)
    assert all(not b.training for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))
    assert m.decoder.last_conv[8].out_channels == 15
    n1 = sum(p.numel() for p in m.get_1x_lr_params())
    n10 = sum(p.numel() for p in m.get_10x_lr_params())
    assert n1 + n10 == sum(p.numel() for p in m.parameters())
    unipose_lstm(backbone="resnet", output_stride=16, num_classes=13, sync_bn=True, freeze_bn=False, stride=8)


def test_partial_checkpoint_load_pattern():
    """unipose.py:78-90: filter a checkpoint by key, update, load."""
    from model.unipose import unipose
    from oracle import unipose_oracle as O
    m = unipose("LSP", num_classes=14)
    ckpt = {"state_dict": O.synth_state_dict(14, 9)}
    ckpt["state_dict"]["not.a.key"] = torch.zeros(1)
    own = m.state_dict()
    own.update({k: v for k, v in ckpt["state_dict"].items() if k in own})
    m.load_state_dict(own)
    assert torch.equal(m.backbone.layer3[7].conv2.weight, ckpt["state_dict"]["backbone.layer3.7.conv2.weight"])


def test_no_cpu_fallback():
    """The product path refuses host tensors instead of silently computing on the CPU."""
    from unipose_amd import _C, ops
    if _C._ALLOW_HOST_POINTERS:
        pytest.skip("emulation backend active in this process")
    with pytest.raises(_C.UniPoseHipError):
        ops.ToNHWC.apply(torch.zeros(1, 3, 8, 8))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "unipose_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(up_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding():
    from unipose_amd import _C
    assert _declared_symbols() == sorted(_C.SIGNATURES)


def test_hip_library_exports_every_declared_symbol():
    """The gfx950 build loads on a GPU-less host and exports the whole C ABI (no compute calls here)."""
    from unipose_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.up_abi_version.restype = ctypes.c_int
    assert lib.up_abi_version() == 10
    # argument validation happens before any launch, so it is testable without a device
    lib.up_last_error.restype = ctypes.c_char_p
    assert lib.up_conv2d_fwd(None, None, None, None, None, None) == -1
    assert b"null" in lib.up_last_error()


def test_conv_desc_validation_without_gpu():
    from unipose_amd import _C
    from unipose_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    lib.up_conv_stats_tiles.argtypes = [ctypes.POINTER(_C.ConvDesc)]
    lib.up_conv2d_bwd_weight_workspace.argtypes = [ctypes.POINTER(_C.ConvDesc)]
    lib.up_conv2d_bwd_weight_workspace.restype = ctypes.c_size_t
    d = _C.ConvDesc(N=32, H=23, W=23, C=256, Cp=256, ldx=256, K=256, R=3, S=3, stride=1, pad=18, dil=18, P=23, Q=23,
                    ldy=256, Kp=256)
    assert lib.up_conv_stats_tiles(ctypes.byref(d)) == (32 * 23 * 23 + 63) // 64     # 64x64 tiles at 23x23
    # whole split-K slabs + 64 bytes + the bias gradient's partial rows (one per 256 pixel rows, round 6: no float atomics)
    extra = 64 + ((32 * 23 * 23 + 255) // 256) * 256 * 4
    assert (lib.up_conv2d_bwd_weight_workspace(ctypes.byref(d)) - extra) % (256 * 9 * 256 * 4) == 0
    d.P = 22                                                                           # inconsistent geometry
    assert lib.up_conv2d_bwd_weight_workspace(ctypes.byref(d)) == 0


def test_deferred_wgrad_context_state():
    """ops.deferred_wgrad is a plain context: no nesting, and an exception inside leaves no state (and no half-summed
    gradients) behind."""
    from unipose_amd import ops
    with pytest.raises(RuntimeError):
        with ops.deferred_wgrad():
            with ops.deferred_wgrad():
                pass
    assert not ops._DEFER["on"] and not ops._DEFER["acc"] and not ops._DEFER["bn"]
    with pytest.raises(ZeroDivisionError):
        with ops.deferred_wgrad():
            ops._DEFER["acc"][1] = (None, None)
            1 / 0
    assert not ops._DEFER["on"] and not ops._DEFER["acc"] and not ops._DEFER["bn"]
    with ops.deferred_wgrad():
        pass
