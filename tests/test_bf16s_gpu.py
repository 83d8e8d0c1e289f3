"""bf16 STORAGE (BASELINE configs[4]: bf16, 736x736, batch 16/GPU) on the MI355X, through the C ABI: activations are bf16
tensors in HBM behind the fp32 stem, bf16 MFMA with fp32 accumulation, fp32 statistics / weights / weight gradients."""
import os

import numpy as np
import pytest
import torch

import bf16s_cases as bc
import glds_cases as gc
import model_cases as mc
from oracle import unipose_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

CONVS = [
    # n, c,    h,  w,   k, r, stride, pad, dil, bias      (SURVEY T1 shapes at 368 / 736 inputs)
    (2, 1024, 23, 23, 256, 1, 1, 0, 1, False),
    (2, 256, 23, 23, 1024, 1, 1, 0, 1, False),
    (2, 256, 46, 46, 256, 3, 1, 1, 1, False),
    (2, 512, 23, 23, 512, 3, 1, 4, 4, False),
    (2, 256, 23, 23, 256, 3, 1, 18, 18, False),
    (2, 128, 92, 92, 128, 3, 2, 1, 1, False),
    (2, 256, 92, 92, 512, 1, 2, 0, 1, False),
    (2, 64, 92, 92, 64, 3, 1, 1, 1, False),
    (2, 320, 46, 46, 256, 3, 1, 1, 1, False),
    (2, 256, 92, 92, 48, 1, 1, 0, 1, False),          # decoder.conv1: K = 48 -> 64 physical channels
    (2, 256, 46, 46, 17, 1, 1, 0, 1, True),           # the output layer: K = 17 -> 32, bias
    (16, 2048, 1, 1, 256, 1, 1, 0, 1, False),         # the global-pool branch: 16 rows
]


@pytest.mark.parametrize("cfg", CONVS, ids=lambda c: "c%d_%dx%d_k%d_r%d_s%d_d%d" % (c[1], c[2], c[3], c[4], c[5], c[6], c[8]))
def test_conv_bf16_storage(cfg):
    bc.conv_case(DEV, *cfg)


@pytest.mark.parametrize("cfg", [
    (4, 256, 23, 23, 48, 3, 1, 1, 1, True, False, True),
    (4, 256, 23, 23, 1024, 1, 1, 0, 1, True, True, True),
    (4, 128, 46, 46, 128, 3, 2, 1, 1, True, False, True),
    (2, 64, 46, 46, 64, 3, 1, 2, 2, True, True, False),
])
def test_conv_bn_bf16_storage(cfg):
    n, c, h, w, k, r, s, p, d, relu, residual, train = cfg
    bc.conv_bn_case(DEV, n, c, h, w, k, r, s, p, d, relu=relu, residual=residual, train=train)


def test_last_convolution_writes_fp32():
    bc.f32_out_case(DEV)
    bc.f32_out_case(DEV, n=1, c=96, h=7, w=6, k=22, r=3, pad=1, seed=9)
    bc.f32_out_case(DEV, n=4, c=256, h=92, w=92, k=17, r=1, pad=0, seed=11)      # the decoder's last layer at 736 / 8


def test_small_ops_bf16_storage():
    bc.small_ops_case(DEV)


@pytest.mark.parametrize("case", gc.SMALL + gc.FULL,
                         ids=lambda c: "n%d_c%d_%dx%d_k%d_r%d_d%d_t%d" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["dil"], c["tile_want"]))
def test_glds_kernel_matches_register_staged_kernel(case):
    """second-generation bf16-storage kernels (direct-to-LDS loads, tap skipping, tap-sorted rows, 16-byte stores, transposing
    LDS reads in the weight gradient) == the register-staged kernels, element for element, at small sizes and at the
    geometries of BASELINE configs[4]"""
    gc.conv_ab(DEV, **case)


@pytest.mark.parametrize("case", gc.BNRED + gc.BNRED_FULL,
                         ids=lambda c: "n%d_c%d_%dx%d_k%d_r%d_t%d_%s" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["tile_want"], "m" if c.get("mask_add") else ("a" if c.get("add") else "n")))
def test_bn_backward_reduction_fused_into_data_gradient_bf16(case):
    gc.bnred_case(DEV, **case)


@pytest.mark.parametrize("case", gc.BIG_SMALL + gc.BIG_FULL,
                         ids=lambda c: "n%d_c%d_%dx%d_k%d_r%d_d%d_s%d" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["dil"], c["stride"]))
def test_big_tile_kernel_matches_glds_kernel(case):
    """third-generation bf16-storage kernel (8 waves on (32 TM) x 256 tiles, bf16s_big.h) == igemm_glds_kernel, element for
    element (outputs, data gradients; BatchNorm partials merged to fp32 round-off), small and at the geometries of configs[4]"""
    gc.conv_ab(DEV, **case)


@pytest.mark.parametrize("case", gc.BIG_BNRED + gc.BIG_BNRED_FULL,
                         ids=lambda c: "n%d_c%d_%dx%d_k%d_r%d_%s" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], "m" if c.get("mask_add") else ("a" if c.get("add") else "n")))
def test_bn_backward_reduction_fused_into_big_tile_data_gradient(case):
    gc.bnred_case(DEV, **case)


def _golden_eval(golden_dir, name, size, B):
    from unipose_amd import ops
    g = np.load(os.path.join(golden_dir, name))
    K, wseed, xseed = (int(v) for v in g["meta"])
    m, _ = mc.build_image_model(K, wseed, DEV)
    m.eval()
    x = O.synth_input((B, 3, size, size), xseed).to(DEV)
    ops.set_conv_math("bf16s")
    try:
        with torch.no_grad():
            y = m(x)
    finally:
        ops.set_conv_math("f32")
    e = O.max_rel(y.cpu(), g["out"])
    _, _, idx = ops.heatmap_argmax(y)
    agree = float((idx.cpu().numpy() == g["argmax"]).mean())
    print(f"{name} in bf16 storage: max_rel {e:.3e}, argmax agreement {agree:.3f}")
    return e, agree


def test_g10_eval_736_bf16_storage(golden_dir):
    """configs[4]'s resolution against the genuine reference: SURVEY 8d allows 5e-2 of the map maximum; with the last convolution and
    the up-sampling in fp32 (round 3) the measured distance is 1.2e-2 with 94 % of the joint argmaxes identical: held to 2.5e-2 / 85 %."""
    e, agree = _golden_eval(golden_dir, "g10_eval_736.npz", 736, 1)
    assert e < 2.5e-2 and agree > 0.85


def test_g1_eval_368_bf16_storage(golden_dir):
    e, agree = _golden_eval(golden_dir, "g1_eval_368.npz", 368, 2)      # measured 1.1e-2 / 90 %
    assert e < 2.5e-2 and agree > 0.85


def test_736_b16_train_step_bf16_storage():
    """configs[4] at full size in bf16 storage: finite loss / gradients for every trained parameter, the saved activations
    really are bf16 (memory of the step), per-sample independence of the eval forward, argmax agrees with torch."""
    from unipose_amd import ops
    K, B, S = 16, 16, 736
    m, _ = mc.build_image_model(K, 3, DEV)
    x = O.synth_input((B, 3, S, S), 51).to(DEV)
    t = O.synth_input((B, K + 1, S // 8, S // 8), 52, "rand").to(DEV)
    ops.set_conv_math("bf16s")
    try:
        m.train()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = ops.mse_loss(m(x), t)
        held = torch.cuda.memory_allocated() - base            # activations saved for backward
        loss.backward()
        assert torch.isfinite(loss.detach()).item()
        for n, p in m.named_parameters():
            if n.startswith("decoder.conv2") or n.startswith("decoder.bn2"):
                assert p.grad is None
            else:
                assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), n
        m.eval()
        with torch.no_grad():
            y = m(x)
            parts = torch.cat([m(x[i:i + 4]) for i in range(0, B, 4)])
    finally:
        ops.set_conv_math("f32")
    print(f"activations held for backward: {held / 2 ** 30:.1f} GiB (bf16 storage; ~2x that in fp32)")
    assert held < 14 * 2 ** 30, held                 # fp32 storage holds ~18 GiB at this size
    assert y.dtype == torch.float32 and y.shape == (B, K + 1, S // 8, S // 8)
    assert O.max_rel(parts.cpu(), y.cpu()) < 1e-5
    _, _, idx = ops.heatmap_argmax(y)
    assert torch.equal(idx.cpu().long(), y.cpu().reshape(B, K + 1, -1).argmax(2))


def test_train_step_b8_within_twice_the_reference_bf16_yardstick(golden_dir):
    """G13: the genuine reference under bf16 autocast / with bf16-rounded stored tensors defines how far a correct bf16
    implementation lands from the fp32 gradients on this input; the HIP bf16-storage step must stay within twice that,
    parameter by parameter"""
    bc.model_train_yardstick_case(DEV, os.path.join(golden_dir, "g13_bf16_yardstick_b8_128.npz"))


@pytest.mark.parametrize("train", [False, True])
def test_lstm_bf16_storage(train):
    """round 5: UniPose-LSTM under bf16 storage (bf16 trunk, fp32 heat-map hand-over, fp32 ConvLSTM state and head), B=2, T=3 at
    128x128, batched frames: against the fp32 path with the head's measured amplification as the yardstick"""
    import model_cases as mc
    l16, l32 = mc.lstm_bf16s_case(torch.device("cuda:0"), B=2, size=128, T=3, train=train)
    print(f"UniPose-LSTM bf16 storage: loss {l16:.6f} vs fp32 {l32:.6f}")
