"""Whole-network parity on a real MI355X: the drop-in modules against (a) the committed golden
vectors produced by the genuine reference and (b) the CPU oracle.  Tolerances: eval heat-maps within
1e-3 of the reference relative to the map maximum (BASELINE.json north_star) with bit-exact joint
argmax; train mode uses the fp64 yardstick of model_cases.yardstick (small-batch BN conditioning)."""
import os

import numpy as np
import pytest
import torch

import model_cases as mc
from oracle import unipose_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-3


def test_g1_eval_368_vs_reference_golden(golden_dir):
    from unipose_amd import ops
    g = np.load(os.path.join(golden_dir, "g1_eval_368.npz"))
    K, wseed, xseed = (int(v) for v in g["meta"])
    m, _ = mc.build_image_model(K, wseed, DEV)
    m.eval()
    x = O.synth_input((2, 3, 368, 368), xseed).to(DEV)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (2, K + 1, 46, 46)
    e = O.max_rel(y.cpu(), g["out"])
    assert e < TOL, e
    assert e < 2e-5, f"fp32 MFMA path should sit at the fp32 noise floor, got {e}"
    _, _, idx = ops.heatmap_argmax(y)
    assert np.array_equal(idx.cpu().numpy(), g["argmax"])            # bit-exact joint index
    # determinism: a second launch of the same graph is bitwise identical
    with torch.no_grad():
        y2 = m(x)
    assert torch.equal(y, y2)


def test_g2_eval_160_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_taps_160.npz"))
    K, wseed, xseed = (int(v) for v in g["meta"])
    m, _ = mc.build_image_model(K, wseed, DEV)
    m.eval()
    x = O.synth_input((2, 3, 160, 160), xseed).to(DEV)
    with torch.no_grad():
        y = m(x)
        m.stride = 1                                                   # optional 8x up-sampling (model/unipose.py:31-32)
        y1 = m(x)
    assert O.max_rel(y.cpu(), g["out"]) < TOL
    assert y1.shape == (2, K + 1, 160, 160)
    assert O.max_rel(y1.cpu()[:, ::4, ::2, ::2], g["out_stride1"]) < TOL


def test_g2_stage_taps_vs_reference_golden(golden_dir):
    """every stage of the 160x160 forward (stem, layer1-4, the WASP cascade x1-x4, the WASP output) against the reference's
    own intermediates, on the MI355X: forward hooks on the drop-in modules"""
    errs = mc.tap_case(DEV, os.path.join(golden_dir, "g2_taps_160.npz"), 160,
                       ("stem", "layer1", "layer2", "layer3", "layer4", "x1", "x2", "x3", "x4", "wasp"), sub=4, tol=TOL)
    assert max(errs.values()) < 1e-4, errs         # fp32 MFMA path: every stage at the fp32 noise floor


def test_g12_output_stride_8_vs_reference_golden(golden_dir):
    """output_stride = 8 (resnet.py:54-56: layer3 at stride 1 / dilation 2, multi-grid unit at dilation 4, 8, 16;
    wasp.py:41-42: dilations 48, 36, 24, 12 — larger than the 20x20 map): output within 1e-3, bit-exact argmax, stage taps"""
    errs = mc.tap_case(DEV, os.path.join(golden_dir, "g12_eval_os8_160.npz"), 160, ("layer2", "layer3", "layer4", "wasp"),
                       sub=8, output_stride=8, tol=TOL)
    assert max(errs.values()) < 1e-4, errs


@pytest.mark.parametrize("batch_frames", [False, True], ids=["per_frame", "batched_frames"])
def test_g5_lstm_368_vs_reference_golden(golden_dir, batch_frames):
    from model.uniposeLSTM import unipose_lstm
    g = np.load(os.path.join(golden_dir, "g5_lstm_368.npz"))
    K, wseed, xseed, cseed = (int(v) for v in g["meta"])
    m = unipose_lstm(num_classes=K)
    m.load_state_dict(O.synth_state_dict(K, wseed, lstm=True))
    m = m.to(DEV).eval()
    m.batch_frames = batch_frames                                       # trunk once on all five frames
    x = O.synth_input((1, 5, 3, 368, 368), xseed).to(DEV)
    cm = O.synth_input((1, 5, 1, 368, 368), cseed, "rand").to(DEV)
    heat = torch.zeros(K + 1, 46, 46, device=DEV)
    cell = torch.zeros(15, 46, 46, device=DEV)
    hide = torch.zeros(15, 46, 46, device=DEV)
    with torch.no_grad():
        for j in range(5):                                              # uniposeLSTM.py:124-128
            heat, cell, hide = m(x, cm, j, heat, hide, cell)
            assert O.max_rel(heat.cpu(), g[f"heat{j}"]) < TOL, j
            assert O.max_rel(cell.cpu(), g[f"cell{j}"]) < TOL, j
            assert O.max_rel(hide.cpu(), g[f"hide{j}"]) < TOL, j
            assert float(heat.min()) >= 0.0                             # final ReLU


@pytest.mark.parametrize("batch_frames", [False, True], ids=["per_frame", "batched_frames"])
def test_g15_lstm_train_368_vs_reference_golden(golden_dir, batch_frames):
    """UniPose-LSTM TRAINING (five-frame unroll, summed loss, one backward) against the genuine reference's own train step"""
    worst = mc.g15_case(DEV, os.path.join(golden_dir, "g15_lstm_train_368.npz"), batch_frames)
    print("g15 worst gradient distance / bound:", worst)


@pytest.mark.parametrize("stride,dilation,inplanes,planes,size", [(2, 1, 256, 128, 46), (1, 1, 64, 64, 23), (1, 2, 1024, 512, 23)])
def test_projection_block_hands_its_data_gradient_to_conv1_gpu(stride, dilation, inplanes, planes, size):
    mc.projection_block_ab_case(DEV, inplanes, planes, stride, dilation, B=4, size=size)


def test_projection_block_vs_float64_gpu():
    print(mc.projection_block_case(DEV, stride=2))          # the emulator's small case (few ReLU decisions near zero)


def test_tap_between_blocks_falls_back_to_the_separate_reduction_gpu():
    print(mc.tapped_block_output_case(DEV, planes=64, B=4, size=23))


def test_hooked_block_output_takes_the_separate_reduction_gpu():
    mc.hooked_block_output_case(DEV)
    mc.hooked_block_output_case(DEV, drop_tensor=True)


def test_bn_backward_reduction_fused_into_data_gradients_gpu():
    launches, worst = mc.fused_reduce_case(DEV, B=8, size=128)
    print("fused BatchNorm-backward reductions per step:", launches, "worst gradient distance fused vs separate:", worst)


@pytest.mark.parametrize("math,B,size", [("f32", 8, 128), ("bf16s", 4, 160)])
def test_bn_finalize_folded_whole_step_gpu(math, B, size):
    fwd, bwd = mc.bn_fold_step_case(DEV, B=B, size=size, math=math)
    print("folded finalize launches per step: forward", fwd, "backward", bwd)
    assert fwd >= 105 and bwd >= 80, (fwd, bwd)


def test_g4_train_128_loss_statistics_and_gradient_norms_vs_reference_golden(golden_dir):
    """Reference train step at 128x128, B=2, dropouts off: loss / output / gradients / running stats."""
    from unipose_amd import ops
    g = np.load(os.path.join(golden_dir, "g4_train_128.npz"))
    K, wseed, xseed, tseed = (int(v) for v in g["meta"])
    m, _ = mc.build_image_model(K, wseed, DEV)
    m.train()
    for d in (m.wasp.dropout, m.decoder.last_conv[3], m.decoder.last_conv[7]):
        d.p = 0.0
    x = O.synth_input((2, 3, 128, 128), xseed).to(DEV)
    t = O.synth_input((2, K + 1, 16, 16), tseed, "rand").to(DEV)
    y = m(x)
    loss = ops.mse_loss(y, t)
    loss.backward()
    assert O.max_rel(y.detach().cpu(), g["out"]) < TOL
    assert abs(float(loss.detach()) - float(g["loss"])) < TOL * abs(float(g["loss"]))
    sd = m.state_dict()
    p = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("rm/"):
            assert O.max_rel(sd[k[3:] + ".running_mean"].cpu(), g[k]) < TOL, k
        elif k.startswith("rv/"):
            assert O.max_rel(sd[k[3:] + ".running_var"].cpu(), g[k]) < TOL, k
    # (the per-gradient comparison of this B = 2 step — 128 samples per channel in the top stages, where one ReLU flip at
    #  round-off moves a gradient by percents — could only be held to 15 % and left in round 4: the reference-held gradient
    #  pins are G11 (B = 8, 128x128), G14 (B = 4, 368x368) and G15 (video model), each at 2x the reference's own yardstick)
    names = sorted(n for n, q in m.named_parameters())
    norms = np.array([p[n].grad.double().norm().item() if p[n].grad is not None else -1.0 for n in names])
    assert np.allclose(norms, g["grad_norms"], rtol=0.1, atol=1e-9)
    assert [n for n, q in m.named_parameters() if q.grad is None] == \
        ["decoder.conv2.weight", "decoder.bn2.weight", "decoder.bn2.bias"]


def test_train_step_vs_oracle_yardstick():
    mc.train_case(DEV, K=16, B=4, size=128)


def test_train_step_368_vs_oracle_yardstick():
    """The headline resolution (23x23 / 46x46 maps: the dilated WASP branches have all their live taps, unlike at 128x128), B = 2
    (the time of this test is the float64 oracle on the host; the genuine reference's B = 4 gradients at this size are G14), with
    the HIP path's ReLU decisions replayed in the oracle: every gradient inside the oracle's own fp32-vs-fp64 yardstick — the stem's
    weight gradient included, at the default floor.  It is a sum of 135 424 products per weight and image that cancel to a small
    value (the BatchNorm in front makes the gradient map sum to zero per channel); until round 4 the split-K merge added its 488
    slabs with plain fp32 additions and landed 2.3e-3 from float64 (floor 5e-3 here, ATen's CPU kernel: 7e-5).  With the compensated
    merge of round 5 (wgrad_reduce_kernel) it lands 1.3e-4 against the oracle's own 9.6e-5 (measured on the MI355X), so the special
    floor is gone."""
    mc.train_case(DEV, K=16, B=2, size=368)


def test_train_step_dropout_masks():
    mc.train_case(DEV, K=16, B=2, size=64, dropout_masks=True)


def test_eval_vs_oracle_odd_sizes():
    assert mc.eval_case(DEV, K=14, B=3, size=208, tol=1e-4) < 1e-4     # 13x13 top maps (odd), B odd
    assert mc.eval_case(DEV, K=16, B=1, size=368, tol=1e-4) < 1e-4     # BASELINE configs[0]


def test_lstm_train_bptt_vs_oracle():
    # two frames: the initial-state path and one recurrent step; three frames (every weight summed over three uses) run in the
    # batched-frames test below and, per frame, on the emulator (the time of these tests is the float64 oracle on the host)
    mc.lstm_case(DEV, size=64, T=2, B=2, train=True)


def test_lstm_train_bptt_batched_frames_vs_oracle():
    """trunk once on all T frames with per-frame BatchNorm statistics (ops.bn_groups): same yardsticks as the per-frame form"""
    mc.lstm_case(DEV, size=64, T=3, B=2, train=True, deferred=True, batch_frames=True)


def test_lstm_batch_generalisation():
    """The reference state is hard-wired to batch 1 (model/uniposeLSTM.py:99-104): a B=2 eval unroll must
    equal two independent B=1 unrolls."""
    from model.uniposeLSTM import unipose_lstm
    K, T, size = 13, 3, 64
    m = unipose_lstm(num_classes=K)
    m.load_state_dict(O.synth_state_dict(K, 4, lstm=True))
    m = m.to(DEV).eval()
    x = O.synth_input((2, T, 3, size, size), 21).to(DEV)
    cm = O.synth_input((2, T, 1, size, size), 22, "rand").to(DEV)

    def unroll(xs, cs):
        hs = size // 8
        heat = torch.zeros(K + 1, hs, hs, device=DEV)
        cell = torch.zeros(K + 2, hs, hs, device=DEV)
        hide = torch.zeros(K + 2, hs, hs, device=DEV)
        out = []
        with torch.no_grad():
            for j in range(T):
                heat, cell, hide = m(xs, cs, j, heat, hide, cell)
                out.append(heat)
        return out
    both = unroll(x, cm)
    for b in range(2):
        one = unroll(x[b:b + 1], cm[b:b + 1])
        for j in range(T):
            assert O.max_rel(both[j][b:b + 1].cpu(), one[j].cpu()) < 1e-5, (b, j)


def test_train_batch1_raises():
    m, _ = mc.build_image_model(14, 1, DEV)
    m.train()
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 64, 64, device=DEV))


def test_second_step_repack_and_counters():
    mc.second_step_case(DEV, B=2, size=64)


def test_full_size_step_properties():
    """BASELINE configs[1] shape (B=32, 368x368, K=16) — too big for the CPU oracle, so size-independent
    properties: finite loss/gradients, every trained parameter gets a gradient, per-sample independence of
    the eval forward (a batch of 32 equals 32/4 batches of 4), heat-map argmax agrees with torch."""
    from unipose_amd import ops
    K, B = 16, 32
    m, _ = mc.build_image_model(K, 3, DEV)
    x = O.synth_input((B, 3, 368, 368), 31).to(DEV)
    t = O.synth_input((B, K + 1, 46, 46), 32, "rand").to(DEV)
    m.train()
    loss = ops.mse_loss(m(x), t)
    loss.backward()
    assert torch.isfinite(loss.detach()).item()
    for n, p in m.named_parameters():
        if n.startswith("decoder.conv2") or n.startswith("decoder.bn2"):
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
    m.eval()
    with torch.no_grad():
        y = m(x)
        parts = torch.cat([m(x[i:i + 4]) for i in range(0, B, 4)])
    # not bitwise: which tiles are K-split (another summation order) depends on the launch's tile count
    assert O.max_rel(parts.cpu(), y.cpu()) < 1e-5
    _, _, idx = ops.heatmap_argmax(y)
    assert torch.equal(idx.cpu().long(), y.cpu().reshape(B, K + 1, -1).argmax(2))


@pytest.mark.parametrize("math,tol", [("bf16x3", 1e-3), ("bf16", 6e-2)])
def test_eval_golden_bf16_operand_modes(golden_dir, math, tol):
    """G1 again with the bf16-MFMA convolution kernels: split-bf16 must meet the fp32 bar (1e-3, bit-exact argmax);
    plain bf16 gets its own tolerance (~2.5e-2) and only reports argmax agreement."""
    from unipose_amd import ops
    g = np.load(os.path.join(golden_dir, "g1_eval_368.npz"))
    K, wseed, xseed = (int(v) for v in g["meta"])
    m, _ = mc.build_image_model(K, wseed, DEV)
    m.eval()
    x = O.synth_input((2, 3, 368, 368), xseed).to(DEV)
    ops.set_conv_math(math)
    try:
        with torch.no_grad():
            y = m(x)
    finally:
        ops.set_conv_math("f32")
    e = O.max_rel(y.cpu(), g["out"])
    _, _, idx = ops.heatmap_argmax(y)
    agree = float((idx.cpu().numpy() == g["argmax"]).mean())
    print(f"math={math}: max_rel {e:.3e}, argmax agreement {agree:.3f}")
    assert e < tol, e
    if math == "bf16x3":
        assert agree == 1.0


def test_lstm_any_num_classes_gpu():
    mc.lstm_case(DEV, K=15, size=64, T=2, B=2)
    mc.lstm_case(DEV, K=15, size=64, T=3, B=2, train=True, deferred=True, batch_frames=True)
