"""Whole-network parity on a GPU-less box: drop-in modules -> host ops -> HIP kernel sources running on
the fiber emulator, against the CPU oracle.  Tiny inputs (the emulator is ~1e5x slower than the GPU)."""
import model_cases as mc


def test_eval_forward_emu(emu_backend):
    assert mc.eval_case(emu_backend, size=64) < 1e-4


def test_eval_forward_stride1_emu(emu_backend):
    mc.eval_case(emu_backend, size=32, B=1, stride=1)


def test_train_dropout_masks_emu(emu_backend):
    mc.train_case(emu_backend, size=32, dropout_masks=True)


def test_bn_backward_reduction_fused_into_data_gradients_emu(emu_backend):
    mc.fused_reduce_case(emu_backend, size=32)


def test_tap_between_blocks_falls_back_to_the_separate_reduction_emu(emu_backend):
    print(mc.tapped_block_output_case(emu_backend))


import pytest


@pytest.mark.parametrize("stride,dilation", [(2, 1), (1, 1), (1, 2)])
def test_projection_block_hands_its_data_gradient_to_conv1_emu(emu_backend, stride, dilation):
    print(mc.projection_block_case(emu_backend, stride=stride, dilation=dilation))


def test_projection_block_hand_over_equals_autograd_add_emu(emu_backend):
    mc.projection_block_ab_case(emu_backend, 64, 32, 2, 1, B=2, size=10)


def test_second_step_repack_and_counters_emu(emu_backend):
    mc.second_step_case(emu_backend)


def test_lstm_eval_emu(emu_backend):
    mc.lstm_case(emu_backend, size=32, T=3, B=1)


def test_lstm_train_bptt_emu(emu_backend):
    mc.lstm_case(emu_backend, size=32, T=2, B=2, train=True)


def test_lstm_train_bptt_deferred_wgrad_emu(emu_backend):
    # three frames: every weight and every BatchNorm affine pair is summed over three uses by the library
    mc.lstm_case(emu_backend, size=32, T=3, B=2, train=True, deferred=True)


def test_lstm_eval_batched_frames_emu(emu_backend):
    # trunk once on all T frames: eval mode (running statistics), per-frame outputs unchanged
    mc.lstm_case(emu_backend, size=32, T=3, B=1, batch_frames=True)


def test_lstm_train_bptt_batched_frames_emu(emu_backend):
    # trunk once on all T frames, BatchNorm statistics per frame (ops.bn_groups), one backward through everything
    mc.lstm_case(emu_backend, size=32, T=3, B=2, train=True, deferred=True, batch_frames=True)


def test_output_stride_8_vs_reference_golden_emu(emu_backend, golden_dir):
    """output_stride = 8 through the drop-in modules (layer3 at dilation 2, multi-grid 4 / 8 / 16, WASP dilations 48 / 36 /
    24 / 12 on a 20x20 map) against the genuine reference's output, argmax and stage taps (G12)"""
    import os
    errs = mc.tap_case(emu_backend, os.path.join(golden_dir, "g12_eval_os8_160.npz"), 160, ("layer2", "layer3", "layer4", "wasp"),
                       sub=8, output_stride=8)
    assert max(errs.values()) < 1e-4, errs


def test_lstm_bf16_storage_eval_emu(emu_backend):
    """round 5: the video model no longer refuses ops.set_conv_math("bf16s") (bf16 trunk, fp32 ConvLSTM state and head)"""
    mc.lstm_bf16s_case(emu_backend, train=False)


def test_lstm_bf16_storage_train_emu(emu_backend):
    mc.lstm_bf16s_case(emu_backend, train=True)


def test_hooked_block_output_takes_the_separate_reduction_emu(emu_backend):
    mc.hooked_block_output_case(emu_backend)
    mc.hooked_block_output_case(emu_backend, drop_tensor=True)


def test_bn_finalize_folded_whole_step(emu_backend):
    fwd, bwd = mc.bn_fold_step_case(emu_backend, B=3, size=64)
    print("folded finalize launches per step: forward", fwd, "backward", bwd)
    # 64x64 inputs, B = 3: the stem, layer1 and layer2.0 see more than 256 rows per channel (the rest: float64 statistics, no fold)
    assert fwd >= 12 and bwd >= 60, (fwd, bwd)


def test_lstm_any_num_classes(emu_backend):
    """the reference's video model accepts any num_classes (model/uniposeLSTM.py:68-91); K + 1 a multiple of 4 (15, 19,
    23 ...) leaves no spare pad channel for the centre map, so the hand-over tensor grows to the next multiple of 4."""
    mc.lstm_case(emu_backend, K=15, size=32, T=2, B=1)
    mc.lstm_case(emu_backend, K=15, size=32, T=3, B=2, train=True, deferred=True, batch_frames=True)
    mc.lstm_case(emu_backend, K=19, size=32, T=2, B=1, batch_frames=True)


def test_lstm_whole_clip_unroll_serves_only_its_own_states_emu(emu_backend):
    mc.lstm_unroll_fallback_case(emu_backend)
