"""up_unipose_forward (whole-graph inference entry, C ABI 9) on the CPU emulator: equal bits to the module's folded forward."""
import plan_cases as pc


def test_plan_forward_equals_folded_module_emu(emu_backend):
    pc.plan_case(emu_backend, K=14, B=1, size=64)


def test_plan_forward_size_not_a_multiple_of_8_emu(emu_backend):
    """52 x 52 gives ceil(52 / 8) = 7 x 7 heat-maps; the plan used to allocate 6 x 6 and write past the end"""
    out = pc.plan_case(emu_backend, K=14, B=1, size=52)
    assert out.shape[-2:] == (7, 7)


def test_plan_forward_output_stride_8_and_box_head_emu(emu_backend):
    pc.plan_case(emu_backend, K=16, B=2, size=48, output_stride=8, bbox=True)


def test_plan_argument_checks(emu_backend):
    import ctypes as C
    from unipose_amd import _C
    from unipose_amd.plan import _Config
    L = _C.lib()
    plan = C.c_void_p()
    assert L.up_unipose_plan_create(C.byref(_Config(1, 64, 64, 32, 15)), C.byref(plan)) != 0      # output stride 32: not built
    assert b"output stride" in L.up_last_error()
    assert L.up_unipose_plan_create(C.byref(_Config(1, 64, 64, 16, 15)), C.byref(plan)) == 0
    n = L.up_unipose_plan_num_convs(plan)
    names = [L.up_unipose_plan_conv_name(plan, i).decode() for i in range(n)]
    assert n == 116 and names[0] == "backbone.conv1" and names[-1] == "decoder.last_conv.8"       # 115 parameters, wasp.conv2 twice
    assert names.count("wasp.conv2") == 2 and len(set(names)) == 115
    assert L.up_unipose_plan_workspace(plan) > 0
    import torch
    buf = torch.zeros(1024)
    assert L.up_unipose_forward(plan, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 1 << 40, 0) != 0   # weights never set
    assert b"never set" in L.up_last_error()
    L.up_unipose_plan_destroy(plan)
