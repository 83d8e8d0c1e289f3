"""Exact-fp32 direct-to-LDS kernels (unipose_amd/csrc/f32_glds.h) on the CPU emulator: the same kernel sources, A/B against the
register-staged igemm_kernel, and the BatchNorm-backward reduction fused into the data gradient against the separate pass."""
import pytest

import glds32_cases as g32

_id = lambda c: "n%d_c%d_%dx%d_k%d_r%d_d%d_t%d" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["dil"], c["tile_want"])


@pytest.mark.parametrize("case", g32.SMALL, ids=_id)
def test_glds32_kernel_matches_register_staged_kernel(emu_backend, case):
    """outputs, BatchNorm partials, data gradients and weight gradients, element for element, for both epilogue forms
    (LDS-transposed 16-byte stores / dword stores)"""
    g32.conv_ab(emu_backend, **case)


@pytest.mark.parametrize("case", g32.SPLIT, ids=_id)
def test_glds32_kernel_tail_split(emu_backend, case):
    """K-split tail tiles: same cut, same merge order as igemm_kernel -> equal results"""
    g32.conv_ab(emu_backend, **case)


@pytest.mark.parametrize("case", g32.WIDE, ids=_id)
def test_glds32_kernel_filters_of_more_than_32_taps(emu_backend, case):
    """the WIDE form (11x11 video head, 7x7): equal to the register-staged per-slice-tap kernel, element for element"""
    g32.conv_ab(emu_backend, **case)


@pytest.mark.parametrize("case", g32.BNRED, ids=_id)
def test_bn_backward_reduction_fused_into_data_gradient(emu_backend, case):
    g32.bnred_case(emu_backend, **case)


_wid = lambda c: "n%d_c%d_%dx%d_k%d_r%d_s%d_d%d_cus%d" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["stride"], c["dil"], c["cus"])


@pytest.mark.parametrize("case", g32.WGRAD, ids=_wid)
def test_wgrad_glds32_kernel_matches_register_staged_kernel(emu_backend, case):
    one_stage = g32.wgrad_ab(emu_backend, **case)
    assert one_stage == (case["cus"] > 0), "the case was meant for the other stage form"
