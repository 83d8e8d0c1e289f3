"""Exact-fp32 direct-to-LDS kernels (unipose_amd/csrc/f32_glds.h) on the MI355X, through the C ABI: A/B against the
register-staged igemm_kernel (element for element) at small and at the real layer geometries of BASELINE configs[1], and the
BatchNorm-backward reduction fused into the data gradient against the separate reduce pass."""
import pytest
import torch

import glds32_cases as g32

pytestmark = pytest.mark.gpu

_id = lambda c: "n%d_c%d_%dx%d_k%d_r%d_d%d_t%d" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["dil"], c["tile_want"])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", g32.SMALL + g32.SPLIT + g32.WIDE, ids=_id)
def test_glds32_kernel_matches_register_staged_kernel(dev, case):
    g32.conv_ab(dev, **case)


@pytest.mark.parametrize("case", g32.FULL, ids=_id)
def test_glds32_kernel_at_the_layer_geometries_of_the_headline(dev, case):
    g32.conv_ab(dev, **case)


@pytest.mark.parametrize("case", g32.BNRED + g32.BNRED_FULL, ids=_id)
def test_bn_backward_reduction_fused_into_data_gradient(dev, case):
    g32.bnred_case(dev, **case)


_wid = lambda c: "n%d_c%d_%dx%d_k%d_r%d_s%d_d%d_cus%d" % (c["n"], c["c"], c["h"], c["w"], c["k"], c["r"], c["stride"], c["dil"], c["cus"])


@pytest.mark.parametrize("case", g32.WGRAD, ids=_wid)
def test_wgrad_glds32_kernel_matches_register_staged_kernel(dev, case):
    g32.wgrad_ab(dev, **case)
