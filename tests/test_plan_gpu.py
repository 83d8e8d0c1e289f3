"""up_unipose_forward (whole-graph inference entry, C ABI 9) on the MI355X: equal bits to the module's folded forward, the G1
reference golden through the plan, and what the single C call buys at small batches."""
import os
import time

import numpy as np
import pytest
import torch

import model_cases as mc
import plan_cases as pc
from oracle import unipose_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_plan_forward_equals_folded_module():
    pc.plan_case(DEV, K=14, B=2, size=368)


def test_plan_forward_output_stride_8_and_box_head():
    pc.plan_case(DEV, K=16, B=2, size=160, output_stride=8, bbox=True)


def test_plan_g1_eval_368_vs_reference_golden(golden_dir):
    """G1 (the genuine reference's eval forward, K=14, B=2, 368x368) through the ONE-call C entry: 1e-3 on the heat-maps
    (measured ~1e-6: the folded weights are rounded once), bit-exact joint argmax."""
    from unipose_amd.plan import UniPosePlan
    g = np.load(os.path.join(golden_dir, "g1_eval_368.npz"))
    K, wseed, xseed = (int(v) for v in g["meta"])
    m, _ = mc.build_image_model(K, wseed, DEV)
    m.eval()
    x = O.synth_input((2, 3, 368, 368), xseed).to(DEV)
    plan = UniPosePlan(m, 2, 368, 368)
    y = plan(x)
    e = O.max_rel(y.cpu(), g["out"])
    print(f"G1 through up_unipose_forward: max_rel {e:.2e}")
    assert e < 1e-3 and e < 2e-5
    assert np.array_equal(y.cpu().reshape(2, K + 1, -1).argmax(2).numpy(), g["argmax"])


def test_plan_latency_report():
    """Inference latency at B = 1 / 8 (368x368): the module's eager forward (one Python call per layer), the same forward as ONE
    hipGraph (unipose_amd.graph), and the ONE C call of the plan.  A report; the only assertion is that the plan is not slower
    than the eager module."""
    from unipose_amd.graph import GraphedForward
    from unipose_amd.plan import UniPosePlan
    m, _ = mc.build_image_model(16, 3, DEV)
    m.eval()

    def ms(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for B in (1, 8):
        x = O.synth_input((B, 3, 368, 368), 9).to(DEV)
        plan = UniPosePlan(m, B, 368, 368)
        out = torch.empty((B, 17, 46, 46), device=DEV)
        with torch.no_grad():
            eager = ms(lambda: m(x))
        g = GraphedForward(m, x)
        graphed = ms(lambda: g(x))
        planned = ms(lambda: plan(x, out=out))
        print(f"inference forward 368x368 B={B}: module eager {eager:.2f} ms, one hipGraph {graphed:.2f} ms, up_unipose_forward {planned:.2f} ms")
        assert planned < eager * 1.05
        g.close()
        plan.close()
