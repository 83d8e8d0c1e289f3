"""Whole-graph inference entry (up_unipose_forward, unipose_amd/plan.py) against the drop-in module: shared by the emulator and
the GPU tests.  The plan issues the launches of the module's folded inference forward, so the comparison is for EQUAL bits;
the module itself is pinned to the reference by the G1 / G2 / G12 goldens."""
import copy

import torch

from oracle import unipose_oracle as O

import model_cases as mc


def plan_case(dev, K=14, B=1, size=64, wseed=1, xseed=5, output_stride=16, bbox=False):
    from unipose_amd import checkpoint
    from unipose_amd.plan import UniPosePlan
    kw = {}
    if output_stride != 16:
        kw["output_stride"] = output_stride
    if bbox:
        kw["bbox"] = True
    m = mc.skeleton("image", K, **kw)
    sd = O.synth_state_dict(K, wseed)
    if bbox:        # the box head's five extra output channels have no synthetic entry: keep the constructor's
        own = m.state_dict()
        sd = {k: (v if v.shape == own[k].shape else own[k]) for k, v in sd.items()}
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = O.synth_input((B, 3, size, size), xseed).to(dev)
    plan = UniPosePlan(m, B, size, size)
    got = plan(x)
    folded = checkpoint.load_folded(copy.deepcopy(m), checkpoint.fold_batchnorm(m))
    with torch.no_grad():
        ref = folded(x)
        unfolded = m(x)
    pairs = list(zip(got, ref)) if bbox else [(got, ref)]
    for g, r in pairs:
        assert g.shape == r.shape
        assert torch.equal(g.cpu(), r.cpu()), float((g - r).abs().max())
    first = torch.cat(got, 1) if bbox else got
    assert O.max_rel(first.cpu(), (torch.cat(unfolded, 1) if bbox else unfolded).cpu()) < 1e-4      # folding itself: one rounding per weight
    # a second call on the same workspace, and wrong shapes / unset weights fail loudly
    again = plan(x)
    assert torch.equal((torch.cat(again, 1) if bbox else again).cpu(), first.cpu())
    try:
        plan(x[:, :, :size - 8])
        raise AssertionError("a mis-shaped input must be refused")
    except ValueError:
        pass
    oh = (size - 1) // 8 + 1
    for bad in (torch.empty((B, first.shape[1], oh - 1, oh), device=dev), torch.empty((B, first.shape[1], oh, oh), device=dev).double(),
                torch.empty((B, first.shape[1], oh, 2 * oh), device=dev)[..., ::2]):
        try:                            # a caller-supplied `out` of the wrong shape / dtype / layout was written past its end
            plan(x, out=bad)
            raise AssertionError("a mis-shaped / mis-typed / strided `out` must be refused")
        except ValueError:
            pass
    own = torch.empty((B, first.shape[1], oh, oh), device=dev)
    assert plan(x, out=own) is own or bbox
    plan.close()
    return first
