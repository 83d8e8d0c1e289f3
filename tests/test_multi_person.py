"""Multi-person decode of the optional box head (utils/uniPose.py:14-200, model/unipose.py:34-35).

G9 holds outputs (and exception types) of the reference's OWN `uniPose_kpts`, extracted from utils/uniPose.py and run
with scipy by tools/make_goldens.py.  The oracle restatement (plain numpy, no scipy) and the HIP path (emulator here,
MI355X with -m gpu) must reproduce them exactly: integer coordinates, same order, same error behaviour."""
import os

import numpy as np
import pytest
import torch

from oracle import unipose_oracle as O

CASES = ["lsp_one", "lsp_two", "mpii_two_noise", "mpii_noise_peaks_raises", "posetrack_three", "ntid_one_rect",
         "lsp_plateau_raises", "lsp_missing_corner_raises", "lsp_empty_box_raises", "lsp_nothing"]
ERRORS = {"IndexError": IndexError, "ValueError": ValueError}


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_multi_person.npz"))


def _check(fn, g9, name):
    maps, ds, err = g9[name + "_maps"], str(g9[name + "_dataset"]), str(g9[name + "_error"])
    if err:
        with pytest.raises(ERRORS[err]):
            fn(maps, ds)
        return
    got = fn(maps, ds)
    want = g9[name + "_kpts"].tolist()
    assert got == want
    if want:
        people = want[-1][0] + 1
        assert len(want) == people * 19            # 14 joints + centre + four corners per person (uniPose.py:161-175)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(g9, name):
    _check(O.unipose_kpts_multi, g9, name)


def test_goldens_cover_both_outcomes(g9):
    errs = [str(g9[n + "_error"]) for n in CASES]
    assert errs.count("") >= 5 and "IndexError" in errs and "ValueError" in errs
    assert len(g9["posetrack_three_kpts"]) == 57


def _hip(dev):
    from unipose_amd import ops
    return lambda maps, ds: ops.uniPose_kpts(torch.from_numpy(np.ascontiguousarray(maps)).to(dev), ds)


def _random_scenes(fn):
    """random boxes on random maps against the oracle (ties in the joint maps included: first maximum of the BOX)"""
    rng = np.random.default_rng(21)
    done = 0
    for trial in range(40):
        h, w = int(rng.integers(12, 40)), int(rng.integers(12, 40))
        maps = rng.standard_normal((1, 20, h, w)).astype(np.float32)
        maps[0, 15:20] = -1.0
        n = int(rng.integers(1, 4))
        ys = np.sort(rng.choice(h - 1, size=2 * n, replace=False))
        xs = np.sort(rng.choice(w - 1, size=2 * n, replace=False))
        for p in range(n):                                      # person p: rows ys[2p]..ys[2p+1]; peaks in list order
            y0, y1, x0, x1 = ys[2 * p], ys[2 * p + 1], xs[2 * p], xs[2 * p + 1]
            for ch, (yy, xx) in zip(range(15, 20), (((y0 + y1) // 2, (x0 + x1) // 2), (y0, x0), (y1, x0), (y0, x1), (y1, x1))):
                maps[0, ch, yy, xx] = 1.0 + p
        maps[0, 3] = np.round(maps[0, 3])                       # many exact ties in one joint channel
        try:
            want = O.unipose_kpts_multi(maps, "LSP")
        except (IndexError, ValueError) as e:
            with pytest.raises(type(e)):
                fn(maps, "LSP")
            continue
        assert fn(maps, "LSP") == want
        done += 1
    assert done >= 10


@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_emu(emu_backend, g9, name):
    _check(_hip(emu_backend), g9, name)


def test_hip_random_scenes_emu(emu_backend):
    _random_scenes(_hip(emu_backend))


def test_unknown_dataset_and_short_maps(emu_backend):
    fn = _hip(emu_backend)
    with pytest.raises(ValueError):
        fn(np.zeros((1, 20, 8, 8), np.float32), "COCO")
    with pytest.raises(IndexError):
        fn(np.zeros((1, 17, 8, 8), np.float32), "LSP")


def _bbox_model(dev):
    """model/unipose.py:34-35 + decoder.py:31: K+5+1 output channels, two returned tensors; the key-point half equals the
    oracle graph run with the same (wider) output layer."""
    from model.unipose import unipose
    K = 14
    sd = O.synth_state_dict(K, 7)
    g = torch.Generator().manual_seed(3)
    sd["decoder.last_conv.8.weight"] = torch.randn(K + 6, 256, 1, 1, generator=g) * 0.05
    sd["decoder.last_conv.8.bias"] = torch.randn(K + 6, generator=g) * 0.1
    m = unipose("LSP", num_classes=K, bbox=True)
    assert m.decoder.last_conv[8].out_channels == K + 6
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x = O.synth_input((1, 3, 64, 64), 4)
    with torch.no_grad():
        kp, box = m(x.to(dev))
        ref = O.unipose_forward(sd, x)
    assert kp.shape == (1, K + 1, 8, 8) and box.shape == (1, 5, 8, 8)
    got = torch.cat([kp, box], 1).cpu()
    assert O.max_rel(got, ref) < 1e-4
    plain = unipose("LSP", num_classes=K)
    assert plain.decoder.last_conv[8].out_channels == K + 1 and not plain.bbox


def test_bbox_head_emu(emu_backend):
    _bbox_model(emu_backend)


# ---- MI355X ----------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_gpu(g9, name):
    _check(_hip(torch.device("cuda:0")), g9, name)


@pytest.mark.gpu
def test_hip_random_scenes_gpu():
    _random_scenes(_hip(torch.device("cuda:0")))


@pytest.mark.gpu
def test_bbox_head_gpu():
    _bbox_model(torch.device("cuda:0"))
