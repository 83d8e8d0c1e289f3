"""Checkpoint interop — host logic, no GPU.  The "authors' checkpoint" is synthesised from the key /
shape list dumped from the genuine reference (tests/golden/g0_state_dict_keys.json)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from unipose_amd import checkpoint as ckpt
from unipose_amd.unipose import unipose
from unipose_amd.uniposeLSTM import unipose as unipose_lstm


def _golden_state(golden_dir, which, seed=0):
    spec = json.load(open(os.path.join(golden_dir, "g0_state_dict_keys.json")))[which]
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape, _ in spec:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) + 0.5
        else:
            sd[k] = torch.randn(shape, generator=g)
    return sd


@pytest.fixture(scope="module")
def image_model():
    return unipose("LSP", num_classes=14, backbone="resnet", output_stride=16, sync_bn=True, freeze_bn=False, stride=8)


def test_reference_checkpoint_round_trip(tmp_path, golden_dir, image_model):
    which = "unipose_K14"
    sd = _golden_state(golden_dir, which)
    path = ckpt.save_checkpoint({"state_dict": sd}, True, str(tmp_path / "authors"))
    assert path.endswith("authors_best.pth.tar") and os.path.exists(path)
    assert ckpt.save_checkpoint({"state_dict": sd}, 0, str(tmp_path / "never")) is None
    assert not os.path.exists(str(tmp_path / "never_best.pth.tar"))
    rep = ckpt.load_checkpoint(image_model, path)
    assert not rep.missing and not rep.unexpected and not rep.skipped and len(rep.loaded) == len(sd)
    own = image_model.state_dict()
    assert all(torch.equal(own[k], sd[k]) for k in sd)
    # DataParallel prefix, a foreign key and a head of another dataset: loaded / reported / skipped
    other = {"module." + k: v for k, v in sd.items()}
    other["module.fc.weight"] = torch.zeros(3)
    head = [k for k in sd if k.endswith("last_conv.8.weight")][0]
    other["module." + head] = torch.zeros(17, 256, 1, 1)
    before = own[head].clone()
    rep = ckpt.load_checkpoint(image_model, other)
    assert rep.unexpected == ["module.fc.weight"] and rep.skipped == ["module." + head] and rep.missing == [head]
    assert torch.equal(image_model.state_dict()[head], before)


def test_torchvision_trunk_file(tmp_path, golden_dir, image_model):
    which = "unipose_K14"
    sd = _golden_state(golden_dir, which, seed=3)
    tv = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    path = str(tmp_path / "resnet101-synthetic.pth")
    torch.save(tv, path)
    rep = ckpt.load_resnet_pretrained(image_model.backbone, path)
    assert sorted(rep.unexpected) == ["fc.bias", "fc.weight"] and not rep.missing and not rep.skipped
    own = image_model.backbone.state_dict()
    assert all(torch.equal(own[k], v) for k, v in tv.items() if not k.startswith("fc."))
    image_model.backbone.load_pretrained(path)                      # the method the constructor argument uses


def test_lstm_prefix_filter(golden_dir):
    which = "unipose_lstm_K13"
    sd = _golden_state(golden_dir, which, seed=5)
    m = unipose_lstm(num_classes=13)
    keep = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("lstm")}
    rep = ckpt.load_checkpoint(m, {"state_dict": sd}, skip_prefix=("lstm",))
    assert rep.skipped and all(k.startswith("lstm") for k in rep.skipped) and sorted(rep.missing) == sorted(rep.skipped)
    own = m.state_dict()
    assert all(torch.equal(own[k], v) for k, v in keep.items())
    assert all(torch.equal(own[k], sd[k]) for k in rep.loaded)


def test_fold_batchnorm(golden_dir, image_model):
    which = "unipose_K14"
    image_model.load_state_dict(_golden_state(golden_dir, which, seed=9))
    pairs = ckpt.conv_bn_pairs(image_model)
    sd = image_model.state_dict()
    n_bn = sum(1 for k in sd if k.endswith("running_mean"))
    assert len(pairs) == n_bn and len({b for _, b in pairs}) == n_bn          # every BatchNorm found exactly once
    assert ("wasp.conv1", "wasp.bn1") in pairs and not any(c == "wasp.conv2" for c, _ in pairs)
    assert ("backbone.layer1.0.downsample.0", "backbone.layer1.0.downsample.1") in pairs
    folded = ckpt.fold_batchnorm(image_model)
    assert not any("running_" in k or ".bn" in k for k in folded)
    mods = dict(image_model.named_modules())
    g = torch.Generator().manual_seed(1)
    for conv, bn in (pairs[0], ("backbone.layer4.2.conv2", "backbone.layer4.2.bn2"), ("wasp.aspp3.atrous_conv", "wasp.aspp3.bn"),
                     ("decoder.last_conv.4", "decoder.last_conv.5")):
        c = mods[conv]
        x = torch.randn((1, c.in_channels, 9, 9), generator=g)
        kw = dict(stride=c.stride, padding=c.padding, dilation=c.dilation)
        ref = F.batch_norm(F.conv2d(x, sd[conv + ".weight"], None, **kw), sd[bn + ".running_mean"],
                           sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"], False, 0.0, mods[bn].eps)
        got = F.conv2d(x, folded[conv + ".weight"], folded[conv + ".bias"], **kw)
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max())), (conv, float((got - ref).abs().max()))


def _folded_vs_unfolded(dev, size, K=14, B=1):
    """fold_batchnorm() -> load_folded(): the BN-free network (one conv + bias (+ residual) (+ ReLU) kernel per layer) against
    the unfolded eval network on the same device; returns (max_rel, folded output, unfolded output)."""
    from oracle import unipose_oracle as O
    sd = O.synth_state_dict(K, 1)
    m = unipose("LSP", num_classes=K)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    folded = ckpt.fold_batchnorm(m)
    f = ckpt.load_folded(unipose("LSP", num_classes=K).to(dev), folded)
    assert sorted(f.state_dict().keys()) == sorted(folded.keys())
    assert not any(isinstance(q, torch.nn.BatchNorm2d) for q in f.modules())
    x = O.synth_input((B, 3, size, size), 11).to(dev)
    with torch.no_grad():
        y0, y1 = m(x), f(x)
    with pytest.raises(NotImplementedError):
        f(x.clone().requires_grad_(True))                 # inference export only
    return O.max_rel(y1.cpu(), y0.cpu()), y1, y0


def test_load_folded_runs_on_the_kernels_emu(emu_backend):
    e, _, _ = _folded_vs_unfolded(emu_backend, 64)
    assert e < 1e-5, e


@pytest.mark.gpu
def test_load_folded_gpu(golden_dir):
    """N1 on the device: folded eval == unfolded eval (<= 1e-5) and == the reference's G1 golden (<= 1e-3, argmax exact)."""
    import numpy as np
    from oracle import unipose_oracle as O
    from unipose_amd import ops
    dev = torch.device("cuda:0")
    e, y1, _ = _folded_vs_unfolded(dev, 368, K=14, B=2)
    assert e < 1e-5, e
    g = np.load(os.path.join(golden_dir, "g1_eval_368.npz"))
    assert tuple(int(v) for v in g["meta"]) == (14, 1, 11)
    assert O.max_rel(y1.cpu(), g["out"]) < 1e-3
    _, _, idx = ops.heatmap_argmax(y1)
    assert np.array_equal(idx.cpu().numpy(), g["argmax"])
