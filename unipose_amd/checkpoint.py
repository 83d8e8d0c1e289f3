"""Checkpoint interop: the reference's own checkpoint conventions around the drop-in modules.

* ``load_checkpoint``  — the key-filtered partial load of ``unipose.py:78-90`` / ``uniposeLSTM.py:77-93`` (a file
  written by ``save_checkpoint`` holds ``{'state_dict': model.state_dict()}``); additionally tolerates the
  ``module.`` prefix of nn.DataParallel files and reports what was skipped instead of dropping it silently.
* ``save_checkpoint``  — ``utils/utils.py:53-56`` (writes ``<filename>_best.pth.tar`` only when ``is_best``).
* ``load_resnet_pretrained`` — ``resnet.py:138-150`` for a torchvision ``resnet101-*.pth`` on disk (no download).
* ``fold_batchnorm``   — inference export: every convolution followed by a BatchNorm becomes weight + bias
  (``w * gamma / sqrt(var + eps)``, ``beta - mean * gamma / sqrt(var + eps)``), the arithmetic the inference fast path
  of ``ops.conv_bn_act`` applies in its convolution epilogue (``up_bn_eval_coeffs``).
* ``load_folded``      — the consumer of that export: turns a freshly built model into the BN-free inference network
  (every BatchNorm2d replaced by ``ops.FoldedBatchNorm``, convolutions gain a bias) whose ``state_dict`` IS the folded
  dict; its forward runs one ``up_conv2d_fwd`` per layer with the bias (+ residual) (+ ReLU) epilogue.

Host-side logic only: nothing here touches the device.
"""
from __future__ import annotations

import os
import re
from collections import OrderedDict
from typing import Dict, Iterable, List, NamedTuple, Optional, Tuple

import torch
import torch.nn as nn


class LoadReport(NamedTuple):
    loaded: List[str]            # keys copied into the model
    missing: List[str]           # model keys the file does not hold (they keep their current values)
    unexpected: List[str]        # file keys the model does not have (the reference drops these silently)
    skipped: List[str]           # file keys excluded by `skip_prefix` or by a shape mismatch


def _unwrap(obj) -> Dict[str, torch.Tensor]:
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]                                    # unipose.py:80
    if not isinstance(obj, dict):
        raise TypeError("checkpoint must be a state_dict or a dict holding one under 'state_dict'")
    return obj


def load_checkpoint(model: nn.Module, source, skip_prefix: Optional[Iterable[str]] = None,
                    map_location="cpu") -> LoadReport:
    """``source``: a path or an already loaded object.  Keys present in the model are copied, the rest is ignored
    exactly like the reference's loop; ``skip_prefix`` is the ``prefix`` filter of ``uniposeLSTM.py:81-88``.
    A tensor whose shape differs from the model's (e.g. a 15-channel LSP head into a 17-channel MPII model) is skipped and
    reported — the reference would raise from ``load_state_dict`` at that point."""
    src = _unwrap(torch.load(source, map_location=map_location) if isinstance(source, (str, bytes, os.PathLike)) or
                  hasattr(source, "read") else source)
    skip = tuple(skip_prefix) if skip_prefix else ()
    own = model.state_dict()
    loaded, unexpected, skipped = [], [], []
    for k, v in src.items():
        name = k[7:] if k.startswith("module.") and k not in own else k
        if name not in own:
            unexpected.append(k)
        elif (skip and name.startswith(skip)) or tuple(v.shape) != tuple(own[name].shape):
            skipped.append(k)
        else:
            own[name] = v
            loaded.append(name)
    model.load_state_dict(own)                                     # unipose.py:89-90
    done = set(loaded)
    return LoadReport(loaded, [k for k in own if k not in done], unexpected, skipped)


def save_checkpoint(state, is_best, filename: str = "checkpoint") -> Optional[str]:
    """utils/utils.py:53-56: called as ``save_checkpoint({'state_dict': model.state_dict()}, best, name)``."""
    if is_best:
        path = filename + "_best.pth.tar"
        torch.save(state, path)
        return path
    return None


def load_resnet_pretrained(backbone: nn.Module, path: str) -> LoadReport:
    """Key-matched load of a torchvision ResNet-101 file into the trunk (``fc.*`` has no counterpart and is ignored,
    resnet.py:144-150)."""
    return load_checkpoint(backbone, path)


_NUM = re.compile(r"^conv(\d*)$")


def conv_bn_pairs(model: nn.Module) -> List[Tuple[str, str]]:
    """(convolution, BatchNorm) module-name pairs as the forward passes apply them: ``convN``/``bnN`` attributes of one
    block, ``atrous_conv``/``bn`` (wasp.py:9-11), and neighbours inside an nn.Sequential (downsample, last_conv,
    global_avg_pool).  ``decoder.conv2``/``bn2`` are listed too although no forward uses them."""
    pairs = []
    for prefix, mod in model.named_modules():
        dot = prefix + "." if prefix else ""
        kids = list(mod.named_children())
        if isinstance(mod, nn.Sequential):
            for (n0, m0), (n1, m1) in zip(kids, kids[1:]):
                if isinstance(m0, nn.Conv2d) and isinstance(m1, nn.BatchNorm2d):
                    pairs.append((dot + n0, dot + n1))
            continue
        named = dict(kids)
        for n, m in kids:
            if not isinstance(m, nn.Conv2d):
                continue
            hit = _NUM.match(n)
            bn = "bn" + hit.group(1) if hit else ("bn" if n == "atrous_conv" else None)
            if bn and isinstance(named.get(bn), nn.BatchNorm2d) and named[bn].num_features == m.out_channels:
                pairs.append((dot + n, dot + bn))
    return pairs


def fold_batchnorm(model: nn.Module) -> "OrderedDict[str, torch.Tensor]":
    """state_dict of the inference network with every BatchNorm folded into its convolution: the BatchNorm entries
    disappear, the convolution gains ``.bias``.  Computed in float64 and rounded once."""
    sd = model.state_dict()
    mods = dict(model.named_modules())
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for conv, bn in conv_bn_pairs(model):
        m = mods[bn]
        g, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
        scale = g / torch.sqrt(sd[bn + ".running_var"].double() + m.eps)
        shift = b - sd[bn + ".running_mean"].double() * scale
        w = sd[conv + ".weight"].double() * scale.view(-1, 1, 1, 1)
        if conv + ".bias" in sd:
            shift = shift + sd[conv + ".bias"].double() * scale
        out[conv + ".weight"] = w.float()
        out[conv + ".bias"] = shift.float()
        for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out.pop(f"{bn}.{leaf}", None)
    return out


def load_folded(model: nn.Module, folded: Dict[str, torch.Tensor]) -> nn.Module:
    """Consume ``fold_batchnorm``'s output: `model` (a freshly built ``unipose`` / ``unipose_lstm``, any device) becomes the
    inference network without BatchNorm layers.  Every (convolution, BatchNorm) pair loses the BatchNorm (replaced by the
    parameter-less ``ops.FoldedBatchNorm`` marker, so attribute paths and Sequential indices stay valid) and the
    convolution gains ``.bias``; afterwards ``model.state_dict()`` has exactly the keys of `folded`, which is loaded
    strictly.  The result is inference-only (``ops.conv_bn_act`` refuses to differentiate through it)."""
    from . import ops
    mods = dict(model.named_modules())
    for conv, bn in conv_bn_pairs(model):
        c = mods[conv]
        if c.bias is None:
            c.bias = nn.Parameter(torch.zeros(c.out_channels, dtype=c.weight.dtype, device=c.weight.device))
        parent, _, leaf = bn.rpartition(".")
        (mods[parent] if parent else model)._modules[leaf] = ops.FoldedBatchNorm()
    missing, unexpected = model.load_state_dict(folded, strict=False)
    if missing or unexpected:
        raise KeyError(f"load_folded: missing {list(missing)[:4]} unexpected {list(unexpected)[:4]}")
    for p in model.parameters():
        p.requires_grad_(False)
    return model.eval()
