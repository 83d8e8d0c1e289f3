"""``unipose`` — drop-in for the reference's ``model.unipose.unipose`` (model/unipose.py:8-38):
same constructor, ``forward(input NCHW) -> (B, num_classes+1, H/8, W/8)``, same state_dict keys."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .modules import build_backbone, build_decoder, build_wasp


class unipose(nn.Module):
    def __init__(self, dataset, backbone="resnet", output_stride=16, num_classes=21, sync_bn=True, freeze_bn=False,
                 stride=8, bbox=False):
        """`bbox=True` (not a reference argument; default off) enables the variant the reference keeps as comments
        (model/unipose.py:34-35, decoder.py:31): num_classes+5+1 output channels, forward returns
        ``(x[:, :num_classes+1], x[:, num_classes+1:])`` = key-point maps and the five box maps."""
        super().__init__()
        self.stride = stride
        self.num_classes = num_classes
        self.bbox = bool(bbox)
        BatchNorm = nn.BatchNorm2d          # the reference ignores sync_bn the same way (model/unipose.py:14)
        self.pool_center = nn.AvgPool2d(kernel_size=9, stride=8, padding=1)   # unused, kept for parity (:18)
        self.backbone = build_backbone(backbone, output_stride, BatchNorm)
        self.wasp = build_wasp(backbone, output_stride, BatchNorm)
        self.decoder = build_decoder(dataset, num_classes, backbone, BatchNorm, self.bbox)
        if freeze_bn:
            self.freeze_bn()

    def forward(self, input):
        with ops.bn_counters(self):
            x = ops.ToNHWC.apply(input)
            x, low = self.backbone(x)
            x = self.wasp(x)
            x = self.decoder(x, low)
            if self.stride != 8:            # optional 8x bilinear up-sampling to the input size (:31-32)
                x = ops.Bilinear.apply(x, input.shape[2], input.shape[3])
            if self.bbox:                   # model/unipose.py:34-35
                y = ops.ToNCHW.apply(x, self.num_classes + 6)
                return y[:, 0:self.num_classes + 1, :, :], y[:, self.num_classes + 1:, :, :]
            return ops.ToNCHW.apply(x, self.num_classes + 1)

    # The reference versions reference an undefined SynchronizedBatchNorm2d (model/unipose.py:42,51,61)
    # and raise NameError; these do what they were meant to.
    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _params_of(self, roots):
        for root in roots:
            for m in root.modules():
                if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                    for p in m.parameters(recurse=False):
                        if p.requires_grad:
                            yield p

    def get_1x_lr_params(self):
        return self._params_of([self.backbone])

    def get_10x_lr_params(self):
        return self._params_of([self.wasp, self.decoder])
