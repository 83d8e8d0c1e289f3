"""UniPose building blocks as drop-in nn.Modules backed by the HIP kernels.

Parameters live in ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` containers under the SAME attribute
names as the reference, so ``state_dict()`` keys, shapes, ``load_state_dict`` of the authors'
checkpoints, ``.modules()`` walks (freeze_bn) and ``.parameters()`` (Adam) behave identically
(SURVEY §8b).  The containers' own ``forward`` is never called: every block's ``forward`` drives
the fused kernels through ``unipose_amd.ops`` on NHWC tensors.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


def _kaiming_all(module: nn.Module):
    """wasp.py:22-31,92-103 / decoder.py:58-64: kaiming_normal_ convs, BN weight 1 / bias 0."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


# (module-wide hooks registered through torch.nn.modules.module.register_module_forward_hook see every block's output too)
_GLOBAL_FWD_HOOKS = getattr(torch.nn.modules.module, "_global_forward_hooks", {})
_GLOBAL_PRE_HOOKS = getattr(torch.nn.modules.module, "_global_forward_pre_hooks", {})


class Bottleneck(nn.Module):
    """resnet.py:5-42.  Three fused conv+BN(+ReLU) stages; the residual add and the last ReLU ride in
    the third stage's apply pass (train) or convolution epilogue (inference)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, dilation=dilation, padding=dilation, bias=False)
        self.bn2 = BatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

    def forward(self, x):
        # identity blocks: the gradient of the skip connection joins the data gradient of conv1 inside its kernel
        grad = torch.is_grad_enabled()
        link = ops.GradLink() if self.downsample is None and x.requires_grad and grad else None
        # BatchNorm-backward reductions ride in the data gradient of the consuming convolution (ops.BnSlot): bn1 in conv2's,
        # bn2 in conv3's, and the previous block's bn3 in this block's conv1 launch when that launch already adds the skip
        # gradient (identity block) — then the block input has no other gradient path.  The slot of this block's output travels as
        # an attribute of the output tensor; nobody picks it up unless the next module is an identity Bottleneck.
        s_in = getattr(x, "_up_bnslot", None) if link is not None else None
        s1, s2, s3 = (ops.BnSlot(), ops.BnSlot(), ops.BnSlot()) if grad else (None, None, None)
        if self._forward_hooks or self._forward_pre_hooks or _GLOBAL_FWD_HOOKS or _GLOBAL_PRE_HOOKS:
            s_in = s3 = None          # a hook may use the block input / output elsewhere: keep autograd's generic path
        # projection blocks: x feeds conv1 AND the down-sampling convolution; the latter's data gradient (computed first: its
        # node is younger) rides into conv1's data-gradient epilogue instead of autograd adding the two (a 277 MB add at layer2.0)
        link_ds = ops.GradLink() if self.downsample is not None and x.requires_grad and grad else None
        y = ops.conv_bn_act(x, self.conv1, self.bn1, relu=True, link_in=link if link is not None else link_ds, slot_in=s_in,
                            slot_out=s1)
        y = ops.conv_bn_act(y, self.conv2, self.bn2, relu=True, slot_in=s1, slot_out=s2)
        if self.downsample is not None:
            x = ops.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False, link_dx=link_ds)
        out = ops.conv_bn_act(y, self.conv3, self.bn3, relu=True, residual=x, link_out=link, slot_in=s2, slot_out=s3)
        if s3 is not None and s3.y is not None:
            out._up_bnslot = s3
        return out


class ResNet(nn.Module):
    """ResNet-101 trunk with output stride 16/8 and the multi-grid last stage (resnet.py:44-136).
    Never downloads weights (the reference does so unconditionally, resnet.py:74-75,142): pass a
    checkpoint through ``load_state_dict`` / ``pretrained_path`` instead."""

    def __init__(self, layers=(3, 4, 23, 3), output_stride=16, BatchNorm=nn.BatchNorm2d, pretrained_path=None):
        super().__init__()
        if output_stride == 16:
            strides, dilations = (1, 2, 2, 1), (1, 1, 1, 2)
        elif output_stride == 8:
            strides, dilations = (1, 2, 1, 1), (1, 1, 2, 4)
        else:
            raise NotImplementedError
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(64, [1] * layers[0], strides[0], dilations[0], BatchNorm)
        self.layer2 = self._stage(128, [1] * layers[1], strides[1], dilations[1], BatchNorm)
        self.layer3 = self._stage(256, [1] * layers[2], strides[2], dilations[2], BatchNorm)
        self.layer4 = self._stage(512, [1, 2, 4], strides[3], dilations[3], BatchNorm)   # MG unit :94-111
        self._init_weight()
        if pretrained_path:
            self.load_pretrained(pretrained_path)

    def _stage(self, planes, grid, stride, dilation, BatchNorm):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 BatchNorm(planes * 4))
        blocks = [Bottleneck(self.inplanes, planes, stride, grid[0] * dilation, down, BatchNorm)]
        self.inplanes = planes * 4
        for g in grid[1:]:
            blocks.append(Bottleneck(self.inplanes, planes, 1, g * dilation, None, BatchNorm))
        return nn.Sequential(*blocks)

    def _init_weight(self):
        for m in self.modules():                          # resnet.py:126-136
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def load_pretrained(self, path):
        """Key-matched partial load of a torchvision resnet101 file (resnet.py:144-150)."""
        src = torch.load(path, map_location="cpu")
        own = self.state_dict()
        own.update({k: v for k, v in src.items() if k in own})
        self.load_state_dict(own)

    def forward(self, x_nhwc):
        x = ops.conv_bn_act(x_nhwc, self.conv1, self.bn1, relu=True)      # the 3-channel stem stays fp32 in every mode
        x = ops.MaxPool3s2.apply(x, ops.storage_dtype())                 # bf16 storage starts here (ops.set_conv_math)
        x = self.layer1(x)
        low = x
        x = self.layer4(self.layer3(self.layer2(x)))
        return x, low


def build_backbone(backbone, output_stride, BatchNorm):
    if backbone == "resnet":
        return ResNet((3, 4, 23, 3), output_stride, BatchNorm)
    raise NotImplementedError


class _AtrousModule(nn.Module):
    """wasp.py:6-31."""

    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size, stride=1, padding=padding, dilation=dilation,
                                     bias=False)
        self.bn = BatchNorm(planes)
        self.relu = nn.ReLU()
        _kaiming_all(self)

    def forward(self, x):
        return ops.conv_bn_act(x, self.atrous_conv, self.bn, relu=True)


class WASP(nn.Module):
    """Waterfall atrous spatial pooling (wasp.py:33-90; video flavour waspVideo.py:33-91, whose
    global-pool branch has no BatchNorm).  The four branch convolutions are a serial cascade."""

    def __init__(self, backbone, output_stride, BatchNorm, video=False):
        super().__init__()
        inplanes = 2048
        if output_stride == 16:
            dil = (24, 18, 12, 6)
        elif output_stride == 8:
            dil = (48, 36, 24, 12)
        else:
            raise NotImplementedError
        self.aspp1 = _AtrousModule(inplanes, 256, 1, 0, dil[0], BatchNorm)
        self.aspp2 = _AtrousModule(256, 256, 3, dil[1], dil[1], BatchNorm)
        self.aspp3 = _AtrousModule(256, 256, 3, dil[2], dil[2], BatchNorm)
        self.aspp4 = _AtrousModule(256, 256, 3, dil[3], dil[3], BatchNorm)
        gap = [nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(inplanes, 256, 1, stride=1, bias=False)]
        if not video:
            gap.append(nn.BatchNorm2d(256))
        gap.append(nn.ReLU())
        self.global_avg_pool = nn.Sequential(*gap)
        self.conv1 = nn.Conv2d(1280, 256, 1, bias=False)
        self.conv2 = nn.Conv2d(256, 256, 1, bias=False)
        self.bn1 = BatchNorm(256)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        self.video = video
        _kaiming_all(self)

    def forward(self, x):
        x1 = self.aspp1(x)
        x2 = self.aspp2(x1)
        x3 = self.aspp3(x2)
        x4 = self.aspp4(x3)
        # the SAME 1x1 weight twice on each branch, nothing in between (wasp.py:72-80): the four branches are stacked
        # along the batch axis, so the eight 1x1 convolutions are TWO implicit GEMMs over 4*B*H*W rows (2 + 2 + 2 launches
        # per step instead of 8 + 8 + 8, each a full-occupancy shape: M = 67 712 at B = 32)
        n = x1.shape[0]
        ys = ops.conv_bias_act(ops.conv_bias_act(torch.cat((x1, x2, x3, x4), 0), self.conv2), self.conv2)
        br = ops.SplitBatch.apply(ys, 4)          # (one gradient concatenation instead of four slice_backward nodes)
        g = ops.GlobalAvgPool.apply(x)
        if self.video:
            g = ops.conv_bias_act(g, self.global_avg_pool[1], relu=True)
        else:
            g = ops.conv_bn_act(g, self.global_avg_pool[1], self.global_avg_pool[2], relu=True)
        g = ops.Bilinear.apply(g, x4.shape[1], x4.shape[2])          # 1x1 -> HxW: a broadcast
        y = ops.ConcatC.apply(0, *br, g)
        y = ops.conv_bn_act(y, self.conv1, self.bn1, relu=True)
        return ops.dropout(y, self.dropout)


def build_wasp(backbone, output_stride, BatchNorm, video=False):
    return WASP(backbone, output_stride, BatchNorm, video)


class Decoder(nn.Module):
    """decoder.py:6-64 (conv2/bn2 are defined-but-unused there too: kept for state_dict parity).  `bbox=True` builds the
    output layer the reference keeps commented out next to the active one (decoder.py:31): five more channels (person
    centre and the four box corners) for the multi-person decode of utils/uniPose.py."""

    def __init__(self, dataset, num_classes, backbone, BatchNorm, bbox=False):
        super().__init__()
        if backbone != "resnet":
            raise NotImplementedError
        self.conv1 = nn.Conv2d(256, 48, 1, bias=False)
        self.bn1 = BatchNorm(48)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(2048, 256, 1, bias=False)
        self.bn2 = BatchNorm(256)
        self.last_conv = nn.Sequential(nn.Conv2d(304, 256, 3, stride=1, padding=1, bias=False), BatchNorm(256),
                                       nn.ReLU(), nn.Dropout(0.5),
                                       nn.Conv2d(256, 256, 3, stride=1, padding=1, bias=False), BatchNorm(256),
                                       nn.ReLU(), nn.Dropout(0.1),
                                       nn.Conv2d(256, num_classes + (6 if bbox else 1), 1, stride=1))
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        _kaiming_all(self)

    def forward(self, x, low):
        lc = self.last_conv
        low = ops.conv_bn_act(low, self.conv1, self.bn1, relu=True)
        low = ops.MaxPool3s2.apply(low)
        x = ops.Bilinear.apply(x, low.shape[1], low.shape[2])
        y = ops.ConcatC.apply(320, x, low)      # 256 + 48 = 304 real channels, zero-padded to a multiple of 32
        y = ops.dropout(ops.conv_bn_act(y, lc[0], lc[1], relu=True), lc[3])
        y = ops.dropout(ops.conv_bn_act(y, lc[4], lc[5], relu=True), lc[7])
        return ops.conv_bias_act(y, lc[8], out_f32=True)      # heat-maps leave as fp32 in every storage mode


def build_decoder(dataset, num_classes, backbone, BatchNorm, bbox=False):
    return Decoder(dataset, num_classes, backbone, BatchNorm, bbox)


class LSTM_0(nn.Module):
    """model/uniposeLSTM.py:9-24.  The three gate convolutions share their input, so they run as ONE
    convolution with the weights stacked along the output channels, then one gate kernel."""

    def __init__(self, inplanes, planes, kernel_size, padding):
        super().__init__()
        self.conv_g_lstm = nn.Conv2d(inplanes, planes, kernel_size, padding=padding)
        self.conv_i_lstm = nn.Conv2d(inplanes, planes, kernel_size, padding=padding)
        self.conv_o_lstm = nn.Conv2d(inplanes, planes, kernel_size, padding=padding)
        self.planes, self.pad = planes, padding

    def forward(self, z):
        cs = (self.conv_g_lstm, self.conv_i_lstm, self.conv_o_lstm)
        w = torch.cat([c.weight for c in cs], 0)
        b = torch.cat([c.bias for c in cs], 0)
        gates = ops.ConvBias.apply(z, w, b, ops.ConvCfg(1, self.pad, 1), False)
        return ops.LSTM0Gates.apply(gates, self.planes)


class LSTM(nn.Module):
    """model/uniposeLSTM.py:27-64.  The eight convolutions (x- and h-path of g,i,o,f) become ONE
    convolution over cat(x, h) with the weights stacked along both channel axes."""

    def __init__(self, inplanes, planes, kernel_size, padding):
        super().__init__()
        for n in "giof":
            setattr(self, f"conv_{n}x_lstm", nn.Conv2d(inplanes, planes, kernel_size, padding=padding))
        for n in "giof":
            setattr(self, f"conv_{n}h_lstm", nn.Conv2d(planes, planes, kernel_size, padding=padding))
        self.inplanes, self.planes, self.pad = inplanes, planes, padding

    def stacked(self):
        """(weight, bias) of the ONE gate convolution over cat(x, h): the eight weights stacked along both channel axes (the x
        part padded to the physical width of z), the x- and h-path biases summed.  Built with torch ops, so gradients flow back to
        the sixteen parameters through autograd."""
        order = "giof"
        padx = ops.rup4(self.inplanes) - self.inplanes
        padh = ops.rup4(self.planes) - self.planes
        wx = torch.cat([getattr(self, f"conv_{n}x_lstm").weight for n in order], 0)
        wh = torch.cat([getattr(self, f"conv_{n}h_lstm").weight for n in order], 0)
        w = torch.cat([nn.functional.pad(wx, (0, 0, 0, 0, 0, padx)), nn.functional.pad(wh, (0, 0, 0, 0, 0, padh))], 1)
        b = torch.cat([getattr(self, f"conv_{n}x_lstm").bias + getattr(self, f"conv_{n}h_lstm").bias
                       for n in order], 0)
        return w, b

    def forward(self, z, prev_hide, prev_cell, stacked=None):
        """stacked: the result of an earlier ``self.stacked()`` in the SAME autograd graph (the frames of one clip unroll share it:
        one cat / pad chain and ONE gradient split per clip instead of one per frame — 4 cat + 2 pad + 16 gradient adds per
        frame in the five-frame step of uniposeLSTM.py:116-133)."""
        w, b = stacked if stacked is not None else self.stacked()
        zh = ops.ConcatC.apply(0, z, prev_hide)
        gates = ops.ConvBias.apply(zh, w, b, ops.ConvCfg(1, self.pad, 1), False)
        return ops.LSTMGates.apply(gates, prev_cell, self.planes)
