"""``unipose`` (alias ``unipose_lstm``) — drop-in for the reference's video model
(model/uniposeLSTM.py:67-147): ResNet-101 + WASP(video) + decoder trunk per frame, ConvLSTM state,
five-convolution head.  The state is generalised from the reference's hard-wired batch 1
(:99-104) to (B, 15, H/8, W/8); B=1 reproduces the reference numerics."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .modules import LSTM, LSTM_0, build_backbone, build_decoder, build_wasp


class _AddCenter(torch.autograd.Function):
    """cat(heat, AvgPool2d(9,8,1)(centermap)) (model/uniposeLSTM.py:114-116): the heat tensor already has
    a spare pad channel, so the pooled centre map is written in place of it."""

    @staticmethod
    def forward(ctx, heat_nhwc, center_nchw, k):
        n, h, w, cp = heat_nhwc.shape
        if k < cp:
            z = heat_nhwc.clone()
        else:       # (k + 1) % 4 == 0 — K = 15, 19, 23 ...: no spare pad channel, the hand-over tensor grows to the next multiple of 4
            z = torch.zeros((n, h, w, ops.rup4(k + 1)), dtype=heat_nhwc.dtype, device=heat_nhwc.device)
            ops._copy2d(heat_nhwc, ops._nhwc_ok(heat_nhwc), 0, z, z.shape[3], 0, n * h * w, cp)
        ctx.cp = cp
        ops.avgpool9s8_into(center_nchw, z, k)
        return z

    @staticmethod
    def backward(ctx, dz):
        if dz.shape[3] != ctx.cp:          # the widened hand-over: the trunk's own channels
            dz = dz[..., :ctx.cp].contiguous()
        return dz, None, None     # channels >= k are ignored upstream (the producer masks them)


class unipose(nn.Module):
    def __init__(self, backbone="resnet", output_stride=16, num_classes=21, sync_bn=True, freeze_bn=False, stride=8):
        super().__init__()
        self.stride = stride
        self.num_classes = num_classes
        self.BatchNorm = nn.BatchNorm2d
        self.backbone = build_backbone(backbone, output_stride, self.BatchNorm)
        self.wasp = build_wasp(backbone, output_stride, self.BatchNorm, video=True)
        self.decoder = build_decoder("Penn_Action", num_classes, backbone, self.BatchNorm)
        c = num_classes + 2                     # joints + background + centre map (15 for Penn Action)
        self.lstm_0 = LSTM_0(c, c, 3, 1)
        self.lstm = LSTM(c, c, 3, 1)
        self.conv1 = nn.Conv2d(c, 128, kernel_size=11, padding=5)
        self.conv2 = nn.Conv2d(128, 128, kernel_size=11, padding=5)
        self.conv3 = nn.Conv2d(128, 128, kernel_size=11, padding=5)
        self.conv4 = nn.Conv2d(128, 128, kernel_size=1, padding=0)
        self.conv5 = nn.Conv2d(128, num_classes + 1, kernel_size=1, padding=0)
        self.pool_center = nn.AvgPool2d(kernel_size=9, stride=8, padding=1)
        # batch_frames: run the trunk ONCE, at iter == 0, on all T frames of the clip batch (the call receives the whole clip
        # tensor anyway) and serve the later calls from that result.  Every frame keeps its own BatchNorm batch statistics
        # (ops.bn_groups) and the running statistics receive the same T updates in the same order, so outputs, gradients and
        # buffers are those of T separate trunk calls — but every convolution runs on B*T images instead of B (B = 8:
        # 268-tile launches become 1300-tile launches).  Off by default: a caller that only ever asks for iter == 0 would pay
        # for T frames and see T running-statistics updates.  unipose_amd.trainer.VideoTrainer and bench.py switch it on.
        self.batch_frames = False
        self.batch_head = True       # with batch_frames: the recurrence of the whole clip at iter == 0, the head once on T * B states (_unroll_clip)
        self._clip = None
        self._frames = None
        self._stacked = None        # the ConvLSTM cell's stacked gate weights of the running clip unroll, see modules.LSTM.forward
        if freeze_bn:
            self.freeze_bn()

    def _trunk(self, x_nchw):
        x = ops.ToNHWC.apply(x_nchw)
        x, low = self.backbone(x)
        x = self.wasp(x)
        return self.decoder(x, low)                                # (N,h,w,16), 14 real channels

    def _trunk_frame(self, input, iter):
        b, T = input.shape[0], input.shape[1]
        if not self.batch_frames or T == 1:
            with ops.bn_counters(self):
                return self._trunk(input[:, iter])
        # the cache holds a reference to the clip tensor itself: "same object, same version" cannot be faked by a recycled
        # id / address.  A caller that passes a different tensor for a later frame simply gets the per-frame trunk.
        # ... and the key carries the in-place version of the trunk's first and last weight: an optimizer step or a
        # load_state_dict between two frames of a clip (both bump every parameter) makes the later frames miss
        # (... and the count of optimizer steps: torch's fused optimizers do not move the version counters, round 5)
        key = (input._version, self.training, torch.is_grad_enabled(), self.backbone.conv1.weight._version,
               self.decoder.last_conv[8].weight._version, ops.OPTIMIZER_STEPS)
        hit = self._frames is not None and self._frames[0] is input and self._frames[1] == key
        if iter == 0:
            xa = input.transpose(0, 1).reshape(T * b, *input.shape[2:])      # frame-major: BatchNorm group g = frame g
            with ops.bn_groups(T if self.training else 1), ops.bn_counters(self):
                self._frames = (input, key, ops.SplitBatch.apply(self._trunk(xa), T))
        elif not hit:
            self._frames = None
            with ops.bn_counters(self):
                return self._trunk(input[:, iter])
        x = self._frames[2][iter]
        if iter == T - 1:
            self._frames = None                                     # the clip is served: nothing outlives the unroll
        return x

    def train(self, mode: bool = True):
        self._frames = None          # a clip that was not served to its last frame must not outlive a mode switch
        self._stacked = None
        self._clip = None
        return super().train(mode)

    def _state(self, t, like, b):
        """(C,h,w) zeros from the first call, or the (B,C,h,w) tensor returned by the previous one."""
        if t.dim() == 3:
            t = t.unsqueeze(0).expand(b, -1, -1, -1)
        return ops.ToNHWC.apply(t.to(like.device))

    def _head(self, hide):
        h = hide
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            h = ops.conv_bias_act(h, conv, relu=True)
        return h

    def _frame_input(self, input, centermap, iter):
        x = self._trunk_frame(input, iter)                         # (B,h,w,16) fp32, 14 real channels
        cpad = ops.rup4(self.num_classes + 1)
        if x.shape[3] != cpad:          # (a bf16-storage convolution pads its output to 32 channels: back to the fp32 layout, whose
            x = x[..., :cpad]           #  one spare pad channel takes the centre map and whose width the stacked gate weights assume)
        return _AddCenter.apply(x, centermap[:, iter], self.num_classes + 1)

    def _unroll_clip(self, input, centermap):
        """batch_frames, iter == 0: the WHOLE clip at once.  The recurrence needs nothing the later calls bring (their states are the
        ones this module returned), so cell / hide of all T frames are computed here — state kept in NHWC between the frames, one
        stacked gate weight — and the five head convolutions (model/uniposeLSTM.py:85-89: two 11x11 128 -> 128 among them, a fifth
        of the step's FLOP) run ONCE on the T * B hidden states instead of T times on B: 2 650-tile launches instead of 530-tile
        ones, one weight gradient per head weight instead of T accumulated ones.  Later calls are served from the result as long
        as they pass back the tensors they were given (identity); anything else takes the per-frame path."""
        T = input.shape[1]
        c = self.num_classes + 2
        cells, hides = [], []
        stacked = None
        for t in range(T):
            z = self._frame_input(input, centermap, t)
            if t == 0:
                cell, hide = self.lstm_0(z)
            else:
                if stacked is None:
                    stacked = self.lstm.stacked()
                cell, hide = self.lstm(z, hides[-1], cells[-1], stacked=stacked)
            cells.append(cell)
            hides.append(hide)
        heats = ops.SplitBatch.apply(self._head(torch.cat(hides, 0)), T)
        out = [(ops.ToNCHW.apply(heats[t], self.num_classes + 1), ops.ToNCHW.apply(cells[t], c), ops.ToNCHW.apply(hides[t], c))
               for t in range(T)]
        key = (input._version, centermap._version, self.training, torch.is_grad_enabled(), ops.OPTIMIZER_STEPS)
        self._clip = (input, centermap, key, out)
        return out[0]

    def forward(self, input, centermap, iter, previous, previousHide, previousCell):
        b = input.shape[0]
        T = input.shape[1]
        if self.batch_frames and self.batch_head and T > 1:
            if iter == 0:
                return self._unroll_clip(input, centermap)
            cl, self._clip = self._clip, (None if iter == T - 1 else self._clip)
            if cl is not None and cl[0] is input and cl[1] is centermap and \
                    cl[2] == (input._version, centermap._version, self.training, torch.is_grad_enabled(), ops.OPTIMIZER_STEPS) and \
                    previousHide is cl[3][iter - 1][2] and previousCell is cl[3][iter - 1][1]:
                return cl[3][iter]
            self._clip = None          # the caller left the unroll this module prepared: per-frame path from here on
        # bf16 storage (ops.set_conv_math("bf16s")): the trunk runs on bf16 tensors like the image model's and hands over fp32
        # heat-maps (the decoder's last convolution writes fp32); the ConvLSTM cell and the head — 15 / 16-channel state, outside
        # the 32-channel granularity of the bf16 kernels — stay fp32 tensors, their 128-channel convolutions on bf16 MFMA operands
        z = self._frame_input(input, centermap, iter)
        x = z
        if iter == 0:
            self._stacked = None
            cell, hide = self.lstm_0(z)
        else:
            # frames 1 .. T-1 of one unroll share ONE stacked gate weight (same autograd graph: rebuilt at iter == 1 and whenever
            # the grad mode or a weight version changed in between)
            key = (torch.is_grad_enabled(), self.training, ops.OPTIMIZER_STEPS, self.lstm.conv_gx_lstm.weight._version,
                   self.lstm.conv_fh_lstm.weight._version)
            if iter == 1 or self._stacked is None or self._stacked[0] != key:
                self._stacked = (key, self.lstm.stacked())
            cell, hide = self.lstm(z, self._state(previousHide, x, b), self._state(previousCell, x, b), stacked=self._stacked[1])
            if iter == input.shape[1] - 1:
                self._stacked = None                                # the clip is served
        h = self._head(hide)
        c = self.num_classes + 2
        return ops.ToNCHW.apply(h, self.num_classes + 1), ops.ToNCHW.apply(cell, c), ops.ToNCHW.apply(hide, c)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def get_1x_lr_params(self):
        for m in self.backbone.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                for p in m.parameters(recurse=False):
                    if p.requires_grad:
                        yield p


unipose_lstm = unipose
