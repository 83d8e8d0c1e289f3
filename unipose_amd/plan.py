"""``UniPosePlan`` — the image model's inference forward as ONE call into the C ABI (``up_unipose_forward``, ABI 9).

The reference's validation / test loops run ``heat = model(input)`` per batch (unipose.py:150-160).  A plan is built once
from a model (any ``unipose`` with ``bbox`` or not, output stride 16 / 8): its BatchNorm layers are folded into the
convolutions (``checkpoint.fold_batchnorm``), the folded weights are packed into plan-owned device memory, the activation
workspace is one torch byte tensor.  ``plan(x)`` then issues the whole network from C — no autograd, no per-layer Python —
and returns the heat-maps (the two slices of ``unipose(bbox=True)`` when the model has the box head).

    model = unipose("MPII", num_classes=16).cuda().eval(); model.load_state_dict(...)
    plan = UniPosePlan(model, batch=8, height=368, width=368)
    heat = plan(images)                      # equal bits to checkpoint.load_folded(...)(images)

The launches are those of the drop-in module's folded forward, so the results agree bit for bit
(tests/test_plan_emu.py, tests/test_plan_gpu.py).  Weights are captured at construction: call ``refresh(model)`` after
changing them.  Inference only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _C
from .checkpoint import fold_batchnorm


class _Config(C.Structure):
    """up_unipose_config"""
    _fields_ = [(n, C.c_int32) for n in ("batch", "height", "width", "output_stride", "out_channels")]


class UniPosePlan:
    def __init__(self, model, batch: int, height: int, width: int):
        if model.training:
            raise ValueError("UniPosePlan captures the inference forward: call model.eval() first")
        if getattr(model, "stride", 8) != 8:
            raise NotImplementedError("UniPosePlan: stride != 8 (the extra 8x up-sampling of model/unipose.py:31-32) is not planned")
        if batch < 1 or height < 8 or width < 8:
            raise ValueError(f"UniPosePlan: batch {batch}, {height} x {width} input")
        w0 = model.backbone.conv1.weight
        if not w0.is_cuda and not _C._ALLOW_HOST_POINTERS:
            raise _C.UniPoseHipError("UniPosePlan needs a CUDA(HIP) model; there is no CPU fallback")
        self.device = w0.device
        self.bbox = bool(getattr(model, "bbox", False))
        self.num_classes = model.num_classes
        out_channels = model.decoder.last_conv[8].out_channels
        os_ = 16 if model.wasp.aspp2.atrous_conv.dilation[0] == 18 else 8
        self.cfg = _Config(batch, height, width, os_, out_channels)
        self._plan = C.c_void_p()
        L = _C.lib()
        _C.check(L.up_unipose_plan_create(C.byref(self.cfg), C.byref(self._plan)), "unipose_plan_create")
        self.batch, self.height, self.width, self.out_channels = batch, height, width, out_channels
        # heat-map size: three ceil-halvings (7x7 stride-2 stem, 3x3 stride-2 max-pool, layer2's stride 2), the arithmetic of
        # plan.hip's build(): ceil(H / 8), NOT H // 8 — the two differ whenever H % 8 != 0 (a 52 x 52 input gives 7 x 7 maps)
        self.out_height, self.out_width = (height - 1) // 8 + 1, (width - 1) // 8 + 1
        self.workspace = torch.empty(max(L.up_unipose_plan_workspace(self._plan), 256) + 256, dtype=torch.uint8, device=self.device)
        self.refresh(model)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def refresh(self, model):
        """(re)load the weights: BatchNorm folded in float64 and rounded once, like checkpoint.fold_batchnorm / load_folded"""
        L = _C.lib()
        folded = fold_batchnorm(model)
        keep = []
        shape = (C.c_int32 * 4)()
        hb = C.c_int32()
        for i in range(L.up_unipose_plan_num_convs(self._plan)):
            name = L.up_unipose_plan_conv_name(self._plan, i).decode()
            _C.check(L.up_unipose_plan_conv_shape(self._plan, i, shape, C.byref(hb)), "unipose_plan_conv_shape")
            w = folded[name + ".weight"].to(self.device, torch.float32).contiguous()
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"UniPosePlan: {name}.weight is {tuple(w.shape)}, the plan expects {tuple(shape)}")
            b = folded.get(name + ".bias")
            if bool(hb.value) != (b is not None):
                raise ValueError(f"UniPosePlan: {name} {'needs' if hb.value else 'must not have'} a bias after folding")
            if b is not None:
                b = b.to(self.device, torch.float32).contiguous()
            _C.check(L.up_unipose_plan_set_conv(self._plan, i, w.data_ptr(), None if b is None else b.data_ptr(), self._stream()),
                     "unipose_plan_set_conv")
            keep += [w, b]
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()       # the staging tensors in `keep` may go now
        del keep

    def __call__(self, x: torch.Tensor, out: torch.Tensor = None):
        if tuple(x.shape) != (self.batch, 3, self.height, self.width) or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError(f"UniPosePlan: input {tuple(x.shape)} {x.dtype} on {x.device}, planned for "
                             f"{(self.batch, 3, self.height, self.width)} float32 on {self.device}")
        x = x.contiguous()
        shape = (self.batch, self.out_channels, self.out_height, self.out_width)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or out.device != self.device or not out.is_contiguous():
            raise ValueError(f"UniPosePlan: `out` is {tuple(out.shape)} {out.dtype} on {out.device} (contiguous: {out.is_contiguous()}), "
                             f"the plan writes a contiguous float32 {shape} on {self.device}")
        ws = self.workspace
        off = (-ws.data_ptr()) % 256
        _C.check(_C.lib().up_unipose_forward(self._plan, x.data_ptr(), out.data_ptr(), ws.data_ptr() + off, ws.numel() - off,
                                             self._stream()), "unipose_forward")
        if self.bbox:                   # model/unipose.py:34-35
            return out[:, 0:self.num_classes + 1], out[:, self.num_classes + 1:]
        return out

    def close(self):
        if self._plan:
            _C.lib().up_unipose_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass
