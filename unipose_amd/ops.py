"""Host-side operators: torch tensors in, C-ABI calls out (``_C.lib()``), wired into autograd.

Tensors inside the network are NHWC fp32 ("pixel-major", see include/unipose_hip.h): shape
(N, H, W, Cp) with Cp a multiple of 4; ``logical`` channel counts smaller than Cp (3, 14, 15, 17 …)
are carried by the weights' shapes, pad channels hold zeros.  PyTorch owns all memory; kernels borrow
raw pointers and run on the current stream.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import weakref
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _C

BN_EPS_DEFAULT = 1e-5


def rup4(c: int) -> int:
    return (c + 3) // 4 * 4


def rup32(c: int) -> int:
    return (c + 31) // 32 * 32


def _dt(t: torch.Tensor) -> int:
    """Element-type code of an activation tensor for the `_t` entry points (UP_DT_F32 / UP_DT_BF16)."""
    return 1 if t.dtype == torch.bfloat16 else 0


def _padk(k: int, like: torch.Tensor) -> int:
    """Physical channel count of a convolution output: multiples of 4 (fp32), of 32 in bf16 storage — there every
    reduction (also the data gradient's, over the OUTPUT channels) must be a whole number of 32-wide K slices."""
    return rup32(k) if like.dtype == torch.bfloat16 else rup4(k)


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev_ok(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda and not _C._ALLOW_HOST_POINTERS:
            raise _C.UniPoseHipError("unipose_amd kernels need CUDA(HIP) tensors; there is no CPU fallback")
        if t.dtype not in (torch.float32, torch.bfloat16, torch.uint8, torch.int32):
            raise TypeError(f"unsupported dtype {t.dtype}")


def _nhwc_ok(t: torch.Tensor):
    n, h, w, c = t.shape
    q = 8 if t.dtype == torch.bfloat16 else 4                           # elements per 16 bytes
    if t.is_contiguous() and c % q == 0 and t.data_ptr() % 16 == 0:     # the common case, without four stride() calls
        return c
    ld = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != w * ld or (n > 1 and t.stride(0) != h * w * ld) or ld % q or \
            t.data_ptr() % 16:
        raise ValueError(f"tensor is not a pixel-contiguous NHWC view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


def _dense(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_WS = {}


def workspace(dev: torch.device, nbytes: int, tag: str = "main") -> torch.Tensor:
    """Stream-ordered scratch (split-K slabs, BN-backward partials); grows monotonically per (device, stream tag)."""
    key = (dev.type, dev.index, tag)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = t
    return t


# --------------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------------
class ConvCfg:
    __slots__ = ("stride", "pad", "dil")

    def __init__(self, stride=1, pad=0, dil=1):
        self.stride, self.pad, self.dil = int(stride), int(pad), int(dil)


def make_desc(x: torch.Tensor, weight: torch.Tensor, cfg: ConvCfg, ldy: Optional[int] = None) -> _C.ConvDesc:
    n, h, w, cp = x.shape
    k, c, r, s = weight.shape
    if cp < c or cp % 4:
        raise ValueError(f"input has {cp} physical channels, weight expects {c}")
    p = (h + 2 * cfg.pad - cfg.dil * (r - 1) - 1) // cfg.stride + 1
    q = (w + 2 * cfg.pad - cfg.dil * (s - 1) - 1) // cfg.stride + 1
    d = _C.ConvDesc()
    # Cp = physical channels of the input (pad channels are zeros and meet zero weights): lets a producer
    # pad e.g. the 304-channel decoder concat to 320 so that 32-wide K slices never straddle a filter tap
    d.N, d.H, d.W, d.C, d.Cp, d.ldx = n, h, w, c, cp, _nhwc_ok(x)
    d.K, d.R, d.S = k, r, s
    d.stride, d.pad, d.dil = cfg.stride, cfg.pad, cfg.dil
    d.P, d.Q = p, q
    d.Kp = _padk(k, x)
    d.ldy = ldy if ldy is not None else _padk(k, x)
    return d


# The batched re-pack of one step (0.44 ms for the 115 weights of the network, on the critical path of the forward since the
# optimizer hook of round 5) is split: the images of the first _REPACK_HEAD parameters (stem + layer1: what the forward needs first)
# are re-packed on the current stream, the rest on the weight-gradient side stream — idle during the forward — while the stem and
# layer1 run; the first request for one of THOSE images makes the current stream wait for the side stream's event.
ASYNC_REPACK = os.environ.get("UNIPOSE_ASYNC_REPACK", "1") != "0"
_REPACK_HEAD = 12


def _settle_repack(ps, dev=None):
    """An earlier split re-pack may still be writing images / reading a job table on the side stream: make the current stream
    wait for it BEFORE any of those buffers is dropped, replaced or re-written (the caching allocator could otherwise hand a
    freed block to a main-stream allocation while the side stream still uses it)."""
    if ps.pending is not None:
        ev = ps.pending[0]
        ps.pending = None
        if dev is None or dev.type == "cuda":
            torch.cuda.current_stream(dev).wait_event(ev)


def _launch_repack(ps, weight, entry_point, what, job_bytes):
    """`ps`: a _PackSet / _Pack16Set whose job table is current; launches the batched re-pack of its STALE jobs — all of them
    after a step of the one optimizer of a process, a sub-table (cached per stale set) when a second model keeps its images."""
    dev = weight.device
    _settle_repack(ps, dev)          # first: the sub-table below may replace one an in-flight launch still reads
    stale = [k for k in ps.order if ps.entries[k][1] != ps.entries[k][0]()._version]
    if len(stale) == len(ps.order):
        table, keys = ps.table, ps.order
    else:
        sig = tuple(stale)
        hit = ps.subtables.get(sig)
        if hit is None:
            if len(ps.subtables) >= 4:
                ps.subtables.clear()
            rows = torch.tensor([ps.order.index(k) for k in stale], dtype=torch.int64, device=ps.table.device)
            hit = ps.subtables[sig] = ps.table.index_select(0, rows).contiguous()
        table, keys = hit, stale
    n = len(keys)
    if n == 0:
        return
    L = _C.lib()
    if ASYNC_REPACK and dev.type == "cuda" and n > 2 * _REPACK_HEAD and not torch.cuda.is_current_stream_capturing():
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        _C.check(getattr(L, entry_point)(table.data_ptr(), _REPACK_HEAD, main.cuda_stream), what)
        side.wait_stream(main)                       # the optimizer's writes (and every reader of the old images) are behind us
        _C.check(getattr(L, entry_point)(table.data_ptr() + _REPACK_HEAD * job_bytes, n - _REPACK_HEAD, side.cuda_stream), what)
        ev = torch.cuda.Event()
        ev.record(side)
        ps.pending = (ev, set(keys[_REPACK_HEAD:]), set())
    else:
        _C.check(getattr(L, entry_point)(table.data_ptr(), n, _stream(weight)), what)


def retire_pending_repacks():
    """Forget the events of finished split re-packs (call after a device synchronisation: graph.GraphedTrainStep does before it
    captures — a capturing stream must not wait for an event recorded outside the capture)."""
    for ps in list(_PACK_CACHE.values()) + list(_PACK16_CACHE.values()):
        ps.pending = None


def _await_repack(ps, key, dev):
    """the current stream is about to read the images of parameter `key`"""
    # (the event stays pending after a reader waited — a reader on ANOTHER stream must wait too, ADVICE r5 — but every stream
    #  waits only once; _settle_repack retires it)
    if ps.pending is not None and key in ps.pending[1]:
        st = torch.cuda.current_stream(dev)
        if st.cuda_stream not in ps.pending[2]:
            st.wait_event(ps.pending[0])
            ps.pending[2].add(st.cuda_stream)


class _PackSet:
    """Packed weight images of every convolution parameter seen on one device.

    The images (forward [K][R*S][Cp], data gradient [C][R*S][Kp]) live in persistent buffers; the parameter's
    in-place version counter tells when they are stale (optimizer steps and load_state_dict bump it).  An optimizer
    step makes ALL of them stale at once, so the first stale hit re-packs every registered parameter with ONE
    up_pack_weights_batched launch driven by a job table in device memory (2 x 115 five-microsecond launches per
    step otherwise)."""

    def __init__(self):
        self.entries = {}          # id(weight) -> [weakref, version, wf, wd, (K, C, Cp, Kp, taps), data_ptr]
        self.table = None          # device job table, rebuilt when the membership changes
        self.order = []
        self.pending = None        # (event, keys): images being re-packed on the side stream, see _launch_repack
        self.subtables = {}        # stale-key tuple -> job table of just those parameters

    def get(self, weight, d):
        key = id(weight)
        nf, nd = d.K * d.R * d.S * d.Cp, d.C * d.R * d.S * d.Kp
        e = self.entries.get(key)
        if e is not None and e[0]() is weight and e[2].numel() == nf and e[3].numel() == nd and \
                e[5] == weight.data_ptr():                 # `param.data = other` keeps the version but moves the storage
            if e[1] != weight._version:
                self.repack(weight)
            _await_repack(self, key, weight.device)
            return e[2], e[3]
        _settle_repack(self, weight.device)     # an entry / the table is about to be replaced
        wf = torch.empty(nf, dtype=torch.float32, device=weight.device)
        wd = torch.empty(nd, dtype=torch.float32, device=weight.device)
        _C.check(_C.lib().up_pack_weights(C.byref(d), _dense(weight).data_ptr(), wf.data_ptr(), wd.data_ptr(),
                                          _stream(weight)), "pack_weights")
        if isinstance(weight, torch.nn.Parameter) and weight.is_contiguous():      # temporaries are packed per use
            if len(self.entries) > 4096:
                self.entries.clear()
            geom = d.stride | (d.R << 4) | (d.S << 10) | (d.pad << 16) | (d.dil << 24)      # up_pack_job.geometry
            self.entries[key] = [weakref.ref(weight), weight._version, wf, wd, (d.K, d.C, d.Cp, d.Kp, d.R * d.S, geom),
                                 weight.data_ptr()]
            self.table = None
        return wf, wd

    def repack(self, weight):
        if self.table is None:
            import numpy as np
            dead = [k for k, e in self.entries.items() if e[0]() is None]
            for k in dead:
                del self.entries[k]
            self.order = list(self.entries)
            job = np.zeros((len(self.order), 6), dtype=np.int64)      # up_pack_job: 3 pointers + 6 int32
            for i, k in enumerate(self.order):
                w, _, wf, wd, (kk, cc, cp, kp, taps, geom), _ = self.entries[k]
                job[i, 0], job[i, 1], job[i, 2] = w().data_ptr(), wf.data_ptr(), wd.data_ptr()
                job[i, 3:].view(np.int32)[:] = (kk, cc, cp, kp, taps, geom)
            self.table = torch.from_numpy(job).to(weight.device)
            self.subtables = {}
        live = [self.entries[k] for k in self.order]
        if any(e[0]() is None for e in live):                         # a parameter died: its pointer is stale
            self.table = None
            return self.repack(weight)
        _launch_repack(self, weight, "up_pack_weights_batched", "pack_weights_batched", 48)
        for e in live:
            e[1] = e[0]()._version


_PACK_CACHE = {}                   # device index -> _PackSet


def _packed(weight: torch.Tensor, d: _C.ConvDesc):
    """(forward image, data-gradient image) of an OIHW parameter, see _PackSet."""
    ps = _PACK_CACHE.get(weight.device.index)
    if ps is None:
        ps = _PACK_CACHE[weight.device.index] = _PackSet()
    return ps.get(weight, d)


# Arithmetic of the convolutions:
#   0  exact fp32 (v_mfma_f32_32x32x2_f32)                                  -- default, parity configuration
#   1  split-bf16: a*b ~= ah*bh + ah*bl + al*bh on bf16 MFMA, fp32-equivalent (2^-16 per product); weight gradients
#      stay on the exact fp32 MFMA
#   2  plain bf16 operands, fp32 accumulation (BASELINE config 5): forward, data gradient AND weight gradient
# Layers whose padded channel count is not a multiple of 32 (stem, 15-channel ConvLSTM convs) stay exact.
#   3  "bf16s": as 2 with bf16 STORAGE — every activation and activation gradient behind the stem's max-pool is a bf16
#      tensor in HBM (half the bytes of every streaming pass and of every operand gather); BatchNorm statistics,
#      accumulators, weights, weight gradients and the optimizer stay fp32.  The kernels dispatch on the tensor dtype,
#      the switch only tells the model where to change the element type (modules.ResNet.forward, unipose.forward).
MATH_F32, MATH_BF16X3, MATH_BF16, MATH_BF16S = 0, 1, 2, 3
MATH_BF16S_F32OUT = 4        # up_conv2d_fwd_bf16 only: bf16 input, fp32 output (the network's last convolution)
CONV_MATH = MATH_F32
_MATH_NAMES = {"f32": 0, "fp32": 0, "bf16x3": 1, "split": 1, "bf16": 2, "bf16s": 3, "bf16_storage": 3}


def set_conv_math(mode):
    global CONV_MATH
    CONV_MATH = _MATH_NAMES.get(mode, mode)
    if CONV_MATH not in (0, 1, 2, 3):
        raise ValueError(f"unknown conv math {mode!r}")


def storage_dtype():
    """Element type of the activations behind the stem (torch.bfloat16 in the bf16-storage configuration)."""
    return torch.bfloat16 if CONV_MATH == MATH_BF16S else torch.float32


if os.environ.get("UNIPOSE_CONV_MATH"):           # e.g. UNIPOSE_CONV_MATH=bf16x3 python -m pytest tests -m gpu
    set_conv_math(os.environ["UNIPOSE_CONV_MATH"])

WGRAD_BF16_ANY_WIDTH = False     # tests only: drive the bf16 weight-gradient kernel at channel counts the model keeps in fp32


class _Pack16Set:
    """bf16 (hi, lo) planes of the forward and data-gradient weight images of every convolution parameter seen on one device
    (arithmetic modes bf16x3 / bf16 / bf16s): the _PackSet scheme — persistent buffers, staleness by version counter (and the
    optimizer hook below), ALL registered parameters re-packed by ONE up_pack_weights_bf16_batched launch at the first stale hit
    (2 x 115 pack launches per step otherwise, ~1 ms of the 37 ms bf16-storage step at 736x736)."""

    def __init__(self):
        self.entries = {}          # id(weight) -> [weakref, version, wf(2, nf) int16, wd(2, nd) int16, (K, C, Cp, Kp, taps), data_ptr]
        self.table = None
        self.order = []
        self.pending = None
        self.subtables = {}
        self.with_lo = True        # the job table carries the lo planes (only the split-bf16 arithmetic reads them)

    def get(self, weight, d):
        key = id(weight)
        nf, nd = d.K * d.R * d.S * d.Cp, d.C * d.R * d.S * d.Kp
        e = self.entries.get(key)
        need_lo = CONV_MATH == MATH_BF16X3
        if need_lo != self.with_lo:        # the arithmetic changed: other planes in the job table, and lo planes may be stale
            self.with_lo, self.table = need_lo, None
            if need_lo:
                for q in self.entries.values():
                    q[1] = -1
        if e is not None and e[0]() is weight and e[2].shape[1] == nf and e[3].shape[1] == nd and e[5] == weight.data_ptr():
            if e[1] != weight._version:
                self.repack(weight)
            _await_repack(self, key, weight.device)
            return e[2], e[3]
        _settle_repack(self, weight.device)     # an entry / the table is about to be replaced
        wf = torch.empty((2, nf), dtype=torch.int16, device=weight.device)
        wd = torch.empty((2, nd), dtype=torch.int16, device=weight.device)
        _C.check(_C.lib().up_pack_weights_bf16(C.byref(d), _dense(weight).data_ptr(), wf[0].data_ptr(), wf[1].data_ptr(),
                                               wd[0].data_ptr(), wd[1].data_ptr(), _stream(weight)), "pack_weights_bf16")
        if isinstance(weight, torch.nn.Parameter) and weight.is_contiguous():      # temporaries are packed per use
            if len(self.entries) > 4096:
                self.entries.clear()
            self.entries[key] = [weakref.ref(weight), weight._version, wf, wd, (d.K, d.C, d.Cp, d.Kp, d.R * d.S), weight.data_ptr()]
            self.table = None
        return wf, wd

    def repack(self, weight):
        if self.table is None:
            import numpy as np
            for k in [k for k, e in self.entries.items() if e[0]() is None]:
                del self.entries[k]
            self.order = list(self.entries)
            job = np.zeros((len(self.order), 8), dtype=np.int64)      # up_pack_job_bf16: 5 pointers + 6 int32
            for i, k in enumerate(self.order):
                w, _, wf, wd, (kk, cc, cp, kp, taps), _ = self.entries[k]
                job[i, :5] = (w().data_ptr(), wf[0].data_ptr(), wf[1].data_ptr() if self.with_lo else 0, wd[0].data_ptr(),
                              wd[1].data_ptr() if self.with_lo else 0)
                job[i, 5:].view(np.int32)[:] = (kk, cc, cp, kp, taps, 0)
            self.table = torch.from_numpy(job).to(weight.device)
            self.subtables = {}
        live = [self.entries[k] for k in self.order]
        if any(e[0]() is None for e in live):                         # a parameter died: its pointer is stale
            self.table = None
            return self.repack(weight)
        _launch_repack(self, weight, "up_pack_weights_bf16_batched", "pack_weights_bf16_batched", 64)
        for e in live:
            e[1] = e[0]()._version


_PACK16_CACHE = {}                 # device index -> _Pack16Set


def _packed_bf16(weight: torch.Tensor, d: _C.ConvDesc):
    """bf16 (hi, lo) planes of the forward and data-gradient weight images, see _Pack16Set."""
    ps = _PACK16_CACHE.get(weight.device.index)
    if ps is None:
        ps = _PACK16_CACHE[weight.device.index] = _Pack16Set()
    return ps.get(weight, d)


def invalidate_packed_weights():
    """Mark every cached weight image stale.  The cache follows a parameter's in-place version counter, but not every writer moves
    it: torch's FUSED optimizers (``Adam(fused=True)``: one multi-tensor kernel through ``torch._fused_adam_``) and in-place edits
    through ``.data`` leave ``_version`` untouched, and a training loop on such an optimizer then ran every convolution on the
    weights of step 0 (found by the G16 trajectory golden, round 5: loss 1.5628 -> 1.5411 where the reference goes to 1.4078).
    Called from a global optimizer post-step hook (below), so a plain ``optimizer.step()`` needs nothing; call it by hand after
    editing weights through ``.data`` or a raw pointer."""
    for ps in list(_PACK_CACHE.values()) + list(_PACK16_CACHE.values()):
        for e in ps.entries.values():
            e[1] = -1


OPTIMIZER_STEPS = 0     # bumped by the hook: a cache key for anything derived from the parameters (unipose_lstm's clip cache)


def _optimizer_stepped(optimizer, _args, _kwargs):
    """only the parameters of the optimizer that stepped go stale: a second model in the process (bench.py's legs, a frozen
    teacher) keeps its images"""
    global OPTIMIZER_STEPS
    OPTIMIZER_STEPS += 1
    try:
        ids = {id(p) for g in optimizer.param_groups for p in g["params"]}
    except Exception:       # noqa: BLE001  (an optimizer without the standard layout: everything goes stale)
        return invalidate_packed_weights()
    for ps in list(_PACK_CACHE.values()) + list(_PACK16_CACHE.values()):
        for k, e in ps.entries.items():
            if k in ids:
                e[1] = -1


try:        # every torch.optim optimizer, fused or not, reports its step here
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_step_hook
    _STEP_HOOK = _reg_step_hook(_optimizer_stepped)
except ImportError:     # (older torch: the version counter alone, i.e. no fused optimizers)
    _STEP_HOOK = None


# UNIPOSE_DEBUG_PACK=1: every use of a cached packed image first re-packs the weight into a scratch image and compares — a weight
# edited behind the cache's back (`.data`, a raw pointer, an optimizer that neither moves version counters nor reports its step)
# raises at the first convolution that reads the stale image instead of silently training on old weights (the round-2..4 bug the
# G16 trajectory golden found).  One extra pack launch and a device-to-host comparison per convolution: a debugging mode.
DEBUG_PACK = os.environ.get("UNIPOSE_DEBUG_PACK", "0") != "0"


def _check_packed(weight, d, wf):
    ref = torch.empty_like(wf)
    _C.check(_C.lib().up_pack_weights(C.byref(d), _dense(weight).data_ptr(), ref.data_ptr(), None, _stream(weight)), "pack_weights")
    if not torch.equal(ref, wf):
        raise _C.UniPoseHipError(
            f"stale packed weight image for a {tuple(weight.shape)} convolution weight: the parameter changed without the cache "
            "noticing (edited through .data / a raw pointer?) — call ops.invalidate_packed_weights() after such edits")


def packed_fwd(weight: torch.Tensor, d: _C.ConvDesc) -> torch.Tensor:
    wf = _packed(weight, d)[0]
    if DEBUG_PACK:
        _check_packed(weight, d, wf)
    return wf


def packed_dgrad(weight: torch.Tensor, d: _C.ConvDesc) -> torch.Tensor:
    return _packed(weight, d)[1]


def conv_fwd_raw(x, weight, cfg, *, scale=None, shift=None, bias=None, residual=None, relu=False, stats=False,
                 out=None, out_f32=False, fold=None):
    """One launch of the implicit-GEMM kernel.  Returns (y, desc, stats_tensor|None).
    out_f32: in bf16 storage, write an fp32 output (no effect on fp32 tensors).
    fold (with stats): a _C.BnFold the launch may carry out itself (BatchNorm finalize by its last workgroup, include/unipose_hip.h
    up_bn_fold); fold.folded tells afterwards whether it did."""
    _dev_ok(x, weight, scale, shift, bias, residual)
    d = make_desc(x, weight, cfg, None if out is None else _nhwc_ok(out))
    out_f32 = out_f32 and x.dtype == torch.bfloat16
    if out is None:
        alloc = torch.zeros if d.ldy != d.K else torch.empty      # pad channels must read as zeros
        out = alloc((d.N, d.P, d.Q, d.ldy), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    ep = _C.ConvEpilogue()
    ep.scale, ep.shift, ep.bias = _ptr(scale), _ptr(shift), _ptr(bias)
    ep.residual = _ptr(residual)
    ep.ldr = _nhwc_ok(residual) if residual is not None else 0
    ep.relu = int(relu)
    st = None
    if x.dtype == torch.bfloat16:
        math = MATH_BF16S
    elif CONV_MATH in (MATH_BF16X3, MATH_BF16, MATH_BF16S) and d.Cp % 32 == 0:
        math = min(CONV_MATH, MATH_BF16)
    else:
        math = MATH_F32
    if stats:
        tiles = _C.lib().up_conv_stats_tiles_math(C.byref(d), math)   # the tile rule depends on the arithmetic
        st = torch.empty((tiles, d.K, 3), dtype=torch.float32, device=x.device)
        ep.stats = st.data_ptr()
        if fold is not None:
            ep.fold = C.pointer(fold)
    if x.dtype == torch.bfloat16:                                  # bf16 storage: the kernels follow the tensor
        if d.Cp % 32 or (residual is not None and residual.dtype != x.dtype) or \
                out.dtype != (torch.float32 if out_f32 else x.dtype) or (out_f32 and (residual is not None or stats)):
            raise NotImplementedError(f"bf16-storage convolution needs 32-aligned input channels (got {d.Cp}) and bf16 "
                                      "residual / output tensors (fp32 output: no residual, no statistics)")
        wf, _ = _packed_bf16(weight, d)
        _C.check(_C.lib().up_conv2d_fwd_bf16(C.byref(d), x.data_ptr(), wf[0].data_ptr(), wf[1].data_ptr(),
                                             out.data_ptr(), C.byref(ep), MATH_BF16S_F32OUT if out_f32 else MATH_BF16S,
                                             _stream(x)), "conv2d_fwd_bf16s")
    elif CONV_MATH in (MATH_BF16X3, MATH_BF16, MATH_BF16S) and d.Cp % 32 == 0:
        wf, _ = _packed_bf16(weight, d)
        _C.check(_C.lib().up_conv2d_fwd_bf16(C.byref(d), x.data_ptr(), wf[0].data_ptr(), wf[1].data_ptr(),
                                             out.data_ptr(), C.byref(ep), min(CONV_MATH, MATH_BF16), _stream(x)),
                 "conv2d_fwd_bf16")
    else:
        wp = packed_fwd(weight, d)
        _C.check(_C.lib().up_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), out.data_ptr(), C.byref(ep),
                                        _stream(x)), "conv2d_fwd")
    return out, d, st


# Forward statistics from per-group tiles (up_conv2d_fwd_grouped) instead of the extra pass over y: built, correct, and OFF —
# the Welford epilogue on 112 launches + 1.3 % tile padding cost 2.4 ms of kernel time for the 1.7 ms pass they remove
# (UniPose-LSTM step 103.4 -> 104.0 ms, profiles/r04_experiments.txt).  The per-group tiling itself is what the data gradient uses.
GROUPED_TILES = os.environ.get("UNIPOSE_GROUPED_TILES", "0") != "0"


def conv_fwd_grouped_raw(x, weight, cfg, groups):
    """Forward convolution over `groups` equal batches stacked along N with every group tiled on its own (up_conv2d_fwd_grouped):
    returns (y, desc, stats[groups][tiles][K][3]) or None when the launch cannot be tiled per group (the caller then uses
    conv_fwd_raw + up_bn_batch_stats_t)."""
    if not GROUPED_TILES or x.dtype != torch.float32 or CONV_MATH != MATH_F32:
        return None
    _dev_ok(x, weight)
    d = make_desc(x, weight, cfg)
    tiles = _C.lib().up_conv_stats_tiles_grouped(C.byref(d), groups)
    if tiles <= 0:
        return None
    alloc = torch.zeros if d.ldy != d.K else torch.empty      # pad channels must read as zeros
    out = alloc((d.N, d.P, d.Q, d.ldy), dtype=x.dtype, device=x.device)
    st = torch.empty((groups, tiles, d.K, 3), dtype=torch.float32, device=x.device)
    wp = packed_fwd(weight, d)
    _C.check(_C.lib().up_conv2d_fwd_grouped(C.byref(d), x.data_ptr(), wp.data_ptr(), out.data_ptr(), st.data_ptr(), groups,
                                            _stream(x)), "conv2d_fwd_grouped")
    return out, d, st


class BnSlot:
    """Hands the BatchNorm-backward REDUCTION of a layer z = relu(bn(y) (+ res)) to the data-gradient launch of the one
    convolution that consumes z (resnet.py:25-33: bn1 -> relu -> conv2, bn2 -> relu -> conv3, and a block's output into the next
    identity block's conv1, whose kernel already adds the skip gradient): that launch produces dz and reduces sum(g), sum(g * xhat)
    per row tile in its epilogue (f32_glds.h BNRED), so the producer's backward runs finalize + apply only.
    The wiring assumes z has no other consumer; should one exist after all (a tap taken between two blocks without a hook), autograd
    adds its gradient to dz either into a new tensor (another address) or IN PLACE (same address, but the tensor's version counter
    moves — the kernels' raw writes never touch it), so the producer's backward uses the sums only if dz has the address AND the
    version the consumer's launch left, else it falls back to its own reduction.  The producer fills y / bits / mean / invstd in
    its forward; the consumer's backward fills partial, dz_ptr and dz_version."""
    __slots__ = ("y", "bits", "mean", "invstd", "C", "partial", "dz_ptr", "dz_version", "groups", "gstride", "dgb", "gsum")

    def __init__(self, y=None, bits=None, mean=None, invstd=None, C=0):
        self.y, self.bits, self.mean, self.invstd, self.C = y, bits, mean, invstd, C
        self.partial, self.dz_ptr, self.dz_version = None, 0, -1
        # (2, C) dgamma / dbeta when the consumer's launch also carried the merge of its partial rows (up_bn_reduce_slot.folded)
        self.dgb = None
        self.gsum = None       # (groups, 2, C) per-group sums of a grouped launch that merged them (up_bn_reduce_slot.gsum)
        # row groups (ops.bn_groups): mean / invstd are the first group's vectors inside coef[groups][4][C], gstride = 4 * C floats
        # to the next group's; the consumer's data gradient is tiled per group and partial is [groups * tiles][C][2]
        self.groups, self.gstride = 1, 0

    def clear(self):
        self.y = self.bits = self.mean = self.invstd = self.partial = self.dgb = self.gsum = None
        self.dz_ptr, self.dz_version = 0, -1
        self.groups, self.gstride = 1, 0

    def matches(self, dz, y):
        return self.partial is not None and self.dz_ptr == dz.data_ptr() and self.dz_version == dz._version and self.y is y


# tests: how often a BatchNorm backward really started from a consumer's sums / a projection's data gradient rode in conv1's launch
HOST_COUNTERS = {"bn_prereduced": 0, "dx_handed_over": 0, "bn_fwd_folded": 0, "bn_bwd_folded": 0}
BN_FUSE_REDUCE = os.environ.get("UNIPOSE_BN_FUSE_REDUCE", "1") != "0"     # development switches (A/B runs)
MASKED_ADDEND = os.environ.get("UNIPOSE_MASKED_ADDEND", "1") != "0"
GROUPED_REDUCE = os.environ.get("UNIPOSE_GROUPED_REDUCE", "1") != "0"     # the fused reduction also inside ops.bn_groups (row groups)
DX_HANDOVER = os.environ.get("UNIPOSE_DX_HANDOVER", "1") != "0"           # projection blocks: downsample's dx rides in conv1's launch
# the fused reduction only on data-gradient launches whose reduction (taps x output channels) is at least this long (A/B knobs)
BNRED_MIN_K = int(os.environ.get("UNIPOSE_BNRED_MIN_K", "0"))
BNRED_MIN_K_BF16 = int(os.environ.get("UNIPOSE_BNRED_MIN_K_BF16", "0"))


def dgrad_extras_tiles(d: _C.ConvDesc, x_shape, dtype) -> int:
    """Row tiles of the data-gradient launch of `d` if its kernel supports the extended epilogue (masked addend, fused
    BatchNorm-backward reduction: up_conv2d_bwd_data_ex), else 0."""
    if dtype == torch.bfloat16:
        math = MATH_BF16S
    elif dtype == torch.float32 and CONV_MATH == MATH_F32:
        math = MATH_F32
    else:
        return 0
    dd = _C.ConvDesc.from_buffer_copy(d)
    dd.ldx = x_shape[3]
    return _C.lib().up_conv2d_bwd_data_tiles_math(C.byref(dd), math)


def conv_bwd_data_raw(dy, weight, d: _C.ConvDesc, x_shape, dev, add=None, bn_slot=None, add_bits=None):
    """dx = dgrad(dy) (+ add: another gradient of the same input, summed in the kernel epilogue).
    bn_slot: see BnSlot (stride 1; a launch that cannot carry the reduction leaves bn_slot.partial None).
    add_bits: ReLU sign bits applied to `add` in the epilogue (the caller made sure with dgrad_extras_tiles that the launch can)."""
    n, h, w, cp = x_shape
    alloc = torch.zeros if cp != d.C else torch.empty
    dx = alloc((n, h, w, cp), dtype=dy.dtype, device=dev)
    dd = _C.ConvDesc.from_buffer_copy(d)
    dd.ldx = cp
    dd.ldy = _nhwc_ok(dy)
    ld_add = _nhwc_ok(add) if add is not None else 0
    # extended epilogue (masked addend / fused BatchNorm-backward reduction): fp32 and bf16 storage
    ex_math = MATH_BF16S if dy.dtype == torch.bfloat16 else (MATH_F32 if CONV_MATH == MATH_F32 else -1)
    tiles = 0
    want_slot = bn_slot is not None and bn_slot.y is not None and bn_slot.C == d.C == cp and bn_slot.y.dtype == dy.dtype and \
        bn_slot.y.shape[:3] == (n, h, w) and d.R * d.S * d.Kp >= (BNRED_MIN_K_BF16 if dy.dtype == torch.bfloat16 else BNRED_MIN_K)
    groups = bn_slot.groups if want_slot else 1
    if (want_slot or add_bits is not None) and ex_math >= 0:
        tiles = _C.lib().up_conv2d_bwd_data_tiles_math(C.byref(dd), ex_math)
        if groups > 1 and tiles > 0:       # every group tiled on its own, or no fused reduction at all
            gt = _C.lib().up_conv2d_bwd_data_tiles_grouped(C.byref(dd), groups) if ex_math == MATH_F32 else 0
            if gt > 0:
                tiles = groups * gt
            else:
                want_slot, groups = False, 1
                if add_bits is None:
                    tiles = 0
    if add_bits is not None and tiles <= 0:
        raise _C.UniPoseHipError("conv_bwd_data_raw: this launch cannot mask its addend (check dgrad_extras_tiles first)")
    if tiles > 0:
        ep = _C.DgradEpilogue()
        ep.add, ep.add_relu_bits, ep.ld_add = _ptr(add), _ptr(add_bits), ld_add
        if want_slot:
            partial = torch.empty((tiles, d.C, 2), dtype=torch.float32, device=dev)
            sl = _C.BnReduceSlot()
            sl.y, sl.relu_bits, sl.mean, sl.invstd = bn_slot.y.data_ptr(), _ptr(bn_slot.bits), bn_slot.mean.data_ptr(), \
                bn_slot.invstd.data_ptr()
            sl.partial, sl.ld, sl.C = partial.data_ptr(), _nhwc_ok(bn_slot.y), d.C
            sl.group_stride = bn_slot.gstride if groups > 1 else 0
            dgb = gsum = None
            # the launch also finishes dgamma / dbeta (last-arriver merge, up_bn_reduce_slot); under ops.deferred_wgrad only for
            # row groups (their backward sums every use into one buffer itself, the ungrouped one uses the accumulating finalize)
            if groups > 1 or not _DEFER["on"]:
                dgb = torch.empty((2, d.C), dtype=torch.float32, device=dev)
                sl.dgamma, sl.dbeta = dgb[0].data_ptr(), dgb[1].data_ptr()
                if groups > 1:
                    gsum = torch.empty((groups, 2, d.C), dtype=torch.float32, device=dev)
                    sl.gsum = gsum.data_ptr()
            ep.bn = C.pointer(sl)
            ep.groups = groups
        if ex_math == MATH_BF16S:
            if d.Kp % 32 or (add is not None and add.dtype != dy.dtype):
                raise NotImplementedError("bf16-storage data gradient needs 32-aligned output channels and a bf16 addend")
            wimg = _packed_bf16(weight, d)[1][0]
        else:
            wimg = packed_dgrad(weight, d)
        _C.check(_C.lib().up_conv2d_bwd_data_ex(C.byref(dd), dy.data_ptr(), wimg.data_ptr(), dx.data_ptr(), C.byref(ep), ex_math,
                                                _stream(dy)), "conv2d_bwd_data_ex")
        if want_slot:
            bn_slot.partial, bn_slot.dz_ptr, bn_slot.dz_version = partial, dx.data_ptr(), dx._version
            bn_slot.dgb = dgb if (dgb is not None and sl.folded) else None
            bn_slot.gsum = gsum if bn_slot.dgb is not None else None
        return dx
    if dy.dtype == torch.bfloat16:
        if d.Kp % 32 or (add is not None and add.dtype != dy.dtype):
            raise NotImplementedError("bf16-storage data gradient needs 32-aligned output channels and a bf16 addend")
        _, wd16 = _packed_bf16(weight, d)
        _C.check(_C.lib().up_conv2d_bwd_data_bf16(C.byref(dd), dy.data_ptr(), wd16[0].data_ptr(), wd16[1].data_ptr(),
                                                  dx.data_ptr(), _ptr(add), ld_add, MATH_BF16S, _stream(dy)),
                 "conv2d_bwd_data_bf16s")
    elif CONV_MATH in (MATH_BF16X3, MATH_BF16, MATH_BF16S) and d.Kp % 32 == 0:
        _, wd16 = _packed_bf16(weight, d)
        _C.check(_C.lib().up_conv2d_bwd_data_bf16(C.byref(dd), dy.data_ptr(), wd16[0].data_ptr(), wd16[1].data_ptr(),
                                                  dx.data_ptr(), _ptr(add), ld_add, min(CONV_MATH, MATH_BF16),
                                                  _stream(dy)), "conv2d_bwd_data_bf16")
    else:
        wd = packed_dgrad(weight, d)
        _C.check(_C.lib().up_conv2d_bwd_data(C.byref(dd), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), _ptr(add),
                                             ld_add, _stream(dy)), "conv2d_bwd_data")
    return dx


class GradLink:
    """Carries the identity-branch gradient of a residual block from the backward of its last stage (which produces
    it) to the backward of its first convolution (which adds it in the data-gradient epilogue), so autograd never
    launches the add of the two gradients of the block input (resnet.py:36-40).  The last stage always runs its
    backward first (its input depends on the first stage's output); a link that is not picked up — the first stage
    needs no input gradient — is returned through autograd as usual."""
    __slots__ = ("grad", "armed", "bits", "masked_ok")

    def __init__(self):
        self.grad = None
        self.armed = False
        # masked hand-over (round 4): when the first convolution's data-gradient kernel can apply a ReLU mask to its addend
        # (masked_ok, set by that convolution's forward), the last stage hands over its UNMASKED dz plus its sign bits instead
        # of writing dz * [z > 0] as a tensor of its own
        self.bits = None
        self.masked_ok = False


def conv_bwd_weight_raw(x, dy, weight_shape, d: _C.ConvDesc, want_bias: bool, ws_tag: str = "main", out=None, accumulate=None,
                        db_out=None):
    """dw (and db) of one convolution.  `out`: an existing fp32 gradient buffer the result is ADDED to by the split-K reduce
    pass itself (up_conv2d_bwd_weight_acc, accumulate = 1): the sum over the uses of a shared weight without an add kernel
    (db_out, when given, is accumulated the same way by the bias gradient's finishing kernel);
    with accumulate=False `out` is simply the destination (a slice of a gradient-exchange bucket, see set_grad_destinations)."""
    if accumulate is None:
        accumulate = out is not None
    dd = _C.ConvDesc.from_buffer_copy(d)
    dd.ldx = _nhwc_ok(x)
    dd.ldy = _nhwc_ok(dy)
    need = _C.lib().up_conv2d_bwd_weight_workspace(C.byref(dd))
    ws = workspace(x.device, need, ws_tag)
    dw = out if out is not None else torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    db = (db_out if db_out is not None else torch.empty(weight_shape[0], dtype=torch.float32, device=x.device)) if want_bias else None
    if x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16:
        math = MATH_BF16S
    elif x.dtype != torch.float32 or dy.dtype != torch.float32:
        raise TypeError(f"weight gradient of mixed element types {x.dtype} / {dy.dtype}")
    else:
        # bf16 operands only where forward and data gradient use them too (32-aligned channel counts): the 3-channel stem and
        # the 15-channel ConvLSTM convolutions stay on the exact fp32 MFMA in every pass
        bf = CONV_MATH in (MATH_BF16, MATH_BF16S) and (WGRAD_BF16_ANY_WIDTH or (dd.Cp % 32 == 0 and dd.Kp % 32 == 0))
        math = MATH_BF16 if bf else MATH_F32
    _C.check(_C.lib().up_conv2d_bwd_weight_acc(C.byref(dd), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(),
                                               ws.numel(), math, int(bool(accumulate)), _stream(x)), "conv2d_bwd_weight")
    return dw, db


# ---- weight gradients on a side stream ------------------------------------------------------------
# dW of a layer has no consumer until the optimizer (or the gradient all-reduce) runs, while dX is on the
# critical path of backward.  The wgrad kernels therefore go to a second HIP stream: their workgroups fill
# the CUs left idle by the tail of the data-gradient kernels and overlap the HBM-bound BatchNorm-backward
# passes of the following layers.  Ordering: side waits for main before each launch (dy, x ready); main
# waits for side (a) at the end of the backward pass (engine callback), (b) before a weight that was already
# seen in this pass returns a second gradient (autograd will ADD the two on the main stream), (c) whenever
# `wgrad_fence()` is called (the data-parallel reducer does before it reads a gradient).
ASYNC_WGRAD = os.environ.get("UNIPOSE_SYNC_WGRAD", "") == ""   # development switch: weight gradients on the main stream
_SIDE = {}
_PASS = {"seen": set(), "task": None, "keep": []}      # task: id of the autograd graph task whose end-of-backward callback is queued
# The side stream reads x and dy of a layer after autograd has dropped them: every such tensor is marked record_stream(side) (the
# caching allocator records an event on the side stream when it is freed and polls it before re-use).  UNIPOSE_KEEP_WGRAD_INPUTS=1
# instead keeps the tensors referenced until the end-of-backward fence has made the main stream wait for the side stream (freed
# once, behind the fence, no events; +18 GB held at B = 32) — measured in round 6: fp32 -0.1 ms, UniPose-LSTM -0.1 ms, bf16
# storage at 736^2 +0.35 ms (profiles/r06_experiments.txt item 9), so the events stay the default.
KEEP_WGRAD_INPUTS = os.environ.get("UNIPOSE_KEEP_WGRAD_INPUTS", "0") != "0"
_DEFER = {"on": False, "acc": {}, "bn": {}, "shared": {}}          # see deferred_wgrad
_SHARED_USES = {}     # id(non-leaf weight) -> forward uses by ConvBias that have not been back-propagated yet (entry dies with the tensor)


def _count_shared_use(weight):
    key = id(weight)
    if key not in _SHARED_USES:
        import weakref
        _SHARED_USES[key] = 0
        weakref.finalize(weight, _SHARED_USES.pop, key, None)
    _SHARED_USES[key] += 1


class deferred_wgrad:
    """``with ops.deferred_wgrad(): loss.backward()`` — for graphs that use a weight more than once (the five-frame unroll of
    the video model, uniposeLSTM.py:116-133).

    The autograd engine sums the gradients of a re-used leaf in its input buffer on the MAIN stream, so without this
    context `conv_bwd_weight` must make the main stream wait for the side stream at every such weight: 4 of the 5 frames
    then run their weight gradients synchronously, followed by one small add per weight and frame.  Inside the context a
    weight-gradient node hands autograd NOTHING: the first gradient of a weight becomes a buffer, later ones are added
    to it on the side stream behind the kernel that produced them, and on exit (the end-of-backward callback has made the
    main stream wait for the side stream by then) every buffer is installed as / added to ``weight.grad``.
    Opt-in because it changes what autograd sees: ``torch.autograd.grad`` w.r.t. such a weight and gradient hooks on it get
    no gradient inside the context (weights with post-accumulate hooks keep the normal path).  A leaf bias travels with its
    weight (round 6: the five head convolutions of the video model, 40 gradient adds and 20 main-waits-for-side per step).
    NON-LEAF weights that several convolutions share (the ConvLSTM cell's stacked gate weights, one cat per clip): every
    forward use is counted (ConvBias.forward); in the backward the uses accumulate into one buffer on the side stream and
    the LAST one hands autograd the sum — one gradient through the cat / pad graph per clip instead of one per frame.  A use
    whose gradient never arrives inside the context is reported on exit (RuntimeError), not dropped."""

    def __enter__(self):
        if _DEFER["on"]:
            raise RuntimeError("ops.deferred_wgrad() does not nest")
        _DEFER["on"], _DEFER["acc"], _DEFER["bn"], _DEFER["shared"] = True, {}, {}, {}
        return self

    def __exit__(self, exc_type, exc, tb):
        acc, bn, shared = _DEFER["acc"], _DEFER["bn"], _DEFER["shared"]
        _DEFER["acc"], _DEFER["bn"], _DEFER["shared"], _DEFER["on"] = {}, {}, {}, False
        if exc_type is None:
            wgrad_fence()                     # (no-op after a completed backward; covers a backward that never ran the callback)
            pending = [e for e in shared.values() if e[2] > 0]
            if pending:
                raise RuntimeError(f"ops.deferred_wgrad: {len(pending)} shared non-leaf weight(s) still wait for "
                                   f"{sum(e[2] for e in pending)} use(s) whose gradient never arrived; their sums were not handed to autograd")
            pairs = []
            for ent in acc.values():
                pairs.append((ent[0], ent[1]))
                if ent[2] is not None:
                    pairs.append((ent[2], ent[3]))
            for gamma, beta, dgb in bn.values():       # BatchNorm affine parameters: sums kept by up_bn_bwd_acc_t
                pairs += [(gamma, dgb[0]), (beta, dgb[1])]
            for p, buf in pairs:
                if p.grad is None:
                    p.grad = buf
                else:
                    p.grad.add_(buf)
        return False


_GRAD_DEST = {"fn": None}


def set_grad_destinations(fn):
    """Data-parallel gradient exchange (unipose_amd.dist): `fn(weight)` returns a FRESH view of the exchange bucket's memory for
    that parameter (or None).  The weight-gradient reduce pass then writes the gradient there directly, autograd installs that
    view as ``weight.grad``, and the bucket needs no gather copy before its all-reduce (190 MB per step otherwise).  None switches
    it off.  Only the first gradient of a weight in a backward pass is redirected; re-used weights keep the normal path.
    Precondition: the redirect is meant for ``loss.backward()`` — with a reducer installed, two ``torch.autograd.grad``
    calls on the same convolution weight (``weight.grad`` stays None there) return tensors that alias the same bucket memory."""
    _GRAD_DEST["fn"] = fn


def cu_mask_stream(dev, mask_bits):
    """A torch stream handle around a HIP stream restricted to the compute units in `mask_bits` (iterable of logical CU indices):
    up_stream_create_cu_mask.  Experiment support (UNIPOSE_SIDE_CUS): the stream is never destroyed."""
    words = (C.c_uint32 * 8)()
    for b in mask_bits:
        words[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    _C.check(_C.lib().up_stream_create_cu_mask(words, 8, C.byref(h)), "stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value, device=dev)


def _side_stream(dev):
    st = _SIDE.get(dev.index)
    if st is None:
        ncu = int(os.environ.get("UNIPOSE_SIDE_CUS", "0"))
        if ncu > 0 and dev.type == "cuda":
            # experiment (profiles/r06_experiments.txt): the weight-gradient stream on `ncu` of the 256 CUs; UNIPOSE_SIDE_CU_STRIDE
            # picks how the bits are spread (1: the lowest ncu bits, k: every k-th bit, wrapping)
            stride = max(1, int(os.environ.get("UNIPOSE_SIDE_CU_STRIDE", "1")))
            bits, b = [], 0
            while len(bits) < ncu:
                if b not in bits:
                    bits.append(b)
                b = (b + stride) % 256
                if b in bits:
                    b = (b + 1) % 256
            st = cu_mask_stream(dev, bits)
        else:
            st = torch.cuda.Stream(device=dev, priority=int(os.environ.get("UNIPOSE_SIDE_PRIO", "0")))
        _SIDE[dev.index] = st
    return st


def wgrad_fence(dev=None):
    """Make the current stream wait for every weight gradient issued so far."""
    want = None
    if dev is not None:                      # torch.device("cuda") has no index: it means the current device
        dev = torch.device(dev)
        if dev.type != "cuda":
            return
        want = dev.index if dev.index is not None else torch.cuda.current_device()
    for idx, st in _SIDE.items():
        if want is None or want == idx:
            torch.cuda.current_stream(st.device).wait_stream(st)


def _end_of_backward():
    wgrad_fence()
    _PASS["seen"].clear()
    _PASS["keep"].clear()          # (behind the fence: whatever re-uses these blocks on the main stream runs after the side stream's reads)
    _PASS["task"] = None


def _hold_for_side(side, *tensors):
    if KEEP_WGRAD_INPUTS:
        _PASS["keep"].extend(tensors)
    else:
        for t in tensors:
            t.record_stream(side)


def _graph_task_id():
    """Id of the running autograd graph task (-1 outside a backward pass).  The engine DROPS queued callbacks when a
    backward raises, so the callback state is keyed on the task: a new task always queues its own fence, whatever an
    aborted earlier pass left behind."""
    try:
        return torch._C._current_graph_task_id()
    except AttributeError:                 # (older torch: fall back to "a backward is running")
        return 0


def _shared_wgrad(x, dy, weight, d, want_bias):
    """ops.deferred_wgrad, non-leaf weight used by several convolutions: accumulate on the side stream, hand over at the last use."""
    dev = dy.device
    key = id(weight)
    uses = _SHARED_USES.get(key, 0)
    if uses > 0:
        _SHARED_USES[key] = uses - 1
    ent = _DEFER["shared"].get(key)
    first = ent is None
    if first and uses <= 1:               # a single use (LSTM_0's gate weights): nothing to share
        return conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias)
    if first:
        ent = [torch.empty(weight.shape, dtype=torch.float32, device=dev),
               torch.empty(weight.shape[0], dtype=torch.float32, device=dev) if want_bias else None, max(uses, 1), weight]
        _DEFER["shared"][key] = ent
    ent[2] -= 1
    if ASYNC_WGRAD and dy.is_cuda and _graph_task_id() != -1:
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        _ensure_end_of_backward(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias, ws_tag="side", out=ent[0], accumulate=not first, db_out=ent[1])
        _hold_for_side(side, x, dy)
        if ent[2] > 0:
            return None, None
        main.wait_stream(side)            # the sum goes to torch operators (cat / pad backward) on the main stream
    else:
        conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias, out=ent[0], accumulate=not first, db_out=ent[1])
        if ent[2] > 0:
            return None, None
    return ent[0], ent[1]


def _ensure_end_of_backward(dev):
    """Queue the end-of-backward fence for the running autograd graph task (once per task); False outside a backward pass."""
    task = _graph_task_id()
    if task == -1:
        return False
    if _PASS["task"] != task:
        if _PASS["task"] is not None:             # an earlier backward died before its callback ran: fence for it now
            wgrad_fence(dev)
            _PASS["seen"].clear()
            _PASS["keep"].clear()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            _PASS["task"] = task
        except RuntimeError:
            return False
    return True


# A/B switches of the round-6 extensions of deferred_wgrad (UniPose-LSTM step, profiles/r06_experiments.txt items 13 / 17): a bias
# travels with its weight, shared non-leaf weights hand over one sum; DEFER_WAIT: main waits for the side stream behind the weight
# gradient of a convolution WITH bias (2: every use, 1: the later uses of a weight, 0: never).  Per-frame head (five uses of the
# video head's 11x11 convolutions): free-running weight gradients 102.4-103.0 ms, with the waits the engine's accumulation used to
# force 99.4-99.9.  With the head batched over the clip (unipose_lstm._unroll_clip: every head weight is used once) 2 / 1 / 0 measure
# 95.4-95.6 / 94.7-95.2 / 94.6-95.1 ms: 1 keeps both.
DEFER_BIAS = os.environ.get("UNIPOSE_DEFER_BIAS", "1") != "0"
DEFER_SHARED = os.environ.get("UNIPOSE_DEFER_SHARED", "1") != "0"
DEFER_WAIT = int(os.environ.get("UNIPOSE_DEFER_WAIT", "1"))


def conv_bwd_weight(x, dy, weight, d, want_bias, bias=None):
    """bias: the bias PARAMETER (for ops.deferred_wgrad, which installs its gradient on exit)."""
    if _DEFER["on"] and DEFER_SHARED and not weight.is_leaf and weight.requires_grad:
        return _shared_wgrad(x, dy, weight, d, want_bias)
    if not (ASYNC_WGRAD and dy.is_cuda and weight.is_leaf):
        # non-leaf weights (the stacked ConvLSTM gate weights) feed torch ops on the main stream right away
        return conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias)
    dev = dy.device
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    task = _graph_task_id()
    if task == -1:                    # not inside an autograd backward pass: stay synchronous
        return conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias)
    if _PASS["task"] != task:
        if _PASS["task"] is not None:             # an earlier backward died before its callback ran: fence for it now
            wgrad_fence(dev)
            _PASS["seen"].clear()
            _PASS["keep"].clear()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            _PASS["task"] = task
        except RuntimeError:
            return conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias)
    # The results are allocated HERE, from the main stream's pool, and written by the side stream behind the wait below; main reads
    # or frees them only behind the end-of-backward fence.  (Rounds 1-5 allocated them inside the side-stream context and marked
    # them record_stream(main): the caching allocator then records one event on the MAIN stream per freed gradient — 229 marker
    # packets = 0.66 ms of main-stream time in every zero_grad, profiles/r06_experiments.txt item 8.)
    bias_ok = not want_bias or (DEFER_BIAS and bias is not None and bias.is_leaf and not getattr(bias, "_post_accumulate_grad_hooks", None))
    defer = _DEFER["on"] and bias_ok and not getattr(weight, "_post_accumulate_grad_hooks", None)
    entry = _DEFER["acc"].get(id(weight)) if defer else None
    dst = None
    if entry is None:
        if _GRAD_DEST["fn"] is not None and not defer and weight.grad is None and id(weight) not in _PASS["seen"]:
            dst = _GRAD_DEST["fn"](weight)
        if dst is None:
            dst = torch.empty(weight.shape, dtype=torch.float32, device=dev)
    db_buf = torch.empty(weight.shape[0], dtype=torch.float32, device=dev) if want_bias else None
    side.wait_stream(main)
    with torch.cuda.stream(side):
        if entry is not None:             # a later use of the weight: the reduce pass adds to the first use's buffer
            conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias, ws_tag="side", out=entry[1], db_out=entry[3])
            _hold_for_side(side, x, dy)
            if want_bias and DEFER_WAIT:
                main.wait_stream(side)
            return None, None
        dw, db = conv_bwd_weight_raw(x, dy, weight.shape, d, want_bias, ws_tag="side", out=dst, accumulate=False, db_out=db_buf)
        if defer:
            _DEFER["acc"][id(weight)] = (weight, dw, bias if want_bias else None, db)
            _hold_for_side(side, x, dy)
            if want_bias and DEFER_WAIT > 1:
                main.wait_stream(side)
            return None, None
    _hold_for_side(side, x, dy)
    key = id(weight)
    if key in _PASS["seen"] or weight.grad is not None:
        main.wait_stream(side)        # autograd is about to accumulate into an earlier, possibly in-flight dW
    _PASS["seen"].add(key)
    return dw, db


_RELU_TRACE = None


def set_relu_trace(lst):
    """Test hook: while a list is installed, every fused ReLU appends its output tensor (NHWC) to it, in
    call order — the oracle replays those sign patterns for flip-free gradient comparisons."""
    global _RELU_TRACE
    _RELU_TRACE = lst


class ConvBias(Function):
    """conv (+bias) (+ReLU) — the LSTM head, gate and final 1x1 convolutions
    (decoder.py:30, model/uniposeLSTM.py:12-14,30-38,85-89,120-124)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg: ConvCfg, relu: bool, out_f32: bool = False):
        y, d, _ = conv_fwd_raw(x, weight, cfg, bias=bias, relu=relu, out_f32=out_f32)
        if relu and _RELU_TRACE is not None:
            _RELU_TRACE.append(y.detach())
        ctx.d, ctx.relu, ctx.has_bias = d, relu, bias is not None
        ctx.bias = bias                       # (the parameter object: deferred_wgrad installs its gradient)
        if ctx.needs_input_grad[1] and not weight.is_leaf:
            _count_shared_use(weight)         # see deferred_wgrad: the last use in the backward hands over the sum
            ctx.shared_w = weight             # (keeps THIS Python object — the key of the count — alive as long as the graph)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _dense(dy)
        if dy.dtype != x.dtype and not ctx.relu:      # fp32 output of a bf16-storage convolution: its gradient goes back to bf16
            dy = dy.to(x.dtype)
        if ctx.relu:
            if dy.dtype != torch.float32:
                # bf16 storage: the only conv + bias + ReLU on bf16 tensors is the video WASP's global-average-pool branch
                # (waspVideo.py:51-53, no BatchNorm there): N x 256 elements — a torch select, not a kernel of its own
                dy = torch.where(y > 0, dy, torch.zeros_like(dy))
            else:
                g = torch.empty_like(dy)
                _C.check(_C.lib().up_relu_bwd(dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), _stream(dy)),
                         "relu_bwd")
                dy = g
        dx = conv_bwd_data_raw(dy, weight, ctx.d, x.shape, x.device) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            # (a shared non-leaf weight: the Python object the forward counted, not a fresh wrapper of the saved tensor)
            dw, db = conv_bwd_weight(x, dy, getattr(ctx, "shared_w", weight), ctx.d, ctx.has_bias and ctx.needs_input_grad[2], bias=ctx.bias)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.reshape(-1, dy.shape[3]).float().sum(0)[:weight.shape[0]]      # fp32 sum for an fp32 bias, also in bf16 storage
        return dx, dw, db, None, None, None


_BN_GROUPS = {"n": 1}
EXACT_STATS_ROWS = 256      # train-mode BatchNorm over at most this many rows per channel: float64 statistics (up_bn_exact_stats_t)


class bn_groups:
    """Inside the context every train-mode conv + BatchNorm normalises `n` equal row groups of its input batch separately
    (frame-major clips: group g = frame g).  The video model uses it to run its trunk ONCE on all T frames of a clip batch —
    every convolution sees a T-times larger batch — while each frame keeps the batch statistics, running-statistics updates
    and gradients of its own module call in the reference loop (uniposeLSTM.py:116-133)."""

    def __init__(self, n: int):
        self.n, self.prev = int(n), 1

    def __enter__(self):
        self.prev, _BN_GROUPS["n"] = _BN_GROUPS["n"], self.n
        return self

    def __exit__(self, exc_type, exc, tb):
        _BN_GROUPS["n"] = self.prev
        return False


class ConvBnAct(Function):
    """conv -> BatchNorm2d -> (+residual) -> (ReLU): Bottleneck.forward resnet.py:22-42, the stem
    :113-116, _AtrousModule wasp.py:16-20, wasp.py:86-88, decoder.py:39-41,52.

    train (batch statistics): conv kernel with Welford partials in its epilogue -> finalize ->
    one apply pass.  eval with grad: same graph with the running statistics."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, rm, rv, cfg: ConvCfg, relu: bool, train: bool, eps: float,
                momentum: float, link_in=None, link_out=None, slot_in=None, slot_out=None, link_dx=None):
        L = _C.lib()
        dev = x.device
        k = weight.shape[0]
        groups = _BN_GROUPS["n"] if train else 1
        if groups > 1:
            # `groups` row groups (frames), each with its own batch statistics: one extra pass over y collects them (a row
            # tile of the convolution may straddle two frames, so its epilogue partials cannot be used)
            if x.shape[0] % groups:
                raise ValueError(f"batch {x.shape[0]} is not a multiple of {groups} BatchNorm groups")
            d0 = make_desc(x, weight, cfg)
            # (a few rows per channel — the branch behind the global average pool — get float64 statistics from y, below)
            fused = conv_fwd_grouped_raw(x, weight, cfg, groups) if (d0.N * d0.P * d0.Q) // groups > EXACT_STATS_ROWS else None
            if fused is not None:                 # every group tiled on its own: the epilogue's partials ARE per group
                y, d, st = fused
                tiles = st.shape[1]
            else:
                y, d, _ = conv_fwd_raw(x, weight, cfg)
            rows = d.N * d.P * d.Q
            rpg = rows // groups
            if rpg <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                                 f"{(d.N // groups, k, d.P, d.Q)}")
            coef = torch.empty((groups, 4, k), dtype=torch.float32, device=dev)   # per group: mean, invstd, scale, shift
            done = False
            if fused is None:
                tiles = 1 if rpg <= EXACT_STATS_ROWS else L.up_bn_batch_stats_tiles(rpg)
                st = torch.empty((groups, tiles, k, 3), dtype=torch.float32, device=dev)
                if rpg <= EXACT_STATS_ROWS:
                    _C.check(L.up_bn_exact_stats_t(y.data_ptr(), d.ldy, rpg, k, groups, _dt(y), st.data_ptr(), _stream(x)), "bn_exact_stats")
                else:       # statistics pass + finalize of every group in ONE launch (the pass's last workgroups merge, bn_fold.h)
                    _C.check(L.up_bn_stats_groups_t(y.data_ptr(), d.ldy, rpg, k, groups, _dt(y), st.data_ptr(), eps, momentum,
                                                    _ptr(rm), _ptr(rv), gamma.data_ptr(), beta.data_ptr(), coef.data_ptr(),
                                                    _stream(x)), "bn_stats_groups")
                    done = True
            if not done:
                _C.check(L.up_bn_finalize_groups(st.data_ptr(), tiles, k, groups, rpg, eps, momentum, _ptr(rm), _ptr(rv),
                                                 gamma.data_ptr(), beta.data_ptr(), coef.data_ptr(), _stream(x)), "bn_finalize_groups")
        elif train:
            small = x.shape[0] * ((x.shape[1] + 2 * cfg.pad - cfg.dil * (weight.shape[2] - 1) - 1) // cfg.stride + 1) * \
                ((x.shape[2] + 2 * cfg.pad - cfg.dil * (weight.shape[3] - 1) - 1) // cfg.stride + 1) <= EXACT_STATS_ROWS
            coef = torch.empty((4, k), dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
            # the convolution's last workgroup per channel column merges the partials and writes coef + the running statistics
            # itself (up_bn_fold); `folded` = 0 (fold switched off, scratch exhausted): the stand-alone merge below, same bits
            fold = None
            if not small:
                fold = _C.BnFold()
                fold.eps, fold.momentum, fold.running_mean, fold.running_var = eps, momentum, _ptr(rm), _ptr(rv)
                fold.gamma, fold.beta = gamma.data_ptr(), beta.data_ptr()
                base, step = coef.data_ptr(), 4 * k
                fold.mean, fold.invstd, fold.scale, fold.shift = base, base + step, base + 2 * step, base + 3 * step
            y, d, st = conv_fwd_raw(x, weight, cfg, stats=not small, fold=fold)
            rows = d.N * d.P * d.Q
            if rows <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                                 f"{(d.N, k, d.P, d.Q)}")          # F.batch_norm's check
            if small:      # a handful of samples per channel (the global-average-pool branch): float64 statistics, see the kernel
                assert rows <= EXACT_STATS_ROWS
                st = torch.empty((1, k, 3), dtype=torch.float32, device=dev)
                _C.check(L.up_bn_exact_stats_t(y.data_ptr(), d.ldy, rows, k, 1, _dt(y), st.data_ptr(), _stream(x)), "bn_exact_stats")
            if fold is not None and fold.folded:
                HOST_COUNTERS["bn_fwd_folded"] += 1
            else:
                _C.check(L.up_bn_finalize(st.data_ptr(), st.shape[0], k, eps, momentum, _ptr(rm), _ptr(rv),
                                          gamma.data_ptr(), beta.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                          coef[2].data_ptr(), coef[3].data_ptr(), _stream(x)), "bn_finalize")
        else:
            y, d, _ = conv_fwd_raw(x, weight, cfg)
            rows = d.N * d.P * d.Q
            coef = torch.empty((4, k), dtype=torch.float32, device=dev)
            coef[0].copy_(rm)
            torch.rsqrt(rv + eps, out=coef[1])
            _C.check(L.up_bn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), eps, k,
                                         coef[2].data_ptr(), coef[3].data_ptr(), _stream(x)), "bn_eval_coeffs")
        z = torch.empty_like(y) if d.ldy == k else torch.zeros_like(y)     # pad channels must read as zeros
        # sign bits of z for the backward passes (1/32 of re-reading z there); only when a backward can follow
        bits = torch.empty(((rows * k + 31) // 32,), dtype=torch.int32, device=dev) \
            if relu and any(ctx.needs_input_grad[:5]) else None
        if residual is not None and residual.dtype != y.dtype:
            raise TypeError(f"residual {residual.dtype} vs convolution output {y.dtype}")
        if groups > 1:
            _C.check(L.up_bn_apply_groups_t(y.data_ptr(), d.ldy, coef.data_ptr(), beta.data_ptr(), _ptr(residual),
                                            _nhwc_ok(residual) if residual is not None else 0, int(relu), z.data_ptr(), d.ldy,
                                            _ptr(bits), rows // groups, k, groups, _dt(y), _stream(x)), "bn_apply_groups")
        else:
            # the centred form ATen evaluates, (y - mean) * (gamma * invstd) + beta: see up_bn_apply_centered_t
            _C.check(L.up_bn_apply_centered_t(y.data_ptr(), d.ldy, coef[0].data_ptr(), coef[2].data_ptr(), beta.data_ptr(),
                                              _ptr(residual), _nhwc_ok(residual) if residual is not None else 0, int(relu),
                                              z.data_ptr(), d.ldy, _ptr(bits), rows, k, _dt(y), _stream(x)), "bn_apply")
        if relu and _RELU_TRACE is not None:
            _RELU_TRACE.append(z.detach())
        ctx.d, ctx.relu, ctx.train, ctx.has_res, ctx.groups = d, relu, train, residual is not None, groups
        ctx.beta = beta                       # (the parameter object: deferred_wgrad installs its gradient)
        ctx.link_in, ctx.link_out = link_in, link_out
        if link_in is not None:
            link_in.armed = True       # this node will compute a data gradient: the producer may hand over
            link_in.masked_ok = MASKED_ADDEND and dgrad_extras_tiles(d, x.shape, x.dtype) > 0
        # BatchNorm-backward reduction by the consumer's data gradient (BnSlot): this layer as the producer ...
        # link_dx: this node's OWN data gradient is handed to the node behind the link (the block's first convolution, which reads
        # the same x: resnet.py:36-37 downsample(x) next to conv1(x)) instead of to autograd, whose engine would add the two
        ctx.link_dx = link_dx if (DX_HANDOVER and link_dx is not None and ctx.needs_input_grad[0]) else None
        ctx.slot_out = None
        if slot_out is not None:
            slot_out.clear()
            if BN_FUSE_REDUCE and groups == 1 and d.ldy == k and (bits is not None or not relu) and \
                    ((y.dtype == torch.float32 and CONV_MATH == MATH_F32) or y.dtype == torch.bfloat16):
                slot_out.y, slot_out.bits, slot_out.mean, slot_out.invstd, slot_out.C = y, bits, coef[0], coef[1], k
                ctx.slot_out = slot_out
            elif BN_FUSE_REDUCE and GROUPED_REDUCE and 1 < groups <= 8 and d.ldy == k and (bits is not None or not relu) and \
                    y.dtype == torch.float32 and CONV_MATH == MATH_F32 and \
                    L.up_bn_bwd_groups_prereduced_ok((d.N * d.P * d.Q) // groups, k, groups, d.ldy):
                # row groups: coef is [groups][4][k] (mean, invstd, scale, shift per group)
                slot_out.y, slot_out.bits, slot_out.mean, slot_out.invstd, slot_out.C = y, bits, coef[0, 0], coef[0, 1], k
                slot_out.groups, slot_out.gstride = groups, 4 * k
                ctx.slot_out = slot_out
        # ... and as the consumer of the layer that produced x
        ctx.slot_in = slot_in if (slot_in is not None and slot_in.y is not None and ctx.needs_input_grad[0]) else None
        # a tensor hook on z can edit dz IN PLACE through .data without moving its version counter (g.data.mul_(2), a clipping
        # hook): BnSlot.matches cannot see that, so a hooked output always takes the separate reduction: conv_bn_act
        # attaches z's hook dictionary to this node (ctx.z_hooks) once z has its grad_fn
        ctx.save_for_backward(x, weight, gamma, y, bits, coef)
        return z

    @staticmethod
    def _slot_usable(ctx, dz, y):
        so = ctx.slot_out
        if so is None or not so.matches(dz, y):
            return False
        # tensor hooks of z: conv_bn_act registered z's hook dictionary with the node right after the forward and keeps it on the
        # node, so hooks added later are seen here even when the Python tensor is long gone (`t = block(x); t.register_hook(h);
        # return next_block(t)`: a dead weak reference said nothing about them)
        hooks = getattr(ctx, "z_hooks", None)
        return hooks is not None and len(hooks) == 0

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, weight, gamma, y, bits, coef = ctx.saved_tensors
        L, d = _C.lib(), ctx.d
        dz = _dense(dz)
        k = weight.shape[0]
        rows = d.N * d.P * d.Q
        fresh = torch.empty_like if d.ldy == k else torch.zeros_like         # pad channels meet zero weights: keep them finite
        dy = fresh(y)
        # the skip gradient dz * [z > 0]: not materialised when the block's first convolution masks its addend itself
        lo = ctx.link_out
        masked = ctx.has_res and lo is not None and lo.armed and lo.masked_ok and ctx.relu and bits is not None and d.ldy == k \
            and (ctx.groups == 1 or GROUPED_REDUCE)      # (row groups: the sign bits are indexed by the global row, like dz)
        dres = fresh(y) if ctx.has_res and not masked else None
        dgb = torch.empty((2, k), dtype=torch.float32, device=x.device)
        if dz.dtype != y.dtype:
            raise TypeError(f"gradient {dz.dtype} vs saved convolution output {y.dtype}")
        if ctx.groups > 1:                    # per-group data gradient, parameter gradients summed over the groups
            rpg = rows // ctx.groups
            ws = workspace(x.device, L.up_bn_bwd_groups_workspace(rpg, k, ctx.groups))
            so = ctx.slot_out
            if so is not None and so.groups == ctx.groups and ConvBnAct._slot_usable(ctx, dz, y):
                # the data-gradient launch that wrote dz (tiled per group) already reduced every group's sums
                partial, so.partial, so.dz_ptr, so.dz_version = so.partial, None, 0, -1
                HOST_COUNTERS["bn_prereduced"] += 1
                done, gsum, so.dgb, so.gsum = so.dgb, so.gsum, None, None
                if done is not None and gsum is not None:      # ... and merged them: the apply pass alone
                    HOST_COUNTERS["bn_bwd_folded"] += 1
                    dgb = done
                    _C.check(L.up_bn_bwd_groups_finalized_t(dz.data_ptr(), d.ldy, _ptr(bits), y.data_ptr(), d.ldy, gamma.data_ptr(),
                                                            coef.data_ptr(), int(ctx.relu), dy.data_ptr(), d.ldy, _ptr(dres), d.ldy,
                                                            gsum.data_ptr(), rpg, k, ctx.groups, _dt(y), _stream(x)),
                             "bn_bwd_groups_finalized")
                    return ConvBnAct._finish_backward(ctx, x, weight, d, dy, dres, dgb, True, (dz, bits) if masked else None)
                _C.check(L.up_bn_bwd_groups_prereduced_t(dz.data_ptr(), d.ldy, _ptr(bits), y.data_ptr(), d.ldy, gamma.data_ptr(),
                                                         coef.data_ptr(), int(ctx.relu), dy.data_ptr(), d.ldy, _ptr(dres), d.ldy,
                                                         dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(), ws.numel(),
                                                         partial.data_ptr(), partial.shape[0] // ctx.groups, rpg, k, ctx.groups,
                                                         _dt(y), _stream(x)), "bn_bwd_groups_prereduced")
                return ConvBnAct._finish_backward(ctx, x, weight, d, dy, dres, dgb, True, (dz, bits) if masked else None)
            _C.check(L.up_bn_bwd_groups_t(dz.data_ptr(), d.ldy, _ptr(bits), y.data_ptr(), d.ldy, gamma.data_ptr(), coef.data_ptr(),
                                          int(ctx.relu), dy.data_ptr(), d.ldy, _ptr(dres), d.ldy, dgb[0].data_ptr(),
                                          dgb[1].data_ptr(), ws.data_ptr(), ws.numel(), rpg, k, ctx.groups, _dt(y), _stream(x)),
                     "bn_bwd_groups")
            return ConvBnAct._finish_backward(ctx, x, weight, d, dy, dres, dgb, True, (dz, bits) if masked else None)
        need = L.up_bn_bwd_workspace(rows, k)
        ws = workspace(x.device, need)
        beta = ctx.beta
        hand_over = True                      # autograd gets dgamma / dbeta
        acc = None
        if _DEFER["on"] and gamma.is_leaf and beta.is_leaf and ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and \
                not getattr(gamma, "_post_accumulate_grad_hooks", None) and not getattr(beta, "_post_accumulate_grad_hooks", None):
            hand_over = False                 # ops.deferred_wgrad: the sums over all uses live in ONE buffer per layer
            entry = _DEFER["bn"].get(id(gamma))
            if entry is None:
                _DEFER["bn"][id(gamma)] = (gamma, beta, dgb)
            else:
                acc = entry[2]
        so = ctx.slot_out
        if ConvBnAct._slot_usable(ctx, dz, y):
            # the data-gradient launch that wrote dz already reduced this layer's sums (BnSlot): finalize + apply
            partial, so.partial, so.dz_ptr, so.dz_version = so.partial, None, 0, -1
            HOST_COUNTERS["bn_prereduced"] += 1
            done, so.dgb = so.dgb, None
            if done is not None and acc is None:      # ... and merged them: dgamma / dbeta are final, the apply pass alone
                HOST_COUNTERS["bn_bwd_folded"] += 1
                dgb = done
                _C.check(L.up_bn_bwd_finalized_t(dz.data_ptr(), d.ldy, _ptr(bits), y.data_ptr(), d.ldy, gamma.data_ptr(),
                                                 coef[0].data_ptr(), coef[1].data_ptr(), int(ctx.relu), int(ctx.train), dy.data_ptr(),
                                                 d.ldy, _ptr(dres), d.ldy, dgb[0].data_ptr(), dgb[1].data_ptr(), rows, k, _dt(y),
                                                 _stream(x)), "bn_bwd_finalized")
                return ConvBnAct._finish_backward(ctx, x, weight, d, dy, dres, dgb, hand_over, (dz, bits) if masked else None)
            _C.check(L.up_bn_bwd_prereduced_t(dz.data_ptr(), d.ldy, _ptr(bits), y.data_ptr(), d.ldy, gamma.data_ptr(),
                                              coef[0].data_ptr(), coef[1].data_ptr(), int(ctx.relu), int(ctx.train), dy.data_ptr(),
                                              d.ldy, _ptr(dres), d.ldy, dgb[0].data_ptr(), dgb[1].data_ptr(),
                                              acc[0].data_ptr() if acc is not None else None,
                                              acc[1].data_ptr() if acc is not None else None, partial.data_ptr(),
                                              partial.shape[0], rows, k, _dt(y), _stream(x)), "bn_bwd_prereduced")
            return ConvBnAct._finish_backward(ctx, x, weight, d, dy, dres, dgb, hand_over, (dz, bits) if masked else None)
        _C.check(L.up_bn_bwd_acc_t(dz.data_ptr(), d.ldy, None, 0, _ptr(bits), y.data_ptr(), d.ldy, gamma.data_ptr(),
                                   coef[0].data_ptr(), coef[1].data_ptr(), int(ctx.relu), int(ctx.train), dy.data_ptr(),
                                   d.ldy, _ptr(dres), d.ldy, dgb[0].data_ptr(), dgb[1].data_ptr(),
                                   acc[0].data_ptr() if acc is not None else None,
                                   acc[1].data_ptr() if acc is not None else None, ws.data_ptr(),
                                   ws.numel(), rows, k, _dt(y), _stream(x)), "bn_bwd")
        return ConvBnAct._finish_backward(ctx, x, weight, d, dy, dres, dgb, hand_over, (dz, bits) if masked else None)

    @staticmethod
    def _finish_backward(ctx, x, weight, d, dy, dres, dgb, hand_over, masked_skip=None):
        add = add_bits = None
        if ctx.link_in is not None:
            li = ctx.link_in
            add, add_bits, li.grad, li.bits, li.armed = li.grad, li.bits, None, None, False
        si = ctx.slot_in
        if si is not None and ctx.link_in is not None and add is None:
            si = None       # the skip gradient was not handed over: autograd will ADD it to dx, which is then not dz yet
        if ctx.needs_input_grad[0]:
            dx = conv_bwd_data_raw(dy, weight, d, x.shape, x.device, add, bn_slot=si, add_bits=add_bits)
        else:
            dx = add if add_bits is None else None      # (armed links belong to nodes that compute a data gradient)
        ld = ctx.link_dx
        if ld is not None and ld.armed and dx is not None and ld.grad is None:
            ld.grad, dx = dx, None                        # the consumer has not run yet (it disarms the link when it does): it adds
            HOST_COUNTERS["dx_handed_over"] += 1
        if masked_skip is not None:                       # unmasked dz + sign bits: the first convolution masks
            ctx.link_out.grad, ctx.link_out.bits = masked_skip
        elif ctx.link_out is not None and ctx.link_out.armed and dres is not None:
            ctx.link_out.grad, dres = dres, None          # the block's first convolution adds it to ITS dx
        dw = conv_bwd_weight(x, dy, weight, d, False)[0] if ctx.needs_input_grad[1] else None   # frozen weight: no launch
        return dx, dw, (dgb[0] if hand_over else None), (dgb[1] if hand_over else None), dres, None, None, None, None, None, \
            None, None, None, None, None, None, None


def conv_bn_act_eval_fused(x, weight, gamma, beta, rm, rv, cfg, relu, residual=None, eps=BN_EPS_DEFAULT):
    """Inference fast path: BatchNorm folded into the convolution epilogue, ONE kernel per layer."""
    k = weight.shape[0]
    coef = torch.empty((2, k), dtype=torch.float32, device=x.device)
    _C.check(_C.lib().up_bn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), eps, k,
                                        coef[0].data_ptr(), coef[1].data_ptr(), _stream(x)), "bn_eval_coeffs")
    y, _, _ = conv_fwd_raw(x, weight, cfg, scale=coef[0], shift=coef[1], residual=residual, relu=relu)
    return y


_BN_COUNT = {"mode": None, "seen": None}


class bn_counters:
    """``with ops.bn_counters(model):`` around a model forward: nn.BatchNorm2d bumps ``num_batches_tracked`` once per
    training forward — 113 one-element kernels per step here.  The first forward records which counters the pass
    touches (decoder.bn2 is never called, SURVEY D9, and must stay 0 like in the reference); later forwards with the
    same train/eval pattern bump all of them with ONE multi-tensor add."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        if _BN_COUNT["mode"] is not None:          # nested: the outer context owns the pass
            self.own = False
            return self
        self.own = True
        bns = self.module.__dict__.get("_up_bn_list")
        if bns is None:                            # the BatchNorm layers of a model do not change: walk the tree once
            bns = self.module.__dict__["_up_bn_list"] = [m for m in self.module.modules()
                                                         if isinstance(m, torch.nn.BatchNorm2d)]
        self.sig = tuple(m.training for m in bns)
        plan = self.module.__dict__.get("_up_bn_plan")
        if plan is not None and plan[0] == self.sig:
            if plan[1]:
                torch._foreach_add_(plan[1], _BN_GROUPS["n"] if any(self.sig) else 1)
            _BN_COUNT["mode"] = "skip"
        else:
            _BN_COUNT["mode"], _BN_COUNT["seen"] = "record", []
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.own:
            if _BN_COUNT["mode"] == "record" and exc_type is None:
                self.module.__dict__["_up_bn_plan"] = (self.sig, list(_BN_COUNT["seen"]))
            _BN_COUNT["mode"], _BN_COUNT["seen"] = None, None
        return False


class FoldedBatchNorm(torch.nn.Identity):
    """Stands where a BatchNorm2d stood after ``checkpoint.load_folded``: its affine map lives in the preceding
    convolution's weight and bias (inference export).  No parameters, no buffers, inference only."""


def conv_bn_act(x, conv, bn, relu=True, residual=None, link_in=None, link_out=None, slot_in=None, slot_out=None, link_dx=None):
    """Dispatch on (bn.training, grad mode) exactly like nn.BatchNorm2d would.  link_in / link_out: see GradLink;
    slot_in / slot_out: see BnSlot (slot_in: the slot the producer of x filled; slot_out: filled here for x's ONE consumer)."""
    weight, cfg = conv.weight, ConvCfg(conv.stride[0], conv.padding[0], conv.dilation[0])
    if isinstance(bn, FoldedBatchNorm):       # BN-folded network: ONE kernel, conv + bias (+ residual) (+ ReLU) epilogue
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
            raise NotImplementedError("a BatchNorm-folded network is an inference export: run it under torch.no_grad()")
        return conv_fwd_raw(x, weight, cfg, bias=conv.bias, residual=residual, relu=relu)[0]
    need_grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or bn.weight.requires_grad)
    train = bn.training
    if not train and not need_grad:
        return conv_bn_act_eval_fused(x, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, cfg, relu,
                                      residual, bn.eps)
    if train and _BN_GROUPS["n"] > 1 and bn.momentum is None:
        raise NotImplementedError("grouped BatchNorm (ops.bn_groups) with momentum=None (cumulative average)")
    if train and bn.track_running_stats and _BN_COUNT["mode"] != "skip":
        bn.num_batches_tracked.add_(_BN_GROUPS["n"])            # one "batch" per group, like separate module calls
        if _BN_COUNT["mode"] == "record":
            _BN_COUNT["seen"].append(bn.num_batches_tracked)
    if bn.momentum is not None:
        mom = bn.momentum
    elif train and bn.track_running_stats:      # nn.BatchNorm2d(momentum=None): cumulative average, factor 1 / batches seen
        mom = 1.0 / float(max(int(bn.num_batches_tracked.item()), 1))      # (already bumped above / by bn_counters)
    else:
        mom = 0.0
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    z = ConvBnAct.apply(x, weight, bn.weight, bn.bias, residual, rm, rv, cfg, relu, train, bn.eps, mom,
                        link_in, link_out, slot_in, slot_out, link_dx)
    node = z.grad_fn
    if node is not None and getattr(node, "slot_out", None) is not None:
        # the node watches z's tensor hooks through the dictionary torch itself files them in (Tensor.register_hook): created
        # and registered here, while z is certainly alive, and kept on the node — see ConvBnAct._slot_usable
        hooks = z._backward_hooks
        if hooks is None:
            hooks = z._backward_hooks = collections.OrderedDict()
            node._register_hook_dict(z)
        node.z_hooks = hooks
    return z


def conv_bias_act(x, conv, relu=False, out_f32=False):
    """out_f32: the network's last convolution — in bf16 storage its output (the heat-maps) is written as fp32"""
    cfg = ConvCfg(conv.stride[0], conv.padding[0], conv.dilation[0])
    return ConvBias.apply(x, conv.weight, conv.bias, cfg, relu, out_f32)


# --------------------------------------------------------------------------------------------
# layout, pooling, resampling, concat, dropout, loss
# --------------------------------------------------------------------------------------------
class SplitBatch(Function):
    """(n * b, ...) -> n tensors of b leading rows each (views).  Plain slicing gives every part a slice_backward node: zeros +
    copy per part and n - 1 full-size gradient adds in the backward (the four WASP branches behind the batched conv2: 4 fills,
    4 copies and 3 adds of 69 MB tensors per step; the five frames of the video model's batched trunk); here the n gradients are
    concatenated once."""

    @staticmethod
    def forward(ctx, x, n):
        b = x.shape[0] // n
        ctx.b, ctx.tail = b, x.shape[1:]
        return tuple(x.narrow(0, i * b, b) for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        like = next((g for g in grads if g is not None), None)
        if like is None:
            return None, None
        parts = [g if g is not None else torch.zeros((ctx.b,) + tuple(ctx.tail), dtype=like.dtype, device=like.device) for g in grads]
        return torch.cat(parts, 0), None


class ToNHWC(Function):
    @staticmethod
    def forward(ctx, x):
        _dev_ok(x)
        x = _dense(x)
        n, c, h, w = x.shape
        y = torch.empty((n, h, w, rup4(c)), dtype=torch.float32, device=x.device)
        _C.check(_C.lib().up_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), n, c, h, w, rup4(c), _stream(x)), "nchw_to_nhwc")
        ctx.c = c
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _dense(dy)
        n, h, w, ld = dy.shape
        dx = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=dy.device)
        _C.check(_C.lib().up_nhwc_to_nchw(dy.data_ptr(), ld, dx.data_ptr(), n, ctx.c, h, w, _stream(dy)), "nhwc_to_nchw")
        return dx


class ToNCHW(Function):
    @staticmethod
    def forward(ctx, x, c: int):
        _dev_ok(x)
        n, h, w, _ = x.shape
        ld = _nhwc_ok(x)
        y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)      # the network's output is always fp32
        _C.check(_C.lib().up_nhwc_to_nchw_t(x.data_ptr(), ld, y.data_ptr(), n, c, h, w, _dt(x), _stream(x)),
                 "nhwc_to_nchw")
        ctx.ld, ctx.dtype = x.shape[3], x.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _dense(dy)
        n, c, h, w = dy.shape
        dx = torch.empty((n, h, w, ctx.ld), dtype=ctx.dtype, device=dy.device)
        _C.check(_C.lib().up_nchw_to_nhwc_t(dy.data_ptr(), dx.data_ptr(), n, c, h, w, ctx.ld, _dt(dx), _stream(dy)),
                 "nchw_to_nhwc")
        return dx, None


class MaxPool3s2(Function):
    """nn.MaxPool2d(3, 2, 1): resnet.py:65,117; decoder.py:33,47."""

    @staticmethod
    def forward(ctx, x, out_dtype=None):
        """`out_dtype`: element type of the result (default: that of x) — the pool behind the fp32 stem is where the
        bf16-storage network changes its element type."""
        _dev_ok(x)
        n, h, w, c = x.shape
        p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, p, q, c), dtype=out_dtype or x.dtype, device=x.device)
        if y.dtype == torch.bfloat16 and c % 8:
            raise ValueError(f"bf16 tensors need channel counts that are multiples of 8, got {c}")
        idx = torch.empty((n, p, q, c), dtype=torch.uint8, device=x.device)
        _C.check(_C.lib().up_maxpool3s2_fwd_t(x.data_ptr(), _nhwc_ok(x), y.data_ptr(), c, idx.data_ptr(), n, h, w, c,
                                              p, q, _dt(x), _dt(y), _stream(x)), "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.hw, ctx.in_dtype = (h, w), x.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = _dense(dy)
        n, p, q, c = dy.shape
        h, w = ctx.hw
        dx = torch.empty((n, h, w, c), dtype=ctx.in_dtype, device=dy.device)
        _C.check(_C.lib().up_maxpool3s2_bwd_t(dy.data_ptr(), c, idx.data_ptr(), dx.data_ptr(), c, n, h, w, c, p, q,
                                              _dt(dy), _dt(dx), _stream(dy)), "maxpool_bwd")
        return dx, None


class Bilinear(Function):
    """F.interpolate(mode='bilinear', align_corners=True): wasp.py:83, decoder.py:49, model/unipose.py:32."""

    @staticmethod
    def forward(ctx, x, p: int, q: int):
        _dev_ok(x)
        n, h, w, c = x.shape
        y = torch.empty((n, p, q, c), dtype=x.dtype, device=x.device)
        _C.check(_C.lib().up_bilinear_fwd_t(x.data_ptr(), _nhwc_ok(x), y.data_ptr(), c, n, h, w, c, p, q, _dt(x),
                                            _stream(x)), "bilinear_fwd")
        ctx.hw = (h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _dense(dy)
        n, p, q, c = dy.shape
        h, w = ctx.hw
        dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
        _C.check(_C.lib().up_bilinear_bwd_t(dy.data_ptr(), c, dx.data_ptr(), c, n, h, w, c, p, q, _dt(dy), _stream(dy)),
                 "bilinear_bwd")
        return dx, None, None


class GlobalAvgPool(Function):
    """nn.AdaptiveAvgPool2d((1,1)): wasp.py:51 -> (N,1,1,C)."""

    @staticmethod
    def forward(ctx, x):
        _dev_ok(x)
        n, h, w, c = x.shape
        y = torch.empty((n, 1, 1, c), dtype=x.dtype, device=x.device)
        _C.check(_C.lib().up_gap_fwd_t(x.data_ptr(), _nhwc_ok(x), y.data_ptr(), n, h * w, c, _dt(x), _stream(x)),
                 "gap_fwd")
        ctx.hw = (h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _dense(dy)
        n, _, _, c = dy.shape
        h, w = ctx.hw
        dx = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
        _C.check(_C.lib().up_gap_bwd_t(dy.data_ptr(), dx.data_ptr(), c, n, h * w, c, _dt(dy), _stream(dy)), "gap_bwd")
        return dx


def _copy2d(src, lds, soff, dst, ldd, doff, rows, c):
    """Strided copy of `c` channels per pixel row (element offsets soff / doff).  The kernel moves 4-byte words: a bf16
    tensor is copied as c/2 words per row (channel counts and offsets of bf16 tensors are multiples of 8)."""
    if src.dtype != dst.dtype:
        raise TypeError(f"copy2d: {src.dtype} -> {dst.dtype}")
    es = src.element_size()
    w = 4 // es                       # elements per 4-byte word
    if c % w or lds % w or ldd % w or soff % w or doff % w:
        raise ValueError("copy2d: bf16 rows must be whole 4-byte words")
    _C.check(_C.lib().up_copy2d(src.data_ptr() + es * soff, lds // w, dst.data_ptr() + es * doff, ldd // w, rows, c // w,
                                _stream(src)), "copy2d")


class ConcatC(Function):
    """torch.cat(dim=1) of NCHW == channel concat of NHWC (wasp.py:84, decoder.py:51,
    model/uniposeLSTM.py:116): strided copies into slices of one buffer; backward slices."""

    @staticmethod
    def forward(ctx, ld_out: int, *xs):
        _dev_ok(*xs)
        n, h, w, _ = xs[0].shape
        widths = [t.shape[3] for t in xs]
        tot = sum(widths)
        alloc = torch.zeros if ld_out > tot else torch.empty
        y = alloc((n, h, w, max(ld_out, tot)), dtype=xs[0].dtype, device=xs[0].device)
        off = 0
        for t, c in zip(xs, widths):
            _copy2d(t, _nhwc_ok(t), 0, y, y.shape[3], off, n * h * w, c)
            off += c
        ctx.widths = widths
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _dense(dy)
        n, h, w, ld = dy.shape
        outs, off = [], 0
        for i, c in enumerate(ctx.widths):
            if ctx.needs_input_grad[i + 1]:
                g = torch.empty((n, h, w, c), dtype=dy.dtype, device=dy.device)
                _copy2d(dy, ld, off, g, c, 0, n * h * w, c)
                outs.append(g)
            else:
                outs.append(None)
            off += c
        return (None, *outs)


class Dropout(Function):
    """nn.Dropout (wasp.py:63,90; decoder.py:25,29).  `ext_mask` (float 0/1, same shape) injects the
    keep decisions for parity tests; otherwise a counter hash of (seed, element index) is used."""

    @staticmethod
    def forward(ctx, x, p: float, seed: int, ext_mask):
        _dev_ok(x, ext_mask)
        x = _dense(x)
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        step = _DROPOUT_STATE["step_dev"]        # (a device counter while a training step is captured / replayed, see graph.py)
        _C.check(_C.lib().up_dropout_fwd_step_t(x.data_ptr(), y.data_ptr(), mask.data_ptr(), _ptr(ext_mask), x.numel(),
                                                float(p), int(seed) & (2 ** 64 - 1), _ptr(step), _dt(x), _stream(x)), "dropout_fwd")
        ctx.p = float(p)
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _dense(dy)
        dx = torch.empty_like(dy)
        _C.check(_C.lib().up_dropout_bwd_t(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), ctx.p, _dt(dy),
                                           _stream(dy)), "dropout_bwd")
        return dx, None, None, None


_DROPOUT_STATE = {"seed": 0x5EED, "calls": 0, "ext": None, "step_dev": None}


def set_dropout_masks(masks):
    """Test hook: list of float (0/1) NHWC masks consumed in call order (None restores the hash RNG)."""
    _DROPOUT_STATE["ext"] = list(masks) if masks is not None else None


def dropout(x, module):
    if not module.training or module.p == 0.0:
        return x
    if module.p >= 1.0:
        raise NotImplementedError("dropout p >= 1")
    ext = None
    if _DROPOUT_STATE["ext"]:
        ext = _DROPOUT_STATE["ext"].pop(0)
    _DROPOUT_STATE["calls"] += 1
    seed = _DROPOUT_STATE["seed"] * 1000003 + _DROPOUT_STATE["calls"]
    return Dropout.apply(x, module.p, seed, ext)


def manual_seed(seed: int):
    _DROPOUT_STATE["seed"], _DROPOUT_STATE["calls"] = int(seed), 0


class MSELoss(Function):
    """nn.MSELoss() (mean) as used at unipose.py:70,117."""

    @staticmethod
    def forward(ctx, y, t):
        _dev_ok(y, t)
        y, t = _dense(y), _dense(t)
        if y.shape != t.shape:
            raise ValueError(f"mse: shape mismatch {tuple(y.shape)} vs {tuple(t.shape)}")
        loss = torch.empty(1, dtype=torch.float32, device=y.device)
        ws = torch.empty(_C.lib().up_mse_workspace(y.numel()) // 4, dtype=torch.float32, device=y.device)
        _C.check(_C.lib().up_mse_fwd(y.data_ptr(), t.data_ptr(), loss.data_ptr(), ws.data_ptr(), y.numel(),
                                     _stream(y)), "mse_fwd")
        ctx.save_for_backward(y, t)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        y, t = ctx.saved_tensors
        dl = dl.reshape(1).contiguous()
        dy = torch.empty_like(y)
        _C.check(_C.lib().up_mse_bwd(y.data_ptr(), t.data_ptr(), dl.data_ptr(), dy.data_ptr(), y.numel(),
                                     _stream(y)), "mse_bwd")
        return dy, None


def mse_loss(y, t):
    return MSELoss.apply(y, t)


def avgpool9s8_into(center_nchw, out_nhwc, coff: int):
    """nn.AvgPool2d(9, 8, 1) of the (N,1,H,W) centre map into channel `coff` of an NHWC buffer
    (model/uniposeLSTM.py:114).  No gradient: the centre map is an input."""
    _dev_ok(center_nchw, out_nhwc)
    c = _dense(center_nchw)
    n, one, h, w = c.shape
    assert one == 1
    _, p, q, ld = out_nhwc.shape
    _C.check(_C.lib().up_avgpool9s8_fwd(c.data_ptr(), out_nhwc.data_ptr(), ld, coff, n, h, w, p, q, _stream(c)),
             "avgpool9s8")


# --------------------------------------------------------------------------------------------
# ConvLSTM gate math
# --------------------------------------------------------------------------------------------
class LSTM0Gates(Function):
    """model/uniposeLSTM.py:17-22 given the stacked gate pre-activations g|i|o."""

    @staticmethod
    def forward(ctx, gates, cg: int):
        _dev_ok(gates)
        n, h, w, ldg = gates.shape
        ldo = rup4(cg)
        alloc = torch.zeros if ldo != cg else torch.empty
        cell = alloc((n, h, w, ldo), dtype=torch.float32, device=gates.device)
        hide = alloc((n, h, w, ldo), dtype=torch.float32, device=gates.device)
        _C.check(_C.lib().up_lstm0_fwd(gates.data_ptr(), ldg, cell.data_ptr(), hide.data_ptr(), ldo, n * h * w, cg,
                                       _stream(gates)), "lstm0_fwd")
        ctx.cg = cg
        ctx.save_for_backward(gates)
        return cell, hide

    @staticmethod
    @once_differentiable
    def backward(ctx, dcell, dhide):
        (gates,) = ctx.saved_tensors
        n, h, w, ldg = gates.shape
        dcell, dhide = _dense(dcell), _dense(dhide)
        dg = torch.zeros_like(gates)
        _C.check(_C.lib().up_lstm0_bwd(gates.data_ptr(), ldg, dcell.data_ptr(), dhide.data_ptr(), dcell.shape[3],
                                       dg.data_ptr(), n * h * w, ctx.cg, _stream(gates)), "lstm0_bwd")
        return dg, None


class LSTMGates(Function):
    """model/uniposeLSTM.py:41-62 given the stacked gate pre-activations g|i|o|f and the previous cell."""

    @staticmethod
    def forward(ctx, gates, cprev, cg: int):
        _dev_ok(gates, cprev)
        n, h, w, ldg = gates.shape
        ldo = rup4(cg)
        alloc = torch.zeros if ldo != cg else torch.empty
        cell = alloc((n, h, w, ldo), dtype=torch.float32, device=gates.device)
        hide = alloc((n, h, w, ldo), dtype=torch.float32, device=gates.device)
        _C.check(_C.lib().up_lstm_fwd(gates.data_ptr(), ldg, cprev.data_ptr(), _nhwc_ok(cprev), cell.data_ptr(),
                                      hide.data_ptr(), ldo, n * h * w, cg, _stream(gates)), "lstm_fwd")
        ctx.cg = cg
        ctx.save_for_backward(gates, cprev, cell)
        return cell, hide

    @staticmethod
    @once_differentiable
    def backward(ctx, dcell, dhide):
        gates, cprev, cell = ctx.saved_tensors
        n, h, w, ldg = gates.shape
        dcell, dhide = _dense(dcell), _dense(dhide)
        dg = torch.zeros_like(gates)
        dcp = torch.zeros_like(cell)
        _C.check(_C.lib().up_lstm_bwd(gates.data_ptr(), ldg, cprev.data_ptr(), _nhwc_ok(cprev), cell.data_ptr(),
                                      dcell.data_ptr(), dhide.data_ptr(), cell.shape[3], dg.data_ptr(),
                                      dcp.data_ptr(), n * h * w, ctx.cg, _stream(gates)), "lstm_bwd")
        return dg, dcp, None


# --------------------------------------------------------------------------------------------
# heat-map argmax
# --------------------------------------------------------------------------------------------
def heatmap_argmax(hm: torch.Tensor):
    """get_max_preds (utils/evaluate.py:32-54) on the device: returns (preds (B,J,2) float32,
    maxvals (B,J,1) float32, idx (B,J) int32)."""
    _dev_ok(hm)
    hm = _dense(hm.detach())
    b, j, h, w = hm.shape
    idx = torch.empty((b, j), dtype=torch.int32, device=hm.device)
    preds = torch.empty((b, j, 2), dtype=torch.float32, device=hm.device)
    mx = torch.empty((b, j, 1), dtype=torch.float32, device=hm.device)
    _C.check(_C.lib().up_heatmap_argmax(hm.data_ptr(), b, j, h, w, idx.data_ptr(), preds.data_ptr(), mx.data_ptr(),
                                        _stream(hm)), "heatmap_argmax")
    return preds, mx, idx


def _as_f64(a, dev):
    t = torch.as_tensor(a, dtype=torch.float64).contiguous()
    t = t.to(dev) if dev is not None else t
    if not t.is_cuda and not _C._ALLOW_HOST_POINTERS:
        raise _C.UniPoseHipError("unipose_amd kernels need CUDA(HIP) tensors; there is no CPU fallback")
    return t


def make_heatmaps(kpt_xy, height: int, width: int, stride: float, sigma: float, device):
    """Target heat-maps of a batch on the device (lsp_lspet_data.py:224-236, mpii_data.py:165-175): kpt_xy (B,K,2) pixel
    coordinates (any array-like; kept in float64 like the loaders' annotations), maps of int(height/stride) x
    int(width/stride); returns (B, K+1, h, w) float32 with the background in channel 0."""
    k = _as_f64(kpt_xy, device)
    b, nk, _ = k.shape
    h, w = int(height / stride), int(width / stride)
    out = torch.empty((b, nk + 1, h, w), dtype=torch.float32, device=k.device)
    _C.check(_C.lib().up_make_heatmaps(k.data_ptr(), b, nk, h, w, float(stride), float(sigma), out.data_ptr(),
                                       _stream(out)), "make_heatmaps")
    return out


def make_centermaps(center_xy, height: int, width: int, sigma: float = 3.0, device=None):
    """Gaussian centre maps (lsp_lspet_data.py:238-242): center_xy (N,2) -> (N, 1, height, width) float32."""
    c = _as_f64(center_xy, device)
    out = torch.empty((c.shape[0], 1, height, width), dtype=torch.float32, device=c.device)
    _C.check(_C.lib().up_make_gaussian_maps(c.data_ptr(), c.shape[0], height, width, float(sigma), out.data_ptr(),
                                            _stream(out)), "make_gaussian_maps")
    return out


def normalize_image(img_hwc: torch.Tensor, mean: float = 128.0, std: float = 256.0):
    """(B,H,W,C) float32 pixels -> (B,C,H,W) (pixel - mean) / std: Mytransforms.to_tensor + normalize as the loaders
    call them (lsp_lspet_data.py:244-245)."""
    _dev_ok(img_hwc)
    x = _dense(img_hwc)
    b, h, w, c = x.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    _C.check(_C.lib().up_normalize_image(x.data_ptr(), b, h, w, c, float(mean), float(std), out.data_ptr(), _stream(x)),
             "normalize_image")
    return out


DATASET_IDS = {"LSP": 0, "COCO": 1, "Penn_Action": 2, "NTID": 3, "PoseTrack": 4, "BBC": 5, "MPII": 6}


def accuracy(output: torch.Tensor, target: torch.Tensor, thr_PCK: float, thr_PCKh: float, dataset: str,
             hm_type: str = "gaussian", threshold: float = 0.5):
    """utils/evaluate.py:58-172 ``accuracy`` with the heat-maps left on the device: both argmaxes and the PCK / PCKh
    arithmetic run as kernels, only the (J,) results and the (B,J,2) predictions come back.  Returns the reference's
    tuple (acc, PCK, PCKh, cnt, pred, visible) as numpy values.  `threshold` is accepted and unused, as in the
    reference (its dist_acc calls always use 0.5, evaluate.py:78)."""
    if hm_type != "gaussian":
        raise NotImplementedError("only hm_type='gaussian' (the reference defines nothing else, evaluate.py:62)")
    if dataset not in DATASET_IDS:
        raise ValueError(f"unknown dataset {dataset!r}")
    pred, _, _ = heatmap_argmax(output)
    tgt, _, _ = heatmap_argmax(target)
    b, j, h, w = output.shape
    res = torch.empty((4, j), dtype=torch.float64, device=output.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=output.device)
    _C.check(_C.lib().up_pck_accuracy(pred.data_ptr(), tgt.data_ptr(), b, j, h, w, DATASET_IDS[dataset], float(thr_PCK),
                                      float(thr_PCKh), res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(),
                                      res[3].data_ptr(), cnt.data_ptr(), _stream(output)), "pck_accuracy")
    r = res.cpu().numpy()
    return r[0], r[1], r[2], int(cnt.item()), pred.cpu().numpy(), r[3]


BOX_CHANNEL0 = {"LSP": 15, "MPII": 17, "PoseTrack": 18, "NTID": 20, "NTID_small": 20}   # utils/uniPose.py:20-49


def uniPose_kpts(maps: torch.Tensor, dataset: str, img_h: float = 368.0, img_w: float = 368.0):
    """utils/uniPose.py:14-200 (multi-person decode of the optional box head) with the maps left on the device: one
    kernel marks the peaks of the centre and the four corner maps, a second one arg-maxes the 14 joint channels inside
    every person's box; only the peak coordinates and the (P,14,2) results cross to the host.  Returns the reference's
    list [[idx, x, y], ...]; `img_h`/`img_w` are accepted and unused like there.  Error behaviour follows numpy's in
    the reference: IndexError when a corner map has fewer peaks than the centre map, ValueError for an empty box."""
    if dataset not in BOX_CHANNEL0:
        raise ValueError(f"no box channels defined for dataset {dataset!r}")
    _dev_ok(maps)
    m = _dense(maps.detach()[0].float())
    c, h, w = m.shape
    f = BOX_CHANNEL0[dataset]
    if c < f + 5:
        raise IndexError(f"index {f + 4} is out of bounds for axis 0 with size {c}")
    mask = torch.empty((5, h, w), dtype=torch.uint8, device=m.device)
    _C.check(_C.lib().up_peak_mask(m[f:f + 5].data_ptr(), 5, h, w, mask.data_ptr(), _stream(m)), "peak_mask")
    nz = torch.nonzero(mask).cpu().tolist()                       # (channel, row, col), lexicographic = row-major per map
    center, tl, bl, tr, br = ([[i, j] for ch, i, j in nz if ch == k] for k in range(5))
    if not center:
        return []
    boxes = []
    for idx in range(len(center)):
        r0, c0 = tl[idx]                                          # IndexError like the reference's list indexing
        r1, c1 = br[idx]
        if r1 <= r0 or c1 <= c0:
            raise ValueError("attempt to get argmax of an empty sequence")
        boxes.append([r0, r1, c0, c1])
        _ = bl[idx], tr[idx]
    bx = torch.tensor(boxes, dtype=torch.int32).to(m.device)
    out = torch.empty((len(boxes), 14, 2), dtype=torch.int32, device=m.device)
    _C.check(_C.lib().up_box_argmax(m.data_ptr(), c, h, w, bx.data_ptr(), len(boxes), 1, 14, out.data_ptr(), _stream(m)),
             "box_argmax")
    rel = out.cpu().tolist()
    kpts = []
    for idx, (r0, _, c0, _) in enumerate(boxes):
        for hh, ww in rel[idx]:
            kpts.append([idx, int(ww + c0), int(hh + r0)])
        for lst in (center, tl, bl, tr, br):
            kpts.append([idx, lst[idx][1], lst[idx][0]])
    return kpts


def get_kpts(maps: torch.Tensor, img_h: float = 368.0, img_w: float = 368.0):
    """utils/utils.py:94-106 on top of the device argmax: [[x, y], ...] for joints 1.. of sample 0."""
    _, _, idx = heatmap_argmax(maps[:1])
    h, w = maps.shape[2], maps.shape[3]
    flat = idx[0].tolist()[1:]
    return [[int((i % w) * img_w / w), int((i // w) * img_h / h)] for i in flat]
