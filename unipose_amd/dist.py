"""Data-parallel gradient exchange: one process per GPU, batch sharded, weights replicated, the parameter
gradients averaged once per step over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm).

The reference has no distributed code at all; BatchNorm statistics stay per-GPU exactly
like its plain nn.BatchNorm2d.  Parameters that never receive a gradient (decoder.conv2 / bn2,
SURVEY D9) are detected on the first step and left out of the exchange.

Two forms (``overlap``):
  * default — after backward: ONE multi-tensor copy of every gradient into flat buffers and one all-reduce per
    buffer (32 MB granularity); measured faster for this model than the hook form (see ``GradAllReducer``).
  * ``overlap=True`` — buckets in the order autograd hands out the gradients (recorded once, on the second step), each
    launched by ONE post-accumulate-grad hook on the parameter that completes it, on the weight-gradient side stream of
    ``unipose_amd.ops``: the main stream — the critical path of backward — never waits for the exchange until ``finish()``.
    Needs ``zero_grad(set_to_none=True)`` (a missing gradient is how a changed graph is noticed).
After the all-reduce ``param.grad`` simply becomes a view of the flat buffer (no copy back, no scaling pass:
the collective averages).
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch
import torch.distributed as dist

_DEBUG = os.environ.get("UP_DP_DEBUG", "")      # development switch: "hook-only" = hooks fire, nothing is exchanged


class _Bucket:
    __slots__ = ("params", "buf", "views", "work")

    def __init__(self, params):
        self.params = params
        # every parameter starts on a 16-byte boundary of the flat buffer: the weight-gradient reduce pass writes its destination
        # with 16-byte stores (ops.set_grad_destinations), and an odd-sized parameter (the 17-element head bias) must not
        # misalign everything behind it.  The pad elements are zero on every rank and ride through the all-reduce untouched.
        q = max(16 // params[0].element_size(), 1)
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + q - 1) // q * q
        self.buf = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.views = [self.buf[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]
        self.work = None


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20, group=None, force: bool = False,
                 overlap: bool = False):
        """force=True runs the whole exchange (broadcast, buckets, collectives) even in a 1-rank group — used to
        exercise the RCCL path on a single-GPU box.

        overlap=False (default): ONE multi-tensor copy of all gradients into a flat buffer and ONE all-reduce
        after backward.  The 190 MB exchange takes ~1-2 ms on xGMI against an 80+ ms fp32 step, while merely
        REGISTERING 345 post-accumulate-grad hooks was measured to cost 8.6 ms per step on the MI355X box
        (profiles/README.md), so the un-overlapped form is the faster one for this model.
        overlap=True: bucketed exchange launched during backward from one hook per BUCKET (see _build)."""
        self.overlap = overlap
        self.direct_write = os.environ.get("UNIPOSE_DP_DIRECT", "1") != "0"   # development switch (A/B runs)
        self._dest = {}
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.buckets: Optional[List[_Bucket]] = None
        self._handles = []
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"              # gloo has no AVG: sum, then scale
        if self.active and backend == "nccl":
            # the weight-gradient side stream needs a hardware queue of its own next to RCCL's streams (ROCm maps HIP streams onto
            # GPU_MAX_HW_QUEUES queues round-robin, default 4): unipose_amd/__init__.py exports 8 at import time, which only takes
            # effect if that import precedes the first HIP call — say so loudly when the value in force is smaller
            import warnings
            hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
            if hwq < 8:
                warnings.warn(f"GradAllReducer: GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')!r} (< 8): the weight-gradient "
                              "side stream may share a hardware queue with RCCL's streams (+10 ... +14 % per step measured); export "
                              "GPU_MAX_HW_QUEUES=8 before the process touches the GPU", RuntimeWarning, stacklevel=2)
        # replicate the initial weights / buffers from rank 0 once (documented choice: BN running
        # statistics are NOT re-broadcast per step; each rank keeps its own like the reference would)
        if self.active:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=group)

    # -- streams -------------------------------------------------------------------------------------
    @staticmethod
    def _streams(dev):
        """(main, side) HIP streams for a CUDA device, (None, None) on CPU."""
        if dev.type != "cuda":
            return None, None
        from . import ops
        return torch.cuda.current_stream(dev), ops._side_stream(dev)

    def _reduce(self, buf):
        if self._avg:
            return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # -- first step: plain post-backward exchange, then build the buckets from what received a grad --
    def _make_buckets(self, ordered):
        self.buckets, cur, size = [], [], 0
        for p in ordered:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        # convolution weights get their gradient written straight into the bucket (no gather copy)
        self._dest = {id(p): (b, i) for b in self.buckets for i, p in enumerate(b.params) if p.dim() == 4}
        if self.buckets and self.buckets[0].buf.is_cuda and self.direct_write:
            from . import ops
            ops.set_grad_destinations(self._destination)

    def _build(self):
        live = [p for p in reversed(self.params) if p.grad is not None]
        self._make_buckets(live)
        if self.overlap:
            # Registering a post-accumulate hook on each of the 342 live parameters costs 8.6 ms of a 67 ms step on the MI355X box
            # (profiles/README.md).  The order in which autograd hands out the gradients is a property of the (static) graph: the
            # NEXT step records it through temporary per-parameter hooks, then the buckets are rebuilt in that order and only the
            # parameter that completes a bucket keeps a hook (a handful per step).
            self._order = []
            self._handles = [p.register_post_accumulate_grad_hook(self._order.append) for p in live]
            self._calibrating = True

    def _launch(self, b, side_stream: bool):
        """gather the bucket's gradients into its flat buffer — those that were not written there in the first place, see
        ops.set_grad_destinations — and start the collective"""
        pairs = [(v, q.grad) for v, q in zip(b.views, b.params) if q.grad.data_ptr() != v.data_ptr()]
        main, side = self._streams(b.buf.device) if side_stream else (None, None)
        if side is not None:
            side.wait_stream(main)                 # BN / bias gradients come from the main stream
            with torch.cuda.stream(side):
                if pairs:
                    torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
                b.work = self._reduce(b.buf)
        else:
            if pairs:
                torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
            b.work = self._reduce(b.buf)

    def _destination(self, p):
        """fresh view of the bucket memory behind parameter p (ops.set_grad_destinations), None for unknown parameters"""
        hit = self._dest.get(id(p))
        if hit is None:
            return None
        b, i = hit
        if b.work is not None:                     # (the bucket is in flight: cannot happen for a first gradient)
            return None
        v = b.views[i]
        return v.view(v.shape)                     # a new tensor object: autograd may adopt it as .grad without a copy

    def _tail_hook(self, bi):
        def hook(_p):
            b = self.buckets[bi]
            if _DEBUG == "hook-only" or b.work is not None:
                return
            # the tail parameter is the LAST of its bucket in the recorded order; if the graph changed and a gradient is still
            # missing (zero_grad(set_to_none=True) leaves None), finish() exchanges this bucket after backward instead
            if all(q.grad is not None for q in b.params):
                self._launch(b, side_stream=True)
        return hook

    def _finish_calibration(self):
        for h in self._handles:
            h.remove()
        seen, order = set(), []
        for p in self._order:                      # first hand-out of every parameter, in autograd's order
            if id(p) not in seen:
                seen.add(id(p))
                order.append(p)
        order += [p for b in self.buckets for p in b.params if id(p) not in seen]
        # every rank must cut the same buckets: rank 0's order wins (the orders agree unless the graphs differ)
        if dist.is_initialized() and self.world > 1:
            index = {id(p): i for i, p in enumerate(self.params)}
            known = [index.get(id(p), -1) for p in order]           # (-1: a hooked tensor that is not one of self.params)
            dev = self.params[0].device              # (not order[0]: a rank whose hooks never fired has an empty order)
            # the verdict is taken COLLECTIVELY before the order is broadcast: a rank that raised on its own would leave the
            # others waiting in the second broadcast — min / max of the count over the ranks, then everyone agrees
            cnt = len(known) if known and min(known) >= 0 else -1      # (an empty order counts as a disagreement, collectively)
            lo, hi = (torch.tensor([cnt], dtype=torch.int64, device=dev) for _ in range(2))
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if int(lo.item()) != int(hi.item()) or int(lo.item()) < 0:
                raise RuntimeError("data-parallel ranks disagree on the set of parameters that receive gradients "
                                   f"(this rank: {cnt}, over the ranks: {int(lo.item())} .. {int(hi.item())})")
            idx = torch.tensor(known, dtype=torch.int64, device=dev)
            dist.broadcast(idx, 0, group=self.group)
            order = [self.params[i] for i in idx.tolist()]
        self._make_buckets(order)
        self._handles = [b.params[-1].register_post_accumulate_grad_hook(self._tail_hook(bi))
                         for bi, b in enumerate(self.buckets)]
        self._calibrating = False

    def finish(self):
        """Call after loss.backward(): exchanges what is not in flight yet, waits; param.grad becomes the averaged bucket view."""
        if not self.active:
            return
        if self.buckets is None:
            grads = [p.grad for p in self.params if p.grad is not None]
            if grads and grads[0].is_cuda:
                from . import ops
                ops.wgrad_fence(grads[0].device)
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            self._build()
            return
        if _DEBUG == "hook-only":
            return
        pending = [b for b in self.buckets if b.work is None]
        if pending:
            if pending[0].buf.is_cuda:
                from . import ops
                ops.wgrad_fence(pending[0].buf.device)
            for b in pending:
                if any(p.grad is None for p in b.params):
                    raise RuntimeError("a bucketed parameter received no gradient this step")
                self._launch(b, side_stream=False)
        for b in self.buckets:
            b.work.wait()                          # current (main) stream waits for the collective
            if not self._avg:
                b.buf.div_(self.world)
            for p, v in zip(b.params, b.views):
                p.grad = v
            b.work = None
        if getattr(self, "_calibrating", False):
            self._finish_calibration()

    def payload_bytes(self) -> int:
        """bytes of gradients exchanged per step (the 16-byte alignment pads of the flat buffers not counted)"""
        return sum(p.numel() * p.element_size() for b in self.buckets for p in b.params) if self.buckets else 0

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._dest:
            from . import ops
            if ops._GRAD_DEST["fn"] == self._destination:
                ops.set_grad_destinations(None)
            self._dest = {}


def shard_seed(base: int, rank: int) -> int:
    """Per-rank synthetic-data seed (seed = base + rank)."""
    return base + rank
