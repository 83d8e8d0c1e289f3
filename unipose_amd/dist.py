"""Data-parallel gradient exchange: one process per GPU, batch sharded, weights replicated, ONE
all-reduce (sum, then /N) of the parameter gradients per step over RCCL/xGMI
(``torch.distributed`` backend "nccl" on ROCm), bucketed in reverse-registration (~ reverse autograd)
order and launched from post-accumulate-grad hooks so the exchange overlaps the rest of backward.

The reference has no distributed code at all; BatchNorm statistics stay per-GPU exactly
like its plain nn.BatchNorm2d.  Parameters that never receive a gradient (decoder.conv2 / bn2,
SURVEY D9) are detected on the first step and left out of the buckets.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "buf", "views", "pending", "work")

    def __init__(self, params):
        self.params = params
        n = sum(p.numel() for p in params)
        self.buf = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.buf[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.work = None


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.buckets: Optional[List[_Bucket]] = None
        self._where = {}
        self._handles = []
        # replicate the initial weights / buffers from rank 0 once (documented choice: BN running
        # statistics are NOT re-broadcast per step; each rank keeps its own like the reference would)
        if self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=group)

    # -- first step: plain post-backward exchange, then build the buckets from what received a grad --
    def _build(self):
        live = [p for p in reversed(self.params) if p.grad is not None]
        self.buckets, cur, size = [], [], 0
        for p in live:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def _hook(self, p):
        bi, pi = self._where[p]
        b = self.buckets[bi]
        if p.grad.is_cuda:
            from . import ops
            ops.wgrad_fence(p.grad.device)      # weight gradients are produced on a side stream
        b.views[pi].copy_(p.grad)
        b.pending -= 1
        if b.pending == 0:
            b.work = dist.all_reduce(b.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Call after loss.backward(): waits for the in-flight buckets and writes averaged grads back."""
        if self.world == 1:
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
        inv = 1.0 / self.world
        for b in self.buckets:
            if b.pending != 0:
                raise RuntimeError("a bucketed parameter received no gradient this step")
            b.work.wait()
            b.buf.mul_(inv)
            for p, v in zip(b.params, b.views):
                p.grad.copy_(v)
            b.pending, b.work = len(b.params), None

    def payload_bytes(self) -> int:
        return sum(b.buf.numel() * b.buf.element_size() for b in self.buckets) if self.buckets else 0

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def shard_seed(base: int, rank: int) -> int:
    """Per-rank synthetic-data seed (seed = base + rank)."""
    return base + rank
