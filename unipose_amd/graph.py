"""Inference forward — and, since round 6, the whole TRAINING step — as ONE hipGraph.

At small batches the forward of the image model is ~130 launches of a few microseconds each and the step is bound by the
host issuing them (B = 1 at 368^2: 3.1 ms wall for 1.3 ms of kernel time).  `GraphedForward` captures one forward —
every convolution (+ folded BatchNorm + ReLU) launch, the pooling / up-sampling / concatenation kernels, the K-split
partial exchange of small launches — into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replays it per call.

    model = unipose(...).cuda().eval()
    fwd = GraphedForward(model, example_input)          # warms up on a private stream, then captures
    heat = fwd(batch)                                   # copies `batch` into the static input, replays, returns the static output

The returned tensors are the graph's static output buffers: consume (or clone) them before the next call.  Shapes are fixed
at capture; the packed weight images the kernels read are captured by address, so call `recapture()` after changing the
parameters (load_state_dict, an optimizer step).  Inference only (`model.eval()`, `torch.no_grad()`).  `close()` (or deleting the object)
releases the graph and — once no other live graph shares the stream handle — the library scratch of its private stream.

Reference call sites: the validation / test loops `unipose.py:150-160`, `uniposeLSTM.py:160-190` (one forward per batch).
"""
import collections
import threading

import torch

# torch hands out stream handles from a 32-entry round-robin pool per device, so two live GraphedForward objects can hold the
# SAME hipStream_t; the library keys its K-split scratch by that handle and a captured graph has the scratch baked in by
# address.  The scratch is therefore released only when the LAST graph captured on a handle closes.
_stream_users = collections.Counter()
_stream_users_lock = threading.Lock()


def _acquire_stream(handle):
    with _stream_users_lock:
        _stream_users[handle] += 1


def _release_stream(handle):
    """True when no other live graph was captured on this handle (the caller may free the library scratch)."""
    with _stream_users_lock:
        _stream_users[handle] -= 1
        if _stream_users[handle] > 0:
            return False
        del _stream_users[handle]
        return True


class GraphedForward:
    def __init__(self, model, *example_args, warmup=3):
        if model.training:
            raise ValueError("GraphedForward captures the inference forward: call model.eval() first")
        tensors = [a for a in example_args if torch.is_tensor(a)]
        if not tensors or not all(t.is_cuda for t in tensors):
            raise TypeError("GraphedForward needs CUDA example inputs (there is no CPU path)")
        self.model = model
        self.device = tensors[0].device
        self.warmup = warmup
        # library-owned scratch (K-split partials, tap-sort tables) is keyed by stream and allocated on first use: warm up
        # on the stream the capture will run on, so nothing allocates while capturing
        self.stream = torch.cuda.Stream(device=self.device)
        _acquire_stream((self.device.index, self.stream.cuda_stream))
        self.static_args = [a.clone() if torch.is_tensor(a) else a for a in example_args]
        self.graph = None
        self.static_out = None
        self.recapture()

    def recapture(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(self.warmup):
                self.model(*self.static_args)
        self.stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph, stream=self.stream):
            out = self.model(*self.static_args)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.graph, self.static_out = graph, out

    def __call__(self, *args):
        if len(args) != len(self.static_args):
            raise TypeError(f"expected {len(self.static_args)} arguments, got {len(args)}")
        for dst, src in zip(self.static_args, args):
            if torch.is_tensor(dst):
                if not torch.is_tensor(src) or src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError(f"argument of shape {getattr(src, 'shape', None)} / {getattr(src, 'dtype', None)}: the graph was "
                                     f"captured for {tuple(dst.shape)} / {dst.dtype}")
                dst.copy_(src, non_blocking=True)
            elif dst != src:
                raise ValueError(f"non-tensor argument {src!r} differs from the captured {dst!r}")
        self.graph.replay()
        return self.static_out

    def close(self):
        """Drop the graph and hand the private stream's library scratch (K-split partials, 16 MB) back; idempotent."""
        if getattr(self, "stream", None) is None:
            return
        from . import _C
        self.graph, self.static_out = None, None
        torch.cuda.synchronize(self.device)
        if _release_stream((self.device.index, self.stream.cuda_stream)):
            _C.lib().up_stream_release(self.stream.cuda_stream)
        self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the process takes the memory with it
            pass


class GraphedTrainStep:
    """The reference's training step (unipose.py:100-131: zero_grad -> forward -> MSE -> backward -> Adam.step) captured as ONE
    hipGraph and replayed per call: the whole-graph training entry of SURVEY 8(b).  ~800 launches on two streams (the weight
    gradients fork to the side stream and join at the end of the backward pass), the batched weight re-pack, the BatchNorm
    counters' multi-tensor add and the optimizer's fused kernels become one graph launch; the host issues nothing per layer.

        model = unipose("MPII", num_classes=16).cuda().train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=True)      # capturable: the step count lives on the device
        step = GraphedTrainStep(model, opt, images, targets)            # warms up (3 eager steps on the capture stream), captures
        loss = step(images, targets)                                    # copies the batch into the static inputs, replays

    Equal bits to the eager step: the same launches with the same arguments in the same per-stream order (tests/test_graph_gpu.py).
    What is fixed at capture: shapes, the arithmetic mode (ops.set_conv_math), every kernel-selection decision, the addresses of
    parameters, gradients (`param.grad` are the graph's static buffers: read them after a replay, do not replace them) and
    optimizer state.  Dropout masks change from replay to replay through a device-side step counter (up_dropout_fwd_step_t).
    Not captured: a data-parallel gradient exchange (capture per rank with the all-reduce is left to torch's own tooling),
    ops.deferred_wgrad / the video model's unroll.  The warm-up steps are REAL optimizer steps on the example batch.
    """

    def __init__(self, model, optimizer, x, target, loss_fn=None, warmup=3, two_streams=False):
        """two_streams: capture the weight gradients on the side stream like the eager step issues them.  Off by default: on
        ROCm 7.2 every cross-stream edge of a replayed graph is expensive (fp32 B = 32: 77.2 ms two-stream replay against 63.4 ms
        single-stream replay and 61.2 ms for the eager two-stream step, profiles/r06_experiments.txt), so the captured step keeps
        everything on one stream; the results are the same bits either way."""
        from . import ops
        self.two_streams = bool(two_streams)
        if not model.training:
            raise ValueError("GraphedTrainStep captures a training step: call model.train() first")
        if not (torch.is_tensor(x) and x.is_cuda and torch.is_tensor(target) and target.is_cuda):
            raise TypeError("GraphedTrainStep needs CUDA example tensors (there is no CPU path)")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep: construct the optimizer with capturable=True (its step counter must live on the "
                                 "device; torch refuses to capture a non-capturable step)")
        self.model, self.optimizer = model, optimizer
        self.loss_fn = loss_fn or ops.mse_loss
        self.device = x.device
        self.stream = torch.cuda.Stream(device=self.device)
        _acquire_stream((self.device.index, self.stream.cuda_stream))
        self.static_x, self.static_t = x.clone(), target.clone()
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)       # dropout's step counter, bumped inside the graph
        self.graph, self.static_loss = None, None
        self._capture(warmup)

    def _one_step(self):
        from . import ops
        prev, prev_async = ops._DROPOUT_STATE["step_dev"], ops.ASYNC_WGRAD
        ops._DROPOUT_STATE["step_dev"] = self.step_dev
        ops.ASYNC_WGRAD = prev_async and self.two_streams
        try:
            self.step_dev.add_(1)
            loss = self.loss_fn(self.model(self.static_x), self.static_t)
            loss.backward()
            self.optimizer.step()
        finally:
            ops._DROPOUT_STATE["step_dev"] = prev
            ops.ASYNC_WGRAD = prev_async
        return loss

    def _capture(self, warmup):
        from . import ops
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        ops._side_stream(self.device)          # the side stream exists before the capture (stream creation is not capturable)
        with torch.cuda.stream(self.stream):
            # eager steps on the capture stream: library scratch, tap / rectangle tables, workspaces, Adam state — and at least
            # TWO: the batched re-pack builds its job table (a host-to-device copy) at the first stale hit, i.e. in the second step
            for _ in range(max(warmup, 2)):
                self.optimizer.zero_grad(set_to_none=True)
                self._one_step()
        self.stream.synchronize()
        torch.cuda.synchronize(self.device)
        ops.retire_pending_repacks()           # (their events were recorded outside the capture)
        # the gradients of the captured step live in the graph's private pool: drop the eager ones first, so that backward
        # allocates (and the graph owns) fresh static .grad tensors
        self.optimizer.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self.stream):
            loss = self._one_step()
        cur.wait_stream(self.stream)
        self.graph, self.static_loss = graph, loss.detach()

    def __call__(self, x=None, target=None):
        for dst, src in ((self.static_x, x), (self.static_t, target)):
            if src is None:
                continue
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f"batch of shape {tuple(src.shape)} / {src.dtype}: the step was captured for {tuple(dst.shape)} / {dst.dtype}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_loss

    def close(self):
        if getattr(self, "stream", None) is None:
            return
        from . import _C
        self.graph, self.static_loss = None, None
        torch.cuda.synchronize(self.device)
        if _release_stream((self.device.index, self.stream.cuda_stream)):
            _C.lib().up_stream_release(self.stream.cuda_stream)
        self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass
