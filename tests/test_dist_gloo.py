"""N>1 path on CPU: world_size 2, gloo backend — the gradient exchange used by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.unused = torch.nn.Linear(4, 4)       # never receives a gradient (like decoder.conv2/bn2)
        self.b = torch.nn.Linear(16, 3)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, q, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unipose_amd.dist import GradAllReducer, shard_seed
    torch.manual_seed(100 + rank)                  # different initial weights: must be overwritten by rank 0's
    net = _Net()
    red = GradAllReducer(net, bucket_bytes=256, overlap=overlap)    # tiny buckets -> several of them
    w0 = [p.detach().numpy().copy() for p in net.parameters()]
    outs = []
    for step in range(3):                          # step 0 = unbucketed path, 1-2 = hooks + buckets
        g = torch.Generator().manual_seed(shard_seed(7, rank) + 10 * step)
        x = torch.randn(5, 8, generator=g)
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        local = [None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()]
        red.finish()
        outs.append((local, [None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()]))
    q.put((rank, w0, outs, red.payload_bytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_grad_allreduce_world2(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w_a, out_a, bytes_a), (_, w_b, out_b, _) = res
    for x, y in zip(w_a, w_b):
        assert np.array_equal(x, y)                # weights replicated from rank 0
    for step in range(3):
        la, ra = out_a[step]
        lb, rb = out_b[step]
        for i in range(len(la)):
            if la[i] is None:
                assert ra[i] is None and rb[i] is None
                continue
            want = (la[i] + lb[i]) / 2
            assert np.allclose(ra[i], want, atol=1e-7) and np.allclose(rb[i], want, atol=1e-7), (step, i)
    n_live = sum(p.numel() for n, p in _Net().named_parameters() if not n.startswith("unused"))
    assert bytes_a == 4 * n_live                   # the unused parameters are not in the exchange


def _trainer_worker(rank, world, port, q):
    """unipose_amd.trainer under WORLD_SIZE=2 (what `torch.distributed.run unipose.py` sets up), kernels on the emulator."""
    import argparse
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), UNIPOSE_NO_TQDM="1", UP_EMU_THREADS="4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "emu"))
    import build_emu
    from unipose_amd import _C
    _C.load(build_emu.build())
    _C._ALLOW_HOST_POINTERS = True
    from unipose_amd.trainer import Trainer, init_distributed
    r, w, dev = init_distributed()                 # gloo on a GPU-less box
    assert (r, w, dev.type) == (rank, world, "cpu") and dist.get_backend() == "gloo"
    torch.manual_seed(50 + rank)                   # different initial weights: rank 0's must win
    args = argparse.Namespace(dataset="LSP", pretrained=None, model_name=None, model_arch="unipose", train_dir=None,
                              val_dir=None, batch_size=2, size=32, train_batches=1, val_batches=1)
    tr = Trainer(args, device=dev)
    assert tr.reducer is not None and tr.reducer.active
    first = next(iter(tr.train_loader))["kpts"].copy()
    w0 = tr.model.wasp.conv2.weight.detach().clone()
    tr.training(0)
    w1 = tr.model.wasp.conv2.weight.detach().clone()
    q.put((rank, first, w0.numpy(), w1.numpy(), tr.iters))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, k_a, w0_a, w1_a, it_a), (_, k_b, w0_b, w1_b, it_b) = res
    assert it_a == it_b == 1
    assert not np.array_equal(k_a, k_b)            # every rank draws its own shard
    assert np.array_equal(w0_a, w0_b)              # weights replicated from rank 0 before the first step
    assert not np.array_equal(w0_a, w1_a)          # the steps moved them
    assert np.array_equal(w1_a, w1_b)              # identical averaged gradients -> identical weights after Adam


def test_bucket_views_are_16_byte_aligned():
    """the weight-gradient reduce pass writes its destination with 16-byte stores, so every parameter's slice of the flat
    exchange buffer starts on a 16-byte boundary whatever odd-sized parameter (the 17-element head bias) sits in front of it; the pad
    elements are zeros and the payload figure counts parameters only."""
    import torch
    from unipose_amd.dist import _Bucket
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in ((17,), (8, 3, 3, 3), (5,), (64, 16, 1, 1), (1,))]
    b = _Bucket(ps)
    base = b.buf.data_ptr()
    assert base % 16 == 0
    for p, v in zip(ps, b.views):
        assert v.shape == p.shape and (v.data_ptr() - base) % 16 == 0
    for p, v in zip(ps, b.views):
        v.copy_(p.detach())
    flat = b.buf.clone()
    covered = torch.zeros_like(flat, dtype=torch.bool)
    for v in b.views:
        off = (v.data_ptr() - base) // 4
        covered[off:off + v.numel()] = True
    assert float(flat[~covered].abs().sum()) == 0.0                      # pads stay zero
    assert int(covered.sum()) == sum(p.numel() for p in ps)
