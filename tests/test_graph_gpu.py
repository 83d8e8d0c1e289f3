"""Inference forward as one hipGraph (unipose_amd/graph.py) on the MI355X."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_forward_matches_eager_and_is_faster():
    from model.unipose import unipose
    from unipose_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=16).to(dev).eval()
    xs = [torch.randn(1, 3, 368, 368, device=dev) for _ in range(3)]
    with torch.no_grad():
        eager = [model(x).clone() for x in xs]
    fwd = GraphedForward(model, xs[0])
    for x, ref in zip(xs, eager):                     # replays with different inputs reproduce the eager results bit for bit
        out = fwd(x)
        assert torch.equal(out, ref)
    out = fwd(xs[0])
    assert torch.equal(out, eager[0])

    def wall(fn, n=30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager_fn():
        with torch.no_grad():
            model(xs[0])

    t_eager, t_graph = wall(eager_fn), wall(lambda: fwd(xs[0]))
    print(f"B=1 368x368 inference forward: eager {t_eager:.3f} ms, one hipGraph {t_graph:.3f} ms")
    assert t_graph < t_eager


def test_graphed_forward_rejects_other_shapes_and_training_mode():
    from model.unipose import unipose
    from unipose_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    model = unipose("MPII", num_classes=16).to(dev)
    x = torch.randn(2, 3, 128, 128, device=dev)
    with pytest.raises(ValueError):
        GraphedForward(model.train(), x)
    fwd = GraphedForward(model.eval(), x)
    with pytest.raises(ValueError):
        fwd(torch.randn(1, 3, 128, 128, device=dev))
    with pytest.raises(TypeError):
        GraphedForward(model, torch.randn(2, 3, 128, 128))


def test_graphed_forward_recapture_follows_new_weights():
    from model.unipose import unipose
    from unipose_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = unipose("MPII", num_classes=16).to(dev).eval()
    x = torch.randn(2, 3, 128, 128, device=dev)
    fwd = GraphedForward(model, x)
    before = fwd(x).clone()
    with torch.no_grad():
        model.decoder.last_conv[-1].weight.mul_(2.0)
        ref = model(x).clone()
    fwd.recapture()
    assert torch.equal(fwd(x), ref) and not torch.equal(ref, before)


def test_graphed_forward_close_returns_the_stream_scratch():
    """Every GraphedForward owns a capture stream, and the library keeps 16 MB of K-split scratch per stream that ran a split
    launch (B = 1: every small launch is split): close() / deletion hands it back, so building graphs in a loop does not grow."""
    from model.unipose import unipose
    from unipose_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=16).to(dev).eval()
    x = torch.randn(1, 3, 128, 128, device=dev)
    with torch.no_grad():
        ref = model(x).clone()

    def used():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info(dev)
        return total - free

    fwd = GraphedForward(model, x)
    assert torch.equal(fwd(x), ref)
    fwd.close()
    fwd.close()                                   # idempotent
    torch.cuda.empty_cache()
    base = used()
    for _ in range(6):
        fwd = GraphedForward(model, x)
        assert torch.equal(fwd(x), ref)
        del fwd
    torch.cuda.empty_cache()
    grown = used() - base
    assert grown < 48 * 2 ** 20, f"{grown / 2 ** 20:.0f} MiB left behind by six graphs"      # 6 x 16 MB would be 96
    with torch.no_grad():
        assert torch.equal(model(x), ref)         # the eager path (its own stream scratch) is untouched


def test_two_live_graphs_on_one_stream_handle_share_the_scratch():
    """torch's stream pool is 32 entries round-robin: after 32 more `torch.cuda.Stream()`s a new GraphedForward gets the handle of a
    live one.  Closing either must not free the K-split scratch the other's captured graph has baked in."""
    from model.unipose import unipose
    from unipose_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=16).to(dev).eval()
    x = torch.randn(1, 3, 128, 128, device=dev)
    with torch.no_grad():
        ref = model(x).clone()
    first = GraphedForward(model, x)
    handle = first.stream.cuda_stream
    twin = None
    keep = [torch.cuda.Stream(device=dev) for _ in range(31)]      # 31 plain streams: the next one closes the 32-entry round
    for _ in range(40):                              # (walk on with whole graphs should the pool be laid out differently)
        g = GraphedForward(model, x, warmup=1)
        if g.stream.cuda_stream == handle:
            twin = g
            break
        g.close()
    del keep
    assert twin is not None, "the stream pool never repeated a handle"
    assert torch.equal(twin(x), ref)
    twin.close()                                     # must NOT release the scratch `first` replays with
    for _ in range(3):
        assert torch.equal(first(x), ref)
    first.close()
