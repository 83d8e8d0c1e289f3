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


@pytest.mark.parametrize("math,two_streams", [("f32", False), ("f32", True), ("bf16s", False)])
def test_graphed_train_step_equals_eager_steps(math, two_streams):
    """unipose_amd.graph.GraphedTrainStep (the whole-graph TRAINING entry): the step captured as one hipGraph — both
    streams, the weight re-pack, the BatchNorm fold tickets, capturable fused Adam — against the same steps issued eagerly:
    loss, gradients, weights, running statistics and Adam state after 3 + 2 steps agree bit for bit (B = 4, 128x128, dropouts 0;
    the one float-atomic quantity, the last convolution's bias gradient, to round-off)."""
    import copy
    from model.unipose import unipose
    from unipose_amd import ops
    from unipose_amd.graph import GraphedTrainStep
    dev = torch.device("cuda:0")
    ops.set_conv_math(math)
    try:
        torch.manual_seed(0)
        base = unipose("MPII", num_classes=16).to(dev).train()
        for d in (base.wasp.dropout, base.decoder.last_conv[3], base.decoder.last_conv[7]):
            d.p = 0.0
        x = torch.randn(4, 3, 128, 128, device=dev)
        t = torch.rand(4, 17, 16, 16, device=dev)
        results = []
        for graphed in (False, True):
            m = copy.deepcopy(base)
            opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True, capturable=True)
            if graphed:
                step = GraphedTrainStep(m, opt, x, t, warmup=3, two_streams=two_streams)
                for _ in range(2):
                    loss = step(x, t)
                loss = loss.clone()
            else:
                for _ in range(5):
                    opt.zero_grad(set_to_none=True)
                    loss = ops.mse_loss(m(x), t)
                    loss.backward()
                    opt.step()
            torch.cuda.synchronize()
            st = opt.state_dict()["state"]
            results.append({"loss": loss.detach().reshape(1).cpu(),
                            **{"w." + n: p.detach().cpu().clone() for n, p in m.named_parameters()},
                            **{"g." + n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None},
                            **{"b." + n: b.detach().cpu().clone() for n, b in m.named_buffers()},
                            **{f"adam.{i}.exp_avg": v["exp_avg"].cpu().clone() for i, v in st.items()}})
            if graphed:
                step.close()
        eager, graph = results
        assert eager.keys() == graph.keys()
        loose = ("decoder.last_conv.8.bias",)          # float-atomic column sum (and the Adam state / weight it feeds)
        bad = []
        for k in eager:
            if torch.equal(eager[k], graph[k]):
                continue
            if any(k.endswith(s) for s in loose) or k == f"adam.{len(list(base.parameters())) - 1}.exp_avg":
                assert torch.allclose(eager[k], graph[k], rtol=1e-4, atol=1e-7), k
                continue
            bad.append(k)
        assert not bad, (len(bad), bad[:6])
        print(f"graphed train step ({math}): loss {float(graph['loss']):.7f} == eager, {len(eager)} tensors equal")
    finally:
        ops.set_conv_math("f32")


def test_graphed_train_step_dropout_masks_differ_between_replays():
    """A replay re-issues the dropout launches with unchanged arguments; the device-side step counter still gives every step its
    own masks (two replays on the same batch produce different losses; with p = 0 they would be the trajectory of test 1)."""
    from model.unipose import unipose
    from unipose_amd.graph import GraphedTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = unipose("MPII", num_classes=16).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=0.0, fused=True, capturable=True)        # lr 0: only the masks change
    x = torch.randn(2, 3, 128, 128, device=dev)
    t = torch.rand(2, 17, 16, 16, device=dev)
    step = GraphedTrainStep(m, opt, x, t, warmup=1)
    losses = [float(step().clone()) for _ in range(4)]
    step.close()
    assert len(set(losses)) == 4, losses
