"""Generate tests/golden/*.npz by running the GENUINE reference model.

Runs ONLY in the development container (needs /root/reference); the fixtures it writes are
data (inputs are re-derived from seeds, outputs are stored) and travel with the repo, the
reference never does.  Usage:  python tools/make_goldens.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
# /root/repo/model is a regular package (import-path shim) and would shadow the reference's
# namespace package `model`: import the reference FIRST with the repo root off sys.path.
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
sys.path.insert(0, REF)

import model.modules.backbone.resnet as R  # noqa: E402  (reference)

R.model_zoo.load_url = lambda *a, **k: {}   # ctor downloads ImageNet weights (resnet.py:142); no network
torch.Tensor.cuda = lambda self, *a, **k: self  # LSTM forward calls .cuda() (model/uniposeLSTM.py:99-104)

from model.unipose import unipose as RefUniPose  # noqa: E402
from model.uniposeLSTM import unipose as RefUniPoseLSTM  # noqa: E402

assert RefUniPose.__module__ == "model.unipose" and "/root/reference" in sys.modules["model.unipose"].__file__
sys.path.insert(1, ROOT)
from oracle import unipose_oracle as O  # noqa: E402

SUB = 4   # stride used to sub-sample large gradient tensors (tests index the same way)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


def ref_image_model(K, seed, output_stride=16):
    m = RefUniPose("LSP", num_classes=K, output_stride=output_stride)
    sd = O.synth_state_dict(K, seed)
    missing = set(m.state_dict()) ^ set(sd)
    assert not missing, missing
    m.load_state_dict(sd)
    return m


def g1_eval_full():
    """G1: eval forward, K=14, B=2, 368x368."""
    m = ref_image_model(14, 1).eval()
    x = O.synth_input((2, 3, 368, 368), 11)
    with torch.no_grad():
        y = m(x)
    flat = y.reshape(2, 15, -1)
    save("g1_eval_368.npz", out=y.numpy(), argmax=flat.argmax(2).numpy().astype(np.int32),
         meta=np.array([14, 1, 11]))


def g2_taps():
    """G2: intermediate taps at 160x160 (hooks on the reference modules)."""
    m = ref_image_model(16, 2).eval()
    x = O.synth_input((2, 3, 160, 160), 12)
    taps = {}

    def hook(name):
        def f(_m, _i, o):
            taps[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy()
        return f
    m.backbone.maxpool.register_forward_hook(hook("stem"))
    for n in ("layer1", "layer2", "layer3", "layer4"):
        getattr(m.backbone, n).register_forward_hook(hook(n))
    for n in ("aspp1", "aspp2", "aspp3", "aspp4"):
        getattr(m.wasp, n).register_forward_hook(hook("x" + n[-1]))
    m.wasp.register_forward_hook(hook("wasp"))
    m.decoder.last_conv[7].register_forward_hook(hook("dec_pre"))
    with torch.no_grad():
        y = m(x)
        y_full = torch.nn.functional.interpolate(y, size=(160, 160), mode="bilinear", align_corners=True)
    # keep the fixture small: store sub-sampled taps (every 4th channel) + checksums
    small = {k: v[:, ::4] for k, v in taps.items()}
    sums = {k + "_abs_sum": np.abs(v).sum(dtype=np.float64) for k, v in taps.items()}
    save("g2_taps_160.npz", out=y.numpy(), out_stride1=y_full.numpy()[:, ::4, ::2, ::2], **small, **sums,
         meta=np.array([16, 2, 12]))


def g4_train():
    """G4: train-mode fwd+bwd, B=2, dropouts p=0, 128x128: loss, grads, BN running stats."""
    K = 16
    m = ref_image_model(K, 3).train()
    m.wasp.dropout.p = 0.0
    m.decoder.last_conv[3].p = 0.0
    m.decoder.last_conv[7].p = 0.0
    x = O.synth_input((2, 3, 128, 128), 13)
    t = O.synth_input((2, K + 1, 16, 16), 14, "rand")
    y = m(x)
    loss = torch.nn.MSELoss()(y, t)
    loss.backward()
    sd = m.state_dict()
    g = dict(m.named_parameters())
    keys = ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight", "backbone.layer2.0.downsample.0.weight",
            "backbone.layer3.5.bn2.weight", "backbone.layer3.5.bn2.bias", "backbone.layer4.2.conv2.weight",
            "wasp.conv2.weight", "wasp.aspp2.atrous_conv.weight", "wasp.global_avg_pool.1.weight",
            "decoder.conv1.weight", "decoder.last_conv.0.weight", "decoder.last_conv.8.weight",
            "decoder.last_conv.8.bias"]
    arrs = {"grad/" + k: g[k].grad.numpy() for k in keys}
    for k in list(arrs):                       # keep the fixture small: sub-sample big tensors
        if arrs[k].size > 100_000:
            arrs[k] = arrs[k][::SUB, ::SUB]
    arrs["grad_norms"] = np.array([g[k].grad.double().norm().item() if g[k].grad is not None else -1.0
                                   for k in sorted(g)])
    for k in ("backbone.bn1", "backbone.layer3.5.bn2", "wasp.bn1", "wasp.global_avg_pool.2", "decoder.last_conv.5"):
        arrs["rm/" + k] = sd[k + ".running_mean"].numpy()
        arrs["rv/" + k] = sd[k + ".running_var"].numpy()
    none_grad = [k for k in sorted(g) if g[k].grad is None]
    print("params without grad:", none_grad)
    save("g4_train_128.npz", out=y.detach().numpy(), loss=np.array(loss.item()), **arrs,
         meta=np.array([K, 3, 13, 14]))
    # train-mode B=1 must raise
    try:
        m(x[:1])
        raise SystemExit("expected failure for train-mode batch 1")
    except ValueError as e:
        print("B=1 train raises:", str(e)[:60])


def g5_lstm():
    """G5: UniPose-LSTM, K=13, T=5, B=1, eval, 368x368 frames (state shapes are hard-wired to 46x46)."""
    K = 13
    m = RefUniPoseLSTM(num_classes=K)
    sd = O.synth_state_dict(K, 4, lstm=True)
    assert not (set(m.state_dict()) ^ set(sd)), set(m.state_dict()) ^ set(sd)
    m.load_state_dict(sd)
    m.eval()
    T = 5
    x = O.synth_input((1, T, 3, 368, 368), 15)
    cm = O.synth_input((1, T, 1, 368, 368), 16, "rand")
    heat = torch.zeros(K + 1, 46, 46)
    cell = torch.zeros(15, 46, 46)
    hide = torch.zeros(15, 46, 46)
    res = {}
    with torch.no_grad():
        for j in range(T):                      # uniposeLSTM.py:124-128 call pattern
            heat, cell, hide = m(x, cm, j, heat, hide, cell)
            res[f"heat{j}"], res[f"cell{j}"], res[f"hide{j}"] = heat.numpy(), cell.numpy(), hide.numpy()
    save("g5_lstm_368.npz", **res, meta=np.array([K, 4, 15, 16]))


def g6_argmax():
    """G6: reference get_max_preds (utils/evaluate.py loaded by path; numpy only) on edge cases."""
    spec = importlib.util.spec_from_file_location("ref_evaluate", os.path.join(REF, "utils", "evaluate.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    rng = np.random.default_rng(5)
    hm = rng.standard_normal((3, 15, 46, 46)).astype(np.float32)
    hm[0, 0] = -1.0                              # all negative & all equal -> idx 0, masked
    hm[0, 1] = 0.0                               # all zero -> masked
    hm[0, 2] = -5.0; hm[0, 2, 45, 45] = 3.0      # max at last index
    hm[0, 3] = 1.0                               # all ties -> first
    hm[0, 4] = 0.0; hm[0, 4, 10, 7] = 2.0; hm[0, 4, 30, 3] = 2.0   # duplicate max -> first
    hm[1, 5, 0, 0] = 100.0
    hm[2, 6] = -np.abs(hm[2, 6])                 # negative everywhere, unique max
    preds, maxvals = ev.get_max_preds(hm)
    save("g6_argmax.npz", hm=hm, preds=preds, maxvals=maxvals)


def g7_accuracy():
    """G7: the reference's PCK / PCKh `accuracy` (utils/evaluate.py:58-172, numpy only: loaded by file path) on random
    heat-map stacks for the three datasets the drivers use, with some target joints at coordinates <= 1 (not counted)."""
    spec = importlib.util.spec_from_file_location("ref_evaluate", os.path.join(REF, "utils", "evaluate.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    rng = np.random.default_rng(7)
    out = {}
    for ds, j in (("LSP", 15), ("MPII", 17), ("Penn_Action", 14)):
        b, hw = 6, 32
        tgt = np.zeros((b, j, hw, hw), np.float32)
        outp = rng.standard_normal((b, j, hw, hw)).astype(np.float32) * 0.1
        for n in range(b):
            for c in range(j):
                ty, tx = rng.integers(0, hw, 2)
                if rng.random() < 0.15:
                    tx = int(rng.integers(0, 2))                  # x <= 1: joint not counted
                tgt[n, c, ty, tx] = 1.0
                dy, dx = rng.integers(-9, 10, 2)                  # prediction near (or far from) the target
                py, px = int(np.clip(ty + dy, 0, hw - 1)), int(np.clip(tx + dx, 0, hw - 1))
                outp[n, c, py, px] = 2.0 + rng.random()
        tgt[1, 3] = 0.0                                           # an all-zero target map: argmax 0, masked to (0, 0)
        out.update({f"{ds}_out": outp, f"{ds}_tgt": tgt})
        for tag, (tk, th) in (("std", (0.2, 0.5)), ("tight", (0.03, 0.12))):   # the drivers' thresholds, and discriminating ones
            acc, pck, pckh, cnt, pred, vis = ev.accuracy(outp, tgt, tk, th, ds)
            out.update({f"{ds}_{tag}_thr": np.array([tk, th]), f"{ds}_{tag}_acc": acc, f"{ds}_{tag}_pck": pck,
                        f"{ds}_{tag}_pckh": pckh, f"{ds}_{tag}_cnt": np.array(cnt), f"{ds}_{tag}_pred": pred,
                        f"{ds}_{tag}_vis": vis})
    save("g7_accuracy.npz", **out)


def g8_targets():
    """G8: target heat-maps and centre maps.  The Gaussian is the reference's OWN function: `guassian_kernel` is extracted
    from utils/utils.py with `ast` and executed here (the module itself cannot be imported: cv2 / torchvision are absent);
    the assembly around it (int(coordinate)/stride centres, clip to 1, < 0.0099 -> 0, background channel) follows
    lsp_lspet_data.py:224-242 statement by statement."""
    import ast
    src = open(os.path.join(REF, "utils", "utils.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "guassian_kernel"][0]
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "reference:utils/utils.py", "exec"), ns)
    gk = ns["guassian_kernel"]
    rng = np.random.default_rng(8)
    height = width = 368
    out = {}
    for tag, stride, sigma, nk in (("lsp", 8, 3, 14), ("penn_sigma1", 8, 1, 13), ("odd_stride4", 4, 3.0, 5)):
        b = 3
        kpt = rng.uniform(-20, 400, size=(b, nk, 2))               # some joints outside the image, fractional coordinates
        kpt[0, 0] = [183.99999999, 92.0]                           # int() truncation right below an integer
        kpt[1, 1] = [0.0, 367.9]
        hs, ws = int(height / stride), int(width / stride)
        hm = np.zeros((b, nk + 1, hs, ws), dtype=np.float32)
        for n in range(b):
            for i in range(nk):
                x = int(kpt[n, i][0]) * 1.0 / stride
                y = int(kpt[n, i][1]) * 1.0 / stride
                m = gk(size_h=hs, size_w=ws, center_x=x, center_y=y, sigma=sigma)
                m[m > 1] = 1
                m[m < 0.0099] = 0
                hm[n, i + 1] = m
            hm[n, 0] = 1.0 - np.max(hm[n, 1:], axis=0)
        out.update({f"{tag}_kpt": kpt, f"{tag}_hm": hm, f"{tag}_cfg": np.array([stride, sigma], dtype=np.float64)})
    centers = np.array([[184.3, 190.7], [10.25, 355.5], [400.0, -3.0]])
    cm = np.zeros((3, 1, 96, 80), dtype=np.float32)                # a small non-square map keeps the fixture small
    for n in range(3):
        m = gk(size_h=96, size_w=80, center_x=centers[n, 0] / 4, center_y=centers[n, 1] / 4, sigma=3)
        m[m > 1] = 1
        m[m < 0.0099] = 0
        cm[n, 0] = m
    out.update(centers=centers / 4, centermaps=cm)
    save("g8_targets.npz", **out)


def g9_multi_person():
    """G9: multi-person decode.  `uniPose_kpts` is extracted from utils/uniPose.py with `ast` (the module imports skimage
    and cv2, which are absent) and executed with the scipy it needs; inputs are synthetic (1, C, 46, 46) maps: Gaussian
    blobs for people (centre + four corners + joints inside the box), plateaus, negatives, and cases where the
    reference raises (a corner peak missing -> IndexError, an empty box -> ValueError)."""
    import ast
    import warnings
    src = open(os.path.join(REF, "utils", "uniPose.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "uniPose_kpts"][0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from scipy.ndimage import binary_erosion, generate_binary_structure, maximum_filter
    ns = {"np": np, "torch": torch, "maximum_filter": maximum_filter, "binary_erosion": binary_erosion,
          "generate_binary_structure": generate_binary_structure}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "reference:utils/uniPose.py", "exec"), ns)
    ref = ns["uniPose_kpts"]
    rng = np.random.default_rng(9)
    first = {"LSP": 15, "MPII": 17, "PoseTrack": 18, "NTID": 20}

    def blob(m, y, x, amp=1.0, sigma=1.5):
        ys, xs = np.mgrid[0:m.shape[0], 0:m.shape[1]]
        m += amp * np.exp(-((ys - y) ** 2 + (xs - x) ** 2) / 2.0 / sigma / sigma).astype(np.float32)

    def scene(dataset, people, noise=0.0, H=46, W=46):
        c = first[dataset] + 5
        maps = (rng.standard_normal((1, c, H, W)).astype(np.float32) * noise - 6.0 * noise).astype(np.float32)   # noise stays negative: clamped away
        f = first[dataset]
        for (y0, x0, y1, x1) in people:
            blob(maps[0, f], (y0 + y1) // 2, (x0 + x1) // 2)       # integer centre: a half-pixel one gives two equal peaks
            blob(maps[0, f + 1], y0, x0)
            blob(maps[0, f + 2], y1, x0)
            blob(maps[0, f + 3], y0, x1)
            blob(maps[0, f + 4], y1, x1)
            for j in range(1, f):
                blob(maps[0, j], rng.uniform(y0, y1), rng.uniform(x0, x1), amp=rng.uniform(0.5, 1.0))
        return maps

    cases = {}
    cases["lsp_one"] = ("LSP", scene("LSP", [(5, 6, 30, 28)]))
    cases["lsp_two"] = ("LSP", scene("LSP", [(3, 4, 20, 18), (24, 22, 43, 41)]))
    cases["mpii_two_noise"] = ("MPII", scene("MPII", [(2, 2, 18, 20), (22, 25, 44, 44)], noise=0.01))
    cases["mpii_noise_peaks_raises"] = ("MPII", scene("MPII", [(2, 2, 18, 20)]) + rng.standard_normal((1, 22, 46, 46)).astype(np.float32) * 0.01)
    cases["posetrack_three"] = ("PoseTrack", scene("PoseTrack", [(1, 1, 12, 12), (15, 16, 28, 30), (31, 30, 44, 45)]))
    cases["ntid_one_rect"] = ("NTID", scene("NTID", [(4, 3, 25, 40)], H=32, W=48))
    m = scene("LSP", [(6, 6, 30, 30)])
    m[0, 15, 17:19, 17:19] = m[0, 15].max() + 1.0        # a 2x2 plateau: four centre peaks, one set of corners
    cases["lsp_plateau_raises"] = ("LSP", m)
    m = scene("LSP", [(5, 6, 30, 28)])
    m[0, 19] = -1.0                                      # no bottom-right peak at all
    cases["lsp_missing_corner_raises"] = ("LSP", m)
    m = scene("LSP", [(5, 6, 30, 28)])
    m[0, 16], m[0, 19] = m[0, 19].copy(), m[0, 16].copy()   # top-left below bottom-right: empty box
    cases["lsp_empty_box_raises"] = ("LSP", m)
    m = np.full((1, 20, 46, 46), -0.5, dtype=np.float32)    # nothing anywhere: no people, empty list
    cases["lsp_nothing"] = ("LSP", m)
    out = {}
    for name, (ds, maps) in cases.items():
        out[name + "_maps"] = maps
        out[name + "_dataset"] = np.array(ds)
        try:
            k = ref(torch.from_numpy(maps.copy()), ds)
            out[name + "_kpts"] = np.asarray(k, dtype=np.int64).reshape(-1, 3)
            out[name + "_error"] = np.array("")
        except (IndexError, ValueError) as e:
            out[name + "_kpts"] = np.zeros((0, 3), dtype=np.int64)
            out[name + "_error"] = np.array(type(e).__name__)
        print("g9", name, ds, out[name + "_kpts"].shape, str(out[name + "_error"]))
    save("g9_multi_person.npz", **out)


def g10_eval_736():
    """G10: eval forward at BASELINE configs[4]'s resolution, K=16, B=1, 736x736 -> (1,17,92,92) (model/unipose.py:27-38)."""
    m = ref_image_model(16, 5).eval()
    x = O.synth_input((1, 3, 736, 736), 41)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (1, 17, 92, 92)
    save("g10_eval_736.npz", out=y.numpy(), argmax=y.reshape(1, 17, -1).argmax(2).numpy().astype(np.int32),
         meta=np.array([16, 5, 41]))


class _ExactStatBN(torch.autograd.Function):
    """Train-mode batch normalisation with float64 statistics and a float64 backward, rounded to float32 once: the same
    function as F.batch_norm to within one rounding per element, but not bitwise ATen — a stand-in for "any other
    correct fp32 implementation" when G11 measures how far such an implementation lands from the reference."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps):
        yd = y.double()
        mean, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=False)
        m32, i32 = mean.float(), (1.0 / torch.sqrt(var + eps)).float()
        z = (y - m32.view(1, -1, 1, 1)) * i32.view(1, -1, 1, 1) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        ctx.save_for_backward(y, gamma, m32, i32)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, m32, i32 = ctx.saved_tensors
        n = y.numel() / y.shape[1]
        xh = (y.double() - m32.double().view(1, -1, 1, 1)) * i32.double().view(1, -1, 1, 1)
        dzd = dz.double()
        s1, s2 = dzd.sum((0, 2, 3)), (dzd * xh).sum((0, 2, 3))
        dy = (dzd - (s1 / n).view(1, -1, 1, 1) - xh * (s2 / n).view(1, -1, 1, 1)) * (gamma.double() * i32.double()).view(1, -1, 1, 1)
        return dy.float(), s2.float(), s1.float(), None


def g11_train_b8(name="g11_train_b8_128.npz", B=8, size=128, wseed=7, xseed=43, tseed=44):
    """G11: a better-conditioned train step than G4 (B=8 at 128x128: 512 samples per channel in the 8x8 stages instead of
    128): loss, output, gradients (same key list and sub-sampling as G4), running statistics — and, per gradient, two
    yardsticks for what ANOTHER correct fp32 implementation can be held to on this input (ReLU decisions at round-off flip
    between any two evaluations and move every gradient below them):
      `noise/...` = relative L2 distance between the reference evaluated in fp32 and in fp64;
      `alt/...`   = relative L2 distance between the reference and the SAME reference modules with every train-mode
                    nn.BatchNorm2d evaluated by `_ExactStatBN` (float64 statistics, one rounding per element)."""
    K = 16
    x = O.synth_input((B, 3, size, size), xseed)
    t = O.synth_input((B, K + 1, size // 8, size // 8), tseed, "rand")
    res = {}
    for dt in (torch.float32, torch.float64, "alt"):
        alt = dt == "alt"
        dt = torch.float32 if alt else dt
        m = ref_image_model(K, wseed).to(dt).train()
        m.wasp.dropout.p = 0.0
        m.decoder.last_conv[3].p = 0.0
        m.decoder.last_conv[7].p = 0.0
        if alt:
            for bn in [q for q in m.modules() if isinstance(q, torch.nn.BatchNorm2d)]:
                bn.forward = (lambda inp, bn=bn: _ExactStatBN.apply(inp, bn.weight, bn.bias, bn.eps))
        y = m(x.to(dt))
        loss = torch.nn.MSELoss()(y, t.to(dt))
        loss.backward()
        res["alt" if alt else dt] = (m, y.detach(), loss.detach())
    m, y, loss = res[torch.float32]
    m64, y64, loss64 = res[torch.float64]
    galt = dict(res["alt"][0].named_parameters())
    g, g64 = dict(m.named_parameters()), dict(m64.named_parameters())
    keys = ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight", "backbone.layer2.0.downsample.0.weight",
            "backbone.layer3.5.bn2.weight", "backbone.layer3.5.bn2.bias", "backbone.layer3.11.conv1.weight",
            "backbone.layer4.2.conv2.weight", "wasp.conv2.weight", "wasp.aspp2.atrous_conv.weight",
            "wasp.global_avg_pool.1.weight", "decoder.conv1.weight", "decoder.last_conv.0.weight",
            "decoder.last_conv.8.weight", "decoder.last_conv.8.bias"]
    arrs = {}
    for k in keys:
        a, a64 = g[k].grad, g64[k].grad
        arrs["noise/" + k] = np.array(float((a.double() - a64).norm() / a64.norm()))
        arrs["alt/" + k] = np.array(float((galt[k].grad.double() - a.double()).norm() / a.double().norm()))
        a = a.numpy()
        arrs["grad/" + k] = a[::SUB, ::SUB] if a.size > 100_000 else a
    names = sorted(g)
    arrs["grad_norms"] = np.array([g[k].grad.double().norm().item() if g[k].grad is not None else -1.0 for k in names])
    arrs["grad_norms64"] = np.array([g64[k].grad.norm().item() if g64[k].grad is not None else -1.0 for k in names])
    sd = m.state_dict()
    for k in ("backbone.bn1", "backbone.layer3.5.bn2", "wasp.bn1", "wasp.global_avg_pool.2", "decoder.last_conv.5"):
        arrs["rm/" + k] = sd[k + ".running_mean"].numpy()
        arrs["rv/" + k] = sd[k + ".running_var"].numpy()
    print("g11 fp32-vs-fp64 output", O.max_rel(y, y64.float()), "loss", float(loss), float(loss64))
    print("g11 gradient noise (rel L2, fp32 vs fp64 reference):", {k: float(arrs["noise/" + k]) for k in keys})
    print("g11 gradient distance of the exact-statistics BatchNorm variant:", {k: float(arrs["alt/" + k]) for k in keys})
    save(name, out=y.numpy(), out_noise=np.array(O.max_rel(y, y64.float())), loss=np.array(loss.item()),
         loss64=np.array(loss64.item()), **arrs, meta=np.array([K, wseed, xseed, tseed, B] + ([size] if size != 128 else [])))


def g14_train_368():
    """G14: the G11 recipe at the HEADLINE resolution (368x368, B=4: 46x46 / 23x23 maps, 2116 samples per channel in the top stages):
    reference gradients with their own fp32-vs-fp64 and exact-statistics-BatchNorm yardsticks, so that the train step is pinned
    against the genuine reference at the size the benchmark runs, not only at 128x128."""
    g11_train_b8("g14_train_b4_368.npz", B=4, size=368, wseed=9, xseed=61, tseed=62)


def g12_eval_os8():
    """G12: eval forward with output_stride = 8 (resnet.py:54-56: layer3 at stride 1 / dilation 2, multi-grid unit at dilation
    4 x [1,2,4]; wasp.py:41-42: dilations [48,36,24,12]), K=14, B=2, 160x160, plus the layer3 / layer4 / wasp taps."""
    m = ref_image_model(14, 5, output_stride=8).eval()
    x = O.synth_input((2, 3, 160, 160), 15)
    taps = {}

    def hook(name):
        def f(_m, _i, o):
            taps[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy()
        return f
    for n in ("layer2", "layer3", "layer4"):
        getattr(m.backbone, n).register_forward_hook(hook(n))
    m.wasp.register_forward_hook(hook("wasp"))
    with torch.no_grad():
        y = m(x)
    assert taps["layer3"].shape[2:] == (20, 20) and taps["layer4"].shape[2:] == (20, 20), taps["layer3"].shape
    flat = y.reshape(2, 15, -1)
    small = {k: v[:, ::8] for k, v in taps.items()}
    sums = {k + "_abs_sum": np.abs(v).sum(dtype=np.float64) for k, v in taps.items()}
    save("g12_eval_os8_160.npz", out=y.numpy(), argmax=flat.argmax(2).numpy().astype(np.int32), **small, **sums,
         meta=np.array([14, 5, 15]))


def g13_bf16_yardstick():
    """G13: what does a CORRECT bf16 implementation get on the G11 train step (K=16, B=8, 128x128, dropout off)?  The genuine
    reference is run three times: fp32; under torch.autocast(bfloat16); and in fp32 arithmetic with every tensor a bf16-STORAGE
    implementation keeps in HBM rounded to bf16 once (convolution weights, the output of every Conv2d / BatchNorm2d / ReLU /
    pooling / up-sampling module, and the gradient flowing into each of them).  Stored per parameter with >= 4096 elements:
    cosine and relative L2 distance of both bf16 gradients to the reference's own fp32 gradient, plus loss and output error.
    The residual branches are damped (every bn3.weight x 0.25, DAMP): with the plain synthetic weights the ReLU decisions of
    101 layers flip under ANY 2^-9 rounding and the yardstick itself lands at cosine 0.66-0.75 (measured), too loose to show
    a wrong gradient path; damped, a correct bf16 implementation sits at 0.91-0.98.
    tests/test_bf16s_gpu.py holds the HIP bf16-storage path to twice the larger of the two distances."""
    K, B = 16, 8
    DAMP = float(os.environ.get("G13_DAMP", "0.25"))
    x = O.synth_input((B, 3, 128, 128), 43)
    t = O.synth_input((B, K + 1, 16, 16), 44, "rand")

    def rb(v):
        return v.to(torch.bfloat16).to(v.dtype)

    class _RoundGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v):
            return rb(v)

        @staticmethod
        def backward(ctx, g):
            return rb(g)

    def build(mode):
        m = ref_image_model(K, 7).train()
        with torch.no_grad():     # damped residual branches (see the docstring)
            for k, v in m.state_dict().items():
                if k.endswith("bn3.weight"):
                    v.mul_(DAMP)
        m.wasp.dropout.p = 0.0
        m.decoder.last_conv[3].p = 0.0
        m.decoder.last_conv[7].p = 0.0
        if mode == "rounded":
            kinds = (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.ReLU, torch.nn.MaxPool2d, torch.nn.AdaptiveAvgPool2d)
            stem = {id(m.backbone.conv1), id(m.backbone.bn1), id(m.backbone.relu)}    # the stem stays fp32 up to the max-pool
            for mod in m.modules():
                if isinstance(mod, kinds) and id(mod) not in stem:
                    mod.register_forward_hook(lambda _m, _i, o: _RoundGrad.apply(o))
                if isinstance(mod, torch.nn.Conv2d) and id(mod) not in stem:
                    mod.register_forward_pre_hook(lambda mm, _i: None)
                    w = mod.weight
                    mod.forward = (lambda inp, mod=mod: torch.nn.functional.conv2d(
                        inp, rb(mod.weight.detach()) + (mod.weight - mod.weight.detach()), mod.bias, mod.stride, mod.padding, mod.dilation))
        return m

    grads, outs, losses = {}, {}, {}
    for mode in ("fp32", "autocast", "rounded"):
        m = build(mode)
        if mode == "autocast":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                y = m(x)
            y = y.float()
        else:
            y = m(x)
        loss = torch.nn.MSELoss()(y, t)
        loss.backward()
        grads[mode] = {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}
        outs[mode], losses[mode] = y.detach(), float(loss)
    names = [k for k, g in grads["fp32"].items() if g.numel() >= 4096]
    arrs = {"names": np.array(names)}
    for mode in ("autocast", "rounded"):
        cos, rel = [], []
        for k in names:
            a, b = grads[mode][k].flatten(), grads["fp32"][k].flatten()
            cos.append(float(a @ b / (a.norm() * b.norm() + 1e-300)))
            rel.append(float((a - b).norm() / (b.norm() + 1e-300)))
        arrs[mode + "_cos"], arrs[mode + "_rel"] = np.array(cos), np.array(rel)
        arrs[mode + "_out"] = np.array(O.max_rel(outs[mode], outs["fp32"]))
        arrs[mode + "_loss"] = np.array(losses[mode])
        print("g13", mode, "loss", losses[mode], "(fp32", losses["fp32"], ") out", float(arrs[mode + "_out"]),
              "cos min / median", min(cos), float(np.median(cos)), "rel max / median", max(rel), float(np.median(rel)))
    save("g13_bf16_yardstick_b8_128.npz", loss=np.array(losses["fp32"]), **arrs, meta=np.array([K, 7, 43, 44, B]))


def g15_lstm_train():
    """G15: UniPose-LSTM TRAINING against the genuine reference: K=13, B=1 (the reference hard-wires its state
    to batch 1), T=5 frames at 368x368, train mode (legal at B=1: the video WASP has no BatchNorm behind its global-average
    pool, waspVideo.py:56-59), the three dropouts at p=0, the loop of uniposeLSTM.py:116-133 — summed MSE over the frames, ONE
    backward.  Stored: per-frame heat-maps, the loss, sampled gradients (ConvLSTM cells, head, trunk), all gradient norms, running
    statistics after the five calls, and per gradient the reference's own fp32-vs-fp64 distance (`noise/`), the yardstick of G11."""
    K, T = 13, 5
    x = O.synth_input((1, T, 3, 368, 368), 71)
    cm = O.synth_input((1, T, 1, 368, 368), 72, "rand")
    t = O.synth_input((1, T, K + 1, 46, 46), 73, "rand")
    res = {}
    sd = O.synth_state_dict(K, 6, lstm=True)     # (made under the float32 default: the generator's stream depends on the dtype)
    for dt in (torch.float32, torch.float64):
        m = RefUniPoseLSTM(num_classes=K)
        assert not (set(m.state_dict()) ^ set(sd))
        m.load_state_dict(sd)
        torch.set_default_dtype(dt)          # the reference creates its state tensors with torch.zeros(...) (uniposeLSTM.py:99-104)
        try:
            m = m.to(dt).train()
            m.wasp.dropout.p = 0.0
            m.decoder.last_conv[3].p = 0.0
            m.decoder.last_conv[7].p = 0.0
            heat = torch.zeros(K + 1, 46, 46)
            cell = torch.zeros(15, 46, 46)
            hide = torch.zeros(15, 46, 46)
            loss, heats = 0.0, []
            for j in range(T):
                heat, cell, hide = m(x.to(dt), cm.to(dt), j, heat, hide, cell)
                loss = loss + torch.nn.MSELoss()(heat, t[:, j].to(dt))
                heats.append(heat.detach())
            loss.backward()
            res[dt] = (m, torch.stack(heats), loss.detach())
        finally:
            torch.set_default_dtype(torch.float32)
    m, heats, loss = res[torch.float32]
    m64, heats64, loss64 = res[torch.float64]
    g, g64 = dict(m.named_parameters()), dict(m64.named_parameters())
    keys = ["lstm_0.conv_g_lstm.weight", "lstm_0.conv_o_lstm.bias", "lstm.conv_gx_lstm.weight", "lstm.conv_fh_lstm.weight",
            "lstm.conv_ih_lstm.bias", "conv1.weight", "conv2.weight", "conv3.weight", "conv4.weight", "conv5.weight", "conv5.bias",
            "backbone.conv1.weight", "backbone.layer3.5.bn2.weight", "backbone.layer3.11.conv1.weight", "wasp.conv2.weight",
            "wasp.global_avg_pool.1.weight", "decoder.last_conv.8.weight"]
    arrs = {}
    for k in keys:
        a, a64 = g[k].grad, g64[k].grad
        arrs["noise/" + k] = np.array(float((a.double() - a64).norm() / a64.norm()))
        a = a.numpy()
        arrs["grad/" + k] = a[::SUB, ::SUB] if a.size > 100_000 else a
    names = sorted(g)
    arrs["grad_norms"] = np.array([g[k].grad.double().norm().item() if g[k].grad is not None else -1.0 for k in names])
    sdm = m.state_dict()
    for k in ("backbone.bn1", "backbone.layer3.5.bn2", "wasp.bn1", "decoder.last_conv.5"):
        arrs["rm/" + k] = sdm[k + ".running_mean"].numpy()
        arrs["rv/" + k] = sdm[k + ".running_var"].numpy()
    arrs["nbt/backbone.bn1"] = np.array(int(sdm["backbone.bn1.num_batches_tracked"]))
    print("g15 fp32-vs-fp64 heat-maps", O.max_rel(heats, heats64.float()), "loss", float(loss), float(loss64))
    print("g15 gradient noise (rel L2, fp32 vs fp64 reference):", {k: round(float(arrs["noise/" + k]), 6) for k in keys})
    save("g15_lstm_train_368.npz", heat=heats.numpy(), heat_noise=np.array(O.max_rel(heats, heats64.float())),
         loss=np.array(loss.item()), loss64=np.array(loss64.item()), **arrs, meta=np.array([K, 6, 71, 72, 73, T]))


G16_WEIGHTS = ("backbone.conv1.weight", "backbone.layer2.0.conv2.weight", "backbone.layer3.11.conv1.weight",
               "wasp.conv2.weight", "decoder.last_conv.0.weight", "decoder.last_conv.8.weight")
G16_BN = ("backbone.bn1", "backbone.layer3.5.bn2", "backbone.layer4.2.bn3", "wasp.bn1", "wasp.global_avg_pool.2",
          "decoder.last_conv.5")


def g16_trajectory(name="g16_adam_3steps_b8_128.npz", K=16, B=8, size=128, steps=3, wseed=11, xseed=71, tseed=72):
    """G16: a full OPTIMISER trajectory of the genuine reference — the loop of unipose.py:100-131 (zero_grad -> forward -> MSE ->
    backward -> Adam(lr 1e-4).step) run `steps` times on one resident batch, dropouts at p = 0: per-step loss, the running
    statistics and num_batches_tracked after the last step, and a strided sample of six weights as their MOVE (w_after - w_initial:
    Adam's first steps are ~lr * sign(g), so the move — not the weight — is what carries information).  The same trajectory in
    float64 gives the yardsticks: `loss_noise` per step, `noise/move/<weight>` = relative L2 distance of the fp32 move from the fp64
    move (elements whose gradient sits at round-off take the other sign in ANY second evaluation), `noise/rm|rv/<bn>`."""
    x = O.synth_input((B, 3, size, size), xseed)
    t = O.synth_input((B, K + 1, size // 8, size // 8), tseed, "rand")
    res = {}
    for dt in (torch.float32, torch.float64):
        m = ref_image_model(K, wseed).to(dt).train()
        m.wasp.dropout.p = 0.0
        m.decoder.last_conv[3].p = 0.0
        m.decoder.last_conv[7].p = 0.0
        w0 = {k: v.detach().clone() for k, v in m.named_parameters()}
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)                 # unipose.py:72
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            loss = torch.nn.MSELoss()(m(x.to(dt)), t.to(dt))
            loss.backward()
            opt.step()
            losses.append(loss.item())
        res[dt] = (m, w0, losses)
    m, w0, losses = res[torch.float32]
    m64, w064, losses64 = res[torch.float64]
    p, p64, sd, sd64 = dict(m.named_parameters()), dict(m64.named_parameters()), m.state_dict(), m64.state_dict()
    arrs = {"loss": np.array(losses), "loss64": np.array(losses64)}
    for k in G16_WEIGHTS:
        mv = (p[k].detach() - w0[k]).flatten()[::7]
        mv64 = (p64[k].detach() - w064[k]).flatten()[::7]
        arrs["move/" + k] = mv.numpy()
        arrs["noise/move/" + k] = np.array(float((mv.double() - mv64).norm() / mv64.norm()))
    for k in G16_BN:
        for tag, key in (("rm", ".running_mean"), ("rv", ".running_var")):
            arrs[f"{tag}/{k}"] = sd[k + key].numpy()
            arrs[f"noise/{tag}/{k}"] = np.array(O.max_rel(sd[k + key], sd64[k + key].float()))
    arrs["num_batches_tracked"] = np.array([int(sd[k + ".num_batches_tracked"]) for k in G16_BN] +
                                           [int(sd["decoder.bn2.num_batches_tracked"])])
    moved = sorted(k for k in p if not torch.equal(p[k].detach(), w0[k]))
    arrs["params_moved"] = np.array(len(moved))
    print("g16 losses", losses, "fp64", losses64)
    print("g16 move noise:", {k: round(float(arrs["noise/move/" + k]), 5) for k in G16_WEIGHTS})
    print("g16 running-stat noise:", {k: float(v) for k, v in arrs.items() if k.startswith("noise/r")})
    print("g16 parameters that moved:", len(moved), "of", len(p))
    save(name, **arrs, meta=np.array([K, wseed, xseed, tseed, B, size, steps]))


def g0_keys():
    """G0: the reference's state_dict contract (names, shapes, dtypes, order) for both models."""
    import json
    out = {}
    for name, m in (("unipose_K14", RefUniPose("LSP", num_classes=14)), ("unipose_lstm_K13", RefUniPoseLSTM(num_classes=13))):
        out[name] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        out[name + "_params"] = [k for k, _ in m.named_parameters()]
    with open(os.path.join(OUT, "g0_state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("wrote g0_state_dict_keys.json", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g0", "g1", "g2", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16"]
    fns = dict(g0=g0_keys, g1=g1_eval_full, g2=g2_taps, g4=g4_train, g5=g5_lstm, g6=g6_argmax, g7=g7_accuracy,
               g8=g8_targets, g9=g9_multi_person, g10=g10_eval_736, g11=g11_train_b8, g12=g12_eval_os8, g13=g13_bf16_yardstick,
               g14=g14_train_368, g15=g15_lstm_train, g16=g16_trajectory)
    for w in which:
        fns[w]()
