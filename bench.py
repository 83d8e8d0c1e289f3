#!/usr/bin/env python
"""images/sec, forward+backward(+Adam), UniPose ResNet-101 on synthetic 368x368 batches (BASELINE.json
configs[1]: K=16, batch 32/GPU, fp32) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU (RCCL via torch.distributed "nccl"); the batch is sharded (weak scaling: 32
images per GPU), weights are replicated, gradients are all-reduced once per step (unipose_amd/dist.py).
A step = zero_grad -> forward -> MSE -> backward -> gradient all-reduce -> Adam, exactly the loop of
the reference's Trainer.training (unipose.py:100-131) with synthetic inputs already resident in HBM.
Rank 0 prints ONE JSON line.  `roofline` is measured live: on every 5th step of the timed region each
MFMA convolution launch is bracketed by hipEvents on its stream (up_profile_begin/enable/end in the C ABI).
`cpu_baseline` times the CPU oracle (a torch-CPU restatement of the reference graph, pinned to the
reference by tests/golden) on this box's host cores on a bounded sample; `stock_gpu_baseline` times the same graph on
this GPU through PyTorch-ROCm eager (MIOpen) — the reference's own way of running; baselines only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The weight-gradient side stream must map to its own hardware queue.  ROCm multiplexes HIP streams onto
# GPU_MAX_HW_QUEUES (default 4) queues round-robin; once RCCL has created its streams the side stream can end up
# sharing a queue with the main stream, which serialises the two (measured: 82 -> 93 ms per step).  Must be set
# before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PROFILE_EVERY = 5                   # steps of the timed region between two launch-by-launch bracketed ones
F32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA (32x32x16)
FLOP_PER_IMAGE_FWD_BWD = 187.7e9    # 3 x 31.279 GMAC x 2 at 368x368, K=16


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(num_classes, size, batch, steps, threads):
    """Reference graph (oracle restatement) on the host cores: the train step (fwd+MSE+bwd) in images/sec as `value`, and
    BASELINE.json configs[0] — K=14, batch 1, eval forward only, the reference's own CPU-runnable case (SURVEY 8d config 1:
    7.0 img/s on 8 threads in the survey container) — as `eval_forward`.  The thread count is chosen by a quick sweep of the
    eval forward (oneDNN stops scaling long before 64 threads on this graph; an oversubscribed run is not a baseline)."""
    from oracle import unipose_oracle as O
    cores = os.cpu_count() or 1
    sd14 = O.synth_state_dict(14, 0)
    x1 = O.synth_input((1, 3, size, size), 0)

    def eval_ms(n_threads, reps):
        torch.set_num_threads(n_threads)
        with torch.no_grad():
            O.unipose_forward(sd14, x1)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                O.unipose_forward(sd14, x1)
                ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    cand = sorted({t for t in (4, 8, 16, 32, 64, threads) if 1 <= t <= max(cores, 1)})
    sweep = {t: round(eval_ms(t, 3), 2) for t in cand}
    best = min(sweep, key=sweep.get)
    ev = eval_ms(best, 10)
    log(f"cpu baseline: eval-forward thread sweep {sweep} -> {best} threads, median of 10: {ev:.1f} ms")
    torch.set_num_threads(best)
    sd = O.clone_sd(O.synth_state_dict(num_classes, 0), requires_grad=True)
    x = O.synth_input((batch, 3, size, size), 1)
    t = O.synth_input((batch, num_classes + 1, size // 8, size // 8), 2, "rand")

    def one():
        for v in sd.values():
            v.grad = None
        y = O.unipose_forward(sd, x, train=True)
        torch.nn.functional.mse_loss(y, t).backward()

    one()
    log("cpu baseline warm-up step done")
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port", "host_cores": cores, "thread_sweep_eval_ms": sweep,
            "sample": f"{steps} train steps (fwd+MSE+bwd, no optimizer) of batch {batch} at {size}x{size}, "
                      f"torch {torch.__version__} CPU, after 1 warm-up step, {best} threads (best of the sweep)",
            "eval_forward": {"value": round(1e3 / ev, 3), "unit": "images/sec", "ms": round(ev, 2), "cores": best,
                             "sample": f"BASELINE.json configs[0]: K=14, batch 1, eval forward at {size}x{size}, median of 10 "
                                       f"after 1 warm-up"}}


def stock_gpu_baseline(dev, num_classes, size, batch, steps=5, warmup=3, benchmark=False, lstm_frames=0):
    """The SAME train step on the SAME GPU through the platform's stock path: the oracle's functional restatement of the
    reference graph executed by PyTorch-ROCm eager (MIOpen / hipBLASLt / ATen kernels, fp32) with torch's fused Adam —
    what a user of the reference gets on an MI355X without this library.  A reported baseline like `cpu_baseline`, never
    `value`.  `cudnn.benchmark` (the reference sets it, unipose.py:56) is OFF by default here: on a fresh box MIOpen's
    exhaustive search took 852 s for this network and bought 7 % (93.7 vs 100.8 ms per step, profiles/r02_s_*, r02_z);
    without it the first steps cost ~40 s of kernel compilation.  The caller runs this in a child process with a time limit."""
    from oracle import unipose_oracle as O
    old = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = benchmark
    try:
        sd = {k: v.to(dev) for k, v in O.synth_state_dict(num_classes, 0, lstm=lstm_frames > 0).items()}
        params = []
        for k, v in sd.items():
            if v.is_floating_point() and "running_" not in k:
                v.requires_grad_(True)
                params.append(v)
        opt = torch.optim.Adam(params, lr=1e-4, fused=True)
        T, hs = lstm_frames, size // 8
        if T:
            x = O.synth_input((batch, T, 3, size, size), 1).to(dev)
            cm = O.synth_input((batch, T, 1, size, size), 3, "rand").to(dev)
            t = O.synth_input((batch, T, num_classes + 1, hs, hs), 2, "rand").to(dev)
        else:
            x = O.synth_input((batch, 3, size, size), 1).to(dev)
            t = O.synth_input((batch, num_classes + 1, hs, hs), 2, "rand").to(dev)

        def one():
            opt.zero_grad(set_to_none=True)
            if T:                                      # uniposeLSTM.py:116-133: T frames, summed MSE, one backward
                h = torch.zeros(batch, num_classes + 2, hs, hs, device=dev)
                c = torch.zeros(batch, num_classes + 2, hs, hs, device=dev)
                loss = 0.0
                for j in range(T):
                    heat, c, h = O.unipose_lstm_forward(sd, x, cm, j, h, c, train=True)
                    loss = loss + torch.nn.functional.mse_loss(heat, t[:, j])
            else:
                loss = torch.nn.functional.mse_loss(O.unipose_forward(sd, x, train=True), t)
            loss.backward()
            opt.step()
            return loss

        t0 = time.perf_counter()
        for _ in range(warmup):
            one()
        torch.cuda.synchronize(dev)
        t_warm = time.perf_counter() - t0
        ts = []
        for _ in range(steps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            one()
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ts)[len(ts) // 2]
        return {"value": round(batch * max(T, 1) * 1e3 / ms, 2), "unit": "images/sec", "ms_per_step": round(ms, 3),
                "kind": "port on the stock GPU path",
                "sample": f"median of {steps} fenced train steps (fwd+MSE+bwd+fused Adam) of batch {batch}"
                          f"{' x %d frames' % T if T else ''} at {size}x{size}, "
                          f"oracle graph on torch {torch.__version__} eager (MIOpen, cudnn.benchmark={benchmark}), fp32, "
                          f"after {warmup} warm-up steps ({t_warm:.1f} s incl. MIOpen's search)"}
    finally:
        torch.backends.cudnn.benchmark = old


def wasp_dilated_leg(dev, batch=32, hw=23, iters=20, bf16=False, dilations=(6, 12, 18)):
    """The quantity BASELINE.json's north_star names for the WASP branch: the dilated 3x3 256->256 convolutions
    (wasp.py:46-49, dilation 6 / 12 / 18 at 23x23; 24 is the video variant's / output-stride-8 value) timed alone with HIP
    events on the launch stream.  Reported per dilation: nominal and effective (taps that touch the image) TFLOP/s against
    the MFMA peak of the arithmetic, and the algorithmic bytes (input once + output once + weights once, SURVEY 8d: 37.0 MB
    at B=32 fp32, 35.8 MB at B=16 736x736 bf16) per second against the 8 TB/s HBM peak.  In fp32 these kernels are
    compute-bound (AI ~540 FLOP/B) and the MFMA fraction is the meaningful one; bf16=True is the same leg in bf16 storage
    at the geometry of configs[4] (46x46, B=16: the "dilated-conv HBM stress" case)."""
    from unipose_amd import _C, ops
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(batch, hw, hw, 256, generator=g).to(dev)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev)
    esize, peak = (2.0, BF16_MFMA_PEAK_TFLOPS) if bf16 else (4.0, F32_MFMA_PEAK_TFLOPS)
    if bf16:
        x = x.to(torch.bfloat16)
    rows = []
    for dil in dilations:
        cfg = ops.ConvCfg(1, dil, dil)
        y, _, _ = ops.conv_fwd_raw(x, w, cfg)
        for _ in range(3):
            ops.conv_fwd_raw(x, w, cfg, out=y)
        torch.cuda.synchronize(dev)
        # per-launch HIP events inside the library (the same instrument as the main roofline figure): the kernel's own
        # duration, free of the Python time between launches
        lib = _C.lib()
        nv = lib.up_profile_variants()
        arr = (ctypes.c_double * (nv * 3))()
        lib.up_profile_begin()
        for _ in range(iters):
            ops.conv_fwd_raw(x, w, cfg, out=y)
        torch.cuda.synchronize(dev)
        _C.check(lib.up_profile_end(arr, nv), "profile_end")
        launches = sum(arr[i * 3] for i in range(nv))
        ms = sum(arr[i * 3 + 1] for i in range(nv)) / max(launches, 1.0)
        live = sum(1 for p in range(hw) for r in (-1, 0, 1) if 0 <= p + r * dil < hw) ** 2 / float(hw * hw * 9)
        flop = 2.0 * batch * hw * hw * 256 * 256 * 9
        nbytes = esize * (x.numel() + batch * hw * hw * 256 + w.numel())
        rows.append({"dilation": dil, "ms": round(ms, 4), "live_tap_fraction": round(live, 3),
                     "nominal_tflops": round(flop / ms / 1e9, 1),
                     "effective_tflops": round(flop * live / ms / 1e9, 1),
                     "effective_mfma_frac": round(flop * live / ms / 1e9 / peak, 3),
                     "algorithmic_gbps": round(nbytes / ms / 1e6, 1),
                     "hbm_frac": round(nbytes / ms / 1e6 / 8000.0, 4)})
    return rows


HBM_PEAK_TBPS = 8.0


def make_workload(dev, lstm, K, B, S, T, seed, emu=False, init=None, graph=False):
    """Model + resident synthetic batch + the train step of the reference's loops (unipose.py:100-131 /
    uniposeLSTM.py:116-133).  Returns (model, optimizer, step(reducer=None)).
    init (tests: the G16 trajectory fixture drives THIS step function): {"state_dict", "x", "t"} replace the random
    initialisation and batch of the image model, "no_dropout" sets the three dropouts to p = 0."""
    from unipose_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    if lstm:
        from model.uniposeLSTM import unipose_lstm
        model = unipose_lstm(num_classes=K).to(dev).train()
        # the trunk runs once on all T frames of the clip batch (every convolution on B*T images), each frame with its own
        # BatchNorm batch statistics: the results of the reference's T calls (tests: lstm_case(batch_frames=True), G5)
        model.batch_frames = os.environ.get("UNIPOSE_BATCH_FRAMES", "1") != "0"
        model.batch_head = os.environ.get("UNIPOSE_BATCH_HEAD", "1") != "0"      # (A/B: the head once on all T * B hidden states)
        x = torch.randn(B, T, 3, S, S, generator=g).to(dev)
        cm = torch.rand(B, T, 1, S, S, generator=g).to(dev)
        t = torch.rand(B, T, K + 1, S // 8, S // 8, generator=g).to(dev)
    else:
        from model.unipose import unipose
        model = unipose("MPII", num_classes=K).to(dev).train()
        x = torch.randn(B, 3, S, S, generator=g).to(dev)
        t = torch.rand(B, K + 1, S // 8, S // 8, generator=g).to(dev)
        if init is not None:
            model.load_state_dict(init["state_dict"])
            x, t = init["x"].to(dev), init["t"].to(dev)
            if init.get("no_dropout"):
                for d in (model.wasp.dropout, model.decoder.last_conv[3], model.decoder.last_conv[7]):
                    d.p = 0.0
    # unipose.py:72 (no weight decay); `fused=True` is torch's own single-kernel implementation of the same update
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=not emu,  # 70.3 vs 71.7 ms/step with the foreach default
                           **({"capturable": True} if graph else {}))
    if graph:       # the whole step as ONE hipGraph (unipose_amd.graph.GraphedTrainStep): same launches, no per-layer host work
        if lstm or emu:
            raise ValueError("--graph: the image model on a GPU")
        from unipose_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, x, t)

        def graphed(reducer=None):
            if reducer is not None:
                raise ValueError("--graph does not capture the data-parallel exchange")
            return gstep()
        return model, opt, graphed

    def step(reducer=None):
        opt.zero_grad(set_to_none=True)
        if lstm:                                               # uniposeLSTM.py:116-133: T frames, ONE backward
            hs = S // 8
            heat = torch.zeros(K + 1, hs, hs, device=dev)
            cell = torch.zeros(K + 2, hs, hs, device=dev)
            hide = torch.zeros(K + 2, hs, hs, device=dev)
            loss = 0.0
            for j in range(T):
                heat, cell, hide = model(x, cm, j, heat, hide, cell)
                loss = loss + ops.mse_loss(heat, t[:, j])
        else:
            loss = ops.mse_loss(model(x), t)
        if lstm:
            with ops.deferred_wgrad():                         # every weight is used T times: see ops.deferred_wgrad
                loss.backward()
        else:
            loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss
    return model, opt, step


def profile_rows(lib, steps_profiled=1):
    nv = lib.up_profile_variants()
    arr = (ctypes.c_double * (nv * 3))()
    from unipose_amd import _C
    _C.check(lib.up_profile_end(arr, nv), "profile_end")
    live = (ctypes.c_double * nv)()
    _C.check(lib.up_profile_live_flops(live, nv), "profile_live_flops")
    rows = []
    for i in range(nv):
        n, ms, fl = arr[i * 3], arr[i * 3 + 1], arr[i * 3 + 2]
        if n:
            name = lib.up_profile_variant_name(i).decode()
            peak = BF16_MFMA_PEAK_TFLOPS if "bf16" in name else F32_MFMA_PEAK_TFLOPS
            # tflops: nominal 2 M N K per launch (SURVEY 8d's convention, the sum over a step is its 6.0 TFLOP);
            # tflops_live_taps: dilated launches charged for the (pixel, tap) pairs that touch the image only
            rows.append({"kernel": name, "launches": int(n), "avg_ms": ms / n, "total_ms": ms,
                         "tflops": fl / ms / 1e9, "tflops_live_taps": (live[i] or fl) / ms / 1e9, "peak_tflops": peak})
    # "dominant" = the variant that carries the most ALGORITHMIC FLOP.  Ranking by summed launch durations is not stable
    # here: the weight gradients run on a second stream, their event-bracketed durations include the time they are
    # starved by the main stream, and which of two kernels has the larger sum flips with the interleaving of the streams.
    rows.sort(key=lambda r: -r["tflops"] * r["total_ms"])
    return rows


def top_by_time(rows):
    """The variant with the largest summed launch duration, next to the FLOP ranking `profile_rows` returns: in the two-stream
    timed region the side-stream weight gradient can be the larger one by time although it carries fewer FLOP."""
    r = max(rows, key=lambda q: q["total_ms"])
    return {"kernel": r["kernel"], "achieved": round(r["tflops"], 2), "frac": round(r["tflops"] / r["peak_tflops"], 4),
            "total_ms": round(r["total_ms"], 3), "launches": r["launches"]}


def other_config_leg(dev, name):
    """A BASELINE.json configuration other than the headline one, timed by the same driver run (rank 0, N=1 only, a few
    seconds each): its own throughput and the roofline of its dominant MFMA kernel from per-launch HIP events."""
    from unipose_amd import _C, ops
    lib = _C.lib()
    if name == "lstm":        # configs[3]: UniPose-LSTM, K=13, 8 clips x 5 frames per GPU, BPTT
        lstm, K, B, S, T, math, steps = True, 13, 8, 368, 5, "f32", 5
        work = (f"UniPose-LSTM ResNet-101 (K={K}) train step: {T}-frame unroll, summed MSE, one backward (BPTT) + Adam, "
                f"synthetic {S}x{S}, batch {B} clips/GPU (BASELINE.json configs[3])")
        flop_img = FLOP_PER_IMAGE_FWD_BWD + 3 * 2 * 8.93e9
    else:                     # configs[4]: 736x736, B=16/GPU, bf16 MFMA arithmetic
        lstm, K, B, S, T, math, steps = False, 16, 16, 736, 1, "bf16s", 5
        work = (f"UniPose ResNet-101 (K={K}) train step: fwd + MSE + bwd + Adam, synthetic {S}x{S}, batch {B}/GPU, bf16 "
                f"storage (activations and their gradients bf16 in HBM behind the fp32 stem) + bf16 MFMA with fp32 "
                f"accumulation; fp32 BatchNorm statistics, weights, weight gradients, Adam (BASELINE.json configs[4])")
        flop_img = FLOP_PER_IMAGE_FWD_BWD * 4.0
    ops.set_conv_math(math)
    try:
        model, opt, step = make_workload(dev, lstm, K, B, S, T, seed=7)
        for _ in range(3):          # the first steps of a new shape set allocate, build tap-order / rectangle tables, pack
            step()
        torch.cuda.synchronize(dev)
        # timed like the headline since round 5: `steps` steps between two fences, twice, the better region reported.  (Until
        # round 4: the median of individually fenced steps — a fence per step exposes the start of every step, which now holds
        # the re-pack of all weights that training loops hide behind the previous step's tail: 39.3 vs 37.3 ms at 736^2.)
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) / steps)
        dt = min(times)
        dt_mean = sum(times) / len(times)
        lib.up_profile_begin()
        step()
        torch.cuda.synchronize(dev)
        rows = profile_rows(lib)
    finally:
        ops.set_conv_math("f32")
    ips = B * T / dt
    out = {"config": {"workload": work, "per_gpu_batch": B, "input": [3, S, S], "frames": T},
           "metric": "images/sec fwd+bwd" + (" (frames)" if lstm else ""), "value": round(ips, 2), "unit": "images/sec",
           "ms_per_step": round(1e3 * dt, 3), "ms_per_step_mean_of_regions": round(1e3 * dt_mean, 3),
           "ms_per_step_by_region": [round(1e3 * v, 3) for v in times], "steps": steps, "warmup": 3,
           "timing": f"{steps} steps between two fences (the headline's method), better of two regions",
           "ms_per_step_by_region": [round(1e3 * v, 2) for v in times],
           "dtype": {"f32": "f32", "bf16": "bf16 operands, fp32 storage", "bf16s": "bf16"}[math],
           "step_tflops": round(ips * flop_img / 1e12, 2)}
    if rows:
        top = rows[0]
        tot_ms = sum(r["total_ms"] for r in rows)
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": round(top["tflops"], 2),
                           "peak": top["peak_tflops"], "unit": "TFLOP/s", "frac": round(top["tflops"] / top["peak_tflops"], 4),
                           "traffic": None, "avg_launch_ms": round(top["avg_ms"], 4), "launches": top["launches"],
                           "mfma_ms_per_step": round(tot_ms, 3), "top_by_time": top_by_time(rows),
                           "by_kernel": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
                                         for r in rows[:6]]}
        if name != "lstm":
            try:      # north_star's number for the dilated branch at the "HBM stress" configuration
                out["roofline"]["wasp_dilated_bf16"] = wasp_dilated_leg(dev, batch=B, hw=S // 16, bf16=True,
                                                                        dilations=(6, 12, 18, 24))
            except Exception as e:      # a reporting extra must never cost the bench line
                log(f"bf16 wasp leg skipped: {type(e).__name__}: {e}")
    del model, opt, step
    return out


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves — the same
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the
    same arguments>` the driver's launcher form uses — and hand its stdout (rank 0's ONE JSON line) and exit status on.
    The parent process never touches the GPU or the process group."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        "HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=368)
    ap.add_argument("--num-classes", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stock-baseline", action="store_true",
                    help="skip the leg that times the same step through PyTorch-ROCm eager (MIOpen) on this GPU")
    ap.add_argument("--stock-baseline-only", action="store_true", help="(child mode) print the stock_gpu_baseline record")
    ap.add_argument("--stock-benchmark", action="store_true", help="stock baseline with cudnn.benchmark = True (minutes of search)")
    ap.add_argument("--stock-timeout", type=int, default=150)
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch hipEvent timing")
    ap.add_argument("--math", default="f32", choices=["f32", "bf16x3", "bf16", "bf16s"],
                    help="arithmetic of the forward/data-gradient convolutions: exact fp32 MFMA (default, the parity "
                         "configuration), split-bf16 fp32-equivalent, or plain bf16 operands")
    ap.add_argument("--model", default="unipose", choices=["unipose", "lstm"],
                    help="unipose = BASELINE configs[1]/[2] (default, the headline metric); lstm = configs[3]: "
                         "UniPose-LSTM, 5-frame clips, one backward through all frames, images/sec counts frames")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--force-dp", action="store_true",
                    help="run the gradient exchange (RCCL all-reduce, buckets, hooks) even with one rank")
    ap.add_argument("--overlap", action="store_true",
                    help="bucketed gradient exchange launched during backward (one hook per 32 MB bucket) instead of one flat "
                         "all-reduce after it; the default for --math bf16s, whose step (< 40 ms) is short enough for the flat "
                         "190 MB exchange to show (--no-overlap switches it off)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--allow-shared-queues", action="store_true",
                    help="run a data-parallel measurement although GPU_MAX_HW_QUEUES < 8 (the weight-gradient stream may share a "
                         "hardware queue with RCCL's streams: +10 ... +14 %% per step measured)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole training step as ONE hipGraph (unipose_amd.graph.GraphedTrainStep; image model, one GPU)")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the extra split-bf16 timing loop")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE configs[3] (LSTM) and configs[4] (736x736 bf16) legs appended at N=1")
    ap.add_argument("--wasp-only", action="store_true", help="only the WASP dilated-convolution roofline leg")
    ap.add_argument("--dry-run-emu", action="store_true",
                    help="TEST INFRASTRUCTURE (tests/test_bench_flow.py): walk the whole control flow of this script — warm-up, "
                         "timed region, exclusive pass, alt-math loop, every barrier and collective — on the CPU kernel "
                         "emulator with the gloo backend.  Prints a line marked dry_run; never a measurement")
    ap.add_argument("--settle", type=int, default=20,
                    help="untimed steps BEFORE the --warmup steps (groups of 5, logged and reported as `settle`): lets a freshly "
                         "touched GPU reach its steady state; 0 switches it off")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=min(os.cpu_count() or 1, 64),
                    help="host threads for the CPU baseline (default: min(cores, 64); more oversubscribes oneDNN)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    emu = args.dry_run_emu
    use_dist = world > 1 or args.force_dp
    if emu:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "emu"))
        import build_emu
        from unipose_amd import _C as _Cemu
        _Cemu.load(build_emu.build())
        _Cemu._ALLOW_HOST_POINTERS = True
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None      # this process only: the script's fences become no-ops
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        from unipose_amd import ops as _ops
        _ops._side_stream(dev)          # create it BEFORE RCCL creates its own streams (hardware-queue assignment)
    if use_dist and not emu:
        # the weight-gradient side stream must keep a hardware queue of its own once RCCL has created its streams:
        # this script exports GPU_MAX_HW_QUEUES=8 before `import torch` unless the caller set something else; a smaller value is
        # a measurement of the queue collision, not of the step — refuse it loudly unless asked for
        hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
        if hwq < 8 and not args.allow_shared_queues:
            raise SystemExit(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')!r}: a data-parallel run needs >= 8 hardware "
                             "queues (side stream vs RCCL's streams, +10 ... +14 % per step otherwise); unset it or pass "
                             "--allow-shared-queues")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from model.unipose import unipose
    from unipose_amd import _C, ops
    from unipose_amd.dist import GradAllReducer, shard_seed
    comm_ranks = None
    if use_dist:       # the ranks that REALLY take part in a collective: an all-reduce of ones (not get_world_size())
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        comm_ranks = int(round(float(ones.item())))
        log(f"process group up: backend {dist.get_backend()}, all-reduce of ones = {comm_ranks} (WORLD_SIZE {world})")
        if comm_ranks != world:
            raise SystemExit(f"all-reduce over {comm_ranks} ranks but WORLD_SIZE={world}")
    if args.wasp_only:
        print(json.dumps({"wasp_dilated": wasp_dilated_leg(dev)}), flush=True)
        return
    if args.stock_baseline_only:
        video = args.model == "lstm"
        print(json.dumps(stock_gpu_baseline(dev, 13 if video and args.num_classes == 16 else args.num_classes, args.size,
                                            8 if video and args.batch == 32 else args.batch, benchmark=args.stock_benchmark,
                                            lstm_frames=args.frames if video else 0)), flush=True)
        return

    K, B, S = args.num_classes, args.batch, args.size
    lstm = args.model == "lstm"
    T = args.frames if lstm else 1
    torch.manual_seed(0)
    if lstm:
        if args.num_classes == 16:
            K = 13                                             # Penn Action joints (configs[3])
        if args.batch == 32:
            B = 8
    if args.graph:          # the capture fixes the arithmetic: set it first; per-launch events cannot ride in a replay
        ops.set_conv_math(args.math)
        args.no_profile = True
    model, opt, step1 = make_workload(dev, lstm, K, B, S, T, seed=shard_seed(0, rank), emu=emu, graph=args.graph)
    ops.manual_seed(shard_seed(0, rank))
    ops.set_conv_math(args.math)
    # default: ONE flat all-reduce after backward (190 MB: ~1-2 ms on xGMI against a 66 ms fp32 step); --overlap: 32 MB
    # buckets launched during backward by one hook per bucket (unipose_amd/dist.py)
    reducer = None
    args.overlap = (args.overlap or args.math == "bf16s") and not args.no_overlap
    if use_dist:
        reducer = GradAllReducer(model, bucket_bytes=(32 << 20) if args.overlap else (256 << 20), force=args.force_dp,
                                 overlap=args.overlap)
    xch = {"ms": 0.0, "n": 0, "on": False}
    if reducer is not None:       # host-side duration of the exchange call (copy into the flat buffer, all-reduce, wait) on
        inner_finish = reducer.finish          # every step of the timed region; with --overlap only what is NOT hidden

        def timed_finish():
            if not xch["on"]:
                return inner_finish()
            if dev.type == "cuda":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                inner_finish()
                e1.record()
                xch.setdefault("events", []).append((e0, e1))
            else:
                t_ = time.perf_counter()
                inner_finish()
                xch["ms"] += (time.perf_counter() - t_) * 1e3
            xch["n"] += 1
        reducer.finish = timed_finish

    def step():
        return step1(reducer)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"model on {dev}, {sum(p.numel() for p in model.parameters())} parameters; warm-up")
    # Settling phase (untimed, before the W warm-up steps; fixed count so that every rank runs the same collectives): a process
    # that starts its timed region within ~2 s of first touching the GPU has read 65-67 ms per step on boxes whose later
    # processes read 61.5-62.4 (r04_y / r04_z / r05_z final sessions: timed region +5.7 %, the exclusive pass right behind it
    # +1.6 %, the loop after that +1 % — a ramp, not a property of the step).  The groups are logged and reported (`settle`).
    settle = {"steps": 0, "ms_per_step_by_group": []}
    if not emu and args.settle > 0:
        for _ in range(args.settle // 5):
            torch.cuda.synchronize(dev)
            ts = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize(dev)
            settle["ms_per_step_by_group"].append(round((time.perf_counter() - ts) * 200.0, 2))
            settle["steps"] += 5
        log(f"settling: {settle['steps']} untimed steps, ms per step by group of 5: {settle['ms_per_step_by_group']}")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize(dev)
        log(f"warm-up step {i} done")
    fence()
    profile = (not args.no_profile) and rank == 0
    if profile:
        _C.lib().up_profile_begin()
    xch["on"] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        if profile:        # every 5th step of the timed region is bracketed launch by launch (~1.7 ms per bracketed step:
            _C.lib().up_profile_enable(1 if i % PROFILE_EVERY == 0 else 0)      # events are recycled, round 4; 62.8 vs 62.3 ms at every 4th)
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    xch["on"] = False
    log(f"timed region: {args.steps} steps in {dt:.3f}s")
    loss_val = float(loss.detach())
    dp = None
    if use_dist:
        # first-run proof for N > 1: every rank's own clock, the exchange's share, and whether the replicas still agree
        for e0, e1 in xch.get("events", []):
            xch["ms"] += e0.elapsed_time(e1)
        mine = torch.tensor([dt, xch["ms"] / max(xch["n"], 1),
                             float(sum(p.detach().double().sum() for p in model.parameters()))],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dp = {"comm_ranks": comm_ranks, "backend": dist.get_backend(), "exchange": "overlap" if args.overlap else "flat",
              "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
              "payload_mb": round(reducer.payload_bytes() / 1e6, 1),
              "ms_per_step_by_rank": [round(1e3 * float(r[0]) / args.steps, 3) for r in allr],
              "exchange_ms_by_rank": [round(float(r[1]), 3) for r in allr],
              "weights_identical_across_ranks": all(float(r[2]) == float(allr[0][2]) for r in allr)}
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    roofline = None
    if profile:
        rows = profile_rows(_C.lib())
        if rows:
            top = rows[0]
            tot_ms = sum(r["total_ms"] for r in rows)
            tot_fl = sum(r["tflops"] * r["total_ms"] for r in rows)
            roofline = {"bound": "mfma", "kernel": top["kernel"], "achieved": round(top["tflops"], 2),
                        "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(top["tflops"] / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                        "avg_launch_ms": round(top["avg_ms"], 4), "launches": top["launches"],
                        "achieved_live_taps": round(top["tflops_live_taps"], 2),
                        "frac_live_taps": round(top["tflops_live_taps"] / F32_MFMA_PEAK_TFLOPS, 4),
                        "flop_convention": "achieved / frac: nominal 2*M*N*K per launch (SURVEY 8d: 187.7 GFLOP per image); *_live_taps: "
                                           "dilated launches (WASP, layer4) charged for their live (pixel, tap) pairs only",
                        "dominant_by": "algorithmic FLOP in the timed region (see profile_rows)",
                        "top_by_time": top_by_time(rows),
                        "all_mfma_kernels": {"achieved": round(tot_fl / tot_ms, 2),
                                             "frac": round(tot_fl / tot_ms / F32_MFMA_PEAK_TFLOPS, 4),
                                             "ms_per_step": round(tot_ms / ((args.steps + PROFILE_EVERY - 1) // PROFILE_EVERY), 3)},
                        "by_kernel": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                                       if k != "peak_tflops"} for r in rows]}
            # HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 passes (profiles/pmc_traffic.json
            # records the last ones and how they were taken); null when no record matches
            try:
                pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                  "pmc_traffic.json")))
                rec = pmc["kernels"].get(top["kernel"])
                if rec and args.math == "f32" and not lstm and S == 368 and B == 32:
                    roofline["traffic"] = round(rec["traffic_mb"] * 1e6)
                    roofline["traffic_unit"] = "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes, " \
                                               f"profiles/{pmc['tag']}_pmc_summary.txt)"
                    roofline["traffic_source"] = (f"builder PMC pass, tag {pmc['tag']} (profiles/pmc_traffic.json): counters need "
                                                  "their own rocprofv3 runs, this value is NOT measured inside this bench run")
            except (OSError, ValueError, KeyError):
                pass

    # The timed region runs the weight gradients on a second stream, so the event-bracketed durations above include
    # time shared with the other stream's kernels.  A few extra steps with everything on ONE stream give each kernel's
    # own duration (same kernels, same shapes, nothing else resident).  EVERY rank runs these steps — they contain the
    # gradient exchange and the barriers of fence() — only rank 0 records events.
    if not args.no_profile:
        was_async, ops.ASYNC_WGRAD = ops.ASYNC_WGRAD, False
        step()
        fence()
        if profile:
            _C.lib().up_profile_begin()
        nx = min(args.steps, 5)
        for _ in range(nx):
            step()
        fence()
        ops.ASYNC_WGRAD = was_async
        if profile:
            xrows = profile_rows(_C.lib())
            same = [r for r in xrows if roofline is not None and r["kernel"] == roofline["kernel"]]
            if same:
                x_ms = sum(r["total_ms"] for r in xrows)
                x_fl = sum(r["tflops"] * r["total_ms"] for r in xrows)
                roofline["exclusive"] = {
                    "note": "same step with both streams serialised: the kernel's own duration",
                    "kernel": roofline["kernel"], "achieved": round(same[0]["tflops"], 2),
                    "frac": round(same[0]["tflops"] / F32_MFMA_PEAK_TFLOPS, 4),
                    "achieved_live_taps": round(same[0]["tflops_live_taps"], 2),
                    "frac_live_taps": round(same[0]["tflops_live_taps"] / F32_MFMA_PEAK_TFLOPS, 4),
                    "avg_launch_ms": round(same[0]["avg_ms"], 4), "launches": same[0]["launches"],
                    "all_mfma_kernels": {"achieved": round(x_fl / x_ms, 2),
                                         "frac": round(x_fl / x_ms / F32_MFMA_PEAK_TFLOPS, 4),
                                         "ms_per_step": round(x_ms / nx, 3)}}

    if roofline is not None and args.math == "f32":
        try:
            roofline["wasp_dilated"] = wasp_dilated_leg(dev)
        except Exception as e:      # a reporting extra must never cost the bench line
            log(f"wasp leg skipped: {e}")

    # second arithmetic on the same workload (reported as `alt_math`, never as `value`): the split-bf16
    # fp32-equivalent convolution kernels (parity 1.6e-5 on the reference golden, argmax bit-exact)
    alt = None
    if args.math == "f32" and not args.no_alt_math:
        ops.set_conv_math("bf16x3")
        for _ in range(2):
            step()
        fence()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dta = time.perf_counter() - ta
        ops.set_conv_math("f32")
        if use_dist:
            tt = torch.tensor([dta], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dta = float(tt.item())
        alt = {"math": "bf16x3: fwd/dgrad convolutions as 3 bf16 MFMAs on (hi,lo)-split fp32 operands, fp32 accumulate; "
                       "wgrad and everything else unchanged (fp32)",
               "value": round(world * B * T * args.steps / dta, 2), "unit": "images/sec",
               "ms_per_step": round(1e3 * dta / args.steps, 3),
               "parity": "tests/test_model_gpu.py::test_eval_golden_bf16_operand_modes (1e-3, bit-exact argmax)"}
        log(f"alt math bf16x3: {alt['value']} images/sec")

    if rank == 0:
        ips = world * B * T * args.steps / dt
        out = {
            "metric": (f"images/sec fwd+bwd, UniPose-LSTM ResNet-101 {T}-frame clips {S}x{S}" if lstm else
                       f"images/sec fwd+bwd, UniPose ResNet-101 {S}x{S}"),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (split-bf16 MFMA, fp32-equivalent)", "bf16": "bf16 operands, fp32 storage",
                      "bf16s": "bf16"}[args.math],
            "data": "synthetic",
            "config": {"workload": (f"UniPose-LSTM ResNet-101 (K={K}) train step: {T}-frame unroll, summed MSE, one "
                                    f"backward (BPTT) + Adam, synthetic {S}x{S}, batch {B} clips/GPU (BASELINE.json "
                                    f"configs[3])" if lstm else
                                    f"UniPose ResNet-101 (K={K}) train step: fwd + MSE + bwd + Adam, synthetic "
                                    f"{S}x{S}, batch {B}/GPU (BASELINE.json configs[1]; configs[2] for N>1)"),
                       "global_batch": world * B, "per_gpu_batch": B, "input": [3, S, S],
                       "parallelism": f"dp{world}", "optimizer": "torch.optim.Adam(lr=1e-4, fused=True)",
                       "arithmetic": {"f32": "fp32 MFMA 32x32x2 everywhere",
                                      "bf16x3": "fwd/dgrad: 3x bf16 MFMA 32x32x16 on (hi,lo) split operands; wgrad: fp32 MFMA",
                                      "bf16": "bf16 MFMA 32x32x16 (fwd, dgrad, wgrad), operands rounded from fp32 tensors",
                                      "bf16s": "bf16 MFMA 32x32x16 (fwd, dgrad, wgrad) on bf16 tensors; fp32 stem"}[args.math]},
            "step_tflops_per_gpu": round(ips / world * (FLOP_PER_IMAGE_FWD_BWD + (3 * 2 * 8.93e9 if lstm else 0.0))
                                         * (S / 368.0) ** 2 / 1e12, 2),
            "loss": loss_val,
        }
        if args.graph:
            out["config"]["step_launch"] = "one hipGraph replay per step (unipose_amd.graph.GraphedTrainStep, capturable fused Adam)"
        if settle["steps"]:
            out["settle"] = settle
        if emu:
            out["dry_run"] = "CPU emulator + gloo: control-flow test only, the numbers mean nothing"
            out["metric"] = "DRY RUN (not a measurement): " + out["metric"]
        if dp:
            out["rccl_ranks"] = dp["comm_ranks"]
            out["data_parallel"] = dp
        if roofline:
            out["roofline"] = roofline
        if alt:
            out["alt_math"] = alt
        if world == 1 and not emu and not args.no_other_configs and not lstm and S == 368 and args.math == "f32":
            out["other_configs"] = []
            for name in ("lstm", "736_bf16"):
                try:
                    log(f"other config: {name}")
                    out["other_configs"].append(other_config_leg(dev, name))
                except Exception as e:      # a reporting extra must never cost the bench line
                    log(f"other config {name} skipped: {type(e).__name__}: {e}")
        if world == 1 and not emu and not args.no_stock_baseline and not lstm and args.math == "f32" and S == 368:
            # child process with a hard time limit: MIOpen compiles / searches kernels on first use and how long that takes
            # is not ours to bound; a reporting extra must never cost the bench line
            import subprocess
            log(f"stock GPU baseline (oracle graph on PyTorch-ROCm eager, child process, limit {args.stock_timeout}s)")
            cmd = [sys.executable, os.path.abspath(__file__), "--stock-baseline-only", "--batch", str(B), "--size", str(S),
                   "--num-classes", str(K)] + (["--stock-benchmark"] if args.stock_benchmark else [])
            try:
                cp = subprocess.run(cmd, capture_output=True, text=True, timeout=args.stock_timeout, start_new_session=True)
                rec = json.loads(cp.stdout.strip().splitlines()[-1])
                out["stock_gpu_baseline"] = rec
                out["vs_stock_gpu"] = round(out["value"] / rec["value"], 3)
            except Exception as e:          # noqa: BLE001  (timeout, MIOpen failure, no JSON)
                log(f"stock GPU baseline skipped: {type(e).__name__}: {str(e)[:200]}")
        if world == 1 and not args.no_cpu_baseline and not lstm:
            log("cpu baseline (oracle on host cores)")
            out["cpu_baseline"] = cpu_baseline(K, S, args.cpu_batch, args.cpu_steps, args.cpu_threads)
        line = json.dumps(out)
    else:
        line = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON is the LAST line on stdout: RCCL prints a version banner through C stdio, which (redirected to a file)
    # would otherwise be flushed after it at process exit
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
