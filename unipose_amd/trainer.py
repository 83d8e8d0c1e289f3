"""Training / validation / test loops around the drop-in modules.

The reference's two drivers cannot run as shipped (a syntax error, imports of modules that do
not exist, a data-loader factory whose arity does not match its call sites).  This module restates what their
``Trainer`` classes do — ``unipose.py:37-231`` (image) and ``uniposeLSTM.py:36-260`` (video) — around the MI355X path:

* hyper-parameters, optimiser, step learning-rate policy and loss exactly as the reference sets them
  (``unipose.py:45-53,70-72``, ``utils/utils.py:42-51``);
* the per-batch order zero_grad -> forward -> MSE -> backward -> step (``unipose.py:100-131``), the five-frame unroll
  with a summed loss and ONE backward (``uniposeLSTM.py:100-138``);
* validation with the running AP / PCK / PCKh means of ``unipose.py:133-196`` / ``uniposeLSTM.py:141-220``, but the
  argmax and the distance arithmetic stay on the device (``ops.accuracy``, N3) instead of a D2H copy + numpy per batch;
* best-model checkpointing through ``checkpoint.save_checkpoint`` (``utils/utils.py:53-56``), key-filtered
  ``--pretrained`` loading (``unipose.py:78-90``);
* targets and input normalisation are produced on the device from key-point annotations when the loader hands over
  raw pixels + annotations (``ops.make_heatmaps`` / ``make_centermaps`` / ``normalize_image``, N2); a loader that
  yields the reference's ready-made ``(input, heatmap, centermap, img_path)`` tuples is accepted as well;
* one process per GPU: with ``WORLD_SIZE`` > 1 the gradients are averaged by ``dist.GradAllReducer`` each step.

The reference's dataset classes need OpenCV and files on disk (out of scope); ``SyntheticPoseData`` stands
in with random annotations of the same shapes and value ranges so that the loops can be exercised anywhere.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import numpy as np
import torch

from . import checkpoint, ops

NUM_CLASSES = {"LSP": 14, "MPII": 16, "Penn_Action": 13, "COCO": 17, "BBC": 7, "NTID": 18, "PoseTrack": 17}


class AverageMeter:
    """utils/utils.py:25-40."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0.0
        self.avg = 0.0
        self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def adjust_learning_rate(optimizer, iters, base_lr, gamma, step_size, policy="step", multiple=(1,)):
    """utils/utils.py:42-51: ``lr = base_lr * gamma ** (iters // step_size)`` written into every parameter group."""
    if policy == "fixed":
        lr = base_lr
    elif policy == "step":
        lr = base_lr * (gamma ** (iters // step_size))
    else:
        raise ValueError(f"unknown learning-rate policy {policy!r}")
    for i, group in enumerate(optimizer.param_groups):
        group["lr"] = lr * multiple[i if i < len(multiple) else -1]
    return lr


class PoseMetrics:
    """The running means of ``Trainer.validation`` (unipose.py:139-177, uniposeLSTM.py:146-200).

    Channel 0 (the mean over visible joints of a batch) is averaged over ALL evaluations, every joint channel only
    over the evaluations in which that joint was visible."""

    def __init__(self, num_classes: int):
        self.n = num_classes
        self.AP = np.zeros(num_classes + 1)
        self.PCK = np.zeros(num_classes + 1)
        self.PCKh = np.zeros(num_classes + 1)
        self.count = np.zeros(num_classes + 1)
        self.evals = 0

    def update(self, acc, acc_PCK, acc_PCKh, visible, video: bool = False):
        i = self.evals
        self.AP[0] = (self.AP[0] * i + acc[0]) / (i + 1)
        self.PCK[0] = (self.PCK[0] * i + acc_PCK[0]) / (i + 1)
        self.PCKh[0] = (self.PCKh[0] * i + acc_PCKh[0]) / (i + 1)
        # the video driver's joint loop starts at channel 0 (uniposeLSTM.py:194) and so overwrites the three
        # lines above whenever visible[0] == 1; the image driver starts at 1 (unipose.py:167)
        for j in range(0 if video else 1, self.n + 1):
            if visible[j] == 1:
                c = self.count[j]
                self.AP[j] = (self.AP[j] * c + acc[j]) / (c + 1)
                self.PCK[j] = (self.PCK[j] * c + acc_PCK[j]) / (c + 1)
                self.PCKh[j] = (self.PCKh[j] * c + acc_PCKh[j]) / (c + 1)
                self.count[j] += 1
        self.evals += 1

    @property
    def mAP(self):
        return self.AP[1:].sum() / self.n

    @property
    def mPCK(self):
        return self.PCK[1:].sum() / self.n

    @property
    def mPCKh(self):
        return self.PCKh[1:].sum() / self.n

    def table(self, dataset: str) -> str:
        """Plain-text stand-in for ``printAccuracies`` (utils/utils.py:354-475 prints one hand-written block per
        dataset): the same three means followed by the per-joint values."""
        lines = [f"mAP:   {self.mAP * 100:6.2f}%   mPCK:  {self.mPCK * 100:6.2f}%   mPCKh: {self.mPCKh * 100:6.2f}%   ({dataset})"]
        for name, v in (("AP", self.AP), ("PCK", self.PCK), ("PCKh", self.PCKh)):
            lines.append(f"{name:5s}" + " ".join(f"{x * 100:6.2f}" for x in v[1:]))
        return "\n".join(lines)


class SyntheticPoseData:
    """Random stand-in for ``getDataloader`` (utils/utils.py:231-352) with the shapes and value ranges of the LSP/MPII
    loaders (lsp_lspet_data.py:208-249): pixel values 0..255, key points anywhere in the image (a negative coordinate
    marks an invisible joint, as in the LSP annotations), a person centre near the middle.  ``frames`` > 0 yields
    clips (B, T, ...) like penn_action_data.py:47-170.  Annotations are host arrays; images, targets and centre maps
    are made on the device by ``DeviceBatcher``."""

    def __init__(self, num_classes: int, batch_size: int, batches: int, size: int = 368, frames: int = 0,
                 seed: int = 0, invisible: float = 0.1, device="cpu"):
        self.k, self.b, self.n, self.size, self.t = num_classes, batch_size, batches, size, frames
        self.seed, self.inv, self.device = seed, invisible, torch.device(device)

    def __len__(self):
        return self.n

    def __iter__(self):
        rng = np.random.default_rng(self.seed)
        lead = (self.b, self.t) if self.t else (self.b,)
        for i in range(self.n):
            kpt = rng.uniform(0, self.size - 1, size=lead + (self.k, 2))
            kpt[rng.uniform(size=lead + (self.k,)) < self.inv] = -1.0
            center = self.size / 2 + rng.uniform(-8, 8, size=lead + (2,))
            g = torch.Generator().manual_seed(self.seed * 100003 + i)
            pixels = torch.randint(0, 256, lead + (self.size, self.size, 3), generator=g, dtype=torch.uint8)
            yield {"pixels": pixels, "kpts": kpt, "center": center, "img_path": [f"synthetic/{i}_{j}" for j in range(self.b)]}


class DeviceBatcher:
    """Turns one loader item into the ``(input, heatmap, centermap)`` device tensors the loops consume.

    * dict with ``pixels`` (…,H,W,3) uint8/float + ``kpts`` (…,K,2) + ``center`` (…,2): normalisation ``(x-128)/256``
      and the Gaussian targets (sigma, stride, 0.0099 cut, background = 1 - max; lsp_lspet_data.py:224-245) run as
      kernels where the loss reads them;
    * the reference's tuple ``(input, heatmap, centermap, img_path)``: moved to the device unchanged."""

    def __init__(self, device, stride: float = 8, sigma: float = 3.0, center_sigma: float = 3.0):
        self.dev, self.stride, self.sigma, self.csigma = torch.device(device), stride, sigma, center_sigma

    def __call__(self, item):
        if isinstance(item, dict):
            px = item["pixels"].to(self.dev)
            lead = px.shape[:-3]
            h, w = px.shape[-3], px.shape[-2]
            x = ops.normalize_image(px.reshape((-1, h, w, 3)).float())
            k = np.asarray(item["kpts"], dtype=np.float64)
            heat = ops.make_heatmaps(k.reshape((-1,) + k.shape[-2:]), h, w, self.stride, self.sigma, self.dev)
            c = np.asarray(item["center"], dtype=np.float64).reshape(-1, 2)
            cm = ops.make_centermaps(c, h, w, self.csigma, self.dev)
            return (x.reshape(lead + x.shape[1:]), heat.reshape(lead + heat.shape[1:]), cm.reshape(lead + cm.shape[1:]))
        inp, heat, cm = item[0], item[1], item[2]
        return inp.to(self.dev), heat.to(self.dev), cm.to(self.dev)


def _rank_seed(base: int) -> int:
    """Every rank of a data-parallel run draws its own synthetic shard (seed = base + 1000 * rank)."""
    return base + 1000 * int(os.environ.get("RANK", "0"))


def _progress(loader, desc):
    try:
        from tqdm import tqdm
        return tqdm(loader, desc=desc, disable=os.environ.get("UNIPOSE_NO_TQDM") == "1")
    except Exception:   # pragma: no cover
        return loader


class _TrainerBase:
    # unipose.py:45-53 / uniposeLSTM.py:45-55
    workers = 1
    weight_decay = 0.0005     # set and never used by the reference; kept for completeness
    momentum = 0.9
    lr = 0.0001
    gamma = 0.333
    step_size = 13275
    stride = 8

    def _setup(self, model, args, device):
        self.args = args
        self.dataset = args.dataset
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.criterion = ops.mse_loss                                  # nn.MSELoss() (unipose.py:70), fused kernel
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.lr)   # unipose.py:72
        self.iters = 0
        self.isBest = 0
        self.bestPCK = 0
        self.bestPCKh = 0
        self.reducer = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            from .dist import GradAllReducer
            self.reducer = GradAllReducer(self.model)
        self.load_report = None
        if getattr(args, "pretrained", None):
            self.load_report = self._load_pretrained(args.pretrained)

    def _lr_step(self):
        return adjust_learning_rate(self.optimizer, self.iters, self.lr, policy="step", gamma=self.gamma,
                                    step_size=self.step_size)

    def _optim_step(self):
        if self.reducer is not None:
            self.reducer.finish()
        ops.wgrad_fence(self.device)          # weight gradients of the side stream (also fenced at the end of backward)
        self.optimizer.step()
        self.iters += 1

    def _after_validation(self, metrics: PoseMetrics):
        print(metrics.table(self.dataset))
        saved = None
        if metrics.mAP > self.isBest:                                  # unipose.py:183-186
            self.isBest = metrics.mAP
            if getattr(self.args, "model_name", None):
                saved = checkpoint.save_checkpoint({"state_dict": self.model.state_dict()}, True, self.args.model_name)
                print("Model saved to " + str(saved))
        self.bestPCKh = max(self.bestPCKh, metrics.mPCKh)
        self.bestPCK = max(self.bestPCK, metrics.mPCK)
        print("Best AP = %.2f%%; PCK = %2.2f%%; PCKh = %2.2f%%" % (self.isBest * 100, self.bestPCK * 100,
                                                                 self.bestPCKh * 100))
        return saved


class Trainer(_TrainerBase):
    """Image model (reference ``unipose.py:37-231``)."""
    batch_size = 8
    sigma = 3

    def __init__(self, args, train_loader: Optional[Iterable] = None, val_loader: Optional[Iterable] = None,
                 device="cuda"):
        from .unipose import unipose
        if args.dataset not in NUM_CLASSES:
            raise ValueError(f"unknown dataset {args.dataset!r}")
        self.numClasses = NUM_CLASSES[args.dataset]                    # unipose.py:57-60
        self.batch_size = getattr(args, "batch_size", None) or self.batch_size
        size = getattr(args, "size", 368)
        dev = torch.device(device)
        self.train_loader = train_loader if train_loader is not None else SyntheticPoseData(
            self.numClasses, self.batch_size, getattr(args, "train_batches", 4), size, seed=_rank_seed(1))
        self.val_loader = val_loader if val_loader is not None else SyntheticPoseData(
            self.numClasses, self.batch_size, getattr(args, "val_batches", 2), size, seed=2)
        self.batcher = DeviceBatcher(dev, self.stride, self.sigma)
        model = unipose(args.dataset, num_classes=self.numClasses, backbone="resnet", output_stride=16, sync_bn=True,
                        freeze_bn=False, stride=self.stride)
        self._setup(model, args, dev)

    def _load_pretrained(self, path):
        return checkpoint.load_checkpoint(self.model, path)            # unipose.py:78-90

    def training(self, epoch):
        train_loss = 0.0
        self.model.train()
        print("Epoch " + str(epoch) + ":")
        bar = _progress(self.train_loader, "train")
        i = -1
        for i, item in enumerate(bar):
            self._lr_step()
            input_var, heatmap_var, _ = self.batcher(item)
            self.optimizer.zero_grad()
            heat = self.model(input_var)
            loss = self.criterion(heat, heatmap_var)
            train_loss += loss.item()
            loss.backward()
            self._optim_step()
            if hasattr(bar, "set_description"):
                bar.set_description("Train loss: %.6f" % (train_loss / ((i + 1) * self.batch_size)))
            if i == 10000:                                             # unipose.py:130-131
                break
        return train_loss / max(i + 1, 1)

    @torch.no_grad()
    def validation(self, epoch):
        self.model.eval()
        metrics = PoseMetrics(self.numClasses)
        val_loss = 0.0
        for i, item in enumerate(_progress(self.val_loader, "val")):
            input_var, heatmap_var, _ = self.batcher(item)
            heat = self.model(input_var)
            val_loss += self.criterion(heat, heatmap_var).item()
            acc, acc_PCK, acc_PCKh, _, _, visible = ops.accuracy(heat, heatmap_var, 0.2, 0.5, self.dataset)
            metrics.update(acc, acc_PCK, acc_PCKh, visible)
        self._after_validation(metrics)
        return metrics

    @torch.no_grad()
    def test(self, pixels_hwc: torch.Tensor):
        """``Trainer.test`` (unipose.py:200-246) without the drawing: (H,W,3) pixel values of an already resized image ->
        key points in image coordinates (``get_kpts`` on the heat-maps up-sampled to the input size)."""
        self.model.eval()
        x = ops.normalize_image(pixels_hwc.to(self.device).float().unsqueeze(0))
        heat = self.model(x)
        h, w = x.shape[2], x.shape[3]
        up = ops.ToNCHW.apply(ops.Bilinear.apply(ops.ToNHWC.apply(heat), h, w), heat.shape[1])
        return ops.get_kpts(up, img_h=float(h), img_w=float(w)), up


class VideoTrainer(_TrainerBase):
    """UniPose-LSTM (reference ``uniposeLSTM.py:36-260``)."""
    batch_size = 1
    sigma = 1                      # uniposeLSTM.py:53
    frame_memory = 5               # uniposeLSTM.py:43

    def __init__(self, args, train_loader: Optional[Iterable] = None, val_loader: Optional[Iterable] = None,
                 device="cuda"):
        from .uniposeLSTM import unipose as unipose_lstm
        self.numClasses = NUM_CLASSES.get(args.dataset, 13)            # Penn_Action: 13 (uniposeLSTM.py:58-59)
        self.batch_size = getattr(args, "batch_size", None) or self.batch_size
        self.frame_memory = getattr(args, "frame_memory", None) or self.frame_memory
        size = getattr(args, "size", 368)
        dev = torch.device(device)
        mk = lambda n, seed: SyntheticPoseData(self.numClasses, self.batch_size, n, size, frames=self.frame_memory,  # noqa: E731
                                               seed=seed)
        self.train_loader = train_loader if train_loader is not None else mk(getattr(args, "train_batches", 2), _rank_seed(1))
        self.val_loader = val_loader if val_loader is not None else mk(getattr(args, "val_batches", 1), 2)
        self.batcher = DeviceBatcher(dev, self.stride, self.sigma)
        model = unipose_lstm(num_classes=self.numClasses, backbone="resnet", output_stride=16, sync_bn=True,
                             freeze_bn=False, stride=self.stride)
        model.batch_frames = True      # this loop always walks all frames of a clip: trunk once per clip batch, not per frame
        self._setup(model, args, dev)

    def _load_pretrained(self, path):
        # uniposeLSTM.py:77-93: key-filtered load with prefix = 'invalid' (drops nothing by name); a tensor whose shape
        # does not fit (the 1x1 output layer of an image checkpoint with another joint count) is skipped and reported
        return checkpoint.load_checkpoint(self.model, path, skip_prefix=("invalid",))

    def _zero_state(self, hw):
        z = lambda c: torch.zeros(c, hw[0], hw[1], device=self.device)  # noqa: E731  (uniposeLSTM.py:116-118)
        return z(self.numClasses + 1), z(self.numClasses + 2), z(self.numClasses + 2)

    def _unroll(self, input_var, heatmap_var, centermap_var, on_frame=None):
        hw = (input_var.shape[-2] // self.stride, input_var.shape[-1] // self.stride)
        heat, cell, hide = self._zero_state(hw)
        loss = 0
        for j in range(self.frame_memory):
            heat, cell, hide = self.model(input_var, centermap_var, j, heat, hide, cell)
            lj = self.criterion(heat, heatmap_var[:, j])
            loss = loss + lj
            if on_frame is not None:
                on_frame(j, heat)
        return loss

    def training(self, epoch):
        train_loss = 0.0
        self.model.train()
        print("Epoch " + str(epoch) + ":")
        bar = _progress(self.train_loader, "train")
        i = -1
        for i, item in enumerate(bar):
            self._lr_step()
            input_var, heatmap_var, centermap_var = self.batcher(item)
            self.optimizer.zero_grad()
            loss = self._unroll(input_var, heatmap_var, centermap_var)
            train_loss += loss.item()
            with ops.deferred_wgrad():                                 # every weight is used once per frame
                loss.backward()                                        # one backward through all frames
            self._optim_step()
            if hasattr(bar, "set_description"):
                bar.set_description("Train loss: %.6f" % (train_loss / ((i + 1) * self.batch_size)))
        return train_loss / max(i + 1, 1)

    @torch.no_grad()
    def validation(self, epoch):
        self.model.eval()
        metrics = PoseMetrics(self.numClasses)
        for item in _progress(self.val_loader, "val"):
            input_var, heatmap_var, centermap_var = self.batcher(item)

            def on_frame(j, heat):
                acc, acc_PCK, acc_PCKh, _, _, visible = ops.accuracy(heat, heatmap_var[:, j], 0.2, 0.5, self.dataset)
                metrics.update(acc, acc_PCK, acc_PCKh, visible, video=True)

            self._unroll(input_var, heatmap_var, centermap_var, on_frame)
        self._after_validation(metrics)
        return metrics


def init_distributed():
    """One process per GPU (torchrun environment): returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if dev.type == "cuda":
            from . import ops as _ops
            _ops._side_stream(dev)          # before RCCL creates its streams (hardware-queue mapping)
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group("gloo")
    return rank, world, dev
