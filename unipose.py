#!/usr/bin/env python
"""Image-model driver: the working counterpart of the reference's ``unipose.py`` (which has a syntax error at :97-98
and calls a loader factory with the wrong arity).  Same flags (``unipose.py:248-254``) plus the ones
needed to run without the LSP/MPII files; the loops live in ``unipose_amd/trainer.py``.

    python unipose.py --dataset MPII --epochs 1 --train_batches 8 --batch_size 32       # synthetic annotations
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 unipose.py --dataset MPII ...
"""
import argparse


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--pretrained", default=None, type=str, dest="pretrained")
    p.add_argument("--dataset", type=str, dest="dataset", default="LSP")
    p.add_argument("--train_dir", default=None, type=str, dest="train_dir")
    p.add_argument("--val_dir", default=None, type=str, dest="val_dir")
    p.add_argument("--model_name", default=None, type=str)
    p.add_argument("--model_arch", default="unipose", type=str)
    # additions
    p.add_argument("--epochs", default=100, type=int, help="reference: 100 (unipose.py:257)")
    p.add_argument("--batch_size", default=None, type=int, help="reference: 8 (unipose.py:48)")
    p.add_argument("--size", default=368, type=int)
    p.add_argument("--train_batches", default=4, type=int, help="synthetic batches per epoch")
    p.add_argument("--val_batches", default=2, type=int)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from unipose_amd.trainer import Trainer, init_distributed
    rank, world, dev = init_distributed()
    if args.train_dir or args.val_dir:
        raise SystemExit("the reference's dataset classes need OpenCV and are not part of this build: pass your own "
                         "loaders to unipose_amd.trainer.Trainer(args, train_loader, val_loader) or omit the flags "
                         "to run on synthetic annotations")
    trainer = Trainer(args, device=dev)
    for epoch in range(0, args.epochs):
        trainer.training(epoch)
        if rank == 0:
            trainer.validation(epoch)
        if world > 1:                 # the other ranks wait here, not inside the next epoch's first gradient all-reduce
            import torch.distributed as dist
            dist.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
