#!/usr/bin/env python
"""Video-model driver: the working counterpart of the reference's ``uniposeLSTM.py`` (imports a module its tree does
not contain, :26; its Penn Action loader cannot run).  Same flags (``uniposeLSTM.py:272-278``);
dataset Penn_Action and five-frame clips are fixed like there (:284-289); the loops live in ``unipose_amd/trainer.py``.
"""
import argparse


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--pretrained", default=None, type=str, dest="pretrained")
    p.add_argument("--dataset", type=str, dest="dataset", default="Penn_Action")
    p.add_argument("--train_dir", default=None, type=str, dest="train_dir")
    p.add_argument("--val_dir", default=None, type=str, dest="val_dir")
    p.add_argument("--model_name", default=None, type=str)
    p.add_argument("--model_arch", default="unipose", type=str)
    # additions
    p.add_argument("--epochs", default=100, type=int)
    p.add_argument("--batch_size", default=None, type=int, help="clips per step; reference: 1 (uniposeLSTM.py:49)")
    p.add_argument("--frame_memory", default=5, type=int)
    p.add_argument("--size", default=368, type=int)
    p.add_argument("--train_batches", default=2, type=int)
    p.add_argument("--val_batches", default=1, type=int)
    args = p.parse_args(argv)
    args.dataset = "Penn_Action"
    return args


def main(argv=None):
    args = parse_args(argv)
    from unipose_amd.trainer import VideoTrainer, init_distributed
    rank, world, dev = init_distributed()
    if args.train_dir or args.val_dir:
        raise SystemExit("the Penn Action loader of the reference is not part of this build: pass your own loaders to "
                         "unipose_amd.trainer.VideoTrainer(args, train_loader, val_loader)")
    trainer = VideoTrainer(args, device=dev)
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
