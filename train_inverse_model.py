#!/usr/bin/env python
"""CLI of the reference's ``train_inverse_model.py`` (flags :21-28, loop :86-200) on the HIP kernels.

    python train_inverse_model.py --batch_size 128 --lr 0.0001 --lr_decay 0.1 --lr_decay_every 200 --weight_decay 0.0 \
        --gpu 0 --logdir debug [--data pairs.feather | --synthetic] [--epochs 199] [--init inverse_model.torch]

Kept from the reference: the flag names/defaults, Adam(lr, weight_decay) + StepLR(lr_decay_every, lr_decay) stepped once per
epoch (:190-199), the model checkpoint ``inverse_model_runs/<logdir>/model-<iteration>.pth`` = ``model.state_dict()``
every 100 minibatches (:115,134-136), cross-entropy loss and accuracy bookkeeping.  Different: the dataset — the
reference reads a Habitat-rendered ``.npy`` index with machine-specific paths (dataloaders/gibson.py:56-66); here the pairs
come from a feather file with ``before_image, after_image, inverse_actions`` columns (the Q-learning data frame) or from
the synthetic generator.  absl is not required (argparse with the same flag names)."""
import argparse
import os

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--bottleneck_size", type=int, default=3)
    ap.add_argument("--lr", type=float, default=0.0001)
    ap.add_argument("--lr_decay", type=float, default=0.1)
    ap.add_argument("--lr_decay_every", type=float, default=200)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--logdir", default="debug")
    ap.add_argument("--data", default=None, help="feather file with before_image, after_image, inverse_actions")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--epochs", type=int, default=199)
    ap.add_argument("--max_batches", type=int, default=0, help="stop after this many minibatches (0 = run all epochs)")
    ap.add_argument("--init", default=None, help="state_dict to start from (e.g. inverse_model.torch)")
    ap.add_argument("--dtype", default=None)
    return ap.parse_args()


def pair_batches(args, device):
    from video_dqn_amd import synth
    if args.synthetic or not args.data:
        n = 16 * args.batch_size
        while True:
            for i in range(n // args.batch_size):
                fb = synth.make_frames_uint8(900 + i, "be", args.batch_size, 1, structured=True)[:, 0]
                fa = synth.make_frames_uint8(900 + i, "ae", args.batch_size, 1, structured=True)[:, 0]
                act = synth.randint(900 + i, "act", (args.batch_size,), 3)
                yield torch.from_numpy(fb).to(device), torch.from_numpy(fa).to(device), torch.from_numpy(act).to(device)
            yield None  # epoch boundary
    else:
        import pandas as pd
        from PIL import Image
        from video_dqn_amd.dataset import resize_center_crop_u8
        df = pd.read_feather(args.data)
        rng = np.random.default_rng(0)
        while True:
            order = rng.permutation(len(df))
            for lo in range(0, len(df) - args.batch_size + 1, args.batch_size):
                rows = df.iloc[order[lo:lo + args.batch_size]]
                be = torch.from_numpy(np.stack([resize_center_crop_u8(Image.open(p)) for p in rows["before_image"]]))
                ae = torch.from_numpy(np.stack([resize_center_crop_u8(Image.open(p)) for p in rows["after_image"]]))
                yield be.to(device), ae.to(device), torch.from_numpy(rows["inverse_actions"].to_numpy().astype(np.int64)).to(device)
            yield None


def main():
    args = parse()
    if args.bottleneck_size != 3:
        raise SystemExit("only bottleneck_size 3 (the published model) is supported")
    torch.cuda.set_device(args.gpu)
    device = torch.device("cuda", args.gpu)
    from video_dqn_amd.inverse_model import InverseActionModel
    from video_dqn_amd.inverse_train import InverseTrainer
    model = InverseActionModel(dtype=args.dtype, device=device, max_batch=args.batch_size)
    if args.init:
        model.load_state_dict(torch.load(args.init, map_location="cpu"), strict=True)
    trainer = InverseTrainer(model, lr=args.lr, weight_decay=args.weight_decay, lr_decay=args.lr_decay, lr_decay_every=args.lr_decay_every)
    out_dir = os.path.join("inverse_model_runs", args.logdir)
    os.makedirs(out_dir, exist_ok=True)
    it = pair_batches(args, device)
    print_every, iteration, seen = 100, 0, 0
    train_loss, train_acc, batch_idx = 0.0, 0, 0
    for epoch in range(1, args.epochs + 1):
        print("Train Epoch: ", epoch)
        batch_idx = 0
        while True:
            batch = next(it)
            if batch is None:
                break
            be, ae, act = batch
            loss, y = trainer.step(be, ae, act)
            train_loss += loss.item()
            train_acc += (y.argmax(dim=1) == act).sum().item()
            seen += 1
            if batch_idx % print_every == 0 and batch_idx != 0:  # :115-136
                iteration += 1
                print("iter: ", iteration, ", train_loss: ", train_loss / print_every, ", train_acc: ",
                      train_acc / (print_every * args.batch_size + args.batch_size))
                torch.save(model.state_dict(), os.path.join(out_dir, "model-{:d}.pth".format(iteration)))
                train_loss, train_acc = 0.0, 0
            batch_idx += 1
            if args.max_batches and seen >= args.max_batches:
                torch.save(model.state_dict(), os.path.join(out_dir, "model-final.pth"))
                return
        trainer.end_epoch()  # scheduler.step() :199
    torch.save(model.state_dict(), os.path.join(out_dir, "model-final.pth"))


if __name__ == "__main__":
    main()
