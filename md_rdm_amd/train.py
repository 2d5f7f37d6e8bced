"""Counterpart of the reference's ``train.py`` (flags :9-26) on the native stack.  Lightning is not
used: the step logic is md_rdm_amd/harness.py (restating network/module.py).  ``--nyu_path DIR`` reads
raw NYU samples (.h5 / .npz) through md_rdm_amd/dataloaders (module.py:19-27: NYUDataset(..., output_size=
(226, 226)), shuffle for train) with the PIL augmentation chain on the GPU; ``--synthetic`` feeds
hash-generated NYU-shaped batches.

  python -m md_rdm_amd.train --synthetic --batch_size 16 --max_steps 20
  python -m md_rdm_amd.train --nyu_path /data/nyudepthv2 --batch_size 16
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m md_rdm_amd.train --synthetic --gpus 8
"""
import os
import random
import sys
import time
from argparse import ArgumentParser

# multi-process GPU work on this ROCm stack (RCCL, --gpus N) needs dmabuf IPC - hipIpcGetMemHandle fails with "invalid argument" in the
# legacy mode; must be in the environment before the first HIP call of the process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch


class ReduceLROnPlateau:
    """module.py:42-46: ReduceLROnPlateau(optimizer, 'max', patience=2) monitoring val_delta1
    (torch defaults: factor 0.1, threshold 1e-4 rel, cooldown 0, min_lr 0)."""

    def __init__(self, optimizer, mode="max", patience=2, factor=0.1, threshold=1e-4):
        self.opt, self.mode, self.patience, self.factor, self.threshold = optimizer, mode, patience, factor, threshold
        self.best, self.bad = None, 0

    def step(self, metric):
        better = self.best is None or (metric > self.best * (1 + self.threshold) if self.mode == "max" else metric < self.best * (1 - self.threshold))
        if better:
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
            if self.bad > self.patience:
                self.opt.lr *= self.factor
                for g in self.opt.small.param_groups:
                    g["lr"] = self.opt.lr
                self.bad = 0

    def state_dict(self):
        return {"best": self.best, "bad": self.bad, "lr": self.opt.lr}

    def load_state_dict(self, sd):
        self.best, self.bad = sd["best"], sd["bad"]


def main(argv=None):
    parser = ArgumentParser("Trains mono depth estimation models (MI355X-native stack)")
    parser.add_argument("--seed", default=None, type=int)
    parser.add_argument("--precision", default=32, type=int, help="32 (default here - the parity configuration): float32 training and validation; 16 (the reference's default, AMP O2, train.py:11,57-58): training in the mixed-precision "
                        "arithmetic mode (model.gemm_bf16 = 3: bf16 operands with float32 accumulation in the GRADIENT GEMMs of the dense blocks; float32 forward, weights, statistics and optimiser) and the VALIDATION forward on the bf16 MFMA path")
    parser.add_argument("--gemm_bf16", type=int, default=None, choices=[0, 1, 2, 3], help="explicit mixed-precision arithmetic mode of the training step: 1 forward + gradient GEMMs, 2 forward only, 3 gradient GEMMs only (what --precision 16 selects)")
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--dev", action="store_true", help="one train + one val step (Lightning fast_dev_run)")
    parser.add_argument("--overfit", action="store_true", help="reuse one batch")
    parser.add_argument("--min_epochs", default=1, type=int)
    parser.add_argument("--max_epochs", default=1, type=int)
    parser.add_argument("--max_steps", default=8, type=int, help="steps per epoch in --synthetic mode")
    parser.add_argument("--metrics", default=["delta1", "delta2", "delta3", "mse", "mae", "log10", "rmse"], nargs="+")
    parser.add_argument("--worker", default=6, type=int)
    parser.add_argument("--find_learning_rate", action="store_true")
    parser.add_argument("--detect_anomaly", action="store_true")
    parser.add_argument("--learning_rate", type=float, default=1e-4)
    parser.add_argument("--batch_size", type=int, default=4)
    parser.add_argument("--nyu_path", type=str, default=None)
    parser.add_argument("--synthetic", action="store_true")
    parser.add_argument("--size", type=int, nargs=2, default=[226, 226], help="input HxW (module.py:19 feeds 226x226)")
    parser.add_argument("--checkpoint_dir", type=str, default=None, help="keep the best checkpoint by val_delta1 here (train.py:41-47: ModelCheckpoint(save_top_k=1, monitor='val_delta1', mode='max'))")
    parser.add_argument("--resume", type=str, default=None, help="Lightning .ckpt or state_dict to start from")
    parser.add_argument("--relative_decoders", type=int, nargs="*", default=[], help="subset of 6 7 8 9: the relative decoders the reference keeps commented out (RDM_Net.py:57-60)")
    args = parser.parse_args(argv)
    if args.precision not in (16, 32):
        raise SystemExit("--precision must be 16 or 32")
    if not args.synthetic and not args.nyu_path:
        raise SystemExit("give --nyu_path DIR (raw .h5 / .npz samples) or --synthetic")
    if args.detect_anomaly:                                  # train.py:28-30
        print("Enabling anomaly detection")
        torch.autograd.set_detect_anomaly(True)
    if args.min_epochs > args.max_epochs:
        raise SystemExit(f"--min_epochs {args.min_epochs} > --max_epochs {args.max_epochs}: min_epochs only holds back an early stop (Lightning), "
                         "and like the reference this trainer has none - it always runs max_epochs")
    if args.seed is None:
        args.seed = random.randrange(4294967295)
    torch.manual_seed(args.seed)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # the reference's `gpus=N` makes Lightning spawn N DDP ranks itself (train.py:55); here one process per GPU is started by the launcher
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: start one rank per GPU, e.g.\n  python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 -m md_rdm_amd.train --gpus {args.gpus} ...")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from . import filler, harness, parallel
    from .metrics import MetricLogger
    from .network.RDM_Net import DepthEstimationNet
    model = DepthEstimationNet(relative_decoders=tuple(args.relative_decoders)).to(dev)
    # --precision 16 (the reference's default, train.py:11,57-58: fp16 AMP O2): mixed-precision ARITHMETIC for the training step.  Mode 3 =
    # bf16 operands in the GRADIENT GEMMs only: logits identical to the f32 step, every gradient tensor's cosine >= 0.9999, loss after four
    # AdamW steps within 0.1 % (tests/test_gpu_mixed.py).  Mode 1 (forward GEMMs rounded as well: logits RMS 3 %, gradient cosine down to
    # 0.84 at the hash-filled initial point, no loss scaling or long-run convergence evidence) stays an explicit choice: --gemm_bf16 1.
    model.gemm_bf16 = (args.gemm_bf16 if args.gemm_bf16 is not None else 3) if args.precision == 16 else (args.gemm_bf16 or 0)
    # the reference trains on the GPU, where depth2label_sid of a non-positive depth (int(NaN)) is 0; the CPU semantics (0x80000000) are what
    # the fixtures pin and stay the library default - training follows the reference's device
    from . import utils as _u
    _u.NAN_LABEL = "cuda"
    resumed = None
    if args.resume:
        from .checkpoint import from_lightning
        _, resumed = from_lightning(model, args.resume)
    model.flatten_parameters()
    sync = parallel.attach(model)
    best_delta1, best_path = None, None
    opt = harness.FusedAdamW(model, lr=args.learning_rate)
    sched = ReduceLROnPlateau(opt, "max", patience=2)
    start_epoch = 0
    if resumed is not None:
        from .checkpoint import restore_training_state
        if restore_training_state(resumed, opt, sched):          # one of OUR checkpoints: moments, step, lr, plateau state, epoch
            start_epoch = int(resumed.get("epoch", -1)) + 1
            best_delta1 = resumed.get("best_val_delta1")
            if rank == 0:
                print(f"resumed optimiser state at step {opt.step_count}, lr {opt.lr:g}, continuing with epoch {start_epoch}", flush=True)
        elif rank == 0:
            print("resumed WEIGHTS only (the checkpoint carries no fused-optimiser state, e.g. a reference Lightning .ckpt)", flush=True)
    logger = MetricLogger(args.metrics if "delta1" in args.metrics else ["delta1"] + list(args.metrics))
    H, W = args.size
    steps = 1 if args.dev else args.max_steps
    train_loader = val_loader = None
    if args.nyu_path:
        from .dataloaders import NYUDataset, PrefetchLoader
        train_loader = PrefetchLoader(NYUDataset(args.nyu_path, split="train", output_size=(H, W)), args.batch_size, seed=args.seed, device=dev,
                                      rank=rank, world=world, workers=args.worker)
        val_loader = PrefetchLoader(NYUDataset(args.nyu_path, split="val", output_size=(H, W)), 1, device=dev, drop_last=False, rank=rank, world=world,
                                    workers=args.worker)
        steps = 1 if args.dev else len(train_loader)

    def train_batches(epoch):
        if train_loader is not None:
            first = None
            for it, (x, y) in enumerate(train_loader):
                if it >= steps:
                    break
                first = first if first is not None else (x, y)
                yield first if args.overfit else (x, y)
            return
        for it in range(steps):
            seed = 1234 + rank if args.overfit else 1234 + rank + 1000 * (epoch * steps + it)
            x, y = filler.synthetic_batch(args.batch_size, H, W, seed=seed)
            yield torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)

    if args.find_learning_rate:                                  # train.py:74-80: the finder runs INSTEAD of fit
        model.train()
        suggestion, lrs, losses = harness.find_learning_rate(model, opt, train_batches(0), sync=sync)
        if rank == 0:
            print("Old learning rate: ", args.learning_rate)
            print("Suggested learning rate: ", suggestion, f"({len(lrs)} steps swept, lr {lrs[0]:.1e} .. {lrs[-1]:.1e})")
        if world > 1:
            dist.destroy_process_group()
        return suggestion

    for epoch in range(start_epoch, args.max_epochs):
        model.train()
        model.set_precision("f32")
        t0 = time.time()
        for it, (x, y) in enumerate(train_batches(epoch)):
            opt.zero_grad()
            loss, parts = harness.training_step(model, x, y)
            loss.backward()
            opt.step(sync=sync)                                  # per bucket as its reduction lands (one process: the plain update)
            if rank == 0:
                print(f"epoch {epoch} step {it} loss {loss.item():.4f} MSE {parts['mse'].item():.4f} Ord_Loss {parts['ord_loss'].item():.4f} "
                      f"Fine_Detail {parts['fine_detail_loss'].item():.4f}", flush=True)
        torch.cuda.synchronize()
        model.eval()
        model.set_precision("bf16" if args.precision == 16 else "f32")
        with torch.no_grad():
            logger.reset()
            if val_loader is not None:
                for vi, (x, y) in enumerate(val_loader):
                    y_hat, y_n = harness.validation_step(model, x, y)
                    logger.log_val(y_hat, y_n)     # the reference compares the (log-domain) recombination with the normalised target as is (module.py:117)
                    if args.dev:
                        break
            else:
                x, y = filler.synthetic_batch(1, H, W, seed=99)
                y_hat, y_n = harness.validation_step(model, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
                logger.log_val(y_hat, y_n)
        d1 = logger.computer.avg("delta1")         # epoch mean of the per-step values, as Lightning's self.log aggregates val_delta1
        sched.step(d1)
        if args.checkpoint_dir and rank == 0 and (best_delta1 is None or d1 > best_delta1):     # save_top_k=1, mode='max'
            from .checkpoint import to_lightning
            os.makedirs(args.checkpoint_dir, exist_ok=True)
            path = os.path.join(args.checkpoint_dir, f"epoch={epoch}-val_delta1={d1:.4f}.ckpt")
            torch.save(to_lightning(model, {"epoch": epoch, "global_step": (epoch + 1) * steps, "best_val_delta1": d1}, optimizer=opt, scheduler=sched), path)
            if best_path and best_path != path and os.path.exists(best_path):
                os.remove(best_path)
            best_delta1, best_path = d1, path
        if rank == 0:
            print(f"epoch {epoch}: {steps * args.batch_size * world / (time.time() - t0):.1f} img/s, val_delta1 {d1:.4f}, lr {opt.lr:g}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
