"""The training/validation step of the reference's Lightning module (network/module.py:49-149),
restated without Lightning so that it runs on this stack: same target preparation, same loss sum,
AdamW(lr) as in ``configure_optimizers`` (:38-47).  ``train.py``'s flags are mirrored in
md_rdm_amd/train.py."""
import ctypes as C

import torch

from . import loss as l
from . import utils as u
from .network import computations as cp
from .network.RDM_Net import DepthEstimationNet


def normalize(batch):
    """module.py:145-149: divide by the geometric mean, exponent 1/H**2 (quick_gm squares its arg)."""
    B, C, H, W = batch.size()
    return cp.gm_normalize(batch, 1.0 / (H * H))


def prepare_target(y):
    """module.py:68,75-78: bicubic to 128x128, invalid pixels -> 1e-4."""
    y = cp.resize(y, 128)
    mask1 = y > 0
    mask2 = (y <= 0) + 1e-4
    return (y * mask1) + mask2


def compute_final_depth(fine_detail_list, target, has_ordinal):
    """module.py:119-133."""
    component_target = cp.decompose_depth_map([], normalize(target), 7)[::-1]
    if has_ordinal:
        ord_components = cp.decompose_depth_map([], normalize(u.depth2label_sid(cp.resize(target, 8), cuda=True)), 3)[::-1]
        component_target[0] = ord_components[0]
    components, loss = cp.optimize_components(fine_detail_list, component_target, True)
    final = cp.recombination(components)
    return final, loss


def compute_ordinal_target(ord_pred, target):
    """module.py:135-143; for non-square head outputs (where the reference raises) the target is
    resized to the head's (h, w)."""
    h, w = ord_pred.shape[2], ord_pred.shape[3]
    target = cp.resize(target, h if h == w else (h, w))
    return u.depth2label_sid(target, cuda=True)


def training_step(model, x, y):
    """module.py:64-97 without logging.  Returns (loss_all, dict of components)."""
    y = prepare_target(y)
    fine_details, ord_depth_pred, ord_label_pred = model(x)
    has_ordinal = fine_details[0].shape[2] == 1
    y_hat = list(fine_details)                      # recombination pops from the list it is given (computations.py:405-407)
    final_depth, fine_detail_loss = compute_final_depth(fine_details, y, has_ordinal=has_ordinal)
    ord_y = compute_ordinal_target(ord_depth_pred, y)
    ord_loss = l.Ordinal_Loss().calc(ord_label_pred, ord_y, cuda=True)
    mse = torch.nn.functional.mse_loss(final_depth, y)
    loss_all = mse + fine_detail_loss + ord_loss
    return loss_all, dict(mse=mse, fine_detail_loss=fine_detail_loss, ord_loss=ord_loss, final_depth=final_depth, y=y, ord_y=ord_y,
                          fine_details=y_hat, ord_depth_pred=ord_depth_pred, ord_label_pred=ord_label_pred)


def validation_step(model, x, y):
    """module.py:99-117: returns (y_hat, normalized target)."""
    y = prepare_target(y)
    fine_details, _, _ = model(x)
    has_ordinal = fine_details[0].shape[2] == 1
    y_hat, _ = compute_final_depth(fine_details, y, has_ordinal=has_ordinal)
    return y_hat, normalize(y)


class FusedAdamW:
    """torch.optim.AdamW(lr) semantics (module.py:41) as one launch per contiguous TRAINABLE range of the model's flat parameter /
    gradient buffers (2 for the reference's live graph) + a tiny torch step for the 4 Weights scalars."""

    def __init__(self, model: DepthEstimationNet, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.model, self.lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = self.v = None
        self.sync, self.mode = None, "after the last stage"        # data parallel: set by step(sync=...)
        self.small = torch.optim.AdamW([p for p in model.weight_layer.parameters() if p.requires_grad], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def zero_grad(self):
        self.small.zero_grad(set_to_none=True)

    def trainable_ranges(self):
        """Maximal contiguous [start, stop) element ranges of the flat buffer that hold tensors which receive a gradient.
        torch.optim.AdamW (the reference's optimiser) skips every parameter whose ``.grad is None``: ``d_1.conv1.*`` (constructed
        for every decoder, used only for id > 5, RDM_Net.py:146,156-157) and whatever ``freeze_encoder()`` froze (:65-67) must
        stay bit-unchanged - no weight decay, no moments.  Alignment padding between two trainable tensors is swept along
        (parameter, gradient and moments are all zero there and stay zero)."""
        _, _, entries = self.model._flat
        key = tuple(p.requires_grad for _, p, _, _, _ in entries)
        if getattr(self, "_ranges_key", None) != key:
            ranges = []
            for k, p, o, n, _ in entries:
                if not p.requires_grad or k.startswith("d_1.conv1."):
                    continue
                if ranges and ranges[-1][2] == o:                 # previous trainable tensor (padded to its 64-float slot) ends here
                    ranges[-1][1], ranges[-1][2] = o + n, o + (n + 63) // 64 * 64
                else:
                    ranges.append([o, o + n, o + (n + 63) // 64 * 64])
            self._ranges, self._ranges_key = [(a, b) for a, b, _ in ranges], key
        return self._ranges

    def _update(self, a, b, grad_scale):
        """AdamW on the trainable parts of the flat element range [a, b) (one launch per contiguous trainable piece)."""
        from . import _lib
        flat, gflat, _ = self.model._flat
        L, st = _lib.lib(), _lib.stream()
        off = C.c_void_p
        for ta, tb in self.trainable_ranges():
            lo, hi = max(a, ta), min(b, tb)
            if lo < hi:
                _lib.check(L.rdm_adamw_fused(off(flat.data_ptr() + 4 * lo), off(gflat.data_ptr() + 4 * lo), off(self.m.data_ptr() + 4 * lo),
                                             off(self.v.data_ptr() + 4 * lo), hi - lo, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                             self.step_count, float(grad_scale), st))

    def step(self, grad_scale=1.0, sync=None):
        """``sync`` = the GradSync of a data-parallel run: the update then runs PER BUCKET as each stage's reduction lands
        (``GradSync.finish(on_reduced=...)``: stage k's wait is a stream dependency and its AdamW launch follows at once, under the
        reductions still in flight; with the "reduce_scatter" exchange on this rank's shard only, the updated parameters are
        all-gathered).  Without it: the monolithic update of the whole buffer with the given ``grad_scale`` (what one process runs:
        2 launches for the reference's live graph, around d_1.conv1)."""
        flat, gflat, _ = self.model._flat
        if self.m is None or self.m.data_ptr() == 0 or self.m.numel() != flat.numel():
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)
        self.step_count += 1
        if sync is not None and sync.world > 1:
            self.sync, self.mode = sync, "per stage, as its reduction lands"
            scale = 1.0 / sync.world
            covered = []

            def on_reduced(stage, ranges):
                for a, b in ranges:
                    self._update(a, b, scale)
                covered.append(sync.slices[stage])
            sync.finish(on_reduced=on_reduced)
            # parameters outside every stage bucket (none for the reference's graph: the stages tile the stack) would be stale replicas
            assert sum(b - a for a, b in covered) >= sum(b - a for a, b in self.trainable_ranges()), "backward stages do not cover the trainable parameters"
        else:
            for a, b in self.trainable_ranges():                  # 2 launches for the reference's live graph (around d_1.conv1)
                self._update(a, b, grad_scale)
        self.model.mark_weights_changed()                         # derived bf16 copies are stale now
        self.small.step()

    def state_dict(self):
        """Moments, step and lr (the ``optimizer_states`` entry of a checkpoint)."""
        if self.sync is not None and self.m is not None:          # "reduce_scatter": each rank holds the moments of its shards only
            self.sync.allgather_shards(self.m)
            self.sync.allgather_shards(self.v)
        return {"step": self.step_count, "lr": self.lr, "exp_avg": None if self.m is None else self.m.detach().cpu().clone(),
                "exp_avg_sq": None if self.v is None else self.v.detach().cpu().clone(), "small": self.small.state_dict()}

    def load_state_dict(self, sd):
        flat = self.model._flat[0]
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        if sd.get("exp_avg") is not None:
            if sd["exp_avg"].numel() != flat.numel():
                raise ValueError("optimizer state does not match the model's flat parameter buffer")
            self.m, self.v = sd["exp_avg"].to(flat.device).clone(), sd["exp_avg_sq"].to(flat.device).clone()
        self.small.load_state_dict(sd["small"])
        for g in self.small.param_groups:
            g["lr"] = self.lr


def find_learning_rate(model, opt, batches, min_lr=1e-8, max_lr=1.0, num_training=100, early_stop_threshold=4.0, beta=0.98, sync=None):
    """The learning-rate range test the reference runs for ``--find_learning_rate`` (train.py:74-80: ``trainer.tuner.lr_find(module)``
    + ``lr_finder.suggestion()``; pytorch_lightning 1.1.7 defaults, third party): ``num_training`` steps with the learning rate swept
    exponentially from ``min_lr`` to ``max_lr``, an exponentially smoothed (beta 0.98, bias-corrected) loss per step, early stop once
    it exceeds ``early_stop_threshold`` x the best; the suggestion is the rate at the steepest descent of the smoothed loss
    (``numpy.gradient(...).argmin()``, skipping the first 10 and the last point).  Weights, BatchNorm buffers and optimiser state are
    restored afterwards, as Lightning does.  Returns (suggested_lr or None, lrs, smoothed losses)."""
    import numpy as np
    flat = model._flat[0]
    saved = (flat.detach().clone(), {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("weight_layer.") or "running_" in k or "num_batches" in k},
             opt.state_dict())
    lrs, losses, avg, best = [], [], 0.0, None
    seen, it = [], iter(batches)
    for i in range(num_training):
        try:
            x, y = next(it)
            if len(seen) < 64:
                seen.append((x, y))                                  # a one-shot generator is cycled from what it yielded
        except StopIteration:
            if not seen:
                raise ValueError("find_learning_rate: no batches")
            x, y = seen[i % len(seen)]
        lr = min_lr * (max_lr / min_lr) ** (i / max(num_training - 1, 1))
        opt.lr = lr
        for g in opt.small.param_groups:
            g["lr"] = lr
        opt.zero_grad()
        loss, _ = training_step(model, x, y)
        loss.backward()
        if sync is not None:
            opt.step(sync=sync)
        else:
            opt.step()
        cur = float(loss.item())
        if sync is not None:
            # data parallel: the early stop below must be a COLLECTIVE decision.  Each rank's shard loss diverges at a different
            # step near lr -> 1; a rank that left the sweep alone would leave the others blocked in the next stage all-reduce.
            # Every rank smooths the rank-mean loss, so all of them run the same number of steps and suggest the same rate.
            cur = sync.mean_scalar(cur)
        avg = beta * avg + (1 - beta) * cur
        smooth = avg / (1 - beta ** (i + 1))
        lrs.append(lr)
        losses.append(smooth)
        if not np.isfinite(smooth) or (i > 0 and best is not None and smooth > early_stop_threshold * best):
            break
        if best is None or smooth < best:
            best = smooth
    with torch.no_grad():                                           # restore: the sweep must not leave a trace
        flat.copy_(saved[0])
        sd = model.state_dict()
        for k, v in saved[1].items():
            sd[k].copy_(v)
    opt.load_state_dict(saved[2])
    if opt.m is not None and saved[2]["exp_avg"] is None:
        opt.m.zero_()
        opt.v.zero_()
    model.mark_weights_changed()
    skip_begin, skip_end = 10, 1
    suggestion = None
    if len(losses) > skip_begin + skip_end + 1:
        window = np.array(losses[skip_begin:-skip_end])
        suggestion = lrs[skip_begin + int(np.gradient(window).argmin())]
    return suggestion, lrs, losses
