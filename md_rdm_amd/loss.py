"""``loss.Ordinal_Loss`` of the reference (loss.py:8-59) as two single-launch kernels."""
import torch

from . import _lib


class _OrdinalLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ord_labels, target):
        N, C, H, W = ord_labels.shape
        P = ord_labels.contiguous()
        T = target.to(torch.int32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=P.device)
        _lib.check(_lib.lib().rdm_ordinal_loss_fwd(_lib.ptr(P), _lib.ptr(T), _lib.ptr(loss), N, C, H * W, _lib.stream()))
        ctx.save_for_backward(P, T)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        P, T = ctx.saved_tensors
        N, C, H, W = P.shape
        dP = torch.empty_like(P)
        gs = g.reshape(1).float().contiguous()
        _lib.check(_lib.lib().rdm_ordinal_loss_bwd(_lib.ptr(P), _lib.ptr(T), _lib.ptr(gs), _lib.ptr(dP), N, C, H * W, _lib.stream()))
        return dP, None


class Ordinal_Loss:
    """-(sum_{k<=t} log P_k + sum_{k>t} log(1-P_k)) / (N*H*W), clamps 1e-8..1e8, float32 logs."""

    def __init__(self):
        self.loss = 0.0

    def calc(self, ord_labels, target, cuda=True):
        if not ord_labels.is_cuda:
            raise _lib.RdmError("Ordinal_Loss runs on the GPU only")
        self.loss = _OrdinalLossFn.apply(ord_labels.double(), target)
        return self.loss
