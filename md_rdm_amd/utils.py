"""The one function of the reference's ``utils.py`` that is on the hot path."""
import torch

from . import _lib


def depth2label_sid(depth, K=90.0, alpha=0.02, beta=10.0, cuda=False):
    """utils.py:195-211 with K=90, alpha=0.02, beta=10 (the only values the path uses): SID label
    int(max(K*log(d/alpha)/log(beta/alpha), 0)), with the reference's float32 constants, one launch."""
    if (K, alpha, beta) != (90.0, 0.02, 10.0):
        raise _lib.RdmError("depth2label_sid: only the reference defaults K=90, alpha=0.02, beta=10 are built")
    if not depth.is_cuda:
        raise _lib.RdmError("depth2label_sid runs on the GPU only")
    d = depth.double().contiguous()
    out = torch.empty(d.shape, dtype=torch.int32, device=d.device)
    _lib.check(_lib.lib().rdm_depth2label_sid(_lib.ptr(d), _lib.ptr(out), d.numel(), _lib.stream()))
    return out
