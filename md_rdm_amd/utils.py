"""The one function of the reference's ``utils.py`` that is on the hot path."""
import torch

from . import _lib


# Label of a NON-POSITIVE depth (log -> NaN -> `.int()`), which the reference leaves to the device it runs on: "cpu" = 0x80000000 (x86; what the
# parity fixtures - generated on the CPU - pin; the default), "cuda" = 0 (what the reference's own GPU training computes, utils.py:205-211 with
# cuda=True).  Either way such a pixel poisons compute_final_depth's geometric mean for its sample, exactly as in the reference (log of a
# negative label or of 0); harness.prepare_target's 1e-4 floor is applied BEFORE the resize to 8x8 and does not prevent the overshoot.
NAN_LABEL = "cpu"


def depth2label_sid(depth, K=90.0, alpha=0.02, beta=10.0, cuda=False):
    """utils.py:195-211 with K=90, alpha=0.02, beta=10 (the only values the path uses): SID label
    int(max(K*log(d/alpha)/log(beta/alpha), 0)), with the reference's float32 constants, one launch."""
    if NAN_LABEL not in ("cpu", "cuda"):
        raise _lib.RdmError("utils.NAN_LABEL must be 'cpu' or 'cuda'")
    if (K, alpha, beta) != (90.0, 0.02, 10.0):
        raise _lib.RdmError("depth2label_sid: only the reference defaults K=90, alpha=0.02, beta=10 are built")
    if not depth.is_cuda:
        raise _lib.RdmError("depth2label_sid runs on the GPU only")
    d = depth.double().contiguous()
    out = torch.empty(d.shape, dtype=torch.int32, device=d.device)
    _lib.check(_lib.lib().rdm_depth2label_sid_ex(_lib.ptr(d), _lib.ptr(out), d.numel(), 0 if NAN_LABEL == "cpu" else 1, _lib.stream()))
    return out
