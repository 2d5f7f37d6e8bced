"""Checkpoint wire format: the reference saves through Lightning's ModelCheckpoint
(train.py:41-47), whose ``state_dict`` holds the network under the attribute name of the LightningModule
(``self.model``, network/module.py:32) - i.e. every key carries a ``model.`` prefix.  The 968 keys underneath
are identical to ours, so trained weights interchange in both directions."""
import torch

PREFIX = "model."


def to_lightning(model, extra=None):
    """{'state_dict': {'model.<key>': tensor}, ...} as a Lightning 1.1.x .ckpt would hold it."""
    sd = {PREFIX + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ckpt = {"state_dict": sd, "pytorch-lightning_version": "1.1.7", "epoch": 0, "global_step": 0}
    ckpt.update(extra or {})
    return ckpt


def from_lightning(model, ckpt, strict=True):
    """Load a Lightning checkpoint (dict or path) or a bare state_dict into a DepthEstimationNet."""
    if isinstance(ckpt, (str, bytes)):
        ckpt = torch.load(ckpt, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt)
    if any(k.startswith(PREFIX) for k in sd):
        sd = {k[len(PREFIX):]: v for k, v in sd.items() if k.startswith(PREFIX)}
    return model.load_state_dict(sd, strict=strict)
