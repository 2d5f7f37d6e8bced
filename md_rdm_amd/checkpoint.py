"""Checkpoint wire format: the reference saves through Lightning's ModelCheckpoint
(train.py:41-47), whose ``state_dict`` holds the network under the attribute name of the LightningModule
(``self.model``, network/module.py:32) - i.e. every key carries a ``model.`` prefix.  The 968 keys underneath
are identical to ours, so trained weights interchange in both directions.

A genuine Lightning 1.1.x ``.ckpt`` is a pickle that also carries non-tensor objects: ``hyper_parameters`` (an ``AttributeDict``
from ``save_hyperparameters``), ``callbacks`` keyed by the ``ModelCheckpoint`` CLASS, ``optimizer_states``, ``lr_schedulers``.
torch >= 2.6 refuses those under ``weights_only=True``; only ``state_dict`` is wanted here, so the file is read with an unpickler
that resolves every non-torch global to an inert stand-in instead of importing (or requiring) pytorch_lightning: nothing but
tensors and containers is ever instantiated from the file.  The unpickler's allow-list is a table of exact (module, name) pairs
(tensor / storage / dtype reconstructors, plain containers, numpy's array reconstructors): `builtins.eval`, `os.system` or any other
callable a crafted file names in a `__reduce__` resolves to the inert stand-in and has no effect (tests/test_boundary.py).

Training state written by ``md_rdm_amd.train`` (``optimizer_states``, ``lr_schedulers``, ``epoch``, ``global_step``) uses the same
top-level keys as Lightning; the optimiser entry is this stack's FusedAdamW state (flat moments), which Lightning could not
consume - resuming a reference run here restores the WEIGHTS, resuming one of our runs restores everything."""
import io
import pickle

import torch

PREFIX = "model."


def to_lightning(model, extra=None, optimizer=None, scheduler=None):
    """{'state_dict': {'model.<key>': tensor}, ...} as a Lightning 1.1.x .ckpt would hold it (+ our optimiser / scheduler state)."""
    sd = {PREFIX + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ckpt = {"state_dict": sd, "pytorch-lightning_version": "1.1.7", "epoch": 0, "global_step": 0}
    if optimizer is not None:
        ckpt["optimizer_states"] = [optimizer.state_dict()]
    if scheduler is not None:
        ckpt["lr_schedulers"] = [scheduler.state_dict()]
    ckpt.update(extra or {})
    return ckpt


class _Inert:
    """Stand-in for any class a foreign checkpoint references (AttributeDict, ModelCheckpoint, Namespace ...): swallows construction
    and state, never executes foreign code."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def __reduce_ex__(self, proto):
        return (_Inert, ())

    def __call__(self, *a, **k):                     # a stand-in that a file tries to CALL (a function it named) returns another stand-in
        return _Inert()

    def __getattr__(self, name):
        raise AttributeError(name)


def _allowed_globals():
    """EXACT (module, name) pairs the stand-in unpickler resolves to real objects: the tensor / storage / dtype reconstructors a
    state_dict needs, plain containers, and numpy's array / scalar reconstructors.  Whole modules are never allowed - `builtins`
    alone would hand a crafted file `eval`, `exec`, `getattr` and `__import__` through a `__reduce__`."""
    import collections
    allowed = {("collections", "OrderedDict"): collections.OrderedDict, ("collections", "Counter"): collections.Counter,
               ("collections", "defaultdict"): collections.defaultdict,
               ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "bytearray"): bytearray, ("builtins", "complex"): complex,
               ("builtins", "slice"): slice, ("builtins", "range"): range, ("builtins", "dict"): dict, ("builtins", "list"): list,
               ("builtins", "tuple"): tuple, ("builtins", "int"): int, ("builtins", "float"): float, ("builtins", "bool"): bool,
               ("builtins", "str"): str, ("builtins", "bytes"): bytes}
    import _codecs
    allowed[("_codecs", "encode")] = _codecs.encode                       # how protocol-2 pickles carry bytes (numpy payloads)
    for name in ("_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_tensor_v3", "_rebuild_parameter", "_rebuild_parameter_with_state",
                 "_rebuild_device_tensor_from_numpy", "_rebuild_device_tensor_from_cpu_tensor", "_rebuild_meta_tensor_no_storage"):
        if hasattr(torch._utils, name):
            allowed[("torch._utils", name)] = getattr(torch._utils, name)
    allowed[("torch", "Size")] = torch.Size
    allowed[("torch", "Tensor")] = torch.Tensor
    allowed[("torch", "device")] = torch.device
    allowed[("torch.nn.parameter", "Parameter")] = torch.nn.parameter.Parameter
    allowed[("torch.serialization", "_get_layout")] = torch.serialization._get_layout
    for name in dir(torch):
        obj = getattr(torch, name)
        if isinstance(obj, torch.dtype) or (name.endswith("Storage") and isinstance(obj, type)):
            allowed[("torch", name)] = obj
    for name in dir(torch.cuda):                                            # checkpoints saved from the GPU name torch.cuda.FloatStorage
        if name.endswith("Storage") and isinstance(getattr(torch.cuda, name), type):
            allowed[("torch.cuda", name)] = getattr(torch.cuda, name)
    allowed[("torch.storage", "UntypedStorage")] = torch.storage.UntypedStorage
    allowed[("torch.storage", "TypedStorage")] = torch.storage.TypedStorage
    try:
        import numpy as np
        core = np._core if hasattr(np, "_core") else np.core
        for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):    # numpy 1.x and 2.x spell the module differently
            allowed[(mod, "_reconstruct")] = core.multiarray._reconstruct
            allowed[(mod, "scalar")] = core.multiarray.scalar
        allowed[("numpy", "ndarray")] = np.ndarray
        allowed[("numpy", "dtype")] = np.dtype
        for name in ("float16", "float32", "float64", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "bool_"):
            allowed[("numpy", name)] = getattr(np, name)
    except (ImportError, AttributeError):            # a numpy without this layout of np._core / np.core: skip the numpy entries, do not abort the load
        pass
    return allowed


_ALLOWED_GLOBALS = None          # built once per process (torch.load wraps the Unpickler in a fresh subclass per call: a cache on type(self) is thrown away)


class _TensorOnlyUnpickler(pickle.Unpickler):
    """Resolves ONLY the exact globals of _allowed_globals(); every other global a file names (pytorch_lightning classes, argparse
    Namespaces - or builtins.eval, os.system, anything a hostile file puts into a __reduce__) becomes the inert stand-in, whose
    construction and state restoration do nothing."""
    def find_class(self, module, name):
        global _ALLOWED_GLOBALS
        if _ALLOWED_GLOBALS is None:
            _ALLOWED_GLOBALS = _allowed_globals()
        return _ALLOWED_GLOBALS.get((module, name), _Inert)


class _TensorOnlyPickle:
    """pickle_module for torch.load: real torch / container types, inert stand-ins for everything else."""
    __name__ = "tensor_only_pickle"
    Unpickler = _TensorOnlyUnpickler
    load = staticmethod(lambda f, **k: _TensorOnlyUnpickler(f, **k).load())
    loads = staticmethod(lambda b, **k: _TensorOnlyUnpickler(io.BytesIO(b), **k).load())
    dump, dumps, Pickler = pickle.dump, pickle.dumps, pickle.Pickler


def load_checkpoint_file(path):
    """The checkpoint dict of ``path``: first the safe tensor-only loader, then (foreign classes present) the stand-in unpickler."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError, ModuleNotFoundError):
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TensorOnlyPickle)


def from_lightning(model, ckpt, strict=True, optimizer=None, scheduler=None):
    """Load a Lightning checkpoint (dict or path) or a bare state_dict into a DepthEstimationNet; returns (load result, ckpt dict).
    ``optimizer`` / ``scheduler``: restored too when the checkpoint carries OUR training state (see module docstring)."""
    if isinstance(ckpt, (str, bytes)):
        ckpt = load_checkpoint_file(ckpt)
    sd = ckpt.get("state_dict", ckpt)
    if any(k.startswith(PREFIX) for k in sd):
        sd = {k[len(PREFIX):]: v for k, v in sd.items() if k.startswith(PREFIX)}
    res = model.load_state_dict(sd, strict=strict)
    return res, ckpt


def restore_training_state(ckpt, optimizer, scheduler=None):
    """Optimiser moments / step / lr and the plateau scheduler from a checkpoint written by md_rdm_amd.train; False when the file
    holds none (a reference checkpoint: weights only)."""
    st = ckpt.get("optimizer_states") if isinstance(ckpt, dict) else None
    if not st or not isinstance(st[0], dict) or "exp_avg" not in st[0]:
        return False
    optimizer.load_state_dict(st[0])
    ls = ckpt.get("lr_schedulers")
    if scheduler is not None and ls and isinstance(ls[0], dict) and "best" in ls[0]:
        scheduler.load_state_dict(ls[0])
    return True
