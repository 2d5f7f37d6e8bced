"""C-ABI boundary: every symbol include/rdm_hip.h declares is exported and bound; argument errors
come back as status codes + messages (no aborts); the product never imports the oracle.  CPU only
(no compute calls)."""
import ctypes as C
import os
import re

import pytest

from conftest import GOLDEN, ROOT

HEADER = os.path.join(ROOT, "include", "rdm_hip.h")


@pytest.fixture(scope="module")
def L():
    from md_rdm_amd import _lib, build
    build.build(verbose=False)
    return _lib.lib()


def test_header_symbols_exported_and_bound(L):
    from md_rdm_amd import _lib
    hdr = open(HEADER).read()
    declared = set(re.findall(r"\b(rdm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    bound = set(_lib.exported_symbols())
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(L, name)


def test_bench_library_is_separate_from_the_product(L):
    """The measurement kernels live in librdm_bench.so (include/rdm_bench.h): declared == bound == exported THERE, and the product
    library exports none of them."""
    from md_rdm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rdm_bench.h")).read()
    declared = set(re.findall(r"\b(rdm_microbench_[a-z0-9_]+)\s*\(", hdr))
    assert declared and declared == set(_lib.bench_symbols())
    B = _lib.bench_lib()
    for name in declared:
        assert hasattr(B, name) and not hasattr(L, name), name
    assert "microbench" not in open(HEADER).read()


def test_every_entry_point_cites_the_reference():
    hdr = open(HEADER).read()
    assert hdr.count("RDM_Net.py:") >= 6 and hdr.count("computations.py:") >= 8
    assert "loss.py:8-59" in hdr and "utils.py:195-211" in hdr and "module.py:41" in hdr


def test_registry_matches_reference_state_dict(L):
    want = [l.split(" ", 1) for l in open(os.path.join(GOLDEN, "state_dict_keys.txt")).read().splitlines()]
    assert L.rdm_net_num_tensors() == len(want) == 968
    total = 0
    for i, (k, rest) in enumerate(want):
        assert L.rdm_net_tensor_name(i).decode() == k
        shape = eval(rest.rsplit(" ", 1)[0])
        n = 1
        for s in shape:
            n *= s
        assert L.rdm_net_tensor_numel(i) == n, k
        is_param = "running" not in k and "num_batches" not in k
        assert bool(L.rdm_net_tensor_is_param(i)) == is_param
        total += n if is_param else 0
    assert total == 90529721


def test_plan_geometry_and_flops(L):
    h = C.c_void_p()
    assert L.rdm_net_create(16, 228, 304, C.byref(h)) == 0
    oh, ow = C.c_int32(), C.c_int32()
    assert L.rdm_net_output_hw(h, C.byref(oh), C.byref(ow)) == 0 and (oh.value, ow.value) == (8, 10)
    assert 4 * 2**30 < L.rdm_net_workspace_bytes(h) < 16 * 2**30
    # 155.7 GFLOP/img forward, 466.5 GFLOP/img per train step at 228x304
    f, b = L.rdm_net_forward_flops(h) / 16 / 1e9, L.rdm_net_backward_flops(h) / 16 / 1e9
    assert abs(f - 155.7) < 0.3 and abs(f + b - 466.5) < 1.0
    L.rdm_net_destroy(h)
    assert L.rdm_net_create(1, 228, 228, C.byref(h)) == 0
    assert abs(L.rdm_net_forward_flops(h) / 1e9 - 118.4) < 0.3
    assert L.rdm_net_output_hw(h, C.byref(oh), C.byref(ow)) == 0 and (oh.value, ow.value) == (8, 8)
    L.rdm_net_destroy(h)


def test_errors_are_status_codes(L):
    h = C.c_void_p()
    assert L.rdm_net_create(0, 228, 228, C.byref(h)) == -1
    assert b"batch" in L.rdm_last_error_string()
    assert L.rdm_net_create(1, 8, 8, C.byref(h)) == -1
    assert L.rdm_dorn_fwd(None, None, None, 1, 90, 64, None) == -1
    assert L.rdm_als_rank1(None, 0, None, 1, 1, 100, 64, 30, None, 0, None) == -1
    from md_rdm_amd._lib import ConvDesc
    d = ConvDesc(1, 8, 8, 20, 20, 16, 16, 3, 3, 1, 1, 1, 1)
    assert L.rdm_conv2d_fwd(C.byref(d), C.c_void_p(256), C.c_void_p(256), None, None, None, C.c_void_p(256), None, None, None) == -1
    assert b"multiple of 16" in L.rdm_last_error_string()
    a, b = C.c_int32(), C.c_int32()
    assert L.rdm_net_segment_range(7, C.byref(a), C.byref(b)) == -1
    assert L.rdm_net_segment_range(0, C.byref(a), C.byref(b)) == 0 and L.rdm_net_tensor_name(b.value).decode() == "d_1.conv2.bias"


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "md_rdm_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "/root/reference" not in txt or f.endswith(".py") and "reference" in txt, f


def test_model_surface_on_cpu():
    """Construction, state_dict keys/shapes/dtypes and the loud failure without a GPU."""
    import torch
    from md_rdm_amd.network import RDM_Net
    from md_rdm_amd import _lib
    m = RDM_Net.DepthEstimationNet()
    sd = m.state_dict()
    want = [l.split(" ", 1) for l in open(os.path.join(GOLDEN, "state_dict_keys.txt")).read().splitlines()]
    assert list(sd.keys()) == [k for k, _ in want]
    for (k, rest), v in zip(want, sd.values()):
        assert rest == f"{tuple(v.shape)} {str(v.dtype).replace('torch.', '')}", k
    assert sum(p.numel() for p in m.parameters()) == 90529721
    assert RDM_Net.use_cuda is True and RDM_Net.freeze_encoder is False
    m.freeze_encoder()
    assert not any(p.requires_grad for p in m.encoder.parameters())
    with pytest.raises(_lib.RdmError):
        m(torch.zeros(1, 3, 228, 228))          # no CPU fallback


def test_lightning_checkpoint_wire_format(tmp_path):
    """model.<key> prefix of the reference's LightningModule (module.py:32, train.py:41-47), both directions."""
    import torch
    from md_rdm_amd import checkpoint, filler
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    a = DepthEstimationNet()
    filler.fill_state_dict(a.state_dict())
    ck = checkpoint.to_lightning(a)
    assert len(ck["state_dict"]) == 968 and all(k.startswith("model.") for k in ck["state_dict"])
    assert "model.encoder.dense_e2.denselayer1.norm1.weight" in ck["state_dict"]
    path = tmp_path / "last.ckpt"
    torch.save(ck, path)
    b = DepthEstimationNet()
    res, _ = checkpoint.from_lightning(b, str(path))
    assert not res.missing_keys and not res.unexpected_keys
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    checkpoint.from_lightning(b, a.state_dict())        # a bare state_dict is accepted too


def test_genuine_lightning_checkpoint_with_foreign_classes_loads_without_lightning(tmp_path):
    """A real Lightning 1.1.x .ckpt (train.py:41-47) pickles non-tensor classes: ``hyper_parameters`` is an AttributeDict
    (save_hyperparameters) and ``callbacks`` is keyed by the ModelCheckpoint CLASS.  torch >= 2.6's weights_only=True refuses them
    and pytorch_lightning is not installed here - the loader must still extract the 968 tensors, without importing anything."""
    import sys
    import types
    import torch
    from md_rdm_amd import checkpoint, filler
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    a = DepthEstimationNet()
    filler.fill_state_dict(a.state_dict())
    names = ["pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing", "pytorch_lightning.callbacks",
             "pytorch_lightning.callbacks.model_checkpoint"]
    mods = {n: types.ModuleType(n) for n in names}

    class AttributeDict(dict):
        pass

    class ModelCheckpoint:
        pass
    AttributeDict.__module__, AttributeDict.__qualname__ = "pytorch_lightning.utilities.parsing", "AttributeDict"
    ModelCheckpoint.__module__, ModelCheckpoint.__qualname__ = "pytorch_lightning.callbacks.model_checkpoint", "ModelCheckpoint"
    mods["pytorch_lightning.utilities.parsing"].AttributeDict = AttributeDict
    mods["pytorch_lightning.callbacks.model_checkpoint"].ModelCheckpoint = ModelCheckpoint
    sys.modules.update(mods)
    try:
        ck = checkpoint.to_lightning(a, {"epoch": 3, "global_step": 1234})
        ck["hyper_parameters"] = AttributeDict(batch_size=4, learning_rate=1e-4, metrics=["delta1"])
        ck["callbacks"] = {ModelCheckpoint: {"best_model_score": torch.tensor(0.5), "best_model_path": "x.ckpt"}}
        ck["optimizer_states"] = [{"state": {0: {"step": 5, "exp_avg": torch.zeros(3), "exp_avg_sq": torch.zeros(3)}}, "param_groups": [{"lr": 1e-4, "params": [0]}]}]
        ck["lr_schedulers"] = [{"best": 0.5, "num_bad_epochs": 1}]
        path = tmp_path / "epoch=3-val_delta1=0.5.ckpt"
        torch.save(ck, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)           # what the old loader did
    b = DepthEstimationNet()
    res, loaded = checkpoint.from_lightning(b, str(path))
    assert not res.missing_keys and not res.unexpected_keys
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert loaded["epoch"] == 3 and loaded["global_step"] == 1234
    assert "pytorch_lightning" not in sys.modules                          # nothing foreign was imported or executed

    class FakeOpt:
        def load_state_dict(self, sd):
            raise AssertionError("a reference (torch.optim) optimizer state must not be pushed into the fused optimiser")
    assert checkpoint.restore_training_state(loaded, FakeOpt()) is False    # weights-only resume for reference checkpoints


def test_hostile_checkpoint_pickle_executes_nothing(tmp_path):
    """`--resume` on a crafted file.  A pickle whose __reduce__ names builtins.eval / exec / getattr / __import__,
    os.system, subprocess or posix.system must have NO side effect through load_checkpoint_file: the stand-in unpickler resolves exact
    (module, name) pairs only, everything else is inert.  The fallback path is forced (the weights_only pass refuses such a file)."""
    import os
    import pickle
    import torch
    from md_rdm_amd import checkpoint
    marker = tmp_path / "pwned"
    code = f"open({str(marker)!r}, 'w').write('x')"

    class ViaEval:
        def __reduce__(self):
            return (eval, (code,))

    class ViaExec:
        def __reduce__(self):
            return (exec, (code,))

    class ViaSystem:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    class ViaImport:
        def __reduce__(self):
            return (__import__, ("os",))

    class ViaGetattr:
        def __reduce__(self):
            return (getattr, (ViaImport(), "system"))

    for i, evil in enumerate([ViaEval(), ViaExec(), ViaSystem(), ViaImport(), ViaGetattr()]):
        path = tmp_path / f"evil{i}.ckpt"
        payload = {"state_dict": {"model.weight_layer.d0": torch.ones(1, 1)}, "hyper_parameters": evil}
        torch.save(payload, path)                                            # zip format
        loaded = checkpoint.load_checkpoint_file(str(path))
        assert not marker.exists(), type(evil).__name__
        assert torch.equal(loaded["state_dict"]["model.weight_layer.d0"], torch.ones(1, 1))
        assert isinstance(loaded["hyper_parameters"], checkpoint._Inert)
        torch.save(payload, path, _use_new_zipfile_serialization=False)     # legacy format
        checkpoint.load_checkpoint_file(str(path))
        assert not marker.exists(), type(evil).__name__
        with open(path, "wb") as fh:                                         # a bare protocol-4 pickle, the reproduction in the advice
            pickle.dump({"x": evil}, fh, protocol=4)
        try:
            checkpoint.load_checkpoint_file(str(path))
        except Exception:
            pass                                                             # not a torch file: refusing it is fine, executing it is not
        assert not marker.exists(), type(evil).__name__
    allowed = checkpoint._allowed_globals()
    assert not any(m == "builtins" and n in ("eval", "exec", "getattr", "__import__", "compile", "open", "setattr") for m, n in allowed)
    assert not any(m.split(".")[0] in ("os", "posix", "subprocess", "sys", "importlib", "shutil") for m, n in allowed)


def test_reduce_lr_on_plateau_matches_torch():
    """module.py:42-46: ReduceLROnPlateau(optimizer, 'max', patience=2) on val_delta1 - same lr trajectory as torch's scheduler."""
    import torch
    from md_rdm_amd.train import ReduceLROnPlateau

    class Opt:
        lr = 1e-4

        class small:
            param_groups = [{"lr": 1e-4}]
    o = Opt()
    mine = ReduceLROnPlateau(o, "max", patience=2)
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=1e-4)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, "max", patience=2)
    seq = [0.30, 0.31, 0.31, 0.309, 0.31, 0.305, 0.4, 0.4, 0.39, 0.4, 0.40001, 0.2, 0.2, 0.2, 0.2, 0.2]
    for v in seq:
        mine.step(v)
        ref.step(v)
        assert abs(o.lr - topt.param_groups[0]["lr"]) <= 1e-18 + 1e-12 * o.lr, (v, o.lr, topt.param_groups[0]["lr"])
        assert o.small.param_groups[0]["lr"] == o.lr
    assert o.lr < 1e-5                                                      # it did decay (twice) on this sequence
    st = mine.state_dict()
    again = ReduceLROnPlateau(Opt(), "max", patience=2)
    again.load_state_dict(st)
    assert again.best == mine.best and again.bad == mine.bad
