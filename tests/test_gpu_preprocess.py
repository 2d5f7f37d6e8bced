"""GPU input pipeline (rdm_nyu_preprocess through md_rdm_amd.dataloaders) against the Pillow-pinned oracle: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import preprocess_cpu as P

pytestmark = pytest.mark.gpu


def _sample(rng, H, W, smooth):
    if smooth:
        yy, xx = np.mgrid[0:H, 0:W]
        base = 127 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
        rgb = np.clip(base[..., None] + rng.normal(0, 20, (H, W, 3)) + np.array([10, -20, 30]), 0, 255).astype(np.uint8)
        depth = (2.0 + 1.5 * np.sin(xx / 91.0 + yy / 57.0) + rng.random((H, W)) * 0.1).astype(np.float32)
    else:
        rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        depth = (rng.random((H, W)) * 9.5 + 0.5).astype(np.float32)
    depth[rng.random((H, W)) < 0.05] = 0.0
    return rgb, depth


def _run(raws, draws, resize, out):
    from md_rdm_amd.dataloaders import nyu
    H, W = raws[0][1].shape
    pre = nyu.NyuGpuPreprocessor(resize, out)
    rgb = torch.from_numpy(np.stack([r for r, _ in raws])).cuda()
    dep = torch.from_numpy(np.stack([d for _, d in raws])).cuda()
    params = [nyu.make_params(d["s"], d["angle"], d["flip"], d["jitter"], (H, W), resize, out) for d in draws]
    x, y = pre(rgb, dep, params)
    torch.cuda.synchronize()
    return x.cpu().numpy(), y.cpu().numpy()


@pytest.mark.parametrize("H,W,out", [(480, 640, (228, 304)), (375, 1242, (228, 304)), (300, 300, (228, 228))])
def test_training_preprocess_bit_exact(H, W, out):
    rng = np.random.default_rng(H + W)
    names = ["brightness", "contrast", "saturation"]
    raws = [_sample(rng, H, W, smooth=i % 2 == 0) for i in range(6)]
    draws = [dict(s=float(rng.uniform(1.0, 1.5)), angle=float(rng.uniform(-5, 5)), flip=bool(rng.uniform() > 0.5),
                  jitter=[(names[j], float(rng.uniform(0.6, 1.4))) for j in rng.permutation(3)]) for _ in raws]
    draws[0] = dict(s=1.0, angle=0.0, flip=False, jitter=[])                                   # identity
    draws[1] = dict(s=1.5, angle=-5.0, flip=True, jitter=[("contrast", 1.4), ("saturation", 0.6)])  # extremes, two ops only
    x, y = _run(raws, draws, 250, out)
    for i, ((rgb, depth), d) in enumerate(zip(raws, draws)):
        xr, yr = P.pil_training_preprocess(rgb, depth, output_size=out, **d)
        assert np.array_equal(x[i], xr), (i, d, int((x[i] != xr).sum()))
        assert np.array_equal(y[i], yr), (i, d, int((y[i] != yr).sum()))


def test_validation_preprocess_bit_exact_and_loader():
    from md_rdm_amd.dataloaders import nyu
    rng = np.random.default_rng(5)
    raws = [_sample(rng, 480, 640, smooth=True) for _ in range(5)]
    ds = nyu.NYUDataset(raws, split="val")
    batches = list(nyu.PrefetchLoader(ds, batch_size=2, drop_last=False))
    assert [b[0].shape[0] for b in batches] == [2, 2, 1]
    x = torch.cat([b[0] for b in batches]).cpu().numpy()
    y = torch.cat([b[1] for b in batches]).cpu().numpy()
    for i, (rgb, depth) in enumerate(raws):
        xr, yr = P.pil_validation_preprocess(rgb, depth)
        assert np.array_equal(x[i], xr) and np.array_equal(y[i], yr)


def test_test_preprocess_bit_exact_through_the_loader():
    """split="test": Resize(500) -> CenterCrop((480, 640)) -> Resize(output_size) (nyu_dataloader.py:289-307)."""
    from md_rdm_amd.dataloaders import nyu
    rng = np.random.default_rng(15)
    raws = [_sample(rng, 480, 640, smooth=i == 0) for i in range(3)]
    for out in [(228, 304), (226, 226)]:
        ds = nyu.NYUDataset(raws, split="test", output_size=out)
        x, y = next(iter(nyu.PrefetchLoader(ds, batch_size=3, drop_last=False)))
        for i, (rgb, depth) in enumerate(raws):
            xr, yr = P.pil_test_preprocess(rgb, depth, output_size=out)
            assert np.array_equal(x[i].cpu().numpy(), xr) and np.array_equal(y[i].cpu().numpy(), yr)


def test_training_loader_shards_and_reproduces_its_draws():
    from md_rdm_amd.dataloaders import nyu
    rng = np.random.default_rng(9)
    raws = [_sample(rng, 480, 640, smooth=False) for _ in range(8)]
    ds = nyu.NYUDataset(raws, split="train")
    seen = []
    for rank in range(2):
        ld = nyu.PrefetchLoader(ds, batch_size=2, seed=3, rank=rank, world=2)
        assert len(ld) == 2
        for xb, yb in ld:
            assert xb.shape == (2, 3, 228, 304) and yb.shape == (2, 1, 228, 304)
            assert torch.isfinite(xb).all() and float(xb.min()) >= 0.0 and float(xb.max()) <= 1.0
            seen.append(xb.shape[0])
    assert sum(seen) == 8
    # the loader's draws are the reference's distributions and are replayable from the seed
    g = np.random.default_rng(0)
    d, p = nyu.draw_training_params(g, (480, 640))
    assert 1.0 <= d["s"] <= 1.5 and -5.0 <= d["angle"] <= 5.0 and sorted(n for n, _ in d["jitter"]) == ["brightness", "contrast", "saturation"]
    assert p.h2 == int(250 * d["s"]) and p.top == int(round((p.h2 - 228) / 2.0))


def test_errors():
    from md_rdm_amd import _lib
    from md_rdm_amd.dataloaders import nyu
    with pytest.raises(ValueError):
        nyu.make_params(0.5, 0.0, False, [], (480, 640), 250, (228, 304))
    with pytest.raises(ValueError):
        nyu.make_params(1.0, 0.0, False, [], (480, 640), 100, (228, 304))           # crop larger than the resized image
    pre = nyu.NyuGpuPreprocessor(50, (40, 60))
    rgb = torch.zeros(1, 480, 640, 3, dtype=torch.uint8, device="cuda")
    dep = torch.zeros(1, 480, 640, device="cuda")
    with pytest.raises(_lib.RdmError, match="down-scale"):
        pre(rgb, dep, [nyu.make_params(1.0, 0.0, False, [], (480, 640), 50, (40, 60))])
    with pytest.raises(_lib.RdmError):
        nyu.NyuGpuPreprocessor()(rgb.cpu(), dep.cpu(), [])


def test_train_script_on_a_raw_dataset(tmp_path, capsys):
    """train.py end to end on raw .npz samples: loader -> GPU augmentation -> train step -> validation epoch mean."""
    from md_rdm_amd import train
    rng = np.random.default_rng(21)
    for split, n in (("train", 4), ("val", 2)):
        d = tmp_path / split / "scene"
        d.mkdir(parents=True)
        for i in range(n):
            rgb, depth = _sample(rng, 480, 640, smooth=True)
            np.savez(d / f"{i:05d}.npz", rgb=rgb, depth=depth + 0.5)
    ck = tmp_path / "ck"
    train.main(["--nyu_path", str(tmp_path), "--batch_size", "2", "--max_epochs", "2", "--seed", "1", "--checkpoint_dir", str(ck)])
    out = capsys.readouterr().out
    assert "epoch 0 step 1 loss" in out and "val_delta1" in out
    assert "nan" not in out.lower()
    saved = list(ck.glob("*.ckpt"))
    assert len(saved) == 1                                           # save_top_k = 1
    ckpt = torch.load(saved[0], map_location="cpu")
    state = ckpt["state_dict"]
    assert len(state) == 968 and all(k.startswith("model.") for k in state)
    # the checkpoint carries the training state too (Lightning's top-level keys): fused-optimiser moments / step / lr, plateau scheduler
    opt_state = ckpt["optimizer_states"][0]
    assert opt_state["step"] == 2 * (ckpt["epoch"] + 1) and opt_state["exp_avg"].numel() > 90_000_000 and float(opt_state["exp_avg"].abs().max()) > 0
    assert "best" in ckpt["lr_schedulers"][0]
    train.main(["--synthetic", "--batch_size", "2", "--seed", "1", "--max_steps", "1", "--max_epochs", str(ckpt["epoch"] + 2), "--resume", str(saved[0])])
    out = capsys.readouterr().out
    assert f"resumed optimiser state at step {opt_state['step']}" in out and f"epoch {ckpt['epoch'] + 1} step 0 loss" in out
    assert f"epoch {ckpt['epoch']} step" not in out                   # continues after the saved epoch, does not repeat it


def test_train_script_synthetic_dev_run(capsys):
    from md_rdm_amd import train
    train.main(["--synthetic", "--dev", "--batch_size", "2", "--seed", "1"])
    assert "epoch 0 step 0 loss" in capsys.readouterr().out


def test_train_script_flags_work_or_raise(capsys):
    """train.py:9-30,55,74-80: no flag is silently ignored."""
    from md_rdm_amd import train
    with pytest.raises(SystemExit) as e:                               # gpus=N needs N launched ranks
        train.main(["--synthetic", "--dev", "--gpus", "8"])
    assert "WORLD_SIZE" in str(e.value) and "torch.distributed.run" in str(e.value)
    with pytest.raises(SystemExit):
        train.main(["--synthetic", "--dev", "--min_epochs", "5", "--max_epochs", "2"])
    with pytest.raises(SystemExit):
        train.main(["--synthetic", "--dev", "--precision", "8"])
    # --detect_anomaly switches autograd anomaly mode on (train.py:28-30) and the native autograd node runs under it
    try:
        train.main(["--synthetic", "--dev", "--batch_size", "2", "--seed", "1", "--detect_anomaly"])
        assert torch.is_anomaly_enabled()
    finally:
        torch.autograd.set_detect_anomaly(False)
    assert "Enabling anomaly detection" in capsys.readouterr().out
    # --precision 16: training in the mixed-precision arithmetic mode (bf16 GEMM operands), the validation forward on the bf16 MFMA path
    train.main(["--synthetic", "--dev", "--batch_size", "2", "--seed", "1", "--precision", "16"])
    out = capsys.readouterr().out
    assert "epoch 0 step 0 loss" in out and "val_delta1" in out and "nan" not in out.lower()


def test_find_learning_rate_sweeps_and_restores(capsys):
    """train.py:74-80 / Lightning lr_find: exponential sweep 1e-8 -> 1, smoothed loss, early stop at 4x the best, suggestion at the
    steepest descent; the model and optimiser come back untouched."""
    from md_rdm_amd import filler, harness
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    dev = torch.device("cuda:0")
    m = DepthEstimationNet()
    filler.fill_state_dict(m.state_dict())
    m = m.to(dev).train()
    m.flatten_parameters()
    m.direct_grads = True
    opt = harness.FusedAdamW(m, lr=1e-4)
    x, y = filler.synthetic_batch(2, 226, 226, seed=4)
    batch = (torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sug, lrs, losses = harness.find_learning_rate(m, opt, [batch], num_training=40)
    assert 12 < len(lrs) <= 40 and abs(lrs[0] - 1e-8) < 1e-20 and all(b > a for a, b in zip(lrs, lrs[1:]))
    assert sug is not None and lrs[10] <= sug <= lrs[-1] and all(np.isfinite(losses[:-1]))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k                           # weights, BatchNorm buffers and counters restored
    assert opt.step_count == 0 and opt.lr == 1e-4 and float(opt.m.abs().max()) == 0
    from md_rdm_amd import train
    assert train.main(["--synthetic", "--overfit", "--batch_size", "2", "--seed", "1", "--find_learning_rate"]) is not None
    out = capsys.readouterr().out
    assert "Old learning rate:" in out and "Suggested learning rate:" in out and "epoch 0 step" not in out   # the finder runs INSTEAD of fit
