"""The numpy restatement of Pillow's resample / rotate / blend arithmetic (oracle/preprocess_cpu.py) against Pillow itself -
the library the reference's torchvision transforms call into (nyu_dataloader.py:240-287).  Bit-exact."""
import numpy as np
import pytest

from oracle import preprocess_cpu as P

PIL = pytest.importorskip("PIL")


def _sample(rng, H, W, smooth):
    if smooth:
        yy, xx = np.mgrid[0:H, 0:W]
        base = 127 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
        rgb = np.clip(base[..., None] + rng.normal(0, 20, (H, W, 3)) + np.array([10, -20, 30]), 0, 255).astype(np.uint8)
        depth = (2.0 + 1.5 * np.sin(xx / 91.0 + yy / 57.0) + rng.random((H, W)) * 0.1).astype(np.float32)
    else:
        rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        depth = (rng.random((H, W)) * 9.5 + 0.5).astype(np.float32)
    depth[rng.random((H, W)) < 0.05] = 0.0          # invalid pixels (module.py:76-78)
    return rgb, depth


def _params(rng):
    names = ["brightness", "contrast", "saturation"]
    return dict(s=float(rng.uniform(1.0, 1.5)), angle=float(rng.uniform(-5, 5)), flip=bool(rng.uniform() > 0.5),
                jitter=[(names[i], float(rng.uniform(0.6, 1.4))) for i in rng.permutation(3)])


@pytest.mark.parametrize("seed", range(6))
def test_training_preprocess_matches_pillow(seed):
    rng = np.random.default_rng(seed)
    rgb, depth = _sample(rng, 480, 640, smooth=seed % 2 == 0)
    p = _params(rng)
    x1, y1 = P.pil_training_preprocess(rgb, depth, **p)
    x2, y2 = P.np_training_preprocess(rgb, depth, **p)
    assert x1.shape == (3, 228, 304) and y1.shape == (1, 228, 304)
    assert np.array_equal(x1, x2)
    assert np.array_equal(y1, y2)


@pytest.mark.parametrize("case", [dict(s=1.0, angle=0.0, flip=False, jitter=[]), dict(s=1.5, angle=-5.0, flip=True, jitter=[("contrast", 1.4)]),
                                  dict(s=1.0, angle=5.0, flip=True, jitter=[("saturation", 0.6), ("brightness", 1.4), ("contrast", 0.6)]),
                                  dict(s=1.25, angle=360.0, flip=False, jitter=[("brightness", 1.0)])])
def test_training_preprocess_edge_parameters(case):
    rgb, depth = _sample(np.random.default_rng(7), 480, 640, smooth=True)
    x1, y1 = P.pil_training_preprocess(rgb, depth, **case)
    x2, y2 = P.np_training_preprocess(rgb, depth, **case)
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2)


def test_validation_preprocess_and_other_raw_sizes():
    rng = np.random.default_rng(3)
    for (H, W, out) in [(480, 640, (228, 304)), (375, 1242, (228, 304)), (300, 300, (228, 228))]:
        rgb, depth = _sample(rng, H, W, smooth=False)
        x1, y1 = P.pil_validation_preprocess(rgb, depth, output_size=out)
        x2, y2 = P.np_validation_preprocess(rgb, depth, output_size=out)
        assert np.array_equal(x1, x2) and np.array_equal(y1, y2)
        p = _params(rng)
        x1, y1 = P.pil_training_preprocess(rgb, depth, output_size=out, **p)
        x2, y2 = P.np_training_preprocess(rgb, depth, output_size=out, **p)
        assert np.array_equal(x1, x2) and np.array_equal(y1, y2)


def test_blend_extrapolation_saturates_like_pillow():
    from PIL import Image, ImageEnhance
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (64, 80, 3)).astype(np.uint8)
    for name, enh in (("brightness", ImageEnhance.Brightness), ("contrast", ImageEnhance.Contrast), ("saturation", ImageEnhance.Color)):
        for f in (0.0, 0.6, 1.0, 1.4, 2.5):
            ref = np.asarray(enh(Image.fromarray(rgb, "RGB")).enhance(f))
            assert np.array_equal(ref, P.np_color_jitter(rgb, [(name, f)])), (name, f)


def test_host_side_parameters_match_the_oracle_conventions():
    """md_rdm_amd.dataloaders.nyu.make_params (host logic of the GPU pipeline, importable without a GPU): Resize sizes,
    centre-crop origin and the 16.16 fixed-point rotation coefficients are the ones the Pillow-pinned oracle derives."""
    from md_rdm_amd.dataloaders import nyu
    rng = np.random.default_rng(4)
    for (H, W, out) in [(480, 640, (228, 304)), (375, 1242, (228, 304)), (300, 300, (228, 228)), (640, 480, (228, 228))]:
        for _ in range(8):
            s, ang = float(rng.uniform(1.0, 1.5)), float(rng.uniform(-5, 5))
            p = nyu.make_params(s, ang, True, [("contrast", 0.9)], (H, W), 250, out)
            h1, w1 = P.resized_hw(H, W, 250)
            assert (h1, w1) == nyu.resized_hw(H, W, 250)
            h2, w2 = P.resized_hw(h1, w1, int(250 * s))
            assert (p.h2, p.w2) == (h2, w2)
            assert (p.top, p.left) == P.center_crop_box(h2, w2, *out)
            assert list(p.rot) == list(P.affine_fixed_coeffs(P.rotate_matrix(ang, w1, h1)))
            assert p.flip == 1 and list(p.op) == [1, -1, -1] and abs(p.depth_div - s) < 1e-6
    ident = nyu.identity_params((480, 640), 250, (228, 304))
    assert list(ident.rot) == [65536, 0, 32768, 0, 65536, 32768] and (ident.h2, ident.w2) == (250, 333) and list(ident.op) == [-1, -1, -1]
    d, p = nyu.draw_training_params(np.random.default_rng(0), (480, 640))
    assert 1.0 <= d["s"] <= 1.5 and -5.0 <= d["angle"] <= 5.0 and sorted(n for n, _ in d["jitter"]) == ["brightness", "contrast", "saturation"]
    with pytest.raises(ValueError):
        nyu.make_params(0.9, 0.0, False, [], (480, 640), 250, (228, 304))
    raws = [(np.zeros((480, 640, 3), np.uint8), np.ones((480, 640), np.float32))] * 3
    ds = nyu.NYUDataset(raws, split="val")
    assert len(ds) == 3 and ds.get_raw(1)[0].shape == (480, 640, 3)
    with pytest.raises(RuntimeError):
        nyu.NYUDataset(raws, split="bogus")


def test_test_preprocess_matches_pillow():
    """nyu_dataloader.py:289-307: Resize(500) -> CenterCrop((480, 640)) -> Resize((h, w))."""
    rng = np.random.default_rng(12)
    for out in [(228, 304), (226, 226)]:
        rgb, depth = _sample(rng, 480, 640, smooth=out[0] == 228)
        x1, y1 = P.pil_test_preprocess(rgb, depth, output_size=out)
        x2, y2 = P.np_test_preprocess(rgb, depth, output_size=out)
        assert x1.shape == (3,) + out and np.array_equal(x1, x2) and np.array_equal(y1, y2)
