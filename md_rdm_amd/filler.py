"""Deterministic, RNG-independent tensor filler.

Synthetic weights / inputs for parity tests and ``bench.py`` must be identical in
the survey container (where the reference is importable), on the GPU box and in
the oracle, so nothing here touches a framework RNG: every element is a
splitmix64 hash of (crc32(key), flat index).  SURVEY.md section 7 step 1 / 8(d).

The value *ranges* are chosen so that a randomly initialised
``DepthEstimationNet`` (reference ``network/RDM_Net.py:25-135``) produces DORN
counts away from zero (a zero count makes the geometric mean 0 -> inf/NaN in
``network/RDM_Net.py:117``).
"""
import zlib

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def unit(key: str, n: int) -> np.ndarray:
    """n float64 values in [0, 1) determined only by (key, index)."""
    seed = np.uint64(zlib.crc32(key.encode("utf-8")))
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + seed * np.uint64(0x100000001) + np.uint64(1)) * _M1
        z ^= z >> np.uint64(30)
        z *= _M2
        z ^= z >> np.uint64(27)
        z *= _M3
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def uniform(key: str, shape, lo: float, hi: float, dtype=np.float32) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * unit(key, n)).astype(dtype).reshape(shape)


def log_uniform(key: str, shape, lo: float, hi: float, dtype=np.float32) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return np.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * unit(key, n)).astype(dtype).reshape(shape)


def state_value(key: str, shape, dtype_name: str = "float32") -> np.ndarray:
    """Value for one state-dict entry of DepthEstimationNet, keyed by its name."""
    shape = tuple(int(s) for s in shape)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if int(np.prod(shape)) == 0 if len(shape) else False:
        return np.zeros(shape, dtype=np.float32)
    leaf = key.split(".")[-1]
    parent = key.split(".")[-2] if "." in key else ""
    if parent.startswith("norm"):
        if leaf == "weight":
            return uniform(key, shape, 0.5, 1.5)
        if leaf == "bias":
            return uniform(key, shape, -0.3, 0.3)
        if leaf == "running_mean":
            return uniform(key, shape, -0.2, 0.2)
        if leaf == "running_var":
            return uniform(key, shape, 0.5, 1.5)
    if key.startswith("weight_layer."):
        return uniform(key, shape, 0.5, 1.5)
    if len(shape) == 4:  # conv / conv-transpose weight: He-uniform on fan_in
        fan_in = shape[1] * shape[2] * shape[3]
        b = float(np.sqrt(6.0 / fan_in))
        return uniform(key, shape, -b, b)
    if leaf == "bias":
        v = uniform(key, shape, -0.1, 0.1)
        if key.endswith("conv2.bias") and shape == (180,):
            # DORN head: pair k compares channel 2k (A) with 2k+1 (B); push the first
            # half of the pairs towards B > A and the rest towards A > B so that the
            # ordinal count sits mid-range with non-trivial margins.
            k = np.arange(90)
            v[1::2] += np.where(k < 45, 0.45, -0.45).astype(np.float32)
            v[0::2] += 0.25
        return v
    return uniform(key, shape, -0.1, 0.1)


def fill_state_dict(state_dict) -> None:
    """In-place deterministic fill of a torch ``state_dict`` (any device)."""
    import torch

    with torch.no_grad():
        for key, t in state_dict.items():
            v = state_value(key, tuple(t.shape))
            t.copy_(torch.from_numpy(v).to(dtype=t.dtype).reshape(t.shape))


def synthetic_batch(batch: int, height: int, width: int, seed: int = 1234):
    """x ~ U[0,1) (B,3,H,W) f32; y ~ U[0.5,9.5) (B,1,H,W) f32 with 5 % zeros."""
    x = uniform(f"x/{seed}/{batch}x{height}x{width}", (batch, 3, height, width), 0.0, 1.0)
    y = uniform(f"y/{seed}/{batch}x{height}x{width}", (batch, 1, height, width), 0.5, 9.5)
    hole = unit(f"hole/{seed}/{batch}x{height}x{width}", batch * height * width).reshape(y.shape) < 0.05
    y[hole] = 0.0
    return x, y


# Seeds of the parity fixtures (tests/golden/make_golden.py, tests/, __graft_entry__.smoke()).  They were searched so that
# every DORN pair decision of the reference on that input has a margin: the decision `clamp(b) > clamp(a)` (RDM_Net.py:330-342)
# is unchanged when both logits move by up to +-2.5e-4 (25x the f32 conv noise at this logit scale).  With such inputs the
# ordinal indices are asserted EQUAL outright - no "skip the near-ties" escape hatch ("fixtures should be built
# with margins").  make_golden.py re-checks the margin on the reference's own logits and refuses to write the fixture otherwise.
MARGIN_SEEDS = {"train228": 7, "eval226": 1, "train228x304": 10}
DORN_MARGIN = 2.5e-4


def dorn_safe_mask(logits, thr=DORN_MARGIN):
    """(B, K, h, w) bool: True where the ordinal decision of the (a, b) logit pair cannot flip under a +-thr perturbation of both."""
    lg = np.asarray(logits, dtype=np.float64)
    a, b = lg[:, 0::2], lg[:, 1::2]

    def c(v):
        return np.clip(v, 1e-8, 1e4)
    always = (c(b - thr) - c(a + thr)) > 0
    never = (c(b + thr) - c(a - thr)) <= 0
    return always | never


def dorn_unsafe_pairs(logits, thr=DORN_MARGIN):
    """Number of (a, b) logit pairs whose ordinal decision could flip under a +-thr perturbation of both logits."""
    return int((~dorn_safe_mask(logits, thr)).sum())


def tap_subsample(t):
    """Element-wise parity lattice for an activation tap: ``t`` is (B, C, H, W) (the reference's NCHW) or (pixels, C) (our pixel-major block
    buffers); returns the float32 values at every r-th pixel (row-major over B, H, W) and every c-th channel, r = (pixels // 61) | 1,
    c = (C // 53) | 1: both odd, so the lattice drifts through every lane / tile position class of the kernels' power-of-two and
    48-multiple tiles; ~60 x 50 values per tap whatever its size (tests/golden/make_golden.py stores them, tests/test_gpu_net.py compares)."""
    import numpy as np
    if hasattr(t, "detach"):
        t = t.detach().cpu()
        if t.dim() == 4:
            t = t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
        t = t.numpy()
    t = np.asarray(t)
    r, c = (t.shape[0] // 61) | 1, (t.shape[1] // 53) | 1
    return np.ascontiguousarray(t[::r, ::c], dtype=np.float32)
