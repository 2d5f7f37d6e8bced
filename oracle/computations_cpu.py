"""ORACLE (test infrastructure, never shipped, never imported by md_rdm_amd/).

CPU restatement in numpy of the post-processing math of az16/MD_RDM
(``network/computations.py``, ``network/RDM_Net.py`` Ordinal_Layer / Quantization /
Weights, ``loss.py``, ``utils.py:195-211``).  Every function cites the reference lines
it follows.  Pinned against fixtures produced by running the reference itself
(tests/golden/make_golden.py -> tests/golden/op_goldens.npz); see tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "md_rdm_amd", "data")


# ----------------------------------------------------------------------------------
# geometric mean / resize / pyramid
# ----------------------------------------------------------------------------------
def quick_gm(t, rc):
    """computations.py:244-255.  NOTE the reference squares ``rc``: exponent = 1/rc**2.
    t: (B, N, 1).  Integer input is promoted to float32 (torch.pow(int64, float)).
    The reference evaluates prod(pow(t, e)) in the tensor dtype; this restatement evaluates
    the same quantity, exp(e * sum(log t)), in float64 and rounds ONCE to that dtype, so it
    sits within one rounding of the exact value (the reference's own float32 chain of
    N pow+mul roundings deviates from it by up to ~1e-6 relative)."""
    t = np.asarray(t)
    dt = np.float32 if t.dtype.kind in "iu" else t.dtype
    e = 1.0 / (rc * rc)
    with np.errstate(divide="ignore"):
        return np.exp(e * np.sum(np.log(t.astype(np.float64)), axis=1)).astype(dt)


def _cubic_weights(t, A=-0.75):
    # cubic convolution coefficients (Keys), the ones ATen's upsample_bicubic2d uses
    def c1(x):
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def c2(x):
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A

    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], axis=-1)


def _bicubic_matrix(n_in, n_out):
    """Row-stochastic (n_out, n_in) matrix of 1-D bicubic interpolation,
    align_corners=False, border indices clamped (no antialias)."""
    scale = n_in / n_out
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(src)
    w = _cubic_weights(src - i0)
    M = np.zeros((n_out, n_in), dtype=np.float64)
    for k in range(4):
        idx = np.clip(i0.astype(np.int64) - 1 + k, 0, n_in - 1)
        np.add.at(M, (np.arange(n_out), idx), w[:, k])
    return M


def resize(depth_map, newsize):
    """computations.py:308-311: ``.double()`` + F.interpolate(size=newsize, bicubic,
    align_corners=False).  An int ``newsize`` yields a SQUARE output whatever the input.
    Bit-exact with the reference's torch CPU path: the rounding sequence is restated in
    oracle/bicubic_aten.c (see its header); ``resize_matrix_form`` below is the same operator as two
    dense matrices (1e-13 agreement, kept as an independent cross-check of the taps and weights)."""
    from . import _native
    x = np.ascontiguousarray(np.asarray(depth_map, dtype=np.float64))
    if isinstance(newsize, int):
        newsize = (newsize, newsize)
    B, Cc, H, W = x.shape
    out = np.empty((B, Cc, newsize[0], newsize[1]), dtype=np.float64)
    _native.lib().rdm_oracle_resize_bicubic_f64(x.ctypes.data, out.ctypes.data, B * Cc, H, W, newsize[0], newsize[1])
    return out


def resize_matrix_form(depth_map, newsize):
    x = np.asarray(depth_map, dtype=np.float64)
    if isinstance(newsize, int):
        newsize = (newsize, newsize)
    My = _bicubic_matrix(x.shape[2], newsize[0])
    Mx = _bicubic_matrix(x.shape[3], newsize[1])
    return np.einsum("oh,bchw,pw->bcop", My, x, Mx, optimize=True)


def upsample(depth_map):
    """computations.py:357-360: ``.double()`` + nearest x2."""
    x = np.asarray(depth_map, dtype=np.float64)
    return np.repeat(np.repeat(x, 2, axis=2), 2, axis=3)


def multi_upsample(depth_map, n):
    """computations.py:362-366 (n == 0 returns the input unchanged, dtype included)."""
    for _ in range(n):
        depth_map = upsample(depth_map)
    return depth_map


def decompose_depth_map(dn, n, relative_map=False):
    """computations.py:368-392.  Returns [F_n, F_{n-1}, ..., F_1, (d_0)] - the reference
    order BEFORE the caller's ``[::-1]``."""
    out = []
    while n >= 1:
        dn_1 = resize(dn, 2 ** (n - 1))
        out.append(np.asarray(dn, dtype=np.float64) / upsample(dn_1))
        dn, n = dn_1, n - 1
    if not relative_map:
        out.append(dn)
    return out


def relative_fine_detail_matrix(fine_detail_rows):
    """computations.py:423-484: bucket candidates by side length, log, stack on dim 1.
    Returns list of (B, n_candidates, H*W) float64 arrays, smallest size first."""
    slots = [[] for _ in range(8)]
    for row in fine_detail_rows:
        for m in row:
            slots[int(round(math.log2(m.shape[2])))].append(m)
    mats = []
    for cands in slots:
        if cands:
            B = cands[0].shape[0]
            mats.append(np.concatenate([np.log(c).reshape(B, 1, -1) for c in cands], axis=1))
    return mats


def make_pred(weights, mats, relative_only=False):
    """computations.py:512-528 (+ RDM_Net.py:443-491): per level
    ``A[i][b].T.float() @ w[i].float()`` reshaped to (B,1,s,s), float32."""
    if relative_only:
        weights = weights[1:]
    out = []
    for i, A in enumerate(mats):
        B, _, M = A.shape
        w = np.asarray(weights[i], dtype=np.float32).reshape(-1, 1)
        s = int(math.sqrt(M))
        out.append(np.stack([A[b].T.astype(np.float32) @ w for b in range(B)]).reshape(B, 1, s, s))
    return out


def recombination(components, n=7):
    """computations.py:394-421: sum of nearest-upsampled log components at 2**n."""
    comps = list(components)
    if comps[0].shape[2] == 1:
        d0 = multi_upsample(comps.pop(0), n)
        result = multi_upsample(comps.pop(0), n - 1)
        for i, c in enumerate(comps):
            result = result + multi_upsample(c, n - (i + 2))
        return d0 + result
    result = multi_upsample(comps.pop(0), n - 1)
    for i, c in enumerate(comps):
        result = result + multi_upsample(c, n - (i + 2))
    return result


def squared_err_sum(yhat, y):
    """computations.py:499-510,530-544: sum_i MSE(yhat_i, y_i) (detached in the reference)."""
    y = list(y)
    if yhat[0].shape[2] > y[0].shape[2]:
        y.pop(0)
    return float(sum(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2) for a, b in zip(yhat, y)))


# ----------------------------------------------------------------------------------
# SID labels / ordinal loss / DORN head
# ----------------------------------------------------------------------------------
def depth2label_sid(depth, K=90.0, alpha=0.02, beta=10.0):
    """utils.py:195-211.  alpha, beta, K are float32 0-dim tensors in the reference, so
    alpha = float32(0.02) and log(beta/alpha) is a float32 value; the tensor arithmetic
    runs in the dtype of ``depth``.  ``.int()`` truncates after max(.,0)."""
    depth = np.asarray(depth)
    dt = depth.dtype if depth.dtype.kind == "f" else np.float32
    a32, b32, k32 = np.float32(alpha), np.float32(beta), np.float32(K)
    den = np.log(b32 / a32, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        label = dt.type(k32) * np.log(depth.astype(dt) / dt.type(a32)) / dt.type(den)
    label = np.maximum(label, dt.type(0))
    return label.astype(np.int32)


def ordinal_loss(P, target):
    """loss.py:8-59.  P (N,C,H,W) float64, target (N,1,H,W) int.  float32 log/sum."""
    N, C, H, W = P.shape
    Kidx = np.arange(C, dtype=np.int64).reshape(1, C, 1, 1)
    m0 = Kidx <= target
    a = np.clip(P[np.broadcast_to(m0, P.shape)], 1e-8, 1e8).astype(np.float32)
    b = np.clip(1.0 - P[~np.broadcast_to(m0, P.shape)], 1e-8, 1e8).astype(np.float32)
    s = np.sum(np.log(a), dtype=np.float32) + np.sum(np.log(b), dtype=np.float32)
    return np.float32(s / np.float32(-(N * H * W)))


def ordinal_loss_grad(P, target):
    """d loss / d P of the expression above (clamp passes gradient inside [1e-8,1e8])."""
    N, C, H, W = P.shape
    Kidx = np.arange(C, dtype=np.int64).reshape(1, C, 1, 1)
    m0 = np.broadcast_to(Kidx <= target, P.shape)
    scale = -1.0 / (N * H * W)
    g = np.zeros_like(P)
    inside0 = (P >= 1e-8) & (P <= 1e8)
    q = 1.0 - P
    inside1 = (q >= 1e-8) & (q <= 1e8)
    with np.errstate(divide="ignore"):
        g0 = (np.float32(1.0) / np.clip(P, 1e-8, 1e8).astype(np.float32)).astype(np.float64)
        g1 = (np.float32(1.0) / np.clip(q, 1e-8, 1e8).astype(np.float32)).astype(np.float64)
    g[m0 & inside0] = scale * g0[m0 & inside0]
    g[~m0 & inside1] = -scale * g1[~m0 & inside1]
    return g


def dorn_ordinal_regression(x):
    """RDM_Net.py:313-345.  x (N,2K,H,W) float32 -> (decode int64 (N,1,H,W), P float64 (N,K,H,W))."""
    x = np.asarray(x, dtype=np.float32)
    a = np.clip(x[:, 0::2], np.float32(1e-8), np.float32(1e4)).astype(np.float64)
    b = np.clip(x[:, 1::2], np.float32(1e-8), np.float32(1e4)).astype(np.float64)
    m = np.maximum(a, b)
    ea, eb = np.exp(a - m), np.exp(b - m)
    P = eb / (ea + eb)
    decode = np.sum(P > 0.5, axis=1, keepdims=True).astype(np.int64)
    return decode, P


def dorn_backward(x, gP):
    """Gradient of sum(P * gP) w.r.t. the float32 logits (clamp gate + pair softmax)."""
    x = np.asarray(x, dtype=np.float32)
    _, P = dorn_ordinal_regression(x)
    t = gP * P * (1.0 - P)
    gate = (x >= np.float32(1e-8)) & (x <= np.float32(1e4))
    dx = np.zeros(x.shape, dtype=np.float64)
    dx[:, 1::2] = t
    dx[:, 0::2] = -t
    return (dx * gate).astype(np.float32)


# ----------------------------------------------------------------------------------
# Lloyd tables, ratio grids, ALS, paging (dormant relative decoders)
# ----------------------------------------------------------------------------------
_TABLE_BY_ID = {3: "008", 4: "016", 5: "032", 6: "064", 7: "128"}


def load_quant_tables():
    """RDM_Net.py:397-442.  The 008 table is missing upstream (.MISSING_LARGE_BLOBS) and is
    DERIVED as 016**2 - labelled derived, not upstream."""
    import scipy.io

    tabs = {}
    for s in ("016", "032", "064", "128"):
        m = scipy.io.loadmat(os.path.join(_DATA, f"depth_ratio_{s}_{s}_quant.mat"))
        tabs[s] = (m[f"depth_ratio_{s}_{s}_quant"][:, 0].copy(), m[f"depth_ratio_{s}_{s}_quant_inv"][:, 0].copy())
    tabs["008"] = (tabs["016"][0] ** 2, tabs["016"][1] ** 2)
    return tabs


def lloyd_quantization(r, quant, inv):
    """RDM_Net.py:286-311: idx = #{i : r >= quant_i} (thresholds cast to r's dtype, as
    torch does for tensor-vs-scalar comparison), r <- inv[idx] cast to r's dtype."""
    r = np.asarray(r)
    q = quant.astype(r.dtype)
    idx = np.sum(r[..., None] >= q, axis=-1)
    return inv[idx].astype(r.dtype), idx


def sparse_comparison_v1_raw(d3):
    """RDM_Net.py:244-257 before Lloyd: R[b,i,j] = d_i * pow(d_j, -1), float32."""
    B = d3.shape[0]
    v = np.asarray(d3, dtype=np.float32).reshape(B, -1)
    return v[:, :, None] * (np.float32(1.0) / v)[:, None, :]


def ratio_grid_raw(dn, dn_1):
    """RDM_Net.py:259-284 + computations.py:269-295 before Lloyd.  For fine pixel (r,c):
    row = dn[r,c] * 1/area, area = ones(h1,w1) with the 3x3 window of dn_1 whose origin is
    clamp(floor(r/2),0,h1-3), clamp(floor(c/2),0,w1-3).  float64 (B, H*W, h1*w1)."""
    B, _, H, W = dn.shape
    h1, w1 = dn_1.shape[2], dn_1.shape[3]
    dn = np.asarray(dn)
    dn_1 = np.asarray(dn_1, dtype=np.float64)
    out = np.empty((B, H * W, h1 * w1), dtype=np.float64)
    for r in range(H):
        for c in range(W):
            rs = int(min(max(r // 2, 0), h1 - 3))
            cs = int(min(max(c // 2, 0), w1 - 3))
            area = np.ones((B, h1, w1), dtype=np.float64)
            area[:, rs:rs + 3, cs:cs + 3] = dn_1[:, 0, rs:rs + 3, cs:cs + 3]
            out[:, r * W + c, :] = dn[:, 0, r, c].astype(np.float64)[:, None] * (1.0 / area.reshape(B, -1))
    return out


def als_rank1(R, n, limit, q_size=None):
    """computations.py:95-155 (paged, q_size = 4**(n-1)) and :38-85 (quadratic,
    q_size = 4**n).  float32 throughout; rmse and the arg-min are BATCH-GLOBAL; the
    q-update uses R.view(B,W,H) (a reinterpretation, not a transpose); the output is the p
    of the FIRST minimal rmse divided by quick_gm(p, H) (exponent 1/H**2)."""
    R = np.ascontiguousarray(R, dtype=np.float32)
    B, H, W = R.shape
    p_s = 2 ** (2 * n)
    q_s = 2 ** (2 * n - 2) if q_size is None else q_size
    assert H == p_s and W == q_s
    p = np.ones((B, p_s, 1), dtype=np.float32)
    q = np.ones((B, q_s, 1), dtype=np.float32)
    reg = np.float32(0.05)

    def rmse(p_, q_):
        return np.float32(np.sqrt(np.mean((p_ @ q_.reshape(B, 1, q_s) - R) ** 2, dtype=np.float32)))

    rec, vec = [rmse(p, q)], [p]
    Rv = R.reshape(B, W, H)
    for _ in range(limit):
        a = (q.reshape(B, 1, q_s) @ q) + reg            # (B,1,1)
        p = (R @ q) @ (np.float32(1.0) / a)             # b @ inverse(A)
        rec.append(rmse(p, q))
        vec.append(p)
        a = (p.reshape(B, 1, p_s) @ p) + reg
        q = (Rv @ p) @ (np.float32(1.0) / a)
    best = vec[int(np.argmin(np.array(rec)))]
    gm = quick_gm(best, H)                              # (B,1)
    out = best / gm.reshape(B, 1, 1)
    s = 2 ** n
    return out.reshape(B, 1, s, s), np.array(rec)


def split_matrix(d_n, d_n_1):
    """computations.py:201-216: row-major 16x16 pages of d_n and 8x8 pages of d_{n-1}."""
    ratio = d_n.shape[2] // 16
    first, second = [], []
    for i in range(ratio):
        for j in range(ratio):
            first.append(d_n[:, :, 16 * i:16 * i + 16, 16 * j:16 * j + 16])
            second.append(d_n_1[:, :, 8 * i:8 * i + 8, 8 * j:8 * j + 8])
    return first, second


def reconstruct(splits):
    """computations.py:218-238 - bug-as-spec: every column block is cat(splits[0:ratio], H),
    pages >= ratio are never used."""
    ratio = int(len(splits) ** 0.5)
    col = np.concatenate(splits[0:ratio], axis=2)
    return np.concatenate([col] * ratio, axis=3)


def relative_decoder_forward(x, decoder_id, tables=None):
    """Ordinal_Layer.forward for DORN=False (RDM_Net.py:358-396); decoder_id in 6..10."""
    tables = tables or load_quant_tables()
    oid = decoder_id - 3
    if oid == 3:
        R, _ = lloyd_quantization(sparse_comparison_v1_raw(x), *tables["008"])
        return als_rank1(R, 3, 30, q_size=64)[0]
    size_prev = {4: 8, 5: 16, 6: 32, 7: 64}[oid]
    dn_1 = resize(x, size_prev)
    q, inv = tables[_TABLE_BY_ID[oid]]
    if oid == 4:
        R, _ = lloyd_quantization(ratio_grid_raw(x, dn_1), q, inv)
        return als_rank1(R, 4, 100)[0]
    pages, pages_1 = split_matrix(x, dn_1)
    filled = []
    for a, b in zip(pages, pages_1):
        R, _ = lloyd_quantization(ratio_grid_raw(a, b), q, inv)
        filled.append(als_rank1(R, 4, 100)[0])
    return reconstruct(filled)


def depth_metrics(pred, target, names=("delta1", "delta2", "delta3", "mse", "mae", "log10", "absrel", "sqrel", "rmse")):
    """metrics.py:58-66 ``MetricComputation.compute`` (clamp pred to 1e-7, keep target > 0) with the metric functions of
    metrics.py:79-116 ('rmse' is RelativeMeanSquareError, :107-110,128) and the library's mse / mae (pytorch_lightning 1.1.7
    functional, third party: mean squared / absolute error).  Computed in the dtype of ``pred`` like the reference."""
    pred = np.asarray(pred)
    target = np.asarray(target).astype(pred.dtype)
    p = np.maximum(pred, pred.dtype.type(1e-07))
    m = target > 0
    assert m.sum() > 0, "invalid target!"
    p, t = p[m], target[m]
    r = np.maximum(p / t, t / p)
    f = {"delta1": lambda: (r < 1.25 ** 1).astype(np.float32).mean(), "delta2": lambda: (r < 1.25 ** 2).astype(np.float32).mean(),
         "delta3": lambda: (r < 1.25 ** 3).astype(np.float32).mean(), "mse": lambda: ((p - t) ** 2).mean(), "mae": lambda: np.abs(p - t).mean(),
         "log10": lambda: np.abs(np.log10(p) - np.log10(t)).mean(), "absrel": lambda: (np.abs(p - t) / t).mean(),
         "sqrel": lambda: ((p - t) ** 2 / t).mean(), "rmse": lambda: np.sqrt((p - t) ** 2 / t).mean()}
    return [float(f[n]()) for n in names]
