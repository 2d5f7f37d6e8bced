"""``network.computations`` of the reference, re-hosted on hand-written gfx950 kernels.

Function names, argument meaning and return conventions follow /root/reference/network/
computations.py so the harness (and the reference's own callers) read unchanged; the bodies call
the C ABI of librdm_hip.so on the caller's HIP stream.  Tensors must live on the GPU: there is no
CPU path here (the CPU restatement lives in oracle/, for tests only).

Pyramids ("lists of fine-detail maps") are produced by one kernel into ONE packed buffer; the list
elements handed back are strided views of it, recognised again (through a registry of what such a view IS, see
``_Pyramid``) by the downstream functions (relative_fine_detail_matrix -> make_pred -> recombination), which then run fused
single-launch kernels on the packed form.
"""
import ctypes as C
import math

import torch

from .. import _lib

_f64 = torch.float64


def _need_gpu(t):
    if not t.is_cuda:
        raise _lib.RdmError("md_rdm_amd.network.computations runs on the GPU only; got a tensor on %s" % t.device)


def _level_off(k):
    return ((1 << (2 * k)) - 1) // 3


class _Pyramid:
    """Packed per-sample pyramid [level0 (1x1) | level1 (2x2) | ...]; ``kind`` tracks what it holds.
    The reference hands pyramids around as Python lists of per-level tensors (computations.py:368-392, 423-484); the level views this
    class gives out are recognised again by ``_pyramid_of`` through a REGISTRY keyed on what a view is - its storage, offset, shape
    and the version counter of the packed buffer at creation - not through attributes hung on tensor objects: a list that was
    re-assembled from views, detached or passed through ``list(...)`` is still recognised, while any in-place write to the packed
    buffer (version bump) or any tensor that is not exactly such a view falls back to the general (unfused) path.
    The registry is a small LRU that OWNS the pyramids it lists: while an entry exists its packed buffer is alive, so its address
    cannot have been handed to another tensor (no false match); an evicted entry only costs the fused path, never correctness."""

    def __init__(self, packed, n_levels, kind, first_level=0, flat=False):
        self.packed, self.n_levels, self.kind, self.first_level, self.flat = packed, n_levels, kind, first_level, flat
        self.version = packed._version

    def level_shape(self, k):
        B, s = self.packed.shape[0], 1 << k
        return (B, 1, s * s) if self.flat else (B, 1, s, s)

    def views(self):
        out = [self.packed[:, _level_off(k):_level_off(k + 1)].view(self.level_shape(k)) for k in range(self.first_level, self.n_levels)]
        key = self._key(out[0])
        _REGISTRY.pop(key, None)
        _REGISTRY[key] = self
        while len(_REGISTRY) > _REGISTRY_CAP:
            _REGISTRY.pop(next(iter(_REGISTRY)))
        return out

    @staticmethod
    def _key(t):
        return (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype)


_REGISTRY = {}                  # insertion-ordered: oldest first
_REGISTRY_CAP = 8               # a training step makes three pyramids (target levels, their log form, the prediction)


def _pyramid_of(tensors):
    """The packed pyramid behind a list of level views (ascending sizes), or None."""
    if not tensors or not isinstance(tensors[0], torch.Tensor):
        return None
    pyr = _REGISTRY.get(_Pyramid._key(tensors[0]))
    if pyr is None:
        return None
    if pyr.packed._version != pyr.version or len(tensors) != pyr.n_levels - pyr.first_level:
        return None
    base, es = pyr.packed.storage_offset(), pyr.packed.stride(0)
    for i, t in enumerate(tensors):                                # every element must BE the corresponding level view
        k = pyr.first_level + i
        s = 1 << k
        if (not isinstance(t, torch.Tensor) or t.dtype != pyr.packed.dtype or tuple(t.shape) != pyr.level_shape(k)
                or t.untyped_storage().data_ptr() != pyr.packed.untyped_storage().data_ptr()
                or t.storage_offset() != base + _level_off(k) or t.stride(0) != es or t.stride(-1) != 1
                or (not pyr.flat and s > 1 and t.stride(2) != s)):
            return None
    return pyr


# ------------------------------------------------------------------------------------------
# geometric mean / resize / pyramid  (computations.py:244-255, :308-311, :357-392)
# ------------------------------------------------------------------------------------------
def gm_normalize(t, exponent):
    """t / exp(exponent * sum log t) per sample, float64.  Fuses quick_gm with the division every
    caller applies (RDM_Net.py:117, module.py:145-149)."""
    _need_gpu(t)
    B = t.shape[0]
    src = t.reshape(B, -1).to(_f64).contiguous()
    dst = torch.empty_like(src)
    _lib.check(_lib.lib().rdm_gm_normalize_f64(_lib.ptr(src), _lib.ptr(dst), None, B, src.shape[1], float(exponent), _lib.stream()))
    return dst.view(t.shape)


def quick_gm(t, rc):
    """computations.py:244-255 - NOTE the reference squares ``rc`` (exponent 1/rc**2).  t: (B,N,1).
    Evaluated as exp(e*sum(log t)) in float64 and rounded once to the reference's result dtype
    (float32 for int/float32 input)."""
    _need_gpu(t)
    B = t.shape[0]
    src = t.reshape(B, -1).to(_f64).contiguous()
    gm = torch.empty(B, dtype=_f64, device=t.device)
    _lib.check(_lib.lib().rdm_gm_normalize_f64(_lib.ptr(src), None, _lib.ptr(gm), B, src.shape[1], 1.0 / (rc * rc), _lib.stream()))
    out_dt = t.dtype if t.dtype in (torch.float32, torch.float64) else torch.float32
    return gm.to(out_dt).view(B, 1)


def resize(depth_map, newsize):
    """computations.py:308-311: ``.double()`` + bicubic (align_corners=False); an int size gives a SQUARE."""
    _need_gpu(depth_map)
    if isinstance(newsize, int):
        newsize = (newsize, newsize)
    B, Cc, H, W = depth_map.shape
    src = depth_map.to(_f64).contiguous()
    dst = torch.empty(B, Cc, newsize[0], newsize[1], dtype=_f64, device=src.device)
    _lib.check(_lib.lib().rdm_resize_bicubic_f64(_lib.ptr(src), _lib.ptr(dst), B * Cc, H, W, newsize[0], newsize[1], _lib.stream()))
    return dst


def upsample(depth_map):
    """computations.py:357-360 (nearest x2, float64) - pure indexing, kept as a device view op."""
    return depth_map.double().repeat_interleave(2, 2).repeat_interleave(2, 3)


def multi_upsample(depth_map, n):
    for _ in range(n):
        depth_map = upsample(depth_map)
    return depth_map


def decompose_depth_map(container, dn, n, relative_map=False):
    """computations.py:368-392: appends [F_n, ..., F_1, (d_0)] to ``container`` (one kernel).
    The returned tensors are views of one packed pyramid (see module docstring)."""
    _need_gpu(dn)
    B, Cc, S, S2 = dn.shape
    assert Cc == 1 and S == S2 == (1 << n), "decompose_depth_map needs a (B,1,2^n,2^n) map"
    src = dn.to(_f64).contiguous()
    per = _level_off(n + 1)
    packed = torch.empty(B, per, dtype=_f64, device=dn.device)
    _lib.check(_lib.lib().rdm_decompose_f64(_lib.ptr(src), _lib.ptr(packed), B, n, _lib.stream()))
    pyr = _Pyramid(packed, n + 1, "levels", first_level=1 if relative_map else 0)
    container.extend(pyr.views()[::-1])
    return container


def idx_from_size(fine_detail_map):
    return int(round(math.log2(fine_detail_map.shape[2])))


def relative_fine_detail_matrix(fine_detail_rows, cuda):
    """computations.py:423-484.  Live graph (one candidate row that came out of
    decompose_depth_map): returns tagged views so make_pred can fuse log + weighting into one
    launch.  General multi-candidate case (dormant decoders d_6..d_9): per-level log + stack with
    device ops."""
    if len(fine_detail_rows) == 1:
        pyr = _pyramid_of(list(fine_detail_rows[0]))
        if pyr is not None and pyr.kind == "levels":
            lp = _Pyramid(pyr.packed, pyr.n_levels, "log_pending", pyr.first_level, flat=True)   # same buffer, (B,1,HW) views: log + weighting pending
            return lp.views()
    slots = [[] for _ in range(8)]
    for row in fine_detail_rows:
        for m in row:
            slots[idx_from_size(m)].append(m)
    return [make_matrix(x, cuda) for x in slots if len(x)]


def make_matrix(list_of_candidates, cuda):
    B = list_of_candidates[0].shape[0]
    return torch.cat([torch.log(x).reshape(B, 1, -1) for x in list_of_candidates], dim=1)


class _FineDetailPred(torch.autograd.Function):
    """yhat_k = float32(log F_k) * w_k on the packed pyramid; gradient only w.r.t. the weights."""

    @staticmethod
    def forward(ctx, packed, n_levels, *w):
        B = packed.shape[0]
        wv = torch.cat([x.reshape(-1)[:1].float() for x in w]).contiguous()
        yhat = torch.empty(B, _level_off(n_levels), dtype=torch.float32, device=packed.device)
        _lib.check(_lib.lib().rdm_fine_detail_pred_f32(_lib.ptr(packed), _lib.ptr(wv), _lib.ptr(yhat), B, n_levels, _lib.stream()))
        ctx.save_for_backward(packed)
        ctx.n_levels = n_levels
        ctx.shapes = [x.shape for x in w]
        return yhat

    @staticmethod
    def backward(ctx, dyhat):
        (packed,) = ctx.saved_tensors
        B = packed.shape[0]
        dw = torch.empty(ctx.n_levels, dtype=torch.float32, device=packed.device)
        _lib.check(_lib.lib().rdm_fine_detail_pred_bwd(_lib.ptr(packed), _lib.ptr(dyhat.contiguous().float()), _lib.ptr(dw), B, ctx.n_levels, _lib.stream()))
        return (None, None) + tuple(dw[i].reshape(s) for i, s in enumerate(ctx.shapes))


class _CandidatesMatvec(torch.autograd.Function):
    """out[b][m] = sum_k float32(A[b][k][m]) * w[k] (the reference's A^T.float() @ w.float() for a level with several candidates);
    gradient only w.r.t. the weights, as in the reference graph (the matrices are detached logs)."""

    @staticmethod
    def forward(ctx, A, w):
        B, K, M = A.shape
        A = A.to(_f64).contiguous()
        wv = w.reshape(-1).float().contiguous()
        out = torch.empty(B, M, dtype=torch.float32, device=A.device)
        _lib.check(_lib.lib().rdm_candidates_matvec_f32(_lib.ptr(A), _lib.ptr(wv), _lib.ptr(out), B, K, M, _lib.stream()))
        ctx.save_for_backward(A)
        ctx.wshape = w.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (A,) = ctx.saved_tensors
        B, K, M = A.shape
        dw = torch.empty(K, dtype=torch.float32, device=A.device)
        _lib.check(_lib.lib().rdm_candidates_matvec_bwd(_lib.ptr(A), _lib.ptr(dout.contiguous().float()), _lib.ptr(dw), B, K, M, _lib.stream()))
        return None, dw.reshape(ctx.wshape)


def make_pred(w, A, cuda, relative_only):
    """computations.py:512-528 (mutates and returns ``A`` like the reference).  One weight per level
    on a packed pyramid -> single fused launch; otherwise the general per-level matvec."""
    weights = w[1:] if relative_only else w
    pyr = _pyramid_of(list(A))
    if pyr is not None and pyr.kind == "log_pending" and pyr.first_level == 0 and all(weights[i].numel() == 1 for i in range(pyr.n_levels)):
        yhat = _FineDetailPred.apply(pyr.packed, pyr.n_levels, *[weights[i] for i in range(pyr.n_levels)])
        yp = _Pyramid(yhat, pyr.n_levels, "yhat", 0)
        for i, v in enumerate(yp.views()):
            A[i] = v
        return A
    for i in range(len(A)):
        B, M = A[i].shape[0], A[i].shape[2]
        s = int(math.sqrt(M))
        if A[i].is_cuda and 1 <= A[i].shape[1] <= 8:
            A[i] = _CandidatesMatvec.apply(A[i], weights[i]).view(B, 1, s, s)
        else:
            A[i] = torch.matmul(A[i].transpose(1, 2).float(), weights[i].float()).view(B, 1, s, s)
    return A


class _Recombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yhat, n_levels, n_out, first_level):
        B = yhat.shape[0]
        out = torch.empty(B, 1, 1 << n_out, 1 << n_out, dtype=_f64, device=yhat.device)
        _lib.check(_lib.lib().rdm_recombine_f64(_lib.ptr(yhat), _lib.ptr(out), B, n_levels, n_out, first_level, _lib.stream()))
        ctx.meta = (B, n_levels, n_out, first_level, yhat.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, n_levels, n_out, first_level, shape = ctx.meta
        dy = torch.empty(shape, dtype=torch.float32, device=dout.device)
        _lib.check(_lib.lib().rdm_recombine_bwd(_lib.ptr(dout.contiguous().double()), _lib.ptr(dy), B, n_levels, n_out, first_level, _lib.stream()))
        return dy, None, None, None


def recombination(list_of_components, n=7):
    """computations.py:394-421: sum of nearest-upsampled components at 2**n (pops like the reference)."""
    pyr = _pyramid_of(list(list_of_components))
    if pyr is not None and pyr.kind == "yhat":
        del list_of_components[:2]
        return _Recombine.apply(pyr.packed, pyr.n_levels, n, pyr.first_level)
    if list_of_components[0].shape[2] == 1:
        d_0 = multi_upsample(list_of_components.pop(0), n)
        result = multi_upsample(list_of_components.pop(0), n - 1)
        for i in range(len(list_of_components)):
            result = result + multi_upsample(list_of_components[i], n - (i + 2))
        return d_0 + result
    result = multi_upsample(list_of_components.pop(0), n - 1)
    for i in range(len(list_of_components)):
        result = result + multi_upsample(list_of_components[i], n - (i + 2))
    return result


def squared_err(yhat, y, cuda):
    """computations.py:530-544."""
    if yhat[0].shape[2] > y[0].shape[2]:
        y.pop(0)
    return [torch.nn.functional.mse_loss(yhat[i].double(), y[i].double()) for i in range(len(yhat))]


def optimize_components(yhat, y, cuda):
    """computations.py:499-510: returns (pred, DETACHED sum of per-level MSEs)."""
    loss = squared_err(yhat, y, cuda)
    return yhat, torch.sum(torch.stack([l.detach() for l in loss]))


# ------------------------------------------------------------------------------------------
# DORN head (RDM_Net.py:313-345) as an autograd node over two launches
# ------------------------------------------------------------------------------------------
class _Dorn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C2, H, W = x.shape
        x = x.contiguous()
        ord_c1 = torch.empty(B, C2 // 2, H, W, dtype=_f64, device=x.device)
        decode = torch.empty(B, 1, H, W, dtype=torch.int64, device=x.device)
        _lib.check(_lib.lib().rdm_dorn_fwd(_lib.ptr(x), _lib.ptr(ord_c1), _lib.ptr(decode), B, C2 // 2, H * W, _lib.stream()))
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(decode)
        return decode, ord_c1

    @staticmethod
    def backward(ctx, _gdecode, gord):
        (x,) = ctx.saved_tensors
        B, C2, H, W = x.shape
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().rdm_dorn_bwd(_lib.ptr(x), _lib.ptr(gord.contiguous()), _lib.ptr(dx), B, C2 // 2, H * W, _lib.stream()))
        return dx


def dorn_ordinal_regression(x):
    _need_gpu(x)
    return _Dorn.apply(x.float())


# ------------------------------------------------------------------------------------------
# relative decoders: ratio grids + Lloyd, ALS, paging
# ------------------------------------------------------------------------------------------
def ratio_grid_lloyd_dense(d3, quant, inv):
    """RDM_Net.py:244-257 + :286-311 (id 3): (B,1,S,S) f32 -> (B,S*S,S*S) f32."""
    _need_gpu(d3)
    B, _, H, W = d3.shape
    n = H * W
    d = d3.float().contiguous()
    R = torch.empty(B, n, n, dtype=torch.float32, device=d.device)
    _lib.check(_lib.lib().rdm_ratio_grid_lloyd_dense(_lib.ptr(d), _lib.ptr(R), B, n, _lib.ptr(quant), _lib.ptr(inv), _lib.stream()))
    return R


def ratio_grid_lloyd_paged(dn, dn_1, quant, inv, quantize=True):
    """RDM_Net.py:259-311 + computations.py:269-295 for every 16x16 page at once:
    dn (B,1,S,S), dn_1 (B,1,S/2,S/2) -> (P,B,256,64) float64, P = (S/16)^2 row-major pages."""
    _need_gpu(dn)
    B, _, S, _ = dn.shape
    a = dn.float().contiguous()
    b = dn_1.to(_f64).contiguous()
    P = (S // 16) ** 2
    R = torch.empty(P, B, 256, 64, dtype=_f64, device=a.device)
    _lib.check(_lib.lib().rdm_ratio_grid_lloyd_paged(_lib.ptr(a), _lib.ptr(b), _lib.ptr(R), B, S, _lib.ptr(quant), _lib.ptr(inv), int(quantize), _lib.stream()))
    return R


def _als(R, groups, batch, rows, cols, limit):
    L = _lib.lib()
    R = R.contiguous()
    nbytes = int(L.rdm_als_workspace_bytes(groups, batch, rows, cols, limit))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=R.device)
    p = torch.empty(groups, batch, rows, dtype=torch.float32, device=R.device)
    _lib.check(L.rdm_als_rank1(_lib.ptr(R), int(R.dtype == _f64), _lib.ptr(p), groups, batch, rows, cols, limit, C.c_void_p(ws.data_ptr()), nbytes, _lib.stream()))
    return p


def als_pages(R, limit=100):
    """(P,B,256,64) -> (P,B,1,16,16): one LDS-resident ALS workgroup per (page, sample); the rmse
    arg-min is batch-global per page, as in the reference's per-page calls."""
    P, B = R.shape[0], R.shape[1]
    return _als(R, P, B, 256, 64, limit).view(P, B, 1, 16, 16)


def als_pages_fused(dn, dn_1, quant, inv, limit=100):
    """ratio_grid_lloyd_paged + als_pages in one call (rdm_als_rank1_paged): every ALS thread forms its row of the quantised grid from
    dn and the 3x3 window of dn_1 - the (P,B,256,64) float64 grid never exists in HBM.  Bit-identical with the two-step path."""
    _need_gpu(dn)
    L = _lib.lib()
    B, _, S, _ = dn.shape
    a = dn.float().contiguous()
    b = dn_1.to(_f64).contiguous()
    P = (S // 16) ** 2
    nbytes = int(L.rdm_als_workspace_bytes(P, B, 256, 64, limit))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    p = torch.empty(P, B, 256, dtype=torch.float32, device=a.device)
    _lib.check(L.rdm_als_rank1_paged(_lib.ptr(a), _lib.ptr(b), _lib.ptr(p), B, S, _lib.ptr(quant), _lib.ptr(inv), limit, C.c_void_p(ws.data_ptr()), nbytes, _lib.stream()))
    return p.view(P, B, 1, 16, 16)


def alternating_least_squares(sparse_m, n, cuda, limit=30, debug=False):
    """computations.py:95-155."""
    _need_gpu(sparse_m)
    B, H, W = sparse_m.shape
    assert H == 2 ** (2 * n) and W == 2 ** (2 * n - 2)
    s = 2 ** n
    return _als(sparse_m, 1, B, H, W, limit).view(B, 1, s, s)


def quadratic_als(sparse_m, cuda, n=3, limit=30, debug=False):
    """computations.py:38-85."""
    _need_gpu(sparse_m)
    B, H, W = sparse_m.shape
    assert H == W == 2 ** (2 * n)
    s = 2 ** n
    return _als(sparse_m, 1, B, H, W, limit).view(B, 1, s, s)


def _pages(t, page):
    """(B,1,S,S) -> list of (S/page)^2 row-major (B,1,page,page) pages, dtype kept (pure indexing: a device view + one copy)."""
    B, _, S, _ = t.shape
    r = S // page
    return list(t.reshape(B, 1, r, page, r, page).permute(2, 4, 0, 1, 3, 5).reshape(r * r, B, 1, page, page).contiguous())


def split_matrix(d_n, d_n_1):
    """computations.py:201-216: 16x16 pages of d_n and the matching 8x8 pages of d_{n-1}, each in its own dtype (the reference
    hands the float64 coarse map through unchanged).  float32 maps go through rdm_page_split_f32."""
    _need_gpu(d_n)
    B, _, S, _ = d_n.shape
    if d_n.dtype != torch.float32:
        pa = _pages(d_n, 16)
    else:
        a = d_n.contiguous()
        pa = torch.empty((S // 16) ** 2, B, 1, 16, 16, dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().rdm_page_split_f32(_lib.ptr(a), _lib.ptr(pa), B, S, 16, _lib.stream()))
        pa = list(pa)
    if d_n_1.dtype != torch.float32:
        pb = _pages(d_n_1, 8)
    else:
        b = d_n_1.contiguous()
        pb = torch.empty((S // 16) ** 2, B, 1, 8, 8, dtype=torch.float32, device=b.device)
        _lib.check(_lib.lib().rdm_page_split_f32(_lib.ptr(b), _lib.ptr(pb), B, S // 2, 8, _lib.stream()))
        pb = list(pb)
    return pa, pb


def reconstruct(splits):
    """computations.py:218-238, bug-as-spec: only the first sqrt(P) pages are used."""
    P = len(splits)
    ratio = int(P ** 0.5)
    B, _, page, _ = splits[0].shape
    pages = torch.stack([s.float() for s in splits]).contiguous()
    out = torch.empty(B, 1, ratio * page, ratio * page, dtype=torch.float32, device=pages.device)
    _lib.check(_lib.lib().rdm_page_reconstruct_f32(_lib.ptr(pages), _lib.ptr(out), B, ratio * page, page, _lib.stream()))
    return out


def find_nans(container):
    return any(bool(torch.any(t.isnan())) for t in container)
