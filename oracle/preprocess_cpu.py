"""TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench_ops' cpu leg; never by the product).

CPU oracle of the reference's NYU input pipeline (dataloaders/nyu_dataloader.py:240-307).

The reference builds it from torchvision transforms on PIL images.  torchvision is not in this image
(and unpinned in the reference's requirements.txt); its PIL-backed functional ops are thin wrappers
over Pillow calls, and Pillow IS here (12.2.0), so the oracle has two layers:

  * ``pil_*``  - the pipeline written with the very Pillow calls torchvision's ops forward to
                 (F.resize -> Image.resize(BILINEAR), F.rotate -> Image.rotate(NEAREST),
                 F.center_crop -> Image.crop, F.hflip -> transpose(FLIP_LEFT_RIGHT),
                 ColorJitter -> ImageEnhance.Brightness/Contrast/Color, to_tensor).  This is the pin.
  * ``np_*``   - a numpy restatement of the Pillow C algorithms behind those calls (Resample.c 8bpc fixed
                 point / 32bpc double accumulation, Geometry.c affine_fixed nearest, Blend.c, the ITU-R
                 601-2 luma of Convert.c), checked bit-for-bit against the pil_ layer in
                 tests/test_oracle_preprocess.py.  The HIP kernels implement exactly this arithmetic.

The random draws of the reference (np.random.uniform for s / angle / flip, torchvision's RNG for the
jitter factors and order) are inputs here: parity is defined for given augmentation parameters.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


# ---------------------------------------------------------------------------------------------
# torchvision size conventions
# ---------------------------------------------------------------------------------------------
def resized_hw(h, w, size):
    """transforms.Resize(int): the smaller edge becomes `size` (nyu_dataloader.py:249,257)."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def center_crop_box(h, w, th, tw):
    """transforms.CenterCrop((th, tw)) on an image at least that large (nyu_dataloader.py:261)."""
    top = int(round((h - th) / 2.0))
    left = int(round((w - tw) / 2.0))
    return top, left


def rotate_matrix(angle, w, h):
    """Image.rotate's inverse affine matrix (Pillow Image.py, expand=False, center=None)."""
    angle = angle % 360.0
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    return m


def affine_fixed_coeffs(m):
    """Geometry.c affine_fixed: 16.16 fixed-point coefficients, the pixel-centre offset folded into a2 / a5."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


# ---------------------------------------------------------------------------------------------
# layer 1: Pillow
# ---------------------------------------------------------------------------------------------
def _pil_resize(img, size):
    from PIL import Image
    w, h = img.size
    nh, nw = resized_hw(h, w, size)
    return img.resize((nw, nh), Image.BILINEAR)


def pil_color_jitter(rgb_img, jitter):
    from PIL import ImageEnhance
    for name, f in jitter:
        enh = {"brightness": ImageEnhance.Brightness, "contrast": ImageEnhance.Contrast, "saturation": ImageEnhance.Color}[name]
        rgb_img = enh(rgb_img).enhance(f)
    return rgb_img


def pil_training_preprocess(rgb, depth, s, angle, flip, jitter, resize=250, output_size=(228, 304)):
    """nyu_dataloader.py:240-272 with the random draws passed in.  rgb (H,W,3) uint8, depth (H,W) float32."""
    from PIL import Image
    depth = (depth / s).astype(np.float32) if depth.dtype == np.float32 else depth / s
    rgb_i = Image.fromarray(rgb, "RGB")
    dep_i = Image.fromarray(np.asarray(depth, dtype=np.float32), "F")
    rgb_i = pil_color_jitter(rgb_i, jitter)
    rgb_i, dep_i = _pil_resize(rgb_i, resize), _pil_resize(dep_i, resize)
    rgb_i, dep_i = rgb_i.rotate(angle), dep_i.rotate(angle)
    s2 = int(resize * s)
    rgb_i, dep_i = _pil_resize(rgb_i, s2), _pil_resize(dep_i, s2)
    w, h = rgb_i.size
    top, left = center_crop_box(h, w, *output_size)
    box = (left, top, left + output_size[1], top + output_size[0])
    rgb_i, dep_i = rgb_i.crop(box), dep_i.crop(box)
    if flip:
        rgb_i, dep_i = rgb_i.transpose(Image.FLIP_LEFT_RIGHT), dep_i.transpose(Image.FLIP_LEFT_RIGHT)
    x = np.asarray(rgb_i, dtype=np.uint8).transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    y = np.asarray(dep_i, dtype=np.float32)[None]
    return x, y


def pil_validation_preprocess(rgb, depth, resize=250, output_size=(228, 304)):
    """nyu_dataloader.py:274-287."""
    from PIL import Image
    rgb_i = _pil_resize(Image.fromarray(rgb, "RGB"), resize)
    dep_i = _pil_resize(Image.fromarray(np.asarray(depth, dtype=np.float32), "F"), resize)
    w, h = rgb_i.size
    top, left = center_crop_box(h, w, *output_size)
    box = (left, top, left + output_size[1], top + output_size[0])
    x = np.asarray(rgb_i.crop(box), dtype=np.uint8).transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    y = np.asarray(dep_i.crop(box), dtype=np.float32)[None]
    return x, y


def pil_test_preprocess(rgb, depth, output_size=(228, 304)):
    """nyu_dataloader.py:289-307: Resize(500) -> CenterCrop((480, 640)) -> Resize(output_size) [a (h, w) tuple: exact size]."""
    from PIL import Image
    rgb_i = _pil_resize(Image.fromarray(rgb, "RGB"), 500)
    dep_i = _pil_resize(Image.fromarray(np.asarray(depth, dtype=np.float32), "F"), 500)
    w, h = rgb_i.size
    top, left = center_crop_box(h, w, 480, 640)
    box = (left, top, left + 640, top + 480)
    rgb_i, dep_i = rgb_i.crop(box), dep_i.crop(box)
    size = (output_size[1], output_size[0])
    rgb_i, dep_i = rgb_i.resize(size, Image.BILINEAR), dep_i.resize(size, Image.BILINEAR)
    x = np.asarray(rgb_i, dtype=np.uint8).transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    y = np.asarray(dep_i, dtype=np.float32)[None]
    return x, y


# ---------------------------------------------------------------------------------------------
# layer 2: numpy restatement of the Pillow C code
# ---------------------------------------------------------------------------------------------
def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs for the bilinear (triangle, support 1) filter over the whole axis.
    Returns (xmin[out], count[out], k[out, ksize]) with k in float64, normalised."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, np.int64)
    counts = np.zeros(out_size, np.int64)
    k = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w = 1.0 - a if a < 1.0 else 0.0
            k[xx, x] = w
            ww += w
        if ww != 0.0:
            k[xx, :xmax] /= ww
        xmins[xx], counts[xx] = xmin, xmax
    return xmins, counts, k


def _fixed(k):
    """normalize_coeffs_8bpc"""
    return np.where(k < 0, np.trunc(-0.5 + k * (1 << PRECISION_BITS)), np.trunc(0.5 + k * (1 << PRECISION_BITS))).astype(np.int64)


def _resample_axis_u8(img, out_size, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    xmins, counts, k = resample_coeffs(img.shape[0], out_size)
    kk = _fixed(k)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(counts[xx]):
            acc += img[xmins[xx] + x] * kk[xx, x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def _resample_axis_f32(img, out_size, axis):
    img = np.moveaxis(img, axis, 0).astype(np.float64)
    xmins, counts, k = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.float32)
    for xx in range(out_size):
        acc = np.zeros(img.shape[1:], np.float64)
        for x in range(counts[xx]):
            acc = acc + img[xmins[xx] + x] * k[xx, x]
        out[xx] = acc.astype(np.float32)
    return np.moveaxis(out, 0, axis)


def np_resize(img, nh, nw):
    """Image.resize((nw, nh), BILINEAR): horizontal pass, then vertical (each skipped when the size is unchanged)."""
    f = _resample_axis_u8 if img.dtype == np.uint8 else _resample_axis_f32
    if nw != img.shape[1]:
        img = f(img, nw, 1)
    if nh != img.shape[0]:
        img = f(img, nh, 0)
    return img


def np_rotate_nearest(img, angle):
    """Image.rotate(angle) = affine transform, NEAREST, fill 0, through Geometry.c's 16.16 fixed-point path."""
    if angle % 360.0 == 0:
        return img.copy()
    h, w = img.shape[:2]
    a0, a1, a2, a3, a4, a5 = affine_fixed_coeffs(rotate_matrix(angle, w, h))
    ys, xs = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    xx = a2 + a0 * xs + a1 * ys
    yy = a5 + a3 * xs + a4 * ys
    xin, yin = xx >> 16, yy >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def np_luma(rgb):
    """Convert.c rgb2l: ITU-R 601-2 luma in 16-bit fixed point."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def np_blend(deg, img, alpha):
    """Blend.c ImagingBlend(degenerate, image, alpha) in float32 arithmetic, truncating cast."""
    a = np.float32(alpha)
    d, i = deg.astype(np.int32), img.astype(np.int32)
    t = d.astype(np.float32) + a * (i - d).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)        # (UINT8) of a value already inside [0, 255]
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def np_color_jitter(rgb, jitter):
    for name, f in jitter:
        if name == "brightness":
            deg = np.zeros_like(rgb)
        elif name == "contrast":
            lum = np_luma(rgb)
            mean = int(lum.astype(np.float64).sum() / lum.size + 0.5)
            deg = np.full_like(rgb, mean)
        else:
            deg = np.repeat(np_luma(rgb)[..., None], 3, axis=2)
        rgb = np_blend(deg, rgb, f)
    return rgb


def np_training_preprocess(rgb, depth, s, angle, flip, jitter, resize=250, output_size=(228, 304)):
    depth = np.asarray(depth / s, dtype=np.float32)
    rgb = np_color_jitter(rgb, jitter)
    h, w = rgb.shape[:2]
    nh, nw = resized_hw(h, w, resize)
    rgb, depth = np_resize(rgb, nh, nw), np_resize(depth, nh, nw)
    rgb, depth = np_rotate_nearest(rgb, angle), np_rotate_nearest(depth, angle)
    nh2, nw2 = resized_hw(nh, nw, int(resize * s))
    rgb, depth = np_resize(rgb, nh2, nw2), np_resize(depth, nh2, nw2)
    top, left = center_crop_box(nh2, nw2, *output_size)
    rgb = rgb[top:top + output_size[0], left:left + output_size[1]]
    depth = depth[top:top + output_size[0], left:left + output_size[1]]
    if flip:
        rgb, depth = rgb[:, ::-1], depth[:, ::-1]
    return rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255), np.ascontiguousarray(depth)[None]


def np_validation_preprocess(rgb, depth, resize=250, output_size=(228, 304)):
    h, w = rgb.shape[:2]
    nh, nw = resized_hw(h, w, resize)
    rgb, depth = np_resize(rgb, nh, nw), np_resize(np.asarray(depth, np.float32), nh, nw)
    top, left = center_crop_box(nh, nw, *output_size)
    rgb = rgb[top:top + output_size[0], left:left + output_size[1]]
    depth = depth[top:top + output_size[0], left:left + output_size[1]]
    return rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255), np.ascontiguousarray(depth)[None]


def np_test_preprocess(rgb, depth, output_size=(228, 304)):
    h, w = rgb.shape[:2]
    nh, nw = resized_hw(h, w, 500)
    rgb, depth = np_resize(rgb, nh, nw), np_resize(np.asarray(depth, np.float32), nh, nw)
    top, left = center_crop_box(nh, nw, 480, 640)
    rgb, depth = rgb[top:top + 480, left:left + 640], depth[top:top + 480, left:left + 640]
    rgb, depth = np_resize(np.ascontiguousarray(rgb), *output_size), np_resize(np.ascontiguousarray(depth), *output_size)
    return rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255), np.ascontiguousarray(depth)[None]
