"""NYU-v2 input pipeline, MI355X-first (reference: dataloaders/nyu_dataloader.py, dataloaders/dataloader.py).

The reference decodes one sample per DataLoader worker and runs ~10 PIL operations on it on the CPU
(``training_preprocess`` :240-272; measured 131 images/s per core).  At 200 images/s per GPU (8 GPUs: 1600/s) that needs ~12 busy host
cores; here the host only reads the raw uint8 / float32 arrays into pinned memory, and the whole
augmentation chain runs as a handful of batch-wide HIP launches (``rdm_nyu_preprocess``), bit-exact
with the Pillow arithmetic of the reference for given random draws.

  NYUDataset           raw samples (rgb uint8 HxWx3, depth float32 HxW) from .h5 files (h5py needed),
                       .npz files, or an in-memory list; same constructor keywords as the reference
                       where they make sense (output_size, resize, split).
  draw_training_params the reference's random draws (s, angle, flip, ColorJitter factors + order)
  NyuGpuPreprocessor   uploads a raw batch + its parameters and runs the C-ABI kernel chain
  PrefetchLoader       background thread: read -> pinned buffers -> async H2D on a copy stream ->
                       GPU preprocessing, one batch ahead of the consumer
"""
import ctypes as C
import math
import os
import queue
import threading

import numpy as np
import torch

from .. import _lib

_OPS = {"brightness": 0, "contrast": 1, "saturation": 2}


class NyuAug(C.Structure):
    """rdm_nyu_aug (include/rdm_hip.h)"""
    _fields_ = [("depth_div", C.c_float), ("rot", C.c_int32 * 6), ("h2", C.c_int32), ("w2", C.c_int32), ("top", C.c_int32),
                ("left", C.c_int32), ("flip", C.c_int32), ("op", C.c_int32 * 3), ("factor", C.c_float * 3), ("crop2", C.c_int32 * 4)]


def resized_hw(h, w, size):
    """torchvision ``Resize(int)``: the smaller edge becomes ``size`` (nyu_dataloader.py:249,257)."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def _rotation_fixed(angle, w, h):
    """Image.rotate(angle) (expand=False, centre = image centre) as Geometry.c's 16.16 fixed-point affine."""
    angle = angle % 360.0
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    if angle == 0.0:
        m = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    else:
        m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
        m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def make_params(s, angle, flip, jitter, in_hw, resize, output_size):
    """One sample's rdm_nyu_aug from the reference's random draws.  ``jitter`` = [(name, factor), ...] in application order."""
    if not 1.0 <= s <= 4.0:
        raise ValueError(f"scale s={s} outside [1, 4] (the reference draws U[1, 1.5])")
    h1, w1 = resized_hw(in_hw[0], in_hw[1], resize)
    h2, w2 = resized_hw(h1, w1, int(resize * s))
    oh, ow = output_size
    if h2 < oh or w2 < ow:
        raise ValueError(f"centre crop {output_size} larger than the resized image {(h2, w2)}")
    p = NyuAug()
    p.depth_div = s
    p.rot[:] = _rotation_fixed(angle, w1, h1)
    p.h2, p.w2 = h2, w2
    p.top, p.left = int(round((h2 - oh) / 2.0)), int(round((w2 - ow) / 2.0))
    p.flip = int(bool(flip))
    ops = list(jitter) + [(None, 1.0)] * (3 - len(jitter))
    for i, (name, f) in enumerate(ops[:3]):
        p.op[i] = _OPS[name] if name is not None else -1
        p.factor[i] = f
    p.crop2[:] = [0, 0, h1, w1]
    return p


def test_params(in_hw, output_size):
    """test_preprocess (:289-307): Resize(500) -> CenterCrop((480, 640)) -> Resize(output_size) to the exact (h, w).
    Use with ``NyuGpuPreprocessor(resize=500, output_size=output_size)``."""
    h1, w1 = resized_hw(in_hw[0], in_hw[1], 500)
    if h1 < 480 or w1 < 640:
        raise ValueError(f"CenterCrop((480, 640)) larger than the resized image {(h1, w1)}")
    p = NyuAug()
    p.depth_div = 1.0
    p.rot[:] = _rotation_fixed(0.0, w1, h1)
    p.h2, p.w2 = output_size
    p.top = p.left = p.flip = 0
    for i in range(3):
        p.op[i], p.factor[i] = -1, 1.0
    p.crop2[:] = [int(round((h1 - 480) / 2.0)), int(round((w1 - 640) / 2.0)), 480, 640]
    return p


def identity_params(in_hw, resize, output_size):
    """validation_preprocess (:274-287): Resize + CenterCrop only."""
    return make_params(1.0, 0.0, False, [], in_hw, resize, output_size)


def draw_training_params(rng, in_hw, resize=250, output_size=(228, 304)):
    """The reference's draws (nyu_dataloader.py:241,247,252,264; ColorJitter(0.4, 0.4, 0.4) shuffles its three ops)."""
    s = float(rng.uniform(1.0, 1.5))
    names = ["brightness", "contrast", "saturation"]
    jitter = [(names[i], float(rng.uniform(0.6, 1.4))) for i in rng.permutation(3)]
    angle = float(rng.uniform(-5.0, 5.0))
    flip = bool(rng.uniform(0.0, 1.0) > 0.5)
    return {"s": s, "angle": angle, "flip": flip, "jitter": jitter}, make_params(s, angle, flip, jitter, in_hw, resize, output_size)


class NYUDataset:
    """Raw NYU samples.  ``path`` may be a directory of sparse-to-dense .h5 files (rgb (3,H,W) uint8, depth (H,W) float32;
    nyu_dataloader.py:169-174, needs h5py), a directory / list of .npz files with arrays ``rgb`` (H,W,3) and ``depth`` (H,W),
    or a list of (rgb, depth) arrays.  ``__getitem__`` returns the RAW pair: preprocessing is the GPU's job."""

    def __init__(self, path, split="train", output_size=(228, 304), resize=250, n_images=-1):
        if split not in ("train", "val", "test"):
            raise RuntimeError("Invalid dataset type: " + split + "\nSupported dataset types are: train, val, test")
        self.split, self.output_size, self.resize = split, tuple(output_size), resize
        if isinstance(path, (list, tuple)) and path and isinstance(path[0], tuple):
            self.images, self._mem = list(range(len(path))), list(path)
        else:
            self._mem = None
            if isinstance(path, (list, tuple)):
                self.images = list(path)
            else:
                root = os.path.join(path, "train" if split == "train" else "val") if os.path.isdir(os.path.join(path, "train")) else path
                self.images = sorted(os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f.endswith((".h5", ".npz")))
        assert len(self.images) > 0, "Found 0 images in subfolders of: " + str(path) + "\n"
        if n_images > 0:
            self.images = self.images[:n_images]

    def __len__(self):
        return len(self.images)

    def get_raw(self, index):
        if self._mem is not None:
            rgb, depth = self._mem[index]
        else:
            f = self.images[index]
            if f.endswith(".npz"):
                with np.load(f) as z:
                    rgb, depth = z["rgb"], z["depth"]
            else:
                try:
                    import h5py
                except ImportError as e:            # no CPU stand-in decoder: say what is missing
                    raise RuntimeError("reading .h5 samples needs h5py, which is not installed; convert to .npz (rgb HxWx3 uint8, depth HxW float32)") from e
                with h5py.File(f, "r") as h:
                    rgb, depth = np.transpose(np.array(h["rgb"]), (1, 2, 0)), np.array(h["depth"])
        return np.ascontiguousarray(rgb, dtype=np.uint8), np.ascontiguousarray(depth, dtype=np.float32)

    __getitem__ = get_raw


class NyuGpuPreprocessor:
    """Batch-wide GPU preprocessing through the C ABI.  All samples of a batch share the raw size."""

    def __init__(self, resize=250, output_size=(228, 304), device="cuda"):
        self.resize, self.output_size, self.device = resize, tuple(output_size), torch.device(device)
        self._ws = None

    def __call__(self, rgb, depth, params):
        """rgb (B,H,W,3) uint8 and depth (B,H,W) float32 device tensors, params: list of NyuAug -> x (B,3,oh,ow), y (B,1,oh,ow)."""
        if not rgb.is_cuda:
            raise _lib.RdmError("NyuGpuPreprocessor runs on the GPU only")
        L = _lib.lib()
        B, H, W, _ = rgb.shape
        h1, w1 = resized_hw(H, W, self.resize)
        oh, ow = self.output_size
        need = L.rdm_nyu_preprocess_workspace_bytes(B, H, W, h1, w1, ow)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=rgb.device)
        arr = (NyuAug * B)(*params)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        aug = host.to(rgb.device)
        x = torch.empty(B, 3, oh, ow, dtype=torch.float32, device=rgb.device)
        y = torch.empty(B, 1, oh, ow, dtype=torch.float32, device=rgb.device)
        _lib.check(L.rdm_nyu_preprocess(_lib.ptr(rgb.contiguous()), _lib.ptr(depth.contiguous()), _lib.ptr(aug), B, H, W, h1, w1, oh, ow,
                                        _lib.ptr(x), _lib.ptr(y), _lib.ptr(self._ws), self._ws.numel(), _lib.stream()))
        return x, y


class PrefetchLoader:
    """Iterate (x, y) training / validation batches: a reader thread fills pinned staging buffers one batch ahead, the
    H2D copies run on a dedicated copy stream, and the augmentation kernels run on the consumer's stream after an event wait."""

    def __init__(self, dataset, batch_size, shuffle=None, seed=0, device="cuda", drop_last=True, rank=0, world=1, workers=0):
        """``workers`` (the reference's ``--worker``, module.py:19-27 DataLoader(num_workers=...)): threads that decode the raw samples of a
        batch in parallel (h5 / npz reads release the GIL); 0 = decode in the reader thread.  The augmentation itself runs on the GPU."""
        self.ds, self.bs, self.device = dataset, batch_size, torch.device(device)
        self.pool = None
        if workers and workers > 0:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=int(workers))
        self.train = dataset.split == "train"
        self.shuffle = self.train if shuffle is None else shuffle
        self.rng = np.random.default_rng(seed + rank)
        self.order_rng = np.random.default_rng(seed)            # same permutation on every rank, disjoint shards
        self.rank, self.world, self.drop_last = rank, world, drop_last
        self.pre = NyuGpuPreprocessor(500 if dataset.split == "test" else dataset.resize, dataset.output_size, device)
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        n = len(self.ds) // self.world
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def _read(self, idx):
        raws = list(self.pool.map(self.ds.get_raw, idx)) if self.pool is not None else [self.ds.get_raw(i) for i in idx]
        H, W = raws[0][1].shape
        rgb = torch.empty(len(idx), H, W, 3, dtype=torch.uint8).pin_memory()
        dep = torch.empty(len(idx), H, W, dtype=torch.float32).pin_memory()
        for j, (r, d) in enumerate(raws):
            rgb[j] = torch.from_numpy(r)
            dep[j] = torch.from_numpy(d)
        if self.train:
            params = [draw_training_params(self.rng, (H, W), self.ds.resize, self.ds.output_size)[1] for _ in idx]
        elif self.ds.split == "test":
            params = [test_params((H, W), self.ds.output_size) for _ in idx]
        else:
            params = [identity_params((H, W), self.ds.resize, self.ds.output_size) for _ in idx]
        return rgb, dep, params

    def __iter__(self):
        order = self.order_rng.permutation(len(self.ds)) if self.shuffle else np.arange(len(self.ds))
        order = order[:len(order) // self.world * self.world][self.rank::self.world]      # equal shards: collectives in the step stay matched
        batches = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        if self.drop_last:
            batches = [b for b in batches if len(b) == self.bs]
        q = queue.Queue(maxsize=2)

        def worker():
            try:
                for b in batches:
                    q.put(self._read(b))
            except BaseException as e:           # surface reader failures in the consumer
                q.put(e)
            q.put(None)

        threading.Thread(target=worker, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            rgb, dep, params = item
            with torch.cuda.stream(self.copy_stream):
                rgb_d, dep_d = rgb.to(self.device, non_blocking=True), dep.to(self.device, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.copy_stream)
            torch.cuda.current_stream().wait_event(done)
            rgb_d.record_stream(torch.cuda.current_stream())
            dep_d.record_stream(torch.cuda.current_stream())
            yield self.pre(rgb_d, dep_d, params)
