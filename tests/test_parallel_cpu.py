"""The N>1 path (segment-wise asynchronous gradient all-reduce) on CPU: world_size-2 gloo."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from md_rdm_amd.parallel import GradSync
    n = 1000
    slices = [(700, 1000), (400, 700), (100, 400), (0, 100)]       # backward order: decoder first
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    w = torch.nn.Parameter(torch.ones(1, 1))
    w.grad = torch.full((1, 1), float(rank + 1))
    sync = GradSync(flat, slices, extra=[w])
    for seg in range(4):
        sync.on_segment(seg)
    scale = sync.finish()
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(flat, want) and abs(scale - 1.0 / world) < 1e-12 and abs(w.grad.item() - (1 + world) / 2) < 1e-6
    # torch DDP's broadcast_buffers: BatchNorm running statistics and counters follow rank 0
    rm, nbt = torch.full((7,), float(rank + 3)), torch.tensor(rank + 5, dtype=torch.int64)
    sb = GradSync(flat, slices, buffers=[rm, nbt, torch.zeros(0)])
    sb.sync_buffers()
    ok = ok and torch.equal(rm, torch.full((7,), 3.0)) and int(nbt) == 5 and nbt.dtype == torch.int64
    # the same with the buffers kept as views of flat storage (DepthEstimationNet._flatten_buffers): broadcast in place, no copies
    fbuf, ibuf = torch.full((12,), float(rank + 1)), torch.full((3,), rank + 7, dtype=torch.int64)
    views = [fbuf[0:5], fbuf[5:12].view(7), ibuf[0:1].view(()), ibuf[1:3]]
    sf = GradSync(flat, slices, buffers=views, flat_buffers=[fbuf, ibuf])
    sf.sync_buffers()
    ok = ok and all(torch.equal(v, torch.full_like(v, 1.0 if v.dtype == torch.float32 else 7)) for v in views)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_gradsync_single_process_is_noop():
    from md_rdm_amd.parallel import GradSync
    flat = torch.ones(10)
    s = GradSync(flat, [(0, 10)] * 4)
    for seg in range(4):
        s.on_segment(seg)
    assert s.finish() == 1.0 and torch.equal(flat, torch.ones(10))


def test_backward_stages_partition_the_stack_into_ddp_sized_buckets():
    """rdm_net_backward_stage: the stages tile the conv stack's tensors exactly once, last registered tensor first, in buckets of
    the size class of DDP's 25 MB default (train.py:55) with a small LAST exchange."""
    import ctypes as C
    from md_rdm_amd import _lib
    L = _lib.lib()
    n = L.rdm_net_num_backward_stages()
    assert 10 <= n <= 20
    nxt, sizes = None, []
    for s in range(n):
        a, b = C.c_int32(), C.c_int32()
        assert L.rdm_net_backward_stage_range(s, C.byref(a), C.byref(b)) == 0
        assert a.value <= b.value and (nxt is None or b.value == nxt - 1)         # contiguous, walking backwards
        nxt = a.value
        sizes.append(4 * sum(L.rdm_net_tensor_numel(i) for i in range(a.value, b.value + 1) if L.rdm_net_tensor_is_param(i)))
    assert nxt == 0                                                               # ... down to encoder.conv_e1.weight
    assert L.rdm_net_tensor_name(L.rdm_net_num_tensors() - 9).decode() == "d_1.conv2.bias"   # stage 0 ends at the last stack tensor (8 weight_layer entries follow)
    assert sum(sizes) == 4 * (90529721 - 4)                                       # every stack parameter exactly once (the 4 Weights scalars are exchanged apart)
    assert all(10e6 < z < 40e6 for z in sizes) and sizes[-1] < 15e6               # the exposed last exchange is the smallest
    assert L.rdm_net_backward_stage_range(n, C.byref(a), C.byref(b)) != 0
