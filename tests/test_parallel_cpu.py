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
