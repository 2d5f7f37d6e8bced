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


def _adamw_ref(p, g, m, v, step, scale, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW's update, elementwise on flat ranges (what rdm_adamw_fused computes; module.py:41)."""
    g = g * scale
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)


def _exchange_worker(rank, world, port, q):
    """The three ways of driving exchange + optimiser (VERDICT r4 item 6 a-c) must leave every rank with the SAME parameters:
    monolithic all-reduce then one update; all-reduce with the update per bucket as it lands; reduce-scatter of the gradient bucket ->
    update of the owned shard -> all-gather of the parameter bucket."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from md_rdm_amd.parallel import GradSync
    n = 3000
    slices = [(2000, 3000), (900, 2000), (0, 900)]                   # 1000 / 1100 / 900 floats: shards of 448 / 512 / 448 + remainders 104 / 76 / 4
    gen = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=gen)
    grads = [torch.randn(n, generator=gen) for _ in range(world)]
    gsum = sum(grads)
    ref_p, ref_m, ref_v = p0.clone(), torch.zeros(n), torch.zeros(n)
    _adamw_ref(ref_p, gsum, ref_m, ref_v, 1, 1.0 / world)
    ok, why = True, []

    def check(name, cond):
        nonlocal ok
        if not cond:
            ok = False
            why.append(name)

    for exchange in ("all_reduce", "reduce_scatter"):
        p, g, m, v = p0.clone(), grads[rank].clone(), torch.zeros(n), torch.zeros(n)
        sync = GradSync(g, slices, exchange=exchange, flat_param=p)
        for st in range(3):
            sync.on_stage(st)
        calls = []

        def on_reduced(stage, ranges):
            calls.append((stage, ranges))
            for a, b in ranges:
                _adamw_ref(p[a:b], g[a:b], m[a:b], v[a:b], 1, 1.0 / world)
        scale = sync.finish(on_reduced=on_reduced)
        check(exchange + ": scale", abs(scale - 1.0 / world) < 1e-12)
        check(exchange + ": one call per stage, in issue order", [c[0] for c in calls] == [0, 1, 2])
        if exchange == "all_reduce":
            check("all_reduce: ranges are the whole bucket", [c[1] for c in calls] == [[s] for s in slices])
            check("all_reduce: gradients are sums everywhere", torch.allclose(g, gsum, rtol=0, atol=1e-6))
        else:
            for (stage, ranges), (a, b) in zip(calls, slices):
                a0, main, chunk = sync.shard(stage)
                check("reduce_scatter: shard geometry", a0 == a and chunk % 64 == 0 and main == chunk * world and 0 <= (b - a) - main < world * 64)
                want = [(a + rank * chunk, a + (rank + 1) * chunk)] + ([(a + main, b)] if a + main < b else [])
                check("reduce_scatter: this rank updates its shard + the remainder", ranges == want)
                own = slice(a + rank * chunk, a + (rank + 1) * chunk)
                check("reduce_scatter: owned gradient shard is the sum", torch.allclose(g[own], gsum[own], rtol=0, atol=1e-6))
            # the moments exist for the owned shards only until they are gathered (checkpointing)
            other = 1 - rank
            a, main, chunk = sync.shard(0)
            check("reduce_scatter: foreign shard's moments untouched before the gather", float(m[a + other * chunk:a + (other + 1) * chunk].abs().max()) == 0.0)
            sync.allgather_shards(m)
            sync.allgather_shards(v)
        check(exchange + ": parameters == torch AdamW fed the averaged gradient", torch.allclose(p, ref_p, rtol=0, atol=1e-6))
        check(exchange + ": moments", torch.allclose(m, ref_m, rtol=0, atol=1e-6) and torch.allclose(v, ref_v, rtol=0, atol=1e-6))
        both = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(both, p)
        check(exchange + ": replicas bit-identical after the step", torch.equal(both[0], both[1]))
    # reduce_scatter WITHOUT a per-stage optimiser = the all-reduce spelled as its two halves: gradients are sums everywhere
    g = grads[rank].clone()
    sync = GradSync(g, slices, exchange="reduce_scatter")
    for st in range(3):
        sync.on_stage(st)
    sync.finish()
    check("reduce_scatter + gradient gather: sums everywhere", torch.allclose(g, gsum, rtol=0, atol=1e-6))
    # asynchronous buffer broadcast: issued, then fenced; the refresh callback fires at the fence, once
    fbuf, ibuf, fired = torch.full((12,), float(rank + 1)), torch.full((3,), rank + 7, dtype=torch.int64), []
    sb = GradSync(g, slices, buffers=[fbuf[0:5], ibuf], flat_buffers=[fbuf, ibuf], on_buffers_changed=lambda: fired.append(1))
    sb.sync_buffers(async_op=True)
    check("async broadcast: callback not yet fired", fired == [])
    check("async broadcast: fence reports work", sb.wait_buffers() is True)
    check("async broadcast: values are rank 0's", torch.equal(fbuf, torch.full((12,), 1.0)) and torch.equal(ibuf, torch.full((3,), 7, dtype=torch.int64)))
    check("async broadcast: callback fired once; nothing left in flight", fired == [1] and sb.wait_buffers() is False)
    q.put((rank, ok, why))
    dist.barrier()
    dist.destroy_process_group()


def test_exchanges_and_per_stage_optimiser_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted(q.get(timeout=120) for _ in ps)
    finally:
        for p in ps:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert res == [(0, True, []), (1, True, [])], res


def test_gradsync_rejects_an_unknown_exchange():
    from md_rdm_amd.parallel import GradSync
    with pytest.raises(ValueError):
        GradSync(torch.zeros(4), [(0, 4)], exchange="ring")


def _lr_worker(rank, world, port, q):
    """find_learning_rate under data parallelism: the shard losses of the two ranks diverge at DIFFERENT sweep
    steps.  Every step runs a collective (the stage all-reduces, here one all_reduce in finish()); if one rank left the sweep
    alone, the other would block in it and this worker would never report."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from md_rdm_amd import harness
    from md_rdm_amd.parallel import GradSync

    class Opt:
        lr, m, v = 1e-4, None, None

        class small:
            param_groups = [{"lr": 1e-4}]

        def zero_grad(self): pass
        def step(self, grad_scale=1.0, sync=None): self.scale = sync.finish() if sync is not None else grad_scale
        def state_dict(self): return {"exp_avg": None}
        def load_state_dict(self, sd): pass

    class Model:
        _flat = (torch.zeros(8), torch.zeros(8), [])
        def state_dict(self): return {}
        def mark_weights_changed(self): pass

    opt, steps = Opt(), []
    blow_at = 40 if rank == 0 else 70                               # rank 0's shard diverges 30 steps before rank 1's

    def fake_step(model, x, y):
        i = len(steps)
        steps.append(i)
        return torch.tensor(1.0 / (1 + 0.05 * i) if i < blow_at else 50.0 * (i - blow_at + 1), requires_grad=True), {}
    harness.training_step = fake_step
    sync = GradSync(Model._flat[1], [(0, 8)], group=None)
    real_finish = sync.finish

    def finish():
        sync.on_segment(0)                                           # the per-step gradient exchange every rank must take part in
        return real_finish()
    sync.finish = finish
    lr, lrs, losses = harness.find_learning_rate(Model(), opt, iter([(None, None)] * 100), sync=sync)
    q.put((rank, len(steps), lr, losses[-1]))
    dist.barrier()
    dist.destroy_process_group()


def test_find_learning_rate_stops_collectively_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_lr_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted(q.get(timeout=90) for _ in ps)
    finally:
        for p in ps:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()                                             # a rank blocked in a collective: exactly the failure this test is about
    (r0, n0, lr0, l0), (r1, n1, lr1, l1) = res
    assert (r0, r1) == (0, 1)
    assert n0 == n1 and 40 < n0 < 100                                # same number of steps on both ranks, stopped early
    assert lr0 == lr1 and lr0 is not None and abs(l0 - l1) < 1e-12   # same smoothed (rank-mean) loss, same suggestion


def test_gradsync_rereads_the_models_live_buffers_after_a_reflatten():
    """attach() hands GradSync CALLABLES, so buffers / gradients re-homed by a later flatten_parameters() are the ones
    that get broadcast / reduced - not the storage captured at attach time."""
    from md_rdm_amd.parallel import GradSync

    class M:
        pass
    m = M()
    m.bufs, m.flat_bufs, m.g = [torch.zeros(3)], [torch.zeros(3)], torch.zeros(4)
    s = GradSync(lambda: m.g, [(0, 4)], buffers=lambda: m.bufs, flat_buffers=lambda: m.flat_bufs)
    first = (s.buffers[0], s.flat_buffers[0], s.flat)
    m.bufs, m.flat_bufs, m.g = [torch.ones(3)], [torch.ones(3)], torch.ones(4)          # "re-flattened"
    assert s.buffers[0] is m.bufs[0] and s.flat_buffers[0] is m.flat_bufs[0] and s.flat is m.g
    assert first[0] is not s.buffers[0] and first[2] is not s.flat


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
