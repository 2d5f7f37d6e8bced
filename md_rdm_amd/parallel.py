"""Data parallelism for the one exchange step of the path: the gradient all-reduce that Lightning's
DDP performs implicitly in the reference (train.py:55 ``gpus=N``).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  The conv-stack gradients live in ONE flat buffer.
The native backward runs in 13 STAGES of consecutive layers (rdm_net_backward_stage: 12-37 MB of gradients each, the size class
of DDP's 25 MB buckets, last layers first); after each stage its contiguous slice is all-reduced asynchronously - RCCL runs on
its own HIP stream, fenced by events against the compute stream - so the reduction of stage k overlaps the backward of stage
k+1 and only the LAST exchange (12 MB: conv_e1 + the first two dense_e2 layers) is exposed.  xGMI is point-to-point (7 links
per GPU): tens-of-MB messages keep every link busy; smaller buckets would be latency bound.
Sums, not means, travel: the optimiser applies 1/world (``grad_scale``), saving a pass over the 362 MB.

Two exchanges (``GradSync(exchange=...)``, ``attach(model, exchange=...)``):
  "all_reduce"      one asynchronous all-reduce per stage (the default).
  "reduce_scatter"  per stage: reduce-scatter of the GRADIENT bucket (every rank ends up owning the sum of one 1/N shard), the optimiser
                    updates only that shard (1/N of the AdamW work and of its 28 B/param of HBM traffic), all-gather of the updated
                    PARAMETER bucket.  Same bytes on the wire as the ring all-reduce, but driven as two collectives that keep all 7 xGMI
                    links of a GPU busy at once.  A bucket is cut into N equal shards of a multiple of 64 floats; the < N*64-float
                    remainder is all-reduced and updated by every rank.  Without a per-stage optimiser (plain ``finish()``) the second
                    half gathers the reduced GRADIENTS instead, i.e. the all-reduce spelled as its two halves.
The optimiser runs PER BUCKET as its reduction lands (``finish(on_reduced=...)``, ``FusedAdamW.step(sync=...)``): stage k's wait is a
stream dependency, its AdamW launch follows at once, so the updates of stages 0 .. n-2 run under the reductions still in flight and only
the last (12 MB) bucket's update is exposed.

BatchNorm buffers: statistics are per-rank in training (local batch, as the reference: no SyncBN).  The reference's DDP
re-broadcasts module buffers from rank 0 before every forward (torch DDP ``broadcast_buffers=True``), i.e. every rank normalises
its running_mean / running_var history to rank 0's.  ``GradSync.sync_buffers()`` does exactly that broadcast in one call and
``attach(..., broadcast_buffers=True)`` (default) runs it at the start of every training forward through a forward pre-hook;
pass False to keep purely local running statistics.  The broadcast is ASYNCHRONOUS: it is issued right after a training forward has
been enqueued (rank 0's statistics are final from then until the next training forward), runs on the communication stream under the whole
backward pass, and the next training forward only waits for its completion - an event fence in front of the first BatchNorm finalisation,
no blocking collective at the head of the forward (the first step, with nothing in flight yet, broadcasts synchronously).
The class is independent of the model so that the N>1 logic is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist


class GradSync:
    ALIGN = 64                   # floats: shards of the reduce-scatter exchange start on the flat buffer's 256-byte tensor alignment

    def __init__(self, flat_grad, slices, extra=(), group=None, buffers=(), flat_buffers=None, on_buffers_changed=None, exchange="all_reduce", flat_param=None):
        """``flat_grad`` / ``buffers`` / ``flat_buffers``: tensors / tensor lists, or CALLABLES returning them - ``attach`` passes callables so that the
        broadcast always sees the module's LIVE buffers (``flatten_parameters`` re-homes them after ``.to()`` / a re-flatten; a list
        captured once would go on broadcasting the orphaned storage).  ``on_buffers_changed``: called after every buffer broadcast
        (derived data such as the bf16 BatchNorm affines must be refreshed)."""
        self._flat, self.slices, self.extra, self.group = flat_grad, list(slices), list(extra), group
        self._buffers = buffers if callable(buffers) else [b for b in buffers if b.numel()]
        self._flat_buffers = flat_buffers if callable(flat_buffers) else (list(flat_buffers) if flat_buffers else None)
        self.on_buffers_changed = on_buffers_changed
        if exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"GradSync: unknown exchange {exchange!r} (all_reduce | reduce_scatter)")
        self.exchange, self._param = exchange, flat_param
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.handles = []
        self._buffer_work = []       # asynchronous buffer broadcasts in flight (waited for at the head of the next training forward)
        self.timing = False          # True: finish() brackets its waits with a device event pair (bench.py: exposed communication time)
        self._events = []

    def bytes_per_step(self):
        """Payload of one step's exchange: the float32 gradient slices of all stages + the small extra tensors."""
        return 4 * (sum(b - a for a, b in self.slices) + sum(t.numel() for t in self.extra))

    def exposed_ms(self):
        """Mean device time between the point where the compute stream reaches finish() and the completion of every reduction (the
        communication the backward pass did NOT hide), over the steps run with ``timing`` on; None if there were none."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._events]
        self._events.clear()
        return sum(ms) / len(ms)

    @property
    def flat(self):
        return self._flat() if callable(self._flat) else self._flat

    @property
    def flat_param(self):
        return self._param() if callable(self._param) else self._param

    def shard(self, stage):
        """Reduce-scatter geometry of a stage's bucket [a, b): (a, main, chunk) - ``main`` = the leading part that splits into ``world``
        equal shards of ``chunk`` floats (a multiple of ALIGN); rank r owns [a + r chunk, a + (r + 1) chunk); [a + main, b) is the
        remainder every rank reduces and updates."""
        a, b = self.slices[stage]
        unit = self.world * self.ALIGN
        main = (b - a) // unit * unit
        return a, main, main // self.world

    def owned_ranges(self, stage):
        """Element ranges of the flat buffers THIS rank's optimiser must update for a stage: everything under "all_reduce"; its own
        shard + the small remainder under "reduce_scatter"."""
        a, b = self.slices[stage]
        if self.world == 1 or self.exchange == "all_reduce":
            return [(a, b)]
        a, main, chunk = self.shard(stage)
        res = [(a + self.rank * chunk, a + (self.rank + 1) * chunk)] if chunk else []
        if a + main < b:
            res.append((a + main, b))
        return res

    @property
    def buffers(self):
        return [b for b in self._buffers() if b.numel()] if callable(self._buffers) else self._buffers

    @property
    def flat_buffers(self):
        fb = self._flat_buffers() if callable(self._flat_buffers) else self._flat_buffers
        return list(fb) if fb else None

    def mean_scalar(self, value):
        """Mean over the ranks of a host scalar (one small all-reduce): every rank gets the SAME number, so control decisions taken
        on it (the learning-rate finder's early stop) are taken together."""
        if self.world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / self.world

    def on_segment(self, stage):
        """Call right after backward stage ``stage`` has been enqueued."""
        if self.world == 1:
            return
        a, b = self.slices[stage]
        flat = self.flat
        if self.exchange == "all_reduce":
            self.handles.append((stage, [dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)]))
            return
        a, main, chunk = self.shard(stage)
        work = []
        if chunk:                    # in place: the output is this rank's shard of the input (the in-place form of NCCL / RCCL)
            work.append(dist.reduce_scatter_tensor(flat[a + self.rank * chunk:a + (self.rank + 1) * chunk], flat[a:a + main], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if a + main < b:
            work.append(dist.all_reduce(flat[a + main:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.handles.append((stage, work))

    on_stage = on_segment

    def _gather(self, tensor, stage):
        """Second half of the reduce-scatter exchange: every rank's shard of ``tensor``'s bucket -> all ranks (asynchronous)."""
        a, main, chunk = self.shard(stage)
        if not chunk:
            return None
        return dist.all_gather_into_tensor(tensor[a:a + main], tensor[a + self.rank * chunk:a + (self.rank + 1) * chunk], group=self.group, async_op=True)

    def allgather_shards(self, tensor):
        """A flat tensor whose owned shards only are valid on each rank (the optimiser's moments under "reduce_scatter") -> valid everywhere
        (checkpointing).  No-op under "all_reduce" / one process."""
        if self.world == 1 or self.exchange == "all_reduce":
            return tensor
        for stage in range(len(self.slices)):
            w = self._gather(tensor, stage)
            if w is not None:
                w.wait()
        return tensor

    def wait_buffers(self):
        """Fence: the calling stream waits for the buffer broadcasts in flight (no host block on RCCL: ``Work.wait`` is a stream
        dependency).  Returns True if there were any."""
        if not self._buffer_work:
            return False
        for w in self._buffer_work:
            w.wait()
        self._buffer_work.clear()
        if self.on_buffers_changed is not None:
            self.on_buffers_changed()
        return True

    def sync_buffers(self, async_op=False):
        """torch DDP's ``broadcast_buffers``: every rank takes rank 0's BatchNorm running statistics / counters.  ``async_op``: issue the
        broadcasts on the communication stream and return; ``wait_buffers()`` fences them (flat buffer storage only - the copy-back of
        the unflattened form needs the data)."""
        buffers = self.buffers
        if self.world == 1 or not buffers:
            return
        self.wait_buffers()                                          # never two generations in flight
        flat_buffers = self.flat_buffers
        if flat_buffers is not None:                                 # the model keeps its buffers in flat storage: broadcast in place
            work = [dist.broadcast(buf, 0, group=self.group, async_op=True) for buf in flat_buffers]
            self._buffer_work.extend(work)
            if not async_op:
                self.wait_buffers()
            return
        floats = [b for b in buffers if b.dtype == torch.float32]
        if floats:                                                   # one coalesced broadcast instead of ~480 tiny ones
            buf = torch.cat([b.reshape(-1) for b in floats])
            dist.broadcast(buf, 0, group=self.group)
            o = 0
            for b in floats:
                b.copy_(buf[o:o + b.numel()].view_as(b))
                o += b.numel()
        others = [b for b in buffers if b.dtype != torch.float32]
        if others:
            buf = torch.cat([b.reshape(-1).to(torch.int64) for b in others])
            dist.broadcast(buf, 0, group=self.group)
            o = 0
            for b in others:
                b.copy_(buf[o:o + b.numel()].view_as(b).to(b.dtype))
                o += b.numel()
        if self.on_buffers_changed is not None:
            self.on_buffers_changed()

    def finish(self, on_reduced=None):
        """Wait for all reductions (and reduce the small extra tensors).  Gradients are SUMS; the optimiser applies 1/world (the returned
        ``grad_scale``), saving a pass over the buffer.
        ``on_reduced(stage, ranges)``: the per-bucket optimiser - called for every stage, in the order the reductions were issued, as soon
        as that stage's reduction is fenced; ``ranges`` = ``owned_ranges(stage)``.  Under "reduce_scatter" the stage's updated PARAMETER
        shards are all-gathered right after the call (needs ``flat_param``); without a callback the reduced GRADIENT shards are."""
        if self.world == 1:
            if on_reduced is not None:
                for stage in range(len(self.slices)):
                    on_reduced(stage, self.owned_ranges(stage))
            return 1.0
        ev = None
        if self.timing and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        small = [t.grad for t in self.extra if t.grad is not None]
        if small:
            buf = torch.cat([g.reshape(-1) for g in small])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            o = 0
            for g in small:
                g.copy_((buf[o:o + g.numel()] / self.world).view_as(g))
                o += g.numel()
        gathers = []
        for stage, work in self.handles:
            for w in work:
                w.wait()
            if on_reduced is not None:
                on_reduced(stage, self.owned_ranges(stage))
            if self.exchange == "reduce_scatter":
                target = self.flat_param if on_reduced is not None else self.flat
                if target is None:
                    raise RuntimeError("GradSync(exchange='reduce_scatter') with a per-stage optimiser needs flat_param (the parameter buffer whose shards are gathered)")
                g = self._gather(target, stage)
                if g is not None:
                    gathers.append(g)
        for g in gathers:
            g.wait()
        self.handles.clear()
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        return 1.0 / self.world


def attach(model, group=None, broadcast_buffers=True, exchange=None):
    """Wire a flattened DepthEstimationNet to a GradSync (direct-gradient fast path).  ``exchange``: "all_reduce" (default) or
    "reduce_scatter" (see the module text); None reads RDM_DP_EXCHANGE."""
    import os
    exchange = exchange or os.environ.get("RDM_DP_EXCHANGE", "all_reduce")
    flat, gflat = model._flat[0], model._flat[1]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, 0, group=group)                 # identical replicas
        for b in (getattr(model, "_flat_buffers", None) or [b for b in model.buffers() if b.numel()]):
            dist.broadcast(b, 0, group=group)
        for p in model.weight_layer.parameters():
            if p.numel():
                dist.broadcast(p.data, 0, group=group)
    sync = GradSync(lambda: model._flat[1], model.stage_slices(), extra=[p for p in model.weight_layer.parameters() if p.requires_grad], group=group,
                    buffers=lambda: list(model.buffers()), flat_buffers=lambda: getattr(model, "_flat_buffers", None),
                    on_buffers_changed=getattr(model, "mark_weights_changed", None), exchange=exchange, flat_param=lambda: model._flat[0])
    model.direct_grads = True
    # one process: no exchange, so no per-stage consumer - the plan then joins its weight-gradient stream once per segment, not per stage
    model.grad_ready_hook = sync.on_stage if sync.world > 1 else None
    if broadcast_buffers and sync.world > 1:
        def _pre(module, args):                                       # fence of the broadcast issued after the previous training forward;
            if module.training and not sync.wait_buffers():          # nothing in flight (first step, or eval forwards since): broadcast now
                sync.sync_buffers()

        def _post(module, args, output):                             # rank 0's statistics are final until the next training forward:
            if module.training:                                      # their broadcast runs under the backward pass
                sync.sync_buffers(async_op=True)
        model._dp_buffer_hook = model.register_forward_pre_hook(_pre)
        model._dp_buffer_post_hook = model.register_forward_hook(_post)
    return sync
