"""Data parallelism for the one exchange step of the path: the gradient all-reduce that Lightning's
DDP performs implicitly in the reference (train.py:55 ``gpus=N``).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  The conv-stack gradients live in ONE flat buffer.
The native backward runs in 13 STAGES of consecutive layers (rdm_net_backward_stage: 12-37 MB of gradients each, the size class
of DDP's 25 MB buckets, last layers first); after each stage its contiguous slice is all-reduced asynchronously - RCCL runs on
its own HIP stream, fenced by events against the compute stream - so the reduction of stage k overlaps the backward of stage
k+1 and only the LAST exchange (12 MB: conv_e1 + the first two dense_e2 layers) is exposed.  xGMI is point-to-point (7 links
per GPU): tens-of-MB messages keep every link busy; smaller buckets would be latency bound.
Sums, not means, travel: the optimiser applies 1/world (``grad_scale``), saving a pass over the 362 MB.

BatchNorm buffers: statistics are per-rank in training (local batch, as the reference: no SyncBN).  The reference's DDP
re-broadcasts module buffers from rank 0 before every forward (torch DDP ``broadcast_buffers=True``), i.e. every rank normalises
its running_mean / running_var history to rank 0's.  ``GradSync.sync_buffers()`` does exactly that broadcast in one call and
``attach(..., broadcast_buffers=True)`` (default) runs it at the start of every training forward through a forward pre-hook;
pass False to keep purely local running statistics.
The class is independent of the model so that the N>1 logic is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, flat_grad, slices, extra=(), group=None, buffers=(), flat_buffers=None, on_buffers_changed=None):
        """``flat_grad`` / ``buffers`` / ``flat_buffers``: tensors / tensor lists, or CALLABLES returning them - ``attach`` passes callables so that the
        broadcast always sees the module's LIVE buffers (``flatten_parameters`` re-homes them after ``.to()`` / a re-flatten; a list
        captured once would go on broadcasting the orphaned storage).  ``on_buffers_changed``: called after every buffer broadcast
        (derived data such as the bf16 BatchNorm affines must be refreshed)."""
        self._flat, self.slices, self.extra, self.group = flat_grad, list(slices), list(extra), group
        self._buffers = buffers if callable(buffers) else [b for b in buffers if b.numel()]
        self._flat_buffers = flat_buffers if callable(flat_buffers) else (list(flat_buffers) if flat_buffers else None)
        self.on_buffers_changed = on_buffers_changed
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.handles = []
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
        self.handles.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    on_stage = on_segment

    def sync_buffers(self):
        """torch DDP's ``broadcast_buffers``: every rank takes rank 0's BatchNorm running statistics / counters."""
        buffers = self.buffers
        if self.world == 1 or not buffers:
            return
        flat_buffers = self.flat_buffers
        if flat_buffers is not None:                                 # the model keeps its buffers in flat storage: broadcast in place
            for buf in flat_buffers:
                dist.broadcast(buf, 0, group=self.group)
            if self.on_buffers_changed is not None:
                self.on_buffers_changed()
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

    def finish(self):
        """Wait for all reductions (and reduce the small extra tensors).  Gradients are SUMS;
        the optimiser applies 1/world (``grad_scale``), saving a pass over the buffer."""
        if self.world == 1:
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
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        return 1.0 / self.world


def attach(model, group=None, broadcast_buffers=True):
    """Wire a flattened DepthEstimationNet to a GradSync (direct-gradient fast path)."""
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
                    on_buffers_changed=getattr(model, "mark_weights_changed", None))
    model.direct_grads = True
    # one process: no exchange, so no per-stage consumer - the plan then joins its weight-gradient stream once per segment, not per stage
    model.grad_ready_hook = sync.on_stage if sync.world > 1 else None
    if broadcast_buffers and sync.world > 1:
        def _pre(module, args):
            if module.training:
                sync.sync_buffers()
        model._dp_buffer_hook = model.register_forward_pre_hook(_pre)
    return sync
