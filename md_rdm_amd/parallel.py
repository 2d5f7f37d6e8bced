"""Data parallelism for the one exchange step of the path: the gradient all-reduce that Lightning's
DDP performs implicitly in the reference (train.py:55 ``gpus=N``).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  The conv-stack
gradients live in ONE flat buffer; the native backward runs in 4 segments (decoder, e4, e3, e2+stem)
and after each one the segment's contiguous slice is all-reduced asynchronously - RCCL runs on its
own HIP stream (fenced by events against the compute stream), so the reduction of segment k overlaps
the backward of segment k+1.  xGMI is point-to-point: few, large (tens of MB) collectives drive all
7 links; the 4 slices are 60-150 MB each at fp32.
The class is independent of the model so that the N>1 logic is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, flat_grad, slices, extra=(), group=None):
        self.flat, self.slices, self.extra, self.group = flat_grad, list(slices), list(extra), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.handles = []

    def on_segment(self, seg):
        """Call right after backward segment ``seg`` has been enqueued."""
        if self.world == 1:
            return
        a, b = self.slices[seg]
        self.handles.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for all reductions (and reduce the small extra tensors).  Gradients are SUMS;
        the optimiser applies 1/world (``grad_scale``), saving a pass over the buffer."""
        if self.world == 1:
            return 1.0
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
        return 1.0 / self.world


def attach(model, group=None):
    """Wire a flattened DepthEstimationNet to a GradSync (direct-gradient fast path)."""
    flat, gflat = model._flat[0], model._flat[1]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, 0, group=group)                 # identical replicas
        for b in model.buffers():
            if b.numel():
                dist.broadcast(b, 0, group=group)
        for p in model.weight_layer.parameters():
            if p.numel():
                dist.broadcast(p.data, 0, group=group)
    sync = GradSync(gflat, model.segment_slices(), extra=[p for p in model.weight_layer.parameters() if p.requires_grad], group=group)
    model.direct_grads = True
    model.grad_ready_hook = sync.on_segment
    return sync
