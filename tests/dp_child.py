"""Helper of tests/test_gpu_dp.py (not a test module): ONE data-parallel rank in a fresh process.
    python tests/dp_child.py RANK WORLD PORT OUTDIR BACKEND [MODE]
MODE: monolithic (finish(), then one AdamW over the whole buffer) | per_stage (AdamW per bucket as its all-reduce lands) |
reduce_scatter (reduce-scatter of the gradient bucket, AdamW on the owned shard, all-gather of the parameter bucket).
Runs attach() + one real training step on its own shard and dumps what the parent needs to check the DP parity definition of
SURVEY.md 8(e).  Started before this process touches the GPU; 'gloo' lets two ranks share the one GPU of a test box (RCCL refuses
two ranks on one device), on a multi-GPU node the same script runs with 'nccl' (= RCCL)."""
import os
import sys

rank, world, port, outdir, backend = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
mode = sys.argv[6] if len(sys.argv) > 6 else "monolithic"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import time
T0 = time.time()
import numpy as np
import torch
import torch.distributed as dist


def note(msg):
    print(f"[rank {rank} +{time.time() - T0:6.1f}s] {msg}", flush=True)


note("torch imported")

ndev = torch.cuda.device_count()
dev = torch.device("cuda", rank % max(ndev, 1))
if backend == "nccl":
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
torch.cuda.set_device(dev)

from md_rdm_amd import filler, harness, parallel
from md_rdm_amd.network.RDM_Net import DepthEstimationNet

model = DepthEstimationNet()
filler.fill_state_dict(model.state_dict())
if rank == 1:                       # replicas must start identical whatever the ranks hold: attach() broadcasts rank 0's state
    with torch.no_grad():
        for p in model.parameters():
            if p.numel():
                p.add_(0.01)
        for n, b in model.named_buffers():
            if b.dtype == torch.float32 and b.numel():
                b.add_(0.05)
note("process group up, model built")
model = model.to(dev).train()
model.flatten_parameters()
sync = parallel.attach(model, exchange="reduce_scatter" if mode == "reduce_scatter" else "all_reduce")
note("attached (broadcast done)")
opt = harness.FusedAdamW(model, lr=1e-4)
flat, gflat, _ = model._flat
init = flat.detach().clone()
x, y = filler.synthetic_batch(2, 228, 228, seed=(7, 14)[rank])      # this rank's shard
opt.zero_grad()
loss, parts = harness.training_step(model, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
loss.backward()
note("backward enqueued")
if mode == "monolithic":
    scale = sync.finish()
    note("gradients reduced")
    gsum = gflat.detach().clone()                                     # all-reduced SUM over the ranks
    small_g = {n: p.grad.detach().clone() for n, p in model.weight_layer.named_parameters() if p.grad is not None}   # already averaged by finish()
    opt.step(grad_scale=scale)
else:
    scale = 1.0 / world
    opt.step(sync=sync)                                               # per bucket: wait -> AdamW (-> all-gather of the parameter shards)
    note("reduced + stepped per stage (%s; optimiser: %s)" % (sync.exchange, opt.mode))
    sync.allgather_shards(gflat)                                      # reduce_scatter left each rank the sums of ITS shards: complete them for the parent's checks
    gsum = gflat.detach().clone()
    small_g = {n: p.grad.detach().clone() for n, p in model.weight_layer.named_parameters() if p.grad is not None}
torch.cuda.synchronize()
note("step done")
sd = model.state_dict()
np.savez(os.path.join(outdir, f"rank{rank}.npz"), init=init.cpu().numpy(), gsum=gsum.cpu().numpy(), post=flat.detach().cpu().numpy(), scale=np.array(scale),
         loss=np.array(loss.item()), n_slices=np.array(len(sync.slices)), n_handles_used=np.array(len(model.stage_slices())),
         rm=sd["encoder.dense_e3.denselayer5.norm2.running_mean"].cpu().numpy(), nbt=sd["encoder.dense_e3.denselayer5.norm2.num_batches_tracked"].cpu().numpy(),
         **{"wl_" + n: g.cpu().numpy() for n, g in small_g.items()},
         **{"wlp_" + n: p.detach().cpu().numpy() for n, p in model.weight_layer.named_parameters() if p.numel()})
note("results written")
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "done", flush=True)
