"""Which Python lines of a training step cause host <-> device copies (each one blocks the host until the stream has drained - torch's memcpy_and_sync ->
hipMemcpyWithStream - and costs the host its lead over the GPU)?  torch.profiler with stacks, filtered to the copy / scalar-extraction ops.
    python tools/sync_debug.py"""
import sys, collections, torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
from md_rdm_amd import filler, harness
from md_rdm_amd.network.RDM_Net import DepthEstimationNet
dev = torch.device("cuda:0")
m = DepthEstimationNet(); filler.fill_state_dict(m.state_dict()); m = m.to(dev).train()
x, y = filler.synthetic_batch(16, 228, 304, seed=1234)
xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
m.flatten_parameters()
opt = harness.FusedAdamW(m, lr=1e-4)
def step():
    opt.zero_grad()
    loss, _ = harness.training_step(m, xg, yg)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
names = collections.Counter(e.name for e in ev)
print({k: v for k, v in names.items() if "emcpy" in k or "item" in k or "local_scalar" in k or "_to_copy" in k or k in ("aten::copy_", "aten::to", "aten::tensor", "aten::lift_fresh", "aten::scalar_tensor", "aten::fill_")})
sites = collections.Counter()
for e in ev:
    if e.name in ("aten::_local_scalar_dense", "aten::item", "aten::_to_copy", "aten::copy_", "aten::scalar_tensor", "aten::tensor", "aten::lift_fresh"):
        st = [s for s in (e.stack or []) if "md_rdm_amd" in s or "bench" in s or "optim" in s]
        shapes = str(e.input_shapes)[:40]
        sites[(e.name, shapes, " <- ".join(s.split("/")[-1][:60] for s in st[:3]))] += 1
for k, v in sites.most_common(40): print(v, k)
