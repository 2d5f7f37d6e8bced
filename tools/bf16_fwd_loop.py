"""Run a few bf16 forwards (for rocprofv3 --kernel-trace / --pmc of BASELINE config 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import filler
from md_rdm_amd.network.RDM_Net import DepthEstimationNet
B = int(os.environ.get("B", "8")); n = int(os.environ.get("N", "3"))
dev = torch.device("cuda:0")
m = DepthEstimationNet(); filler.fill_state_dict(m.state_dict()); m = m.to(dev).eval().set_precision("bf16")
x, _ = filler.synthetic_batch(B, 228, 304, seed=1234)
xg = torch.from_numpy(x).to(dev)
with torch.no_grad():
    for _ in range(n):
        m._native_forward_bf16(xg)
torch.cuda.synchronize()
print("done")
