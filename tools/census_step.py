"""Kernel-variant census of one train step (rdm_census_*): `python tools/census_step.py [B H W]` prints {variant: launches} as JSON."""
import json, sys, torch
sys.path.insert(0, ".")
from md_rdm_amd import _lib, filler, harness
from md_rdm_amd.network.RDM_Net import DepthEstimationNet
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 228, 304)
L = _lib.lib()
m = DepthEstimationNet(); filler.fill_state_dict(m.state_dict()); m = m.cuda().train()
x, y = filler.synthetic_batch(B, H, W, seed=1234)
L.rdm_census_reset(); L.rdm_census_enable(1)
loss, _ = harness.training_step(m, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()); loss.backward(); torch.cuda.synchronize()
print(json.dumps(_lib.census(), indent=1, sort_keys=True))
