"""-m gpu: WSM decoder block (RDM_Net.py:163-236) on the conv family vs fixtures produced by running
the reference's WSMLayer (reduced widths; channel counts that are NOT multiples of 16 exercise the
zero-padding path).  Tolerance 2e-5 of the tensor's max (f32 accumulation order)."""
import numpy as np
import pytest
import torch

from md_rdm_amd import filler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,k,lid,raw,hw", [(64, 16, 2, 128, 8), (32, 32, 3, 64, 16)])
def test_wsm_layer_forward_backward(wsm_gold, cin, k, lid, raw, hw):
    from md_rdm_amd.network.wsm import WSMLayer
    dev = torch.device("cuda:0")
    m = WSMLayer(cin, k, k, lid)
    sd = m.state_dict()
    assert [f"{k_} {tuple(v.shape)}" for k_, v in sd.items()] == [str(s) for s in wsm_gold[f"wsm{lid}_keys"]]
    for key, t in sd.items():
        t.copy_(torch.from_numpy(filler.state_value(f"wsm{lid}." + key, tuple(t.shape))))
    m = m.to(dev)
    x = torch.from_numpy(filler.uniform(f"wsm{lid}.x", (2, raw, hw, hw), -1.0, 1.0)).to(dev).requires_grad_(True)
    y = m(x)
    ref = wsm_gold[f"wsm{lid}_out"]
    assert y.shape == ref.shape
    assert np.abs(y.detach().cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()
    gy = torch.from_numpy(filler.uniform(f"wsm{lid}.gy", tuple(y.shape), -1.0, 1.0)).to(dev)
    (y * gy).sum().backward()
    dx = x.grad.detach().cpu()
    st = np.array([dx.double().mean().item(), dx.double().abs().mean().item(), dx.double().abs().max().item()])
    np.testing.assert_allclose(st, wsm_gold[f"wsm{lid}_dx_stats"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(dx[0, :4, :2, :2].numpy(), wsm_gold[f"wsm{lid}_dx_head"], rtol=2e-4, atol=2e-5 * st[2])
    for name, p in m.named_parameters():
        want = float(wsm_gold[f"wsm{lid}_gradnorm__{name}"])
        assert abs(p.grad.double().norm().item() - want) <= 2e-4 * want + 1e-9, name
