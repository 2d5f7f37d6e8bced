"""-m gpu, runs LAST (file name): self-checking coverage of the kernel variants the headline geometry selects."""
import pytest
import torch

import conftest

pytestmark = pytest.mark.gpu
NEED = {"test_gpu_conv": 4 + 4 + 8 + 2 + 4, "test_gpu_wino": 6 + 6 + 6, "test_gpu_xsplit": 6 + 5 + 5 + 4 + 4}          # headline-size parity cases each operator-level module must have run


def test_every_headline_kernel_variant_was_launched_by_a_parity_test():
    """Run one B=16 228x304 train step (the bench geometry) with the census on and collect the kernel VARIANTS the plan selects (tile
    shapes, halo lengths, epilogues, Winograd or direct, split or not); every one of them must have been launched by an oracle
    comparison of tests/test_gpu_conv.py, tests/test_gpu_wino.py or tests/test_gpu_xsplit.py (each module's census is snapshotted when it
    ends; the launches of whole-network tests do not count).  Both backward modes are stepped: the shipped one (split-precision bf16x3
    gradient kernels in dense_e2 / dense_e3) and exact f32.  Skipped when those modules were only run in part."""
    from md_rdm_amd import _lib, filler, harness
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    for mod, need in NEED.items():
        if len(conftest.RAN.get(mod, ())) < need or mod not in conftest.CENSUS:
            pytest.skip(f"{mod}: only {len(conftest.RAN.get(mod, ()))} of {need} headline parity cases ran in this process")
    by_tests = {}
    for mod in NEED:
        for k, v in conftest.CENSUS[mod].items():
            by_tests[k] = by_tests.get(k, 0) + v
    L = _lib.lib()
    L.rdm_census_reset()
    L.rdm_census_enable(1)
    try:
        dev = torch.device("cuda:0")
        m = DepthEstimationNet()
        filler.fill_state_dict(m.state_dict())
        m = m.to(dev).train()
        x, y = filler.synthetic_batch(16, 228, 304, seed=1234)
        for mode in ("bf16x3", "f32"):
            m.backward_precision = mode
            m.forward_split = mode == "bf16x3"          # the shipped configuration, then exact-f32 MFMA everywhere
            loss, _ = harness.training_step(m, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
            loss.backward()
            torch.cuda.synchronize()
        step = _lib.census()
    finally:
        L.rdm_census_enable(0)
    assert len(step) >= 10 and any("px256" in k for k in step) and any(
        k.startswith("conv1x1_dma256_kernel") for k in step) and any(k.startswith("conv3x3_wino_fwd_kernel") for k in step) and any(k.startswith("conv3x3_wino_wgrad_kernel") for k in step), sorted(step)
    for name in ("xs_wgrad1x1_kernel", "xs_dgrad3x3_kernel", "xs_dgrad1x1_kernel", "xs_wgrad3x3_kernel", "xs_fwd1x1_kernel"):
        assert any(k.startswith(name) for k in step), (name, sorted(step))
    missing = sorted(k for k in step if by_tests.get(k, 0) == 0)
    assert not missing, "kernel variants of the headline step that no operator-level parity test launched: %r" % missing
