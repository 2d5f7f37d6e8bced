"""-m gpu: the data-parallel path as a SYSTEM (the reference's implicit DDP, train.py:55): two fresh child ranks
(tests/dp_child.py, started before they touch the GPU) run attach() + one real training step on DIFFERENT shards; the parent then
checks the DP parity definition:
  * replicas: rank 1 starts from perturbed weights / buffers and must leave the step bit-identical to rank 0 (attach broadcast +
    identical reduced gradients + identical optimiser);
  * exchange: the all-reduced gradient equals the sum of the two per-shard gradients a single process computes on the same shards
    (per stage slice, to float32 noise: the kernels accumulate with atomics, so two evaluations differ by rounding);
  * update: the post-step weights equal torch.optim.AdamW fed (sum / world) - exactly what "a single-process run fed the averaged
    gradients" produces - to 1 ulp-level tolerance, d_1.conv1.* untouched;
  * buffers: BatchNorm running statistics follow rank 0's shard (torch DDP broadcast_buffers semantics), not an average.
The box has ONE GPU: the ranks exchange over gloo (RCCL refuses two ranks on one device); with >= 2 GPUs the same test uses nccl."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from md_rdm_amd import filler
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_train_step_matches_single_process_with_averaged_gradients(tmp_path):
    assert torch.cuda.is_available()
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = str(29600 + os.getpid() % 1500)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")      # the child ranks (tests/dp_child.py) share the card: dmabuf IPC, as bench.py / train.py set for themselves
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_child.py"), str(r), "2", port, str(tmp_path), backend],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    print("\n".join(o[-1500:] for o in outs))                        # phase timings of the ranks (pytest -s)
    r0, r1 = ({k: v for k, v in np.load(tmp_path / f"rank{r}.npz").items()} for r in range(2))     # materialise once: NpzFile re-reads per access
    assert int(r0["n_slices"]) == 13 and float(r0["scale"]) == 0.5
    # ---- replicas stay identical ----
    np.testing.assert_array_equal(r0["init"], r1["init"])          # attach() broadcast rank 0's parameters over rank 1's perturbed ones
    np.testing.assert_array_equal(r0["gsum"], r1["gsum"])
    np.testing.assert_array_equal(r0["post"], r1["post"])
    for k in r0:
        if k.startswith("wlp_") or k.startswith("wl_"):
            np.testing.assert_array_equal(r0[k], r1[k])
    assert float(r0["loss"]) != float(r1["loss"])                  # ... although they saw different shards
    # ---- single process: the same two shards, one after the other ----
    from md_rdm_amd import harness
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    dev = torch.device("cuda:0")
    m = DepthEstimationNet()
    filler.fill_state_dict(m.state_dict())
    m = m.to(dev).train()
    m.flatten_parameters()
    m.direct_grads = True
    flat, gflat, entries = m._flat
    np.testing.assert_array_equal(flat.cpu().numpy(), r0["init"])
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    shard_g, wl_g, rm_after = [], [], []
    for r in range(2):
        m.load_state_dict(sd0)                                     # both shards start from the same weights and BatchNorm buffers
        x, y = filler.synthetic_batch(2, 228, 228, seed=(7, 14)[r])   # seeds with finite losses (no zero ordinal count)
        for p in m.weight_layer.parameters():
            p.grad = None
        loss, _ = harness.training_step(m, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
        loss.backward()
        shard_g.append(gflat.detach().clone())
        wl_g.append({n: p.grad.detach().clone() for n, p in m.weight_layer.named_parameters() if p.grad is not None})
        rm_after.append(m.state_dict()["encoder.dense_e3.denselayer5.norm2.running_mean"].cpu().numpy())
        assert abs(loss.item() - float((r0, r1)[r]["loss"])) <= 1e-5 * abs(loss.item())     # rank r computed the single-GPU result on ITS shard
    want_sum = (shard_g[0] + shard_g[1]).cpu().numpy()
    for a, b in m.stage_slices():                                  # the exchange, bucket by bucket
        num, den = np.linalg.norm(r0["gsum"][a:b] - want_sum[a:b]), np.linalg.norm(want_sum[a:b])
        assert num <= 2e-2 * den, (a, b, num / den)
    for n in wl_g[0]:
        np.testing.assert_allclose(r0["wl_" + n], ((wl_g[0][n] + wl_g[1][n]) / 2).cpu().numpy(), rtol=2e-3, atol=1e-7)
    # ---- the update: torch.optim.AdamW fed exactly the averaged gradient the ranks reduced ----
    ref_p = [torch.nn.Parameter(torch.from_numpy(r0["init"][o:o + n].copy())) for k, p, o, n, g in entries]
    for rp, (k, p, o, n, g) in zip(ref_p, entries):
        rp.grad = None if k.startswith("d_1.conv1.") else torch.from_numpy(r0["gsum"][o:o + n] * 0.5)
    torch.optim.AdamW(ref_p, lr=1e-4).step()
    for rp, (k, p, o, n, g) in zip(ref_p, entries):
        got, want = r0["post"][o:o + n], rp.detach().numpy()
        if k.startswith("d_1.conv1."):
            np.testing.assert_array_equal(got, r0["init"][o:o + n])
        else:
            assert np.abs(got - want).max() <= 2e-8 + 1e-6 * np.abs(want).max(), k
    # ---- buffers: rank 0's statistics (DDP broadcast_buffers), not rank 1's and not a mean ----
    np.testing.assert_allclose(r0["rm"], rm_after[0], rtol=1e-5, atol=1e-7)
    assert np.abs(rm_after[0] - rm_after[1]).max() > 1e-5
    assert int(r0["nbt"]) == 1


def test_bench_two_ranks_prints_one_weak_scaling_json_line():
    """The driver's N > 1 launch of bench.py, rehearsed at N = 2: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`
    (one rank per GPU over RCCL when the box has two; on the one-GPU box both ranks share the card and exchange over gloo).  Rank 0 must
    print exactly ONE JSON line: whole-job throughput over both ranks, weak scaling, the max-over-ranks step time, global batch = 2 x
    the per-GPU batch; the extra single-GPU configurations and the CPU baseline stay out of a multi-rank run."""
    import json
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = str(31200 + os.getpid() % 1500)
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)          # bench.py sets the dmabuf IPC mode itself (first lines of the file), before any HIP call
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--backend", backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]      # images of BOTH ranks / the slowest rank's time
    assert d["cpu_baseline"] is None and "extra_configs" not in d and "extra_configs" not in d["config"]
    assert d["roofline"]["bound"] == "mfma" and np.isfinite(d["config"]["loss"])
    # the line explains its own communication: payload of the 13-stage exchange (every conv-stack gradient once + the 4 scalars) and the
    # part of it the backward pass did not hide
    comm = d["config"]["comm"]
    assert comm["stages"] == 13 and 350e6 < comm["allreduce_bytes_per_step"] < 380e6
    assert comm["exposed_wait_ms_per_step"] >= 0.0 and "side_stream_join" in comm
