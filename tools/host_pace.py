"""Host-side enqueue time of one training step, phase by phase, WITHOUT synchronising (the GPU runs behind or beside): if the host needs about as long to
enqueue a phase as the GPU to execute it, that phase is launch-paced, not kernel-paced.
    python tools/host_pace.py"""
import sys, time, torch
sys.path.insert(0, ".")
from md_rdm_amd import filler, harness
from md_rdm_amd.network.RDM_Net import DepthEstimationNet
dev = torch.device("cuda:0")
m = DepthEstimationNet(); filler.fill_state_dict(m.state_dict()); m = m.to(dev).train()
x, y = filler.synthetic_batch(16, 228, 304, seed=1234)
xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
m.flatten_parameters()
opt = harness.FusedAdamW(m, lr=1e-4)
def step(rec=None):
    t = [time.perf_counter()]
    opt.zero_grad()
    yy = harness.prepare_target(yg); t.append(time.perf_counter())
    out = m(xg); t.append(time.perf_counter())
    fine_details, ord_depth_pred, ord_label_pred = out
    has_ordinal = fine_details[0].shape[2] == 1
    final_depth, fdl = harness.compute_final_depth(fine_details, yy, has_ordinal=has_ordinal)
    ord_y = harness.compute_ordinal_target(ord_depth_pred, yy)
    ord_loss = harness.l.Ordinal_Loss().calc(ord_label_pred, ord_y, cuda=True)
    mse = torch.nn.functional.mse_loss(final_depth, yy)
    loss = mse + fdl + ord_loss; t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if rec is not None: rec.append([b - a for a, b in zip(t, t[1:])])
for _ in range(3): step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(10): step(rec)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
names = ["zero_grad + target prep", "forward (native enqueue + tail)", "losses", "backward (autograd -> native enqueue)", "optimizer"]
avg = [sum(r[i] for r in rec) / len(rec) * 1e3 for i in range(5)]
print("host ms per step, unsynchronised:", {n: round(a, 2) for n, a in zip(names, avg)}, "sum", round(sum(avg), 2))
print(f"10 steps: host returned after {t_host * 1e2:.2f} ms/step, GPU done after {t_all * 1e2:.2f} ms/step")
# the same steps with an event at every phase boundary: GPU time of each phase step by step, and how far the host ran ahead of the GPU at the
# end of each step (host clock at the record call against the GPU clock of the event, both from the common start)
def step_ev(ev, host):
    def mark():
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e); host.append(time.perf_counter())
    mark(); opt.zero_grad(); yy = harness.prepare_target(yg)
    out = m(xg); mark()
    fine_details, ord_depth_pred, ord_label_pred = out
    final_depth, fdl = harness.compute_final_depth(fine_details, yy, has_ordinal=fine_details[0].shape[2] == 1)
    ord_y = harness.compute_ordinal_target(ord_depth_pred, yy)
    ord_loss = harness.l.Ordinal_Loss().calc(ord_label_pred, ord_y, cuda=True)
    loss = torch.nn.functional.mse_loss(final_depth, yy) + fdl + ord_loss; mark()
    loss.backward(); mark()
    opt.step(); mark()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.cuda.synchronize()
ev, host = [], []
for _ in range(N): step_ev(ev, host)
torch.cuda.synchronize()
print("step  forward  losses  backward  optimizer | step ms | host ahead of the GPU at step end, ms")
for i in range(N):
    e = ev[5 * i:5 * i + 5]
    d = [a.elapsed_time(b) for a, b in zip(e, e[1:])]
    nxt = ev[5 * i + 5] if i + 1 < N else None
    whole = e[0].elapsed_time(nxt) if nxt is not None else sum(d)
    lead = ev[0].elapsed_time(e[4]) - (host[5 * i + 4] - host[0]) * 1e3
    print(f"{i:4d}  {d[0]:7.2f} {d[1]:7.2f} {d[2]:9.2f} {d[3]:10.2f} | {whole:7.2f} | {lead:7.2f}")
