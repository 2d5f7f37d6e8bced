"""Where is the GPU idle inside one training step?  From a rocprofv3 --kernel-trace CSV (kernel trace only: the step then runs within 2 % of its unprofiled
time): the union of the busy intervals of ALL streams of the last whole step, the time with no kernel resident per millisecond of the step, and the
largest single holes with the kernels either side of them.
    python tools/step_idle.py OUT/train_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_im2col_stem" in r["Kernel_Name"]]
seq = rows[idx[-2]:idx[-1] + 1]
t0 = int(seq[0]["Start_Timestamp"])
T = int(seq[-1]["Start_Timestamp"]) - t0
holes, busy_end, last = [], int(seq[0]["End_Timestamp"]), seq[0]
for r in seq[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        holes.append((busy_end - t0, s - busy_end, last["Kernel_Name"][:70], r["Kernel_Name"][:70]))
    if e > busy_end:
        busy_end, last = e, r
idle = sum(h[1] for h in holes)
print(f"step {T / 1e6:.2f} ms, {len(seq) - 1} kernels, no kernel resident for {idle / 1e6:.2f} ms in {len(holes)} holes")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 20e3), (20e3, 50e3), (50e3, 1e12)):
    sel = [h for h in holes if lo <= h[1] < hi]
    print(f"  holes of {lo / 1e3:4.0f}-{min(hi, 1e9) / 1e3:.0f} us: {len(sel):5d}  total {sum(h[1] for h in sel) / 1e6:6.3f} ms")
print("per millisecond of the step: idle us (kernels started)")
nms = T // 1000000 + 1
per, cnt = [0.0] * nms, [0] * nms
for at, d, _, _ in holes:
    per[min(nms - 1, at // 1000000)] += d / 1e3
for r in seq[:-1]:
    cnt[min(nms - 1, (int(r["Start_Timestamp"]) - t0) // 1000000)] += 1
for i in range(nms):
    print(f"  {i:3d} ms  idle {per[i]:6.1f} us   {cnt[i]:4d} kernels")
print("largest holes")
for at, d, a, b in sorted(holes, key=lambda h: -h[1])[:25]:
    print(f"  +{at / 1e6:7.3f} ms  {d / 1e3:7.1f} us  after {a}\n{'':30s}before {b}")
