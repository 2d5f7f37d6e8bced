"""Phase timeline of ONE training step from a rocprofv3 kernel trace (both streams), with the conv FLOPs of each phase:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o train -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline
    python tools/step_timeline.py OUT/train_kernel_trace.csv
Phases are cut at the transition / pooling kernels of the main stream (B=16, 228x304 geometry for the FLOP column)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_im2col_stem" in r["Kernel_Name"]]
seq = rows[idx[-2]:idx[-1]]
t0 = int(seq[0]["Start_Timestamp"])
us = lambda r, k: (int(r[k]) - t0) / 1e3
marks = [("stem", 0.0)]
names = iter(["dense_e2 fwd", "dense_e3 fwd", "dense_e4 fwd", "decoder fwd", "head + losses", "decoder bwd", "dense_e4 bwd", "dense_e3 bwd", "dense_e2 bwd", "stem bwd + AdamW"])
for r in seq:
    n = r["Kernel_Name"]
    if any(k in n for k in ("k_maxpool3s2(", "k_trans_pool(", "k_dorn_fwd", "k_dorn_bwd", "k_trans_pool_bwd_reduce", "k_maxpool3s2_bwd")):
        marks.append((next(names), us(r, "Start_Timestamp")))
end = us(rows[idx[-1]], "Start_Timestamp")
# conv GFLOP per phase at B=16 228x304 (forward; backward = 2x)
B = 16
geo = {"dense_e2": (B * 57 * 76, 96, 6, 2736), "dense_e3": (B * 29 * 38, 384, 12, 1392), "dense_e4": (B * 15 * 19, 1056, 36, 720), "decoder": (B * 8 * 10, 1056, 24, 384)}
def gf(name):
    M, c0, L, cb = geo[name]
    return sum(2.0 * M * ((c0 + 48 * i) * cb + 9 * cb * 48) for i in range(L)) / 1e9
print(f"one step = {end / 1e3:.2f} ms, {len(seq)} kernel launches (main stream {sum(1 for r in seq if r['Stream_Id'] == seq[0]['Stream_Id'])})")
for (nm, a), (_, b) in zip(marks, marks[1:] + [("end", end)]):
    key = nm.split(" ")[0]
    fl = gf(key) * (2 if "bwd" in nm else 1) if key in geo else 0.0
    busy = {}
    for r in seq:
        s, e = us(r, "Start_Timestamp"), us(r, "End_Timestamp")
        if s >= a and s < b:
            busy[r["Stream_Id"]] = busy.get(r["Stream_Id"], 0.0) + (e - s)
    extra = f"  {fl:7.0f} GFLOP -> {fl / (b - a) * 1e3:6.1f} TFLOP/s = {fl / (b - a) * 1e3 / 157.3 * 100:4.1f} % of the f32 MFMA peak" if fl else ""
    print(f"{nm:18s} {a / 1e3:7.2f} .. {b / 1e3:7.2f} ms  ({(b - a) / 1e3:6.2f} ms; kernel time main {busy.get(seq[0]['Stream_Id'], 0) / 1e3:6.2f} ms, side {sum(v for k, v in busy.items() if k != seq[0]['Stream_Id']) / 1e3:6.2f} ms){extra}")
