"""Decoder d_1 (RDM_Net.py:144: 24 dense layers at 8x10, B=16 -> 1280 pixels) in one training step, from a rocprofv3 kernel trace of bench.py:
span, kernel time per stream, idle gaps of the dependent chain and launches per layer - the numbers behind DESIGN.md's bound on what a
persistent cooperative kernel could gain.   python tools/decoder_trace.py OUT/train_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_im2col_stem" in r["Kernel_Name"]]
seq = rows[idx[-2]:idx[-1]]
main = seq[0]["Stream_Id"]
def phase(start_pred, end_pred, nth_start=0):
    s = [i for i, r in enumerate(seq) if start_pred(r["Kernel_Name"])][nth_start]
    e = next(i for i in range(s + 1, len(seq)) if end_pred(seq[i]["Kernel_Name"]))
    return seq[s:e]
def report(name, ks, layers):
    t0, t1 = int(ks[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in ks)
    m = [r for r in ks if r["Stream_Id"] == main]
    busy_m = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in m)
    busy_s = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks if r["Stream_Id"] != main)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(m, m[1:])]
    pos = [g for g in gaps if g > 0]
    mf = [r for r in ks if any(k in r["Kernel_Name"] for k in ("conv_", "conv3x3", "conv1x1"))]
    conv = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mf)
    print(f"{name}: span {(t1 - t0) / 1e3:.0f} us, {len(ks)} launches ({len(m)} on the dependent chain = {len(m) / layers:.1f} per layer), chain kernel time {busy_m / 1e3:.0f} us, "
          f"side-stream kernel time {busy_s / 1e3:.0f} us, idle between chain kernels {sum(pos) / 1e3:.0f} us in {len(pos)} gaps (median {sorted(pos)[len(pos) // 2] / 1e3:.2f} us), "
          f"MFMA kernels {len(mf)} launches / {conv / 1e3:.0f} us")
    by = {}
    for r in m:
        n = r["Kernel_Name"].replace("void ", "").replace("rdm::", "").split("(")[0][:48]
        d = by.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"    {n:48s} x{c:4d}  {d / 1e3:8.1f} us  ({d / c / 1e3:.1f} us each)")
    by = {}
    for r in ks:
        if r["Stream_Id"] == main: continue
        n = r["Kernel_Name"].replace("void ", "").replace("rdm::", "").split("(")[0][:48]
        d = by.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"    (side) {n:41s} x{c:4d}  {d / 1e3:8.1f} us  ({d / c / 1e3:.1f} us each)")
e4f = phase(lambda n: "k_trans_pool(" in n, lambda n: "k_trans_pool(" in n, nth_start=1)
report("dense_e4 forward (36 layers + its transition conv)", e4f, 36)
fwd = phase(lambda n: "k_trans_pool(" in n, lambda n: "k_dorn_fwd" in n, nth_start=2)
report("decoder forward (24 layers)", fwd, 24)
bwd = phase(lambda n: "k_dorn_bwd" in n, lambda n: "k_trans_pool_bwd_reduce" in n)
report("decoder backward (24 layers + head)", bwd, 24)
e4b = phase(lambda n: "k_trans_pool_bwd_reduce" in n, lambda n: "k_trans_pool_bwd_reduce" in n, nth_start=0)
report("dense_e4 backward (transition + 36 layers)", e4b, 36)
