"""Kernel-by-kernel listing of the two host-paced phases of one training step (head + losses; stem backward + AdamW + the step boundary) from a rocprofv3
kernel trace: start offset, duration, gap to the previous kernel, name.
    python tools/phase_kernels.py OUT/train_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_im2col_stem" in r["Kernel_Name"]]
seq = rows[idx[-2]:idx[-1] + 3]
t0 = int(seq[0]["Start_Timestamp"])
def phase(start_key, stop_key, title, after_last=False):
    i0 = next(i for i, r in enumerate(seq) if start_key in r["Kernel_Name"])
    i1 = len(seq) if stop_key is None else next(i for i, r in enumerate(seq) if i > i0 and stop_key in r["Kernel_Name"])
    print(title)
    prev_end = int(seq[i0 - 1]["End_Timestamp"])
    for r in seq[i0:i1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  stream {r['Stream_Id']}  {r['Kernel_Name'][:90]}")
        prev_end = max(prev_end, e)
phase("k_dorn_fwd", "k_dorn_bwd", "== head + losses ==")
phase("k_maxpool3s2_bwd", None, "== stem backward + AdamW + step boundary ==")
