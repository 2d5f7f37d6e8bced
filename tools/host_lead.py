"""How far ahead of the GPU is the host?  From one rocprofv3 run with --hip-runtime-trace --kernel-trace: for chosen kernels of the last step, the time between
the hipLaunchKernel call that enqueued them and their start on the GPU.
    python tools/host_lead.py OUT/x_hip_api_trace.csv OUT/x_kernel_trace.csv"""
import csv, sys
api = [r for r in csv.DictReader(open(sys.argv[1])) if r["Function"] in ("hipLaunchKernel", "hipExtLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel")]
ker = list(csv.DictReader(open(sys.argv[2])))
by_corr = {r["Correlation_Id"]: r for r in api}
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ker) if "k_im2col_stem" in r["Kernel_Name"]]
seq = ker[idx[-2]:idx[-1]]
t0 = int(seq[0]["Start_Timestamp"])
out = []
for i, r in enumerate(seq):
    a = by_corr.get(r["Correlation_Id"])
    if a is None: continue
    lead = (int(r["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    out.append(((int(r["Start_Timestamp"]) - t0) / 1e3, lead, r["Kernel_Name"][:60]))
print(f"{len(out)} of {len(seq)} kernels of the step matched to their launch call")
step = max(1, len(out) // 60)
for t, lead, n in out[::step]:
    print(f"  GPU +{t / 1e3:7.2f} ms   enqueued {lead / 1e3:8.3f} ms before it started   {n}")
