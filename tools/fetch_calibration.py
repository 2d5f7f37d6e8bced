"""FETCH_SIZE calibration for the 3x3 dgrad's epilogue: the halo kernel reads its mask operand X one dword per lane in
64-byte segments, a width MI355X_MICROARCH.md leaves uncalibrated.  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cal -o c -- python3 tools/fetch_calibration.py
and summarise with `python tools/fetch_calibration.py --summarise gpurun_out/pmc_cal/c_counter_collection.csv`.
Dispatches: (1) float4 stream copy of X (known bytes, wide loads), (2) dense_e2 3x3 dgrad WITHOUT the mask epilogue, (3) the same WITH it.
(3) - (2) is the counter's view of reading X exactly once through the epilogue's access pattern."""
import ctypes as C, csv, json, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "FETCH_SIZE"]
    B, H, W, Cb = 16, 57, 76, 2736
    xb = B * H * W * Cb * 4
    per = {}
    for r in rows:
        per.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "")), 0.0)
        per[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))] += float(r["Counter_Value"]) * 1024.0
    copy = [v for (d, k), v in sorted(per.items()) if "k_stream_copy" in k]
    halo = [v for (d, k), v in sorted(per.items()) if "conv3x3_halo_kernel" in k]
    nomask, mask = halo[-2], halo[-1]
    out = {"x_bytes": xb, "copy_fetch_raw": copy[-1], "copy_factor": xb / copy[-1],
           "dgrad_nomask_fetch_raw": nomask, "dgrad_mask_fetch_raw": mask, "epilogue_x_fetch_raw": mask - nomask,
           "epilogue_factor": xb / (mask - nomask),
           "dgrad_main_loop_bytes_wide_corrected": 2 * nomask, "dgrad_total_bytes": 2 * nomask + xb,
           "compulsory_bytes": B * H * W * 48 * 4 + 9 * 48 * Cb * 4 + xb}
    out["traffic_over_compulsory"] = out["dgrad_total_bytes"] / out["compulsory_bytes"]
    json.dump(out, sys.stdout, indent=1)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, ptr, stream, check
L = _lib.lib()
if os.environ.get("RDM_VARIANT"): L.rdm_debug_variant(int(os.environ["RDM_VARIANT"]))
dev = torch.device("cuda")
B, H, W, Cb = 16, 57, 76, 2736
M = B * H * W
Y = torch.randn(M, Cb, device=dev); dZ = torch.empty(M, Cb, device=dev)
sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
w3 = torch.randn(9, 48, Cb, device=dev) * 0.01; g48 = torch.randn(M, 48, device=dev)
s0 = torch.zeros(Cb, dtype=torch.float64, device=dev); s1 = torch.zeros_like(s0)
d3 = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
big = torch.empty(300 * 2**20 // 4, device=dev)          # > 256 MiB streamed between dispatches: nothing of X survives in the Infinity Cache
def flush(): big.fill_(1.0)
for rep in range(2):
    flush(); check(_lib.bench_lib().rdm_microbench_copy(ptr(Y), ptr(dZ), M * Cb, stream()))
    flush(); check(L.rdm_conv2d_dgrad(C.byref(d3), ptr(g48), ptr(w3), ptr(dZ), Cb, None, 0, None, None, None, None, stream()))
    flush(); check(L.rdm_conv2d_dgrad(C.byref(d3), ptr(g48), ptr(w3), ptr(dZ), Cb, ptr(Y), Cb, ptr(sc), ptr(sh), ptr(s0), ptr(s1), stream()))
torch.cuda.synchronize()
print("done")
