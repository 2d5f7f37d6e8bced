"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - they do not fit one pass on gfx950)
into HBM-side bytes per kernel and per training step.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/hbm_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv 3 > profiles/rNN_hbm_traffic.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B / lane) streaming reads at 64 bytes, so the read side is doubled.  The
correction is checked in the same pass on k_adamw, whose traffic is known exactly (4 reads + 3 writes of
the flat parameter buffer).  Loads narrower than 16 B / lane are uncalibrated: the doubled figure is an upper bound.
"""
import collections
import csv
import json
import sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != name:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0
    return agg


def main():
    fetch, write, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    n_params = 90529721
    kernels = []
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        kernels.append({"kernel": k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), "launches_per_step": round(fetch[k][0] / steps, 2),
                        "fetch_raw_bytes_per_step": round(fetch[k][1] / steps), "write_bytes_per_step": round(write.get(k, [0, 0.0])[1] / steps)})
    conv = [k for k in kernels if "conv_" in k["kernel"] or "conv3x3" in k["kernel"] or "conv1x1" in k["kernel"]]
    adam = [k for k in kernels if "k_adamw" in k["kernel"]][0]
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py, batch 16, 228x304",
        "steps_profiled": steps,
        "calibration": {"kernel": "k_adamw", "expected_read_bytes": 4 * 4 * n_params, "fetch_raw_bytes": adam["fetch_raw_bytes_per_step"],
                        "expected_write_bytes": 3 * 4 * n_params, "write_bytes": adam["write_bytes_per_step"],
                        "read_correction": 2.0},
        "conv_kernels": {"launches_per_step": round(sum(k["launches_per_step"] for k in conv), 1),
                         "read_bytes_per_step": round(2.0 * sum(k["fetch_raw_bytes_per_step"] for k in conv)),
                         "write_bytes_per_step": round(sum(k["write_bytes_per_step"] for k in conv))},
        "all_kernels": {"read_bytes_per_step": round(2.0 * sum(k["fetch_raw_bytes_per_step"] for k in kernels)),
                        "write_bytes_per_step": round(sum(k["write_bytes_per_step"] for k in kernels))},
        "per_kernel": kernels[:40],
    }
    c = out["conv_kernels"]
    c["bytes_per_launch"] = round((c["read_bytes_per_step"] + c["write_bytes_per_step"]) / c["launches_per_step"])
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
