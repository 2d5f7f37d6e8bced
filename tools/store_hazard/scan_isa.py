#!/usr/bin/env python3
"""Audit of compiled gfx950 ISA for the store-data hazard tools/store_hazard/store_hazard_probe.hip demonstrates: a MUBUF store of more than
64 bits whose soffset is an SGPR, with a VALU write to one of its data VGPRs closer than WAIT wait states behind it.  hipcc exempts exactly
that form from the >64-bit store-data hazard (LLVM GCNHazardRecognizer::createsVALUHazard); on MI355X the exemption does not hold
(profiles/r05_store_hazard_probe.txt: lanes 12-15 of every 16 store the overwritten dword).

    python tools/store_hazard/scan_isa.py [file.s ...]     (no arguments: compiles every md_rdm_amd/csrc/*.hip to ISA first)
Exit code 1 if a hazard is found."""
import glob
import os
import re
import subprocess
import sys

WAIT = 2                                 # wait states LLVM itself leaves for the non-exempt form on gfx940+
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
INSN = re.compile(r"^\s*(v_|s_|buffer_|global_|flat_|scratch_|ds_|tbuffer_|image_|exp\b)")
VDST = re.compile(r"^\s*v_\w+\s+(v\[(\d+):(\d+)\]|v(\d+))")


def regs_written(line):
    m = VDST.match(line)
    if not m or line.lstrip().startswith(("v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane", "v_nop")):
        return set()
    if m.group(2):
        return set(range(int(m.group(2)), int(m.group(3)) + 1))
    return {int(m.group(4))}


def wait_states(line):
    m = re.match(r"^\s*s_nop\s+(\d+)", line)
    return int(m.group(1)) + 1 if m else 1


def scan(path):
    lines = [l for l in open(path).read().split("\n")]
    # instruction lines: compiler output is tab-indented, inline-asm text keeps whatever indentation its string had (often none)
    code = [(i, l) for i, l in enumerate(lines) if INSN.match(l)]
    found, kernel = [], "?"
    names = {i: l.split(":")[0] for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)}
    for k, (i, l) in enumerate(code):
        m = STORE.match(l)
        if not m or not m.group(5).startswith("s"):          # immediate soffset: the form hipcc already protects
            continue
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        dist = 0
        for j, l2 in code[k + 1:k + 1 + WAIT]:
            if dist >= WAIT:
                break
            hit = data & regs_written(l2)
            if hit:
                kernel = max((n for n in names if n < i), default=None)
                found.append((path, i + 1, names.get(kernel, "?"), l.strip(), l2.strip(), dist))
                break
            dist += wait_states(l2)
    return found


def main():
    files = sys.argv[1:]
    if not files:
        out = os.path.join(ROOT, "scratch", "isa")
        os.makedirs(out, exist_ok=True)
        for src in sorted(glob.glob(os.path.join(ROOT, "md_rdm_amd", "csrc", "*.hip"))):
            dst = os.path.join(out, os.path.basename(src)[:-4] + ".s")
            if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
                subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
                                "-I" + os.path.join(ROOT, "include"), src, "-o", dst], check=True, stderr=subprocess.DEVNULL)
            files.append(dst)
    bad = []
    for f in files:
        bad += scan(f)
    for path, line, kern, st, wr, dist in bad:
        print(f"{os.path.basename(path)}:{line} [{kern[:70]}] {st}  <- {dist} wait state(s) ->  {wr}")
    print(f"{len(files)} ISA files scanned, {len(bad)} unprotected >64-bit buffer stores with an SGPR soffset")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
