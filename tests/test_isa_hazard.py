"""CPU test (hipcc cross-compiles gfx950 without a GPU): no kernel of the library may contain the store-data hazard that
tools/store_hazard/store_hazard_probe.hip demonstrates on MI355X - a MUBUF store of more than 64 bits with an SGPR soffset whose data
registers are overwritten in the next issue slots (hipcc exempts that form from the wait states it inserts; the hardware does not:
profiles/r05_store_hazard_probe.txt).  Sources that can emit such a store (they mention a buffer store) are compiled to ISA and scanned."""
import glob
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools", "store_hazard"))
import scan_isa  # noqa: E402


def _isa(src, out_dir):
    dst = os.path.join(out_dir, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
                    src, "-o", dst], check=True, stderr=subprocess.DEVNULL)
    return dst


def test_no_unprotected_wide_buffer_store_with_scalar_offset(tmp_path):
    srcs = [f for f in sorted(glob.glob(os.path.join(ROOT, "md_rdm_amd", "csrc", "*.hip"))) if "buffer_store" in open(f).read()]
    assert any(f.endswith("elementwise.hip") for f in srcs)                     # k_bn_bwd_apply stores 16 bytes per lane at a scalar row offset
    bad = []
    for f in srcs:
        bad += scan_isa.scan(_isa(f, str(tmp_path)))
    assert not bad, bad


def test_the_scanner_sees_the_hazard_in_the_reproducer(tmp_path):
    """Variant 0 of the reproducer IS the hazard (hand-written asm: store, then an immediate overwrite of its data); variant 1 leaves ONE wait
    state (enough on the hardware, fewer than the two the scanner asks for - what hipcc leaves for the non-exempt form); 2 and 3 are protected."""
    s = _isa(os.path.join(ROOT, "tools", "store_hazard", "store_hazard_probe.hip"), str(tmp_path))
    found = scan_isa.scan(s)
    assert len(found) >= 1 and all("buffer_store_dwordx4 v[100:103]" in f[3] for f in found), found
    kernels = {f[2] for f in found}
    assert any("Li0E" in k for k in kernels) and not any("Li2E" in k or "Li3E" in k for k in kernels), kernels
