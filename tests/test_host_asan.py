"""SURVEY.md section 5 (sanitizers): the HOST side of librdm_hip - registry, plan construction / workspace layout, backward stage
table, size queries and every argument-validation path - built from the library's own sources with -fsanitize=address
(--cuda-host-only: no device code, no GPU) and run under AddressSanitizer + LeakSanitizer.  GPU-side sanitizers are not available
on this pool; the device kernels are covered by the NaN-poisoned padding / guard columns in the -m gpu operator tests."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("make") is None, reason="needs hipcc and make")
def test_host_side_of_the_c_abi_under_address_sanitizer():
    d = os.path.join(ROOT, "tests", "host_asan")
    r = subprocess.run(["make", "-s", "-j4", "-C", d], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:abort_on_error=0")
    r = subprocess.run([os.path.join(d, "_build", "host_asan")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "host-side ASAN check ok: 968 tensors, 90529721 parameters" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr
