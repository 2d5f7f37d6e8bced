"""In-tree build of librdm_hip.so (hipcc, gfx950 only).  `python -m md_rdm_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "librdm_hip.so")
BENCH_SRC = os.path.join(CSRC, "bench")
BENCH_LIB = os.path.join(HERE, "librdm_bench.so")           # measurement kernels of tools/ and bench_ops.py: not the product
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


# per-file extra flags (none at present; measured on wino.hip: -fno-slp-vectorize, i.e. scalar v_add_f32 instead of v_pk_add_f32 in the
# in-register Winograd transforms, is 1-3 % SLOWER, so the default vectoriser stays on)
EXTRA = {}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """RDM_DEV_VARIANTS=1 in the environment compiles the development A/B switch in (rdm_debug_variant); the default ("ship") build has
    none.  Switching mode rebuilds everything."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    mode = "dev" if os.environ.get("RDM_DEV_VARIANTS", "0") not in ("", "0") else "ship"
    stamp = os.path.join(OBJ, ".mode")
    if not os.path.exists(stamp) or open(stamp).read().strip() != mode:
        force = True
    flags = FLAGS + (["-DRDM_DEV_VARIANTS"] if mode == "dev" else [])
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "rdm_hip.h")]
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, src[:-4] + ".o")
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + flags + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose:
            print(f"[build] {src}", flush=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB} ({mode})", flush=True)
    # the development tools' measurement kernels: their own library next to the product, resolving set_error / the launch counter from it
    bsrcs = sorted(f for f in os.listdir(BENCH_SRC) if f.endswith(".hip")) if os.path.isdir(BENCH_SRC) else []
    bobjs = []
    for src in bsrcs:
        obj = os.path.join(OBJ, "bench_" + src[:-4] + ".o")
        bobjs.append(obj)
        if force or _stale(obj, [os.path.join(BENCH_SRC, src)] + headers + [os.path.join(HERE, "..", "include", "rdm_bench.h")]):
            r = subprocess.run([hipcc] + flags + ["-c", os.path.join(BENCH_SRC, src), "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on bench/{src}:\n{r.stderr}")
            if verbose:
                print(f"[build] bench/{src}", flush=True)
    if bobjs and (force or _stale(BENCH_LIB, bobjs + [LIB])):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", BENCH_LIB] + bobjs + ["-L" + HERE, "-lrdm_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[build] linked {BENCH_LIB}", flush=True)
    with open(stamp, "w") as fh:
        fh.write(mode)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
