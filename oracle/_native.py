"""ORACLE (test infrastructure): loader of the C restatements in oracle/*.c (built by oracle/Makefile into
oracle/_build/liboracle.so; ``__graft_entry__.build()`` builds it, and it is rebuilt here on demand - gcc, < 1 s)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.rdm_oracle_resize_bicubic_f64.restype = None
        L.rdm_oracle_resize_bicubic_f64.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 5
        _lib = L
    return _lib
