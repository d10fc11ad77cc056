"""ctypes access to the C++ host adapter (dindel_tgi_amd/host/libdindel_host.so: dindel::LikelihoodEngine, the mirror of
DetInDel::computeLikelihoods) — used by bench.py's end-to-end leg and by the tests.  No Python implementation behind it."""
import ctypes as C
import os
import subprocess

from . import capi

HOST_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
LIB = os.environ.get("DD_HOST_LIB") or os.path.join(HOST_DIR, "libdindel_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        capi.load()                     # libdindel_hmm.so (and torch's HIP runtime) first
        srcs = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith((".cpp", ".hpp"))]
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-s", "-C", HOST_DIR])
        _lib = C.CDLL(LIB)
    return _lib


def bench_batch(W, H=8, R=200, L=100, HL=120, seed=0, eager=False, keep_alignments=True, faster=False, device=0, reps=2):
    """ddh_bench_batch (host/host_capi.cpp): W synthetic windows as the reference's own objects through
    LikelihoodEngine::computeLikelihoodsBatch; best wall time of `reps` calls after a warm-up call."""
    lib = load()
    out = (C.c_double * 8)()
    lib.ddh_bench_batch.argtypes = [C.c_int] * 5 + [C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    rc = lib.ddh_bench_batch(W, H, R, L, HL, seed, (1 if eager else 0) | (0 if keep_alignments else 2) | (4 if faster else 0),
                             device, reps, out)
    if rc != 0:
        raise RuntimeError("ddh_bench_batch failed")
    return dict(seconds=out[0], pack=out[1], device=out[2], finish=out[3], ll00=out[4], errors=int(out[5]), windows=W)
