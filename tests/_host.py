"""ctypes access to the C++ host adapter (dindel_tgi_amd/host/libdindel_host.so) for the tests."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

from dindel_tgi_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "dindel_tgi_amd", "host")
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


def _params(p):
    pd = (C.c_double * 6)(p.pError, p.pMut, p.pFirstgLO, p.mapQualThreshold, p.checkBaseQualThreshold, p.capMapQualFast)
    pi = (C.c_int * 3)(p.maxLengthDel, p.padCover, p.bMid)
    return pd, pi


def rebuild(hap, read, qual, hpos, params, hap_indels=(), faster=False):
    lib = load()
    L = len(read)
    q = np.ascontiguousarray(np.broadcast_to(np.asarray(qual, np.float64), (L,)))
    hp = np.asarray(hpos, np.int16)
    pd, pi = _params(params)
    hv = np.asarray([v for t in hap_indels for v in t], np.int32)
    out = C.create_string_buffer(1 << 20)
    n = lib.ddh_rebuild_json(hap.encode(), read.encode(), q.ctypes.data_as(capi.c_f64p),
                             hp.ctypes.data_as(C.POINTER(C.c_short)), L, pd, pi,
                             hv.ctypes.data_as(capi.c_i32p), len(hap_indels) | (0x10000 if faster else 0), out, len(out))
    assert n > 0
    return json.loads(out.value.decode())


def compute_window(haps, reads, quals, mapq, pos_first, unmapped, left_pos, params, device=0, faster=False):
    lib = load()
    q = np.ascontiguousarray(np.concatenate([np.broadcast_to(np.asarray(x, np.float64), (len(r),)) for x, r in zip(quals, reads)]))
    mq = np.asarray(mapq, np.float64)
    pf = np.asarray(pos_first, np.float64)
    um = np.asarray(unmapped, np.int32)
    pd, pi = _params(params)
    out = C.create_string_buffer(1 << 24)
    fn = lib.ddh_compute_window_faster_json if faster else lib.ddh_compute_window_json
    n = fn("\n".join(haps).encode(), "\n".join(reads).encode(), q.ctypes.data_as(capi.c_f64p),
                                    mq.ctypes.data_as(capi.c_f64p), pf.ctypes.data_as(capi.c_f64p),
                                    um.ctypes.data_as(capi.c_i32p), C.c_uint(left_pos & 0xFFFFFFFF), pd, pi, device, out, len(out))
    assert n > 0, n
    return json.loads(out.value.decode())


def filter_window(haps, hap_vars, reads, quals, mapq, pos_first, rflags, left_pos, params, do_filter=True, device=0, faster=False):
    """computeLikelihoods + filterHaplotypes through the C++ host adapter.  hap_vars: per hap list of
    (key, kind, leftFlankRead, rightFlankRead)."""
    lib = load()
    q = np.ascontiguousarray(np.concatenate([np.broadcast_to(np.asarray(x, np.float64), (len(r),)) for x, r in zip(quals, reads)]))
    mq = np.asarray(mapq, np.float64); pf = np.asarray(pos_first, np.float64); rf = np.asarray(rflags, np.int32)
    hv = []
    for vs in hap_vars:
        hv.append(len(vs))
        for t in vs:
            hv += list(t)
    hv = np.asarray(hv, np.int32)
    pd, pi = _params(params)
    out = C.create_string_buffer(1 << 24)
    n = lib.ddh_filter_window_json("\n".join(haps).encode(), hv.ctypes.data_as(capi.c_i32p), "\n".join(reads).encode(),
                                   q.ctypes.data_as(capi.c_f64p), mq.ctypes.data_as(capi.c_f64p), pf.ctypes.data_as(capi.c_f64p),
                                   rf.ctypes.data_as(capi.c_i32p), C.c_uint(left_pos & 0xFFFFFFFF), pd, pi, params.maxMismatch,
                                   (1 if do_filter else 0) | (2 if faster else 0), device, out, len(out))
    assert n > 0, n
    return json.loads(out.value.decode())


def compute_window_mates(haps, reads, quals, mapq, pos_first, unmapped, left_pos, params, mate, lib_counts, device=0):
    """computeLikelihoods with mapUnmappedReads: mate = per read (flags, matePos, mateLen, lib); lib_counts = list of histograms."""
    lib = load()
    q = np.ascontiguousarray(np.concatenate([np.broadcast_to(np.asarray(x, np.float64), (len(r),)) for x, r in zip(quals, reads)]))
    mq = np.asarray(mapq, np.float64); pf = np.asarray(pos_first, np.float64); um = np.asarray(unmapped, np.int32)
    mt = np.ascontiguousarray(np.asarray(mate, np.int32).reshape(-1))
    lc = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64) for c in lib_counts]))
    ls = np.asarray([len(c) for c in lib_counts], np.int32)
    pd, pi = _params(params)
    out = C.create_string_buffer(1 << 24)
    n = lib.ddh_compute_window_mates_json("\n".join(haps).encode(), "\n".join(reads).encode(), q.ctypes.data_as(capi.c_f64p),
                                          mq.ctypes.data_as(capi.c_f64p), pf.ctypes.data_as(capi.c_f64p),
                                          um.ctypes.data_as(capi.c_i32p), C.c_uint(left_pos & 0xFFFFFFFF), pd, pi, device,
                                          mt.ctypes.data_as(capi.c_i32p), lc.ctypes.data_as(capi.c_f64p),
                                          ls.ctypes.data_as(capi.c_i32p), len(lib_counts), out, len(out))
    assert n > 0, n
    return json.loads(out.value.decode())
