"""ctypes access to the C++ host adapter (dindel_tgi_amd/host/libdindel_host.so) for the tests."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

from dindel_tgi_amd import capi

from dindel_tgi_amd import hostlib

load = hostlib.load


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


def batch(windows, params, faster=False, keep_alignments=True, device=0):
    """LikelihoodEngine::computeLikelihoodsBatch over dindel_tgi_amd.batch.Window objects, eager and lazy (ddh_batch_json):
    {"windows": [{"error": str, "ll": [...], "onHap": [...]}], "mismatch": lazy-vs-eager differences}."""
    lib = load()
    haps = [h for w in windows for h in w.haps]
    reads = [r for w in windows for r in w.reads]
    nh = np.asarray([len(w.haps) for w in windows], np.int32)
    nr = np.asarray([len(w.reads) for w in windows], np.int32)
    q = np.ascontiguousarray(np.concatenate([np.asarray(r.qual, np.float64).reshape(-1) for r in reads] + [np.zeros(1)]))
    mq = np.asarray([r.mapQual for r in reads] + [0.0], np.float64)
    pf = np.asarray([float(r.start) for r in reads] + [0.0], np.float64)
    um = np.asarray([int(r.unmapped) for r in reads] + [0], np.int32)
    lp = np.asarray([w.hap_start & 0xFFFFFFFF for w in windows], np.uint32)
    pd, pi = _params(params)
    out = C.create_string_buffer(1 << 26)
    n = lib.ddh_batch_json(len(windows), nh.ctypes.data_as(capi.c_i32p), nr.ctypes.data_as(capi.c_i32p),
                           "\n".join(haps).encode(), "\n".join(r.seq for r in reads).encode(), q.ctypes.data_as(capi.c_f64p),
                           mq.ctypes.data_as(capi.c_f64p), pf.ctypes.data_as(capi.c_f64p), um.ctypes.data_as(capi.c_i32p),
                           lp.ctypes.data_as(capi.c_u32p), pd, pi, (1 if faster else 0) | (0 if keep_alignments else 2), device,
                           out, len(out))
    assert n > 0, n
    return json.loads(out.value.decode())


from dindel_tgi_amd.hostlib import bench_batch  # noqa: E402,F401
