#!/usr/bin/env python3
"""Latency of the literal drop-in use: one window per dd_compute_likelihoods call (host pointers)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.batch import alloc_result
lib = capi.load()
p = capi.params_cli_defaults()
for H, R in ((4, 50), (8, 200), (8, 1000)):
    pb = synth.generate(1, H=H, R=R, seed=5)
    arrs, res = alloc_result(pb)
    b = pb.ctypes_batch()
    for _ in range(3):
        lib.dd_compute_likelihoods(C.byref(p), C.byref(b), C.byref(res), 0)
    t0 = time.perf_counter(); n = 50
    for _ in range(n):
        rc = lib.dd_compute_likelihoods(C.byref(p), C.byref(b), C.byref(res), 0)
    dt = (time.perf_counter() - t0) / n
    print("1 window x %d haps x %d reads: %.3f ms per call = %.0f windows/s, %.3e cells/s" % (H, R, dt * 1e3, 1 / dt, pb.cells / dt))
