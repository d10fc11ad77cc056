#!/usr/bin/env python3
"""Ragged 'real-data-like' batch through the host-pointer API: haplotype lengths 100..170 (one in ~12 needs the K=3
tiling), read lengths 36/76/100/150 mixed.  Compares length-class launches (default) with a single batch-wide
launch plan (DD_NO_LENGTH_CLASSES=1)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.batch import PackedBatch, alloc_result
rng = np.random.default_rng(4)
parts = []
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for i in range(NW):
    hl = int(np.clip(rng.normal(118, 12), 100, 170))
    L = int(rng.choice([36, 76, 100, 100, 100, 150]))
    parts.append(synth.generate(1, H=int(rng.integers(2, 9)), R=int(rng.integers(20, 300)), L=L, hap_len=hl, seed=1000 + i,
                                vary_read_len=(i % 5 == 0)))
def cat(parts):
    a = {k: [] for k in PackedBatch.FIELDS}
    offs = dict(win_hap_off=0, win_read_off=0, hap_seq_off=0, hap_var_off=0, read_seq_off=0)
    out = {k: [] for k in PackedBatch.FIELDS}
    for k in offs: out[k].append(np.zeros(1, np.int64))
    for p in parts:
        for k in PackedBatch.FIELDS:
            v = p.a[k]
            if k in offs:
                out[k].append(v[1:].astype(np.int64) + offs[k]); offs[k] += int(v[-1])
            elif k in ("qual_table", "mapq_table"):
                out[k] = [v]
            else:
                out[k].append(v)
    return PackedBatch(**{k: np.concatenate(v) for k, v in out.items()})
pb = cat(parts)
lib = capi.load()
p = capi.params_cli_defaults()
arrs, res = alloc_result(pb)
b = pb.ctypes_batch()
for _ in range(3):
    t0 = time.perf_counter(); rc = lib.dd_compute_likelihoods(C.byref(p), C.byref(b), C.byref(res), 0); dt = time.perf_counter() - t0
    assert rc == 0, capi.last_error()
print("single " if os.environ.get("DD_NO_LENGTH_CLASSES") else "classes=" + os.environ.get("DD_LENGTH_CLASSES", "kl"), "pairs", pb.n_pairs, "max_hap", pb.max_hap_len, "max_read", pb.max_read_len,
      "seconds %.4f" % dt, "cells/s %.3e" % (pb.cells / dt), "checksum %.6f" % float(arrs["ll"][:pb.n_pairs].sum()))
