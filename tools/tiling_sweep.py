#!/usr/bin/env python3
"""The lane-tiling cliff, tracked: configs[1]-shaped batches (8 haplotypes x 200 reads of 100 bp) at haplotype lengths between and at the
steps of the tilings (a wavefront sweeps 64 K / pairs-per-wave positions per pair whatever the haplotype needs).  One JSON line per point:
longest haplotype, the launch(es) the library chose, cells/s.   python tools/tiling_sweep.py [L] [maxLengthDel]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mld = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for hap in (25, 40, 59, 62, 77, 91, 97, 110, 120, 122, 124, 127, 137, 147, 155, 157, 170, 187, 192, 197, 219, 250, 315):
    n = max(40, int(1500 * 120 / hap))
    pb = synth.generate(n, H=8, R=200, L=L, hap_len=hap, seed=11, max_indel=3)
    p = capi.params_cli_defaults(); p.maxLengthDel = mld
    dev = DeviceBatch(pb, p, "cuda:0")
    for _ in range(3):
        dev.launch()
    torch.cuda.synchronize()
    best = None
    for _rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            dev.launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        best = ms if best is None else min(best, ms)
    log = capi.launch_log()
    print(json.dumps(dict(L=L, mld=mld, max_hap=pb.max_hap_len, pairs=pb.n_pairs, ms=round(best, 3), cells_per_s=float("%.4g" % (pb.cells / best * 1e3)),
                          launches=[dict(K=r["K"], ppw=r["pairs_per_wave"], bt="hbm" if r["gbt"] else "lds", fold=r["fold"], waves=r["waves"], max_hap=r["max_hap"]) for r in log])), flush=True)
