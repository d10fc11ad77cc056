#!/usr/bin/env python3
"""Throughput against reads per window (coverage): the per-haplotype setup of a workgroup (state symbols, homopolymer
tables, per-lane constants) is amortised over the window's reads, so thin windows pay relatively more."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

for faster in (False, True):
    for R in (2, 5, 10, 20, 50, 100, 200, 1000):
        n = max(4, int(3.2e6 / (8 * R)))
        pb = synth.generate(n, H=8, R=R, L=100, hap_len=120, seed=11)
        p = capi.params_cli_defaults()
        dev = DeviceBatch(pb, p, "cuda:0")
        launch = dev.launch_faster if faster else dev.launch
        launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        g = capi.last_launch()
        print(json.dumps(dict(model="faster" if faster else "main", reads_per_window=R, windows=n, pairs=pb.n_pairs, ms=round(ms, 2),
                              cells_per_s=float("%.4g" % (pb.cells / ms * 1e3)), waves=g["waves"], grid=g["grid"], split=g["split"])), flush=True)
        del dev
