#!/usr/bin/env python3
"""Throughput against the number of windows in the batch (configs[1] shape): small batches cannot fill 256 CUs with whole
haplotypes, so the launcher splits a haplotype's reads over several workgroups (n_split) at the price of repeating the
per-haplotype setup."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

for faster in (False, True):
    for n in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
        pb = synth.generate(n, H=8, R=200, L=100, hap_len=120, seed=13)
        p = capi.params_cli_defaults()
        dev = DeviceBatch(pb, p, "cuda:0")
        launch = dev.launch_faster if faster else dev.launch
        launch(); torch.cuda.synchronize()
        reps = 20 if n <= 64 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        g = capi.last_launch()
        print(json.dumps(dict(model="faster" if faster else "main", windows=n, pairs=pb.n_pairs, ms=round(ms, 3),
                              cells_per_s=float("%.4g" % (pb.cells / ms * 1e3)), waves=g["waves"], grid=g["grid"], split=g["split"])), flush=True)
        del dev
