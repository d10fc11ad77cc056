#!/usr/bin/env python3
"""BASELINE.json configs[4]: stress sweep — 250 bp reads, 16 haplotypes/window, maxLengthDel=10 (D=11),
haplotype lengths 120/160/200 (+ the configs[1] shape for reference): throughput, LDS per wave and the
resulting occupancy, HBM roofline fraction.  Prints one JSON line per point; run on the GPU box."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

points = [dict(L=100, H=8, R=200, hap=120, mld=5, n=2000),
          dict(L=100, H=8, R=200, hap=120, mld=10, n=2000),
          dict(L=250, H=16, R=200, hap=120, mld=10, n=400),
          dict(L=250, H=16, R=200, hap=160, mld=10, n=300),
          dict(L=250, H=16, R=200, hap=200, mld=10, n=300),
          dict(L=250, H=16, R=200, hap=120, mld=5, n=400),
          dict(L=150, H=16, R=200, hap=160, mld=10, n=400)]
for pt in points:
    pb = synth.generate(pt["n"], H=pt["H"], R=pt["R"], L=pt["L"], hap_len=pt["hap"], seed=99, max_indel=3)
    p = capi.params_cli_defaults()
    p.maxLengthDel = pt["mld"]
    dev = DeviceBatch(pb, p, "cuda:0")
    dev.launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dev.launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    assert int((dev.out["status"][:pb.n_pairs] != 0).sum()) == 0
    g = capi.last_launch()
    blocks_cu = min((160 * 1024) // g["lds_block"], 32 // g["waves"])
    reg_cap = 12 if g["K"] <= 2 else 8 if g["K"] == 3 else 4
    bpp = 4.0 * pt["L"] + 48 + pb.max_hap_len / pt["R"]
    print(json.dumps(dict(point=pt, pairs=pb.n_pairs, ms=ms, cells_per_s=pb.cells / ms * 1e3, pairs_per_s=pb.n_pairs / ms * 1e3,
                          K=g["K"], D_build=g["D"] % 100, backpointers="hbm-scratch" if g["D"] >= 100 else "lds", scratch_MB=dev.ws_bytes / 1e6, waves_per_wg=g["waves"], lds_per_wave=g["lds_wave"], lds_per_wg=g["lds_block"],
                          waves_per_cu_by_lds=blocks_cu * g["waves"], waves_per_cu_by_regs=reg_cap, hbm_GBps_algorithmic=bpp * pb.n_pairs / ms / 1e6,
                          hbm_frac=bpp * pb.n_pairs / ms / 1e6 / 8000.0)))
    del dev
