#!/usr/bin/env python3
"""Which kind of raggedness costs what: the real-shaped batch of bench.py's `ragged` leg with one property after the other made uniform
(reads per window, read length, haplotypes per window, trimmed reads), haplotypes kept on the K = 2 tiling (116..126 bp) unless --all.
One JSON line per variant: cells/s and the launches with their durations (DD_LAUNCH_TIMING=1)."""
import json
import os
import sys

os.environ["DD_LAUNCH_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2400
narrow = {} if "--all" in sys.argv else dict(max_extra=0, max_indel=5)
variants = [("ragged", {}), ("no_trimmed", dict(trimmed=False)), ("reads=127", dict(fix_reads=127)), ("read_len=100", dict(fix_read_len=100)),
            ("haps=7", dict(fix_haps=7)), ("reads=127,len=100", dict(fix_reads=127, fix_read_len=100)),
            ("reads=127,len=100,haps=7,no_trimmed", dict(fix_reads=127, fix_read_len=100, fix_haps=7, trimmed=False))]
p = capi.params_cli_defaults()
for name, kw in variants:
    pb = synth.generate_ragged(n, **narrow, **kw)
    dev = DeviceBatch(pb, p, "cuda:0")
    for _ in range(2):
        dev.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dev.launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps(dict(variant=name, pairs=pb.n_pairs, cells=pb.cells, ms=round(ms, 2), cells_per_s=float("%.4g" % (pb.cells / ms * 1e3)),
                          launches=[dict(K=r["K"], ppw=r["pairs_per_wave"], bt="hbm" if r["gbt"] else "lds", fold=r["fold"], waves=r["waves"], split=r["split"],
                                         haps=r["n_haps"], reads=[r["min_read"], r["max_read"]], grid=r["grid"], dyn=r["dynamic"], rpw=r["reads_per_wave"], ms=r["us"] / 1e3) for r in capi.launch_log()])), flush=True)
