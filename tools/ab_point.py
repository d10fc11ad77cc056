#!/usr/bin/env python3
"""A/B helper: time one shape with the library given in DD_LIB_PATH (diagnostic builds of the same sources)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch
L, H, R, hap, mld, n = [int(v) for v in sys.argv[1:7]]
pb = synth.generate(n, H=H, R=R, L=L, hap_len=hap, seed=99)
p = capi.params_cli_defaults(); p.maxLengthDel = mld
dev = DeviceBatch(pb, p, "cuda:0")
dev.launch(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dev.launch(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
g = capi.last_launch()
print(os.path.basename(os.environ.get("DD_LIB_PATH", "default")), sys.argv[1:7], "gbt" if g["D"] >= 100 else "lds", "waves/wg", g["waves"],
      "ms %.2f" % min(ts), "cells/s %.3e" % (pb.cells / min(ts) * 1e3))
