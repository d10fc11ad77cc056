#!/usr/bin/env python3
"""Do two likelihood launches on two streams overlap on the GPU?  Two resident batches of N windows each: both on one stream, then one
per stream; wall time per pair of launches (HIP events on each stream + a host clock)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = capi.params_cli_defaults()
devs = [DeviceBatch(synth.generate(n, H=8, R=200, L=100, hap_len=120, seed=s), p, "cuda:0") for s in (1, 2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for d in devs:
    d.launch()
torch.cuda.synchronize()


def run(streams, reps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for d, st in zip(devs, streams):
            d.launch(stream=st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("%d windows per batch: two launches on one stream %.2f ms, on two streams %.2f ms (one launch alone: %.2f ms)"
      % (n, run([s[0], s[0]]), run(s), run([s[0]][:1] * 1) if False else 0.0))
t0 = time.perf_counter()
for _ in range(10):
    devs[0].launch(stream=s[0])
torch.cuda.synchronize()
print("one launch: %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
