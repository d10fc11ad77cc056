#!/usr/bin/env python3
"""Does the launch plan (LDS back-pointer tile vs HBM scratch, capi.cpp make_plan) pick the faster build?  For a grid of
(read length, haplotype length, maxLengthDel) the default plan is timed against both forced variants (DD_FORCE_GBT=0/1,
where the shape allows them).  Prints one line per point and flags points where the default loses more than 5 %."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch


def time_point(pb, p):
    dev = DeviceBatch(pb, p, "cuda:0")
    for _ in range(4):                      # the points are a few milliseconds each and the host generates the next batch in between:
        dev.launch()                        # without this the first variant of a point is timed on a card that has just clocked down
    torch.cuda.synchronize()
    best = None
    for _rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            dev.launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        best = ms if best is None else min(best, ms)
    g = capi.last_launch()
    return best, ("hbm" if g["D"] >= 100 else "lds"), g["K"]


# haplotype lengths: every lane tiling near full (43 ... 763 after the variants' indels) and, since round 4, the lengths between
# the tilings' steps (62 / 126 / 190 bp), where real windows sit
HAPS = (40, 62, 77, 97, 120, 124, 127, 137, 147, 157, 180, 192, 197, 250, 380, 500, 760) if "--full" in sys.argv else (40, 120, 180, 250, 380, 500, 760)
bad = 0
for mld in (5, 10):
    for L in (36, 76, 100, 110, 120, 130, 140, 150, 250, 400, 700, 1000):
        for hap in HAPS:
            pairs_target = 2.5e5 * (100 * 120) / (L * hap)
            R = 100
            n = max(2, int(pairs_target / (4 * R)))
            pb = synth.generate(n, H=4, R=R, L=L, hap_len=hap, seed=5, max_indel=3)
            p = capi.params_cli_defaults(); p.maxLengthDel = mld
            res = {}
            for tag, env in (("default", None), ("lds", "0"), ("hbm", "1")):
                if env is None:
                    os.environ.pop("DD_FORCE_GBT", None)
                else:
                    os.environ["DD_FORCE_GBT"] = env
                ms, kind, K = time_point(pb, p)
                if tag != "default" and kind != tag:
                    continue                      # that variant does not exist for this shape
                res[tag] = (ms, kind, K)
            os.environ.pop("DD_FORCE_GBT", None)
            best = min(v[0] for k, v in res.items() if k != "default")
            loss = res["default"][0] / best - 1.0
            flag = "  <-- default loses %.0f %%" % (100 * loss) if loss > 0.05 else ""
            bad += loss > 0.05
            print(json.dumps(dict(mld=mld, L=L, hap=pb.max_hap_len, pairs=pb.n_pairs, K=res["default"][2], default=res["default"][1],
                                  ms={k: round(v[0], 2) for k, v in res.items()}, cells_per_s=float("%.3g" % (pb.cells / res["default"][0] * 1e3)))) + flag, flush=True)
print("points where the default plan loses > 5 %:", bad)
