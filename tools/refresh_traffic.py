#!/usr/bin/env python3
"""tools/refresh_traffic.py BENCH_JSON PMC_JSON [SOURCE_LABEL]

The un-profiled bench line of tools/profile_round.sh is written before the PMC passes of the same run, so its `roofline.traffic` still
points at the previous round-end file (or reads "stale" when the kernel sources changed since).  This fills it in from the PMC summary of
the same run — under bench.py's own rule: same kernel, same pairs per launch, and the summary's `source_id` equal to the hash of the
kernel sources of this tree (`capi.kernel_source_id`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dindel_tgi_amd import capi

bench_path, pmc_path = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else os.path.relpath(os.path.abspath(pmc_path), ROOT)
bench = json.load(open(bench_path))
pmc = json.load(open(pmc_path))
roof = bench["roofline"]
kernel_sub = "dd_faster_kernel" if "dd_faster_kernel" in roof.get("kernel", "") else "dd_hmm_kernel"
here = capi.kernel_source_id(kernel_sub)
n_pairs = bench["config"]["pairs_per_gpu"]
done = False
if pmc.get("source_id") == here and kernel_sub in json.dumps(pmc.get("kernel", "")):
    for cfg in pmc.values():
        if isinstance(cfg, dict) and cfg.get("pairs_per_launch") == n_pairs and "hbm_bytes_raw" in cfg:
            roof["traffic"] = cfg["hbm_bytes_raw"]
            roof["traffic_source"] = label + " (PMC passes of the same run)"
            done = True
if not done:
    sys.exit("refresh_traffic: %s does not match this tree's kernel (%s) or the line's %d pairs" % (pmc_path, here, n_pairs))
json.dump(bench, open(bench_path, "w"))
print("roofline.traffic = %.3e bytes per launch (%s)" % (roof["traffic"], roof["traffic_source"]))
