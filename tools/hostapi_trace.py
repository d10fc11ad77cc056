#!/usr/bin/env python3
"""Timeline of one LikelihoodEngine::computeLikelihoodsBatch call on the GPU: run under
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/hostapi_trace.py run
then
    python3 tools/hostapi_trace.py report DIR
prints, for the last call, when the first kernel started and the last ended, the time no kernel was running in between, and the
copies before / after (what the call costs on top of its kernels)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    from dindel_tgi_amd import hostlib
    r = hostlib.bench_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 10000, reps=1)
    print(r)
else:
    d = sys.argv[2]
    kern = [r for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) for r in csv.DictReader(open(f))]
    cop = [r for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True) for r in csv.DictReader(open(f))]
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kern)
    cs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in cop)
    # the last call = the kernels after the largest gap
    gaps = [(ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)]
    cut = max(gaps)[1] + 1 if gaps else 0
    last = ks[cut:]
    t0, t1 = last[0][0], max(e for _s, e, _n in last)
    busy, cur_s, cur_e = 0, last[0][0], last[0][1]
    for s, e, _n in last[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("kernels of the last call: %d, first start -> last end %.2f ms, some kernel running %.2f ms, idle in between %.2f ms" %
          (len(last), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    names = {}
    for s, e, n in last:
        names[n[:60]] = names.get(n[:60], 0) + (e - s)
    for n, t in sorted(names.items(), key=lambda x: -x[1])[:5]:
        print("  %-60s %.2f ms summed" % (n, t / 1e6))
    prev_end = ks[cut - 1][1] if cut else t0
    before = [(s, e, k) for s, e, k in cs if prev_end < s < t0]
    after = [(s, e, k) for s, e, k in cs if s >= t0]
    if before:
        print("copies before the first kernel: %d, %.2f ms from first start to last end" % (len(before), (max(e for _s, e, _k in before) - before[0][0]) / 1e6))
        print("  first copy start -> first kernel start: %.2f ms" % ((t0 - before[0][0]) / 1e6))
    if after:
        print("copies after the first kernel started: %d, last copy ends %.2f ms after the last kernel" % (len(after), (max(e for _s, e, _k in after) - t1) / 1e6))
