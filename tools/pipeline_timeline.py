#!/usr/bin/env python3
"""GPU timeline of the window loop from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- dindel_tgi_amd/host/dindel_gpu ...
    python3 tools/pipeline_timeline.py DIR
prints how long some kernel was running between the first and the last kernel, the idle gaps (count / total / largest), and per kernel
name the launches, summed and average duration."""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = [r for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) for r in csv.DictReader(open(f))]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ks[0][0], max(e for _s, e, _n in ks)
busy, gaps, cur_s, cur_e = 0, [], ks[0][0], ks[0][1]
for s, e, _n in ks[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels: %d; first start -> last end %.1f ms; some kernel running %.1f ms (%.1f %%); idle %.1f ms in %d gaps (largest %.2f ms, %d gaps > 0.1 ms holding %.1f ms)"
      % (len(ks), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(gaps) / 1e6, len(gaps), max(gaps) / 1e6 if gaps else 0,
         sum(1 for g in gaps if g > 1e5), sum(g for g in gaps if g > 1e5) / 1e6))
names = {}
for s, e, n in ks:
    k = n.split("(")[0][:70]
    a = names.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
for n, (c, t) in sorted(names.items(), key=lambda x: -x[1][1])[:8]:
    print("  %-70s %6d launches %9.2f ms summed %8.3f ms avg" % (n, c, t / 1e6, t / 1e6 / c))
# overlap: time with >= 2 kernels in flight
ev = sorted([(s, 1) for s, _e, _n in ks] + [(e, -1) for _s, e, _n in ks])
depth, last, two = 0, ev[0][0], 0
for t, k in ev:
    if depth >= 2:
        two += t - last
    depth += k
    last = t
print("two or more kernels in flight: %.1f ms" % (two / 1e6))
if len(sys.argv) > 2:                                    # a slice of the timeline, one line per kernel: python3 tools/pipeline_timeline.py DIR FROM_MS TO_MS
    lo, hi = float(sys.argv[2]) * 1e6 + t0, float(sys.argv[3]) * 1e6 + t0
    print("columns:", ",".join(rows[0].keys()))
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= s <= hi:
            print("%10.3f %10.3f ms  q=%s  %s  grid=%s wg=%s" % ((s - t0) / 1e6, (e - t0) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:40],
                                                            r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
