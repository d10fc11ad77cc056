#!/usr/bin/env python3
"""Throughput of the batched window loop (host/dindel_gpu: BAM -> reads -> GPU likelihoods -> diploidGLF -> .glf.txt) on a synthetic
sample: W windows of 120 bp, 400 bp apart, 8 candidate haplotypes each (reference + 7 single-indel variants), ~200 reads of 100 bp
per window drawn from the reference and one variant haplotype (a heterozygous site).  Writes the BAM / window / haplotype files under
--dir, runs the driver and prints windows/s with the driver's own stage split.

    python tools/n2_pipeline_bench.py [--windows 2000] [--reads 200] [--batch 256] [--dir /tmp/n2bench] [--faster]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests import _bamwriter as bw

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=2000)
ap.add_argument("--reads", type=int, default=200)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dir", default="/tmp/n2bench")
ap.add_argument("--faster", action="store_true")
ap.add_argument("--keep", action="store_true", help="reuse the files of an earlier run with the same parameters")
ap.add_argument("--sweep", action="store_true", help="run a list of batch sizes / thread counts instead of one configuration")
ap.add_argument("--extra", default="", help="further driver options, blank-separated")
args = ap.parse_args()
os.makedirs(args.dir, exist_ok=True)
bam, vf, hf = [os.path.join(args.dir, n) for n in ("reads.bam", "windows.txt", "haps.txt")]
tag = os.path.join(args.dir, "params.json")
want = dict(windows=args.windows, reads=args.reads)
if not (args.keep and os.path.exists(tag) and json.load(open(tag)) == want):
    t0 = time.time()
    rng = np.random.default_rng(2026)
    W, step, first = args.windows, 400, 5000
    n_ref = first + W * step + 5000
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_ref)]
    refs = ref.tobytes().decode()
    windows, fixture, recs = [], [], []
    rid = 0
    for wi in range(1, W + 1):
        left = first + (wi - 1) * step
        right = left + 120
        hap0 = refs[left:right + 1]
        offs = sorted(int(o) for o in rng.choice(np.arange(40, 81), 7, replace=False))
        haps, cands = [], []
        for o in offs:
            n = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                seq = hap0[o:o + n]
                haps.append((hap0[:o] + hap0[o + n:], o, "-" + seq, "V I %d -%s %d %d %d %d %d %d %d %d" % (o, seq, o, o + n - 1, o - 1, o, o, o + n - 1, o - 1, o)))
            else:
                seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
                haps.append((hap0[:o] + seq + hap0[o:], o, "+" + seq, "V I %d +%s %d %d %d %d %d %d %d %d" % (o, seq, o, o, o, o + n - 1, o, o, o - 1, o + n)))
            cands.append("%d,%s" % (left + o, haps[-1][2]))
        windows.append("20 %d %d %s" % (left, right, " ".join(cands)))
        fixture.append("W %d %d %d" % (wi, left, right))
        refs_at = lambda skip: ["V %s %d *REF %d %d %d %d %d %d %d %d" % (k, o, o, o, o, o, o, o, o, o) for o in offs if o != skip for k in "IS"]
        fixture += ["H " + hap0] + refs_at(-1)
        for h, o, _v, vline in haps:
            fixture += ["H " + h, vline, "V S %d *REF %d %d %d %d %d %d %d %d" % (o, o, o, o, o, o, o, o, o)] + refs_at(o)
        alt_h, alt_o, alt_v, _ = haps[int(rng.integers(0, 7))]
        alt_full = refs[:left] + alt_h + refs[right + 1:]
        dlen = len(alt_h) - len(hap0)
        for _ in range(args.reads):
            p = int(rng.integers(left - 60, left + 80))
            if rng.random() < 0.5:
                seq, cigar, pos = refs[p:p + 100], "100M", p
            else:
                seq = alt_full[p:p + 100]
                cut = left + alt_o - p
                pos = p
                if cut <= 0:
                    cigar, pos = "100M", p - dlen                     # right of the event: shifted on the reference
                    if dlen > 0 and cut > -dlen:
                        continue
                elif dlen < 0:
                    cigar = "100M" if cut >= 100 else "%dM%dD%dM" % (cut, -dlen, 100 - cut)
                else:
                    cigar = "100M" if cut + dlen >= 100 else "%dM%dI%dM" % (cut, dlen, 100 - cut - dlen)
            recs.append(dict(qname="q%07d" % rid, flag=int(rng.integers(0, 2)) * 16, pos=pos, mapq=60, cigar=cigar, seq=seq, qual=[30] * 100,
                             mtid=-1, mpos=-1, isize=0, tags={}))
            rid += 1
    recs.sort(key=lambda r: r["pos"])
    bw.write_bam(bam, "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:20\tLN:%d\n" % n_ref, [("20", n_ref)], [(0, r) for r in recs], block_bytes=60000)
    open(vf, "w").write("\n".join(windows) + "\n")
    open(hf, "w").write("\n".join(fixture) + "\n")
    json.dump(want, open(tag, "w"))
    print("generated %d windows, %d reads in %.1f s" % (W, len(recs), time.time() - t0), flush=True)
host = os.path.join(ROOT, "dindel_tgi_amd", "host")
subprocess.check_call(["make", "-s", "-C", host])
env = dict(os.environ)
try:
    import torch
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
except ImportError:
    pass
base = [os.path.join(host, "dindel_gpu"), "--bamFile", bam, "--varFile", vf, "--hapFile", hf, "--outputFile", os.path.join(args.dir, "out"), "--timing"]
if args.sweep:
    configs = [["--batchWindows", str(b), "--computeThreads", str(c), "--prepareThreads", str(p), "--reduceThreads", str(r), "--packThreads", str(k)] + f
               for f in ([], ["--faster"]) for (b, c, p, r, k) in ((256, 2, 6, 6, 4), (256, 2, 8, 4, 4), (256, 2, 10, 4, 4), (128, 2, 8, 4, 4), (256, 3, 8, 4, 2))]
else:
    configs = [["--batchWindows", str(args.batch)] + (["--faster"] if args.faster else []) + args.extra.split()]
for cfg in configs:
    for rep in range(2):
        t0 = time.time()
        out = subprocess.run(base + cfg, env=env, capture_output=True, text=True)
        dt = time.time() - t0
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(1)
        rows = open(os.path.join(args.dir, "out.glf.txt")).read().split("\n")
        calls = sum(1 for l in rows if " dip.map " in l)
        skipped = sum(1 for l in rows[1:] if l and not l.startswith("ok "))
        timing = [l for l in out.stdout.split("\n") if l.startswith("timing:")]
        print(json.dumps(dict(rep=rep, windows=args.windows, options=" ".join(cfg), seconds=round(dt, 3), windows_per_s=round(args.windows / dt, 1),
                              dip_map_lines=calls, skipped=skipped, driver=timing[-1] if timing else None)), flush=True)
