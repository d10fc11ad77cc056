#!/usr/bin/env python3
"""Throughput of the batched window loop (host/dindel_gpu: BAM -> reads -> GPU likelihoods -> diploidGLF -> .glf.txt) on a synthetic
sample: W windows of 120 bp, 400 bp apart, 8 candidate haplotypes each (reference + 7 single-indel variants), ~200 reads of 100 bp
per window drawn from the reference and one variant haplotype (a heterozygous site).  Writes the BAM / window / haplotype files under
--dir, runs the driver and prints windows/s with the driver's own stage split.

    python tools/n2_pipeline_bench.py [--windows 2000] [--reads 200] [--batch 256] [--dir /tmp/n2bench] [--faster]
"""
import argparse
import hashlib
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
ap.add_argument("--vcf", action="store_true", help="afterwards: dindel_glf2vcf on the .glf.txt, and the calls against the variants the reads were drawn with")
ap.add_argument("--ragged", action="store_true", help="windows of 90-330 bp, 2-12 haplotypes, 20-400 reads of 60-150 bp each (default: uniform 120 bp / 8 / 200 x 100 bp)")
ap.add_argument("--procs", type=int, default=min(16, os.cpu_count() or 1), help="processes writing the sample")
args = ap.parse_args()
os.makedirs(args.dir, exist_ok=True)
bam, vf, hf = [os.path.join(args.dir, n) for n in ("reads.bam", "windows.txt", "haps.txt")]
tag = os.path.join(args.dir, "params.json")
want = dict(windows=args.windows, reads=args.reads, ragged=bool(args.ragged))
FIRST, STEP, BLOCK = 5000, 400, 60000


def reference(W):
    n_ref = FIRST + W * STEP + 5000
    return np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(2026).integers(0, 4, n_ref)].tobytes().decode()


def gen_chunk(job):
    """Windows [w0, w1) (1-based): their window-file lines, fixture lines and reads — the reads already encoded as BAM records, sorted,
    and compressed into BGZF blocks of BLOCK bytes that belong to this chunk alone (the parent only concatenates)."""
    w0, w1, W, n_reads, ragged = job
    refs = reference(W)
    windows, fixture, blobs, pos, end, truth = [], [], [], [], [], []
    for wi in range(w0, w1):
        rng = np.random.default_rng([2026, wi])
        left = FIRST + (wi - 1) * STEP
        width = int(rng.integers(90, 331)) if ragged else 120
        n_var = int(rng.integers(1, 12)) if ragged else 7
        n_here = int(rng.integers(20, 401)) if ragged else n_reads
        right = left + width
        hap0 = refs[left:right + 1]
        offs = sorted(int(o) for o in rng.choice(np.arange(width // 3, 2 * width // 3 + 1), n_var, replace=False))
        haps, cands = [], []
        for o in offs:
            n = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                seq = hap0[o:o + n]
                haps.append((hap0[:o] + hap0[o + n:], o, "-" + seq, "V I %d -%s %d %d %d %d %d %d %d %d" % (o, seq, o, o + n - 1, o - 1, o, o, o + n - 1, o - 1, o),
                             list(range(o)) + list(range(o + n, len(hap0)))))
            else:
                seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
                haps.append((hap0[:o] + seq + hap0[o:], o, "+" + seq, "V I %d +%s %d %d %d %d %d %d %d %d" % (o, seq, o, o, o, o + n - 1, o, o, o - 1, o + n),
                             list(range(o)) + [-1] * n + list(range(o, len(hap0)))))
            cands.append("%d,%s" % (left + o, haps[-1][2]))
        windows.append("20 %d %d %s" % (left, right, " ".join(cands)))
        fixture.append("W %d %d %d" % (wi, left, right))
        refs_at = lambda skip: ["V %s %d *REF %d %d %d %d %d %d %d %d" % (k, o, o, o, o, o, o, o, o, o) for o in offs if o != skip for k in "IS"]
        fixture += ["H " + hap0, "A " + " ".join(map(str, range(len(hap0))))] + refs_at(-1)      # A: the haplotype's own alignment to the window (hap.ml.hpos)
        for h, o, _v, vline, amap in haps:
            fixture += ["H " + h, "A " + " ".join(map(str, amap)), vline, "V S %d *REF %d %d %d %d %d %d %d %d" % (o, o, o, o, o, o, o, o, o)] + refs_at(o)
        alt_h, alt_o, alt_v, _, _amap = haps[int(rng.integers(0, n_var))]
        truth.append("%d %d %d %s" % (wi, left + alt_o, len(alt_h) - len(hap0), alt_v))      # the heterozygous variant of this window
        alt_full = refs[left - 200:left] + alt_h + refs[right + 1:right + 400]      # the alternative chromosome around the window
        dlen = len(alt_h) - len(hap0)
        mine = []
        for k in range(n_here):
            L = int(rng.integers(60, 151)) if ragged else 100
            p = int(rng.integers(left - 60, left + width - 40))
            if rng.random() < 0.5:
                seq, cigar, at = refs[p:p + L], "%dM" % L, p
            else:
                seq = alt_full[p - (left - 200):p - (left - 200) + L]
                cut = left + alt_o - p
                at = p
                if cut <= 0:
                    cigar, at = "%dM" % L, p - dlen                    # right of the event: shifted on the reference
                    if dlen > 0 and cut > -dlen:
                        continue
                elif dlen < 0:
                    cigar = "%dM" % L if cut >= L else "%dM%dD%dM" % (cut, -dlen, L - cut)
                else:
                    cigar = "%dM" % L if cut + dlen >= L else "%dM%dI%dM" % (cut, dlen, L - cut - dlen)
            quals = [int(q) for q in rng.integers(10, 41, L)] if ragged else [30] * L
            mine.append(dict(qname="q%d_%d" % (wi, k), flag=int(rng.integers(0, 2)) * 16, pos=at, mapq=int(rng.choice([20, 40, 60])) if ragged else 60, cigar=cigar, seq=seq,
                             qual=quals, mtid=-1, mpos=-1, isize=0, tags={}))
        mine.sort(key=lambda r: r["pos"])
        for r in mine:
            b, e = bw.encode_record(0, r)
            blobs.append(b); pos.append(r["pos"]); end.append(e)
    u = b"".join(blobs)
    lens = np.array([len(b) for b in blobs], dtype=np.int64)
    comp, cstart = [], []
    total = 0
    for o in range(0, len(u), BLOCK):
        blk = bw.bgzf_block(u[o:o + BLOCK])
        cstart.append(total); total += len(blk); comp.append(blk)
    return windows, fixture, b"".join(comp), np.array(cstart, dtype=np.int64), np.array(pos, dtype=np.int64), np.array(end, dtype=np.int64), lens, truth


def write_sample(W, n_reads, procs, ragged=False):
    import multiprocessing as mp
    import struct
    n_ref = FIRST + W * STEP + 5000
    per = max(1, (W + 4 * procs - 1) // (4 * procs))
    jobs = [(w, min(w + per, W + 1), W, n_reads, ragged) for w in range(1, W + 1, per)]
    with mp.Pool(procs) as pool:
        parts = pool.map(gen_chunk, jobs)
    header_text = "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:20\tLN:%d\n" % n_ref
    hdr = b"BAM\1" + struct.pack("<I", len(header_text)) + header_text.encode() + struct.pack("<I", 1) + struct.pack("<I", 3) + b"20\0" + struct.pack("<I", n_ref)
    out = [bw.bgzf_block(hdr)]
    c0 = len(out[0])
    beg_v, end_v, pos_all, end_all = [], [], [], []
    starts = []                                   # compressed offset of each chunk
    for (_w, _f, comp, cstart, pos, end, lens, _t) in parts:
        starts.append(c0)
        c0 += len(comp)
    eof_c = c0
    for k, (_w, _f, comp, cstart, pos, end, lens, _t) in enumerate(parts):
        out.append(comp)
        ub = np.concatenate([[0], np.cumsum(lens)[:-1]])           # uncompressed start of each record inside the chunk
        ue = ub + lens
        ulen = int(ue[-1]) if len(ue) else 0
        nxt = starts[k + 1] if k + 1 < len(parts) else eof_c
        voff = lambda uo: np.where(uo >= ulen, nxt << 16, ((starts[k] + cstart[np.minimum(uo // BLOCK, len(cstart) - 1)]) << 16) | (uo % BLOCK))
        beg_v.append(voff(ub)); end_v.append(voff(ue)); pos_all.append(pos); end_all.append(end)
    out.append(bw.bgzf_block(b""))
    with open(bam, "wb") as f:
        for o in out:
            f.write(o)
    beg_v, end_v, pos_all, end_all = [np.concatenate(x) for x in (beg_v, end_v, pos_all, end_all)]
    # BAI: bins (reg2bin) with chunks = runs of records adjacent in the file, and the 16-kb linear index
    e1 = end_all - 1
    bins = np.zeros(len(pos_all), dtype=np.int64)
    done = np.zeros(len(pos_all), dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((pos_all >> shift) == (e1 >> shift))
        bins[hit] = base + (pos_all[hit] >> shift)
        done |= hit
    order = np.lexsort((np.arange(len(bins)), bins))
    sb, si = bins[order], order
    brk = np.concatenate([[True], (sb[1:] != sb[:-1]) | (si[1:] != si[:-1] + 1)])
    run_start = np.flatnonzero(brk)
    run_end = np.concatenate([run_start[1:], [len(sb)]]) - 1
    bai = bytearray(b"BAI\1" + struct.pack("<I", 1))
    run_bins = sb[run_start]
    ubins, first_run = np.unique(run_bins, return_index=True)
    counts = np.diff(np.concatenate([first_run, [len(run_bins)]]))
    bai += struct.pack("<I", len(ubins))
    for b, f0, n in zip(ubins, first_run, counts):
        bai += struct.pack("<II", int(b), int(n))
        for j in range(f0, f0 + n):
            bai += struct.pack("<QQ", int(beg_v[si[run_start[j]]]), int(end_v[si[run_end[j]]]))
    n_intv = int(e1.max() >> 14) + 1
    lin = np.full(n_intv, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(lin, pos_all >> 14, beg_v)
    np.minimum.at(lin, e1 >> 14, beg_v)
    last = 0
    bai += struct.pack("<I", n_intv)
    for w in range(n_intv):
        if lin[w] != np.iinfo(np.int64).max:
            last = int(lin[w])
        bai += struct.pack("<Q", last)
    open(bam + ".bai", "wb").write(bytes(bai))
    open(vf, "w").write("\n".join(l for p in parts for l in p[0]) + "\n")
    open(hf, "w").write("\n".join(l for p in parts for l in p[1]) + "\n")
    open(os.path.join(args.dir, "truth.txt"), "w").write("\n".join(l for p in parts for l in p[7]) + "\n")
    refs = reference(W)                                          # FASTA + .fai for the glf -> VCF step
    with open(os.path.join(args.dir, "ref.fa"), "w") as f:
        f.write(">20\n")
        off = f.tell()
        f.write("\n".join(refs[i:i + 60] for i in range(0, len(refs), 60)) + "\n")
    open(os.path.join(args.dir, "ref.fa.fai"), "w").write("20\t%d\t%d\t60\t61\n" % (len(refs), off))
    return len(pos_all)


if not (args.keep and os.path.exists(tag) and json.load(open(tag)) == want):
    t0 = time.time()
    n_written = write_sample(args.windows, args.reads, args.procs, args.ragged)
    json.dump(want, open(tag, "w"))
    print("generated %d windows, %d reads in %.1f s" % (args.windows, n_written, time.time() - t0), flush=True)
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
        text = open(os.path.join(args.dir, "out.glf.txt")).read()
        rows = text.split("\n")
        calls = sum(1 for l in rows if " dip.map " in l)
        skipped = sum(1 for l in rows[1:] if l and not l.startswith("ok "))
        timing = [l for l in out.stdout.split("\n") if l.startswith("timing:")]
        print(json.dumps(dict(rep=rep, windows=args.windows, options=" ".join(cfg), seconds=round(dt, 3), windows_per_s=round(args.windows / dt, 1),
                              dip_map_lines=calls, skipped=skipped, md5=hashlib.md5(text.encode()).hexdigest(), driver=timing[-1] if timing else None)), flush=True)

if args.vcf:
    lst = os.path.join(args.dir, "glf_files.txt")
    open(lst, "w").write(os.path.join(args.dir, "out.glf.txt") + "\n")
    vcf = os.path.join(args.dir, "calls.vcf")
    t0 = time.time()
    subprocess.check_call([os.path.join(host, "dindel_glf2vcf"), "-i", lst, "-o", vcf, "-r", os.path.join(args.dir, "ref.fa"), "-s", "SAMPLE"], stdout=subprocess.DEVNULL)
    dt = time.time() - t0
    calls = {}
    for l in open(vcf):
        if l.startswith("#"):
            continue
        c = l.rstrip("\n").split("\t")
        calls.setdefault(int(c[1]), []).append((len(c[4]) - len(c[3]), c[9].split(":")[0], c[6]))
    truth = [l.split() for l in open(os.path.join(args.dir, "truth.txt")) if l.strip()]
    found = het = 0
    for _wi, pos, dlen, _v in truth:
        hit = [x for x in calls.get(int(pos), []) if x[0] == int(dlen)]
        found += bool(hit)
        het += bool([x for x in hit if x[1] == "0/1"])
    n_calls = sum(len(v) for v in calls.values())
    print(json.dumps(dict(step="glf2vcf", seconds=round(dt, 3), vcf_records=n_calls, windows=len(truth), true_variant_called=found, called_heterozygous=het,
                          other_records=n_calls - found)), flush=True)
