"""N2 end to end on the GPU: `dindel_gpu` (host/dindel_gpu.cpp: detectIndels restated as prepare-N / compute / reduce-N) turns a
BAM this test writes, a window file and a haplotype fixture into a .glf.txt; `dindel_glf2vcf` turns that into a VCF.
Checked: the planted heterozygous deletion is called (genotype, allele, position, candidate flag); `qual` and the genotype
quality equal a recomputation from the ORACLE's log-likelihoods of the same reads (DInDel.cpp:3083-3118, :3238-3268) to the
six digits the table prints; windows that cannot be processed get the reference's skipped line; preparing windows ahead in
batches gives byte-identical output to one window at a time; the --faster model runs through the same loop.
Parity unpinned beyond the oracle comparison: the reference has no fixtures for the window loop."""
import ctypes as C
import json
import math
import os
import subprocess

import numpy as np
import pytest

from dindel_tgi_amd import capi, hostlib
from tests import _bamwriter as bw
from tests import _oracle
from tests.test_glf_vcf_cpu import GLF_COLUMNS, write_fasta

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dindel_tgi_amd", "host")


def add_logs(a, b):
    return a + math.log(1.0 + math.exp(b - a)) if a > b else b + math.log(1.0 + math.exp(a - b))


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("n2")
    rng = np.random.default_rng(77)
    r = list(rng.choice(list("ACGT"), 12000))
    for i in range(3, len(r)):                     # no homopolymer longer than 3: the VCF's hp filter stays out of the picture
        if r[i] == r[i - 1] == r[i - 2] == r[i - 3]:
            r[i] = "ACGT"[("ACGT".index(r[i]) + 1 + i % 3) % 4]
    ref = "".join(r)
    fasta = str(tmp / "ref.fa")
    write_fasta(fasta, [("20", ref)])
    windows, fixture, recs = [], [], []
    rid = 0
    # three windows with a heterozygous 2-bp deletion at offset 60, one homozygous 3-bp insertion, one window whose reference
    # haplotype is too short (hapSize error.), one without reads
    # (windows sit right of position 2,200: getReads fetches from leftPos - maxInsertSize - 200 in unsigned arithmetic, DInDel.cpp:928)
    spec = [(5000, "del", 0.5), (5400, "del", 0.5), (5800, "ins", 1.0), (6200, "del", 0.5), (6600, "short", 0.5), (9000, "empty", 0.0)]
    for wi, (left, kind, frac) in enumerate(spec, start=1):
        right = left + 120
        hap0 = ref[left:right + 1]
        if kind == "ins":
            insseq = "GAT"
            alt = ref[:left + 60] + insseq + ref[left + 60:]
            hap1 = hap0[:60] + insseq + hap0[60:]
            var = "+" + insseq
            v1 = "V I 60 %s 60 60 60 62 60 60 59 63" % var           # inserted bases 60..62 of the haplotype, flanked by 59 | 63
        else:
            alt = ref[:left + 60] + ref[left + 62:]
            hap1 = hap0[:60] + hap0[62:]
            var = "-" + hap0[60:62]
            v1 = "V I 60 %s 60 61 59 60 60 61 59 60" % var            # deletion between haplotype bases 59 | 60
        windows.append("20 %d %d %d,%s" % (left, right, left + 60, var))
        # A records: every haplotype base's offset on the window's reference sequence (hap.ml.hpos; -1 = inserted base)
        a0 = "A " + " ".join(map(str, range(121)))
        a1 = "A " + " ".join(map(str, list(range(60)) + ([-1, -1, -1] + list(range(60, 121)) if kind == "ins" else list(range(62, 121)))))
        if kind == "short":
            fixture += ["W %d %d %d" % (wi, left, right), "H ACG", "V I 60 *REF 60 60 60 60 60 60 60 60", "H " + hap1, v1]
        else:
            fixture += ["W %d %d %d" % (wi, left, right), "H " + hap0, a0, "V I 60 *REF 60 60 60 60 60 60 60 60", "V S 60 *REF 60 60 60 60 60 60 60 60",
                        "H " + hap1, v1, "V S 60 *REF 60 60 60 60 60 60 60 60"] + ([a1] if wi != 4 else [])     # window 4: the variant haplotype lacks its A record
        if kind == "empty":
            continue
        for _ in range(40):
            from_alt = rng.random() < frac
            p = int(rng.integers(left - 60, left + 75))
            if from_alt:
                cut = left + 60 - p                                     # read bases before the event
                if kind == "ins":
                    seq = alt[p:p + 100]
                    cigar = "100M" if cut <= 0 or cut >= 97 else "%dM3I%dM" % (cut, 97 - cut)
                    if cut <= 0:
                        continue
                else:
                    seq = alt[p:p + 100]
                    cigar = "100M" if cut <= 0 or cut >= 100 else "%dM2D%dM" % (cut, 100 - cut)
                    if cut <= 0:
                        p2 = p + 2                                      # a read right of the deletion sits two bases further on the reference
                        seq, p = alt[p:p + 100], p2
            else:
                seq, cigar = ref[p:p + 100], "100M"
            s = list(seq)
            if rng.random() < 0.1:
                k = int(rng.integers(0, 100)); s[k] = "ACGT"[("ACGT".index(s[k]) + 1) % 4]
            recs.append(dict(qname="q%04d" % rid, flag=int(rng.choice([0, 16])), pos=p, mapq=60, cigar=cigar, seq="".join(s), qual=[30] * 100,
                             mtid=-1, mpos=-1, isize=0, tags={}))
            rid += 1
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp / "reads.bam")
    bw.write_bam(bam, "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:20\tLN:12000\n", [("20", 12000)], [(0, r) for r in recs])
    vf, hf = str(tmp / "windows.txt"), str(tmp / "haps.txt")
    open(vf, "w").write("\n".join(windows) + "\n")
    open(hf, "w").write("\n".join(fixture) + "\n")
    subprocess.check_call(["make", "-s", "-C", HOST])
    return dict(tmp=tmp, ref=ref, fasta=fasta, bam=bam, vf=vf, hf=hf, spec=spec, recs=recs)


def run_driver(scene, prefix, *extra):
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = str(scene["tmp"] / prefix)
    subprocess.check_call([os.path.join(HOST, "dindel_gpu"), "--bamFile", scene["bam"], "--varFile", scene["vf"], "--hapFile", scene["hf"],
                           "--outputFile", out, "--quiet", *extra], env=env, stderr=subprocess.DEVNULL)
    lines = open(out + ".glf.txt").read().split("\n")
    assert lines[0].split(" ") == GLF_COLUMNS
    return out + ".glf.txt", [dict(zip(GLF_COLUMNS, l.split(" "))) for l in lines[1:] if l]


def test_driver_calls_match_oracle_recomputation(scene):
    path, rows = run_driver(scene, "a", "--batchWindows", "64")
    by_index = {}
    for r in rows:
        by_index.setdefault(int(r["index"]), []).append(r)
    # windows 5 / 6: skipped lines with the reference's messages (DInDel.cpp:1366-1395), coordinates of the window FILE
    assert [r["msg"] for r in by_index[5]] == ["error_hapSize_error."] and by_index[5][0]["lpos"] == "6600"
    assert [r["msg"] for r in by_index[6]] == ["error_too_few_reads"] and by_index[6][0]["analysis_type"] == "NA"
    lib = hostlib.load()
    lib.ddh_get_reads_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_double, C.c_char_p, C.c_int]
    p = capi.params_cli_defaults()
    hap_lines = open(scene["hf"]).read().split("\n")
    for wi, (left, kind, _f) in enumerate(scene["spec"][:4], start=1):
        dm = [r for r in by_index[wi] if r["analysis_type"] == "dip.map"]
        dp = [r for r in by_index[wi] if r["analysis_type"] == "dip"]
        assert len(dm) == 1 and len(dp) == 1 and dm[0]["msg"] == "ok", [(r["msg"], r["analysis_type"]) for r in by_index[wi]]
        haps = [l[2:] for l in hap_lines[hap_lines.index("W %d %d %d" % (wi, left, left + 120)):][:7] if l.startswith("H ")]
        out = C.create_string_buffer(1 << 24)
        win = (C.c_int * 2)(left, left + 120)
        prm = (C.c_int * 4)(10000, 500, 20, 0)
        assert lib.ddh_get_reads_json(scene["bam"].encode(), b"", b"20", win, 1, prm, 0.99, out, len(out)) > 0
        reads = json.loads(out.value.decode())[0]["reads"]
        # the oracle on the reads the window selected, in their order
        ll = [[_oracle.pair(h, r[6], [phred_q] * len(r[6]), r[2], int(r[7]), left, p, unmapped=bool(r[5]))[0].ll for r in reads]
              for h in haps for phred_q in [1.0 - 10 ** -3.0]]
        pp = {}
        for h1, h2 in ((0, 0), (0, 1), (1, 1)):
            s = 0.0
            for r in range(len(reads)):
                s += math.log(0.5) + add_logs(ll[h1][r], ll[h2][r])
            pp[(h1, h2)] = s + (0.0 if (h1, h2) == (0, 0) else math.log(1.0 / 10000.0))      # one candidate indel without a prior: log(priorIndel)
        ll_ref = pp[(0, 0)]
        best = max(((0, 1), (1, 1)), key=lambda k: pp[k])
        qual = -10.0 * (ll_ref - add_logs(pp[best], ll_ref)) / math.log(10.0)
        alt_best = max(v for k, v in pp.items() if k != best)            # every other pair has a different genotype at this site
        genoqual = -10.0 * (alt_best - add_logs(pp[best], alt_best)) / math.log(10.0)
        row = dm[0]
        assert row["qual"] == "%g" % qual and row["glf"] == "%s:%g" % ("0/1" if best == (0, 1) else "1/1", genoqual), (wi, row["qual"], qual)
        assert best == ((1, 1) if kind == "ins" else (0, 1))
        assert row["realigned_position"] == str(left + 60) and row["was_candidate_in_window"] == "1" and row["num_reads"] == str(len(reads))
        assert row["nref_all"] == ("+GAT" if kind == "ins" else "-" + scene["ref"][left + 60:left + 62])
        assert int(row["var_coverage_forward"]) + int(row["var_coverage_reverse"]) > 5
        assert dp[0]["nref_all"] == row["nref_all"] and int(dp[0]["nBQT"]) > 1000 and dp[0]["numOffAll"] == "0"
        assert dp[0]["glf"].count(":") == 3                            # 0/0, 0/1, 1/1 log-likelihoods


def test_driver_batching_is_exact_and_vcf_follows(scene):
    a = open(run_driver(scene, "b1", "--batchWindows", "1")[0]).read()
    b = open(run_driver(scene, "b64", "--batchWindows", "64")[0]).read()
    c = open(run_driver(scene, "b2", "--batchWindows", "2")[0]).read()
    assert a == b == c
    # batches prepared side by side (each worker with its own BAM handle and read buffer) and reduced on several threads
    d = open(run_driver(scene, "b2p", "--batchWindows", "2", "--prepareThreads", "4", "--reduceThreads", "3")[0]).read()
    e = open(run_driver(scene, "b1p", "--batchWindows", "1", "--prepareThreads", "1", "--reduceThreads", "1")[0]).read()
    assert a == d == e
    # the engines dealt out over a device list (here the same GPU twice: one box, one card)
    f = open(run_driver(scene, "b2d", "--batchWindows", "2", "--devices", "0,0", "--computeThreads", "3")[0]).read()
    assert a == f
    lst = str(scene["tmp"] / "glfs.txt")
    open(lst, "w").write(str(scene["tmp"] / "b64.glf.txt") + "\n")
    vcf = str(scene["tmp"] / "calls.vcf")
    subprocess.check_call([os.path.join(HOST, "dindel_glf2vcf"), "-i", lst, "-o", vcf, "-r", scene["fasta"], "-s", "S1"], stdout=subprocess.DEVNULL)
    body = [l.split("\t") for l in open(vcf).read().split("\n") if l and not l.startswith("#")]
    assert [(l[0], int(l[1])) for l in body] == [("20", 5060), ("20", 5460), ("20", 5860), ("20", 6260)]
    ref = scene["ref"]
    assert body[0][3] == ref[5059:5062] and body[0][4] == ref[5059] and body[0][9].startswith("0/1:")        # deletion: REF 3 bases, ALT the anchor
    assert body[2][3] == ref[5859] and body[2][4] == ref[5859] + "GAT" and body[2][9].startswith("1/1:")   # insertion
    assert all(l[6] == "PASS" for l in body)


def test_driver_skips_windows_over_the_hap_read_product(scene):
    """--maxHapReadProd (DInDel.cpp:395-399): a window with more haplotypes x reads than the limit is skipped with the reference's message
    and the others are called as before."""
    full = open(run_driver(scene, "hp_full")[0]).read().split("\n")
    n_reads = {int(l.split(" ")[1]): int(l.split(" ")[11]) for l in full[1:] if " dip.map " in l}
    assert len(set(n_reads.values())) > 1
    limit = 2 * min(n_reads.values()) + 1                      # two haplotypes per window: the thinnest window stays under the limit
    path, rows = run_driver(scene, "hp", "--maxHapReadProd", str(limit))
    by = {}
    for r in rows:
        by.setdefault(int(r["index"]), []).append(r)
    for wi, n in n_reads.items():
        if 2 * n > limit:
            assert [r["msg"] for r in by[wi]] == ["error_skipped_numhap_times_numread>%d" % limit], wi
        else:
            assert [l for l in full[1:] if l.split(" ")[1:2] == [str(wi)]] == [l for l in open(path).read().split("\n")[1:] if l.split(" ")[1:2] == [str(wi)]]


def test_driver_faster_model_runs_the_same_loop(scene):
    _path, rows = run_driver(scene, "f", "--faster")
    dm = [r for r in rows if r["analysis_type"] == "dip.map"]
    assert [int(r["index"]) for r in dm] == [1, 2, 3, 4] and all(float(r["qual"]) > 20 for r in dm)
    assert [r["msg"] for r in rows if r["index"] == "5"] == ["error_hapSize_error."]


def clip_to_window(pos, cigar, left, right):
    """The truth alignment as a realignment inside [left, right] can state it: read bases on reference positions outside the window's
    haplotypes hang off them (LO / RO) and come out soft-clipped; the position is that of the first base kept."""
    ops, ref, out, new_pos = bw.parse_cigar(cigar), pos, [], None
    for l, op in ops:
        if op == 0:                                           # M: split into clipped | kept | clipped
            lo = max(0, min(l, left - ref)); hi = max(0, min(l, ref + l - (right + 1)))
            for n, o in ((lo, 4), (l - lo - hi, 0), (hi, 4)):
                if n:
                    if o == 0 and new_pos is None:
                        new_pos = ref + lo
                    out.append([n, o])
            ref += l
        else:
            out.append([l, op])
            if op == 2:
                ref += l
    merged = []
    for n, o in out:
        if merged and merged[-1][1] == o:
            merged[-1][0] += n
        else:
            merged.append([n, o])
    return new_pos, "".join("%d%s" % (n, bw.CIGAR_OPS[o]) for n, o in merged)


def test_driver_writes_realigned_bam(scene):
    """--outputRealignedBAM (DInDel.cpp:589-620, :670-725): one BAM per window with every read of the window, realigned through the
    better haplotype of the most likely pair.  The sample's reads were cut from the two haplotypes, so for a read with enough sequence
    on both sides of the event the new record must say where it came from: position and CIGAR of the truth."""
    path, rows = run_driver(scene, "ra", "--outputRealignedBAM")
    assert [r["msg"] for r in rows if r["index"] == "4"][-1] == "error_Haplotype_has_not_been_aligned!"
    plain = open(run_driver(scene, "ra0")[0]).read().split("\n")
    with_bam = open(path).read().split("\n")
    # window 4 wrote its calls and then failed in getCIGAR: the skipped line follows them, as in the reference's catch
    assert [l for l in with_bam if not l.startswith("error_Haplotype")] == plain
    truth = {r["qname"]: r for r in scene["recs"]}
    header0, refs0, _ = bw.read_bam(scene["bam"])
    seen = 0
    for wi, (left, kind, _frac) in enumerate(scene["spec"], start=1):
        name = "%s.ra.%d_20_%d_%d.bam" % (str(scene["tmp"] / "ra"), wi, left + 20, left + 120 - 20)
        if wi >= 4:
            assert not os.path.exists(name)                  # getCIGAR threw / the window was skipped before
            continue
        header, refs, recs = bw.read_bam(name)
        assert (header, refs) == (header0, refs0)
        kept = [dict(zip(GLF_COLUMNS, l.split(" "))) for l in with_bam if l.split(" ")[1:2] == [str(wi)] and " dip.map " in l]
        assert len(recs) == int(kept[0]["num_reads"])
        n_event = 0
        for r in recs:
            t = truth[r["qname"]]
            assert (r["seq"], r["qual"], r["flag"], r["mapq"], r["mpos"]) == (t["seq"], t["qual"], t["flag"], t["mapq"], t["mpos"])
            assert sum(l for l, op in bw.parse_cigar(r["cigar"]) if op in (0, 1, 4)) == 100
            assert r["isize"] == r["pos"] - r["mpos"]
            cig = bw.parse_cigar(t["cigar"])
            if len(cig) == 3 and min(cig[0][0], cig[2][0]) >= 12:          # the event with 12 bases either side
                assert (r["pos"], r["cigar"]) == clip_to_window(t["pos"], t["cigar"], left, left + 120), (wi, r["qname"])
                n_event += 1
            elif len(cig) == 1 and t["pos"] + 100 <= left + 60 - 5:        # ends left of the event
                assert (r["pos"], r["cigar"]) == clip_to_window(t["pos"], "100M", left, left + 120), (wi, r["qname"])
        assert n_event >= 5
        seen += 1
    assert seen == 3
    # --faster has no realigned output, like the reference (`params.outputRealignedBAM && params.slower`)
    run_driver(scene, "raf", "--outputRealignedBAM", "--faster")
    assert not [f for f in os.listdir(str(scene["tmp"])) if f.startswith("raf.ra.")]


def test_driver_with_library_file_matches_oracle_recomputation(tmp_path):
    """--libFile switches the insert-size prior on (DInDel.cpp:4268-4272): read pairs, mapped reads with an unmapped mate and the
    unmapped mates themselves go through getReads (mates found by name), the packing of mate position / length / library, and the
    prior at the join.  The driver's qual and genotype quality equal a recomputation from the ORACLE's log-likelihoods of the reads
    the window selected, library table included."""
    from dindel_tgi_amd.batch import ReadRec, Window, pack
    rng = np.random.default_rng(99)
    r = list(rng.choice(list("ACGT"), 12000))
    for i in range(3, len(r)):
        if r[i] == r[i - 1] == r[i - 2] == r[i - 3]:
            r[i] = "ACGT"[("ACGT".index(r[i]) + 1 + i % 3) % 4]
    ref = "".join(r)
    left = 6000
    hap0 = ref[left:left + 121]
    hap1 = hap0[:60] + hap0[62:]
    alt = ref[:left + 60] + ref[left + 62:]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    recs = []

    def cut(p, from_alt):                            # (pos, cigar, seq) of a 100-bp read starting at reference position p
        if not from_alt:
            return p, "100M", ref[p:p + 100]
        c = left + 60 - p
        if c <= 0:
            return p, "100M", ref[p:p + 100]
        return p, ("100M" if c >= 100 else "%dM2D%dM" % (c, 100 - c)), alt[p:p + 100]

    for k in range(36):                              # proper pairs: first mate over the window, second one an insert size further right
        from_alt = k % 2 == 0
        p1 = int(rng.integers(left - 50, left + 55))
        p2 = p1 + int(rng.normal(300, 15))
        a, b = cut(p1, from_alt), cut(p2, False)
        recs.append(dict(qname="p%02d" % k, flag=99, pos=a[0], mapq=60, cigar=a[1], seq=a[2], qual=[30] * 100, mtid=0, mpos=b[0], isize=p2 - p1 + 100, tags={"RG": "g1"}))
        recs.append(dict(qname="p%02d" % k, flag=147, pos=b[0], mapq=60, cigar=b[1], seq=b[2], qual=[30] * 100, mtid=0, mpos=a[0], isize=-(p2 - p1 + 100), tags={"RG": "g1"}))
    for k in range(8):                               # a mapped read over the window whose mate did not map: the mate sits at the same position
        p1 = int(rng.integers(left - 40, left + 40))
        a = cut(p1, k % 2 == 0)
        rev = 16 if k % 4 < 2 else 0
        mate_seq = "".join(comp[c] for c in reversed(ref[p1 + 20:p1 + 96])) if k % 3 else "".join(rng.choice(list("ACGT"), 76))
        recs.append(dict(qname="u%02d" % k, flag=1 + 8 + 64 + rev, pos=a[0], mapq=50, cigar=a[1], seq=a[2], qual=[30] * 100, mtid=0, mpos=a[0], isize=0, tags={"RG": "g1"}))
        recs.append(dict(qname="u%02d" % k, flag=1 + 4 + 128 + (0 if k % 2 else 16) + (32 if rev else 0), pos=a[0], mapq=0, cigar="", seq=mate_seq, qual=[30] * 76,
                         mtid=0, mpos=a[0], isize=0, tags={"RG": "g1"}))
    for k in range(10):                              # single-end reads
        a = cut(int(rng.integers(left - 50, left + 55)), k % 2 == 0)
        recs.append(dict(qname="s%02d" % k, flag=int(rng.choice([0, 16])), pos=a[0], mapq=60, cigar=a[1], seq=a[2], qual=[30] * 100, mtid=-1, mpos=-1, isize=0, tags={}))
    recs.sort(key=lambda x: x["pos"])
    bam = str(tmp_path / "pairs.bam")
    bw.write_bam(bam, "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:20\tLN:12000\n@RG\tID:g1\tSM:s\tLB:libA\n", [("20", 12000)], [(0, x) for x in recs])
    counts = np.round(1000 * np.exp(-0.5 * ((np.arange(700) - 300) / 20.0) ** 2)).astype(int)
    libf = str(tmp_path / "libs.txt")
    open(libf, "w").write("#LIB libA\n" + "".join("%d %d\n" % (i, c) for i, c in enumerate(counts)))
    vf, hf = str(tmp_path / "w.txt"), str(tmp_path / "h.txt")
    open(vf, "w").write("20 %d %d %d,-%s\n" % (left, left + 120, left + 60, hap0[60:62]))
    open(hf, "w").write("\n".join(["W 1 %d %d" % (left, left + 120), "H " + hap0, "V I 60 *REF 60 60 60 60 60 60 60 60", "V S 60 *REF 60 60 60 60 60 60 60 60",
                                   "H " + hap1, "V I 60 -%s 60 61 59 60 60 61 59 60" % hap0[60:62], "V S 60 *REF 60 60 60 60 60 60 60 60"]) + "\n")
    subprocess.check_call(["make", "-s", "-C", HOST])
    scene = dict(tmp=tmp_path, bam=bam, vf=vf, hf=hf)
    _path, rows = run_driver(scene, "lib", "--libFile", libf)
    dm = [x for x in rows if x["analysis_type"] == "dip.map"]
    assert len(dm) == 1 and dm[0]["msg"] == "ok", rows
    # the reads the window selected, with the insert-size prior's inputs
    lib = hostlib.load()
    lib.ddh_get_reads_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_double, C.c_char_p, C.c_int]
    out = C.create_string_buffer(1 << 24)
    assert lib.ddh_get_reads_json(bam.encode(), libf.encode(), b"20", (C.c_int * 2)(left, left + 120), 1, (C.c_int * 4)(10000, 500, 20, 1), 0.99, out, len(out)) > 0
    sel = json.loads(out.value.decode())[0]["reads"]
    assert int(dm[0]["num_reads"]) == len(sel) and sum(x[5] for x in sel) >= 3 and int(dm[0]["num_unmapped_realigned"]) >= 0
    by_key = {(x["qname"], bool(x["flag"] & 4), x["pos"]): x for x in recs}
    # Library::calcProb (Library.hpp:78-128) of the histogram
    mode = max(i for i, c in enumerate(counts) if c == counts.max())
    maxins = min(25 * mode, len(counts))
    probs = np.maximum(counts[:maxins] / float(counts[:maxins].sum()), 1e-10)
    srt = np.sort(probs)
    acc, p95 = 0.0, float(srt[-1])
    for x in range(len(srt) - 1, 0, -1):
        acc += srt[x]
        if acc > 0.95:
            p95 = float(srt[x])
            break
    reads = []
    for q, pos, mq, mate_pos, mate_len, unmapped, seq, pstat in sel:
        x = by_key[(q, bool(unmapped), pos)]
        reads.append(ReadRec(seq, [1.0 - 10 ** -3.0] * len(seq), mq, int(pstat), unmapped=bool(unmapped), paired=bool(x["flag"] & 1), mate_unmapped=bool(x["flag"] & 8),
                             mate_reverse=bool(x["flag"] & 32), mate_same_tid=x["mtid"] == 0, mate_pos=mate_pos, mate_len=mate_len if x["flag"] & 1 else -1, lib=0))
    p = capi.params_cli_defaults()
    p.mapUnmappedReads = 1
    want = _oracle.batch(p, pack([Window(left, [hap0, hap1], reads)], libraries=[(probs, p95)]))
    R = len(reads)
    ll = [want["ll"][h * R:(h + 1) * R] for h in range(2)]
    pp = {}
    for h1, h2 in ((0, 0), (0, 1), (1, 1)):
        s = 0.0
        for k in range(R):
            s += math.log(0.5) + add_logs(ll[h1][k], ll[h2][k])
        pp[(h1, h2)] = s + (0.0 if (h1, h2) == (0, 0) else math.log(1.0 / 10000.0))
    best = max(((0, 1), (1, 1)), key=lambda k: pp[k])
    qual = -10.0 * (pp[(0, 0)] - add_logs(pp[best], pp[(0, 0)])) / math.log(10.0)
    alt_best = max(v for k, v in pp.items() if k != best)
    genoqual = -10.0 * (alt_best - add_logs(pp[best], alt_best)) / math.log(10.0)
    assert best == (0, 1)
    assert dm[0]["qual"] == "%g" % qual and dm[0]["glf"] == "0/1:%g" % genoqual, (dm[0]["qual"], qual, dm[0]["glf"], genoqual)
    # and the prior matters: without the library table the same reads give another number
    plain = capi.params_cli_defaults()
    ll0 = _oracle.batch(plain, pack([Window(left, [hap0, hap1], reads)]))["ll"]
    assert not np.array_equal(np.asarray(ll0), np.asarray(want["ll"]))


def test_driver_ragged_sample_is_batching_invariant(tmp_path):
    """Windows of 90-330 bp with 2-12 haplotypes (several lane tilings of the kernel in one batch), 20-400 reads of 60-150 bp with mixed
    base and mapping qualities (tools/n2_pipeline_bench.py --ragged): the .glf.txt does not depend on how the windows are cut into
    batches or on the thread counts, for either model."""
    import sys
    root = os.path.dirname(HOST.rstrip("/")).rsplit("/", 1)[0]
    d = str(tmp_path / "ragged")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "n2_pipeline_bench.py"), "--windows", "400", "--ragged", "--dir", d, "--procs", "2"],
                       capture_output=True, text=True)
    assert "generated 400 windows" in r.stdout, r.stdout + r.stderr
    scene = dict(tmp=tmp_path, bam=d + "/reads.bam", vf=d + "/windows.txt", hf=d + "/haps.txt")
    for model in ([], ["--faster"]):
        a = open(run_driver(scene, "rg_a", *model)[0]).read()
        b = open(run_driver(scene, "rg_b", "--batchWindows", "1", "--prepareThreads", "1", "--computeThreads", "1", "--reduceThreads", "1", *model)[0]).read()
        c = open(run_driver(scene, "rg_c", "--batchWindows", "23", "--prepareThreads", "3", "--computeThreads", "3", "--reduceThreads", "5", *model)[0]).read()
        assert a == b == c
        # engines that find further batches waiting put them into the same launch (--mergeBatches, default 4 / 2): off, and wide
        d1 = open(run_driver(scene, "rg_d", "--batchWindows", "5", "--mergeBatches", "1", *model)[0]).read()
        d2 = open(run_driver(scene, "rg_e", "--batchWindows", "5", "--mergeBatches", "9", "--computeThreads", "1", *model)[0]).read()
        assert a == d1 == d2
        rows = [l.split(" ") for l in a.split("\n")[1:] if l]
        assert len({l[1] for l in rows if l[2] == "dip.map"}) > 350                       # nearly every window is called


def test_window_loop_calls_the_simulated_variants(tmp_path):
    """BAM -> .glf.txt -> VCF on a sample whose reads were drawn from the reference and from ONE of each window's candidate haplotypes
    (tools/n2_pipeline_bench.py --vcf): nearly every window's VCF record is that variant, at its position, heterozygous."""
    import sys
    root = os.path.dirname(HOST.rstrip("/")).rsplit("/", 1)[0]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "n2_pipeline_bench.py"), "--windows", "500", "--dir", str(tmp_path / "s"), "--procs", "2", "--vcf"],
                       capture_output=True, text=True)
    last = [json.loads(l) for l in r.stdout.split("\n") if l.startswith("{")][-1]
    assert last["step"] == "glf2vcf" and last["windows"] == 500, r.stdout + r.stderr
    assert last["true_variant_called"] >= 485 and last["called_heterozygous"] == last["true_variant_called"] and last["vcf_records"] <= 510


def test_late_skip_with_two_pools_is_re_prepared(tmp_path):
    """--bamFiles with two pools and windows whose likelihood step throws ("hapSize error.": a 2-bp haplotype against maxLengthDel 5): the
    reference empties its read buffer behind such a window (DInDel.cpp:1404-1405), which changes the order of the next windows' reads inside
    mapping-quality ties.  The pipeline, which prepared those windows ahead, must write what the loop run window by window writes
    (--windowByWindow: the writer redoes every window itself, one after the other), and hand the same reads in the same order to every window."""
    from tests.test_n2_pools_cpu import DRIVER, _env, _scene
    s = _scene(tmp_path, n_ref=60000, n_reads=16000)
    bad = (4, 9, 10, s["n"] // 2)
    lines = open(s["hf"]).read().split("\n")
    out, w = [], 0
    for ln in lines:                                       # the first haplotype of the chosen windows becomes 2 bp long
        if ln.startswith("W "):
            w = int(ln.split()[1])
            first = True
        if ln.startswith("H ") and w in bad and first:
            ln, first = "H AC", False
        elif ln.startswith("H "):
            first = False
        out.append(ln)
    open(s["hf"], "w").write("\n".join(out))

    def run(tag, *extra):
        d = tmp_path / tag
        d.mkdir()
        env = _env()
        env["DINDEL_DUMP_READS"] = str(d / "w")
        r = subprocess.run([DRIVER, "--bamFiles", s["list"], "--varFile", s["vf"], "--hapFile", s["hf"], "--outputFile", str(d / "out")] + list(extra),
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(str(d / "out.glf.txt")).read(), [open(str(d / ("w.%d" % (i + 1)))).read() for i in range(s["n"])], r.stdout
    glf_truth, dumps_truth, _ = run("wbw", "--windowByWindow", "--batchWindows", "16", "--prepareThreads", "2")
    assert glf_truth.count("error_hapSize_error.") == len(bad)
    for batch, threads in ((5, 3), (64, 2)):
        glf, dumps, stdout = run("b%d" % batch, "--batchWindows", str(batch), "--prepareThreads", str(threads))
        assert "re-prepared behind late skips" in stdout
        assert dumps == dumps_truth and glf == glf_truth, "batches of %d windows" % batch
