"""C++ host adapter on the GPU: LikelihoodEngine::computeLikelihoods (mirror of DetInDel::computeLikelihoods,
DInDel.cpp:1707-1739) against the golden vectors and the oracle, including the reference's error strings."""
import json
import os

import numpy as np
import pytest

from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import ReadRec, Window
from tests import _host, _oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))["cases"]


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_engine_on_kat(case):
    p = capi.dd_params.from_dict(case["params"])
    res = _host.compute_window([case["hap"]], [case["read"]], [case["q"]], [case["mapQual"]], [float(case["pos"])], [0],
                               case["hapStart"], p)
    ml = res["liks"][0][0]
    for k in ("ll", "llOn", "llOff"):
        if k in case:
            assert ml[k] == pytest.approx(case[k], rel=1e-13, abs=0)
    if "hpos" in case:
        assert ml["hpos"] == case["hpos"]
    if "indels" in case:
        assert [[i[0], i[1]] for i in ml["indels"]] == case["indels"]
    if "snps" in case:
        assert ml["snps"] == case["snps"]
    for k in ("offHap", "offHapHMQ", "nBQT", "numMismatch"):
        if k in case:
            assert ml[k] == case[k]
    assert res["onHap"] == [0 if ml["offHapHMQ"] else 1]


def test_engine_window_matches_oracle_and_onhap():
    H = "ACGTTGCATGCCGATAGGCTTAACCGGTTTTTTACGATCGATGCAAGTCCGTA"
    haps = [H, H[:25] + H[27:], H[:30] + "GG" + H[30:]]
    reads = [H[10:40], H[5:25] + H[27:45], "GTCA" * 7 + "GT", H[20:30] + "GG" + H[30:50], "TTGACCA" + H[0:25]]
    quals = [0.999, 0.99, 0.999, 0.9999, 0.999]
    mapq = [0.9999, 0.999, 0.9, 0.9999, 0.99]
    pos = [1010.0, 1005.0, 1010.0, 1020.0, 993.0]
    p = capi.params_cli_defaults()
    res = _host.compute_window(haps, reads, quals, mapq, pos, [0, 0, 0, 0, 1], 1000, p)
    on = [0] * len(reads)
    for h, hap in enumerate(haps):
        for r, read in enumerate(reads):
            o, hpos = _oracle.pair(hap, read, quals[r], mapq[r], int(pos[r]), 1000, p, unmapped=(r == 4))
            ml = res["liks"][h][r]
            assert ml["ll"] == o.ll and ml["llOn"] == o.llOn and ml["llOff"] == o.llOff
            assert ml["hpos"] == hpos
            assert (ml["offHap"], ml["offHapHMQ"]) == (o.offHap, o.offHapHMQ)
            _oracle.assert_record_variants(ml, o, hap, read, (h, r))
            if not o.offHapHMQ:
                on[r] = 1
    assert res["onHap"] == on


def test_engine_throws_reference_strings():
    p = capi.params_cli_defaults()
    res = _host.compute_window(["ACG"], ["ACGT"], [0.999], [0.9999], [0.0], [0], 0, p)
    assert res == {"throw": "hapSize error."}


def test_engine_filter_haplotypes_matches_reference_loop():
    """computeLikelihoods + filterHaplotypes (DInDel.cpp:1932-2100) on the GPU flags vs a Python restatement of the
    reference loop run on the oracle's hpos."""
    rng = np.random.default_rng(3)
    ref = "".join(rng.choice(list("ACGT"), 100))
    hapD = ref[:48] + ref[51:]                      # 3-bp deletion: flanks 47 | 48 in the haplotype
    hapI = ref[:55] + "GATTA" + ref[55:]            # 5-bp insertion at 55..59: flanks 54 | 60
    haps = [ref, hapD, hapI, ref[:48] + ref[51:55] + "GATTA" + ref[55:]]
    hap_vars = [[], [(48, 1, 47, 48)], [(55, 2, 54, 60)], [(48, 1, 47, 48), (52, 2, 51, 57)]]
    reads, quals, mapq, pos, flags = [], [], [], [], []
    for i in range(40):
        src = haps[i % 4]
        off = int(rng.integers(0, len(src) - 40))
        s = list(src[off:off + 40])
        if i % 7 == 0:
            s[int(rng.integers(0, 40))] = "A"
        reads.append("".join(s)); quals.append(0.999); mapq.append(0.9999); pos.append(1000.0 + off)
        flags.append([0, 2, 1, 5][i % 4])           # fwd, reverse, unmapped(mate fwd), unmapped(mate reverse)
    p = capi.params_cli_defaults()
    res = _host.filter_window(haps, hap_vars, reads, quals, mapq, pos, flags, 1000, p)
    # ---- Python restatement on oracle results ----
    nh = len(haps)
    filtered = [0] * nh
    cover = {}
    for h, hap in enumerate(haps):
        sel = []
        al = {}
        for r, read in enumerate(reads):
            o, hpos = _oracle.pair(hap, read, quals[r], mapq[r], int(pos[r]), 1000, p, unmapped=bool(flags[r] & 1))
            al[r] = hpos
            if not o.offHapHMQ and o.numIndels == 0:
                sel.append(r)
        all_cov = True
        for (key, kind, lf, rfl) in hap_vars[h]:
            pav = (key, "-A" if kind == 1 else "+A")
            cover.setdefault(pav, [set() for _ in range(2 * nh)])
            left, right = lf - p.padCover, rfl + p.padCover
            ln = right - left + 1
            covered = False
            for r in sel:
                strand = (0 if (flags[r] & 4) else 1) if (flags[r] & 1) else (1 if (flags[r] & 2) else 0)
                c, nmm = set(), 0
                for b, hb in enumerate(al[r]):
                    if left <= hb <= right:
                        c.add(hb)
                        nmm += (hap[hb] != reads[r][b]) and (kind == 2 or hap[hb] != "N")
                if len(c) >= ln and nmm <= p.maxMismatch:
                    cover[pav][h + strand * nh].add(r)
                    covered = True
            if not covered:
                all_cov = False
                break
        if not all_cov:
            filtered[h] = 1
    want_cov = []
    for pav in sorted(cover):
        rf, rr = set(), set()
        for h in range(nh):
            if not filtered[h]:
                rf |= cover[pav][h]; rr |= cover[pav][h + nh]
        want_cov.append([pav[0], pav[1], len(rf), len(rr)])
    assert res["filtered"] == filtered
    assert res["coverage"] == want_cov
    assert sum(c[2] + c[3] for c in want_cov) > 0


# ---------------- --faster model: LikelihoodEngine::computeLikelihoodsFaster (DInDel.cpp:1790-1833) ----------------
@pytest.mark.parametrize("case", [c for c in KAT if "ll_fast" in c], ids=lambda c: c["name"])
def test_engine_faster_on_kat(case):
    p = capi.dd_params.from_dict(case["params"])
    res = _host.compute_window([case["hap"]], [case["read"]], [case["q"]], [case["mapQual"]], [float(case["pos"])], [0],
                               case["hapStart"], p, faster=True)
    ml = res["liks"][0][0]
    assert ml["ll"] == pytest.approx(case["ll_fast"], rel=1e-14, abs=0)
    assert (ml["offHap"], ml["offHapHMQ"]) == (0, 0) and res["onHap"] == [1]


def test_engine_faster_window_matches_oracle():
    """ll, hpos, firstBase/lastBase and the variants ObservationModelS::reportVariants lists (Faster.cpp:579-681)."""
    H = "ACGTTGCATGCCGATAGGCTTAACCGGTTTTTTACGATCGATGCAAGTCCGTAGGATCCATTAGCAGGATACCAGTTAG"
    haps = [H, H[:25] + H[28:], H[:30] + "GGAT" + H[30:]]
    reads = [H[10:50], H[5:25] + H[28:55], "GTCA" * 9 + "GT", H[20:30] + "GGAT" + H[30:60], "TTGACCA" + H[0:35], H[40:] + "ACGTACG",
             H[12:30] + "T" + H[31:52]]
    quals = [0.999, 0.99, 0.999, 0.9999, 0.999, 0.999, 0.9999]
    mapq = [0.9999, 0.999, 0.9, 0.9999, 0.99, 0.9999, 0.999]
    pos = [1010.0, 1005.0, 1010.0, 1020.0, 993.0, 1040.0, 1012.0]
    for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
        res = _host.compute_window(haps, reads, quals, mapq, pos, [0] * len(reads), 1000, p, faster=True)
        assert res["onHap"] == [1] * len(reads)
        n_indel = 0
        for h, hap in enumerate(haps):
            for r, read in enumerate(reads):
                o, hpos = _oracle.pair_fast(hap, read, quals[r], mapq[r], int(pos[r]), 1000, p)
                ml = res["liks"][h][r]
                assert ml["ll"] == o.ll and ml["hpos"] == hpos
                assert (ml["firstBase"], ml["lastBase"]) == (o.firstBase, o.lastBase)
                assert (ml["offHap"], ml["offHapHMQ"], ml["numIndels"], ml["nBQT"]) == (0, 0, 0, 0)
                _oracle.assert_record_variants(ml, o, hap, read, (h, r))   # keys, strings, haplotype and read coordinates
                n_indel += o.n_indel
        assert n_indel > 0


def test_engine_faster_throws_reference_strings():
    p = capi.params_cli_defaults()
    assert _host.compute_window(["ACG"], ["ACGT"], [0.999], [0.9999], [0.0], [0], 0, p, faster=True) == {"throw": "hapSize error."}
    assert _host.compute_window(["ACGTACGTACGT"], ["ACG"], [0.999], [0.9999], [0.0], [0], 0, p, faster=True) == \
        {"throw": "HapHash string too short"}


def test_engine_fast_unpack_equals_oracle():
    """runBatch fills gap-free, mismatch-free pairs from the device's counters without the per-base walk: every record,
    whichever way it was filled, must equal the oracle's reportVariants (ObservationModelFB.cpp:1351-1475) — counters,
    hpos, and the keys / strings / coordinates of ml.indels and ml.snps."""
    rng = np.random.default_rng(11)
    ref = "".join(rng.choice(list("ACGT"), 110))
    haps = [ref, ref[:50] + ref[53:], ref[:60] + "TTG" + ref[60:], ref[:30] + "N" + ref[31:]]
    reads, quals, mapq, pos = [], [], [], []
    for i in range(60):
        src = haps[i % 3]
        L = int(rng.integers(20, 70))
        off = int(rng.integers(-10, len(src) - 15))
        s = [src[j] if 0 <= j < len(src) else str(rng.choice(list("ACGT"))) for j in range(off, off + L)]
        if i % 5 == 0:
            s[int(rng.integers(0, L))] = "A"
        reads.append("".join(s)); quals.append(float(rng.choice([0.999, 0.9, 0.96])))
        mapq.append(0.9999); pos.append(1000.0 + off)
    p = capi.params_cli_defaults()
    res = _host.compute_window(haps, reads, quals, mapq, pos, [0] * len(reads), 1000, p)
    n_plain = 0
    for h, hap in enumerate(haps):
        for r, read in enumerate(reads):
            ml = res["liks"][h][r]
            o, hpos = _oracle.pair(hap, read, quals[r], mapq[r], int(pos[r]), 1000, p)
            for k in ("numIndels", "numMismatch", "nBQT", "nmmBQT", "nMMLeft", "nMMRight", "firstBase", "lastBase", "mLogBQ"):
                assert ml[k] == getattr(o, k), (h, r, k)
            assert ml["hpos"] == hpos
            _oracle.assert_record_variants(ml, o, hap, read, (h, r))
            assert ml["align"] == _oracle.expected_align(o, hap, read)
            n_plain += (not ml["indels"] and not ml["snps"])
    assert 20 < n_plain < len(haps) * len(reads)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_engine_random_windows_both_models(seed):
    """Adversarial windows (tests/test_gpu_fuzz.make_windows: tiny alphabets, N / IUPAC bytes, indel-carrying and junk reads,
    reads hanging off either end, 1-3-bp reads, reads that are inserted as a whole) through LikelihoodEngine: every record
    equals the oracle's values, including the keys, strings and coordinates of ml.indels / ml.snps of both models."""
    from tests.test_gpu_fuzz import make_windows
    rng = np.random.default_rng(7000 + seed)
    p = capi.params_cli_defaults() if seed % 2 else capi.params_struct_defaults()
    n_pairs = n_whole_ins = 0
    ws = make_windows(rng, 6, 90, 70, min_hap=p.maxLengthDel)
    # the advisor's case (a read that is inserted as a whole: its key is the join's state, not a neighbour's) and 1-3-bp reads
    q1 = [0.9999]
    ws.append(Window(1000, ["A" * 30, "A" * 12 + "C" + "A" * 17], [ReadRec("C", q1, 0.9999, 1010), ReadRec("G", q1, 0.99, 1000),
                                                                   ReadRec("CG", q1 * 2, 0.9999, 1005), ReadRec("TTT", q1 * 3, 0.9999, 1020),
                                                                   ReadRec("GGGGGGGG", q1 * 8, 0.9999, 1008)]))
    for w in ws:
        reads = [r.seq for r in w.reads]
        quals = [list(r.qual) for r in w.reads]
        mapq = [r.mapQual for r in w.reads]
        pos = [float(r.start) for r in w.reads]
        um = [int(r.unmapped) for r in w.reads]
        res = _host.compute_window(w.haps, reads, quals, mapq, pos, um, w.hap_start, p)
        resf = _host.compute_window(w.haps, reads, quals, mapq, pos, um, w.hap_start, p, faster=True)
        for h, hap in enumerate(w.haps):
            for r, rd in enumerate(w.reads):
                o, hpos = _oracle.pair(hap, rd.seq, rd.qual, rd.mapQual, rd.start, w.hap_start, p, unmapped=rd.unmapped)
                ml = res["liks"][h][r]
                assert (ml["ll"], ml["llOn"], ml["llOff"]) == (o.ll, o.llOn, o.llOff) and ml["hpos"] == hpos
                assert (ml["numIndels"], ml["numMismatch"], ml["nBQT"], ml["nmmBQT"], ml["nMMLeft"], ml["nMMRight"]) == \
                    (o.numIndels, o.numMismatch, o.nBQT, o.nmmBQT, o.nMMLeft, o.nMMRight)
                assert (ml["firstBase"], ml["lastBase"], ml["offHap"], ml["offHapHMQ"]) == (o.firstBase, o.lastBase, o.offHap, o.offHapHMQ)
                _oracle.assert_record_variants(ml, o, hap, rd.seq, (seed, h, r))
                assert ml["align"] == _oracle.expected_align(o, hap, rd.seq)
                n_whole_ins += bool(hpos) and all(x == capi.DD_HPOS_INS for x in hpos)
                if "liks" in resf:
                    f, fh = _oracle.pair_fast(hap, rd.seq, rd.qual, rd.mapQual, rd.start, w.hap_start, p)
                    mf = resf["liks"][h][r]
                    assert mf["ll"] == f.ll and mf["hpos"] == fh and (mf["firstBase"], mf["lastBase"]) == (f.firstBase, f.lastBase)
                    _oracle.assert_record_variants(mf, f, hap, rd.seq, ("faster", seed, h, r))
                n_pairs += 1
        if any(len(r.seq) < 4 for r in w.reads):
            assert resf == {"throw": "HapHash string too short"}, (str(resf)[:300], [len(r.seq) for r in w.reads], [len(h) for h in w.haps])
    assert n_pairs > 15 and n_whole_ins > 0


@pytest.mark.parametrize("faster", [False, True])
@pytest.mark.parametrize("keep", [True, False])
def test_batch_lazy_views_equal_eager_records_and_oracle(faster, keep):
    """computeLikelihoodsBatch over many windows at once: the lazy WindowLikelihoods view (scalars straight from the result
    block, full records from get() — recomputed per window when the batch kept no alignments) equals the eager
    MLAlignment records pair by pair, the log-likelihoods equal the oracle's, and a window that cannot be processed (767-bp
    haplotype; too short a haplotype; a 3-bp read for the --faster model) fails alone with the reference's string."""
    from tests.test_gpu_fuzz import make_windows
    rng = np.random.default_rng(99)
    p = capi.params_cli_defaults()
    ws = make_windows(rng, 14, 90, 70, min_hap=p.maxLengthDel)
    if faster:
        ws = [w for w in ws if all(len(r.seq) >= 4 for r in w.reads)]
    q = [0.999]
    hapL = "".join(rng.choice(list("ACGT"), 767))
    ws.insert(3, Window(1000, [hapL], [ReadRec(hapL[:50], q * 50, 0.9999, 1000)]))                     # outside the kernel limits
    ws.insert(7, Window(1000, ["ACG", "ACGTACGTAC"], [ReadRec("ACGTA", q * 5, 0.9999, 1000)]))         # hapSize error.
    if faster:
        ws.insert(9, Window(1000, ["ACGTACGTACGTA"], [ReadRec("ACG", q * 3, 0.9999, 1000)]))           # HapHash string too short
    res = _host.batch(ws, p, faster=faster, keep_alignments=keep)
    assert res["mismatch"] == 0
    errs = [w["error"] for w in res["windows"]]
    assert errs[3].startswith("window outside the GPU kernel limits") and errs[7] == "hapSize error."
    if faster:
        assert errs[9] == "HapHash string too short"
    assert sum(1 for e in errs if e) == (3 if faster else 2)
    n = 0
    for w, rw in zip(ws, res["windows"]):
        if rw["error"]:
            continue
        i = 0
        for hap in w.haps:
            for rd in w.reads:
                if faster:
                    o, _ = _oracle.pair_fast(hap, rd.seq, rd.qual, rd.mapQual, rd.start, w.hap_start, p)
                else:
                    o, _ = _oracle.pair(hap, rd.seq, rd.qual, rd.mapQual, rd.start, w.hap_start, p, unmapped=rd.unmapped)
                assert rw["ll"][i] == o.ll, (hap, rd.seq)
                i += 1
                n += 1
    assert n >= 60


def test_engine_bench_hook_runs():
    """The end-to-end leg bench.py reports (ddh_bench_batch): lazy, lazy without alignments and eager agree on ll[0][0][0]."""
    a = _host.bench_batch(48, reps=1)
    b = _host.bench_batch(48, reps=1, keep_alignments=False)
    c = _host.bench_batch(48, reps=1, eager=True)
    assert a["errors"] == b["errors"] == c["errors"] == 0
    assert a["ll00"] == b["ll00"] == c["ll00"] and a["ll00"] < 0

