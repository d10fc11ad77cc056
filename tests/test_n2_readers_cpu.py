"""N2 — host readers and the window's read selection, CPU only:
  * BGZF + BAM + BAI reader (host/bam_reader.cpp: the formats of the SAM/BAM specification, bam_fetch's traversal) on BAM files
    this test writes itself (tests/_bamwriter.py) — every field of every record, and region queries against a brute-force
    overlap filter;
  * window file (VariantFile::getLineVector, reference VariantFile.hpp:188-289) and library file (Library.hpp:143-241);
  * DetInDel::getReads (reference DInDel.cpp:885-1262) on a hand-built scenario with one read per filter branch.
Parity unpinned: the reference ships no fixtures for these and cannot be built here (libbam is absent); expectations are
hand-derived from the cited code."""
import ctypes as C
import json

import numpy as np
import pytest

from dindel_tgi_amd import hostlib
from tests import _bamwriter as bw


def call_json(fn, *args, cap=1 << 26):
    out = C.create_string_buffer(cap)
    n = fn(*args, out, cap)
    assert n > 0, n
    return json.loads(out.value.decode())


@pytest.fixture(scope="module")
def lib():
    L = hostlib.load()
    L.ddh_bam_fetch_json.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.ddh_parse_inputs_json.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    L.ddh_get_reads_pools_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_double, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.ddh_get_reads_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_double, C.c_char_p, C.c_int]
    return L


def random_records(rng, n, ref_len, rgs):
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 120))
        pos = int(rng.integers(0, ref_len - 300))
        kind = rng.random()
        if kind < 0.6:
            cigar = "%dM" % L
        elif kind < 0.75:
            a = int(rng.integers(5, L - 10)); d = int(rng.integers(1, 30))
            cigar = "%dM%dD%dM" % (a, d, L - a)
        elif kind < 0.9:
            a = int(rng.integers(5, L - 12)); ins = int(rng.integers(1, 6))
            cigar = "%dM%dI%dM" % (a, ins, L - a - ins)
        else:
            s = int(rng.integers(1, 10))
            cigar = "%dS%dM" % (s, L - s)
        flag = int(rng.choice([0, 16, 99, 147, 83, 163, 1024, 512, 2048 + 16, 4 + 8]))
        if flag & 4:
            cigar = ""
        recs.append(dict(qname="r%05d" % i, flag=flag, pos=pos, mapq=int(rng.integers(0, 61)), cigar=cigar,
                         seq="".join(rng.choice(list("ACGTN"), L, p=[.24, .24, .24, .24, .04])), qual=[int(q) for q in rng.integers(2, 42, L)],
                         mtid=-1, mpos=int(rng.integers(-1, ref_len)), isize=int(rng.integers(-500, 500)),
                         tags=({"RG": str(rng.choice(rgs))} if rng.random() < 0.8 else {})))
    return sorted(recs, key=lambda r: r["pos"])


def test_bam_reader_every_field_and_region_queries(lib, tmp_path):
    rng = np.random.default_rng(2024)
    refs = [("20", 250000), ("21", 90000)]
    header = "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:20\tLN:250000\n@SQ\tSN:21\tLN:90000\n@RG\tID:g1\tSM:s\tLB:libA\n@RG\tID:g2\tLB:libB\tSM:s\n@RG\tID:g3\tSM:s\n"
    recs = [(0, r) for r in random_records(rng, 700, 250000, ["g1", "g2", "g3", "gX"])] + [(1, r) for r in random_records(rng, 150, 90000, ["g1"])]
    path = str(tmp_path / "t.bam")
    bw.write_bam(path, header, refs, recs, block_bytes=3000)
    got = call_json(lib.ddh_bam_fetch_json, path.encode(), b"20", -1, 0)
    assert got["targets"] == [["20", 250000], ["21", 90000]] and len(got["records"]) == len(recs)
    lib_of = {"g1": "libA", "g2": "libB"}
    for (tid, r), g in zip(recs, got["records"]):
        cig = bw.parse_cigar(r["cigar"])
        assert (g["qname"], g["tid"], g["pos"], g["flag"], g["mapq"], g["mtid"], g["mpos"], g["isize"]) == \
            (r["qname"], tid, r["pos"], r["flag"], r["mapq"], r["mtid"], r["mpos"], r["isize"])
        assert g["seq"] == r["seq"] and g["qual"] == r["qual"] and g["cigar"] == [(l << 4) | op for l, op in cig]
        assert g["end"] == (r["pos"] + bw.ref_len(cig) if cig else r["pos"] + 1)          # bam_calend / Read::getEndPos
        assert g["lib"] == lib_of.get(r.get("tags", {}).get("RG"))                          # bam_get_library: LB of the record's @RG, else NULL
    for tid, name in ((0, b"20"), (1, b"21")):
        mine = [r for t, r in recs if t == tid]
        for beg, end in [(0, 1000), (16000, 17000), (16383, 16385), (100000, 180000), (249000, 250000), (5000, 5001), (0, 250000), (40000, 40000)]:
            want = [r["qname"] for r in mine if (r["pos"] + (bw.ref_len(bw.parse_cigar(r["cigar"])) if r["cigar"] else 1)) > beg and r["pos"] < end]
            if beg >= end:
                want = []                         # reg2bins returns no bin for an empty region: nothing is fetched
            g = call_json(lib.ddh_bam_fetch_json, path.encode(), name, beg, end)
            assert [x["qname"] for x in g["records"]] == want, (name, beg, end)
    assert call_json(lib.ddh_bam_fetch_json, path.encode(), b"22", 0, 10) == {"throw": "Cannot find ID!"}      # MyBam::getTID
    assert call_json(lib.ddh_bam_fetch_json, str(tmp_path / "none.bam").encode(), b"20", 0, 10) == {"throw": "Cannot open BAM file."}


def test_window_file_and_library_file(lib, tmp_path):
    vf = tmp_path / "windows.txt"
    vf.write_text("20 1000 1120 1060,-AC 1075,+T,0.01 1080;A=>G;0.5;1\n"
                  "\n"
                  "20 2000 2120 2060,-ACGT,-1,0 #comment 2070,+A\n"
                  "20 3000 3120 3060,xyz\n"                       # unrecognised variant: the line is skipped
                  "21 500 620 560,R=>A %rest\n")
    lf = tmp_path / "libs.txt"
    counts = [0, 0, 1, 5, 20, 50, 20, 5, 1, 0]
    lf.write_text("#LIB libA\n" + "".join("%d %d\n" % (i, c) for i, c in enumerate(counts)) + "#LIB libB\n" + "".join("%d 1\n" % i for i in range(300)))
    got = call_json(lib.ddh_parse_inputs_json, str(vf).encode(), 0, str(lf).encode())
    w = got["windows"]
    assert [(x["tid"], x["leftPos"], x["rightPos"], x["centerPos"]) for x in w] == [("20", 1000, 1120, 1060), ("20", 2000, 2120, 2060), ("21", 500, 620, 560)]
    # [startHap, string, endHap, freq, addComb]: a deletion ends at start + length - 1 (Variant.hpp:105-109)
    assert w[0]["variants"] == [[1060, "-AC", 1061, -1, 0], [1075, "+T", 1075, 0.01, 0], [1080, "A=>G", 1080, 0.5, 1]]
    assert w[1]["variants"] == [[2060, "-ACGT", 2063, -1, 0]] and w[2]["variants"] == [[560, "R=>A", 560, -1, 0]]
    one = call_json(lib.ddh_parse_inputs_json, str(vf).encode(), 1, b"")["windows"]
    assert one[0]["variants"][0][:3] == [1059, "-AC", 1060] and one[0]["leftPos"] == 1000          # only variant positions shift
    L = got["libraries"]
    assert set(L) == {"single_end", "libA", "libB"}
    # libA: mode at 5 -> maxins = min(25 * 5, 10) = 10; probabilities normalised with a 1e-10 floor (Library.hpp:78-128)
    assert L["libA"][0] == 10 and L["libA"][2] == pytest.approx(1e-10) and L["libA"][3] == pytest.approx(50 / 102)
    assert L["single_end"][0] == 2000 and L["single_end"][2] == pytest.approx(1 / 2000)
    assert L["libB"][0] == 300 and got["maxInsertSize"] == 2000


def test_window_and_library_file_corner_cases(lib, tmp_path):
    """The behaviours of the two text formats that come from the reference reading them with formatted stream input
    (VariantFile.hpp:188-289, Library.hpp:143-241) — the parsers here are written from the format, so each one is pinned."""
    def windows(text, one_based=0):
        f = tmp_path / "w.txt"
        f.write_text(text)
        return call_json(lib.ddh_parse_inputs_json, str(f).encode(), one_based, b"")
    assert windows("20 10 20\n20 30")["windows"] == []                        # no candidates / line ends after two words: skipped
    assert windows("20 \n") == {"throw": "Cannot read left boundary of region."}        # a blank after the last word is not the end of the line
    assert windows("20 x 20 15,+A\n") == {"throw": "Cannot read left boundary of region."}
    assert windows("20 10 y 15,+A\n") == {"throw": "Cannot read left boundary of region."}    # the same text for the right boundary
    assert windows("20 10abc 20 15,+A\n")["windows"][0]["leftPos"] == 10       # a numeric prefix is a number
    got = windows("20 10 20 15 16,+A\n")["windows"]                            # a candidate without separator: reported, the rest kept
    assert [v[:2] for v in got[0]["variants"]] == [[16, "+A"]]
    assert windows("20 10 20 15,,+A\n")["windows"] == []                       # {"15", ",+A"}: unrecognised variant, the line is dropped
    assert windows("20 10 20 15,+A,zz\n")["windows"] == [] and windows("20 10 20 15,+A,0.1,q\n")["windows"] == []
    assert windows("20 10 20 q,+A\n")["windows"] == []
    assert windows("20 10 20 0,+A\n", 1)["windows"][0]["variants"][0][0] == -1  # one-based 0 wraps like the reference's uint32
    got = windows("20 10 20 15;+AC;0.25 %16,+T 17,-G\n")["windows"]
    assert got[0]["variants"] == [[15, "+AC", 15, 0.25, 0]]

    def libs(text):
        f = tmp_path / "l.txt"
        f.write_text(text)
        return call_json(lib.ddh_parse_inputs_json, b"", 0, str(f).encode())
    rows = lambda n, first=0: "".join("%d 2\n" % i for i in range(first, first + n))
    L = libs("#LIB a\n" + rows(30) + "\n#LIB b\n" + rows(40))["libraries"]       # the first empty line ends the file
    assert set(L) == {"single_end", "a"} and L["a"][0] == 30
    L = libs(rows(3) + "#LIB a\n" + rows(27, 3))["libraries"]                    # rows in front of the first header run on into it
    assert set(L) == {"single_end", "a"} and L["a"][0] == 30
    L = libs("#LIB a\n#LIB b\n" + rows(12))["libraries"]                         # a header after no rows only renames
    assert set(L) == {"single_end", "b"} and L["b"][0] == 12
    assert libs("#LIB a\n" + rows(5) + "#LIB a\n" + rows(5)) == {"throw": "Library error"}
    assert libs("#LIB single_end\n" + rows(5)) == {"throw": "Library error"}
    assert libs("#LIB a\n0 1\n2 1\n") == {"throw": "Library error."}
    assert libs("#LIB a\n0 1\n1 -1\n") == {"throw": "Library error."}
    assert libs("#LIB\n0 1\n") == {"throw": "Cannot read library name "}
    assert call_json(lib.ddh_parse_inputs_json, b"", 0, str(tmp_path / "absent").encode())["throw"].startswith("Cannot open variant file ")


def mk(qname, pos, flag=0, mapq=60, L=100, cigar=None, mtid=-1, mpos=-1, seq=None):
    return dict(qname=qname, flag=flag, pos=pos, mapq=mapq, cigar=("%dM" % L if cigar is None else cigar), seq=seq or "ACGT" * (L // 4) + "A" * (L % 4),
                qual=[30] * L, mtid=mtid, mpos=mpos, isize=0, tags={})


def test_get_reads_filter_branches(lib, tmp_path):
    """Window [10000, 10120], minReadOverlap 20, only the default single_end library (maxDev 2000): one read per branch of
    DInDel.cpp:1095-1213, then the sort / cut of :1218-1227 and the thresholds of :1256-1260."""
    recs = [
        mk("single_a", 9950),                                                   # unpaired: mtid != tid is only counted (:1111-1115) -> kept
        mk("single_b", 10060, mapq=40),                                         # kept, lower mapping quality sorts later
        mk("pair1", 9990, flag=99, mtid=0, mpos=10200), mk("pair1", 10200, flag=147, mtid=0, mpos=9990),   # both mates fetched: kept, mateLen set
        mk("orphan", 10010, flag=99, mtid=0, mpos=30000),                       # mate outside the fetched region: filtered (:1138-1141)
        mk("lowq", 10020, mapq=10),                                             # 1 - 10^-1 = 0.9 < 0.99: cut by the threshold
        mk("dup", 10030, flag=1024), mk("qcfail", 10031, flag=512), mk("suppl", 10032, flag=2048),   # never fetched (Read.hpp:392)
        mk("short_overlap", 9915),                                              # 9915 + 100 < 10000 + 20: filtered (:1104)
        mk("right_edge", 10101),                                                # pos > 10120 - 20: filtered
        mk("toolong", 10000, L=600),                                            # > maxReadLength 500 (:1100)
        mk("mate_um", 10040, flag=1 + 8 + 64, mtid=0, mpos=10040), mk("mate_um", 10040, flag=1 + 4 + 128, cigar="", mtid=0, mpos=10040),   # mapped read + its unmapped mate
        mk("far_left", 7000),                                                   # fetched region starts at 10000 - 2000 - 200
    ]
    recs.sort(key=lambda r: r["pos"])
    path = str(tmp_path / "g.bam")
    bw.write_bam(path, "@SQ\tSN:20\tLN:100000\n", [("20", 100000)], [(0, r) for r in recs])
    win = (C.c_int * 2)(10000, 10120)
    prm = (C.c_int * 4)(10000, 500, 20, 0)
    got = call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", win, 1, prm, 0.99)
    reads = got[0]["reads"]
    names = [r[0] for r in reads]
    # pair1's second mate (10200) lies right of 10120 - 20: it is filtered itself (:1104) but still serves as the first one's mate
    assert sorted(names) == ["mate_um", "pair1", "single_a", "single_b"]
    assert names[-1] == "single_b" and reads[-1][2] == pytest.approx(1 - 1e-4)              # sorted by mapping quality, descending
    by = {(r[0], r[1]): r for r in reads}
    assert by[("pair1", 9990)][3:5] == [10200, 100] and ("pair1", 10200) not in by            # matePos / mateLen from the mate found by name
    assert by[("mate_um", 10040)][3:6] == [10040, 100, 0]                                   # mate unmapped: matePos = own pos, mateLen from the unmapped mate
    assert by[("single_a", 9950)][7] == 9950.0                                              # posStat.first of a 100M read = its position
    # thresholds
    prm2 = (C.c_int * 4)(4, 500, 20, 0)
    assert call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", win, 1, prm2, 0.99)[0] == {"throw": "above_read_count_threshold"}
    assert call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", win, 1, prm, 0.9999999)[0] == {"throw": "too_few_reads"}
    win_narrow = (C.c_int * 2)(10000, 10050)
    assert call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", win_narrow, 1, prm, 0.99)[0] == {"throw": "Choose a larger width or a smaller minReadOverlap."}
    # consecutive windows reuse the read buffer: the second window's reads equal those of a fresh selection
    wins = (C.c_int * 4)(9980, 10100, 10000, 10120)
    two = call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", wins, 2, prm, 0.99)
    assert two[1] == got[0]


def _pairs_scene(rng, n_ref):
    """Single reads, proper pairs, pairs with a far or foreign mate, mapped reads with an unmapped mate, flagged records."""
    recs, k = [], 0
    bases = lambda L: "".join(rng.choice(list("ACGT"), L))
    for _ in range(900):
        k += 1
        name, pos, L = "q%04d" % k, int(rng.integers(3000, n_ref - 1000)), int(rng.integers(40, 130))
        mapq = int(rng.choice([0, 10, 20, 30, 40, 60, 60, 60]))
        kind = rng.random()
        if kind < 0.35:
            recs.append(mk(name, pos, flag=int(rng.choice([0, 16])), mapq=mapq, L=L, seq=bases(L)))
        elif kind < 0.65:                                                     # proper pair, both mapped
            p2, L2 = pos + int(rng.integers(0, 400)), int(rng.integers(40, 130))
            recs.append(mk(name, pos, flag=99, mapq=mapq, L=L, seq=bases(L), mtid=0, mpos=p2))
            recs.append(mk(name, p2, flag=147, mapq=int(rng.choice([20, 60])), L=L2, seq=bases(L2), mtid=0, mpos=pos))
        elif kind < 0.72:                                                     # the mate is far away or on another chromosome
            far = rng.random() < 0.5
            recs.append(mk(name, pos, flag=97, mapq=mapq, L=L, seq=bases(L), mtid=0 if far else 1, mpos=pos + 50000 if far else 77))
        elif kind < 0.90:                                                     # mapped read + unmapped mate placed at the same position
            rev = int(rng.choice([0, 16]))
            recs.append(mk(name, pos, flag=1 + 8 + 64 + rev, mapq=mapq, L=L, seq=bases(L), mtid=0, mpos=pos))
            recs.append(mk(name, pos, flag=1 + 4 + 128 + int(rng.choice([0, 16])), mapq=0, L=L, cigar="", seq=bases(L), mtid=0, mpos=pos))
        elif kind < 0.94:
            recs.append(mk(name, pos, flag=int(rng.choice([1024, 512, 2048])), mapq=mapq, L=L, seq=bases(L)))
        elif kind < 0.97:
            recs.append(mk(name, pos, flag=1 + 32, mapq=mapq, L=L, seq=bases(L), mtid=0, mpos=-1))      # paired, mate "mapped", position unknown
        else:
            recs.append(mk(name, pos, mapq=mapq, L=150, cigar="60M30D90M", seq=bases(150)))
    recs.sort(key=lambda r: r["pos"])                                         # stable: a pair at one position keeps its order
    return recs


@pytest.mark.parametrize("map_unmapped", [0, 1])
def test_get_reads_against_the_python_restatement(lib, tmp_path, map_unmapped):
    """Consecutive windows (overlapping, adjacent, far apart, left of maxInsert + 200, empty) over a mixed sample: every window's
    outcome equals tests/_getreads_oracle.py — the thrown message, or the reads in mapping-quality order (ties as sets: their
    order is std::sort's)."""
    from tests import _getreads_oracle as go
    rng = np.random.default_rng(77 + map_unmapped)
    n_ref = 60000
    recs = _pairs_scene(rng, n_ref)
    path = str(tmp_path / "p.bam")
    bw.write_bam(path, "@SQ\tSN:20\tLN:%d\n@SQ\tSN:21\tLN:1000\n" % n_ref, [("20", n_ref), ("21", 1000)], [(0, r) for r in recs])
    windows, left = [(1500, 1620), (2100, 2230)], 3000                          # the first two: leftPos - 2200 wraps (nothing is fetched)
    while left < n_ref - 2000:
        windows.append((left, left + int(rng.integers(60, 200))))
        left += int(rng.choice([0, 30, 150, 400, 900, 2500, 6000]))
    flat = (C.c_int * (2 * len(windows)))(*[v for w in windows for v in w])
    for max_reads, max_len, thr in ((10000, 500, 0.99), (10000, 120, 0.5), (12, 500, 0.99)):
        prm = (C.c_int * 4)(max_reads, max_len, 20, map_unmapped)
        got = call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", flat, len(windows), prm, thr, cap=1 << 26)
        want = go.run_windows(recs, windows, max_reads=max_reads, max_read_length=max_len, map_unmapped=bool(map_unmapped), map_qual_threshold=thr)
        assert len(got) == len(want)
        n_ok = 0
        for w, (g, e) in enumerate(zip(got, want)):
            if isinstance(e, dict):
                assert g == e, (w, windows[w])
                continue
            n_ok += 1
            g = [(r[0], r[1], r[2], r[3], r[4], r[5], r[6]) for r in g["reads"]]
            assert [x[2] for x in g] == pytest.approx([x[2] for x in e], rel=0, abs=0), (w, windows[w])
            assert sorted(g) == sorted(e), (w, windows[w])
        assert n_ok >= 5 and any(isinstance(e, dict) for e in want)


def test_two_bam_pools_buffer_order_and_selection(lib, tmp_path):
    """--bamFiles: two files as pools of one read buffer (DInDel.cpp:976-1003, Read::fetchFuncVectorPooled).  The buffer's order — the
    survivors of the windows before, then each pool's new records — depends on the windows walked so far; after every window it and the
    selection (with poolID) equal tests/_getreads_oracle.py extended to pools.  A one-line list is the single-file result."""
    from tests import _getreads_oracle as go
    rng = np.random.default_rng(5)
    n_ref = 40000
    scene = _pairs_scene(rng, n_ref)
    pools = [[], []]
    by_name = {}
    for r in scene:                                                        # mates stay in one file
        pools[by_name.setdefault(r["qname"], int(rng.integers(0, 2)))].append(r)
    paths = []
    for k, recs in enumerate(pools):
        paths.append(str(tmp_path / ("pool%d.bam" % k)))
        bw.write_bam(paths[-1], "@SQ\tSN:20\tLN:%d\n" % n_ref, [("20", n_ref)], [(0, r) for r in recs])
    windows, left = [], 3000
    while left < n_ref - 2000:
        windows.append((left, left + int(rng.integers(60, 200))))
        left += int(rng.choice([0, 30, 150, 400, 900, 2500, 5000]))
    flat = (C.c_int * (2 * len(windows)))(*[v for w in windows for v in w])
    for max_reads, thr in ((10000, 0.99), (14, 0.5)):                      # the second: windows skipped (above_read_count_threshold) reset the buffer
        prm = (C.c_int * 4)(max_reads, 500, 20, 1)
        got = call_json(lib.ddh_get_reads_pools_json, "\n".join(paths).encode(), b"", b"20", flat, len(windows), prm, thr, b"", 1, cap=1 << 27)
        want = go.run_windows(pools, windows, with_buffer=True, max_reads=max_reads, map_unmapped=True, map_qual_threshold=thr)
        n_ok = n_thrown = 0
        for w, (g, (e, buf)) in enumerate(zip(got, want)):
            assert [tuple(x) for x in g["buffer"]] == buf, (w, windows[w])
            if isinstance(e, dict):
                assert g["throw"] == e["throw"], (w, windows[w])
                n_thrown += 1
                continue
            rows = [(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[8]) for x in g["reads"]]
            assert [x[2] for x in rows] == [x[2] for x in e], w                  # mapping qualities descending, as selected
            assert sorted(rows) == sorted(e), (w, windows[w])                      # ties of one mapping quality: as sets (std::sort's order)
            n_ok += 1
        assert n_ok > 5 and (max_reads > 100 or n_thrown >= 2)
        assert any(len({b for _q, b in buf}) == 2 and [b for _q, b in buf] != sorted(b for _q, b in buf) for _e, buf in want)   # orders that are not pool-major occur
    # one pool given as a list of one file: the single-file hook's result
    prm = (C.c_int * 4)(10000, 500, 20, 1)
    one = call_json(lib.ddh_get_reads_pools_json, paths[0].encode(), b"", b"20", flat, len(windows), prm, 0.99, b"", 0, cap=1 << 27)
    assert one == call_json(lib.ddh_get_reads_json, paths[0].encode(), b"", b"20", flat, len(windows), prm, 0.99, cap=1 << 27)


def test_fetches_on_one_handle_match_fresh_ones(lib, tmp_path):
    """BamFile::fetch keeps inflated blocks and, for ascending regions, resumes where the previous fetch found its first
    overlapping record: a sequence of fetches on one handle returns what a fresh handle returns for each region."""
    rng = np.random.default_rng(5)
    recs = random_records(rng, 3000, 120000, ["g1"])
    for r in recs[::7]:
        r["cigar"], r["seq"], r["qual"] = "20M%dN20M" % int(rng.integers(1000, 40000)), "A" * 40, [30] * 40      # long spliced reads sit in high-level bins
    path = str(tmp_path / "s.bam")
    bw.write_bam(path, "@SQ\tSN:20\tLN:200000\n", [("20", 200000)], [(0, r) for r in recs], block_bytes=1500)
    regions, beg = [], 0
    while beg < 119000:
        regions.append((beg, beg + int(rng.integers(1, 3000))))
        beg += int(rng.integers(0, 2500))
    regions += [(50000, 50100), (100, 900), (100, 900), (60000, 60001), (59990, 70000)]         # going back, repeating
    flat = (C.c_int * (2 * len(regions)))(*[v for r in regions for v in r])
    L = lib
    L.ddh_bam_fetch_seq_json.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    seq = call_json(L.ddh_bam_fetch_seq_json, path.encode(), b"20", flat, len(regions), cap=1 << 26)
    ends = [r["pos"] + (bw.ref_len(bw.parse_cigar(r["cigar"])) if r["cigar"] else 1) for r in recs]
    for (b, e), got in zip(regions, seq):
        fresh = call_json(lib.ddh_bam_fetch_json, path.encode(), b"20", b, e)
        assert got == [x["qname"] for x in fresh["records"]], (b, e)
        assert got == [r["qname"] for r, en in zip(recs, ends) if en > b and r["pos"] < e], (b, e)


def test_position_statistics_follow_the_cigar(lib, tmp_path):
    """Read::computePositionStatistics (Read.hpp:261-306): mean offset of the matched segments, e.g. 40M2D60M -> 60 * 2 / 100 = 1.2."""
    recs = [mk("del", 10000, cigar="40M2D60M"), mk("ins", 10001, cigar="30M5I65M"), mk("clip", 10002, cigar="10S90M"), mk("plain", 10003)]
    path = str(tmp_path / "p.bam")
    bw.write_bam(path, "@SQ\tSN:20\tLN:100000\n", [("20", 100000)], [(0, r) for r in recs])
    win = (C.c_int * 2)(10000, 10120)
    prm = (C.c_int * 4)(10000, 500, 20, 0)
    reads = {r[0]: r for r in call_json(lib.ddh_get_reads_json, path.encode(), b"", b"20", win, 1, prm, 0.99)[0]["reads"]}
    assert reads["del"][7] == pytest.approx(10000 + 1.2) and reads["plain"][7] == 10003.0
    assert reads["ins"][7] == pytest.approx(10001 + 65 * (30 - 30) / 95.0)                    # insertions do not advance the position
    assert reads["clip"][7] == pytest.approx(10002 + 90 * 10 / 90.0)                           # a soft clip does


def test_haplotype_fixture_file_large_and_broken(lib, tmp_path):
    """The haplotype file stands in for getHaplotypes' output; files above 1 MB are parsed in stretches by several workers.  Every
    window comes out as written (a repeated index: the later record wins), and a broken line is reported with its line number."""
    rng = np.random.default_rng(8)
    lines, want = ["# made by the test"], {}
    for w in [5, 1, 9] + list(range(10, 1500)) + [9]:                     # out of order, and 9 twice
        left = int(rng.integers(1000, 100000))
        lines.append("W %d %d %d" % (w, left, left + 120))
        haps = []
        for h in range(int(rng.integers(1, 9))):
            seq = "".join(rng.choice(list("ACGT"), int(rng.integers(80, 160))))
            lines.append("H " + seq)
            vs = {}
            for _ in range(int(rng.integers(0, 5))):
                kind, key = str(rng.choice(["I", "S"])), int(rng.integers(0, 100))
                f = [int(x) for x in rng.integers(-1, 120, 8)]
                s = str(rng.choice(["*REF", "+AC", "-T", "A=>G"]))
                lines.append(("V %s %d %s " % (kind, key, s) + " ".join(map(str, f))) if rng.random() < 0.8 else ("V\t%s  %d %s\t" % (kind, key, s) + "  ".join(map(str, f)) + " "))
                vs[(kind, key)] = [kind, key, s] + f                      # the same (kind, key) again replaces the entry
            haps.append([seq, [vs[k] for k in sorted(vs, key=lambda k: (k[0] != "I", k[1]))]])
        if rng.random() < 0.05:
            lines.append("")
        want[w] = [w, left, left + 120, haps]
    text = "\n".join(lines) + "\n"
    assert len(text) > (1 << 20)
    path = str(tmp_path / "haps.txt")
    open(path, "w").write(text)
    lib.ddh_fixture_json.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    ask = sorted(want) + [0, 4000]
    got = call_json(lib.ddh_fixture_json, path.encode(), (C.c_int * len(ask))(*ask), len(ask), cap=1 << 27)
    assert got[-2:] == [None, None]
    for w, g in zip(ask[:-2], got[:-2]):
        assert g == want[w], w
    # a broken record far into the file: the message names the line
    bad_at = len(lines) - 40
    while not lines[bad_at].startswith("V"):
        bad_at -= 1
    broken = list(lines)
    broken[bad_at] = broken[bad_at].rsplit(None, 1)[0] + " x7"
    open(path, "w").write("\n".join(broken) + "\n")
    assert call_json(lib.ddh_fixture_json, path.encode(), (C.c_int * 1)(1), 1)[0] == want[1]        # windows are parsed when asked for: window 1 is fine
    last_w = max(i for i in range(bad_at) if lines[i].startswith("W "))
    got = call_json(lib.ddh_fixture_json, path.encode(), (C.c_int * 1)(int(lines[last_w].split()[1])), 1)
    assert got == {"throw": "Cannot read variant record in line %d of %s" % (bad_at + 1, path)}
    broken_w = list(lines)
    broken_w[last_w] = "W 12 x 5"
    open(path, "w").write("\n".join(broken_w) + "\n")
    assert call_json(lib.ddh_fixture_json, path.encode(), (C.c_int * 1)(1), 1) == {"throw": "Cannot read window record in line %d of %s" % (last_w + 1, path)}
    open(path, "w").write("H ACGT\n")
    assert call_json(lib.ddh_fixture_json, path.encode(), (C.c_int * 1)(1), 1) == {"throw": "Cannot read haplotype record in line 1 of %s" % path}
    assert call_json(lib.ddh_fixture_json, str(tmp_path / "nope").encode(), (C.c_int * 1)(1), 1) == {"throw": "Cannot open haplotype file %s" % (tmp_path / "nope")}


def test_realigned_bam_writer_round_trip(lib, tmp_path):
    """writeRealignedBAMFile (DInDel.cpp:670-725) on the own BGZF writer: the header is the input's; a read placed on a haplotype gets
    the new CIGAR, pos = refPos and isize = refPos - mpos with its bin, name, bases, qualities and tags untouched; the others are copied
    byte for byte.  The output is read back with Python's gzip (BGZF = a series of gzip members ending in the empty block)."""
    rng = np.random.default_rng(31)
    header = "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:20\tLN:250000\n@RG\tID:g1\tSM:s\tLB:libA\n"
    recs = [r for r in random_records(rng, 900, 250000, ["g1"]) if not (r["flag"] & 4)]
    path, out = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    bw.write_bam(path, header, [("20", 250000)], [(0, r) for r in recs], block_bytes=3000)
    beg, end = 20000, 200000
    inside = [r for r in recs if r["pos"] + bw.ref_len(bw.parse_cigar(r["cigar"])) > beg and r["pos"] < end]
    assert len(inside) > 400                                  # > 64 KB of records: several BGZF blocks
    on = [int(rng.random() < 0.7) for _ in inside]
    new_cigs, ref_pos = [], []
    for r in inside:
        L = len(r["seq"])
        a = int(rng.integers(1, L - 1))
        pick = int(rng.integers(0, 4))
        new_cigs.append([[(0, L)], [(4, a), (0, L - a)], [(0, a), (2, int(rng.integers(1, 40))), (0, L - a)], [(0, a), (1, 1), (0, L - a - 1)]][pick])
        ref_pos.append(int(rng.integers(-1, 240000)))
    off = np.cumsum([0] + [len(c) for c in new_cigs]).astype(np.int32)
    flat = np.array([v for c in new_cigs for op in c for v in op], dtype=np.int32)
    lib.ddh_write_realigned.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    err = C.create_string_buffer(256)
    as_p = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
    n = lib.ddh_write_realigned(path.encode(), b"20", beg, end, out.encode(), as_p(on), as_p(flat), as_p(off), as_p(ref_pos), len(inside), err, 256)
    assert n == len(inside), err.value
    text0, refs0, recs0 = bw.read_bam(path)
    text1, refs1, recs1 = bw.read_bam(out)
    assert (text1, refs1) == (text0, refs0) == (header, [("20", 250000)])
    assert open(out, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")      # the end-of-file block
    orig = {r["qname"]: r for r in recs0}
    assert [r["qname"] for r in recs1] == [r["qname"] for r in inside]
    for r1, flag, cig, rp in zip(recs1, on, new_cigs, ref_pos):
        r0 = orig[r1["qname"]]
        if not flag:
            assert r1["raw"] == r0["raw"]
            continue
        assert r1["cigar"] == "".join("%d%s" % (l, bw.CIGAR_OPS[op]) for op, l in cig)
        assert (r1["pos"], r1["isize"]) == (rp, rp - r0["mpos"])
        for k in ("tid", "qname", "mapq", "bin", "flag", "mtid", "mpos", "seq", "qual", "aux"):
            assert r1[k] == r0[k], k
    # what the reference checks before writing
    n = lib.ddh_write_realigned(path.encode(), b"20", beg, end, str(tmp_path / "no_such_dir" / "x.bam").encode(), as_p(on), as_p(flat), as_p(off), as_p(ref_pos),
                                len(inside), err, 256)
    assert n == -1 and err.value.decode() == "Cannot open bamfile %s for writing!" % (tmp_path / "no_such_dir" / "x.bam")
    # the reader reads what the writer wrote, index-free
    got = call_json(lib.ddh_bam_fetch_json, path.encode(), b"20", -1, 0)
    assert len(got["records"]) == len(recs0)


def test_filter_read_aux(lib, tmp_path):
    """--filterReadAux (DInDel.cpp:1233-1243): the reads whose auxiliary fields — printed as Read::getAuxData prints them (Read.hpp:223-256:
    "\\tXX" then A:c / i:number / f:number / Z:text; a signed byte comes out unsigned, as there) — contain the text are dropped, or with a
    leading '+' are the only ones kept; the read-count checks come afterwards."""
    tagsets = [{"RG": "g1", "NM": ("C", 3), "XT": ("A", "R")}, {"RG": "g2", "NM": ("C", 0), "XT": ("A", "U"), "XS": ("s", -7)},
               {"NM": ("c", -3), "XF": ("f", 1.5), "XI": ("i", -123456), "XU": ("I", 4000000000), "XH": ("S", 65000)}, {}]
    recs = []
    for k in range(24):
        r = mk("r%02d" % k, 9950 + 3 * k)
        r["tags"] = tagsets[k % 4]
        recs.append(r)
    path = str(tmp_path / "aux.bam")
    bw.write_bam(path, "@SQ\tSN:20\tLN:100000\n@RG\tID:g1\tLB:l1\n@RG\tID:g2\tLB:l1\n", [("20", 100000)], [(0, r) for r in recs])
    lib.ddh_get_reads_aux_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_double, C.c_char_p, C.c_char_p, C.c_int]
    win, prm = (C.c_int * 2)(10000, 10120), (C.c_int * 4)(10000, 500, 20, 0)
    names = lambda f: sorted(r[0] for r in call_json(lib.ddh_get_reads_aux_json, path.encode(), b"", b"20", win, 1, prm, 0.99, f)[0]["reads"])
    every = ["r%02d" % k for k in range(24)]
    aux_text = {0: "\tRGZ:g1\tNMi:3\tXTA:R", 1: "\tRGZ:g2\tNMi:0\tXTA:U\tXSi:-7", 2: "\tNMi:253\tXFf:1.5\tXIi:-123456\tXUi:4000000000\tXHi:65000", 3: ""}
    assert names(b"") == every and names(b"x") == every                                  # one character: no filter (size() > 1, :1234)
    for f in ("-XTA:R", "+XTA:R", "-RGZ:g", "+NMi:253", "-i:-7", "+f:1.5\tXIi:-123456", "qXUi:4000000000", "+\tXHi:65000", "-NMi:0\t"):
        match = f[1:]
        keep = [n for k, n in enumerate(every) if ((match in aux_text[k % 4]) == (f[0] == "+"))]
        assert names(f.encode()) == keep, f
    assert call_json(lib.ddh_get_reads_aux_json, path.encode(), b"", b"20", win, 1, prm, 0.99, b"+no such text")[0] == {"throw": "too_few_reads"}
