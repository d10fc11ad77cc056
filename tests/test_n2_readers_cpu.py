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
