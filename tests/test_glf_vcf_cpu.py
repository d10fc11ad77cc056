"""N3 — the output surface: the `.glf.txt` table writer (host/glf_output.hpp; reference OutputData.hpp:28-114,
DInDel.hpp:262-276, DInDel.cpp:1361-1401, :3277-3301, :3616-3650) and the glf -> VCF converter (host/glf_to_vcf.cpp,
`dindel_glf2vcf`; reference python/mergeOutputDiploid.py).

Parity unpinned: the reference has no fixtures for either and its Python-2 scripts cannot run here; expected values are
hand-derived from the cited code (literal lines below) and cross-checked with an independent Python-3 restatement
(tests/_vcf_oracle.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from dindel_tgi_amd import hostlib
from tests import _vcf_oracle

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dindel_tgi_amd", "host")
GLF_COLUMNS = ("msg index analysis_type tid lpos rpos center_position realigned_position was_candidate_in_window ref_all nref_all num_reads "
               "post_prob_variant qual est_freq logZ hapfreqs indidx msq numOffAll num_indel num_cover_forward num_cover_reverse "
               "num_unmapped_realigned var_coverage_forward var_coverage_reverse nBQT nmmBQT mLogBQ nMMLeft nMMRight glf").split()


def fmt(x):
    lib = hostlib.load()
    out = C.create_string_buffer(64)
    lib.ddh_format_double.argtypes = [C.c_double, C.c_char_p, C.c_int]
    assert lib.ddh_format_double(x, out, 64) > 0
    return out.value.decode()


def test_glf_cells_print_like_default_ostream():
    """OutputData::Line::set formats through `stringstream << x` (OutputData.hpp:82-84): doubles get 6 significant digits."""
    assert fmt(12.3456789) == "12.3457" and fmt(0.000012345678) == "1.23457e-05" and fmt(1234567.0) == "1.23457e+06"
    assert fmt(100.0) == "100" and fmt(-0.5) == "-0.5" and fmt(37.999999999) == "38" and fmt(0.0) == "0"
    assert fmt(float("inf")) == "inf" and fmt(float("nan")) in ("nan", "-nan")
    rng = np.random.default_rng(5)
    for x in np.concatenate([rng.normal(0, 50, 200), 10.0 ** rng.uniform(-12, 12, 200)]):
        assert fmt(float(x)) == "%g" % x
    # the cells are formatted without a stream per cell; the hook reports a MISMATCH if that differs from `stringstream << x`
    for x in [float("-inf"), -float("nan"), -0.0, 5e-324, 1e-310, 1.7976931348623157e308, 999999.5, 9999995.0, 0.1 + 0.2, -123456789.0, 2147483647.0, -2147483648.0, 4294967295.0]:
        assert not fmt(x).startswith("MISMATCH"), (x, fmt(x))


def test_glf_table_header_rows_and_na_defaults(tmp_path):
    lib = hostlib.load()
    path = str(tmp_path / "demo.glf.txt")
    vals = (C.c_double * 5)(37.123456, 45.6789012, 12.987654321, -1234.56789, -4500.0)
    lib.ddh_glf_demo.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double)]
    assert lib.ddh_glf_demo(path.encode(), b"hapSize error.", vals) == 5
    lines = open(path).read().split("\n")
    assert lines[0].split(" ") == GLF_COLUMNS and lines[6] == "" and len(lines) == 7  # DInDel.hpp:262-276, one endl per line
    assert lines[4] == lines[2] and lines[5] == lines[3]                           # the window loop's text writers (GlfText) = the Line objects
    for bad in ((float("nan"), 1.0, 2.0, 3.0, 0.0), (1e300, -1e-300, 5e-324, float("inf"), float("-inf"))):    # what "%.6g" makes of odd values
        p2 = str(tmp_path / "odd.glf.txt")
        assert lib.ddh_glf_demo(p2.encode(), b"x", (C.c_double * 5)(*bad)) == 5
        odd = open(p2).read().split("\n")
        assert odd[4] == odd[2] and odd[5] == odd[3]
    rows = [dict(zip(GLF_COLUMNS, l.split(" "))) for l in lines[1:4]]
    # skipped window: only msg / index / tid / lpos / rpos are set, blanks of the thrown string become '_' (DInDel.cpp:1366-1395)
    assert rows[0]["msg"] == "error_hapSize_error." and (rows[0]["index"], rows[0]["tid"], rows[0]["lpos"], rows[0]["rpos"]) == ("7", "20", "1000123", "1000243")
    assert all(v == "NA" for k, v in rows[0].items() if k not in ("msg", "index", "tid", "lpos", "rpos"))
    # dip.map call (DInDel.cpp:3277-3301)
    want = dict(msg="ok", index="8", analysis_type="dip.map", tid="20", lpos="2000000", rpos="2000120", center_position="2000060",
                realigned_position="2000058", was_candidate_in_window="1", nref_all="-AC", num_reads="173", qual="37.1235", indidx="0",
                msq="45.6789", num_cover_forward="11", num_cover_reverse="9", num_unmapped_realigned="2", var_coverage_forward="12",
                var_coverage_reverse="10", glf="0/1:12.9877")
    assert rows[1] == {k: want.get(k, "NA") for k in GLF_COLUMNS}
    # per-position "dip" line (DInDel.cpp:3616-3650): mLogBQ is written divided by nBQT
    assert rows[2]["analysis_type"] == "dip" and rows[2]["logZ"] == "-1234.57" and rows[2]["mLogBQ"] == "-0.3" and rows[2]["nBQT"] == "15000"
    assert rows[2]["glf"] == "0/0:-310.5,0/1:-250.25,1/1:-400" and rows[2]["qual"] == "NA" and rows[2]["numOffAll"] == "4"


def test_fast_six_digit_formatter_equals_printf():
    """GlfText formats doubles with its own "%.6g" (scaled by an exact power of ten, trusted only away from rounding boundaries, snprintf
    otherwise): two million values incl. exact ties and their neighbours give printf's text."""
    lib = hostlib.load()
    lib.ddh_format_g6_check.argtypes = [C.c_ulonglong, C.c_int, C.c_char_p, C.c_int]
    out = C.create_string_buffer(256)
    for seed in (1, 2026):
        assert lib.ddh_format_g6_check(seed, 150000, out, 256) == 0, out.value.decode()


# ---------------- glf -> VCF ----------------
def write_fasta(path, seqs, width=60):
    """FASTA + .fai (name, length, offset of the first base, bases per line, bytes per line)."""
    fai = []
    with open(path, "w") as f:
        for name, s in seqs:
            f.write(">%s\n" % name)
            off = f.tell()
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(s), off, width, width + 1))
    open(path + ".fai", "w").write("".join(fai))


def glf_row(**kw):
    d = dict(msg="ok", index="1", analysis_type="dip.map", tid="1", lpos="100", rpos="220", center_position="160", realigned_position="160",
             was_candidate_in_window="1", nref_all="-AC", num_reads="50", qual="30", indidx="0", msq="40", num_cover_forward="3",
             num_cover_reverse="4", num_unmapped_realigned="0", var_coverage_forward="5", var_coverage_reverse="6", glf="0/1:33.7")
    d.update({k: str(v) for k, v in kw.items()})
    return " ".join(d.get(c, "NA") for c in GLF_COLUMNS)


@pytest.fixture(scope="module")
def vcf_case(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("vcf")
    rng = np.random.default_rng(12)
    def rnd(n):
        s = list(rng.choice(list("ACGT"), n))
        for i in range(3, n):                         # no accidental homopolymers of 4: the hp filter is placed on purpose below
            if s[i] == s[i - 1] == s[i - 2] == s[i - 3]:
                s[i] = "ACGT"[("ACGT".index(s[i]) + 1 + i % 3) % 4]
        return "".join(s)
    chr1 = list(rnd(600))
    chr1[300:315] = "A" * 15                        # a 15-bp homopolymer: hp10 fires around here
    chr1[400:406] = "G" * 6                         # a 6-bp run: HP reported, no filter
    seqs = [("1", "".join(chr1)), ("2", rnd(300)), ("10", rnd(300)), ("X", rnd(300)), ("MT", rnd(200)), ("GL000207.1", rnd(200))]
    ref = str(tmp / "ref.fa")
    write_fasta(ref, seqs)
    rows1 = [
        glf_row(realigned_position=160, nref_all="-AC", qual="30.9", glf="0/1:33.7"),                       # deletion, PASS, int(float(qual)) = 30
        glf_row(realigned_position=100, nref_all="+TTG", qual="1e2", glf="1/1:99.99"),                      # insertion, scientific notation
        glf_row(realigned_position=59, nref_all="-ACGTA", qual="45"),                                       # deletion across a FASTA line end (60 bases per line)
        glf_row(realigned_position=200, nref_all="A=>C", qual="25"),                                        # SNP only: POS + 1
        glf_row(realigned_position=210, nref_all="A=>C,A=>G", glf="1/2:12"),                                # two SNPs: both ALTs lose the anchor base
        glf_row(realigned_position=220, nref_all="-AC,+G", glf="1/2:50.5"),                                 # het deletion / insertion
        glf_row(realigned_position=230, nref_all="-AC,-AC", glf="1/1:60"),                                  # the same allele twice -> one ALT
        glf_row(realigned_position=240, nref_all="A=>D,-ACG"),                                              # an ALT with a 'D' becomes <DEL>
        glf_row(realigned_position=250, nref_all="R=>D"),                                                   # dropped (:210)
        glf_row(realigned_position=305, nref_all="-A", qual="50"),                                          # inside the 15-bp run: hp10
        glf_row(realigned_position=306, nref_all="+A", qual="12"),                                          # hp10 and q20
        glf_row(realigned_position=402, nref_all="-G", qual="19.99"),                                       # q20 (19), HP=6 reported
        glf_row(realigned_position=450, qual="0.99"),                                                       # int(float(qual)) = 0 < 1: dropped
        glf_row(realigned_position=451, qual="1.0", glf="0/1:0.4"),                                         # kept, q20, GQ 0
        glf_row(realigned_position=460, was_candidate_in_window=0),                                         # not a candidate: dropped
        glf_row(realigned_position=461, analysis_type="dip"),                                               # the per-position line: not a call
        glf_row(msg="error_hapSize_error.", analysis_type="NA", realigned_position="NA", qual="NA", glf="NA", nref_all="NA"),   # skipped window
        glf_row(realigned_position=160, nref_all="+GG", qual="22", glf="0/1:21"),                           # second call at position 160: file order kept
        glf_row(realigned_position=500, nref_all="*REF,-AC", glf="0/1:30"),                                 # a reference allele in the list
        glf_row(realigned_position=30, nref_all="+C", var_coverage_forward="7,2", var_coverage_reverse="8,1"),   # first entries of the coverage lists
    ]
    rows2 = [glf_row(tid="X", realigned_position=50, nref_all="-TT"), glf_row(tid="2", realigned_position=77, nref_all="+A"),
             glf_row(tid="MT", realigned_position=20, nref_all="-C"), glf_row(tid="10", realigned_position=90, nref_all="C=>T"),
             glf_row(tid="GL000207.1", realigned_position=60, nref_all="+AT"), glf_row(tid="2", realigned_position=30, nref_all="-G")]
    g1, g2 = str(tmp / "a.glf.txt"), str(tmp / "b.glf.txt")
    open(g1, "w").write(" ".join(GLF_COLUMNS) + "\n" + "\n".join(rows1) + "\n")
    open(g2, "w").write(" ".join(GLF_COLUMNS) + "\n" + "\n".join(rows2) + "\n")
    lst = str(tmp / "files.txt")
    open(lst, "w").write(g1 + "\n" + g2 + "\n")
    return dict(tmp=tmp, ref=ref, lst=lst, seqs=dict(seqs))


def run_tool(case, out, *extra):
    subprocess.check_call(["make", "-s", "-C", HOST, "dindel_glf2vcf"])
    subprocess.check_call([os.path.join(HOST, "dindel_glf2vcf"), "-i", case["lst"], "-o", out, "-r", case["ref"], *extra],
                          stdout=subprocess.DEVNULL)
    return open(out).read()


def test_glf2vcf_equals_restatement_on_every_branch(vcf_case):
    """26 .glf.txt rows over two files and six sequences: every filter / allele / skip branch of the script."""
    got = run_tool(vcf_case, str(vcf_case["tmp"] / "out.vcf"), "-s", "NA12878")
    want_path = str(vcf_case["tmp"] / "want.vcf")
    _vcf_oracle.merge_output(vcf_case["lst"], "NA12878", vcf_case["ref"], 10, want_path)
    assert got == open(want_path).read()
    body = [l for l in got.split("\n") if l and not l.startswith("#")]
    assert len(body) == 21                                                           # 26 rows - R=>D, qual<1, non-candidate, dip, skipped
    assert [l.split("\t")[0] for l in body] == ["1"] * 15 + ["2", "2", "10", "X", "GL000207.1", "MT"]      # 1..22, X, Y, then the others
    assert [int(l.split("\t")[1]) for l in body[:15]] == sorted(int(l.split("\t")[1]) for l in body[:15])


def test_glf2vcf_literal_lines(vcf_case):
    """Lines derived by hand from mergeOutputDiploid.py:35-154 for the fixture's sequence."""
    s1 = vcf_case["seqs"]["1"]
    got = run_tool(vcf_case, str(vcf_case["tmp"] / "out2.vcf"))
    lines = {(l.split("\t")[0], l.split("\t")[1], l.split("\t")[4]): l for l in got.split("\n") if l and not l.startswith("#")}
    def ref(pos1, n):
        return s1[pos1 - 1:pos1 - 1 + n]
    # deletion of 2 bases reported at the base before it: REF = 3 reference bases, ALT = the first one; QUAL int(30.9), GQ int(33.7)
    assert lines[("1", "160", ref(160, 1))] == "1\t160\t.\t%s\t%s\t30\tPASS\tDP=50;NF=5;NR=6;NRS=3;NFS=4;HP=%d\tGT:GQ\t0/1:33" % (
        ref(160, 3), ref(160, 1), _vcf_oracle.homopolymer_length(list(s1[160 - 25:160 + 25]), 25))
    # SNP: one base to the right, single-base REF / ALT
    assert lines[("1", "201", "C")].split("\t")[:6] == ["1", "201", ".", ref(201, 1), "C", "25"]
    # two alleles, deletion then insertion
    assert lines[("1", "220", ref(220, 1) + "," + ref(220, 1) + "G" + ref(221, 2))].split("\t")[3] == ref(220, 3)
    # <DEL>
    l = lines[("1", "240", "<DEL>," + ref(240, 1))]
    assert l.split("\t")[3] == ref(240, 4) and l.endswith("GT:GQ\t0/1:33")
    # filters
    assert lines[("1", "305", "A")].split("\t")[6] == "hp10" and lines[("1", "306", "AA")].split("\t")[6] == "hp10;q20"
    assert lines[("1", "402", "G")].split("\t")[5:7] == ["19", "q20"] and ";HP=6\t" in lines[("1", "402", "G")]
    assert lines[("1", "451", ref(451, 1))].endswith("0/1:0")
    # coverage lists: first entries only
    assert "NF=7;NR=8;" in lines[("1", "30", ref(30, 1) + "C" + "")]
    header = [l for l in got.split("\n") if l.startswith("#")]
    assert header[0] == "##fileformat=VCFv4.0" and header[-1].endswith("FORMAT\tSAMPLE") and len(header) == 17
    assert '##FILTER=<ID=q20,Description="Quality below 20">' in header


def test_glf2vcf_options_and_errors(vcf_case, tmp_path):
    got = run_tool(vcf_case, str(tmp_path / "o.vcf"), "-f", "40", "--maxHPLen", "5")
    assert '##FILTER=<ID=q40,' in got and '##FILTER=<ID=hp5,' in got
    body = [l for l in got.split("\n") if l and not l.startswith("#")]
    # --maxHPLen only changes the header: processDiploidGLFFile calls getVCFString without it (:219), the filter stays hp10
    assert not any("hp5" in l.split("\t")[6] for l in body) and any(l.split("\t")[6] == "hp10;q40" for l in body)
    assert sum("q40" in l.split("\t")[6] for l in body) > sum("q20" in l.split("\t")[6] for l in run_tool(vcf_case, str(tmp_path / "p.vcf")).split("\n") if l and not l.startswith("#"))
    tool = os.path.join(HOST, "dindel_glf2vcf")
    assert subprocess.call([tool, "-o", "x", "-r", vcf_case["ref"]], stderr=subprocess.DEVNULL) == 1          # Please specify --inputFiles
    missing = str(tmp_path / "missing.txt")
    open(missing, "w").write("/nonexistent/file.glf.txt\n")
    assert subprocess.call([tool, "-i", missing, "-o", str(tmp_path / "q.vcf"), "-r", vcf_case["ref"]], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1
