"""oracle/_ref: the three reference headers that build with the standard library alone (ReadIndelErrorModel.hpp, Utils.hpp,
Variant.hpp), compiled from /root/reference as they are (oracle/ref_bits.cpp, `make -C oracle _ref`).  The restatement, the
product's host tables and the C++ mirror types are checked against the reference's OWN code here — exact equality.
The path's translation units themselves cannot be built in this image (bam.h, Boost), see DESIGN.md §2."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from dindel_tgi_amd import capi
from tests import _host, _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libdd_ref_bits.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs the reference tree)")


@pytest.fixture(scope="module")
def ref():
    lib = C.CDLL(REF)
    lib.ref_hp_error.restype = C.c_double
    lib.ref_add_logs.restype = C.c_double
    lib.ref_add_logs.argtypes = [C.c_double, C.c_double]
    return lib


def test_homopolymer_error_model(ref, lib):
    """ReadIndelErrorModel::getViterbiHPError (ReadIndelErrorModel.hpp:36-50): the oracle's copy and the logs in the
    product's host table (dd_build_tables, T_HP) for every run length the table holds."""
    o = _oracle.load()
    o.ddo_hp_error.restype = C.c_double
    for n in range(-2, 200):
        assert o.ddo_hp_error(n) == ref.ref_hp_error(n), n
    p = capi.params_cli_defaults()
    q = np.array([0.999]); mq = np.array([0.9999])
    out = np.zeros(capi.DD_TABLE_DOUBLES)
    assert lib.dd_build_tables(C.byref(p), q.ctypes.data_as(capi.c_f64p), 1, mq.ctypes.data_as(capi.c_f64p), 1, out.ctypes.data_as(capi.c_f64p)) > 0
    t_hp = 32 + 4 * 256 + 4 * 256
    for n in range(1, 64):
        perr = ref.ref_hp_error(n)
        assert out[t_hp + 2 * n] == math.log(perr) and out[t_hp + 2 * n + 1] == math.log(1.0 - perr), n


def test_add_logs(ref):
    """addLogs (Utils.hpp:29-38): the oracle's and the C++ host mirror's against the reference's inline function."""
    o = _oracle.load(); o.ddo_add_logs.restype = C.c_double; o.ddo_add_logs.argtypes = [C.c_double, C.c_double]
    h = _host.load(); h.ddh_add_logs.restype = C.c_double; h.ddh_add_logs.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(2)
    vals = np.concatenate([-rng.random(300) * 800, [-0.0, 0.0, -1e-300, -745.2, -1e4, -np.inf]])
    pairs = [(a, b) for a in vals[:60] for b in vals]
    # the host mirror answers equal arguments and arguments more than 36.75 apart without exp / log: the neighbourhood of both
    # shortcuts, signed zeros, infinities and NaN
    base = [-1e-9, -3.5, -123.456, -700.0, 0.0, -0.0, 5.0]
    for a in base:
        pairs += [(a, a), (a, np.nextafter(a, -np.inf)), (np.nextafter(a, -np.inf), a)]
        for d in (36.0, 36.7, 36.73, 36.7368, 36.74, 36.7499999, 36.75, np.nextafter(36.75, 40.0), 36.76, 37.0, 40.0, 745.0, 1e6):
            pairs += [(a, a - d), (a - d, a)]
    special = [np.inf, -np.inf, np.nan, 0.0, -0.0, -1.0]
    pairs += [(a, b) for a in special for b in special]
    same = lambda x, y: (x == y and math.copysign(1.0, x) == math.copysign(1.0, y)) or (math.isnan(x) and math.isnan(y))
    for a, b in pairs:
        want = ref.ref_add_logs(a, b)
        for got in (o.ddo_add_logs(a, b), h.ddh_add_logs(a, b)):
            assert same(got, want), (a, b, got, want)


def test_aligned_variant_mirror(ref):
    """AlignedVariant (Variant.hpp:78-175): string parsing and isCovered of the C++ mirror type against the reference class."""
    h = _host.load()
    rng = np.random.default_rng(3)
    strs = ["+A", "+ACGT", "-T", "-GATTACA", "A=>C", "T=>D", "*REF", "-", "+", "AA>C", "X=>Y"]

    def call(lib, fn, s, args):
        t, ln = C.c_int(-9), C.c_int(-9)
        seq = C.create_string_buffer(64)
        r = getattr(lib, fn)(s.encode(), *args, C.byref(t), C.byref(ln), seq, 64)
        return r, t.value, ln.value, seq.value.decode()

    n_ok = 0
    for s in strs:
        for _ in range(40):
            args = [int(x) for x in rng.integers(-5, 60, 7)]
            want = call(ref, "ref_aligned_variant", s, args)
            got = call(h, "ddh_aligned_variant", s, args)
            assert got == want, (s, args, got, want)           # includes -1: both throw "Unrecognized variant" (Variant.hpp:68)
            n_ok += got[0] >= 0
    assert n_ok > 200


def test_struct_defaults_and_hpos_codes(ref, lib):
    """ObservationModelParameters::setDefaultValues (ObservationModel.hpp:39-64) against dd_params_struct_defaults, and the
    MLAlignment hpos codes / constructor zeros (MLAlignment.hpp:31-46) against the header's DD_HPOS_* values."""
    d = (C.c_double * 6)(); i = (C.c_int * 7)()
    ref.ref_obs_params_defaults(d, i)
    p = capi.dd_params()
    lib.dd_params_struct_defaults(C.byref(p))
    assert [p.pError, p.pMut, p.pFirstgLO, p.mapQualThreshold, p.checkBaseQualThreshold, p.capMapQualFast] == list(d)
    assert [p.maxLengthDel, p.maxLengthDel, p.padCover, p.bMid, p.forceReadOnHaplotype, p.mapUnmappedReads, p.maxMismatch] == list(i)
    codes = (C.c_int * 4)(); dd = (C.c_double * 3)(); ii = (C.c_int * 5)()
    ref.ref_mlalignment(codes, dd, ii)
    assert list(codes) == [-1, -2, -3, -4]            # DD_HPOS_INS, DD_HPOS_DEL, DD_HPOS_LO, DD_HPOS_RO (include/dindel_hmm.h)
    assert list(dd) == [0.0, 0.0, 0.0] and list(ii) == [0, 0, 0, 0, -1]


def _window_lines(fn, path, one_based):
    import json
    cap = 1 << 24
    out = C.create_string_buffer(cap)
    n = fn(str(path).encode(), one_based, out, cap)
    assert n > 0, n
    return json.loads(out.value.decode())


def test_window_file_parser_against_the_reference_parser(ref, tmp_path, capfd):
    """host/window_io.cpp's VariantFile::getLineVector (written from the format) against the reference's OWN parser
    (VariantFile.hpp:188-289, compiled as it is into oracle/_ref): the same files go through both, call by call of the loop
    `while (!vf.eof())` — tid, leftPos, rightPos, centerPos, every candidate's position / string / end / type / length / sequence /
    prior / add-combinatorially flag, which lines are skipped, which string is thrown and where, what a trailing blank, an empty line,
    a line without newline at the end of the file, a comment word, separators ';' and ',', repeated separators, a numeric prefix, a
    one-based position 0 and prior / flag fields that do not parse do."""
    from dindel_tgi_amd import hostlib
    host = hostlib.load()
    for fn in (ref.ref_window_lines_json, host.ddh_window_lines_json):
        fn.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        fn.restype = C.c_int
    rng = np.random.default_rng(77)
    bases = "ACGT"

    def variant():
        kind = rng.integers(0, 11)
        if kind < 4:
            return "+" + "".join(rng.choice(list(bases), rng.integers(1, 6)))
        if kind < 8:
            return "-" + "".join(rng.choice(list(bases), rng.integers(1, 9)))
        if kind < 10:
            a, b = rng.choice(list(bases), 2, replace=False)
            return "%s=>%s" % (a, b)
        return ["*REF", "R", "+", "-", "A", "X=>Y", "+N", "-acgt", "G=>", "T=>TT"][int(rng.integers(0, 10))]      # odd ones: some parse, some throw

    def candidate(pos):
        sep = ";" if rng.random() < 0.2 else ","
        w = "%d%s%s" % (pos, sep, variant())
        r = rng.random()
        if r < 0.3:
            w += sep + ["0.25", "1e-3", ".5", "7", "-1", "x", "0.1z", ""][int(rng.integers(0, 8))]
            if rng.random() < 0.5:
                w += sep + ["0", "1", "2", "-3", "q", "1x", ""][int(rng.integers(0, 7))]
        return w

    def line():
        left = int(rng.integers(0, 10 ** 6))
        words = [str(rng.choice(["20", "chr1", "X", "GL000207.1"])), str(left), str(left + int(rng.integers(100, 200)))]
        for _ in range(int(rng.integers(0, 5))):
            words.append(candidate(left + int(rng.integers(30, 90))))
        r = rng.random()
        if r < 0.06:
            words.insert(int(rng.integers(3, len(words) + 1)), rng.choice(["#rest", "%note", "15", "q,+A", ",+A", "15,,+A", "15,", ";;", "0,+T", "4294967296,+T", "-5,+T", "12abc,+T"]))
        text = " ".join(words) if rng.random() < 0.9 else "\t".join(words)
        r = rng.random()
        if r < 0.08:
            text += " "                                   # a blank behind the last word: not "at the end" for formatted stream input
        elif r < 0.12:
            text = " " + text
        elif r < 0.16:
            text = text.replace(" ", "  ", 1)
        return text

    files = []
    for f in range(12):                                   # ~400 generated lines over a dozen files ...
        lines = [line() for _ in range(int(rng.integers(20, 50)))]
        for _ in range(int(rng.integers(0, 3))):
            lines.insert(int(rng.integers(0, len(lines))), rng.choice(["", "20", "20 100", "20 100 200", "# comment line"]))
        files.append("\n".join(lines) + ("\n" if f % 3 else ""))
    files += ["20 10 20\n20 30", "20 \n", "20 x 20 15,+A\n", "20 10 y 15,+A\n20 10 20 16,+C\n", "20 10abc 20 15,+A\n", "20 10 20 15 16,+A\n",       # ... and the corners
              "20 10 20 15,,+A\n", "20 10 20 15,+A,zz\n", "20 10 20 15,+A,0.1,q\n", "20 10 20 q,+A\n", "20 10 20 0,+A\n", "20 10 20 15;+AC;0.25 %16,+T 17,-G\n",
              "", "\n", "\n\n20 10 20 15,+A\n", "20 10 20 15,+A", "20 10 20 15,+A \n", "20 10 20  15,+A\t16,-CC\n", "20 -10 -5 3,+A\n", "20 10 20 15,*REF 16,A=>C\n",
              "20 1e3 20 15,+A\n", "20 10 20 15,+A,1e400\n", "20 10 20 15,+A,nan\n", "20 10 20 15,+A,0x10\n", "20 10 20 99999999999,+A\n", "20 2147483648 5 15,+A\n"]
    n_calls = n_windows = n_throws = 0
    for i, text in enumerate(files):
        path = tmp_path / ("w%d.txt" % i)
        path.write_bytes(text.encode())
        for one_based in (0, 1):
            want = _window_lines(ref.ref_window_lines_json, path, one_based)
            got = _window_lines(host.ddh_window_lines_json, path, one_based)
            assert got == want, (i, one_based, text[:200])
            n_calls += len(want["calls"]); n_windows += sum(1 for c in want["calls"] if c != "skipped"); n_throws += "throw" in want
    capfd.readouterr()                                     # (both parsers report skipped lines on stderr / stdout)
    assert n_calls > 800 and n_windows > 300 and n_throws >= 6, (n_calls, n_windows, n_throws)
