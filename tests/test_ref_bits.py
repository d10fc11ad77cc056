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
