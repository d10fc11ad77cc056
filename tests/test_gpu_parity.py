"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bar: integer / byte / index outputs exact; fp64 log-likelihoods required BIT-EQUAL to the oracle (the
kernel evaluates the reference's sums in the reference's order and takes no log on the device), which
is far inside BASELINE.json's 1e-4 relative tolerance.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.batch import ReadRec, Window, alloc_result, pack
from tests import _oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))["cases"]
INT_KEYS = ["offHap", "offHapHMQ", "numIndels", "numMismatch", "nBQT", "nmmBQT", "nMMLeft", "nMMRight", "firstBase",
            "lastBase", "hpos", "var_covered", "status", "onHap", "var_fcov"]
F64_KEYS = ["ll", "llOn", "llOff", "mLogBQ"]


def run_host_api(lib, params, pb, device=0):
    arrs, res = alloc_result(pb, fill=None)
    b = pb.ctypes_batch()
    rc = lib.dd_compute_likelihoods(C.byref(params), C.byref(b), C.byref(res), device)
    assert rc == 0, capi.last_error()
    return arrs


def assert_same(got, want, pb, rel=0.0):
    ok = want["status"][:pb.n_pairs] != capi.DD_PAIR_HAPSIZE
    assert np.array_equal(got["status"][:pb.n_pairs], want["status"][:pb.n_pairs])
    for k in INT_KEYS:
        if k in ("hpos", "var_covered", "var_fcov", "onHap", "status"):
            n = {"hpos": pb.hpos_len, "var_covered": pb.var_cov_len, "var_fcov": pb.var_cov_len, "onHap": pb.n_reads,
                 "status": pb.n_pairs}[k]
            if k == "hpos" and not ok.all():
                continue
            assert np.array_equal(got[k][:n], want[k][:n]), k
        else:
            assert np.array_equal(got[k][:pb.n_pairs][ok], want[k][:pb.n_pairs][ok]), k
    for k in F64_KEYS:
        g, w = got[k][:pb.n_pairs][ok], want[k][:pb.n_pairs][ok]
        if rel == 0.0:
            bad = np.nonzero(g != w)[0]
            assert bad.size == 0, (k, bad[:5], g[bad[:5]], w[bad[:5]])
        else:
            np.testing.assert_allclose(g, w, rtol=rel, atol=0)


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_kat_through_c_abi(lib, case):
    """The reference's own known answers through dd_compute_likelihoods."""
    p = capi.dd_params.from_dict(case["params"])
    L = len(case["read"])
    w = Window(hap_start=case["hapStart"], haps=[case["hap"]],
               reads=[ReadRec(case["read"], [case["q"]] * L, case["mapQual"], case["pos"])])
    pb = pack([w])
    got = run_host_api(lib, p, pb)
    assert got["status"][0] == 0
    for k in ("ll", "llOn", "llOff"):
        if k in case:
            assert got[k][0] == pytest.approx(case[k], rel=1e-13, abs=0), k
    for k in ("offHap", "offHapHMQ", "nBQT", "numMismatch"):
        if k in case:
            assert int(got[k][0]) == case[k], k
    if "hpos" in case:
        assert capi.hpos_reference_codes(got["hpos"][:L]).tolist() == case["hpos"]
        if "indels" in case:      # inserted bases carry the key of their insertion: the KAT's ml.indels key
            keys = sorted({capi.DD_HPOS_INS_KEY0 - int(v) for v in got["hpos"][:L] if v < capi.DD_HPOS_INS_KEY0})
            assert keys == sorted(k for k, s in case["indels"] if s[0] == "+")
    if "indels" in case:
        assert int(got["numIndels"][0]) == len(case["indels"])


@pytest.mark.parametrize("cfg", [
    dict(n=3, H=4, R=50, L=100, hap_len=120, seed=1),                                   # BASELINE config[0] shape
    dict(n=4, H=8, R=40, L=100, hap_len=120, seed=2, mixed_quals=True),                 # config[1] shape, Phred 2..41
    dict(n=6, H=3, R=33, L=100, hap_len=120, seed=3, vary_read_len=True, mixed_quals=True),   # ragged reads
    dict(n=5, H=5, R=17, L=36, hap_len=40, seed=4, mixed_quals=True),                   # K=1, short reads
    dict(n=3, H=4, R=21, L=100, hap_len=170, seed=5, mixed_quals=True),                 # K=3
    dict(n=2, H=3, R=9, L=150, hap_len=250, seed=6, mixed_quals=True, sub_rate=0.02),   # K=4
])
def test_parity_synthetic(lib, cfg):
    cfg = dict(cfg)
    n = cfg.pop("n")
    pb = synth.generate(n, **cfg)
    p = capi.params_cli_defaults()
    got = run_host_api(lib, p, pb)
    want = _oracle.batch(p, pb, nthreads=8)
    assert_same(got, want, pb)


def test_parity_struct_default_params(lib):
    """maxLengthDel=10 -> D=11 specialisation."""
    pb = synth.generate(3, H=4, R=25, L=100, hap_len=120, seed=11, mixed_quals=True, max_indel=8)
    p = capi.params_struct_defaults()
    assert_same(run_host_api(lib, p, pb), _oracle.batch(p, pb, nthreads=8), pb)


@pytest.mark.parametrize("mld", [0, 1, 3, 7, 11])
def test_parity_generic_D(lib, mld):
    """maxLengthDel values that take the generic (masked D=12) build."""
    pb = synth.generate(2, H=3, R=20, L=60, hap_len=80, seed=20 + mld, mixed_quals=True, max_indel=6)
    p = capi.params_cli_defaults()
    p.maxLengthDel = mld
    assert_same(run_host_api(lib, p, pb), _oracle.batch(p, pb, nthreads=8), pb)


@pytest.mark.parametrize("mld", [12, 13, 15, 16, 21, 30, 31])
def test_parity_long_deletions(lib, mld):
    """maxLengthDel 12..31: the D = 32 build (7-bit back-pointer fields, jump constants formed on the fly) on every lane tiling it has
    (K = 1..9), reads on both sides of long deletions, a haplotype shorter than maxLengthDel (hapSize error.)."""
    from dindel_tgi_amd.batch import pack
    from tests.test_gpu_edge_cases import reads_from, rnd
    ws = []
    for hs in (mld, 40, 70, 126, 150, 200, 260, 330, 400, 470, 574):
        if hs < mld:
            continue
        hap = rnd(hs)
        cut = hs // 2
        dl = min(mld, max(1, hs - cut - 2))
        haps = [hap, hap[:cut] + hap[cut + dl:], hap[:cut] + hap[cut + max(1, dl // 2):]] if hs > 2 * mld + 4 else [hap]
        reads = reads_from(hap, 5, min(100, max(20, hs)), junk=0.1) + (reads_from(haps[1], 4, min(90, max(20, hs - dl))) if len(haps) > 1 else [])
        ws.append(Window(1000, haps, reads))
    ws.append(Window(1000, [rnd(max(1, mld - 1)), rnd(60 if mld <= 60 else mld + 5)], reads_from(rnd(80), 3, 50)))     # first haplotype: hapSize error.
    pb = pack(ws)
    p = capi.params_cli_defaults()
    p.maxLengthDel = mld
    got = run_host_api(lib, p, pb)
    assert all(r["D"] == 32 and r["pairs_per_wave"] == 1 and r["gbt"] == 1 for r in capi.launch_log())
    assert_same(got, _oracle.batch(p, pb, nthreads=8), pb)


def test_device_pointer_path_matches_host_path(lib):
    import torch
    from dindel_tgi_amd.device import DeviceBatch
    pb = synth.generate(5, H=6, R=30, seed=31, mixed_quals=True)
    p = capi.params_cli_defaults()
    dev = DeviceBatch(pb, p, "cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        dev.launch()
    s.synchronize()
    assert_same(dev.results(), run_host_api(lib, p, pb), pb)


@pytest.mark.parametrize("n_windows, faster", [(6, False), (400, False), (6, True), (400, True)])
def test_pinned_outputs_are_written_in_place(lib, n_windows, faster):
    """Output arrays in page-locked host memory (dd_host_alloc) are stored by the kernels directly — no copy in HBM, no copy back;
    pageable arrays go through HBM and a device -> host copy.  Both ways give the same bytes, for a small batch (one staged
    transfer) and for one above the 64 MB staging limit (chunked, double-buffered), for both models."""
    from dindel_tgi_amd.batch import result_lengths, RESULT_DTYPES
    pb = synth.generate(n_windows, H=8, R=150, L=100, hap_len=120, seed=21, mixed_quals=True)
    p = capi.params_cli_defaults()
    call = lib.dd_compute_likelihoods_faster if faster else lib.dd_compute_likelihoods
    arrs, res = alloc_result(pb, fill=None)
    b = pb.ctypes_batch()
    assert call(C.byref(p), C.byref(b), C.byref(res), 0) == 0, capi.last_error()
    assert lib.dd_last_direct_outputs() == 0
    lib.dd_host_alloc.restype = C.c_void_p
    lib.dd_host_alloc.argtypes = [C.c_size_t]
    lib.dd_host_free.argtypes = [C.c_void_p]
    n = result_lengths(pb)
    pinned, res2, ptrs = {}, capi.dd_result(), []
    for k, typ in capi.RESULT_FIELDS:
        dt = np.dtype(RESULT_DTYPES[k])
        cnt = max(n[k], 1)
        ptr = lib.dd_host_alloc(cnt * dt.itemsize)
        assert ptr
        ptrs.append(ptr)
        pinned[k] = np.frombuffer((C.c_char * (cnt * dt.itemsize)).from_address(ptr), dtype=dt)
        pinned[k][...] = 0x55 if dt.kind != "f" else -1.25
        setattr(res2, k, C.cast(C.c_void_p(ptr), typ))
    try:
        assert call(C.byref(p), C.byref(b), C.byref(res2), 0) == 0, capi.last_error()
        assert lib.dd_last_direct_outputs() >= 10
        for k, _typ in capi.RESULT_FIELDS:
            a, c = arrs[k][:n[k]], pinned[k][:n[k]]
            assert np.array_equal(a.view(np.uint8), np.asarray(c).view(np.uint8)), k
        # the same through the several-devices entry point (two window blocks, each on its own host thread, both writing into the
        # one pinned set of arrays)
        for k, _typ in capi.RESULT_FIELDS:
            pinned[k][...] = 0x33 if np.dtype(RESULT_DTYPES[k]).kind != "f" else -2.5
        devs = (C.c_int * 2)(0, 0)
        multi = lib.dd_compute_likelihoods_faster_multi if faster else lib.dd_compute_likelihoods_multi
        assert multi(C.byref(p), C.byref(b), C.byref(res2), devs, 2) == 0, capi.last_error()
        for k, _typ in capi.RESULT_FIELDS:
            a, c = arrs[k][:n[k]], pinned[k][:n[k]]
            assert np.array_equal(a.view(np.uint8), np.asarray(c).view(np.uint8)), ("multi", k)
    finally:
        pinned.clear()
        for ptr in ptrs:
            lib.dd_host_free(ptr)
