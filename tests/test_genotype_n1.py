"""Next row N1: the MAP haplotype-pair step of diploidGLF (reference DInDel.cpp:3062-3120).

CPU: host C++ diploidPairPosteriors against a direct Python restatement of the reference loop.
GPU: dd_pair_sums (device read-sums over the ll array) against the oracle's glibc evaluation, 1e-12 relative
(the device's exp/log may differ from glibc in the last ulp; the VCF only ever sees int(qual))."""
import ctypes as C
import math

import numpy as np
import pytest

from dindel_tgi_amd import capi, synth
from tests import _host, _oracle


def add_logs(l1, l2):
    if l1 > l2:
        return l1 + math.log(1.0 + math.exp(l2 - l1))
    return l2 + math.log(1.0 + math.exp(l1 - l2))


def py_pair_posteriors(nh, pair_sum, prior, filtered, ncand):
    post = np.zeros(nh * nh)
    mi, mn, pi, pn = -math.inf, -math.inf, [-1, -1], [-1, -1]
    for h1 in range(nh):
        if filtered[h1]:
            continue
        for h2 in range(h1, nh):
            if filtered[h2]:
                continue
            pp = pair_sum[h1 * nh + h2] + prior[h1 * nh + h2]
            post[h1 * nh + h2] = pp
            if pp > mi and (ncand[h1] > 0 or ncand[h2] > 0):
                mi, pi = pp, [h1, h2]
            if pp > mn and (ncand[h1] == 0 and ncand[h2] == 0):
                mn, pn = pp, [h1, h2]
    qual = -10.0 * (mn - add_logs(mi, mn)) / math.log(10.0)
    return post, pi, pn, mi, mn, qual


def host_pair_posteriors(nh, pair_sum, prior, filtered, ncand):
    lib = _host.load()
    ps = np.ascontiguousarray(pair_sum, np.float64); pr = np.ascontiguousarray(prior, np.float64)
    f = np.ascontiguousarray(filtered, np.int32); nc = np.ascontiguousarray(ncand, np.int32)
    post = np.zeros(nh * nh); pairs = np.zeros(4, np.int32); vals = np.zeros(3)
    rc = lib.ddh_pair_posteriors(nh, ps.ctypes.data_as(capi.c_f64p), pr.ctypes.data_as(capi.c_f64p),
                                 f.ctypes.data_as(capi.c_i32p), nc.ctypes.data_as(capi.c_i32p),
                                 post.ctypes.data_as(capi.c_f64p), pairs.ctypes.data_as(capi.c_i32p),
                                 vals.ctypes.data_as(capi.c_f64p))
    return rc, post, pairs, vals


def test_host_pair_posteriors_matches_reference_loop():
    rng = np.random.default_rng(5)
    for trial in range(50):
        nh = int(rng.integers(2, 9))
        ps = -rng.random(nh * nh) * 500 - 100
        pr = np.log(rng.choice([1e-4, 1e-3, 1.0], nh * nh))
        filtered = (rng.random(nh) < 0.2).astype(np.int32)
        ncand = rng.integers(0, 2, nh).astype(np.int32)
        rc, post, pairs, vals = host_pair_posteriors(nh, ps, pr, filtered, ncand)
        want = py_pair_posteriors(nh, ps, pr, filtered, ncand)
        if want[1] == [-1, -1]:
            assert rc == -1                      # "Could not find indel allele" (DInDel.cpp:3121)
            continue
        assert rc == 0
        assert np.array_equal(post, want[0])
        assert pairs.tolist() == want[1] + want[2]
        assert vals[0] == want[3] and vals[1] == want[4]
        assert vals[2] == want[5] or (math.isnan(vals[2]) and math.isnan(want[5]))


def test_oracle_pair_sums_is_the_reference_loop():
    pb = synth.generate(2, H=3, R=7, L=30, hap_len=40, seed=3)
    p = capi.params_cli_defaults()
    res = _oracle.batch(p, pb)
    lib = _oracle.load()
    b = pb.ctypes_batch()
    out = np.zeros(2 * 9)
    lib.ddo_pair_sums(C.byref(b), res["ll"].ctypes.data_as(capi.c_f64p), out.ctypes.data_as(capi.c_f64p))
    ll = res["ll"]
    for w in range(2):
        base = int(pb.win_pair_off[w])
        for h1 in range(3):
            for h2 in range(3):
                s = 0.0
                if h2 >= h1:
                    for r in range(7):
                        s += math.log(0.5) + add_logs(ll[base + h1 * 7 + r], ll[base + h2 * 7 + r])
                assert out[w * 9 + h1 * 3 + h2] == s


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(n=4, H=8, R=200), dict(n=7, H=3, R=65, vary_read_len=True, mixed_quals=True),
                                 dict(n=5, H=1, R=3), dict(n=3, H=5, R=1)])
def test_device_pair_sums(lib, cfg):
    from tests.test_gpu_parity import run_host_api
    cfg = dict(cfg)
    pb = synth.generate(cfg.pop("n"), L=60, hap_len=70, seed=21, **cfg)
    p = capi.params_cli_defaults()
    got = run_host_api(lib, p, pb)
    b = pb.ctypes_batch()
    hh = np.zeros(pb.n_windows + 1, np.int64)
    assert lib.dd_pair_sum_offsets(C.byref(b), hh.ctypes.data_as(capi.c_i64p)) == 0
    H = np.diff(pb.a["win_hap_off"]).astype(np.int64)
    assert np.array_equal(hh, np.concatenate([[0], np.cumsum(H * H)]))
    dev = np.zeros(int(hh[-1]))
    assert lib.dd_pair_sums(C.byref(b), got["ll"].ctypes.data_as(capi.c_f64p), dev.ctypes.data_as(capi.c_f64p), 0) == 0, capi.last_error()
    want = np.zeros_like(dev)
    _oracle.load().ddo_pair_sums(C.byref(b), got["ll"].ctypes.data_as(capi.c_f64p), want.ctypes.data_as(capi.c_f64p))
    np.testing.assert_allclose(dev, want, rtol=1e-12, atol=0)     # tolerance: device exp/log vs glibc
    assert (dev[want == 0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(n=40, H=8, R=60), dict(n=30, H=3, R=33, vary_read_len=True, mixed_quals=True),
                                 dict(n=6, H=1, R=3), dict(n=9, H=13, R=5)])
def test_device_map_pairs(lib, cfg):
    """dd_map_pairs: read sums + MAP indel / no-indel pairs + qual per window on the device, against the Python
    restatement of DInDel.cpp:3073-3118 fed with the oracle's read sums.  Pair indices exact (the test priors take few
    distinct values, so equal posteriors between pairs DO occur through equal read sums of identical haplotypes);
    values within 1e-12 (device exp/log vs glibc)."""
    from tests.test_gpu_parity import run_host_api
    cfg = dict(cfg)
    n = cfg.pop("n")
    pb = synth.generate(n, L=60, hap_len=70, seed=31, **cfg)
    p = capi.params_cli_defaults()
    got = run_host_api(lib, p, pb)
    b = pb.ctypes_batch()
    hh = np.zeros(n + 1, np.int64)
    lib.dd_pair_sum_offsets(C.byref(b), hh.ctypes.data_as(capi.c_i64p))
    ns = int(hh[-1])
    rng = np.random.default_rng(7)
    prior = np.log(rng.choice([1e-4, 1e-3, 1.0], ns))
    nhap = pb.n_haps
    filtered = (rng.random(nhap) < 0.15).astype(np.uint8)
    ncand = rng.integers(0, 2, nhap).astype(np.int32)
    sums = np.zeros(ns); post = np.zeros(ns); pairs = np.zeros(4 * n, np.int32); vals = np.zeros(3 * n)
    rc = lib.dd_map_pairs(C.byref(b), got["ll"].ctypes.data_as(capi.c_f64p), prior.ctypes.data_as(capi.c_f64p),
                          filtered.ctypes.data_as(C.POINTER(C.c_uint8)), ncand.ctypes.data_as(capi.c_i32p),
                          sums.ctypes.data_as(capi.c_f64p), post.ctypes.data_as(capi.c_f64p), pairs.ctypes.data_as(capi.c_i32p),
                          vals.ctypes.data_as(capi.c_f64p), 0)
    assert rc == 0, capi.last_error()
    want_sums = np.zeros(ns)
    _oracle.load().ddo_pair_sums(C.byref(b), got["ll"].ctypes.data_as(capi.c_f64p), want_sums.ctypes.data_as(capi.c_f64p))
    np.testing.assert_allclose(sums, want_sums, rtol=1e-12, atol=0)
    hoff = pb.a["win_hap_off"]
    n_none = 0
    for w in range(n):
        h0, h1 = int(hoff[w]), int(hoff[w + 1])
        nh = h1 - h0
        sl = slice(int(hh[w]), int(hh[w + 1]))
        # index decisions are made on the device's own sums (they are what its comparisons see)
        wpost, pi, pn, mi, mn, qual = py_pair_posteriors(nh, sums[sl], prior[sl], filtered[h0:h1], ncand[h0:h1])
        assert np.array_equal(post[sl], wpost)
        assert pairs[4 * w:4 * w + 4].tolist() == pi + pn, w
        assert vals[3 * w] == mi and vals[3 * w + 1] == mn
        if pi == [-1, -1]:
            n_none += 1
        if math.isfinite(qual):
            assert vals[3 * w + 2] == pytest.approx(qual, rel=1e-12, abs=1e-12)
        else:
            assert vals[3 * w + 2] == qual or (math.isnan(qual) and math.isnan(vals[3 * w + 2]))
    assert n_none < n


def py_filter_flags(pb, params, res):
    """filterHaplotypes' per-(haplotype variant, read) coverage test restated in Python straight from the reference
    loop (DInDel.cpp:1951-2054), set-based like the reference; b < L (the reference's b <= L is out of bounds)."""
    a = pb.a
    out = np.zeros(pb.var_cov_len, np.uint8)
    for w in range(pb.n_windows):
        h0, h1 = a["win_hap_off"][w], a["win_hap_off"][w + 1]
        r0, r1 = a["win_read_off"][w], a["win_read_off"][w + 1]
        R = r1 - r0
        SL = int(a["read_seq_off"][r1] - a["read_seq_off"][r0])
        for h in range(h0, h1):
            hap = bytes(a["hap_seq"][a["hap_seq_off"][h]:a["hap_seq_off"][h + 1]])
            v0, v1 = a["hap_var_off"][h], a["hap_var_off"][h + 1]
            for r in range(r0, r1):
                p = int(pb.win_pair_off[w]) + (h - h0) * R + (r - r0)
                s0, s1 = int(a["read_seq_off"][r]), int(a["read_seq_off"][r + 1])
                read = bytes(a["read_seq"][s0:s1])
                hp0 = int(pb.win_hpos_off[w]) + (h - h0) * SL + (s0 - int(a["read_seq_off"][r0]))
                hpos = res["hpos"][hp0:hp0 + (s1 - s0)]
                sel = (not res["offHapHMQ"][p]) and res["numIndels"][p] == 0
                for i in range(v1 - v0):
                    lf, rf, kind = pb.hap_var_flank[3 * (v0 + i):3 * (v0 + i) + 3]
                    left, right = lf - params.padCover, rf + params.padCover
                    ln = right - left + 1
                    cov = 0
                    if sel and kind != 0:
                        c, nmm = set(), 0
                        for b in range(len(hpos)):
                            hb = int(hpos[b])
                            if left <= hb <= right:
                                c.add(hb)
                                if kind == 1:
                                    nmm += hap[hb:hb + 1] != b"N" and hap[hb] != read[b]
                                else:
                                    nmm += hap[hb] != read[b]
                        cov = int(len(c) >= ln and nmm <= params.maxMismatch)
                    out[int(pb.win_varcov_off[w]) + (v0 - a["hap_var_off"][h0]) * R + (r - r0) * (v1 - v0) + i] = cov
    return out


def test_oracle_filter_flags_match_python_restatement():
    pb = synth.generate(4, H=5, R=30, L=60, hap_len=90, seed=13, sub_rate=0.02, mixed_quals=True)
    p = capi.params_cli_defaults()
    res = _oracle.batch(p, pb, nthreads=4)
    want = py_filter_flags(pb, p, res)
    assert np.array_equal(res["var_fcov"][:pb.var_cov_len], want)
    assert 0 < want.sum() < want.size                       # both outcomes occur


@pytest.mark.gpu
def test_device_filter_flags(lib):
    from tests.test_gpu_parity import run_host_api
    for seed, kw in ((1, dict()), (2, dict(sub_rate=0.05, mixed_quals=True)), (3, dict(vary_read_len=True))):
        pb = synth.generate(6, H=6, R=40, L=80, hap_len=110, seed=seed, **kw)
        for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
            got = run_host_api(lib, p, pb)
            want = _oracle.batch(p, pb, nthreads=8)
            assert np.array_equal(got["var_fcov"][:pb.var_cov_len], want["var_fcov"][:pb.var_cov_len])
            assert np.array_equal(got["var_fcov"][:pb.var_cov_len], py_filter_flags(pb, p, got))
