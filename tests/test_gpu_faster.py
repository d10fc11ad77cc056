"""GPU parity for the --faster model: dd_compute_likelihoods_faster / dd_launch_device_faster
against the ObservationModelS restatement in oracle/ (ddo_pair_fast / ddo_batch_fast) and the three SURVEY KATs.

Bar: ll bit-equal (the kernel sums the reference's terms in the reference's order), every integer output exact.
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
KAT = [c for c in json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))["cases"] if "ll_fast" in c]
CHECKED = ["ll", "status", "offHap", "offHapHMQ", "firstBase", "lastBase", "numIndels", "numMismatch", "llOn", "llOff", "mLogBQ"]


def run_faster(lib, params, pb, device=0):
    arrs, res = alloc_result(pb, fill=None)
    b = pb.ctypes_batch()
    rc = lib.dd_compute_likelihoods_faster(C.byref(params), C.byref(b), C.byref(res), device)
    assert rc == 0, capi.last_error()
    return arrs


def assert_same_faster(got, want, pb):
    np_ = pb.n_pairs
    assert np.array_equal(got["status"][:np_], want["status"][:np_])
    ok = want["status"][:np_] == capi.DD_PAIR_OK
    for k in CHECKED:
        g, w = got[k][:np_][ok], want[k][:np_][ok]
        bad = np.nonzero(g != w)[0]
        assert bad.size == 0, (k, bad[:5], g[bad[:5]], w[bad[:5]])
    if ok.all():
        assert np.array_equal(got["hpos"][:pb.hpos_len], want["hpos"][:pb.hpos_len])
    else:                                   # hpos of a failed pair is not written by either side
        a, pair, hp = pb.a, 0, 0
        for w in range(len(a["win_hap_off"]) - 1):
            r0, r1 = int(a["win_read_off"][w]), int(a["win_read_off"][w + 1])
            offs = a["read_seq_off"][r0:r1 + 1] - a["read_seq_off"][r0]
            for _h in range(int(a["win_hap_off"][w]), int(a["win_hap_off"][w + 1])):
                for i in range(r1 - r0):
                    if ok[pair + i]:
                        lo, hi = hp + int(offs[i]), hp + int(offs[i + 1])
                        assert np.array_equal(got["hpos"][lo:hi], want["hpos"][lo:hi]), (w, _h, i)
                pair += r1 - r0
                hp += int(offs[-1])
    assert np.array_equal(got["var_covered"][:pb.var_cov_len], want["var_covered"][:pb.var_cov_len])
    assert np.array_equal(got["var_fcov"][:pb.var_cov_len], want["var_fcov"][:pb.var_cov_len])
    assert np.array_equal(got["onHap"][:pb.n_reads], want["onHap"][:pb.n_reads])


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_faster_kat_through_c_abi(lib, case):
    p = capi.dd_params.from_dict(case["params"])
    L = len(case["read"])
    w = Window(hap_start=case["hapStart"], haps=[case["hap"]],
               reads=[ReadRec(case["read"], [case["q"]] * L, case["mapQual"], case["pos"])])
    got = run_faster(lib, p, pack([w]))
    assert got["status"][0] == 0
    assert got["ll"][0] == pytest.approx(case["ll_fast"], rel=1e-14, abs=0)
    assert int(got["offHap"][0]) == 0 and int(got["offHapHMQ"][0]) == 0


@pytest.mark.parametrize("cfg", [
    dict(n=3, H=4, R=50, L=100, hap_len=120, seed=101),
    dict(n=4, H=8, R=40, L=100, hap_len=120, seed=102, mixed_quals=True, max_indel=8),
    dict(n=6, H=3, R=33, L=100, hap_len=120, seed=103, vary_read_len=True, mixed_quals=True),
    dict(n=5, H=5, R=17, L=36, hap_len=40, seed=104, mixed_quals=True, sub_rate=0.05),
    dict(n=3, H=4, R=21, L=100, hap_len=170, seed=105, mixed_quals=True, sub_rate=0.03),
    dict(n=2, H=3, R=9, L=250, hap_len=400, seed=106, mixed_quals=True, sub_rate=0.02),
    dict(n=2, H=3, R=12, L=70, hap_len=50, seed=107, mixed_quals=True),                 # reads longer than the haplotype
    dict(n=2, H=2, R=600, L=40, hap_len=60, seed=108, mixed_quals=True, vary_read_len=True),   # > 2 chunks of 256 reads per window
    dict(n=1, H=2, R=7, L=700, hap_len=760, seed=109, mixed_quals=True, sub_rate=0.01),  # pair area too large for 4 pairs per wavefront
    dict(n=1, H=2, R=5, L=1024, hap_len=766, seed=110, mixed_quals=True, sub_rate=0.3),  # maximum shape; alphas reach the -1000 floor
])
@pytest.mark.parametrize("defaults", ["cli", "struct"])
def test_faster_parity_synthetic(lib, cfg, defaults):
    cfg = dict(cfg)
    n = cfg.pop("n")
    pb = synth.generate(n, **cfg)
    p = capi.params_cli_defaults() if defaults == "cli" else capi.params_struct_defaults()
    assert_same_faster(run_faster(lib, p, pb), _oracle.batch(p, pb, nthreads=8, faster=True), pb)


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), n))


def test_faster_repeats_n_bases_and_short_reads(lib):
    """Low-complexity haplotypes (many tied diagonals -> the top-15 cut and the EPS hysteresis matter), N bases,
    reads shorter than the 4-mer (status DD_PAIR_NAN), reads far outside the window, haplotype shorter than a k-mer."""
    rng = np.random.default_rng(7)
    wins = []
    for i in range(6):
        unit = _rand_seq(rng, int(rng.integers(1, 5)))
        hap0 = (unit * 80)[:int(rng.integers(30, 140))]
        hap1 = _rand_seq(rng, 20) + hap0[:60] + _rand_seq(rng, 20, "ACGTN")
        hap2 = _rand_seq(rng, int(rng.integers(1, 9)))                # may be shorter than the k-mer / maxLengthDel
        reads = []
        for j in range(24):
            L = int(rng.integers(1, 90)) if j % 5 == 0 else int(rng.integers(20, 110))
            src = hap0 if j % 2 else hap1
            o = int(rng.integers(0, max(1, len(src) - 5)))
            s = (src[o:o + L] + _rand_seq(rng, L))[:L]
            s = "".join(c if rng.random() > 0.04 else "ACGTN"[int(rng.integers(0, 5))] for c in s)
            q = (1.0 - 10.0 ** (-rng.integers(2, 42, L) / 10.0)).tolist()
            pos = 1000 + o + int(rng.integers(-3, 4)) if j % 7 else int(rng.integers(0, 5000))
            reads.append(ReadRec(s, q, 1.0 - 10.0 ** (-int(rng.integers(0, 61)) / 10.0), pos))
        wins.append(Window(hap_start=1000, haps=[hap0, hap1, hap2], reads=reads,
                           hap_vars=[[(10, 12)], [(25, 25), (3, 40)], []],
                           hap_var_flanks=[[(9, 13, 1)], [(24, 26, 2), (0, 4, 1)], []]))
    pb = pack(wins)
    for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
        want = _oracle.batch(p, pb, nthreads=8, faster=True)
        assert (want["status"][:pb.n_pairs] != 0).any() and (want["status"][:pb.n_pairs] == 0).any()
        assert_same_faster(run_faster(lib, p, pb), want, pb)


@pytest.mark.parametrize("groups", ["1", "2"])
def test_faster_fewer_pairs_per_wavefront(lib, groups, monkeypatch):
    """Geometry used when four pair areas do not fit in LDS (not reachable at today's DD_MAX_* limits): forced here."""
    monkeypatch.setenv("DD_FAST_GROUPS", groups)
    pb = synth.generate(3, H=3, R=21, L=60, hap_len=80, seed=150 + int(groups), mixed_quals=True, vary_read_len=True)
    p = capi.params_cli_defaults()
    got = run_faster(lib, p, pb)
    ll = np.zeros(8, np.int32)
    lib.dd_last_launch(C.byref((C.c_int32 * 8).from_buffer(ll)))
    assert int(ll[0]) == int(groups)
    assert_same_faster(got, _oracle.batch(p, pb, nthreads=8, faster=True), pb)


def test_faster_device_pointer_path(lib):
    import torch
    from dindel_tgi_amd.device import DeviceBatch
    pb = synth.generate(5, H=6, R=30, seed=131, mixed_quals=True)
    p = capi.params_cli_defaults()
    dev = DeviceBatch(pb, p, "cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        dev.launch_faster()
    s.synchronize()
    got = dev.results()
    want = run_faster(lib, p, pb)
    for k in CHECKED + ["hpos", "var_covered", "onHap"]:
        n = {"hpos": pb.hpos_len, "var_covered": pb.var_cov_len, "onHap": pb.n_reads}.get(k, pb.n_pairs)
        assert np.array_equal(got[k][:n], want[k][:n]), k
    # and the main model still runs on the same resident batch afterwards (the model switch is per call)
    dev.launch()
    torch.cuda.synchronize()
    want_fb = _oracle.batch(p, pb, nthreads=8)
    assert np.array_equal(dev.results()["ll"][:pb.n_pairs], want_fb["ll"][:pb.n_pairs])


def test_faster_full_size_properties(lib):
    """At a config[1]-like size (oracle too slow to run whole): spot-check 64 windows against the oracle and check
    size-independent properties on the rest: ll <= 0 and finite, offHap flags 0, hpos in range, duplicated windows
    (synth.tile) give identical results."""
    base = synth.generate(64, H=8, R=200, seed=140, mixed_quals=True)
    pb = synth.tile(base, 16)
    p = capi.params_cli_defaults()
    got = run_faster(lib, p, pb)
    want = _oracle.batch(p, base, nthreads=16, faster=True)
    nb = base.n_pairs
    ll = got["ll"][:pb.n_pairs].reshape(16, nb)
    assert (ll == want["ll"][:nb][None, :]).all()
    assert np.isfinite(ll).all() and (ll <= 0).all()
    assert not got["offHap"][:pb.n_pairs].any() and not got["offHapHMQ"][:pb.n_pairs].any()
    hp = got["hpos"][:pb.hpos_len].reshape(16, base.hpos_len)
    assert (hp == want["hpos"][:base.hpos_len][None, :]).all()
    assert got["onHap"][:pb.n_reads].all()


@pytest.mark.parametrize("seed,max_hap,max_read,mld", [(1, 60, 40, 5), (2, 140, 120, 5), (3, 60, 60, 10), (4, 100, 80, 0),
                                                        (5, 200, 170, 10), (6, 30, 300, 3), (7, 400, 90, 5), (8, 700, 250, 11)])
def test_faster_fuzz(lib, seed, max_hap, max_read, mld):
    """The adversarial generator of test_gpu_fuzz.py (tiny alphabets -> many tied diagonals and exact value ties, N bases,
    indel-carrying and junk reads, reads of 1..3 bases, starts far outside the window, uint32 wrap) through the --faster
    model; haplotypes shorter than maxLengthDel make whole haplotypes fail with DD_PAIR_HAPSIZE."""
    from tests.test_gpu_fuzz import make_windows
    rng = np.random.default_rng(3000 + seed)
    ws = make_windows(rng, 100, max_hap, max_read, min_hap=1, with_vars=(seed % 2 == 1))
    p = capi.params_cli_defaults()
    p.maxLengthDel = mld
    p.padCover = int(rng.integers(0, 4))
    p.capMapQualFast = float(rng.choice([5.0, 45.0, 200.0]))
    pb = pack(ws)
    assert_same_faster(run_faster(lib, p, pb), _oracle.batch(p, pb, nthreads=8, faster=True), pb)


def test_faster_launch_is_graph_capturable(lib):
    import torch
    from dindel_tgi_amd.device import DeviceBatch
    pb = synth.generate(3, H=4, R=50, seed=160, mixed_quals=True)
    p = capi.params_cli_defaults()
    dev = DeviceBatch(pb, p, "cuda:0")
    dev.launch_faster()                            # warm-up outside capture (sets the kernel attribute)
    want = {k: v.copy() for k, v in dev.results().items()}
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            dev.launch_faster(stream=s)
    torch.cuda.synchronize()
    for t in dev.out.values():
        t.zero_()
    for _ in range(2):
        g.replay()
    got = dev.results()
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def test_concurrent_host_threads_both_models(lib):
    """The host-pointer entry points keep their device arena, streams and the model switch per host thread: three
    threads hammer both models on different batches at once; every result equals the single-threaded one."""
    import threading
    from tests.test_gpu_parity import run_host_api
    p = capi.params_cli_defaults()
    pbs = [synth.generate(6 + i, H=4 + i, R=40, seed=170 + i, mixed_quals=True) for i in range(3)]
    want = [(run_host_api(lib, p, pb), run_faster(lib, p, pb)) for pb in pbs]
    errors = []

    def work(i):
        try:
            for rep in range(6):
                a = run_faster(lib, p, pbs[i]) if (rep + i) % 2 else run_host_api(lib, p, pbs[i])
                w = want[i][1] if (rep + i) % 2 else want[i][0]
                for k in ("ll", "status", "firstBase", "lastBase", "offHap"):
                    assert np.array_equal(a[k][:pbs[i].n_pairs], w[k][:pbs[i].n_pairs]), (i, rep, k)
                assert np.array_equal(a["hpos"][:pbs[i].hpos_len], w["hpos"][:pbs[i].hpos_len]), (i, rep)
        except Exception as e:          # noqa: BLE001 - surfaced below
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
