"""mapUnmappedReads: the library insert-size prior of paired reads at the join (reference ObservationModelFB.cpp:279-303,
Library.hpp:60-66; switched on by --libFile, DInDel.cpp:4268-4272).

CPU: the oracle's prior term against a direct Python evaluation of the reference formula — a read whose only change is
the mate position moves its log-likelihood by exactly the change of (pinsert + constants) when the alignment stays put.
GPU: bit-equality with the oracle on batches mixing every combination of the mate flags."""
import math

import numpy as np
import pytest

from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import ReadRec, Window, pack
from tests import _oracle


def library(rng, maxins, mode):
    counts = np.exp(-0.5 * ((np.arange(maxins) - mode) / (0.15 * mode + 5.0)) ** 2) + 1e-4 * rng.random(maxins)
    probs = np.maximum(counts / counts.sum(), 1e-10)                      # Library::calcProb (Library.hpp:100-111)
    srt = np.sort(probs)
    acc, p95 = 0.0, srt[-1]
    for x in range(len(srt) - 1, 0, -1):                                  # (:116-123)
        acc += srt[x]
        if acc > 0.95:
            p95 = srt[x]
            break
    return probs, float(p95)


def windows_with_mates(rng, n, libs):
    ws = []
    for _ in range(n):
        hap = "".join(rng.choice(list("ACGT"), int(rng.integers(60, 150))))
        hap2 = hap[:30] + hap[33:]
        reads = []
        for _r in range(int(rng.integers(4, 14))):
            src = hap if rng.random() < 0.5 else hap2
            L = int(rng.integers(20, 90))
            off = int(rng.integers(-10, len(src) - 10))
            s = "".join(src[j] if 0 <= j < len(src) else str(rng.choice(list("ACGT"))) for j in range(off, off + L))
            if rng.random() < 0.15:
                s = "".join(rng.choice(list("ACGT"), L))                  # the orphan mate of a mapped read: sequence unrelated
            lib = int(rng.integers(0, len(libs)))
            rev = bool(rng.random() < 0.5)
            mode = int(np.argmax(libs[lib][0]))
            mpos = 1000 + off + (mode if not rev else -mode) + int(rng.integers(-60, 60))
            reads.append(ReadRec(s, (1.0 - 10.0 ** (-rng.integers(5, 41, L) / 10.0)).tolist(),
                                 1.0 - 10.0 ** (-int(rng.integers(0, 61)) / 10.0), 1000 + off,
                                 unmapped=bool(rng.random() < 0.2), paired=bool(rng.random() < 0.85),
                                 mate_unmapped=bool(rng.random() < 0.1), mate_reverse=rev,
                                 mate_same_tid=bool(rng.random() < 0.9), mate_pos=mpos,
                                 mate_len=int(rng.choice([-1, 36, 76, 100])), lib=lib))
        ws.append(Window(1000, [hap, hap2], reads))
    return ws


def test_oracle_prior_term_is_the_reference_formula():
    rng = np.random.default_rng(5)
    probs, p95 = library(rng, 600, 300)
    hap = "".join(rng.choice(list("ACGT"), 120))
    read = hap[20:80]
    p = capi.params_cli_defaults()
    p.mapUnmappedReads = 1
    base = dict(seq=read, qual=[0.999] * 60, mapQual=0.9999, start=1020)

    def ll_of(**kw):
        w = Window(1000, [hap], [ReadRec(**base, **kw)])
        return _oracle.batch(p, pack([w], libraries=[(probs, p95)]))

    plain = ll_of()                                                        # not paired: pinsert = 0
    assert plain["hpos"][:60].tolist() == list(range(20, 80))
    for rev, mpos, mlen in [(False, 1320, 60), (True, 700, 60), (False, 1021, 50), (True, 5000, 76), (False, 1320, -1)]:
        got = ll_of(paired=True, mate_reverse=rev, mate_same_tid=True, mate_pos=mpos, mate_len=mlen)
        assert got["hpos"][:60].tolist() == list(range(20, 80))
        if mlen == -1:
            assert got["ll"][0] == plain["ll"][0]
            continue
        # the join state is (on haplotype base x, not inserted) with x = hpos[bMid]+1; bMid = 30 for this placement
        bMid, x = 30, 20 + 30 + 1
        d = abs(1000 + x - bMid - (mpos + mlen)) if rev else abs(1000 + x + 60 - bMid - mpos)
        pin = math.log(probs[min(d, len(probs) - 1)])
        # prior = (pinsert + log(1-pOff)) + logpIns: the same sum with pinsert = 0 gives the plain value
        assert got["ll"][0] == pytest.approx(plain["ll"][0] + pin, rel=1e-13)
    # mate on another chromosome / unmapped mate: no prior
    assert ll_of(paired=True, mate_same_tid=False, mate_pos=1300, mate_len=60)["ll"][0] == plain["ll"][0]
    assert ll_of(paired=True, mate_same_tid=True, mate_unmapped=True, mate_pos=1300, mate_len=60)["ll"][0] == plain["ll"][0]
    # off-haplotype state takes the 95th-percentile probability (:289, :298)
    junk = "".join(rng.choice(list("ACGT"), 60))
    w = Window(1000, [hap], [ReadRec(junk, [0.999] * 60, 0.9, 1020, paired=True, mate_same_tid=True, mate_pos=1320, mate_len=60),
                             ReadRec(junk, [0.999] * 60, 0.9, 1020)])
    res = _oracle.batch(p, pack([w], libraries=[(probs, p95)]))
    assert res["offHap"][0] == 1 and res["offHap"][1] == 1
    assert res["llOff"][0] == pytest.approx(res["llOff"][1] + math.log(p95), rel=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_insert_prior_gpu_parity(lib, seed):
    from tests.test_gpu_parity import assert_same, run_host_api
    rng = np.random.default_rng(400 + seed)
    libs = [library(rng, int(rng.integers(50, 900)), int(rng.integers(20, 400))) for _ in range(int(rng.integers(1, 4)))]
    pb = pack(windows_with_mates(rng, 25, libs), libraries=libs)
    for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
        p.mapUnmappedReads = 1
        got = run_host_api(lib, p, pb)
        want = _oracle.batch(p, pb, nthreads=8)
        assert_same(got, want, pb)
        p.mapUnmappedReads = 0                                            # same arrays, option off: the prior must vanish
        off = run_host_api(lib, p, pb)
        assert_same(off, _oracle.batch(p, pb, nthreads=8), pb)
        assert (off["ll"][:pb.n_pairs] != got["ll"][:pb.n_pairs]).any()


@pytest.mark.gpu
def test_faster_model_ignores_the_insert_prior(lib):
    """ObservationModelS has no insert-size prior (Faster.cpp never reads the library): same results with the option on."""
    from tests.test_gpu_faster import run_faster
    rng = np.random.default_rng(31)
    libs = [library(rng, 300, 120)]
    pb = pack(windows_with_mates(rng, 8, libs), libraries=libs)
    p = capi.params_cli_defaults()
    off = run_faster(lib, p, pb)
    p.mapUnmappedReads = 1
    on = run_faster(lib, p, pb)
    assert np.array_equal(on["ll"][:pb.n_pairs], off["ll"][:pb.n_pairs])
    assert np.array_equal(on["hpos"][:pb.hpos_len], off["hpos"][:pb.hpos_len])
    bare = pack(windows_with_mates(np.random.default_rng(31), 8, libs))          # no mate arrays at all: still fine for --faster
    assert bare.mate is None
    run_faster(lib, p, bare)


@pytest.mark.gpu
def test_insert_prior_device_pointer_path(lib):
    import torch
    from dindel_tgi_amd.device import DeviceBatch
    rng = np.random.default_rng(77)
    libs = [library(rng, 500, 250)]
    pb = pack(windows_with_mates(rng, 10, libs), libraries=libs)
    p = capi.params_cli_defaults()
    p.mapUnmappedReads = 1
    dev = DeviceBatch(pb, p, "cuda:0")
    dev.launch()
    torch.cuda.synchronize()
    want = _oracle.batch(p, pb, nthreads=8)
    res = dev.results()
    for k in ("ll", "llOn", "llOff", "offHap", "offHapHMQ"):
        assert np.array_equal(res[k][:pb.n_pairs], want[k][:pb.n_pairs]), k
    assert np.array_equal(res["hpos"][:pb.hpos_len], want["hpos"][:pb.hpos_len])


@pytest.mark.gpu
def test_insert_prior_through_the_cpp_adapter():
    """dindel::LikelihoodEngine with Read mate fields and dindel::Library (Library.hpp:78-128 restated) against the oracle
    fed with a Python evaluation of the same histogram normalisation."""
    from tests import _host
    rng = np.random.default_rng(9)
    counts = np.floor(1000 * np.exp(-0.5 * ((np.arange(400) - 180) / 30.0) ** 2)) + 1.0
    z = 0.0
    for c in counts:
        z += c                                                           # sequential sum, like Library::calcProb
    probs = np.maximum(counts / z, 1e-10)
    srt = np.sort(probs)
    acc, p95 = 0.0, srt[-1]
    for x in range(len(srt) - 1, 0, -1):
        acc += srt[x]
        if acc > 0.95:
            p95 = float(srt[x])
            break
    ws = windows_with_mates(rng, 1, [(probs, p95)])
    w = ws[0]
    p = capi.params_cli_defaults()
    p.mapUnmappedReads = 1
    want = _oracle.batch(p, pack(ws, libraries=[(probs, p95)]), nthreads=4)
    mate = [((1 if r.paired else 0) | (2 if r.mate_unmapped else 0) | (4 if r.mate_reverse else 0) | (8 if r.mate_same_tid else 0),
             r.mate_pos, r.mate_len, 0) for r in w.reads]
    res = _host.compute_window_mates(w.haps, [r.seq for r in w.reads], [r.qual for r in w.reads], [r.mapQual for r in w.reads],
                                     [float(r.start) for r in w.reads], [int(r.unmapped) for r in w.reads], 1000, p, mate, [counts])
    R = len(w.reads)
    for h in range(len(w.haps)):
        for r in range(R):
            ml = res["liks"][h][r]
            assert ml["ll"] == want["ll"][h * R + r] and ml["llOff"] == want["llOff"][h * R + r], (h, r)
            assert (ml["offHap"], ml["offHapHMQ"]) == (int(want["offHap"][h * R + r]), int(want["offHapHMQ"][h * R + r]))
