"""Randomised GPU-vs-oracle parity over many small adversarial windows: tiny alphabets (ties everywhere), N bases,
reads with inserted / deleted / mutated segments, random start positions (overlapping, far away, wrapped), every
quality from a 50-entry table, every maxLengthDel, bMid overrides.  Bit-equality is required, as everywhere."""
import numpy as np
import pytest

from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import ReadRec, Window, pack, phred_to_prob
from tests import _oracle
from tests.test_gpu_parity import assert_same, run_host_api

pytestmark = pytest.mark.gpu


def make_windows(rng, n, max_hap, max_read, min_hap, with_vars=False):
    quals = np.concatenate([phred_to_prob(np.arange(0, 45)), [0.5, 0.25, 0.95, 0.951, 0.949]])
    mapqs = [1e-16, 0.3, 0.9, 0.99, 0.9999, 1 - 1e-10, 1 - 1e-16]
    ws = []
    for _ in range(n):
        alpha = list(rng.choice(["A", "AC", "ACG", "ACGT", "ACGTN", "AT", "ACGTRYn"], 1)[0])
        hl = int(rng.integers(min_hap, max_hap + 1))
        ref = "".join(rng.choice(alpha, hl))
        if "N" not in alpha and rng.random() < 0.2:                       # a run of N (changeINStoN)
            p0 = int(rng.integers(0, hl)); ref = ref[:p0] + "N" * min(3, hl - p0) + ref[p0 + min(3, hl - p0):]
        haps = [ref]
        for _h in range(int(rng.integers(0, 4))):
            p0 = int(rng.integers(0, hl)); ln = int(rng.integers(1, 8))
            h = ref[:p0] + ref[p0 + ln:] if rng.random() < 0.5 else ref[:p0] + "".join(rng.choice(list("ACGT"), ln)) + ref[p0:]
            if len(h) >= min_hap:
                haps.append(h)
        reads = []
        for _r in range(int(rng.integers(1, 9))):
            src = haps[int(rng.integers(0, len(haps)))]
            L = int(rng.integers(1, max_read + 1))
            off = int(rng.integers(-L, len(src) + 3))
            s = [src[i] if 0 <= i < len(src) else str(rng.choice(list("ACGT"))) for i in range(off, off + L)]
            k = rng.random()
            if k < 0.2 and L > 4:                                          # delete a few read bases
                p0 = int(rng.integers(1, L - 2)); del s[p0:p0 + int(rng.integers(1, 4))]
            elif k < 0.4:                                                  # insert a few
                p0 = int(rng.integers(0, len(s) + 1)); s[p0:p0] = list(rng.choice(list("ACGTN"), int(rng.integers(1, 5))))
            elif k < 0.5:
                s = list(rng.choice(list("ACGT"), len(s)))                 # junk
            for i in range(len(s)):
                if rng.random() < 0.03:
                    s[i] = str(rng.choice(list("ACGTNR")))
            if not s:
                s = ["A"]
            start = 1000 + off if rng.random() < 0.8 else int(rng.choice([0, 5, 900, 2000, 0xFFFFFFFF, 1000 + len(src)]))
            reads.append(ReadRec("".join(s), rng.choice(quals, len(s)), float(rng.choice(mapqs)), start, unmapped=bool(rng.random() < 0.1)))
        if with_vars:        # random haplotype variants: (startRead, endRead) and (leftFlank, rightFlank, kind) incl. edge values
            hv, hf = [], []
            for h in haps:
                k = int(rng.integers(0, 4))
                v, f = [], []
                for _v in range(k):
                    a = int(rng.integers(-2, len(h) + 2)); b = a + int(rng.integers(0, 9))
                    v.append((a, b))
                    fl = int(rng.choice([-1, 0, 1, 2, 3, a - 1, a])); fr = fl + int(rng.integers(0, 12))
                    f.append((fl, fr, int(rng.integers(0, 3))))
                hv.append(v); hf.append(f)
            ws.append(Window(1000, haps, reads, hap_vars=hv, hap_var_flanks=hf))
        else:
            ws.append(Window(1000, haps, reads))
    return ws


@pytest.mark.parametrize("seed,max_hap,max_read,mld,bmid", [(1, 60, 40, 5, -1), (2, 140, 120, 5, -1), (3, 60, 60, 10, -1),
                                                              (4, 100, 80, 0, -1), (5, 100, 80, 1, -1), (6, 90, 70, 7, -1),
                                                              (7, 90, 70, 11, -1), (8, 130, 100, 5, 0), (9, 130, 100, 5, 7),
                                                              (10, 200, 170, 10, -1), (11, 30, 300, 3, -1)])
def test_fuzz(lib, seed, max_hap, max_read, mld, bmid):
    rng = np.random.default_rng(1000 + seed)
    ws = make_windows(rng, 120, max_hap, max_read, min_hap=max(mld, 1), with_vars=(seed % 2 == 0))
    p = capi.params_cli_defaults()
    p.maxLengthDel = mld
    p.bMid = bmid
    pb = pack(ws)
    got = run_host_api(lib, p, pb)
    want = _oracle.batch(p, pb, nthreads=8)
    assert_same(got, want, pb)


@pytest.mark.parametrize("force", ["0", "1"])
@pytest.mark.parametrize("seed,min_hap,max_hap,mld", [(21, 12, 60, 3), (22, 12, 60, 11), (23, 70, 125, 7), (24, 70, 125, 10),
                                                       (25, 130, 190, 5), (26, 130, 190, 9), (27, 200, 250, 2), (28, 200, 250, 10),
                                                       (29, 260, 380, 5), (30, 400, 500, 4), (31, 560, 700, 8), (32, 3, 40, 1)])
def test_fuzz_every_build(lib, monkeypatch, force, seed, min_hap, max_hap, mld):
    """Same fuzz, but pinning the back-pointer placement (DD_FORCE_GBT=0: LDS tile, 1: HBM scratch = register-lean
    build) so that every (K, D-build, variant) instantiation meets adversarial input, not only the one the plan picks."""
    monkeypatch.setenv("DD_FORCE_GBT", force)
    rng = np.random.default_rng(2000 + seed)
    ws = make_windows(rng, 40, max_hap, 90, min_hap=min_hap)
    p = capi.params_cli_defaults()
    p.maxLengthDel = mld
    pb = pack(ws)
    got = run_host_api(lib, p, pb)
    assert (capi.last_launch()["D"] >= 100) == (force == "1")
    want = _oracle.batch(p, pb, nthreads=8)
    assert_same(got, want, pb)
