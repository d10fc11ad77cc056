"""CPU-only checks of the --faster restatement (oracle/dd_oracle.c: ddo_pair_fast / ddo_batch_fast) on adversarial
windows: the batch driver against the per-pair function, plus properties of ObservationModelS that hold for any input
(Faster.cpp:491/:529 make offHap / offHapHMQ always false; hpos values stay inside the haplotype; a read identical to a
haplotype segment scores higher on that haplotype than on a shuffled one).  Runs under ASan via tests/sanitize_cpu.sh."""
import numpy as np

from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import ReadRec, Window, pack
from tests import _oracle
from tests.test_gpu_fuzz import make_windows


def test_batch_driver_equals_pair_function():
    rng = np.random.default_rng(77)
    ws = make_windows(rng, 25, 120, 90, min_hap=1)
    p = capi.params_cli_defaults()
    pb = pack(ws)
    res = _oracle.batch(p, pb, nthreads=4, faster=True)
    pair = hp = 0
    n_ok = 0
    for w in ws:
        SL = sum(len(r.seq) for r in w.reads)
        for h in w.haps:
            off = 0
            for r in w.reads:
                o, hpos = _oracle.pair_fast(h, r.seq, r.qual, r.mapQual, r.start, w.hap_start, p)
                assert res["status"][pair] == o.status
                if o.status == 0:
                    n_ok += 1
                    assert res["ll"][pair] == o.ll
                    got = res["hpos"][hp + off:hp + off + len(r.seq)]
                    assert got.tolist() == _oracle.keyed_hpos(o, hpos)            # inserted bases carry their key
                    assert capi.hpos_reference_codes(got).tolist() == hpos       # ... and map back to MLAlignment's codes
                    assert (res["firstBase"][pair], res["lastBase"][pair]) == (o.firstBase, o.lastBase)
                    assert res["offHap"][pair] == 0 and res["offHapHMQ"][pair] == 0
                    on = [x for x in hpos if x >= 0]
                    assert all(x < len(h) for x in on)
                    assert (o.firstBase, o.lastBase) == ((min(on), max(on)) if on else (-1, -1))
                pair += 1
                off += len(r.seq)
            hp += SL
    assert n_ok > 200


def test_matching_haplotype_scores_higher():
    rng = np.random.default_rng(78)
    p = capi.params_cli_defaults()
    for _ in range(20):
        hap = "".join(rng.choice(list("ACGT"), 120))
        other = "".join(rng.permutation(list(hap)))
        o = int(rng.integers(0, 40))
        read = hap[o:o + 70]
        a, _ = _oracle.pair_fast(hap, read, 0.999, 0.9999, 1000 + o, 1000, p)
        b, _ = _oracle.pair_fast(other, read, 0.999, 0.9999, 1000 + o, 1000, p)
        assert a.status == 0 and b.status == 0 and a.ll > b.ll and a.ll > -1.0


def test_short_reads_and_short_haplotypes_status():
    p = capi.params_cli_defaults()
    w = Window(1000, ["ACGTACGTACGTAAA", "ACG"], [ReadRec("ACG", [0.99] * 3, 0.99, 1000), ReadRec("ACGTAC", [0.99] * 6, 0.99, 1000)])
    res = _oracle.batch(p, pack([w]), faster=True)
    assert res["status"][:4].tolist() == [capi.DD_PAIR_NAN, 0, capi.DD_PAIR_HAPSIZE, capi.DD_PAIR_HAPSIZE]
    assert res["onHap"][:2].tolist() == [0, 1]
