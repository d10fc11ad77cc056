"""Host logic: packing, slicing into rank shards, synthetic generator determinism."""
import numpy as np

from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.batch import ReadRec, Window, pack, pair_slices
from dindel_tgi_amd.shard import window_block


def test_pack_roundtrip_and_quality_dedup():
    w0 = Window(100, ["ACGTACGTAC", "ACGTCGTAC"], [ReadRec("ACGT", [0.9, 0.99, 0.999, 0.9], 0.99, 98),
                                                   ReadRec("CGTAC", [0.999] * 5, 0.9, 103, unmapped=True)],
                hap_vars=[[], [(3, 4)]])
    w1 = Window(2 ** 32 - 1, ["TTTTTT"], [ReadRec("TT", [0.5, 0.5], 0.99, -1)])
    pb = pack([w0, w1])
    a = pb.a
    assert pb.n_windows == 2 and pb.n_haps == 3 and pb.n_reads == 3 and pb.n_pairs == 2 * 2 + 1
    assert bytes(a["hap_seq"]) == b"ACGTACGTACACGTCGTACTTTTTT"
    assert a["read_seq_off"].tolist() == [0, 4, 9, 11]
    assert sorted(a["qual_table"].tolist()) == [0.5, 0.9, 0.99, 0.999]
    assert [a["qual_table"][i] for i in a["read_qidx"][:4]] == [0.9, 0.99, 0.999, 0.9]
    assert a["read_flags"].tolist() == [0, 1, 0]
    assert a["read_start"][2] == 0xFFFFFFFF and a["win_hap_start"][1] == 0xFFFFFFFF
    assert pb.hpos_len == 2 * 9 + 1 * 2 and pb.var_cov_len == 1 * 2
    assert pb.cells == (10 + 9) * 9 + 6 * 2
    assert pair_slices(pb, 1) == (4, 1, 1, 18, 2, 9)


def test_slices_partition_the_batch():
    pb = synth.generate(9, H=3, R=7, L=30, hap_len=40, seed=3, vary_read_len=True, mixed_quals=True)
    world = 4
    tot_pairs = tot_cells = 0
    for r in range(world):
        w0, w1 = window_block(pb.n_windows, r, world)
        s = pb.slice_windows(w0, w1)
        assert s.n_windows == w1 - w0
        assert s.n_pairs == pb.win_pair_off[w1] - pb.win_pair_off[w0]
        assert bytes(s.a["read_seq"]) == bytes(pb.a["read_seq"][pb.a["read_seq_off"][pb.a["win_read_off"][w0]]:
                                                               pb.a["read_seq_off"][pb.a["win_read_off"][w1]]])
        tot_pairs += s.n_pairs
        tot_cells += s.cells
    assert tot_pairs == pb.n_pairs and tot_cells == pb.cells
    blocks = [window_block(10, r, 4) for r in range(4)]
    assert blocks == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_synth_is_deterministic_and_shaped():
    a = synth.generate(3, H=8, R=200, seed=1)
    b = synth.generate(3, H=8, R=200, seed=1)
    c = synth.generate(3, H=8, R=200, seed=2)
    assert all(np.array_equal(a.a[k], b.a[k]) for k in a.a)
    assert not np.array_equal(a.a["read_seq"], c.a["read_seq"])
    assert a.n_pairs == 3 * 8 * 200 and a.max_read_len == 100 and 117 <= a.max_hap_len <= 123
    assert set(bytes(a.a["hap_seq"])) <= set(b"ACGT")


def test_slice_windows_carries_variants_flanks_and_mates():
    """PackedBatch.slice_windows (how ranks shard a job) must carry every optional array: haplotype variants, flank
    intervals, mate / library inputs.  Oracle results of the slices, concatenated, equal those of the whole batch."""
    from tests import _oracle
    from tests.test_gpu_fuzz import make_windows
    from tests.test_insert_prior import library
    rng = np.random.default_rng(21)
    ws = make_windows(rng, 9, 70, 50, min_hap=5, with_vars=True)
    libs = [library(rng, 200, 80), library(rng, 50, 20)]
    for w in ws:
        for r in w.reads:
            r.paired = True; r.mate_same_tid = bool(rng.random() < 0.8); r.mate_reverse = bool(rng.random() < 0.5)
            r.mate_pos = int(r.start % 2 ** 30) + int(rng.integers(-100, 100)); r.mate_len = int(rng.choice([-1, 50])); r.lib = int(rng.integers(0, 2))
    pb = pack(ws, libraries=libs)
    p = capi.params_cli_defaults()
    p.mapUnmappedReads = 1
    full = _oracle.batch(p, pb, nthreads=4)
    cuts = [0, 2, 3, 7, 9]
    parts = [_oracle.batch(p, pb.slice_windows(a, b), nthreads=2) for a, b in zip(cuts[:-1], cuts[1:])]
    sl = [pb.slice_windows(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    for k, n in (("ll", "n_pairs"), ("hpos", "hpos_len"), ("var_covered", "var_cov_len"), ("var_fcov", "var_cov_len"), ("onHap", "n_reads")):
        cat = np.concatenate([r[k][:getattr(s, n)] for r, s in zip(parts, sl)])
        assert np.array_equal(cat, full[k][:getattr(pb, n)]), k
    assert sum(s.var_cov_len for s in sl) == pb.var_cov_len > 0
