"""GPU parity on the edge cases of the path (bit-equal to the oracle, through the C ABI):
empty / ragged windows, extreme sizes, N bases, homopolymers, repeats with exact ties, bMid corner cases,
non-overlapping and unmapped reads, quality extremes, the hapSize error status."""
import ctypes as C

import numpy as np
import pytest

from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import ReadRec, Window, alloc_result, pack, phred_to_prob
from tests import _oracle
from tests.test_gpu_parity import assert_same, run_host_api

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(12345)


def rnd(n, alphabet="ACGT"):
    return "".join(RNG.choice(list(alphabet), n))


def mutate(s, rate=0.02):
    out = list(s)
    for i in range(len(out)):
        if RNG.random() < rate:
            out[i] = RNG.choice(list("ACGT"))
    return "".join(out)


def reads_from(hap, n, L, start0=1000, q=0.999, mq=0.9999, junk=0.1):
    reads = []
    for _ in range(n):
        Lr = int(L)
        off = int(RNG.integers(-Lr // 2, max(1, len(hap) - Lr // 2)))
        seq = "".join(hap[i] if 0 <= i < len(hap) else RNG.choice(list("ACGT")) for i in range(off, off + Lr))
        if RNG.random() < junk:
            seq = rnd(Lr)
        reads.append(ReadRec(mutate(seq), [q] * Lr, mq, start0 + off))
    return reads


def check(lib, windows, params=None):
    p = params or capi.params_cli_defaults()
    pb = pack(windows)
    got = run_host_api(lib, p, pb)
    want = _oracle.batch(p, pb, nthreads=8)
    assert_same(got, want, pb)
    return pb, got


def test_ragged_and_empty_windows(lib):
    h1, h2 = rnd(90), rnd(131)
    ws = [Window(1000, [h1, h1[:40] + h1[43:]], reads_from(h1, 7, 50)),
          Window(5000, [h2], []),                                   # window without reads
          Window(7000, [], reads_from(h2, 3, 30)),                  # window without haplotypes
          Window(9000, [h2, h2[:60] + "TT" + h2[60:], h2[:70] + h2[71:]],
                 [ReadRec(h2[10:11], [0.99], 0.99, 9010),           # L = 1
                  ReadRec(h2[10:12], [0.99, 0.9], 0.99, 9010),      # L = 2
                  ReadRec(h2[5:125], [0.999] * 120, 0.9999, 9005)])]
    pb, got = check(lib, ws)
    assert pb.n_pairs == 2 * 7 + 0 + 0 + 3 * 3
    assert got["onHap"][7:10].tolist() == [0, 0, 0]                 # reads of the hap-less window
    # the same degenerate windows through the --faster model and the N1 entry points
    from tests.test_gpu_faster import assert_same_faster, run_faster
    p = capi.params_cli_defaults()
    assert_same_faster(run_faster(lib, p, pb), _oracle.batch(p, pb, nthreads=4, faster=True), pb)
    b = pb.ctypes_batch()
    hh = np.zeros(pb.n_windows + 1, np.int64)
    assert lib.dd_pair_sum_offsets(C.byref(b), hh.ctypes.data_as(capi.c_i64p)) == 0
    assert hh.tolist() == [0, 4, 5, 5, 14]
    ns = int(hh[-1])
    prior = np.zeros(ns); filt = np.zeros(pb.n_haps, np.uint8); nc = np.ones(pb.n_haps, np.int32)
    sums = np.zeros(ns); post = np.zeros(ns); pairs = np.zeros(4 * pb.n_windows, np.int32); vals = np.zeros(3 * pb.n_windows)
    rc = lib.dd_map_pairs(C.byref(b), got["ll"].ctypes.data_as(capi.c_f64p), prior.ctypes.data_as(capi.c_f64p),
                          filt.ctypes.data_as(C.POINTER(C.c_uint8)), nc.ctypes.data_as(capi.c_i32p), sums.ctypes.data_as(capi.c_f64p),
                          post.ctypes.data_as(capi.c_f64p), pairs.ctypes.data_as(capi.c_i32p), vals.ctypes.data_as(capi.c_f64p), 0)
    assert rc == 0, capi.last_error()
    assert sums[4] == 0.0                                           # the read-less window: an empty sum
    assert pairs[8:12].tolist() == [-1, -1, -1, -1]                 # the haplotype-less window: no pair at all
    want = np.zeros(ns)
    _oracle.load().ddo_pair_sums(C.byref(b), got["ll"].ctypes.data_as(capi.c_f64p), want.ctypes.data_as(capi.c_f64p))
    np.testing.assert_allclose(sums, want, rtol=1e-12, atol=0)


def test_n_bases_and_iupac_in_reads(lib):
    hap = rnd(100)
    hapN = hap[:50] + "NNN" + hap[53:]                              # --changeINStoN style haplotype
    reads = reads_from(hap, 12, 60)
    reads.append(ReadRec(hap[20:45] + "N" + hap[46:80], [0.999] * 60, 0.9999, 1020))
    reads.append(ReadRec(hap[20:45] + "R" + hap[46:60] + "nY" + hap[62:80], [0.99] * 60, 0.999, 1020))
    reads.append(ReadRec("N" * 40, [0.9] * 40, 0.99, 1030))
    check(lib, [Window(1000, [hap, hapN], reads)])


def test_homopolymers_incl_last_base_quirk(lib):
    haps = []
    for run in (1, 2, 3, 5, 9, 10, 11, 15, 30, 60):
        left, right = rnd(35), rnd(30)
        haps.append(left + "A" * run + right)                       # run in the middle
        haps.append(left + rnd(20) + "T" * run)                     # run ending at the last base: E[hapSize-1] (:1702)
    ws = []
    for i in range(0, len(haps), 4):
        hs = haps[i:i + 4]
        rs = []
        for h in hs:
            rs += reads_from(h, 4, 45)
            mid = len(h) // 2
            rs.append(ReadRec(h[mid - 20:mid] + h[mid + 1:mid + 21], [0.999] * 40, 0.9999, 1000 + mid - 20))   # 1-bp deletion
        ws.append(Window(1000, hs, rs))
    check(lib, ws)


def test_repeats_exact_ties(lib):
    """Tandem repeats: many alignments tie exactly; pins updateMax's tie-break and the near-tie join replay."""
    ws = []
    for unit in ("AC", "CAG", "T", "AAG", "ACGT"):
        hap = rnd(15) + unit * (60 // len(unit)) + rnd(15)
        rep = unit * 40
        reads = [ReadRec(rep[:k], [0.999] * k, 0.9999, 1015) for k in (12, 24, 30)]
        reads += [ReadRec(unit * 40, [0.99] * len(unit * 40), 0.99, 1000)]                 # longer than the repeat
        reads += reads_from(hap, 6, 40)
        ws.append(Window(1000, [hap, hap[:20] + hap[20 + len(unit):], rnd(len(hap))], reads))
    ws.append(Window(1000, ["A" * 80], [ReadRec("A" * 30, [0.999] * 30, 0.9999, 1010), ReadRec("A" * 90, [0.99] * 90, 0.9, 990)]))
    for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
        check(lib, ws, p)


def test_bmid_corner_cases(lib):
    hap = rnd(120)
    reads = [ReadRec(hap[0:50], [0.999] * 50, 0.9999, 1000 - 49),       # overlaps only its last base
             ReadRec(hap[70:120], [0.999] * 50, 0.9999, 1000 + 120),    # starts exactly at hapEnd
             ReadRec(hap[70:120], [0.999] * 50, 0.9999, 1000 + 121),    # beyond hapEnd -> L/2
             ReadRec(hap[0:50], [0.999] * 50, 0.9999, 1000 - 50),       # ends before hapStart -> L/2
             ReadRec(hap[30:80], [0.999] * 50, 0.9999, 0xFFFFFFFF),     # uint32(-1.0) start
             ReadRec(hap[30:80], [0.999] * 50, 0.9999, 3),              # far left of the window
             ReadRec(hap[30:80], [0.999] * 50, 0.9999, 1030, unmapped=True)]
    ws = [Window(1000, [hap, hap[:60] + hap[62:]], reads), Window(10, [hap], reads[:3]), Window(0xFFFFFF00, [hap], reads)]
    check(lib, ws)
    for bmid in (0, 10, 49, 500):
        p = capi.params_cli_defaults()
        p.bMid = bmid
        check(lib, ws[:1], p)


def test_quality_extremes_and_full_tables(lib):
    hap = rnd(110)
    quals = np.concatenate([phred_to_prob(np.arange(0, 94)), [1e-16, 0.5, 0.25, 0.95, 0.950000001, 0.949999999]])
    mqs = [1e-16, 0.5, 0.9, 0.99, 1 - 1e-10, 1 - 1e-11, 1 - 1e-16]
    reads = []
    for i in range(40):
        L = 50
        off = int(RNG.integers(0, 60))
        q = RNG.choice(quals, L)
        reads.append(ReadRec(mutate(hap[off:off + L], 0.05), q, mqs[i % len(mqs)], 1000 + off))
    check(lib, [Window(1000, [hap, hap[:55] + "GGG" + hap[55:]], reads)])
    # 256 distinct base qualities in one batch
    q256 = np.linspace(0.3, 0.99999, 256)
    reads = [ReadRec(hap[10:74], q256[64 * i:64 * i + 64], 0.999, 1010) for i in range(4)]
    check(lib, [Window(1000, [hap], reads)])


def test_hapsize_error_status_in_mixed_batch(lib):
    p = capi.params_cli_defaults()                                  # maxLengthDel = 5
    hap = rnd(80)
    ws = [Window(1000, [hap, "ACGT"], reads_from(hap, 5, 30),       # second haplotype shorter than maxLengthDel; its
                 hap_vars=[[(20, 22)], [(1, 2), (0, 3)]],           # coverage flags must come back as 0, not stale memory
                 hap_var_flanks=[[(19, 23, 1)], [(0, 3, 2), (1, 2, 1)]]),
          Window(2000, ["ACGTA", hap], reads_from(hap, 4, 30, start0=2000))]   # hapSize == maxLengthDel is allowed
    pb0 = pack(ws)
    arrs, res = alloc_result(pb0, fill=None)
    for k in ("var_covered", "var_fcov"):
        arrs[k][:] = 7                                              # host-side poison: every entry has to be overwritten
    assert lib.dd_compute_likelihoods(C.byref(p), C.byref(pb0.ctypes_batch()), C.byref(res), 0) == 0
    assert arrs["var_covered"][:pb0.var_cov_len].max() <= 1 and arrs["var_fcov"][:pb0.var_cov_len].max() <= 1
    pb, got = check(lib, ws, p)
    st = got["status"][:pb.n_pairs]
    assert st[:5].tolist() == [0] * 5 and st[5:10].tolist() == [capi.DD_PAIR_HAPSIZE] * 5
    assert (st[10:] == 0).all()


@pytest.mark.parametrize("hs,L,n", [(62, 36, 6), (63, 100, 4), (126, 100, 4), (127, 100, 3), (200, 150, 3), (300, 100, 2),
                                     (400, 101, 2), (700, 120, 1), (766, 64, 1)])
def test_all_lane_tilings(lib, hs, L, n):
    """Haplotype lengths on both sides of every K boundary (K = 1,2,3,4,6,8,12), up to the 766-bp limit."""
    hap = rnd(hs)
    var = hap[:hs // 2] + hap[hs // 2 + 2:]
    check(lib, [Window(1000, [hap, var], reads_from(hap, n, L) + reads_from(var, n, L))])
    p = capi.params_struct_defaults()                               # D = 11
    check(lib, [Window(1000, [hap], reads_from(hap, n, min(L, 100)))], p)


def test_maximum_shape_runs_with_hbm_backpointers(lib):
    """766-bp haplotype x 1024-bp reads: the back-pointer tile (512 KiB per pair) cannot live in LDS; the library
    switches to the HBM-scratch build and the result is still bit-equal to the oracle."""
    hap = rnd(766)
    var = hap[:300] + hap[304:]
    reads = reads_from(hap, 2, 1024, junk=0.0) + reads_from(var, 1, 1000, junk=0.0) + reads_from(hap, 1, 37)
    pb, got = check(lib, [Window(1000, [hap, var], reads)])
    assert capi.last_launch()["D"] >= 100                   # HBM-scratch build was selected
    pb, got = check(lib, [Window(1000, [hap[:400]], reads_from(hap[:400], 3, 600, junk=0.0))], capi.params_struct_defaults())
    assert capi.last_launch()["D"] >= 100


def test_long_reads_config5_shape(lib):
    """BASELINE configs[4] shape (250-bp reads, maxLengthDel=10) at small scale."""
    p = capi.params_cli_defaults()
    p.maxLengthDel = 10
    for hs in (120, 160, 200):
        hap = rnd(hs)
        haps = [hap, hap[:hs // 2] + hap[hs // 2 + 3:], hap[:hs // 2] + "ACG" + hap[hs // 2:]]
        check(lib, [Window(1000, haps, reads_from(hap, 6, 250) + reads_from(haps[1], 6, 250))], p)


def test_low_likelihood_pairs_redo_with_ro_chain(lib):
    """Pairs whose best log-likelihood is below -99 cannot use the speculative pass that leaves the RO sink chain
    out (kernel comment at the Dec loop): long, very low quality reads force the redo path; results stay bit-equal."""
    hap = rnd(120)
    reads = []
    for i in range(10):
        L = 250
        off = int(RNG.integers(-100, 100))
        seq = "".join(hap[j] if 0 <= j < len(hap) else RNG.choice(list("ACGT")) for j in range(off, off + L))
        q = phred_to_prob([2 + (i % 3)])[0]
        reads.append(ReadRec(mutate(seq, 0.2), [q] * L, [1 - 1e-16, 0.5, 0.999999][i % 3], 1000 + off))
    reads += reads_from(hap, 4, 100)                      # ordinary pairs in the same batch take the fast pass
    pb, got = check(lib, [Window(1000, [hap, hap[:60] + hap[64:]], reads)])
    assert (got["ll"][:pb.n_pairs] < -99).sum() >= 10 and (got["ll"][:pb.n_pairs] > -99).sum() >= 4
    check(lib, [Window(1000, [hap], reads)], capi.params_struct_defaults())


def test_mixed_length_classes_in_one_batch(lib):
    """A ragged batch spanning several haplotype-length (K) classes and read-length classes, including one that
    needs the HBM-scratch build: the host path launches every class separately; results equal the oracle's and
    equal a one-class-at-a-time evaluation."""
    ws = []
    for hs, Ls in ((50, (30, 64, 65)), (120, (100, 36, 150)), (170, (100, 250)), (130, (128, 129)), (400, (600, 90)), (4, (20,))):
        hap = rnd(hs)
        var = hap[:hs // 2] + hap[hs // 2 + 1:] if hs > 10 else hap
        reads = []
        for L in Ls:
            reads += reads_from(hap, 3, L, junk=0.0)
        ws.append(Window(1000, [hap, var], reads))
    # one window mixing short and long haplotypes (candidate injection can do this) and all read lengths
    hap = rnd(125)
    ws.append(Window(1000, [hap, hap + rnd(60), hap[:60]], reads_from(hap, 4, 40) + reads_from(hap, 4, 100) + reads_from(hap, 2, 300)))
    pb, got = check(lib, ws)
    st = got["status"][:pb.n_pairs]
    assert (st == capi.DD_PAIR_HAPSIZE).sum() == 2 * 3          # the 4-bp haplotypes (maxLengthDel = 5), every read once


def test_hbm_scratch_tile_reuse_many_reads_per_wave(lib):
    """HBM-scratch build under tile reuse: every wave walks many reads of different lengths through the same scratch
    tile (stale-cache hazards would corrupt the traceback of later reads)."""
    p = capi.params_cli_defaults()
    p.maxLengthDel = 10
    hap = rnd(125)
    haps = [hap, hap[:60] + hap[66:], hap[:70] + "ACGTT" + hap[70:]]
    reads = []
    for L in (250, 180, 251, 200, 170, 233):
        reads += reads_from(hap, 12, L, junk=0.05) + reads_from(haps[1], 10, L, junk=0.0)
    pb, got = check(lib, [Window(1000, haps, reads), Window(5000, haps[::-1], reads[::-1])], p)
    assert capi.last_launch()["D"] >= 100


def test_indel_length_ladder_and_end_insertions(lib):
    """SURVEY §8(c) fixture plan: reads carrying a deletion of every length 1..maxLengthDel+1 (the last one is not
    reachable by a single jump), insertions of 1..4 bases at the first / last read base and right at the haplotype
    ends, for maxLengthDel 5 and 10; the number of indels the path reports is cross-checked, not only parity."""
    for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
        D = p.maxLengthDel
        hap = rnd(140)
        reads = []
        for ln in range(1, D + 2):
            s = hap[20:60] + hap[60 + ln:100 + ln]                       # deletion of ln bases after hap base 59
            reads.append(ReadRec(s, [0.999] * len(s), 0.9999, 1020))
        for ln in range(1, 5):
            ins = rnd(ln)
            for s, pos in ((ins + hap[30:90], 1030 - ln), (hap[30:90] + ins, 1030), (ins + hap[0:50], 1000 - ln),
                           (hap[90:140] + ins, 1090), (hap[30:60] + ins + hap[60:90], 1030)):
                reads.append(ReadRec(s, [0.999] * len(s), 0.9999, pos))
        pb, got = check(lib, [Window(1000, [hap], reads)], p)
        nind = got["numIndels"][:pb.n_pairs]
        assert (nind[:D] == 1).all()                                       # one deletion each, lengths 1..maxLengthDel
        assert got["offHap"][D] == 1 or nind[D] != 1                      # maxLengthDel+1: no single-jump explanation


def test_mapping_quality_ladder_and_cap(lib):
    """Mapping qualities from Phred 0 to 120 (mapQualThreshold caps the prior at Phred 100, ObservationModelFB.cpp:276)
    on reads that match, mismatch and do not belong: llOn / llOff / offHap follow the prior bit for bit."""
    hap = rnd(120)
    good, bad = hap[10:110], rnd(100)
    reads = []
    for ph in list(range(0, 64, 3)) + [80, 99, 100, 101, 120, 150]:
        mq = min(1.0 - 10.0 ** (-ph / 10.0), 1.0 - 1e-16) if ph else 0.0
        reads.append(ReadRec(good, [0.999] * 100, mq, 1010))
        reads.append(ReadRec(bad, [0.999] * 100, mq, 1010))
        reads.append(ReadRec(mutate(good, 0.15), [0.99] * 100, mq, 1010))
    pb, got = check(lib, [Window(1000, [hap, hap[:60] + hap[63:]], reads)])
    assert got["offHap"][:pb.n_pairs].any() and not got["offHap"][:pb.n_pairs].all()


def test_iupac_and_soft_masked_haplotype_bytes(lib):
    """The reference compares characters: a haplotype 'R' matches a read 'R' and nothing else, lower case differs from
    upper case, only a haplotype 'N' is a wildcard.  The kernel works on symbol ids from dd_build_symbol_lut."""
    alpha = "ACGTNRYKMacgtn"
    wins = []
    for i in range(8):
        hap = "".join(RNG.choice(list(alpha), int(RNG.integers(30, 150)), p=[.2, .2, .2, .2, .03, .02, .02, .01, .01, .03, .03, .02, .02, .01]))
        hap2 = hap[:20] + hap[23:]
        reads = []
        for _ in range(12):
            src = hap if RNG.random() < 0.5 else hap2
            L = int(RNG.integers(10, 80))
            off = int(RNG.integers(-5, max(1, len(src) - 10)))
            s = "".join(src[j] if 0 <= j < len(src) else RNG.choice(list("ACGT")) for j in range(off, off + L))
            s = "".join(c if RNG.random() > 0.05 else RNG.choice(list("ACGTNRWSacg")) for c in s)
            reads.append(ReadRec(s, phred_to_prob(RNG.integers(2, 42, L)), 0.999, 1000 + off))
        wins.append(Window(1000, [hap, hap2], reads, hap_vars=[[], [(19, 20)]], hap_var_flanks=[[], [(19, 20, 1)]]))
    for p in (capi.params_cli_defaults(), capi.params_struct_defaults()):
        pb, got = check(lib, wins, p)
        assert got["numMismatch"][:pb.n_pairs].any()
    # the --faster model compares raw bytes as well
    from tests.test_gpu_faster import assert_same_faster, run_faster
    pb = pack(wins)
    p = capi.params_cli_defaults()
    assert_same_faster(run_faster(lib, p, pb), _oracle.batch(p, pb, nthreads=8, faster=True), pb)


def test_device_pointer_path_uses_length_classes(lib):
    """dd_launch_device with the dd_build_length_classes summary: a ragged batch (haplotypes 40…400 bp, reads 30…300 bp)
    gets per-class plans and still equals the oracle; the class summary itself is checked against numpy."""
    import torch
    from dindel_tgi_amd.device import DeviceBatch
    wins = []
    for hl, L in ((50, 36), (120, 100), (170, 100), (120, 250), (300, 150), (400, 300), (61, 30), (63, 161)):
        hap = rnd(hl)
        wins.append(Window(1000, [hap, hap[:hl // 2] + hap[hl // 2 + 2:]], reads_from(hap, 9, L)))
    pb = pack(wins)
    p = capi.params_cli_defaults()
    dev = DeviceBatch(pb, p, "cuda:0")
    c = dev.classes
    hl = np.diff(pb.a["hap_seq_off"])
    cls = np.searchsorted(capi.HAP_CLASS_BOUNDS, hl, side="left")
    launches = [c.launch[i] for i in range(c.n_launches)]
    assert sorted({L.hap_class for L in launches}) == sorted(set(cls.tolist()))          # every window has reads of one length: one launch per tiling ...
    assert c.list_len == pb.n_haps and sum(L.list_len for L in launches) == pb.n_haps   # ... and interval
    rl = np.diff(pb.a["read_seq_off"])
    assert max(L.max_read_len for L in launches) == int(rl.max()) and min(L.min_read_len for L in launches) == 1
    dev.launch()
    torch.cuda.synchronize()
    assert_same(dev.results(), _oracle.batch(p, pb, nthreads=8), pb)


@pytest.mark.parametrize("faster", [False, True])
def test_unsupported_window_fails_alone(lib, faster):
    """A window outside the kernel limits (haplotype > 766 bp, read > 1024 bp, empty read) gets DD_PAIR_UNSUPPORTED on its
    own pairs; every other window of the batch equals the oracle — host-pointer path, device-pointer path (per-class
    launches) and both models.  (Round 1 failed the whole batch: one 767-bp haplotype in a fuzz round.)"""
    import torch
    from dindel_tgi_amd.batch import pair_slices
    from dindel_tgi_amd.device import DeviceBatch
    p = capi.params_cli_defaults()
    hapA, hapB, hapL = rnd(110), rnd(300), rnd(767)
    good1 = Window(1000, [hapA, hapA[:50] + hapA[52:]], reads_from(hapA, 9, 70))
    good2 = Window(1000, [hapB], reads_from(hapB, 5, 200, junk=0.0) + reads_from(hapB, 3, 60))
    bad_hap = Window(1000, [hapA, hapL], reads_from(hapA, 4, 50))
    bad_read = Window(1000, [hapA], reads_from(hapA, 2, 60) + [ReadRec(rnd(1025), [0.99] * 1025, 0.99, 1000)])
    bad_empty = Window(1000, [hapA], [ReadRec("", [], 0.99, 1000)] + reads_from(hapA, 2, 60))
    ws = [good1, bad_hap, good2, bad_read, bad_empty, good1]
    bad = [1, 3, 4]
    pb = pack(ws)
    want = _oracle.batch(p, pack([w if i not in bad else Window(1000, [hapA], []) for i, w in enumerate(ws)]), nthreads=4, faster=faster)
    pw = pack([w if i not in bad else Window(1000, [hapA], []) for i, w in enumerate(ws)])

    def check_result(got):
        for w in range(len(ws)):
            p0, H, R, h0, SL, _ = pair_slices(pb, w)
            r0 = int(pb.a["win_read_off"][w])
            if w in bad:
                assert (got["status"][p0:p0 + H * R] == capi.DD_PAIR_UNSUPPORTED).all()
                assert (got["ll"][p0:p0 + H * R] == 0).all() and got["offHapHMQ"][p0:p0 + H * R].all()
                assert not got["onHap"][r0:r0 + R].any()
                continue
            q0, _, _, g0, _, _ = pair_slices(pw, w)
            s0 = int(pw.a["win_read_off"][w])
            for k in ("ll", "llOn", "llOff", "mLogBQ", "status", "offHap", "offHapHMQ", "numIndels", "firstBase", "lastBase"):
                assert np.array_equal(got[k][p0:p0 + H * R], want[k][q0:q0 + H * R]), (w, k)
            assert np.array_equal(got["hpos"][h0:h0 + H * SL], want["hpos"][g0:g0 + H * SL]), w
            assert np.array_equal(got["onHap"][r0:r0 + R], want["onHap"][s0:s0 + R]), w

    arrs, res = alloc_result(pb, fill=None)
    fn = lib.dd_compute_likelihoods_faster if faster else lib.dd_compute_likelihoods
    assert fn(C.byref(p), C.byref(pb.ctypes_batch()), C.byref(res), 0) == 0, capi.last_error()
    check_result(arrs)
    dev = DeviceBatch(pb, p, "cuda:0")
    assert dev.n_skipped == 3
    (dev.launch_faster if faster else dev.launch)()
    torch.cuda.synchronize()
    check_result(dev.results())
    # every window skipped: the call still succeeds and marks them all
    pb2 = pack([bad_hap, bad_read])
    arrs, res = alloc_result(pb2, fill=None)
    assert fn(C.byref(p), C.byref(pb2.ctypes_batch()), C.byref(res), 0) == 0, capi.last_error()
    assert (arrs["status"][:pb2.n_pairs] == capi.DD_PAIR_UNSUPPORTED).all() and not arrs["onHap"][:pb2.n_reads].any()


@pytest.mark.parametrize("faster", [False, True])
def test_window_with_bytes_beyond_the_symbol_table_fails_alone(lib, faster):
    """More than 26 distinct non-ACGTN byte values in the haplotypes of a batch: the main kernel's symbol table cannot number them all, and
    the windows that hold an unnumbered value get DD_PAIR_UNSUPPORTED — the other windows equal the oracle (round 2 failed the call).  The
    --faster kernel compares the bytes themselves and computes every window."""
    import torch
    from dindel_tgi_amd.batch import pair_slices
    from dindel_tgi_amd.device import DeviceBatch
    p = capi.params_cli_defaults()
    hapA, hapB = rnd(110), rnd(140)
    odd = "".join(chr(c) for c in range(98, 127) if chr(c) not in "cgtn")                 # 25 values b ... ~
    assert len(set(odd)) == 25
    good1 = Window(1000, [hapA, hapA[:50] + hapA[52:]], reads_from(hapA, 9, 70))
    hapLow, hapMarks, hapTilde = hapB[:60] + odd[:24] + hapB[60:], hapA[:30] + "!#" + hapA[30:], hapB[:70] + "~" + hapB[70:]
    lower = Window(1000, [hapLow, hapB], reads_from(hapLow, 6, 80))                                # numbered values only
    marks = Window(1000, [hapMarks], reads_from(hapMarks, 5, 60))                                  # two small values: numbered first
    tilde = Window(1000, [hapB, hapTilde], reads_from(hapTilde, 7, 90))                            # '~' is the 27th in byte order
    ws = [good1, lower, marks, tilde, good1]
    pb = pack(ws)
    want = _oracle.batch(p, pb, nthreads=4, faster=faster)
    bad = [] if faster else [3]

    def check_result(got):
        for w in range(len(ws)):
            p0, H, R, h0, SL, _ = pair_slices(pb, w)
            r0 = int(pb.a["win_read_off"][w])
            if w in bad:
                assert (got["status"][p0:p0 + H * R] == capi.DD_PAIR_UNSUPPORTED).all()
                assert (got["ll"][p0:p0 + H * R] == 0).all() and not got["onHap"][r0:r0 + R].any()
                continue
            for k in ("ll", "llOn", "llOff", "mLogBQ", "status", "offHap", "offHapHMQ", "numIndels", "firstBase", "lastBase"):
                assert np.array_equal(got[k][p0:p0 + H * R], want[k][p0:p0 + H * R]), (w, k)
            assert np.array_equal(got["hpos"][h0:h0 + H * SL], want["hpos"][h0:h0 + H * SL]), w
            assert np.array_equal(got["onHap"][r0:r0 + R], want["onHap"][r0:r0 + R]), w

    arrs, res = alloc_result(pb, fill=None)
    fn = lib.dd_compute_likelihoods_faster if faster else lib.dd_compute_likelihoods
    assert fn(C.byref(p), C.byref(pb.ctypes_batch()), C.byref(res), 0) == 0, capi.last_error()
    check_result(arrs)
    if not faster:                                       # device-pointer path: dd_screen_windows' flags travel with the batch
        dev = DeviceBatch(pb, p, "cuda:0")
        assert dev.n_skipped == 1
        dev.launch()
        torch.cuda.synchronize()
        check_result(dev.results())
        # without '~' everything is numbered and computed
        pb3 = pack(ws[:3])
        arrs, res = alloc_result(pb3, fill=None)
        assert fn(C.byref(p), C.byref(pb3.ctypes_batch()), C.byref(res), 0) == 0, capi.last_error()
        assert (arrs["status"][:pb3.n_pairs] != capi.DD_PAIR_UNSUPPORTED).all()


@pytest.mark.parametrize("faster", [False, True])
def test_multi_device_entry_equals_single_call(lib, faster):
    """dd_compute_likelihoods_multi with devices = {0, 0} / {0, 0, 0} (two / three window blocks, each under its own host
    thread, arena and streams, concurrently on the one GPU of this box) writes exactly the single call's arrays."""
    from tests.test_gpu_fuzz import make_windows
    rng = np.random.default_rng(606)
    p = capi.params_cli_defaults()
    ws = make_windows(rng, 90, 160, 120, min_hap=max(p.maxLengthDel, 4), with_vars=True)
    if faster:
        ws = [Window(w.hap_start, w.haps, [r for r in w.reads if len(r.seq) >= 4], hap_vars=w.hap_vars, hap_var_flanks=w.hap_var_flanks) for w in ws]
    pb = pack(ws)
    b = pb.ctypes_batch()
    one, res1 = alloc_result(pb, fill=None)
    fn1 = lib.dd_compute_likelihoods_faster if faster else lib.dd_compute_likelihoods
    fnm = lib.dd_compute_likelihoods_faster_multi if faster else lib.dd_compute_likelihoods_multi
    assert fn1(C.byref(p), C.byref(b), C.byref(res1), 0) == 0, capi.last_error()
    for devs in ([0, 0], [0, 0, 0]):
        many, resm = alloc_result(pb, fill=None)
        d = np.asarray(devs, np.int32)
        for _rep in range(2):                      # the second call reuses the workers' cached arenas
            assert fnm(C.byref(p), C.byref(b), C.byref(resm), d.ctypes.data_as(capi.c_i32p), len(devs)) == 0, capi.last_error()
        n = {"hpos": pb.hpos_len, "var_covered": pb.var_cov_len, "var_fcov": pb.var_cov_len, "onHap": pb.n_reads}
        ok = one["status"][:pb.n_pairs] == 0
        for k in one:
            if k == "hpos" and not ok.all():
                continue                           # hpos of a failed pair is not written by either call
            m = n.get(k, pb.n_pairs)
            assert np.array_equal(one[k][:m], many[k][:m]), (devs, k)
    bad = np.asarray([0, 99], np.int32)
    assert fnm(C.byref(p), C.byref(b), C.byref(resm), bad.ctypes.data_as(capi.c_i32p), 2) == capi.DD_ERR_NO_DEVICE
    assert "block 1" in capi.last_error()


def test_multi_device_block_of_windows_without_haplotypes(lib):
    """A block that holds only windows with reads but no haplotypes has no pair to compute; its reads' onHap must still come back 0, as the
    single-device call (whose onHap kernel covers every read) leaves them — also when there are more devices than windows with pairs."""
    h = rnd(100)
    full = lambda start: Window(start, [h, h[:50] + h[52:]], reads_from(h, 12, 40, start0=start))
    empty = lambda start: Window(start, [], reads_from(h, 9, 40, start0=start))
    ws = [full(1000), empty(3000), empty(4000), empty(5000), full(7000)]
    pb = pack(ws)
    b = pb.ctypes_batch()
    p = capi.params_cli_defaults()
    one, res1 = alloc_result(pb, fill=0x55)
    assert lib.dd_compute_likelihoods(C.byref(p), C.byref(b), C.byref(res1), 0) == 0, capi.last_error()
    assert one["onHap"][12:12 + 27].max() == 0 and one["onHap"][:12].max() == 1
    for devs in ([0, 0, 0], [0, 0, 0, 0, 0], [0] * 6):
        many, resm = alloc_result(pb, fill=0x55)
        d = np.asarray(devs, np.int32)
        assert lib.dd_compute_likelihoods_multi(C.byref(p), C.byref(b), C.byref(resm), d.ctypes.data_as(capi.c_i32p), len(devs)) == 0, capi.last_error()
        assert np.array_equal(one["onHap"][:pb.n_reads], many["onHap"][:pb.n_reads]), devs
        assert np.array_equal(one["ll"][:pb.n_pairs], many["ll"][:pb.n_pairs]), devs
    # a batch without any pair: onHap zero-filled by the single-device entry too
    pe = pack([empty(3000), empty(4000)])
    be = pe.ctypes_batch()
    none, rese = alloc_result(pe, fill=0x55)
    assert lib.dd_compute_likelihoods(C.byref(p), C.byref(be), C.byref(rese), 0) == 0, capi.last_error()
    assert none["onHap"][:pe.n_reads].max() == 0


def test_k3_scratch_build_variants(lib, monkeypatch):
    """K = 3 at D = 6: the scratch build compiled for 3 waves per SIMD (since the end of round 4 for every read length: the item counter made it the
    faster one below 90 bp too), for reads so long that LDS keeps fewer than 12 waves on the CU (> ~250 bp) its 2-waves-per-SIMD variant, and —
    DD_FORCE_GBT=0 — the LDS build that short reads ran on before: all equal the oracle."""
    p = capi.params_cli_defaults()
    hap = rnd(175)                                                    # 159..190 bp: three positions per lane of a whole wavefront
    alt = hap[:70] + hap[73:]
    for L, force, want_name in ((76, None, "dd_hmm_kernel<3, 6, true, false, 0, 1>"), (76, "0", "dd_hmm_kernel<3, 6, false, false, 0, 1>"),
                                (160, None, "dd_hmm_kernel<3, 6, true, false, 0, 1>"), (330, None, "dd_hmm_kernel<3, 6, true, false, 2, 1>")):
        if force is None:
            monkeypatch.delenv("DD_FORCE_GBT", raising=False)
        else:
            monkeypatch.setenv("DD_FORCE_GBT", force)
        pb = pack([Window(1000, [hap, alt], reads_from(hap, 14, L, junk=0.1) + reads_from(alt, 6, L, junk=0.0))])
        got = run_host_api(lib, p, pb)
        assert lib.dd_kernel_name().decode() == want_name, (L, lib.dd_kernel_name().decode())
        assert_same(got, _oracle.batch(p, pb, nthreads=8), pb)


@pytest.mark.parametrize("hs", [30, 59, 60, 61, 62, 100, 118, 120, 121, 123, 124, 125, 126])
def test_folded_end_states_equal_the_one_lane_blocks(lib, hs, monkeypatch):
    """Round 3: for K <= 2 at D = 6 the right->middle pass carries the LO / RO end states in the generic candidate code
    (hmm_kernel.hip, FOLD) when the haplotypes leave the last position idle (64 K >= Hs + 3; RO is folded too when 64 K >= Hs + 5 and it
    sits in its lane's last slot); haplotype lengths either side of both bounds, both parities of RO's slot, reads hanging over both ends (LO / RO stay runs, entering and leaving the haplotype): the
    FOLD build, the build with the one-lane blocks (DD_NO_FOLD) and the oracle agree bit for bit."""
    hap = rnd(hs)
    alt = hap[:hs // 2] + hap[hs // 2 + 1:]                       # one base deleted: the longer haplotype decides the build
    reads = reads_from(hap, 40, 36 if hs < 62 else 80, junk=0.15)
    reads += [ReadRec(rnd(12) + hap[:30], [0.999] * 42, 0.9999, 1000 - 12), ReadRec(hap[-30:] + rnd(15), [0.99] * 45, 0.999, 1000 + hs - 30),
              ReadRec(rnd(40), [0.9] * 40, 0.5, 1000 + hs + 500)]
    ws = [Window(1000, [hap, alt], reads)]
    pb = pack(ws)
    p = capi.params_cli_defaults()
    folded = run_host_api(lib, p, pb)
    name_folded = lib.dd_kernel_name().decode()
    monkeypatch.setenv("DD_NO_FOLD", "1")
    plain = run_host_api(lib, p, pb)
    name_plain = lib.dd_kernel_name().decode()
    monkeypatch.delenv("DD_NO_FOLD")
    want = _oracle.batch(p, pb, nthreads=8)
    assert_same(folded, want, pb)
    assert_same(plain, want, pb)
    longest = max(len(hap), len(alt))
    _, G, K = next(c for c in capi.HAP_CLASSES if longest <= c[0])
    fits = G == 1 and 64 * K >= longest + 3                      # (the half-wave builds keep the one-lane blocks)
    assert name_plain.endswith("false, 0, %d>" % G) and name_folded.endswith(("true, 0, %d>" if fits else "false, 0, %d>") % G), (name_folded, name_plain)
    if K == 2:
        # the two other builds that carry the fold: maxLengthDel = 10 on the LDS build of D = 11 (since the end of round 4 only under DD_FORCE_GBT=0: the
        # plan takes the scratch build there, which has no folded variant) and reads long enough for the scratch build at D = 6
        p10 = capi.params_cli_defaults(); p10.maxLengthDel = 10
        pl = pack([Window(1000, [hap, alt], reads_from(hap, 14, 170, junk=0.1) + reads_from(hap, 4, 200, junk=0.0))])
        for params, batch, tag, force, folds in ((p10, pb, "11, true", None, False), (p10, pb, "11, false", "0", fits), (p, pl, "6, true", None, fits)):
            if force is None:
                monkeypatch.delenv("DD_FORCE_GBT", raising=False)
            else:
                monkeypatch.setenv("DD_FORCE_GBT", force)
            got = run_host_api(lib, params, batch)
            name = lib.dd_kernel_name().decode()
            monkeypatch.setenv("DD_NO_FOLD", "1")
            ref = run_host_api(lib, params, batch)
            monkeypatch.delenv("DD_NO_FOLD")
            monkeypatch.delenv("DD_FORCE_GBT", raising=False)
            w2 = _oracle.batch(params, batch, nthreads=8)
            assert_same(got, w2, batch)
            assert_same(ref, w2, batch)
            assert ("<2, %s, " % tag) in name and name.endswith("true, 0, 1>" if folds else "false, 0, 1>"), name
