"""GPU parity of the half-wave builds (two reads of a haplotype side by side on the 32-lane halves of a wavefront, hmm_kernel.hip G = 2):
bit-equal to the oracle and to the whole-wavefront builds (DD_NO_HALF=1), through the C ABI.  Covers every half tiling in use (K = 1, 3, 5
positions per lane), the three D builds, both back-pointer placements, windows with an odd number of reads (the second pair of the last
wavefront is missing), reads of very different lengths and bMid in one window (padded trip counts), reads of another length class,
haplotypes shorter than maxLengthDel, and more reads than one ordering chunk holds."""
import numpy as np
import pytest

from dindel_tgi_amd import capi
from dindel_tgi_amd.batch import ReadRec, Window, pack, phred_to_prob
from tests import _oracle
from tests.test_gpu_parity import assert_same, run_host_api
from tests.test_gpu_edge_cases import reads_from, rnd

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(4242)
HALF_K = {20: 1, 30: 1, 63: 3, 80: 3, 94: 3, 127: 5, 140: 5, 158: 5}


def windows_for(hs, n_reads, lens=(36, 100)):
    hap = rnd(hs)
    cut = max(1, hs // 2)
    var = hap[:cut] + hap[cut + min(2, hs - cut - 1):]
    ins = hap[:cut] + "GA" + hap[cut:] if hs + 2 <= max(k for k in HALF_K if HALF_K[k] == HALF_K[hs]) else var
    reads = []
    for i in range(n_reads):
        src = (hap, var, ins)[i % 3]
        L = int(RNG.integers(max(8, lens[0]), lens[1] + 1))
        reads += reads_from(src, 1, L, q=float(phred_to_prob([int(RNG.integers(5, 41))])[0]))
    return [Window(1000, [hap, var, ins], reads), Window(1000, [var], reads[:1]), Window(1000, [hap, ins], reads[:5])]


@pytest.mark.parametrize("force", [None, "0", "1"])
@pytest.mark.parametrize("hs", sorted(HALF_K))
def test_half_wave_builds_equal_oracle_and_whole_wave_builds(lib, monkeypatch, hs, force):
    if force is not None:
        monkeypatch.setenv("DD_FORCE_GBT", force)
    ws = windows_for(hs, 23)
    pb = pack(ws)
    for mld in (5, 10, 11):
        p = capi.params_cli_defaults()
        p.maxLengthDel = mld
        got = run_host_api(lib, p, pb)
        log = capi.launch_log()
        assert any(r["pairs_per_wave"] == 2 and r["K"] == HALF_K[hs] for r in log), log
        want = _oracle.batch(p, pb, nthreads=8)
        assert_same(got, want, pb)
        monkeypatch.setenv("DD_NO_HALF", "1")
        whole = run_host_api(lib, p, pb)
        assert all(r["pairs_per_wave"] == 1 for r in capi.launch_log())
        monkeypatch.delenv("DD_NO_HALF")
        assert_same(got, whole, pb)


def test_half_wave_more_reads_than_one_chunk_and_two_read_classes(lib):
    """700 reads of 30..170 bp on a 140-bp haplotype: three ordering chunks per workgroup, two read-length classes (<= 160 bp / longer) whose
    launches each skip the other's reads."""
    hap = rnd(140)
    var = hap[:70] + hap[73:]
    reads = []
    for i in range(700):
        reads += reads_from((hap, var)[i & 1], 1, int(RNG.integers(30, 171)))
    pb = pack([Window(1000, [hap, var], reads)])
    p = capi.params_cli_defaults()
    got = run_host_api(lib, p, pb)
    log = capi.launch_log()
    assert len(log) == 2 and all(r["pairs_per_wave"] == 2 and r["K"] == 5 for r in log), log
    assert_same(got, _oracle.batch(p, pb, nthreads=8), pb)


def test_half_wave_hapsize_error_and_empty_windows(lib):
    """maxLengthDel > haplotype length: every pair of that haplotype gets DD_PAIR_HAPSIZE from the ordering pass; a window without reads and
    one without haplotypes ride along."""
    short, hap = rnd(4), rnd(80)
    ws = [Window(1000, [short, hap], reads_from(hap, 7, 40)), Window(1000, [hap], []), Window(1000, [], reads_from(hap, 3, 40)),
          Window(1000, [hap, short], reads_from(hap, 1, 50))]
    p = capi.params_cli_defaults()
    pb = pack(ws)
    got = run_host_api(lib, p, pb)
    assert_same(got, _oracle.batch(p, pb, nthreads=8), pb)
    st = got["status"][:pb.n_pairs]
    assert st[:7].tolist() == [capi.DD_PAIR_HAPSIZE] * 7 and (st[7:14] == 0).all() and st[-1] == capi.DD_PAIR_HAPSIZE


def test_half_wave_fuzz_with_variants_and_mates(lib):
    from tests.test_gpu_fuzz import make_windows
    for seed, max_hap, mld in ((1, 94, 5), (2, 158, 5), (3, 158, 10), (4, 30, 3), (5, 150, 11)):
        rng = np.random.default_rng(7000 + seed)
        ws = make_windows(rng, 60, max_hap, 120, min_hap=max(mld, 1), with_vars=True)
        p = capi.params_cli_defaults()
        p.maxLengthDel = mld
        pb = pack(ws)
        got = run_host_api(lib, p, pb)
        assert any(r["pairs_per_wave"] == 2 for r in capi.launch_log())
        assert_same(got, _oracle.batch(p, pb, nthreads=8), pb)
