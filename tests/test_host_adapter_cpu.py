"""C++ host adapter, CPU part: LikelihoodEngine::rebuildAlignment reconstructs the MLAlignment record
(reportVariants, ObservationModelFB.cpp:1351-1475) from hpos.  hpos comes from the oracle here."""
import json
import os

import numpy as np
import pytest

from dindel_tgi_amd import capi, synth
from tests import _host, _oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))["cases"]


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_rebuild_matches_kat_variant_strings(case):
    p = capi.dd_params.from_dict(case["params"])
    o, hpos = _oracle.pair(case["hap"], case["read"], case["q"], case["mapQual"], case["pos"], case["hapStart"], p)
    ml = _host.rebuild(case["hap"], case["read"], case["q"], _oracle.keyed_hpos(o, hpos), p)
    assert ml["hpos"] == hpos
    assert ml == _host.rebuild(case["hap"], case["read"], case["q"], hpos, p)   # the reference's bare INS code works here too
    if "indels" in case:
        assert [[i[0], i[1]] for i in ml["indels"]] == case["indels"]
    if "snps" in case:
        assert ml["snps"] == case["snps"]
    for k in ("numIndels", "numMismatch", "nBQT", "nmmBQT", "nMMLeft", "nMMRight", "firstBase", "lastBase"):
        assert ml[k] == getattr(o, k), k
    assert ml["mLogBQ"] == o.mLogBQ


def test_rebuild_matches_oracle_on_random_pairs():
    pb = synth.generate(3, H=4, R=30, L=60, hap_len=70, seed=77, mixed_quals=True, sub_rate=0.03, max_indel=4)
    p = capi.params_cli_defaults()
    a = pb.a
    n_ins = n_del = 0
    for w in range(pb.n_windows):
        for h in range(a["win_hap_off"][w], a["win_hap_off"][w + 1]):
            hap = bytes(a["hap_seq"][a["hap_seq_off"][h]:a["hap_seq_off"][h + 1]]).decode()
            for r in range(a["win_read_off"][w], a["win_read_off"][w + 1]):
                s0, s1 = a["read_seq_off"][r], a["read_seq_off"][r + 1]
                read = bytes(a["read_seq"][s0:s1]).decode()
                q = a["qual_table"][a["read_qidx"][s0:s1]]
                o, hpos = _oracle.pair(hap, read, q, a["mapq_table"][a["read_mqidx"][r]], int(a["read_start"][r]),
                                       int(a["win_hap_start"][w]), p)
                ml = _host.rebuild(hap, read, q, _oracle.keyed_hpos(o, hpos), p)
                for k in ("numIndels", "numMismatch", "nBQT", "nmmBQT", "nMMLeft", "nMMRight", "firstBase", "lastBase"):
                    assert ml[k] == getattr(o, k), (k, h, r)
                assert ml["mLogBQ"] == o.mLogBQ
                want = {}
                for i in range(o.n_indel):          # map semantics: a later variant at the same key overwrites
                    pos, ln, rp = o.indel_pos[i], o.indel_len[i], o.indel_rpos[i]
                    want[pos] = "+" + read[rp:rp + ln] if ln > 0 else "-" + hap[pos:pos - ln]
                    n_ins += ln > 0
                    n_del += ln < 0
                assert {i[0]: i[1] for i in ml["indels"]} == want
                _oracle.assert_record_variants(ml, o, hap, read, (h, r))
                assert ml["align"] == _oracle.expected_align(o, hap, read) and ml["hpos"] == hpos
    assert n_ins > 5 and n_del > 5          # the sample really exercises both kinds


def test_covered_flags_follow_isCovered():
    p = capi.params_cli_defaults()          # padCover = 2
    hap = "ACGTTGCATGCCGATAGGCTTAACCGGTTTTTTACGATCGATGCAAGTCCGTA"
    read = hap[10:40]
    o, hpos = _oracle.pair(hap, read, 0.999, 0.9999, 1010, 1000, p)
    ml = _host.rebuild(hap, read, 0.999, hpos, p, hap_indels=[(5, 12, 37), (6, 11, 37), (7, 12, 38)])
    # firstBase=10, lastBase=39: covered iff 10+2<=startRead and 39-2>=endRead (Variant.hpp:125-128)
    assert ml["hapIndelCovered"] == [[5, 1], [6, 0], [7, 0]]


def test_rebuild_matches_oracle_on_adversarial_pairs():
    """rebuildAlignment / rebuildAlignmentFaster against the oracle's variant lists on the adversarial windows of the fuzz
    tests plus 1-3-bp reads, reads that match no haplotype base and reads that are inserted as a whole (whose ml.indels key
    is the join's state and cannot be read off neighbouring hpos entries: hap A x 30, read C -> {1: "+C"}): keys, strings,
    haplotype and read coordinates, align string, counters.  0 differences allowed."""
    from dindel_tgi_amd.batch import ReadRec, Window
    from tests.test_gpu_fuzz import make_windows
    rng = np.random.default_rng(4242)
    p = capi.params_cli_defaults()
    ws = make_windows(rng, 150, 90, 70, min_hap=p.maxLengthDel)
    q = [0.9999]
    ws.append(Window(1000, ["A" * 30, "A" * 12 + "C" + "A" * 17, "ACGTACGTACGTACGTACGTAAAAA"],
                     [ReadRec("C", q, 0.9999, 1010), ReadRec("G", q, 0.99, 1000), ReadRec("CG", q * 2, 0.9999, 1005),
                      ReadRec("TTT", q * 3, 0.9999, 1020), ReadRec("GGGGGGGG", q * 8, 0.9999, 1008), ReadRec("T", q, 0.5, 5),
                      ReadRec("CCCCCCCCCCCCCCCCCCCC", q * 20, 0.9999, 1002)]))
    n = n_whole = n_fast = n_fast_ins = 0
    for w in ws:
        for hap in w.haps:
            for rd in w.reads:
                o, hpos = _oracle.pair(hap, rd.seq, rd.qual, rd.mapQual, rd.start, w.hap_start, p, unmapped=rd.unmapped)
                ml = _host.rebuild(hap, rd.seq, rd.qual, _oracle.keyed_hpos(o, hpos), p)
                _oracle.assert_record_variants(ml, o, hap, rd.seq, (hap, rd.seq))
                assert ml["hpos"] == hpos and ml["align"] == _oracle.expected_align(o, hap, rd.seq)
                for k in ("numIndels", "numMismatch", "nBQT", "nmmBQT", "nMMLeft", "nMMRight", "firstBase", "lastBase", "mLogBQ"):
                    assert ml[k] == getattr(o, k), (k, hap, rd.seq)
                n += 1
                n_whole += all(x == capi.DD_HPOS_INS for x in hpos)
                if len(rd.seq) >= 4 and len(hap) >= 4:
                    f, fh = _oracle.pair_fast(hap, rd.seq, rd.qual, rd.mapQual, rd.start, w.hap_start, p)
                    if f.status == 0:
                        mf = _host.rebuild(hap, rd.seq, rd.qual, _oracle.keyed_hpos(f, fh), p, faster=True)
                        _oracle.assert_record_variants(mf, f, hap, rd.seq, ("faster", hap, rd.seq))
                        assert mf["hpos"] == fh and (mf["firstBase"], mf["lastBase"]) == (f.firstBase, f.lastBase)
                        n_fast += 1
                        n_fast_ins += any(f.indel_len[i] > 0 for i in range(f.n_indel))
    assert n > 1000 and n_whole >= 3 and n_fast > 500 and n_fast_ins > 20


def test_whole_read_insertion_key_is_the_join_state():
    """The advisor's reproducer (round 1): hap A x 30, read C, Q40 -> indels {1: "+C"} (ObservationModelFB.cpp:1380)."""
    p = capi.params_cli_defaults()
    o, hpos = _oracle.pair("A" * 30, "C", [0.9999], 0.9999, 1010, 1000, p)
    assert hpos == [capi.DD_HPOS_INS] and (o.n_indel, o.indel_pos[0], o.indel_len[0]) == (1, 1, 1)
    ml = _host.rebuild("A" * 30, "C", [0.9999], _oracle.keyed_hpos(o, hpos), p)
    assert ml["indels"] == [[1, "+C", 1, 1, 0, 0]]
