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
    ml = _host.rebuild(case["hap"], case["read"], case["q"], hpos, p)
    assert ml["hpos"] == hpos
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
                ml = _host.rebuild(hap, read, q, hpos, p)
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
                assert [s[0] for s in ml["snps"]] == sorted({o.snp_pos[i] for i in range(o.n_snp)})
    assert n_ins > 5 and n_del > 5          # the sample really exercises both kinds


def test_covered_flags_follow_isCovered():
    p = capi.params_cli_defaults()          # padCover = 2
    hap = "ACGTTGCATGCCGATAGGCTTAACCGGTTTTTTACGATCGATGCAAGTCCGTA"
    read = hap[10:40]
    o, hpos = _oracle.pair(hap, read, 0.999, 0.9999, 1010, 1000, p)
    ml = _host.rebuild(hap, read, 0.999, hpos, p, hap_indels=[(5, 12, 37), (6, 11, 37), (7, 12, 38)])
    # firstBase=10, lastBase=39: covered iff 10+2<=startRead and 39-2>=endRead (Variant.hpp:125-128)
    assert ml["hapIndelCovered"] == [[5, 1], [6, 0], [7, 0]]
