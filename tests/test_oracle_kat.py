"""Pins the CPU oracle to the known-answer vectors of SURVEY.md §8(c) (tests/golden/survey_kat.json).

The expected values were captured by the survey from the compiled reference
(ObservationModelFBMaxErr::calcLikelihood, ObservationModelFB.cpp:1068-1073).  The restatement uses the
same operations in the same order, so ll is required to agree to 1e-13 relative (in practice: every
printed digit), and hpos / flags exactly.
"""
import json
import os

import pytest

from dindel_tgi_amd import capi
from tests import _oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))["cases"]


def variant_strings(case, o, hpos):
    """Rebuild reportVariants' indel / snp strings (ObservationModelFB.cpp:1392-1393, 1411-1414, 1449-1450)."""
    hap, read = case["hap"], case["read"]
    indels = []
    for i in range(o.n_indel):
        pos, ln, rpos = o.indel_pos[i], o.indel_len[i], o.indel_rpos[i]
        indels.append([pos, "+" + read[rpos:rpos + ln]] if ln > 0 else [pos, "-" + hap[pos:pos - ln]])
    snps = [[o.snp_pos[i], hap[o.snp_pos[i]] + "=>" + read[o.snp_rpos[i]]] for i in range(o.n_snp)]
    return indels, snps


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_kat(case):
    p = capi.dd_params.from_dict(case["params"])
    o, hpos = _oracle.pair(case["hap"], case["read"], case["q"], case["mapQual"], case["pos"], case["hapStart"], p)
    assert o.status == 0
    for k in ("ll", "llOn", "llOff"):
        if k in case:
            assert getattr(o, k) == pytest.approx(case[k], rel=1e-13, abs=0), k
    for k in ("offHap", "offHapHMQ", "nBQT", "numMismatch"):
        if k in case:
            assert getattr(o, k) == case[k], k
    if "hpos" in case:
        assert hpos == case["hpos"]
    indels, snps = variant_strings(case, o, hpos)
    if "indels" in case:
        assert indels == case["indels"]
        assert o.numIndels == len(case["indels"])
    if "snps" in case:
        assert snps == case["snps"]
    if "snp_positions" in case:
        assert [s[0] for s in snps] == case["snp_positions"]


def test_kat_phred_table_equals_literal_quality():
    """q=0.999 / 0.99 literals of the KATs are what Read.hpp:143-148 produces from Phred 30 / 20."""
    import numpy as np
    from dindel_tgi_amd.batch import phred_to_prob
    assert phred_to_prob([30])[0] == 0.999
    assert phred_to_prob([20])[0] == 0.99
    assert phred_to_prob([40])[0] == 1 - 1e-4


def test_hapsize_error_status():
    p = capi.params_cli_defaults()
    o, _ = _oracle.pair("ACG", "ACGT", 0.999, 0.9999, 0, 0, p)     # maxLengthDel=5 > hapSize=3
    assert o.status == capi.DD_PAIR_HAPSIZE


def test_faster_model_kat():
    """ObservationModelS ("--faster", Faster.cpp) values of SURVEY §8(c): S1 under struct and CLI parameters, S2."""
    import ctypes as C
    import numpy as np
    lib = _oracle.load()
    n = 0
    for case in KAT:
        if "ll_fast" not in case:
            continue
        p = capi.dd_params.from_dict(case["params"])
        o, hpos = _oracle.pair_fast(case["hap"], case["read"], case["q"], case["mapQual"], case["pos"], case["hapStart"], p)
        assert o.ll == pytest.approx(case["ll_fast"], rel=1e-13, abs=0)
        assert o.offHap == 0 and o.offHapHMQ == 0          # always false in this model (Faster.cpp:491, :529)
        n += 1
    assert n == 3


def test_oracle_regression_vectors():
    """tests/golden/oracle_regression.json (made by make_oracle_regression.py): the restatement's own outputs, bit for bit."""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_regression.json")))
    for c in d["cases"]:
        p = capi.params_cli_defaults() if c["defaults"] == "cli" else capi.params_struct_defaults()
        p.maxLengthDel = c["maxLengthDel"]
        o, hpos = _oracle.pair(c["hap"], c["read"], c["qual"], c["mapQual"], c["start"], c["hapStart"], p, unmapped=c["unmapped"])
        assert (o.status, o.offHap, o.offHapHMQ, o.numIndels) == (c["status"], c["offHap"], c["offHapHMQ"], c["numIndels"])
        assert (o.ll.hex(), o.llOn.hex(), o.llOff.hex()) == (c["ll"], c["llOn"], c["llOff"])
        assert hpos == c["hpos"]
        f, fh = _oracle.pair_fast(c["hap"], c["read"], c["qual"], c["mapQual"], c["start"], c["hapStart"], p)
        assert f.status == c["fast_status"]
        if f.status == 0:
            assert f.ll.hex() == c["fast_ll"] and fh == c["fast_hpos"]


@pytest.mark.parametrize("case", [c for c in KAT if "ll_fbmax" in c], ids=lambda c: c["name"])
def test_sibling_model_fbmax_kat(case):
    """SURVEY §8(c) also records ObservationModelFBMax's log-likelihood for S1 / S2.  That model is not on the production
    path; restating its two message-passing functions lets three more reference numbers pin everything both models share
    (Init / bMid, emissions, bMid priors, the join, updateMax)."""
    p = capi.dd_params.from_dict(case["params"])
    o, _ = _oracle.pair_fbmax(case["hap"], case["read"], case["q"], case["mapQual"], case["pos"], case["hapStart"], p)
    assert o.status == 0
    assert o.ll == pytest.approx(case["ll_fbmax"], rel=1e-15, abs=0)
