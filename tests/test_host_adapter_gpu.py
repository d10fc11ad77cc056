"""C++ host adapter on the GPU: LikelihoodEngine::computeLikelihoods (mirror of DetInDel::computeLikelihoods,
DInDel.cpp:1707-1739) against the golden vectors and the oracle, including the reference's error strings."""
import json
import os

import numpy as np
import pytest

from dindel_tgi_amd import capi
from tests import _host, _oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))["cases"]


@pytest.mark.parametrize("case", KAT, ids=[c["name"] for c in KAT])
def test_engine_on_kat(case):
    p = capi.dd_params.from_dict(case["params"])
    res = _host.compute_window([case["hap"]], [case["read"]], [case["q"]], [case["mapQual"]], [float(case["pos"])], [0],
                               case["hapStart"], p)
    ml = res["liks"][0][0]
    for k in ("ll", "llOn", "llOff"):
        if k in case:
            assert ml[k] == pytest.approx(case[k], rel=1e-13, abs=0)
    if "hpos" in case:
        assert ml["hpos"] == case["hpos"]
    if "indels" in case:
        assert [[i[0], i[1]] for i in ml["indels"]] == case["indels"]
    if "snps" in case:
        assert ml["snps"] == case["snps"]
    for k in ("offHap", "offHapHMQ", "nBQT", "numMismatch"):
        if k in case:
            assert ml[k] == case[k]
    assert res["onHap"] == [0 if ml["offHapHMQ"] else 1]


def test_engine_window_matches_oracle_and_onhap():
    H = "ACGTTGCATGCCGATAGGCTTAACCGGTTTTTTACGATCGATGCAAGTCCGTA"
    haps = [H, H[:25] + H[27:], H[:30] + "GG" + H[30:]]
    reads = [H[10:40], H[5:25] + H[27:45], "GTCA" * 7 + "GT", H[20:30] + "GG" + H[30:50], "TTGACCA" + H[0:25]]
    quals = [0.999, 0.99, 0.999, 0.9999, 0.999]
    mapq = [0.9999, 0.999, 0.9, 0.9999, 0.99]
    pos = [1010.0, 1005.0, 1010.0, 1020.0, 993.0]
    p = capi.params_cli_defaults()
    res = _host.compute_window(haps, reads, quals, mapq, pos, [0, 0, 0, 0, 1], 1000, p)
    on = [0] * len(reads)
    for h, hap in enumerate(haps):
        for r, read in enumerate(reads):
            o, hpos = _oracle.pair(hap, read, quals[r], mapq[r], int(pos[r]), 1000, p, unmapped=(r == 4))
            ml = res["liks"][h][r]
            assert ml["ll"] == o.ll and ml["llOn"] == o.llOn and ml["llOff"] == o.llOff
            assert ml["hpos"] == hpos
            assert (ml["offHap"], ml["offHapHMQ"]) == (o.offHap, o.offHapHMQ)
            if not o.offHapHMQ:
                on[r] = 1
    assert res["onHap"] == on


def test_engine_throws_reference_strings():
    p = capi.params_cli_defaults()
    res = _host.compute_window(["ACG"], ["ACGT"], [0.999], [0.9999], [0.0], [0], 0, p)
    assert res == {"throw": "hapSize error."}
