#!/usr/bin/env python3
"""Writes tests/golden/survey_kat.json: the known-answer vectors recorded in SURVEY.md §8(c).

Provenance: inputs are the reference's own dead self-test vectors (DInDel.cpp:4003-4004, :4040-4041;
"S1", "S2") and the survey's K1..K8 cases; expected values are the ones the survey captured from the
compiled reference (ObservationModelFBMaxErr::calcLikelihood) and printed in SURVEY.md §8(c).  This
script only transcribes that table — it runs neither the reference nor the oracle.
"""
import json, os

H = "ACGTTGCATGCCGATAGGCTTAACCGGTTTTTTACGATCGATGCAAGTCCGTA"
S1_HAP = "ATCGATTCGTGATATATATATTCAATGTAGTCGCTAG"
S1_READ = "ATCGATTCGTGATAATATTCAATGTAGTCGCTAG"
S2_HAP = ("AAAATCACCAACACTTCATAATCTATTTTTTCCCCTGAGGAACTTCCTAAAATGAATAAAAAAAAACCCCAGCCACATCTGCATTTGCAAACAGGAAACTCTGCAAGCC"
          "ATACTAAGACCAAAGCTTAGTT")
S2_READ = "CAAACAGGAAACTCTGCAAGCCATACTAAGACCAAAGCTTAGTTA"
STRUCT = dict(pError=1e-4, pMut=1e-4, pFirstgLO=0.01, mapQualThreshold=100.0, checkBaseQualThreshold=0.95,
              maxLengthDel=10, padCover=5, bMid=-1, forceReadOnHaplotype=0, mapUnmappedReads=0, maxMismatch=1,
              capMapQualFast=40.0)
CLI = dict(pError=5e-4, pMut=1e-5, pFirstgLO=0.01, mapQualThreshold=100.0, checkBaseQualThreshold=0.95,
           maxLengthDel=5, padCover=2, bMid=-1, forceReadOnHaplotype=0, mapUnmappedReads=0, maxMismatch=2,
           capMapQualFast=45.0)
r = H[10:40]
flip = {"A": "C", "C": "A"}
rng = lambda a, b: list(range(a, b + 1))
cases = [
    dict(name="S1_struct", hap=S1_HAP, read=S1_READ, pos=0, hapStart=0, q=0.99, mapQual=1 - 1e-16, params=STRUCT,
         ll=-12.208726495298166, llOff=-23.458611539743039, hpos=rng(0, 13) + rng(17, 36),
         indels=[[14, "-TAT"]], offHap=0, ll_fast=-9.472245272269241,       # ll_fast: ObservationModelS (--faster)
         ll_fbmax=-11.934075176806624),                                    # ll_fbmax: sibling model ObservationModelFBMax
    dict(name="S1_cli", hap=S1_HAP, read=S1_READ, pos=0, hapStart=0, q=0.99, mapQual=1 - 1e-16, params=CLI,
         ll=-12.206837301998762, llOff=-23.469926307808155, ll_fast=-8.3736536261351855, ll_fbmax=-10.342040525271681),
    dict(name="S2_struct", hap=S2_HAP, read=S2_READ, pos=0, hapStart=0, q=0.99, mapQual=1 - 1e-16, params=STRUCT,
         ll=-0.34501268318605621, hpos=rng(87, 130) + [-4], ll_fast=-0.34673884362349849, ll_fbmax=-0.35103905873783248),
    dict(name="K1_exact", hap=H, read=r, pos=1010, hapStart=1000, q=0.999, mapQual=1 - 1e-4, params=CLI,
         ll=-0.026245983382989002, llOff=-9.3988325473015166, hpos=rng(10, 39), nBQT=30),
    dict(name="K2_mismatch", hap=H, read=r[:12] + flip[r[12]] + r[13:], pos=1010, hapStart=1000, q=0.999,
         mapQual=1 - 1e-4, params=CLI, ll=-7.2109849934460657, snps=[[22, "A=>C"]], numMismatch=1),
    dict(name="K3_ins2", hap=H, read=r[:15] + "GG" + r[15:], pos=1010, hapStart=1000, q=0.999, mapQual=1 - 1e-4,
         params=CLI, ll=-9.4113986923231998, llOn=-9.9936481360025802, offHap=1, offHapHMQ=0,
         hpos=rng(10, 24) + [-1, -1] + rng(25, 39), indels=[[25, "+GG"]]),
    dict(name="K4_hp_del1", hap=H, read=H[20:27] + H[28:48], pos=1020, hapStart=1000, q=0.999, mapQual=1 - 1e-4,
         params=CLI, ll=-7.8463691437164433, hpos=rng(20, 30) + rng(32, 47), indels=[[31, "-T"]]),
    dict(name="K5_left_overhang", hap=H, read="TTGACCA" + H[0:25], pos=993, hapStart=1000, q=0.999, mapQual=1 - 1e-4,
         params=CLI, ll=-0.029419857367153013, hpos=[-3] * 7 + rng(0, 24)),
    dict(name="K6_right_overhang", hap=H, read=H[30:52] + "GGATCCA", pos=1030, hapStart=1000, q=0.999,
         mapQual=1 - 1e-4, params=CLI, ll=-7.2098772026701807, hpos=rng(30, 52) + [-4] * 6, snp_positions=[52]),
    dict(name="K7_junk_mq0.9", hap=H, read="GTCA" * 7 + "GT", pos=1010, hapStart=1000, q=0.999, mapQual=0.9,
         params=CLI, ll=-2.4910772683192692, llOn=-15.594698622147538, offHap=1, offHapHMQ=0),
    dict(name="K8_q20", hap=H, read=r, pos=1010, hapStart=1000, q=0.99, mapQual=1 - 1e-4, params=CLI,
         ll=-0.22958502382338133),
]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "survey_kat.json")
with open(out, "w") as f:
    json.dump(dict(source="SURVEY.md §8(c): values captured by the survey from the compiled reference "
                          "(ObservationModelFBMaxErr::calcLikelihood); transcribed, not regenerated",
                   cases=cases), f, indent=1)
print("wrote", out, len(cases), "cases")
