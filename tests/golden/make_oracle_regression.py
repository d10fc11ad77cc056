#!/usr/bin/env python3
"""Writes tests/golden/oracle_regression.json: outputs of THIS repository's CPU restatement (oracle/) on seeded adversarial
pairs, for both models.  These are NOT reference outputs (only survey_kat.json holds those); they freeze the oracle's
behaviour at the state that was pinned to the reference KATs and cross-checked against the GPU in the fuzz campaigns, so
that a later edit of oracle/dd_oracle.c that changes any result is caught by a CPU test."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
from dindel_tgi_amd import capi
from tests import _oracle
from tests.test_gpu_fuzz import make_windows

rng = np.random.default_rng(20261004)
cases = []
for i, (mld, defaults) in enumerate([(5, "cli"), (10, "struct"), (0, "cli"), (3, "cli"), (11, "struct")]):
    p = capi.params_cli_defaults() if defaults == "cli" else capi.params_struct_defaults()
    p.maxLengthDel = mld
    for w in make_windows(rng, 6, 70, 60, min_hap=max(mld, 1)):
        for h in w.haps[:2]:
            for r in w.reads[:3]:
                o, hpos = _oracle.pair(h, r.seq, r.qual, r.mapQual, r.start, w.hap_start, p, unmapped=r.unmapped)
                f, fh = _oracle.pair_fast(h, r.seq, r.qual, r.mapQual, r.start, w.hap_start, p)
                cases.append(dict(defaults=defaults, maxLengthDel=mld, hap=h, read=r.seq, qual=[float(q) for q in r.qual],
                                  mapQual=float(r.mapQual), start=int(r.start), hapStart=int(w.hap_start), unmapped=bool(r.unmapped),
                                  ll=o.ll.hex(), llOn=o.llOn.hex(), llOff=o.llOff.hex(), offHap=o.offHap, offHapHMQ=o.offHapHMQ,
                                  numIndels=o.numIndels, hpos=hpos, status=o.status,
                                  fast_status=f.status, fast_ll=f.ll.hex() if f.status == 0 else None, fast_hpos=fh if f.status == 0 else None))
json.dump(dict(note="oracle regression vectors (hex floats); not reference outputs", cases=cases),
          open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=0)
print(len(cases), "cases")
