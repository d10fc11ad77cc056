"""Every lane tiling x D build, with more items than the persistent grid holds (a workgroup takes several items one after the other): GPU vs oracle,
one line per launch mode (what tests/test_gpu_persistent_rounds.py asserts, as a sweep with the launch geometry printed).
    tools/multi_round_sweep.py MAXLENGTHDEL [CLASS,CLASS,... [READ_LEN]]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dindel_tgi_amd import capi, synth
from tests import _oracle
from tests.test_gpu_parity import run_host_api
mld = int(sys.argv[1])
Ks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(16))
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = capi.load()
bounds = [30, 62, 94, 126, 158, 190, 222, 254, 318, 382, 446, 510, 574, 638, 702, 766]
for c in Ks:
    if mld > 11 and bounds[c] > 574: continue
    p = capi.params_cli_defaults(); p.maxLengthDel = mld
    hap_len = bounds[c] - 4
    if hap_len <= mld: continue
    pb = synth.generate(14, H=8, R=48, L=L, hap_len=hap_len, seed=77 + c, max_indel=3, sub_rate=0.01, mixed_quals=True)
    ref = _oracle.batch(p, pb, nthreads=16)
    for env in ({}, {"DD_DYNAMIC": "1", "DD_FORCE_GBT": "0"}, {"DD_FORCE_GBT": "1"}, {"DD_NO_HALF": "1"}, {"DD_NO_HALF": "1", "DD_DYNAMIC": "1", "DD_FORCE_GBT": "0"}):
        for k in ("DD_DYNAMIC", "DD_FORCE_GBT", "DD_NO_HALF"): os.environ.pop(k, None)
        os.environ.update(env)
        got = run_host_api(lib, p, pb)
        log = capi.launch_log()
        bad = {}
        for k in ("ll", "status", "firstBase", "lastBase", "numIndels", "offHap"):
            a, b = got[k][:pb.n_pairs], ref[k][:pb.n_pairs]
            bad[k] = int((a.view(np.uint8).reshape(len(a), -1) != b.view(np.uint8).reshape(len(b), -1)).any(axis=1).sum())
        rounds = [round(l["n_haps"] * l["split"] / max(1, l["grid"]), 2) for l in log]
        print("mld", mld, "class", bounds[c], "L", L, "env", env, "launch", [(l["K"], l["pairs_per_wave"], l["D"], "hbm" if l["gbt"] else "lds", l["fold"], l["waves"], l["grid"], l["split"], l["n_haps"], l["dynamic"]) for l in log],
              "rounds", rounds, "BAD" if any(bad.values()) else "ok", bad if any(bad.values()) else "", flush=True)
