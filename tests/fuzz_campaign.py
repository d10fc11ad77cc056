#!/usr/bin/env python3
"""Long randomised GPU-vs-oracle parity campaign (both models), beyond what the test-suite budget allows.
  python tests/fuzz_campaign.py [--seconds 300] [--seed0 0]
Every round draws a shape class, parameters and adversarial windows (tests/test_gpu_fuzz.make_windows plus
synth.generate mixes), runs the C ABI and the oracle (16 threads) and requires bit-equality.  Prints one line per
round; exits non-zero at the first mismatch after dumping the seed."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.batch import pack
from tests import _oracle
from tests.test_gpu_fuzz import make_windows
from tests.test_gpu_parity import assert_same, run_host_api
from tests.test_gpu_faster import assert_same_faster, run_faster

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--seed0", type=int, default=0)
ap.add_argument("--long-deletions", action="store_true", help="every round draws maxLengthDel from 12..31 (the D = 32 build) instead of one in ten")
args = ap.parse_args()
lib = capi.load()
t_end = time.time() + args.seconds
rnd = 0
pairs = 0
while time.time() < t_end:
    seed = args.seed0 + rnd
    rng = np.random.default_rng(50000 + seed)
    kind = int(rng.integers(0, 3))
    p = capi.params_cli_defaults() if rng.random() < 0.5 else capi.params_struct_defaults()
    p.maxLengthDel = int(rng.integers(0, 12)) if rng.random() < 0.9 else int(rng.integers(12, 32))   # 12..31: the D = 32 build (haplotypes up to 574 bp)
    if args.long_deletions:
        p.maxLengthDel = int(rng.integers(12, 32))
    p.padCover = int(rng.integers(0, 6))
    p.maxMismatch = int(rng.integers(0, 4))
    p.pError = float(rng.choice([5e-4, 1e-4, 1e-2, 0.2]))
    p.pMut = float(rng.choice([1e-5, 1e-4, 1e-2]))
    p.mapQualThreshold = float(rng.choice([100.0, 40.0, 10.0]))
    p.capMapQualFast = float(rng.choice([45.0, 40.0, 5.0, 300.0]))
    if rng.random() < 0.15:
        p.bMid = int(rng.integers(0, 30))
    if kind == 0:
        max_hap = int(rng.choice([23, 40, 62, 87, 126, 151, 190, 215, 254, 400, 755]))   # make_windows adds up to 7 inserted bases; every lane tiling incl. the half-wave ones
        if p.maxLengthDel > 11 and max_hap > 560:
            max_hap = 560
        ws = make_windows(rng, int(rng.integers(5, 60)), max_hap, int(rng.choice([30, 100, 160, 300, 700])), min_hap=1,
                          with_vars=bool(rng.random() < 0.6))
        libs = None
        if rng.random() < 0.5:                                          # insert-size prior inputs (mapUnmappedReads)
            from tests.test_insert_prior import library
            libs = [library(rng, int(rng.integers(1, 900)), int(rng.integers(1, 400))) for _ in range(int(rng.integers(1, 4)))]
            p.mapUnmappedReads = int(rng.random() < 0.85)
            for w in ws:
                for r in w.reads:
                    r.paired = bool(rng.random() < 0.8); r.mate_unmapped = bool(rng.random() < 0.1)
                    r.mate_reverse = bool(rng.random() < 0.5); r.mate_same_tid = bool(rng.random() < 0.9)
                    mp = int(rng.choice([r.start + int(rng.integers(-500, 500)), -1, 0, 2 ** 31 - 5]))
                    r.mate_pos = (mp + 2 ** 31) % 2 ** 32 - 2 ** 31          # int32, as bam->core.mpos
                    r.mate_len = int(rng.choice([-1, 0, 36, 100]))
                    r.lib = int(rng.integers(0, len(libs)))
        pb = pack(ws, libraries=libs)
    elif kind == 1:
        pb = synth.generate(int(rng.integers(1, 8)), H=int(rng.integers(1, 17)), R=int(rng.integers(1, 300)),
                            L=int(rng.integers(8, 260)), hap_len=int(rng.integers(24, 400)), seed=seed,
                            max_indel=int(rng.integers(1, 9)), sub_rate=float(rng.choice([1e-3, 0.02, 0.2])),
                            vary_read_len=bool(rng.random() < 0.5), mixed_quals=bool(rng.random() < 0.7))
    else:
        pb = synth.generate(int(rng.integers(20, 200)), H=int(rng.integers(2, 9)), R=int(rng.integers(10, 60)),
                            L=int(rng.choice([36, 76, 100])), hap_len=int(rng.integers(40, 170)), seed=seed,
                            mixed_quals=True, sub_rate=float(rng.choice([1e-3, 0.01])))
    try:
        got = run_host_api(lib, p, pb)
        assert_same(got, _oracle.batch(p, pb, nthreads=16), pb)
        gotf = run_faster(lib, p, pb)
        assert_same_faster(gotf, _oracle.batch(p, pb, nthreads=16, faster=True), pb)
        if rnd % 3 == 1:            # the same calls with the outputs in page-locked memory (written by the kernels in place)
            import ctypes as C
            from dindel_tgi_amd.batch import alloc_result_pinned, result_lengths
            n = result_lengths(pb)
            hb = pb.ctypes_batch()
            for call, ref in ((lib.dd_compute_likelihoods, got), (lib.dd_compute_likelihoods_faster, gotf)):
                arrs, res, release = alloc_result_pinned(pb)
                try:
                    assert call(C.byref(p), C.byref(hb), C.byref(res), 0) == 0, capi.last_error()
                    assert lib.dd_last_direct_outputs() >= 10, "outputs were not written in place"
                    okp = ref["status"][:pb.n_pairs] == 0
                    assert np.array_equal(arrs["status"][:pb.n_pairs], ref["status"][:pb.n_pairs]), "pinned status"
                    assert np.array_equal(arrs["onHap"][:pb.n_reads], ref["onHap"][:pb.n_reads]), "pinned onHap"
                    for k in ("ll", "llOn", "llOff", "offHap", "offHapHMQ", "firstBase", "lastBase", "numIndels", "numMismatch", "nBQT", "nmmBQT", "nMMLeft", "nMMRight"):
                        assert np.array_equal(arrs[k][:pb.n_pairs][okp].view(np.uint8), ref[k][:pb.n_pairs][okp].view(np.uint8)), ("pinned", k)
                    if okp.all():
                        for k in ("hpos", "var_covered", "var_fcov", "mLogBQ"):
                            assert np.array_equal(arrs[k][:n[k]].view(np.uint8), ref[k][:n[k]].view(np.uint8)), ("pinned", k)
                finally:
                    release()
        if rnd % 4 == 0:            # device-pointer entry points (one batch-wide plan) against the host path (length classes)
            from dindel_tgi_amd.device import DeviceBatch
            import torch
            dev = DeviceBatch(pb, p, "cuda:0")
            ok = got["status"][:pb.n_pairs] == 0
            for launch, ref in ((dev.launch, got), (dev.launch_faster, gotf)):
                launch()
                torch.cuda.synchronize()
                res = dev.results()
                okm = ref["status"][:pb.n_pairs] == 0
                for k in ("ll", "status", "firstBase", "lastBase", "numIndels", "offHap"):
                    assert np.array_equal(res[k][:pb.n_pairs][okm], ref[k][:pb.n_pairs][okm]), ("device path", k)
                assert np.array_equal(res["onHap"][:pb.n_reads], ref["onHap"][:pb.n_reads]), "device path onHap"
            del dev
    except AssertionError as e:
        print("MISMATCH at seed", seed, "kind", kind, "params", {f: getattr(p, f) for f, _ in p._fields_}, flush=True)
        print(str(e)[:2000], flush=True)
        sys.exit(1)
    pairs += pb.n_pairs
    print("round %d seed %d kind %d pairs %d (total %d) max_hap %d max_read %d mld %d ok" %
          (rnd, seed, kind, pb.n_pairs, pairs, pb.max_hap_len, pb.max_read_len, p.maxLengthDel), flush=True)
    rnd += 1
print("campaign ok: %d rounds, %d pairs per model" % (rnd, pairs))
