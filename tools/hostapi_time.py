"""dd_compute_likelihoods (host pointers, pageable) on 10,000 windows x 8 haplotypes x 200 reads of 100 bp: windows/s, best of three warm calls.
    tools/hostapi_time.py [HAP_LEN]        (DD_DYNAMIC=0/1 for the A/B in profiles/r04/item_counter_uniform_ab.txt)"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.batch import alloc_result
lib = capi.load()
pb = synth.tile(synth.generate(100, H=8, R=200, L=100, hap_len=int(sys.argv[1]) if len(sys.argv) > 1 else 120, seed=5), 100)
p = capi.params_cli_defaults()
arrs, res = alloc_result(pb); hb = pb.ctypes_batch()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); rc = lib.dd_compute_likelihoods(C.byref(p), C.byref(hb), C.byref(res), 0); ts.append(time.perf_counter() - t0); assert rc == 0
print({k: v for k, v in os.environ.items() if k.startswith("DD_DYN")}, sys.argv[1:], "windows/s %.0f (best of last 3: %.4f s)" % (pb.n_windows / min(ts[1:]), min(ts[1:])), flush=True)
