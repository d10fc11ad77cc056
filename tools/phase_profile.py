#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of dd_hmm_kernel from s_memtime stamps (-DDD_STAMPS build).

Builds a SEPARATE library (gpurun_out/libdindel_hmm_stamps.so) — the product library never contains
stamps — runs one batch and prints the share of wave-cycles per phase.  Read the shares, not the run time
(cdna_hip_programming.md §7 'In-kernel stamps')."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dindel_tgi_amd import capi, synth

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdindel_hmm_stamps.so")
src = os.path.join(ROOT, "dindel_tgi_amd", "csrc")
only = os.environ.get("DD_ONLY", "-DDD_ONLY_K=2 -DDD_ONLY_D=6")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-DDD_STAMPS",
       *only.split(), "-shared", "-o", so, os.path.join(src, "hmm_kernel.hip"), os.path.join(src, "genotype_kernel.hip"), os.path.join(src, "faster_kernel.hip"), os.path.join(src, "capi.cpp")]
subprocess.check_call(cmd)
capi.LIB_PATH = so
lib = capi.load()
from dindel_tgi_amd.device import DeviceBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
# shape: PP_SHAPE="H R L hap maxLengthDel" (default: configs[1]); the stamped build must hold the (K, D) the plan picks (DD_ONLY)
H, R, L, hap, mld = [int(x) for x in os.environ.get("PP_SHAPE", "8 200 100 120 5").split()]
if os.environ.get("PP_RAGGED"):          # real-shaped windows whose haplotypes all sit on the K = 2 tiling (116..126 bp)
    pb = synth.generate_ragged(n, max_extra=0, max_indel=5)
else:
    pb = synth.generate(n, H=H, R=R, L=L, hap_len=hap, seed=3)
params = capi.params_cli_defaults()
params.maxLengthDel = mld
dev = DeviceBatch(pb, params, "cuda:0")
dbg = torch.zeros(16, dtype=torch.int64, device="cuda:0")
lib.dd_debug_set_stamp_buffer(C.c_void_p(dbg.data_ptr()))
dev.launch(); torch.cuda.synchronize()
dbg.zero_()
dev.launch(); torch.cuda.synchronize()
v = dbg.cpu().numpy().astype(float)
names = ["(unused)", "bMid+stage read", "Dec passes", "Inc passes", "join", "traceback", "hpos+counters+mLogBQ", "coverage+outputs"]
tot = v[:8].sum()
for i, nme in enumerate(names):
    print("%-18s %14.0f cycles  %5.1f %%   %9.0f cycles/pair" % (nme, v[i], 100 * v[i] / tot, v[i] / pb.n_pairs))
print("total %.0f cycles/pair (wave time, incl. stamp overhead)" % (tot / pb.n_pairs))
print("kernel", capi.last_launch(), "shape", (H, R, L, hap, mld), "launches", [(r["K"], r["pairs_per_wave"], r["gbt"], r["waves"], r["split"], r["n_haps"], r["min_read"], r["max_read"]) for r in capi.launch_log()])
