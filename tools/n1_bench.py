#!/usr/bin/env python3
"""N1 on the device at configs[1] scale: read sums (dd_pair_sums_device) + MAP pairs / qual (dd_map_pairs_device) over the ll
array the likelihood kernel left in HBM.  Prints kernel times (HIP events on the launch stream) and windows/s."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
lib = capi.load()
pb = synth.generate(n, H=8, R=200, L=100, hap_len=120, seed=0x9E3779B9)
p = capi.params_cli_defaults()
dev = DeviceBatch(pb, p, "cuda:0")
dev.launch(); torch.cuda.synchronize()
hb = pb.ctypes_batch()
hh = np.zeros(n + 1, np.int64)
lib.dd_pair_sum_offsets(C.byref(hb), hh.ctypes.data_as(capi.c_i64p))
ns = int(hh[-1])
hh_d = torch.from_numpy(hh).cuda()
sums = torch.zeros(ns, dtype=torch.float64, device="cuda")
post = torch.zeros(ns, dtype=torch.float64, device="cuda")
prior = torch.full((ns,), float(np.log(1e-3)), dtype=torch.float64, device="cuda")
filt = torch.zeros(pb.n_haps, dtype=torch.uint8, device="cuda")
ncand = torch.from_numpy((np.arange(pb.n_haps) % 8 != 0).astype(np.int32)).cuda()      # haplotype 0 of a window = reference
pairs = torch.zeros(4 * n, dtype=torch.int32, device="cuda")
vals = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()


def run():
    rc = lib.dd_pair_sums_device(C.byref(dev.db), C.c_void_p(hh_d.data_ptr()), ns, C.c_void_p(dev.out["ll"].data_ptr()),
                                 C.c_void_p(sums.data_ptr()), C.c_void_p(st.cuda_stream))
    assert rc == 0, capi.last_error()
    rc = lib.dd_map_pairs_device(C.byref(dev.db), hh_d.data_ptr(), sums.data_ptr(), prior.data_ptr(), filt.data_ptr(), ncand.data_ptr(),
                                 post.data_ptr(), pairs.data_ptr(), vals.data_ptr(), st.cuda_stream)
    assert rc == 0, capi.last_error()


run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
terms = pb.n_reads * 36            # addLogs evaluations per launch: 36 unordered haplotype pairs x reads of the window
print(json.dumps(dict(windows=n, slots=ns, ms_per_launch=ms, windows_per_s=n / ms * 1e3, addlogs_per_s=terms / ms * 1e3,
                      bytes_read_GBps=(pb.n_pairs * 8 * 8.0) / ms / 1e6, qual0=float(vals[2].item()), pair0=pairs[:4].tolist())))
