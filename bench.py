#!/usr/bin/env python3
"""bench.py — read x haplotype HMM likelihood throughput on MI355X (BASELINE.json metric).

One *step* = one pass of the hot path (dd_launch_device: every (haplotype, read) pair of the batch ->
log-likelihoods, flags, hpos, QC counters) over one batch of synthetic windows that is already resident
in HBM.  Workload at N=1 is BASELINE.json configs[1]: 10,000 windows x 8 haplotypes x 200 reads
(100 bp, Q30), CLI-default model parameters.  With N>1 every rank owns its own contiguous block of
10,000 windows (weak scaling), and each step ends with the RCCL gather of the per-pair log-likelihoods
and off-haplotype flags to rank 0 that north_star names.  `--total-windows T` switches to strong scaling
(BASELINE.json configs[3]: one job of T windows, rank r takes the r-th contiguous block, in sub-batches
that fit the int32 read-base offsets of a batch).

`value` is the kernel path on HBM-resident inputs.  At N=1 the line also carries what SURVEY §8(d) defines
windows/s on: `windows_per_s_incl_copies` (dd_compute_likelihoods with host pointers: H2D + kernels + D2H)
and `windows_per_s_end_to_end` (dindel::LikelihoodEngine::computeLikelihoodsBatch on the reference's own
C++ objects: pack + H2D + kernels + D2H + per-window status scan, records delivered as lazy views), with
the split and the eager-record rate under `end_to_end`, and `window_loop`: the batched window loop of the diploid analysis (BAM ->
read selection -> kernels -> diploidGLF -> .glf.txt, host/dindel_gpu) on a synthetic 100,000-window sample the leg writes itself
(`windows_per_s` by the driver's own clock from set-up to the last line written, `steady_windows_per_s` once the pipeline is full).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows 10000] [--total-windows T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Launched plainly with `--gpus N` (N > 1, no WORLD_SIZE in the environment) the script starts the second form itself: the
parent touches no GPU, runs `python -m torch.distributed.run --nproc-per-node N` on this file as a fresh child process and exits
with the child's code (`self_launch`).  Inside a rank, WORLD_SIZE must equal --gpus: anything else is an error, never a silent
single-GPU run.

At N > 1 the line carries three figures: `value` — every rank runs --windows windows per step and gathers ll + flags to rank 0
over RCCL (weak; what the driver's scaling table is computed from: the per-GPU work is the N = 1 workload); `configs3` — ONE pass
over one job of --configs3-windows (default 1,000,000 = BASELINE.json configs[3]) windows split into contiguous blocks per rank
(strong; a single pass because 25 driver steps of a 41 GPU-second job would not "finish within minutes"); `in_process` — the same
N devices driven from ONE process through dd_compute_likelihoods_multi (no collective; the form that matches the reference's
single process, DInDel.cpp:4074), run by rank 0 after the ranks have finished.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start N ranks as a fresh child (this process has made no GPU call and
    makes none), hand its stdout / stderr through (rank 0 prints the one JSON line) and return its exit code."""
    import argparse as _ap
    import subprocess
    pre = _ap.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    known, _ = pre.parse_known_args(argv)
    if known.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


if __name__ == "__main__":
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import numpy as np
import torch
import torch.distributed as dist

from dindel_tgi_amd import capi, synth
from dindel_tgi_amd.device import DeviceBatch

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# fp64 VALU ceiling for this recurrence, SURVEY.md §8(d): ~32 fp64 add/cmp/select per cell against
# 256 CU x 4 SIMD x 16 fp64 lanes/clk x 2.4 GHz = 39.3e12 non-FMA fp64 op/s
VALU_CELLS_PER_S = 39.3e12 / 32.0


def algorithmic_bytes_per_pair(L, Hs, R):
    """L bases + L quals + 16 (mapQual, start, flags) + Hs/R + 32 (ll, llOn, llOff, packed
    flags/counters) + 2L (hpos int16) = 4L + 48 + Hs/R."""
    return 4.0 * L + 48.0 + float(Hs) / R


def measured_traffic(n_pairs, kernel_sub="dd_hmm_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, profiles/r*/..._pmc.json) — only when the profiled launch had
    exactly this many pairs AND the file was measured on the kernel sources of this tree (`source_id`, a hash of
    csrc/hmm_kernel.{h,hip}; a file of another build is reported as stale, never quoted); otherwise None.
    bench.py itself cannot collect PMC counters."""
    import glob
    best, stale = None, None
    here = capi.kernel_source_id(kernel_sub)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if kernel_sub not in json.dumps(d.get("kernel", "")):
            continue
        for cfg in d.values():
            if isinstance(cfg, dict) and cfg.get("pairs_per_launch") == n_pairs and "hbm_bytes_raw" in cfg:
                if d.get("source_id") == here:
                    best = dict(bytes=cfg["hbm_bytes_raw"], source=os.path.relpath(f, ROOT))
                else:
                    stale = dict(bytes=None, source="stale: %s was measured on another build of the kernel (source_id %s, this tree %s)"
                                                    % (os.path.relpath(f, ROOT), d.get("source_id"), here))
    return best or stale


def cpu_baseline(pb, params, seconds_target=15.0, faster=False):
    """The oracle (CPU restatement, kind 'port') on a bounded sample of the same workload."""
    from tests import _oracle
    threads = max(1, min(16, os.cpu_count() or 1))
    n_win = min(pb.n_windows, 2 * threads)          # calibration pass: every thread busy
    t0 = time.time()
    _oracle.batch(params, pb, nthreads=threads, first_window=0, n_win=n_win, faster=faster)
    dt = time.time() - t0
    per_win = dt / n_win
    n_win = int(max(threads, min(pb.n_windows, seconds_target / max(per_win, 1e-9))))
    t0 = time.time()
    _oracle.batch(params, pb, nthreads=threads, first_window=0, n_win=n_win, faster=faster)
    dt = time.time() - t0
    a = pb.a
    h1, r1 = int(a["win_hap_off"][n_win]), int(a["win_read_off"][n_win])
    cells = 0
    hl = np.diff(a["hap_seq_off"]).astype(np.int64)
    rl = np.diff(a["read_seq_off"]).astype(np.int64)
    for w in range(n_win):
        cells += int(hl[a["win_hap_off"][w]:a["win_hap_off"][w + 1]].sum()) * int(rl[a["win_read_off"][w]:a["win_read_off"][w + 1]].sum())
    # the same restatement on ONE thread, on a smaller sample (SURVEY §8(d) asks for both figures)
    n1 = max(1, min(pb.n_windows, int(3.0 / max(per_win * threads, 1e-9))))
    t0 = time.time()
    _oracle.batch(params, pb, nthreads=1, first_window=0, n_win=n1, faster=faster)
    dt1 = time.time() - t0
    cells1 = 0
    for w in range(n1):
        cells1 += int(hl[a["win_hap_off"][w]:a["win_hap_off"][w + 1]].sum()) * int(rl[a["win_read_off"][w]:a["win_read_off"][w + 1]].sum())
    return dict(value=cells / dt, unit="cells/s", cores=threads, kind="port", value_1_thread=cells1 / dt1,
                sample_1_thread="first %d windows, %.1f s" % (n1, dt1),
                sample="first %d of the %d windows (%d pairs), oracle/dd_oracle.c with %d OpenMP threads, %.1f s"
                       % (n_win, pb.n_windows, int(pb.win_pair_off[n_win]), threads, dt),
                windows_per_s=n_win / dt)


def window_loop_leg(faster, windows=100000):
    """The batched window loop (host/dindel_gpu: BAM -> read selection -> GPU likelihoods -> diploidGLF -> .glf.txt) on a synthetic
    sample written by tools/n2_pipeline_bench.py, as a child process; the driver's own clock (set-up to last line written)."""
    import shutil
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="dd_window_loop_")
    try:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "n2_pipeline_bench.py"), "--windows", str(windows), "--dir", d] + (["--faster"] if faster else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        best = None
        for line in r.stdout.split("\n"):
            if line.startswith("{"):
                row = json.loads(line)
                wall = float(row["driver"].split("wall=")[1].split()[0])
                if best is None or wall < best["seconds"]:
                    f = lambda key: float(row["driver"].split(key + "=")[1].split()[0].rstrip(")"))
                    best = {"seconds": wall, "windows_per_s": windows / wall, "stages": row["driver"], "lines_written": row["dip_map_lines"] * 8 + 1,
                            # the rate once the pipeline is full (from the batch that completed the first fifth of the windows to the last one)
                            "steady_windows_per_s": f("steady_windows_per_s") if "steady_windows_per_s=" in row["driver"] else None,
                            # host CPU per window, summed over the threads of each stage: read selection, packing, diploidGLF + lines
                            "cpu_ms_per_window": {"prepare": 1e3 * f("prepare") / windows, "pack": 1e3 * f("pack") / windows, "reduce": 1e3 * f("work") / windows}}
        if best is None:
            return {"error": (r.stderr or r.stdout)[-400:]}
        best["what"] = ("%d windows x 8 haplotypes x ~200 reads of 100 bp from a coordinate-sorted BAM through getReads, the likelihood kernels, "
                        "diploidGLF and the .glf.txt writer (dindel_tgi_amd/host/dindel_gpu); best of two runs, the driver's clock" % windows)
        return best
    except Exception as e:                                   # the headline figures do not depend on this leg
        return {"error": repr(e)[:400]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ragged_leg(params, device, windows=2400, steps=3):
    """The shapes real windows have (synth.generate_ragged: 121-bp-and-up reference haplotypes as python/makeWindows.py:72-75 cuts them, 2-12
    candidate haplotypes of different lengths per window, 20-400 reads of 36 / 76 / 100 / 150 bp, mixed qualities), resident in HBM, one
    dd_launch_device per step = one launch per (haplotype-length class, read-length class).  Beside the headline figure, never part of it."""
    try:
        pb = synth.generate_ragged(windows)
        dev = DeviceBatch(pb, params, device)
        dev.launch()
        torch.cuda.synchronize(device)
        ev = []
        for _ in range(steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dev.launch(); e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize(device)
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        log = capi.launch_log()
        st = dev.out["status"][:pb.n_pairs]
        assert int((st != 0).sum().item()) == 0, "non-OK pair status in the ragged batch"
        # lane utilisation of a launch: states its haplotypes have / positions their wavefronts sweep
        hl = np.diff(pb.a["hap_seq_off"]).astype(np.int64)
        rl = np.diff(pb.a["read_seq_off"]).astype(np.int64)
        hw = np.repeat(np.arange(pb.n_windows), np.diff(pb.a["win_hap_off"]))
        rw = np.repeat(np.arange(pb.n_windows), np.diff(pb.a["win_read_off"]))
        launches = []
        ci_of_hap = lambda mh: int(np.searchsorted(capi.HAP_CLASS_BOUNDS, mh, side="left"))
        for rec in log:
            ci = int(np.searchsorted(capi.HAP_CLASS_BOUNDS, rec["max_hap"], side="left"))    # the launch's lane tiling = haplotype-length class
            hsel = (hl > (capi.HAP_CLASS_BOUNDS[ci - 1] if ci else 0)) & (hl <= capi.HAP_CLASS_BOUNDS[ci])
            # the haplotypes of exactly the tiling's full length may run apart (capi.cpp build_launch_classes: the others then take the folded build)
            apart = any(q is not rec and ci_of_hap(q["max_hap"]) == ci and (q["min_read"], q["max_read"] > 160) == (rec["min_read"], rec["max_read"] > 160)
                        and (q["max_hap"] == capi.HAP_CLASS_BOUNDS[ci]) != (rec["max_hap"] == capi.HAP_CLASS_BOUNDS[ci]) for q in log)
            if apart:
                hsel &= (hl == capi.HAP_CLASS_BOUNDS[ci]) if rec["max_hap"] == capi.HAP_CLASS_BOUNDS[ci] else (hl < capi.HAP_CLASS_BOUNDS[ci])
            rsel = (rl >= rec["min_read"]) & (rl <= rec["max_read"])
            # read classes 0 / 1 of a tiling split the WINDOWS by their longest read up to 160 bp (capi.cpp build_launch_classes): a launch
            # that starts at 1 bp owns the windows whose longest such read is above the longest read of the tiling's shorter launch
            ci_of = lambda mh: int(np.searchsorted(capi.HAP_CLASS_BOUNDS, mh, side="left"))
            below = [q["max_read"] for q in log if ci_of(q["max_hap"]) == ci and q["min_read"] == 1 and q["max_read"] < rec["max_read"]]
            wmax = np.zeros(pb.n_windows, np.int64)
            np.maximum.at(wmax, rw[rl <= 160], rl[rl <= 160])
            if rec["min_read"] == 1:
                rsel &= (wmax[rw] > (max(below) if below else 0)) & (wmax[rw] <= rec["max_read"])
            rsum = np.bincount(rw[rsel], weights=rl[rsel], minlength=pb.n_windows)
            cells = float((hl[hsel] * rsum[hw[hsel]]).sum())
            pos = 64 * rec["K"] // rec["pairs_per_wave"]
            launches.append({"K": rec["K"], "pairs_per_wave": rec["pairs_per_wave"], "D": rec["D"], "bt": "hbm" if rec["gbt"] else "lds", "fold": bool(rec["fold"]),
                             "haplotypes": rec["n_haps"], "waves": rec["waves"], "read_split": rec["split"], "dynamic_items": bool(rec["dynamic"]), "hap_len": [int(hl[hsel].min()), int(hl[hsel].max())] if hsel.any() else None,
                             "read_len": [rec["min_read"], rec["max_read"]], "share_of_cells": cells / pb.cells,
                             "lane_utilisation": float((hl[hsel] + 2).mean() / pos) if hsel.any() else None,
                             "ms": rec["us"] / 1e3 if rec["us"] >= 0 else None})
        return {"what": "%d windows as the reference's pipeline shapes them (synth.generate_ragged: haplotypes %d-%d bp, %d-%d per window, reads %d-%d bp, "
                        "%d-%d per window; mixed qualities), HBM-resident, one dd_launch_device per step" %
                        (pb.n_windows, int(hl.min()), int(hl.max()), int(np.diff(pb.a["win_hap_off"]).min()), int(np.diff(pb.a["win_hap_off"]).max()),
                         int(rl.min()), int(rl.max()), int(np.diff(pb.a["win_read_off"]).min()), int(np.diff(pb.a["win_read_off"]).max())),
                "cells_per_s": pb.cells / (ms * 1e-3), "windows_per_s": pb.n_windows / (ms * 1e-3), "pairs": pb.n_pairs, "cells": pb.cells,
                "ms_per_step": ms, "steps": steps, "launches": launches}
    except Exception as e:                          # noqa: BLE001  (the headline figures do not depend on this leg)
        return {"error": repr(e)[:400]}


def in_process_leg(pb, params, world, devices, dev, args):
    """N devices driven from ONE process: dd_compute_likelihoods_multi on a batch of N x --windows windows held in host memory (contiguous
    window blocks balanced by cells, one host thread + arena + streams per device, no collective) — the form that matches the reference's
    single process (DInDel.cpp:4074).  Host pointers in, host pointers out: H2D and D2H of every output are inside the time.  The extra
    leg never takes the headline down: an error is reported in the line instead."""
    try:
        import ctypes as C
        from dindel_tgi_amd.batch import RESULT_DTYPES, alloc_result, result_lengths
        lib = capi.load()
        # host memory of the leg: the tiled batch and its (pageable) result arrays, ~0.41 MB per configs[1] window — rank 0 keeps it
        # under --in-process-gib (default 8): at N = 8 that is 2,400 windows per device instead of 10,000 (32 GB)
        per_window = (sum(np.dtype(RESULT_DTYPES[k]).itemsize * n for k, n in result_lengths(pb).items()) + pb.read_bases * 2 + pb.hap_bases) / max(pb.n_windows, 1)
        per_dev = int(max(1, min(pb.n_windows, args.in_process_gib * 2 ** 30 / per_window / world)))
        big = synth.tile(pb if per_dev == pb.n_windows else pb.slice_windows(0, per_dev), world)
        arrs, res = alloc_result(big)
        hb = big.ctypes_batch()
        devs = (C.c_int32 * len(devices))(*devices)
        fn = lib.dd_compute_likelihoods_faster_multi if args.faster else lib.dd_compute_likelihoods_multi
        for _ in range(2):                          # second call: page tables of the result arrays and the per-device arenas are warm
            t0 = time.perf_counter()
            rc = fn(C.byref(params), C.byref(hb), C.byref(res), devs, len(devices))
            dt = time.perf_counter() - t0
            if rc != 0:
                return {"error": "dd_compute_likelihoods_multi rc=%d: %s" % (rc, capi.last_error())}
        n = int(big.n_pairs // world)               # block 0 of the tiled batch is (the front of) rank 0's own batch: same numbers as its resident launch
        same = bool(np.array_equal(arrs["ll"][:n], dev.out["ll"][:n].cpu().numpy())) and bool((arrs["status"][:big.n_pairs] == 0).all())
        host_gib = sum(a.nbytes for a in arrs.values()) / 2 ** 30
        lib.dd_release_cache()
        return {"what": "dd_compute_likelihoods_multi(devices=%s) on %d windows held in (pageable) host memory: one process, one host thread per "
                        "device, contiguous window blocks, no collective; H2D + kernels + D2H of every output" % (list(devices), big.n_windows),
                "seconds": dt, "cells_per_s": big.cells / dt, "windows_per_s": big.n_windows / dt, "devices": list(devices),
                "windows_per_device": per_dev, "host_result_gib": host_gib, "equals_resident_launch": same}
    except Exception as e:                          # noqa: BLE001
        return {"error": repr(e)[:400]}


def build_plan(total_windows, rank, world, args, params, device, seed=0x9E3779B9):
    """Strong scaling: rank r's contiguous block of a `total_windows` job as [(DeviceBatch, repeats)] — sub-batches that fit the int32
    read-base offsets of a batch; only min(sub-batch, 2500) windows are generated, the rest are replicas (synthetic data either way)."""
    from dindel_tgi_amd.shard import window_block
    w0, w1 = window_block(total_windows, rank, world)
    n_mine = w1 - w0
    n_sub = max(1, -(-n_mine // args.max_batch_windows))
    bs = n_mine // n_sub                       # n_sub sub-batches of bs windows (+ one of `rem` windows)
    rem = n_mine - bs * n_sub
    gen = min(bs, 2500)
    base = synth.generate(gen, H=args.haps, R=args.reads, L=args.read_len, hap_len=args.hap_len, seed=seed + rank)
    reps = -(-bs // gen)
    pb = synth.tile(base, reps).slice_windows(0, bs) if reps * gen != bs else synth.tile(base, reps)
    plan = [(DeviceBatch(pb, params, device), n_sub)]
    if rem:
        plan.append((DeviceBatch(pb.slice_windows(0, rem), params, device), 1))
    return plan, n_mine


def launch_only_rehearsal(args, world, rank):
    """`--rehearse` in a process that sees no GPU (the CPU test of the launcher): rendezvous over gloo, the same gather of per-pair
    records with the same shapes on zero-filled host tensors, one JSON line from rank 0 — and no kernel: `value` is null."""
    pb = synth.generate(min(args.windows, 50), H=args.haps, R=args.reads, L=args.read_len, hap_len=args.hap_len, seed=0x9E3779B9 + rank)
    n = pb.n_pairs
    send = {"ll": torch.full((n,), float(rank), dtype=torch.float64), "offHap": torch.zeros(n, dtype=torch.uint8), "offHapHMQ": torch.zeros(n, dtype=torch.uint8)}
    for k, v in send.items():
        bufs = [torch.empty_like(v) for _ in range(world)] if rank == 0 else None
        dist.gather(v, bufs, dst=0)
        if rank == 0 and k == "ll":
            assert all(float(b[0]) == float(r) for r, b in enumerate(bufs)), "gather order"
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "read-haplotype HMM cells/s", "value": None, "unit": "cells/s", "n_gpus": world, "ranks": dist.get_world_size(),
                          "backend": dist.get_backend(), "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "rehearsal": "launch-only: this process sees no GPU, no kernel ran (the likelihood path has no CPU fallback)",
                          "config": {"workload": "launcher + rendezvous + gather only", "pairs_per_rank": n}}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=10000, help="windows per GPU")
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--reads", type=int, default=200)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--hap-len", type=int, default=120)
    ap.add_argument("--max-length-del", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-window-loop", action="store_true", help="skip the BAM -> .glf.txt window-loop leg (N=1)")
    ap.add_argument("--no-ragged", action="store_true", help="skip the real-shaped (ragged) batch leg (N=1)")
    ap.add_argument("--ragged-only", action="store_true",
                    help="N=1: time only the real-shaped batch (--steps launches of it) and print its JSON — for rocprofv3 runs of that leg's kernels")
    ap.add_argument("--ragged-windows", type=int, default=2400)
    ap.add_argument("--host-api", action="store_true", help="(kept for old command lines: the host-API legs now run by default at N=1)")
    ap.add_argument("--kernel-only", action="store_true",
                    help="skip the legs beside the headline figure: at N=1 the host API (copies included) and the C++ adapter end to end, "
                         "at N>1 the configs[3] pass and the one-process leg")
    ap.add_argument("--total-windows", type=int, default=0,
                    help="strong scaling as the HEADLINE figure: ONE job of this many windows split over the ranks in contiguous blocks, "
                         "every step a full pass (BASELINE.json configs[3] = 1000000); default 0 = weak scaling with --windows per GPU")
    ap.add_argument("--configs3-windows", type=int, default=1000000,
                    help="N>1: size of the job whose single strong-scaling pass is reported under `configs3` (0 = skip)")
    ap.add_argument("--max-batch-windows", type=int, default=50000,
                    help="strong scaling: largest sub-batch a rank keeps resident (a batch holds < 2^31 read bases)")
    ap.add_argument("--in-process-gib", type=float, default=8.0,
                    help="N>1: host memory (GiB) the one-process leg's tiled batch and result arrays may take on rank 0")
    ap.add_argument("--faster", action="store_true",
                    help="time the secondary --faster model (ObservationModelS, SURVEY row A13) instead of the headline path; "
                         "the JSON line then names that model in `metric` and is not the BASELINE metric")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo backend, every rank on cuda:0 (checks the N>1 code path; "
                         "the number it prints is not a multi-GPU measurement).  Without any GPU: launcher + rendezvous + gather only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d — run `python bench.py --gpus N` (it starts the ranks itself) or "
                 "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (world, args.gpus))
    if args.rehearse:
        local_rank = 0
    host_group = None
    cpu_base = None
    if world > 1 and rank == 0 and not args.no_cpu_baseline and not args.rehearse and args.total_windows == 0:
        # the CPU baseline belongs in the N > 1 line too (a SCALE record without it is incomplete): rank 0 times it BEFORE the process
        # group exists — the other ranks wait in the rendezvous meanwhile, no collective kernel spins on their GPUs
        p0 = capi.params_cli_defaults()
        p0.maxLengthDel = args.max_length_del
        cpu_base = cpu_baseline(synth.generate(min(args.windows, 400), H=args.haps, R=args.reads, L=args.read_len, hap_len=args.hap_len, seed=0x9E3779B9),
                                p0, faster=args.faster)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            if not torch.cuda.is_available():
                return launch_only_rehearsal(args, world, rank)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
            host_group = dist.new_group(backend="gloo")     # host-side waits that must not park a spinning kernel on a GPU
    assert torch.cuda.is_available(), "bench.py needs a GPU (the likelihood path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    params = capi.params_cli_defaults()
    params.maxLengthDel = args.max_length_del
    if args.ragged_only:
        assert world == 1, "--ragged-only is an N=1 leg"
        print(json.dumps({"metric": "read-haplotype HMM cells/s, real-shaped (ragged) batch", "unit": "cells/s", "n_gpus": 1, "dtype": "f64", "data": "synthetic",
                          **ragged_leg(params, device, windows=args.ragged_windows, steps=max(args.steps, 1))}))
        return
    # rank r owns windows [r*W, (r+1)*W) of the job; its block is generated from seed+r
    strong = args.total_windows > 0
    if strong:
        plan, windows_this_rank = build_plan(args.total_windows, rank, world, args, params, device)
        dev = plan[0][0]
        pb = dev.pb
    else:
        pb = synth.generate(args.windows, H=args.haps, R=args.reads, L=args.read_len, hap_len=args.hap_len,
                            seed=0x9E3779B9 + rank)
        dev = DeviceBatch(pb, params, device)
        plan = [(dev, 1)]
        windows_this_rank = args.windows
    n_pairs, cells = pb.n_pairs, pb.cells
    step_pairs = sum(d.pb.n_pairs * k for d, k in plan)
    step_cells = sum(d.pb.cells * k for d, k in plan)

    GATHER = ("ll", "offHap", "offHapHMQ")          # per-pair records the downstream reduction consumes

    def make_gather(main_dev):
        """The step's one exchange: gather of the per-pair records of `main_dev`-sized (or shorter) batches to rank 0."""
        bufs = None
        if world > 1 and rank == 0:
            gdev = "cpu" if args.rehearse else device
            bufs = {k: [torch.empty(main_dev.out[k].shape, dtype=main_dev.out[k].dtype, device=gdev) for _ in range(world)] for k in GATHER}

        def gather(d):
            for k in GATHER:
                src = d.out[k].cpu() if args.rehearse else d.out[k]
                n = src.numel()                     # a remainder sub-batch: same collective on the front of the buffers
                dist.gather(src, [b[:n] for b in bufs[k]] if rank == 0 else None, dst=0)
        return gather, bufs

    def timed(the_plan, gather, steps, warmup, events=None):
        """`warmup` untimed + `steps` timed passes over the_plan, barrier + synchronize either side, MAX over ranks; seconds."""
        main_dev = the_plan[0][0]

        def step(ev):
            for d, k in the_plan:
                for _ in range(k):
                    if ev is not None and d is main_dev:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    (d.launch_faster if args.faster else d.launch)()
                    if ev is not None and d is main_dev:
                        e1.record()
                        ev.append((e0, e1))
                    if world > 1:
                        gather(d)
        for _ in range(warmup):
            step(None)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(events)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    gather, gather_bufs = make_gather(dev)
    ev = []
    elapsed = timed(plan, gather, args.steps, args.warmup, ev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")   # one launch of the main (sub-)batch

    # sanity: every pair finished with status OK and a finite negative log-likelihood
    res_status = dev.out["status"][:n_pairs]
    assert int((res_status != 0).sum().item()) == 0, "non-OK pair status in the bench batch"
    assert bool(torch.isfinite(dev.out["ll"][:n_pairs]).all().item())

    if world > 1 and rank == 0 and len(plan) == 1:
        # the gathered block of rank 0 is its own result (sanity of the collective's layout)
        assert torch.equal(gather_bufs["ll"][0][:n_pairs].cpu(), dev.out["ll"][:n_pairs].cpu())

    # ---- N > 1: ONE pass over the configs[3] job (strong scaling), every rank its contiguous block ----
    configs3 = None
    if world > 1 and not strong and not args.kernel_only and args.configs3_windows > 0:
        del gather_bufs
        plan3, mine3 = build_plan(args.configs3_windows, rank, world, args, params, device, seed=0xC0F1C53)
        gather3, _bufs3 = make_gather(plan3[0][0])
        plan3[0][0].launch()                                  # the kernels are warm; this touches the new buffers once
        gather3(plan3[0][0])
        dt3 = timed(plan3, gather3, 1, 0)
        cells3 = sum(d.pb.cells * k for d, k in plan3) * (args.configs3_windows / max(mine3, 1))
        configs3 = {"what": "ONE pass over one job of %d windows x %d haplotypes x %d reads (BASELINE.json configs[3]): rank r computes the r-th "
                            "contiguous block in sub-batches and gathers ll + flags of every sub-batch to rank 0" % (args.configs3_windows, args.haps, args.reads),
                    "scaling": "strong", "seconds": dt3, "cells_per_s": cells3 / dt3, "windows_per_s": args.configs3_windows / dt3,
                    "windows_per_gpu": mine3, "sub_batches_per_gpu": [[d.pb.n_windows, k] for d, k in plan3]}
        del plan3, gather3, _bufs3
        torch.cuda.empty_cache()

    # ---- N > 1: the same devices from ONE process (dd_compute_likelihoods_multi), run by rank 0 while the others wait on the host ----
    in_process = None
    if world > 1 and not strong and not args.kernel_only:
        (dist.barrier(group=host_group) if host_group is not None else dist.barrier())
        if rank == 0:
            in_process = in_process_leg(pb, params, world, [0] * world if args.rehearse else list(range(world)), dev, args)
        (dist.barrier(group=host_group) if host_group is not None else dist.barrier())

    if rank == 0:
        if strong:      # every rank's block has the same per-window shape: the job's cells = this rank's x (job windows / its windows)
            total_cells = int(step_cells * (args.total_windows / max(windows_this_rank, 1))) * args.steps
            total_windows = args.total_windows * args.steps
        else:
            total_cells = cells * world * args.steps
            total_windows = args.windows * world * args.steps
        value = total_cells / elapsed
        bpp = algorithmic_bytes_per_pair(args.read_len, args.hap_len, args.reads)
        default_shape = (args.windows, args.haps, args.reads, args.read_len, args.hap_len, args.max_length_del) == (10000, 8, 200, 100, 120, 5)
        achieved = bpp * n_pairs / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "read-haplotype HMM cells/s", "value": value, "unit": "cells/s",
            "n_gpus": world, "ranks": dist.get_world_size() if world > 1 else 1,
            "backend": ("%s (RCCL over xGMI)" % dist.get_backend() if dist.get_backend() == "nccl" else dist.get_backend()) if world > 1 else None,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("one job of %d windows over %d GPU(s) (BASELINE.json configs[3] shape), " % (args.total_windows, world) if strong
                                    else "%d windows/GPU " % args.windows) +
                                   "x %d haplotypes x %d reads, %d bp reads (Q30), %d bp haplotypes, maxLengthDel=%d%s"
                                   % (args.haps, args.reads, args.read_len, args.hap_len, args.max_length_del,
                                      " (BASELINE.json configs[1])" if default_shape and not strong else ""),
                       "windows_per_gpu": windows_this_rank, "pairs_per_gpu": step_pairs, "cells_per_gpu": step_cells,
                       "sub_batches_per_gpu": [[d.pb.n_windows, k] for d, k in plan],
                       "sharding": "contiguous window blocks per rank; gather of ll+flags to rank 0" if world > 1 else "single GPU"},
            "windows_per_s": total_windows / elapsed,
            "pairs_per_s": total_cells / elapsed / (cells / n_pairs),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": (measured_traffic(n_pairs) or {}).get("bytes"),
                         "traffic_source": (measured_traffic(n_pairs) or {}).get("source"),
                         "algorithmic_bytes_per_launch": bpp * n_pairs,
                         "kernel": capi.load().dd_kernel_name().decode(), "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_pair": bpp,
                         "note": "scalar max-plus fp64 recurrence: bound by fp64 VALU/LDS, not HBM (SURVEY §8d); "
                                 "valu_fp64 gives the binding ceiling"},
            "valu_fp64": {"achieved_cells_per_s": cells / (kern_ms * 1e-3), "ceiling_cells_per_s": VALU_CELLS_PER_S,
                          "frac": cells / (kern_ms * 1e-3) / VALU_CELLS_PER_S},
        }
        if configs3 is not None:
            out["configs3"] = configs3
        if in_process is not None:
            out["in_process"] = in_process
        if world == 1 and not strong and not args.kernel_only:
            import ctypes as C
            from dindel_tgi_amd.batch import alloc_result
            lib = capi.load()
            arrs, res = alloc_result(pb)
            hb = pb.ctypes_batch()
            for _ in range(2):                      # second call: page tables of the result arrays are warm
                t0 = time.perf_counter()
                rc = (lib.dd_compute_likelihoods_faster if args.faster else lib.dd_compute_likelihoods)(C.byref(params), C.byref(hb), C.byref(res), local_rank)
                dt = time.perf_counter() - t0
                assert rc == 0, capi.last_error()
            out["host_api"] = {"seconds": dt, "cells_per_s": cells / dt, "windows_per_s": args.windows / dt,
                               "note": "dd_compute_likelihoods with (pageable) host pointers: H2D of the batch, kernels, D2H of every output"}
            out["windows_per_s_incl_copies"] = args.windows / dt
            # the C++ drop-in: LikelihoodEngine::computeLikelihoodsBatch on vector<Haplotype> / vector<Read> objects of the same shape
            from dindel_tgi_amd import hostlib
            kw = dict(H=args.haps, R=args.reads, L=args.read_len, HL=args.hap_len, faster=args.faster, device=local_rank)
            lazy = hostlib.bench_batch(args.windows, reps=2, **kw)
            lean = hostlib.bench_batch(args.windows, reps=2, keep_alignments=False, **kw)
            eager = hostlib.bench_batch(min(args.windows, 2000), reps=1, eager=True, **kw)
            assert lazy["errors"] == lean["errors"] == eager["errors"] == 0
            out["windows_per_s_end_to_end"] = args.windows / lazy["seconds"]
            out["end_to_end"] = {
                "what": "dindel::LikelihoodEngine::computeLikelihoodsBatch (C++ mirror of DetInDel::computeLikelihoods, DInDel.cpp:1707-1739) "
                        "on %d windows of the same shape held as the reference's objects: pack + H2D + kernels + D2H + per-window "
                        "status scan; records delivered as lazy views (every scalar in place, MLAlignment maps built on demand)" % args.windows,
                "lazy_records": {"windows_per_s": args.windows / lazy["seconds"], **{k: lazy[k] for k in ("seconds", "pack", "device", "finish")}},
                "lazy_records_no_alignments": {"windows_per_s": args.windows / lean["seconds"], **{k: lean[k] for k in ("seconds", "pack", "device", "finish")},
                                               "note": "hpos (45 % of the result bytes) not copied back; a window is recomputed if a consumer asks for an alignment"},
                "eager_records": {"windows_per_s": eager["windows"] / eager["seconds"], "windows": eager["windows"],
                                  **{k: eager[k] for k in ("seconds", "pack", "device", "finish")},
                                  "note": "every MLAlignment rebuilt (maps, strings, vectors) as the literal drop-in does"},
                "frac_of_kernel_only": (args.windows / lazy["seconds"]) / (total_windows / elapsed)}
            if not args.no_window_loop:
                out["window_loop"] = window_loop_leg(args.faster)
        if world == 1 and not strong and not args.no_ragged and not args.faster:
            out["ragged"] = ragged_leg(params, device, windows=args.ragged_windows)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pb, params, faster=args.faster)
        elif cpu_base is not None:                        # N > 1: timed by rank 0 before the ranks met (see above)
            out["cpu_baseline"] = cpu_base
        if args.faster:
            out["metric"] = "read-haplotype cells/s, --faster model (ObservationModelS)"
            out["roofline"]["kernel"] = "dd_faster_kernel"
            tr = measured_traffic(n_pairs, "dd_faster_kernel") or {}
            out["roofline"]["traffic"] = tr.get("bytes")
            out["roofline"]["traffic_source"] = tr.get("source")
            out["roofline"]["note"] = "k-mer voting + <=16-diagonal Viterbi: instruction-issue bound like the headline kernel"
            del out["valu_fp64"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
