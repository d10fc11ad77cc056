"""Flat (CSR) batches of realignment windows for the C ABI (include/dindel_hmm.h: dd_batch / dd_result).

A *window* is what DetInDel::computeLikelihoods receives for one call (DInDel.cpp:1707): the candidate
haplotypes, the reads fetched for the window and leftPos.  A batch packs many windows so one launch
fills the GPU.  Pure numpy host logic; no compute.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi


@dataclass
class ReadRec:
    """The fields of Read the path uses (Read.hpp)."""
    seq: str
    qual: Sequence[float]          # P(base correct) per base — Read::qual
    mapQual: float                 # Read::mapQual (probability)
    start: int                     # uint32_t(read.posStat.first)
    unmapped: bool = False         # bam->core.flag & BAM_FUNMAP
    # mate / library inputs of the insert-size prior (mapUnmappedReads; ObservationModelFB.cpp:279-292)
    paired: bool = False           # read.isPaired()
    mate_unmapped: bool = False    # read.mateIsUnmapped()
    mate_reverse: bool = False     # read.mateIsReverse()
    mate_same_tid: bool = False    # bam->core.tid == bam->core.mtid
    mate_pos: int = -1             # read.matePos
    mate_len: int = -1             # read.mateLen (-1 unknown)
    lib: int = 0                   # index into the batch's libraries


@dataclass
class Window:
    hap_start: int                                  # leftPos
    haps: List[str]
    reads: List[ReadRec]
    hap_vars: Optional[List[List[Tuple[int, int]]]] = None   # per hap: (startRead, endRead) of hap.indels then hap.snps
    hap_var_flanks: Optional[List[List[Tuple[int, int, int]]]] = None   # per hap variant: (leftFlankRead, rightFlankRead, kind 0/1=DEL/2=INS)


def phred_to_prob(phred):
    """Read.hpp:127-131, 143-148: 1-10^(-Q/10) clamped to [1e-16, 1-1e-16]."""
    q = 1.0 - np.power(10.0, -np.asarray(phred, dtype=np.float64) / 10.0)
    return np.clip(q, 1e-16, 1.0 - 1e-16)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class PackedBatch:
    """Numpy arrays in dd_batch layout plus the derived offsets consumers need."""

    FIELDS = ["win_hap_off", "win_read_off", "win_hap_start", "hap_seq_off", "hap_seq", "hap_var_off", "hap_var",
              "read_seq_off", "read_seq", "read_qidx", "read_mqidx", "read_start", "read_flags",
              "qual_table", "mapq_table"]

    def __init__(self, hap_var_flank=None, mate=None, **arrays):
        self.hap_var_flank = None if hap_var_flank is None else np.ascontiguousarray(hap_var_flank, dtype=np.int32)
        # mate: None, or dict(read_mate_pos, read_mate_len, read_lib, lib_off, lib_prob, lib_p95) for mapUnmappedReads
        self.mate = None
        if mate is not None:
            dtm = dict(read_mate_pos=np.int32, read_mate_len=np.int32, read_lib=np.uint8, lib_off=np.int32, lib_prob=np.float64, lib_p95=np.float64)
            self.mate = {k: np.ascontiguousarray(mate[k], dtype=dtm[k]) for k in dtm}
        self.a = {}
        dt = dict(win_hap_off=np.int32, win_read_off=np.int32, win_hap_start=np.uint32, hap_seq_off=np.int32,
                  hap_seq=np.uint8, hap_var_off=np.int32, hap_var=np.int32, read_seq_off=np.int32,
                  read_seq=np.uint8, read_qidx=np.uint8, read_mqidx=np.uint8, read_start=np.uint32,
                  read_flags=np.uint8, qual_table=np.float64, mapq_table=np.float64)
        for k in self.FIELDS:
            self.a[k] = np.ascontiguousarray(arrays[k], dtype=dt[k])
        a = self.a
        self.n_windows = len(a["win_hap_start"])
        self.n_haps = len(a["hap_seq_off"]) - 1
        self.n_reads = len(a["read_seq_off"]) - 1
        H = np.diff(a["win_hap_off"]).astype(np.int64)
        R = np.diff(a["win_read_off"]).astype(np.int64)
        rso = a["read_seq_off"].astype(np.int64)
        SL = rso[a["win_read_off"][1:]] - rso[a["win_read_off"][:-1]]
        hvo = a["hap_var_off"].astype(np.int64)
        NV = hvo[a["win_hap_off"][1:]] - hvo[a["win_hap_off"][:-1]]
        z = np.zeros(1, np.int64)
        self.win_pair_off = np.concatenate([z, np.cumsum(H * R)])
        self.win_hpos_off = np.concatenate([z, np.cumsum(H * SL)])
        self.win_varcov_off = np.concatenate([z, np.cumsum(NV * R)])
        self.n_pairs = int(self.win_pair_off[-1])
        self.hpos_len = int(self.win_hpos_off[-1])
        self.var_cov_len = int(self.win_varcov_off[-1])
        hl = np.diff(a["hap_seq_off"]).astype(np.int64)
        rl = np.diff(a["read_seq_off"]).astype(np.int64)
        self.max_hap_len = int(hl.max()) if len(hl) else 0
        self.max_read_len = int(rl.max()) if len(rl) else 0
        # cells = sum over pairs of L*Hs = sum_w (sum_h Hs)(sum_r L)
        hs_cum = np.concatenate([z, np.cumsum(hl)])
        SH = hs_cum[a["win_hap_off"][1:]] - hs_cum[a["win_hap_off"][:-1]]
        self.cells = int((SH * SL).sum())
        self.read_bases = int(rl.sum())
        self.hap_bases = int(hl.sum())

    def ctypes_batch(self):
        a = self.a
        b = capi.dd_batch()
        b.n_windows = self.n_windows
        b.win_hap_off = _ptr(a["win_hap_off"], capi.c_i32p)
        b.win_read_off = _ptr(a["win_read_off"], capi.c_i32p)
        b.win_hap_start = _ptr(a["win_hap_start"], capi.c_u32p)
        b.hap_seq_off = _ptr(a["hap_seq_off"], capi.c_i32p)
        b.hap_seq = C.cast(a["hap_seq"].ctypes.data, C.c_char_p)
        b.hap_var_off = _ptr(a["hap_var_off"], capi.c_i32p)
        b.hap_var = _ptr(a["hap_var"], capi.c_i32p)
        b.read_seq_off = _ptr(a["read_seq_off"], capi.c_i32p)
        b.read_seq = C.cast(a["read_seq"].ctypes.data, C.c_char_p)
        b.read_qidx = _ptr(a["read_qidx"], capi.c_u8p)
        b.read_mqidx = _ptr(a["read_mqidx"], capi.c_u8p)
        b.read_start = _ptr(a["read_start"], capi.c_u32p)
        b.read_flags = _ptr(a["read_flags"], capi.c_u8p)
        b.n_qual = len(a["qual_table"])
        b.qual_table = _ptr(a["qual_table"], capi.c_f64p)
        b.n_mapq = len(a["mapq_table"])
        b.mapq_table = _ptr(a["mapq_table"], capi.c_f64p)
        if self.hap_var_flank is not None and len(self.hap_var_flank):
            b.hap_var_flank = _ptr(self.hap_var_flank, capi.c_i32p)
        if self.mate is not None:
            m = self.mate
            b.read_mate_pos = _ptr(m["read_mate_pos"], capi.c_i32p)
            b.read_mate_len = _ptr(m["read_mate_len"], capi.c_i32p)
            b.read_lib = _ptr(m["read_lib"], capi.c_u8p)
            b.n_libs = len(m["lib_p95"])
            b.lib_off = _ptr(m["lib_off"], capi.c_i32p)
            b.lib_prob = _ptr(m["lib_prob"], capi.c_f64p)
            b.lib_p95 = _ptr(m["lib_p95"], capi.c_f64p)
        return b

    def slice_windows(self, w0, w1):
        """Contiguous window block [w0,w1) as its own batch (how ranks shard a job)."""
        a = self.a
        h0, h1 = int(a["win_hap_off"][w0]), int(a["win_hap_off"][w1])
        r0, r1 = int(a["win_read_off"][w0]), int(a["win_read_off"][w1])
        hs0, hs1 = int(a["hap_seq_off"][h0]), int(a["hap_seq_off"][h1])
        rs0, rs1 = int(a["read_seq_off"][r0]), int(a["read_seq_off"][r1])
        v0, v1 = int(a["hap_var_off"][h0]), int(a["hap_var_off"][h1])
        return PackedBatch(
            win_hap_off=a["win_hap_off"][w0:w1 + 1] - h0, win_read_off=a["win_read_off"][w0:w1 + 1] - r0,
            win_hap_start=a["win_hap_start"][w0:w1], hap_seq_off=a["hap_seq_off"][h0:h1 + 1] - hs0,
            hap_seq=a["hap_seq"][hs0:hs1], hap_var_off=a["hap_var_off"][h0:h1 + 1] - v0,
            hap_var=a["hap_var"][2 * v0:2 * v1], read_seq_off=a["read_seq_off"][r0:r1 + 1] - rs0,
            read_seq=a["read_seq"][rs0:rs1], read_qidx=a["read_qidx"][rs0:rs1], read_mqidx=a["read_mqidx"][r0:r1],
            read_start=a["read_start"][r0:r1], read_flags=a["read_flags"][r0:r1],
            qual_table=a["qual_table"], mapq_table=a["mapq_table"],
            hap_var_flank=None if self.hap_var_flank is None else self.hap_var_flank[3 * v0:3 * v1],
            mate=None if self.mate is None else dict(self.mate, read_mate_pos=self.mate["read_mate_pos"][r0:r1],
                                                     read_mate_len=self.mate["read_mate_len"][r0:r1], read_lib=self.mate["read_lib"][r0:r1]))


def pack(windows: Sequence[Window], libraries=None) -> PackedBatch:
    """Pack Window objects; dedups base/mapping quality doubles into the <=256-entry tables.
    libraries: optional list of (probs, p95) — Library::getProb table and getNinetyFifthPctProb() — enables the mate arrays."""
    qmap, mqmap = {}, {}
    win_hap_off, win_read_off, win_hap_start = [0], [0], []
    hap_seq_off, hap_seq, hap_var_off, hap_var = [0], [], [0], []
    hap_var_flank, any_flank = [], any(w.hap_var_flanks is not None for w in windows)
    read_seq_off, read_seq, read_qidx, read_mqidx, read_start, read_flags = [0], [], [], [], [], []
    mate_pos, mate_len, read_lib = [], [], []
    for w in windows:
        win_hap_start.append(w.hap_start & 0xFFFFFFFF)
        for hi, h in enumerate(w.haps):
            hap_seq.append(h.encode())
            hap_seq_off.append(hap_seq_off[-1] + len(h))
            vs = w.hap_vars[hi] if w.hap_vars is not None else []
            for s, e in vs:
                hap_var += [int(s), int(e)]
            fl = w.hap_var_flanks[hi] if w.hap_var_flanks is not None else [(0, 0, 0)] * len(vs)
            assert len(fl) == len(vs)
            for t in fl:
                hap_var_flank += [int(t[0]), int(t[1]), int(t[2])]
            hap_var_off.append(hap_var_off[-1] + len(vs))
        win_hap_off.append(win_hap_off[-1] + len(w.haps))
        for r in w.reads:
            if len(r.qual) != len(r.seq):
                raise ValueError("qual/seq length mismatch")
            read_seq.append(r.seq.encode())
            read_seq_off.append(read_seq_off[-1] + len(r.seq))
            for q in r.qual:
                q = float(q)
                if q not in qmap:
                    qmap[q] = len(qmap)
                read_qidx.append(qmap[q])
            mq = float(r.mapQual)
            if mq not in mqmap:
                mqmap[mq] = len(mqmap)
            read_mqidx.append(mqmap[mq])
            read_start.append(int(r.start) & 0xFFFFFFFF)
            read_flags.append((1 if r.unmapped else 0) | (2 if r.paired else 0) | (4 if r.mate_unmapped else 0) |
                              (8 if r.mate_reverse else 0) | (16 if r.mate_same_tid else 0))
            mate_pos.append(int(r.mate_pos)); mate_len.append(int(r.mate_len)); read_lib.append(int(r.lib))
        win_read_off.append(win_read_off[-1] + len(w.reads))
    if len(qmap) > 256 or len(mqmap) > 256:
        raise ValueError("more than 256 distinct base or mapping qualities in one batch")
    qt = np.zeros(max(len(qmap), 1)); mt = np.zeros(max(len(mqmap), 1))
    for q, i in qmap.items():
        qt[i] = q
    for q, i in mqmap.items():
        mt[i] = q
    return PackedBatch(
        win_hap_off=win_hap_off, win_read_off=win_read_off, win_hap_start=np.array(win_hap_start, dtype=np.uint32),
        hap_seq_off=hap_seq_off, hap_seq=np.frombuffer(b"".join(hap_seq), dtype=np.uint8),
        hap_var_off=hap_var_off, hap_var=np.array(hap_var, dtype=np.int32),
        read_seq_off=read_seq_off, read_seq=np.frombuffer(b"".join(read_seq), dtype=np.uint8),
        read_qidx=np.array(read_qidx, dtype=np.uint8), read_mqidx=np.array(read_mqidx, dtype=np.uint8),
        read_start=np.array(read_start, dtype=np.uint32), read_flags=np.array(read_flags, dtype=np.uint8),
        qual_table=qt, mapq_table=mt,
        hap_var_flank=np.array(hap_var_flank, dtype=np.int32) if any_flank else None,
        mate=None if libraries is None else dict(
            read_mate_pos=mate_pos, read_mate_len=mate_len, read_lib=read_lib,
            lib_off=np.concatenate([[0], np.cumsum([len(pr) for pr, _ in libraries])]),
            lib_prob=np.concatenate([np.asarray(pr, np.float64) for pr, _ in libraries]),
            lib_p95=[p95 for _, p95 in libraries]))


RESULT_DTYPES = dict(ll=np.float64, llOn=np.float64, llOff=np.float64, mLogBQ=np.float64, offHap=np.uint8,
                     offHapHMQ=np.uint8, numIndels=np.int16, numMismatch=np.int16, nBQT=np.int16, nmmBQT=np.int16,
                     nMMLeft=np.int16, nMMRight=np.int16, firstBase=np.int16, lastBase=np.int16, hpos=np.int16,
                     var_covered=np.uint8, status=np.int32, onHap=np.uint8, var_fcov=np.uint8)


def result_lengths(pb: PackedBatch):
    n = {k: pb.n_pairs for k in RESULT_DTYPES}
    n["hpos"] = pb.hpos_len
    n["var_covered"] = pb.var_cov_len
    n["var_fcov"] = pb.var_cov_len
    n["onHap"] = pb.n_reads
    return n


def alloc_result(pb: PackedBatch, fill=None):
    """Host result arrays + the dd_result that points at them."""
    n = result_lengths(pb)
    arrs = {}
    res = capi.dd_result()
    for k, typ in capi.RESULT_FIELDS:
        arr = np.zeros(max(n[k], 1), dtype=RESULT_DTYPES[k])
        if fill is not None:
            arr[...] = fill
        arrs[k] = arr
        setattr(res, k, arr.ctypes.data_as(typ))
    return arrs, res


def alloc_result_pinned(pb: PackedBatch, fill=0x55):
    """Like alloc_result, with every array in page-locked host memory (dd_host_alloc): the kernels then write the outputs in place.
    Returns (arrays, dd_result, release); call release() when the arrays are no longer used."""
    import ctypes as C
    lib = capi.load()
    n = result_lengths(pb)
    arrs, ptrs = {}, []
    res = capi.dd_result()
    for k, typ in capi.RESULT_FIELDS:
        dt = np.dtype(RESULT_DTYPES[k])
        cnt = max(n[k], 1)
        ptr = lib.dd_host_alloc(cnt * dt.itemsize)
        if not ptr:
            for q in ptrs:
                lib.dd_host_free(q)
            raise MemoryError("dd_host_alloc failed")
        ptrs.append(ptr)
        arrs[k] = np.frombuffer((C.c_char * (cnt * dt.itemsize)).from_address(ptr), dtype=dt)
        if fill is not None:
            arrs[k].view(np.uint8)[...] = fill
        setattr(res, k, C.cast(C.c_void_p(ptr), typ))

    def release():
        arrs.clear()
        while ptrs:
            lib.dd_host_free(ptrs.pop())
    return arrs, res, release


def pair_slices(pb: PackedBatch, w: int):
    """Index helpers for window w: (pair0, H, R, hpos0, SL, read_seq_off0)."""
    a = pb.a
    h0, h1 = int(a["win_hap_off"][w]), int(a["win_hap_off"][w + 1])
    r0, r1 = int(a["win_read_off"][w]), int(a["win_read_off"][w + 1])
    SL = int(a["read_seq_off"][r1]) - int(a["read_seq_off"][r0])
    return int(pb.win_pair_off[w]), h1 - h0, r1 - r0, int(pb.win_hpos_off[w]), SL, int(a["read_seq_off"][r0])
