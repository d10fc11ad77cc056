"""Device-resident batches: torch owns HBM and the stream, the C ABI (dd_launch_device) does the work.

PyTorch is plumbing here (allocation, streams, torch.distributed); no torch op touches the data path.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .batch import PackedBatch, RESULT_DTYPES, result_lengths

_TORCH_DT = {np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8, np.dtype(np.int16): torch.int16,
             np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}


def _to_dev(arr, device):
    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint32:                      # torch has no uint32 arithmetic; ship the bits as int32
        a = a.view(np.int32)
    if a.size == 0:
        a = np.zeros(1, a.dtype)
    if not a.flags.writeable:
        a = a.copy()
    return torch.from_numpy(a).to(device, non_blocking=False)


class DeviceBatch:
    """A PackedBatch resident in HBM + the dd_device_batch that points at it."""

    def __init__(self, pb: PackedBatch, params: capi.dd_params, device="cuda:0"):
        lib = capi.load()
        self.pb, self.params, self.device = pb, params, torch.device(device)
        a = pb.a
        t = {}
        for k in ["win_hap_off", "win_read_off", "win_hap_start", "hap_seq_off", "hap_seq", "hap_var_off", "hap_var",
                  "read_seq_off", "read_seq", "read_qidx", "read_mqidx", "read_start", "read_flags"]:
            t[k] = _to_dev(a[k], self.device)
        hb = pb.ctypes_batch()
        hap_window = np.zeros(max(pb.n_haps, 1), np.int32)
        po = np.zeros(pb.n_windows + 1, np.int64); ho = np.zeros_like(po); vo = np.zeros_like(po)
        rc = lib.dd_build_index(C.byref(hb), hap_window.ctypes.data_as(capi.c_i32p), po.ctypes.data_as(capi.c_i64p),
                                ho.ctypes.data_as(capi.c_i64p), vo.ctypes.data_as(capi.c_i64p))
        if rc != 0:
            raise RuntimeError("dd_build_index: " + capi.last_error())
        tables = np.zeros(capi.DD_TABLE_DOUBLES, np.float64)
        rc = lib.dd_build_tables(C.byref(params), hb.qual_table, hb.n_qual, hb.mapq_table, hb.n_mapq,
                                 tables.ctypes.data_as(capi.c_f64p))
        if rc < 0:
            raise RuntimeError("dd_build_tables: " + capi.last_error())
        t["hap_window"] = _to_dev(hap_window, self.device)
        t["win_pair_off"] = _to_dev(po, self.device)
        t["win_hpos_off"] = _to_dev(ho, self.device)
        t["win_varcov_off"] = _to_dev(vo, self.device)
        t["tables"] = _to_dev(tables, self.device)
        lut = np.zeros(256, np.uint8)
        if lib.dd_build_symbol_lut(C.byref(hb), lut.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
            raise RuntimeError("dd_build_symbol_lut: " + capi.last_error())
        t["sym_lut"] = _to_dev(lut, self.device)
        if pb.mate is not None and params.mapUnmappedReads:
            lp = np.zeros(max(len(pb.mate["lib_prob"]), 1)); l95 = np.zeros(len(pb.mate["lib_p95"]))
            if lib.dd_build_library_tables(C.byref(hb), lp.ctypes.data_as(capi.c_f64p), l95.ctypes.data_as(capi.c_f64p)) != 0:
                raise RuntimeError("dd_build_library_tables: " + capi.last_error())
            for k in ("read_mate_pos", "read_mate_len", "read_lib", "lib_off"):
                t[k] = _to_dev(pb.mate[k], self.device)
            t["lib_logprob"] = _to_dev(lp, self.device)
            t["lib_log95"] = _to_dev(l95, self.device)
        if pb.hap_var_flank is not None and len(pb.hap_var_flank):
            t["hap_var_flank"] = _to_dev(pb.hap_var_flank, self.device)
        # ragged batches: per-class launch plans (haplotype length x read length), as the host-pointer path does by itself
        # windows outside the kernel limits are marked (DD_PAIR_UNSUPPORTED), not computed: flags + the maxima of the rest
        skip = np.zeros(max(pb.n_windows, 1), np.uint8)
        ok_max = (C.c_int32 * 2)()
        self.n_skipped = lib.dd_screen_windows(C.byref(hb), skip.ctypes.data_as(capi.c_u8p), C.byref(ok_max))
        if self.n_skipped < 0:
            raise RuntimeError("dd_screen_windows: " + capi.last_error())
        if self.n_skipped:
            t["win_skip"] = _to_dev(skip, self.device)
        self.classes = capi.dd_length_classes()
        hcl = np.zeros(max(pb.n_haps, 1) * capi.N_READ_CLASSES, np.int32)
        if lib.dd_build_length_classes(C.byref(hb), skip.ctypes.data_as(capi.c_u8p) if self.n_skipped else None, C.byref(params),
                                       hcl.ctypes.data_as(capi.c_i32p), C.byref(self.classes)) != 0:
            raise RuntimeError("dd_build_length_classes: " + capi.last_error())
        t["hap_class_list"] = _to_dev(hcl[:max(self.classes.list_len, 1)], self.device)
        self.t = t
        db = capi.dd_device_batch()
        db.n_windows, db.n_haps, db.n_reads = pb.n_windows, pb.n_haps, pb.n_reads
        db.max_hap_len, db.max_read_len = max(int(ok_max[0]), 1), max(int(ok_max[1]), 1)
        nr = np.diff(a["win_read_off"])
        db.max_window_reads = int(nr[skip[:pb.n_windows] == 0].max()) if pb.n_windows and (skip[:pb.n_windows] == 0).any() else 0
        for k, v in t.items():
            setattr(db, k, v.data_ptr())
        db.n_qual, db.n_mapq = hb.n_qual, hb.n_mapq
        db.classes = C.addressof(self.classes)
        self.db = db
        # results
        n = result_lengths(pb)
        self.out = {k: torch.zeros(max(n[k], 1), dtype=_TORCH_DT[np.dtype(RESULT_DTYPES[k])], device=self.device)
                    for k, _ in capi.RESULT_FIELDS}
        dr = capi.dd_device_result()
        for k, _ in capi.RESULT_FIELDS:
            setattr(dr, k, self.out[k].data_ptr())
        self.dr = dr
        self._n = n
        # device scratch the launch needs for this shape (back-pointer tiles in HBM for long reads); 0 if none
        self.ws_bytes = int(lib.dd_workspace_bytes(C.byref(params), C.byref(db)))
        self.ws = torch.empty(max(self.ws_bytes, 8), dtype=torch.uint8, device=self.device)

    def launch(self, stream=None):
        """Enqueue the path on `stream` (default: torch's current stream on this device). Asynchronous."""
        lib = capi.load()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        rc = lib.dd_launch_device(C.byref(self.params), C.byref(self.db), C.byref(self.dr),
                                  C.c_void_p(self.ws.data_ptr()), self.ws_bytes, C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("dd_launch_device rc=%d: %s" % (rc, capi.last_error()))

    def launch_faster(self, stream=None):
        """The --faster model (ObservationModelS) on the same resident batch. Asynchronous."""
        lib = capi.load()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        rc = lib.dd_launch_device_faster(C.byref(self.params), C.byref(self.db), C.byref(self.dr), C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError("dd_launch_device_faster rc=%d: %s" % (rc, capi.last_error()))

    def results(self):
        """Host copies (numpy) of the outputs, trimmed to their logical lengths."""
        torch.cuda.synchronize(self.device)
        return {k: self.out[k][:self._n[k]].cpu().numpy() for k, _ in capi.RESULT_FIELDS}
