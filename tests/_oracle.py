"""ctypes wrapper of oracle/libdd_oracle.so — the CPU restatement used ONLY as the checker in tests,
smoke() and bench.py's cpu_baseline leg (see oracle/dd_oracle.c header)."""
import ctypes as C
import os
import subprocess

import numpy as np

from dindel_tgi_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.environ.get("DD_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libdd_oracle.so")
MAXV = 1056


class ddo_out(C.Structure):
    _fields_ = ([("ll", C.c_double), ("llOn", C.c_double), ("llOff", C.c_double), ("mLogBQ", C.c_double)] +
                [(n, C.c_int32) for n in ("offHap offHapHMQ numIndels numMismatch nBQT nmmBQT nMMLeft nMMRight "
                                          "firstBase lastBase bMid status").split()] +
                [("n_indel", C.c_int32), ("indel_pos", C.c_int32 * MAXV), ("indel_len", C.c_int32 * MAXV),
                 ("indel_rpos", C.c_int32 * MAXV), ("n_snp", C.c_int32), ("snp_pos", C.c_int32 * MAXV),
                 ("snp_rpos", C.c_int32 * MAXV)])


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "dd_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(LIB)
        _lib.ddo_pair.argtypes = [C.c_char_p, C.c_int, C.c_char_p, capi.c_f64p, C.c_int, C.c_double, C.c_uint32,
                                  C.c_uint32, C.c_int, C.POINTER(capi.dd_params), C.POINTER(ddo_out),
                                  C.POINTER(C.c_int)]
        _lib.ddo_pair_fast.argtypes = [C.c_char_p, C.c_int, C.c_char_p, capi.c_f64p, C.c_int, C.c_double, C.c_uint32,
                                       C.c_uint32, C.POINTER(capi.dd_params), C.POINTER(ddo_out), C.POINTER(C.c_int)]
        _lib.ddo_pair_fbmax.argtypes = _lib.ddo_pair.argtypes
        _lib.ddo_pair_sums.argtypes = [C.POINTER(capi.dd_batch), capi.c_f64p, capi.c_f64p]
        _lib.ddo_batch.argtypes = [C.POINTER(capi.dd_params), C.POINTER(capi.dd_batch), C.POINTER(capi.dd_result),
                                   C.c_int, C.c_int64, C.c_int64]
        _lib.ddo_batch_fast.argtypes = _lib.ddo_batch.argtypes
    return _lib


def pair(hap, read, qual, mapQual, read_start, hap_start, params, unmapped=False):
    """One (haplotype, read) pair -> (ddo_out, hpos list)."""
    lib = load()
    L = len(read)
    q = np.ascontiguousarray(np.broadcast_to(np.asarray(qual, dtype=np.float64), (L,)))
    hp = (C.c_int * max(L, 1))()
    o = ddo_out()
    lib.ddo_pair(hap.encode(), len(hap), read.encode(), q.ctypes.data_as(capi.c_f64p), L, float(mapQual),
                 read_start & 0xFFFFFFFF, hap_start & 0xFFFFFFFF, 1 if unmapped else 0, C.byref(params),
                 C.byref(o), hp)
    return o, list(hp)[:L]


def pair_fbmax(hap, read, qual, mapQual, read_start, hap_start, params, unmapped=False):
    """Sibling model ObservationModelFBMax (KAT cross-check only) -> (ddo_out, hpos list)."""
    lib = load()
    L = len(read)
    q = np.ascontiguousarray(np.broadcast_to(np.asarray(qual, dtype=np.float64), (L,)))
    hp = (C.c_int * max(L, 1))()
    o = ddo_out()
    lib.ddo_pair_fbmax(hap.encode(), len(hap), read.encode(), q.ctypes.data_as(capi.c_f64p), L, float(mapQual),
                       read_start & 0xFFFFFFFF, hap_start & 0xFFFFFFFF, 1 if unmapped else 0, C.byref(params),
                       C.byref(o), hp)
    return o, list(hp)[:L]


def pair_fast(hap, read, qual, mapQual, read_start, hap_start, params):
    """ObservationModelS (--faster) for one pair -> (ddo_out, hpos list)."""
    lib = load()
    L = len(read)
    q = np.ascontiguousarray(np.broadcast_to(np.asarray(qual, dtype=np.float64), (L,)))
    hp = (C.c_int * max(L, 1))()
    o = ddo_out()
    lib.ddo_pair_fast(hap.encode(), len(hap), read.encode(), q.ctypes.data_as(capi.c_f64p), L, float(mapQual),
                      read_start & 0xFFFFFFFF, hap_start & 0xFFFFFFFF, C.byref(params), C.byref(o), hp)
    return o, list(hp)[:L]


def batch(params, pb, nthreads=1, first_window=0, n_win=-1, faster=False):
    """Whole PackedBatch through the oracle -> dict of numpy result arrays (same layout as the product).
    faster=True: the ObservationModelS restatement (computeLikelihoodsFaster)."""
    from dindel_tgi_amd.batch import alloc_result
    lib = load()
    arrs, res = alloc_result(pb)
    b = pb.ctypes_batch()
    fn = lib.ddo_batch_fast if faster else lib.ddo_batch
    rc = fn(C.byref(params), C.byref(b), C.byref(res), nthreads, first_window, n_win)
    assert rc == 0
    return arrs


def keyed_hpos(o, hpos):
    """hpos of a per-pair call (the reference's codes: every inserted base -1) -> the product's coding, in which an
    inserted base carries the key of its insertion (DD_HPOS_INS_KEY0 - pos, include/dindel_hmm.h)."""
    out = list(hpos)
    assert o.n_indel < MAXV
    for i in range(o.n_indel):
        if o.indel_len[i] > 0:
            for b in range(o.indel_rpos[i], o.indel_rpos[i] + o.indel_len[i]):
                assert out[b] == capi.DD_HPOS_INS
                out[b] = capi.DD_HPOS_INS_KEY0 - o.indel_pos[i]
    assert capi.DD_HPOS_INS not in out
    return out


def expected_variants(o, hap, read):
    """ml.indels / ml.snps as reportVariants builds them (ObservationModelFB.cpp:1375-1453, Faster.cpp:606-661) from the
    oracle's variant lists: ({pos: [string, startHap, endHap, startRead, endRead]}, {pos: string}); std::map semantics, a
    later variant at the same key replaces the earlier one."""
    assert o.n_indel < MAXV and o.n_snp < MAXV
    indels, snps = {}, {}
    for i in range(o.n_indel):
        pos, ln, rp = o.indel_pos[i], o.indel_len[i], o.indel_rpos[i]
        if ln > 0:
            indels[pos] = ["+" + read[rp:rp + ln], pos, pos, rp, rp + ln - 1]
        else:
            indels[pos] = ["-" + hap[pos:pos - ln], pos, pos - ln - 1, rp, rp + 1]
    for i in range(o.n_snp):
        snps[o.snp_pos[i]] = hap[o.snp_pos[i]] + "=>" + read[o.snp_rpos[i]]
    return indels, snps


def assert_record_variants(ml, o, hap, read, ctx=None):
    """The indel / SNP maps of a JSON MLAlignment record (tests/_host.py) against the oracle: keys, strings, coordinates."""
    indels, snps = expected_variants(o, hap, read)
    assert {i[0]: i[1:] for i in ml["indels"]} == indels, (ctx, ml["indels"], indels)
    assert {s[0]: s[1] for s in ml["snps"]} == snps, (ctx, ml["snps"], snps)


def expected_align(o, hap, read):
    """ml.align (ObservationModelFB.cpp:1355, :1428, :1443): 'R' per haplotype base, the read base where a SNP sits, 'D'
    over deleted bases (clipped to the haplotype: a deletion that reaches RO would write past the string's end)."""
    al = ["R"] * len(hap)
    ev = [(o.snp_rpos[i], 0, i) for i in range(o.n_snp)] + [(o.indel_rpos[i], 1, i) for i in range(o.n_indel) if o.indel_len[i] < 0]
    for _rp, kind, i in sorted(ev):          # in read order; within one base the SNP is written before the deletion
        if kind == 0:
            al[o.snp_pos[i]] = read[o.snp_rpos[i]]
        else:
            for y in range(o.indel_pos[i], min(o.indel_pos[i] - o.indel_len[i], len(hap))):
                al[y] = "D"
    return "".join(al)
