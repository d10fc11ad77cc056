"""N4 (second half): dindel::getCIGAR (reference DInDel.cpp:728-882) — CPU only.

The C++ function against a Python restatement of the reference's loop on random alignments (including the code pairs the
reference's if-chain silently ignores), plus hand-checked cases."""
import ctypes as C

import numpy as np
import pytest

from tests import _host

M, I, D, S = 0, 1, 2, 4
INS, LO, RO = -1, -3, -4
MSGS = ["Haplotype has not been aligned!", "Read is not properly aligned!", "Error(1)!", "Error(2)!", "Error(3)!", "Error(4)!",
        "How is this possible? (1)"]


def cpp_cigar(hap_ref_pos, hpos, ref_start):
    lib = _host.load()
    hr = np.ascontiguousarray(hap_ref_pos, np.int32)
    hp = np.ascontiguousarray(hpos, np.int16)
    out = np.zeros(4 * len(hp) + 8, np.int32)
    rp = C.c_int(0)
    n = lib.ddh_get_cigar(hr.ctypes.data_as(C.POINTER(C.c_int)), len(hr), hp.ctypes.data_as(C.POINTER(C.c_short)), len(hp),
                          ref_start, out.ctypes.data_as(C.POINTER(C.c_int)), len(out), C.byref(rp))
    if n < 0:
        return MSGS[-n - 1]
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)], rp.value


def py_cigar(hap_ref_pos, hpos, ref_start):
    """DInDel.cpp:744-876, line by line."""
    npos = [hap_ref_pos[h] if h >= 0 else h for h in hpos]
    n = len(hpos)
    cig = []
    b = n - 1
    while b >= 0 and npos[b] < 0:
        b -= 1
    last = b
    if last < 0:
        return [(S, n)], -1
    b = 0
    while npos[b] < 0:
        b += 1
    if b > 0:
        cig.append((S, b))
    prev = npos[b]
    ref_pos = ref_start + prev
    op, ln = M, 1
    while b < last:
        chp, nhp = npos[b], npos[b + 1]
        if nhp == INS:
            if chp == INS:
                if op != I:
                    return "Error(1)!"
                ln += 1
            elif chp >= 0:
                if op != M:
                    return "Error(2)!"
                cig.append((M, ln)); ln = 1; op = I; prev = chp
            else:
                return "How is this possible? (1)"
        elif chp >= 0 and nhp >= 0 and nhp - chp == 1:
            if op != M:
                return "Error(3)!"
            ln += 1; prev = nhp
        elif chp >= 0 and nhp >= 0 and nhp - chp > 1:
            if op != M:
                return "Error(4)!"
            cig.append((M, ln)); cig.append((D, nhp - chp - 1)); op = M; ln = 1; prev = nhp
        elif chp == INS and nhp - prev == 1:
            cig.append((I, ln)); op = M; ln = 1; prev = nhp
        elif chp == INS and nhp - prev > 1:
            cig.append((I, ln)); cig.append((D, nhp - prev - 1)); op = M; ln = 1; prev = nhp
        b += 1
    cig.append((op, ln))
    if n - 1 - last > 0:
        cig.append((S, n - 1 - last))
    return cig, ref_pos


def test_hand_checked_cases():
    ident = list(range(100))                                   # haplotype == reference
    assert cpp_cigar(ident, list(range(10, 40)), 5000) == ([(M, 30)], 5010)
    assert cpp_cigar(ident, [LO] * 3 + list(range(0, 20)) + [RO] * 2, 7) == ([(S, 3), (M, 20), (S, 2)], 7)
    assert cpp_cigar(ident, list(range(10, 20)) + [INS, INS] + list(range(20, 30)), 0) == ([(M, 10), (I, 2), (M, 10)], 10)
    assert cpp_cigar(ident, list(range(10, 20)) + list(range(23, 33)), 0) == ([(M, 10), (D, 3), (M, 10)], 10)
    assert cpp_cigar(ident, list(range(10, 20)) + [INS] + list(range(22, 30)), 0) == ([(M, 10), (I, 1), (D, 2), (M, 8)], 10)
    assert cpp_cigar(ident, [LO] * 12, 0) == ([(S, 12)], -1)
    # a haplotype carrying a 3-bp deletion w.r.t. the reference: its bases 50.. sit on reference 53..
    hap_del = list(range(50)) + list(range(53, 103))
    assert cpp_cigar(hap_del, list(range(40, 60)), 1000) == ([(M, 10), (D, 3), (M, 10)], 1040)
    # a haplotype carrying a 2-bp insertion: its bases 50,51 have no reference position
    hap_ins = list(range(50)) + [INS, INS] + list(range(50, 98))
    assert cpp_cigar(hap_ins, list(range(45, 60)), 0) == ([(M, 5), (I, 2), (M, 8)], 45)
    assert cpp_cigar(ident, list(range(5)) + [7], 0)[0] == [(M, 5), (D, 2), (M, 1)]
    assert cpp_cigar(ident, [INS] + list(range(5)), 0) == ([(S, 1), (M, 5)], 0)        # leading insertion is clipped (:788)


def test_random_alignments_match_the_reference_loop():
    rng = np.random.default_rng(4)
    n_err = 0
    for trial in range(3000):
        hl = int(rng.integers(5, 80))
        hap_ref, pos = [], int(rng.integers(0, 5))
        for _ in range(hl):                                    # haplotype-to-reference map with indels
            u = rng.random()
            if u < 0.06:
                hap_ref.append(INS)
            else:
                pos += 1 if u < 0.94 else int(rng.integers(2, 6))
                hap_ref.append(pos)
        L = int(rng.integers(1, 60))
        hpos, h = [], int(rng.integers(-3, hl))
        for _ in range(L):
            u = rng.random()
            if h < 0 or u < 0.03:
                hpos.append(LO if len(hpos) < L // 2 else RO)
            elif u < 0.10:
                hpos.append(INS)
            else:
                hpos.append(min(h, hl - 1))
            h += 1 if rng.random() < 0.9 else int(rng.integers(-1, 5))     # also backwards / repeated positions
        want = py_cigar(hap_ref, hpos, 1000)
        got = cpp_cigar(hap_ref, hpos, 1000)
        assert got == want, (hap_ref, hpos, got, want)
        n_err += isinstance(want, str)
    assert 0 < n_err < 3000
