"""host/fast_inflate.cpp — the raw-DEFLATE decoder the BGZF reader tries before zlib — against zlib itself: streams made by zlib at every level
and strategy (stored, fixed and dynamic Huffman blocks, long and short matches, overlapping copies, long codes) decode to the same bytes;
damaged streams are declined or give bytes that the block's CRC would reject, and never touch memory outside the output buffer (the test runs
under ASan in tests/sanitize_cpu.sh)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from dindel_tgi_amd import hostlib


@pytest.fixture(scope="module")
def check():
    lib = hostlib.load()
    lib.ddh_fast_inflate_check.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    return lambda comp, want: lib.ddh_fast_inflate_check(comp, len(comp), want, len(want))


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    return c.compress(data) + c.flush()


def samples(rng):
    yield b""
    yield b"A"
    yield b"ACGT" * 16000                                                    # long matches at distance 4
    yield bytes(65280)                                                       # one byte repeated: distance 1, length 258 runs
    yield bytes(rng.integers(0, 256, 65280, dtype=np.uint8))                 # incompressible: stored blocks at level 0, literals elsewhere
    yield bytes(rng.integers(0, 4, 60000, dtype=np.uint8))                   # 2 bits of entropy per byte
    yield bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 65000)])
    text = b"".join(b"q%07d\t%d\t%dM\t%s\n" % (i, 5000 + 3 * i, 100, bytes(np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, 50)])) for i in range(700))
    yield text[:65280]
    skew = rng.choice(256, 65000, p=np.array([2.0 ** -(1 + i // 2) for i in range(256)]) / sum(2.0 ** -(1 + i // 2) for i in range(256)))
    yield bytes(skew.astype(np.uint8))                                       # very skewed alphabet: code lengths up to 15
    for n in (1, 2, 3, 7, 8, 9, 257, 258, 259, 1000, 32768, 32769, 65535, 65536):
        yield bytes(rng.integers(0, 3, n, dtype=np.uint8))


def test_streams_from_zlib_decode_to_the_same_bytes(check):
    rng = np.random.default_rng(12)
    n_ok = 0
    for data in samples(rng):
        for level in (0, 1, 2, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                comp = raw_deflate(data, level, strategy)
                got = check(comp, data)
                assert got == 1, (len(data), level, strategy)                                    # nothing zlib writes is declined or decoded differently
                n_ok += got
        for wbits, mem in ((-9, 1), (-12, 4)):                                                   # small windows, small hash: other block shapes
            assert check(raw_deflate(data, 6, zlib.Z_DEFAULT_STRATEGY, wbits, mem), data) == 1
    assert n_ok > 600


def test_several_blocks_and_sync_flushes(check):
    rng = np.random.default_rng(13)
    parts = [bytes(rng.integers(0, 5, int(rng.integers(1, 4000)), dtype=np.uint8)) for _ in range(40)]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b""
    for i, p in enumerate(parts):
        comp += c.compress(p)
        if i % 3 == 0:
            comp += c.flush(zlib.Z_SYNC_FLUSH)                                # empty stored blocks in between
        elif i % 3 == 1:
            comp += c.flush(zlib.Z_FULL_FLUSH)
    comp += c.flush()
    assert check(comp, b"".join(parts)) == 1


def test_damaged_streams_are_declined_or_wrong_but_harmless(check):
    rng = np.random.default_rng(14)
    data = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20000)]) + bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    comp = raw_deflate(data)
    assert check(comp, data) == 1
    assert check(comp[:-1], data) == 0 and check(comp[:len(comp) // 2], data) == 0             # truncated
    assert check(comp, data[:-1]) == 0 and check(comp, data + b"x") == 0                        # announces another length
    assert check(b"", data) == 0 and check(b"\x07", b"") == 0                                   # block type 3
    wrong = 0
    for _ in range(400):
        bad = bytearray(comp)
        for _k in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        r = check(bytes(bad), data)
        assert r in (-1, 0, 1)
        wrong += r == -1
    assert wrong > 0                                                                             # decoded to other bytes: what the CRC is for


def test_crc32_equals_zlib():
    """fastCrc32 (carry-less-multiplication folding, byte table for the tail and for CPUs without PCLMULQDQ) against zlib.crc32: every length
    around the 16- and 64-byte steps, unaligned starts, and continuation from a running value."""
    lib = hostlib.load()
    lib.ddh_fast_crc32.argtypes = [C.c_uint, C.c_char_p, C.c_int]
    lib.ddh_fast_crc32.restype = C.c_uint
    rng = np.random.default_rng(15)
    blob = bytes(rng.integers(0, 256, 70000, dtype=np.uint8))
    for n in list(range(0, 300)) + [511, 512, 513, 1023, 1024, 4095, 4096, 4097, 65279, 65280, 65535, 65536, 70000]:
        for off in (0, 1, 7):
            part = blob[off:off + n]
            assert lib.ddh_fast_crc32(0, part, len(part)) == zlib.crc32(part), (n, off)
    a, b = blob[:12345], blob[12345:40000]
    assert lib.ddh_fast_crc32(lib.ddh_fast_crc32(0, a, len(a)), b, len(b)) == zlib.crc32(a + b)
    assert lib.ddh_fast_crc32(0, bytes(1000), 1000) == zlib.crc32(bytes(1000)) and lib.ddh_fast_crc32(0, b"\xff" * 777, 777) == zlib.crc32(b"\xff" * 777)


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, v, n):                       # a field, least significant bit first (RFC 1951 3.1.1)
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def huff(self, code, n):                    # a Huffman code, most significant bit first
        for i in range(n - 1, -1, -1):
            self.bits((code >> i) & 1, 1)

    def done(self):
        if self.n:
            self.bits(0, 8 - self.n)
        return bytes(self.out)


def _one_code_distance_stream(dist_code_len):
    """A dynamic block whose distance tree has ONE code, of `dist_code_len` bits: 'A', a match (length 3, distance 1), end of block."""
    w = _Bits()
    w.bits(1, 1); w.bits(2, 2)                                  # BFINAL, BTYPE = dynamic
    w.bits(1, 5); w.bits(0, 5); w.bits(14, 4)                   # HLIT = 258 codes, HDIST = 1 code, HCLEN = 18 lengths
    cl = {18: 1, 2: 2, 1: 2}                                    # code-length alphabet: 18 -> '0', 1 -> '10', 2 -> '11'
    for sym in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1):
        w.bits(cl.get(sym, 0), 3)
    zeros = lambda n: (w.huff(0, 1), w.bits(n - 11, 7))
    one = lambda: w.huff(0b10, 2)
    two = lambda: w.huff(0b11, 2)
    zeros(65); one()                                            # literal 65 'A': 1 bit
    zeros(138); zeros(52)                                       # literals 66..255 unused
    two(); two()                                                # 256 (end of block) and 257 (length 3): 2 bits each
    (one if dist_code_len == 1 else two)()                      # the single distance code
    w.huff(0, 1)                                                # 'A'
    w.huff(0b11, 2); w.huff(0, dist_code_len)                   # length 3, distance code 0 (distance 1)
    w.huff(0b10, 2)                                             # end of block
    return w.done()


def test_one_code_distance_tree_only_with_a_one_bit_code(check):
    """RFC 1951 3.2.7 / zlib's inftrees (`left > 0 && max != 1` is an error): an incomplete distance tree is legal only as ONE code of ONE
    bit.  The own decoder takes exactly what zlib takes; the two-bit variant is declined (and would go to zlib, which rejects the block)."""
    ok, bad = _one_code_distance_stream(1), _one_code_distance_stream(2)
    assert zlib.decompress(ok, -15) == b"AAAA"
    with pytest.raises(zlib.error):
        zlib.decompress(bad, -15)
    assert check(ok, b"AAAA") == 1
    assert check(bad, b"AAAA") == 0
