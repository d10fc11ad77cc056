"""TEST INFRASTRUCTURE ONLY: a minimal BAM + BAI writer (SAM/BAM specification) so that tests can build their own alignment
files — there is no samtools in the image.  BGZF blocks are deliberately small so that records straddle block boundaries."""
import struct
import zlib

NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(s):
    out, n = [], ""
    for ch in s:
        if ch.isdigit():
            n += ch
        else:
            out.append((int(n), CIGAR_OPS.index(ch)))
            n = ""
    return out


def ref_len(cig):
    return sum(l for l, op in cig if op in (0, 2, 3, 7, 8))


def bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize) + c +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def encode_record(tid, r):
    """r: dict(qname, flag, pos, mapq, cigar (string, "" = none), seq, qual (list of Phred), mtid, mpos, isize, tags={'RG': 'x'})"""
    cig = parse_cigar(r.get("cigar", ""))
    seq = r["seq"]
    end = r["pos"] + (ref_len(cig) if cig else 1)
    b = struct.pack("<iiBBHHHiiii", tid, r["pos"], len(r["qname"]) + 1, r.get("mapq", 60), reg2bin(r["pos"], end), len(cig), r["flag"], len(seq),
                    r.get("mtid", -1), r.get("mpos", -1), r.get("isize", 0))
    b += r["qname"].encode() + b"\0"
    for l, op in cig:
        b += struct.pack("<I", (l << 4) | op)
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= NT16.index(ch) << (4 if i % 2 == 0 else 0)
    b += bytes(packed) + bytes(r["qual"])
    for k, v in r.get("tags", {}).items():                  # "text" = type Z; (type, value) for A c C s S i I f
        if isinstance(v, str):
            b += k.encode() + b"Z" + v.encode() + b"\0"
        else:
            t, x = v
            b += k.encode() + t.encode() + (x.encode() if t == "A" else struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[t], x))
    return struct.pack("<I", len(b)) + b, end


def write_bam(path, header_text, refs, records, block_bytes=3000):
    """refs: [(name, length)]; records: [(tid, dict)] sorted by (tid, pos).  Writes path and path + '.bai'."""
    u = bytearray(b"BAM\1" + struct.pack("<I", len(header_text)) + header_text.encode() + struct.pack("<I", len(refs)))
    for name, ln in refs:
        u += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    spans = []                                    # (tid, pos, end, ustart, uend)
    for tid, r in records:
        b, end = encode_record(tid, r)
        spans.append((tid, r["pos"], end, len(u), len(u) + len(b)))
        u += b
    ustart, cstart, out = [], [], bytearray()
    for o in range(0, len(u), block_bytes):
        ustart.append(o)
        cstart.append(len(out))
        out += bgzf_block(bytes(u[o:o + block_bytes]))
    eof_c = len(out)
    out += bgzf_block(b"")
    open(path, "wb").write(bytes(out))

    def voff(uo):
        if uo >= len(u):
            return eof_c << 16
        k = uo // block_bytes
        return (cstart[k] << 16) | (uo - ustart[k])
    bai = bytearray(b"BAI\1" + struct.pack("<I", len(refs)))
    for tid in range(len(refs)):
        bins, linear = {}, {}
        for (t, pos, end, us, ue) in spans:
            if t != tid:
                continue
            bins.setdefault(reg2bin(pos, end), []).append([voff(us), voff(ue)])
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                linear[w] = min(linear.get(w, 1 << 62), voff(us))
        bai += struct.pack("<I", len(bins))
        for b in sorted(bins):
            merged = []
            for ch in bins[b]:                    # consecutive records of a bin form one chunk
                if merged and merged[-1][1] == ch[0]:
                    merged[-1][1] = ch[1]
                else:
                    merged.append(ch)
            bai += struct.pack("<II", b, len(merged))
            for beg, end in merged:
                bai += struct.pack("<QQ", beg, end)
        n_intv = (max(linear) + 1) if linear else 0
        bai += struct.pack("<I", n_intv)
        last = 0
        for w in range(n_intv):
            last = linear.get(w, last)            # empty windows repeat the previous offset, as samtools does
            bai += struct.pack("<Q", last)
    open(path + ".bai", "wb").write(bytes(bai))


def read_bam(path):
    """The whole file decoded: (header text as stored, [(name, length)], [record dict with tid / bin / raw fields as well]).  BGZF is a
    series of gzip members, which the gzip module reads as one stream."""
    import gzip
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<I", d, 4)
    text = d[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref, = struct.unpack_from("<I", d, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<I", d, o)
        name = d[o + 4:o + 4 + l_name - 1].decode()
        ln, = struct.unpack_from("<I", d, o + 4 + l_name)
        refs.append((name, ln))
        o += 8 + l_name
    recs = []
    while o < len(d):
        size, = struct.unpack_from("<I", d, o)
        b = d[o + 4:o + 4 + size]
        o += 4 + size
        tid, pos, l_qname, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, isize = struct.unpack_from("<iiBBHHHiiii", b, 0)
        p = 32
        qname = b[p:p + l_qname - 1].decode()
        p += l_qname
        cig = struct.unpack_from("<%dI" % n_cig, b, p)
        p += 4 * n_cig
        seq = "".join(NT16[(b[p + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = list(b[p:p + l_seq])
        p += l_seq
        recs.append(dict(tid=tid, pos=pos, qname=qname, mapq=mapq, bin=bin_, flag=flag, mtid=mtid, mpos=mpos, isize=isize, seq=seq, qual=qual,
                         cigar="".join("%d%s" % (c >> 4, CIGAR_OPS[c & 15]) for c in cig), aux=bytes(b[p:]), raw=bytes(b)))
    return text, refs, recs
