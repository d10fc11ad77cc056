"""TEST INFRASTRUCTURE ONLY: DetInDel::getReads (reference DInDel.cpp:885-1262) restated in Python on plain record dicts (the ones
tests/_bamwriter.py writes), with bam_fetch as a brute-force overlap scan.  Used to check host/get_reads.cpp; not part of the product.

Only what the default libraries need is modelled: every read's library has maxInsertSize `max_insert` (Library.hpp defaults)."""
from tests import _bamwriter as bw

U32 = 1 << 32


def _end(r):
    cig = bw.parse_cigar(r["cigar"]) if r["cigar"] else []
    return r["pos"] + bw.ref_len(cig) if cig else r["pos"] + 1                      # Read::getEndPos, Read.hpp:185-188


def _flip(seq):
    comp = {"A": "T", "T": "A", "C": "G", "G": "C"}
    return "".join(comp.get(c, c) for c in reversed(seq))                           # Read::reverse + Read::complement, Read.hpp:205-227


class Fetcher:
    def __init__(self, records, max_reads=10000, max_read_length=500, min_read_overlap=20, map_unmapped=False, map_qual_threshold=0.99, max_insert=2000):
        # one chromosome; either one list of records sorted by pos, or one such list per BAM pool (the reference's myBams)
        self.pools = records if records and isinstance(records[0], list) else [records]
        self.p = dict(maxReads=max_reads, maxReadLength=max_read_length, minReadOverlap=min_read_overlap, mapUnmapped=map_unmapped, thr=map_qual_threshold)
        self.max_insert = max_insert
        self.buffer, self.old_left, self.old_right_fetch, self.reset = [], 0, 0, True

    def window_done(self, skipped, left):                                          # DInDel.cpp:1401-1408
        self.reset, self.old_left = skipped, left

    def fetch(self, beg, end):                                                     # bam_fetch + Read::fetchFuncVectorPooled (Read.hpp:388-412)
        beg, end = beg - U32 if beg >= 1 << 31 else beg, end - U32 if end >= 1 << 31 else end     # the int arguments of bam_fetch
        if beg >= end:
            return []
        # pool after pool into one vector (DInDel.cpp:981-993), each record remembering its pool (Read::poolID)
        return [(b, r) for b, records in enumerate(self.pools) for r in records
                if _end(r) > beg and r["pos"] < end and not (r["flag"] & (1024 | 512 | 2048))]

    def get_reads(self, left, right):
        p = self.p
        if right - left < 3 * p["minReadOverlap"]:
            raise ValueError("Choose a larger width or a smaller minReadOverlap.")
        max_dev = self.max_insert
        right_fetch = right_most = (right + max_dev) % U32
        left_fetch = left_most = (left - max_dev - 200) % U32                      # unsigned arithmetic, :923-927
        if self.reset:
            self.buffer = []
            self.old_right_fetch = right_fetch
        else:
            self.buffer = [(b, r) for b, r in self.buffer if not (r["pos"] % U32 < left_most)]
            if left_most < self.old_right_fetch:
                left_fetch = self.old_right_fetch
        if left_fetch <= right_fetch:
            new = self.fetch(left_fetch, right_fetch)
            if len(self.buffer) + len(new) > p["maxReads"] * 100:
                raise ValueError("Too many reads in region")
            self.old_right_fetch = right_fetch
            self.buffer += [(b, r) for b, r in new if r["pos"] % U32 >= left_fetch]
        count = {}
        for _b, r in self.buffer:
            count[r["qname"]] = count.get(r["qname"], 0) + 1
            if count[r["qname"]] > 2:
                raise ValueError("duplicate reads!")
        reads = [dict(rec=r, pool=b, qname=r["qname"], pos=r["pos"], size=len(r["seq"]), seq=r["seq"], mapQual=1.0 - 10.0 ** (-r["mapq"] / 10.0), matePos=r["mpos"], mateLen=-1,
                      unmapped=bool(r["flag"] & 4), mateUnmapped=bool(r["flag"] & 8), paired=bool(r["flag"] & 1), reverse=bool(r["flag"] & 16), end=_end(r))
                 for b, r in self.buffer]
        for x in reads:
            x["mapQual"] = min(max(x["mapQual"], 1e-16), 1.0 - 1e-16)              # Read.hpp:127-131
        mapped, unmapped = {}, {}
        for i, x in enumerate(reads):
            (unmapped if x["unmapped"] else mapped).setdefault(x["qname"], []).append(i)
        min_q = max(p["thr"], 0.0)
        for r, x in enumerate(reads):
            filt = x["size"] > p["maxReadLength"]
            if x["end"] % U32 < left_most or x["pos"] % U32 > right_most:
                filt = True
            if not x["unmapped"]:
                if x["pos"] + x["size"] < left + p["minReadOverlap"] or x["pos"] > right - p["minReadOverlap"]:
                    filt = True
                elif not x["mateUnmapped"]:
                    if x["rec"]["mtid"] != 0:                                      # the test chromosome is tid 0
                        pass
                    else:
                        filt = True
                        for idx in mapped[x["qname"]]:
                            if idx != r:
                                x["mateLen"], x["matePos"], filt = reads[idx]["size"], reads[idx]["pos"], False
                                if x["matePos"] != x["rec"]["mpos"]:
                                    raise ValueError("matepos inconsistency!")
                else:
                    x["matePos"] = x["pos"]
                    filt = True
                    for idx in unmapped.get(x["qname"], []):
                        if idx != r:
                            x["mateLen"], filt = reads[idx]["size"], False
            elif p["mapUnmapped"]:
                lst = mapped.get(x["qname"])
                if lst is None:
                    filt = True
                else:
                    if len(lst) != 1:
                        raise ValueError("UNMAPPED READ HAS MORE THAN ONE MATE!")
                    m = reads[lst[0]]
                    rl, rr = ((m["pos"] - self.max_insert) % U32, m["pos"] % U32) if m["reverse"] else (m["pos"] % U32, (m["pos"] + self.max_insert) % U32)
                    if rr > left and rl < right:
                        filt = False
                        x["mapQual"], x["matePos"], x["mateLen"] = m["mapQual"], m["pos"], m["size"]
                        if x["reverse"] == m["reverse"]:
                            x["seq"] = _flip(x["seq"])
                    else:
                        filt = True
            else:
                filt = True
            if filt:
                x["mapQual"] = -1.0
        reads.sort(key=lambda x: -x["mapQual"])                                    # the order inside a tie is std::sort's: compared as sets by the test
        out = []
        for x in reads[:p["maxReads"]]:
            if x["mapQual"] < min_q:
                break
            if x["matePos"] == -1 and x["paired"] and not x["mateUnmapped"]:
                x["matePos"] = x["pos"]
            out.append(x)
        if len(out) < 2:
            raise ValueError("too_few_reads")
        if len(out) >= p["maxReads"]:
            raise ValueError("above_read_count_threshold")
        return out


def run_windows(records, windows, with_buffer=False, **kw):
    """What the hook ddh_get_reads_json reports for consecutive windows: per window {"throw": msg} or the selected reads.
    with_buffer: (outcome, [(qname, pool) of the buffer in order]) per window, the selected reads carrying their pool."""
    f = Fetcher(records, **kw)
    res = []
    for left, right in windows:
        skipped = False
        try:
            got = f.get_reads(left, right)
            out = [(x["qname"], x["pos"], x["mapQual"], x["matePos"], x["mateLen"], int(x["unmapped"]), x["seq"]) + ((x["pool"],) if with_buffer else ()) for x in got]
        except ValueError as e:
            out = {"throw": str(e)}
            skipped = True
        res.append((out, [(r["qname"], b) for b, r in f.buffer]) if with_buffer else out)
        f.window_done(skipped, left)
    return res
