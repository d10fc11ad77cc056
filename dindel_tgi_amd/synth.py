"""Deterministic synthetic realignment windows in the shape BASELINE.json / SURVEY.md §8(d) quote:

  window   = random `hap_len`-bp reference haplotype + (H-1) variants of it, each carrying one 1..3 bp
             insertion or deletion near the centre (what candidate-haplotype construction produces),
  reads    = R reads of L bp sampled from a random haplotype of the window at start offsets uniform in
             [-L/2, Hs-L/2) (so reads overhang both ends), bases outside the haplotype random,
             substitution rate 1e-3, all Phred `phred`, mapping quality Phred `mapq_phred`,
  hapStart = 1000, read start = 1000 + offset.

Numpy PCG64 with a fixed seed — reproducible on any host; no file or network input.
"""
import numpy as np

from .batch import PackedBatch, phred_to_prob

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def generate(n_windows, H=8, R=200, L=100, hap_len=120, seed=0x9E3779B9, max_indel=3, sub_rate=1e-3,
             phred=30, mapq_phred=40, hap_start=1000, vary_read_len=False, mixed_quals=False) -> PackedBatch:
    rng = np.random.Generator(np.random.PCG64(seed))
    hap_chunks, hap_lens, hap_var_off, hap_var, hap_var_flank = [], [], [0], [], []
    read_seq = np.empty(0, np.uint8)
    read_chunks, read_lens, read_start = [], [], []
    for _ in range(n_windows):
        ref = ACGT[rng.integers(0, 4, hap_len)]
        haps = [ref]
        hvars = [[]]
        hflank = [[]]
        for _h in range(1, H):
            ln = int(rng.integers(1, max_indel + 1))
            pos = int(rng.integers(hap_len // 2 - 10, hap_len // 2 + 10))
            if rng.random() < 0.5:   # deletion of ln bases at pos
                haps.append(np.concatenate([ref[:pos], ref[pos + ln:]]))
                hvars.append([(pos - 1, pos)])          # flanking bases in the haplotype
                hflank.append([(pos - 1, pos, 1)])      # (leftFlankRead, rightFlankRead, DEL)
            else:                    # insertion of ln random bases before pos
                ins = ACGT[rng.integers(0, 4, ln)]
                haps.append(np.concatenate([ref[:pos], ins, ref[pos:]]))
                hvars.append([(pos, pos + ln - 1)])
                hflank.append([(pos - 1, pos + ln, 2)])  # INS
        for h, v, f in zip(haps, hvars, hflank):
            hap_chunks.append(h)
            hap_lens.append(len(h))
            for s, e in v:
                hap_var += [s, e]
            for t in f:
                hap_var_flank += list(t)
            hap_var_off.append(hap_var_off[-1] + len(v))
        # reads (vectorised over the window's R reads)
        src = rng.integers(0, H, R)
        if vary_read_len:
            lens = rng.integers(max(8, L // 3), L + 1, R)
        else:
            lens = np.full(R, L)
        hl = np.array([len(h) for h in haps])
        hmat = np.zeros((H, int(hl.max())), np.uint8)
        for i, h in enumerate(haps):
            hmat[i, :len(h)] = h
        hlr = hl[src]
        off = np.floor(rng.random(R) * hlr).astype(np.int64) - lens // 2      # uniform in [-L/2, Hs-L/2)
        idx = off[:, None] + np.arange(L)[None, :]
        inside = (idx >= 0) & (idx < hlr[:, None])
        seq = ACGT[rng.integers(0, 4, (R, L))]
        gathered = hmat[src[:, None], np.clip(idx, 0, hmat.shape[1] - 1)]
        seq = np.where(inside, gathered, seq)
        sub = rng.random((R, L)) < sub_rate
        seq = np.where(sub, ACGT[rng.integers(0, 4, (R, L))], seq)
        keep = np.arange(L)[None, :] < lens[:, None]
        read_chunks.append(seq[keep])
        read_lens.append(lens)
        read_start.append(((hap_start + off) & 0xFFFFFFFF).astype(np.uint32))
    n_reads = n_windows * R
    read_seq = np.concatenate(read_chunks) if read_chunks else np.empty(0, np.uint8)
    if mixed_quals:
        phreds = np.arange(2, 42)
        qual_table = phred_to_prob(phreds)
        read_qidx = rng.integers(0, len(phreds), len(read_seq)).astype(np.uint8)
        mq_phreds = np.array([0, 3, 10, 20, 29, 37, 40, 60, 150])
        mapq_table = phred_to_prob(mq_phreds)
        read_mqidx = rng.integers(0, len(mq_phreds), n_reads).astype(np.uint8)
    else:
        qual_table = phred_to_prob([phred])
        read_qidx = np.zeros(len(read_seq), np.uint8)
        mapq_table = phred_to_prob([mapq_phred])
        read_mqidx = np.zeros(n_reads, np.uint8)
    z = np.zeros(1, np.int64)
    return PackedBatch(
        win_hap_off=np.arange(n_windows + 1) * H, win_read_off=np.arange(n_windows + 1) * R,
        win_hap_start=np.full(n_windows, hap_start, np.uint32),
        hap_seq_off=np.concatenate([z, np.cumsum(hap_lens)]), hap_seq=np.concatenate(hap_chunks) if hap_chunks else np.empty(0, np.uint8),
        hap_var_off=hap_var_off, hap_var=np.array(hap_var, np.int32),
        read_seq_off=np.concatenate([z, np.cumsum(np.concatenate(read_lens))]) if read_lens else z, read_seq=read_seq, read_qidx=read_qidx,
        read_mqidx=read_mqidx, read_start=np.concatenate(read_start) if read_start else np.empty(0, np.uint32), read_flags=np.zeros(n_reads, np.uint8),
        qual_table=qual_table, mapq_table=mapq_table, hap_var_flank=np.array(hap_var_flank, np.int32))


def tile(pb: PackedBatch, reps: int) -> PackedBatch:
    """Repeat a batch `reps` times (distinct windows with identical content) — lets bench.py reach the
    10k-window configuration without spending minutes in the Python generator."""
    a = pb.a
    def rep_off(off):
        off = off.astype(np.int64)
        body = off[1:] - off[0]
        step = int(off[-1] - off[0])
        return np.concatenate([np.zeros(1, np.int64)] + [body + k * step for k in range(reps)])
    return PackedBatch(
        win_hap_off=rep_off(a["win_hap_off"]), win_read_off=rep_off(a["win_read_off"]),
        win_hap_start=np.tile(a["win_hap_start"], reps), hap_seq_off=rep_off(a["hap_seq_off"]),
        hap_seq=np.tile(a["hap_seq"], reps), hap_var_off=rep_off(a["hap_var_off"]), hap_var=np.tile(a["hap_var"], reps),
        read_seq_off=rep_off(a["read_seq_off"]), read_seq=np.tile(a["read_seq"], reps),
        read_qidx=np.tile(a["read_qidx"], reps), read_mqidx=np.tile(a["read_mqidx"], reps),
        read_start=np.tile(a["read_start"], reps), read_flags=np.tile(a["read_flags"], reps),
        qual_table=a["qual_table"], mapq_table=a["mapq_table"],
        hap_var_flank=None if pb.hap_var_flank is None else np.tile(pb.hap_var_flank, reps))


def concat(parts) -> PackedBatch:
    """Several batches (same quality / mapping-quality tables) as one: the windows of parts[0], then parts[1], ..."""
    offs = dict(win_hap_off=0, win_read_off=0, hap_seq_off=0, hap_var_off=0, read_seq_off=0)
    out = {k: [] for k in PackedBatch.FIELDS}
    for k in offs:
        out[k].append(np.zeros(1, np.int64))
    flank = []
    for p in parts:
        for k in PackedBatch.FIELDS:
            v = p.a[k]
            if k in offs:
                out[k].append(v[1:].astype(np.int64) + offs[k])
                offs[k] += int(v[-1])
            elif k in ("qual_table", "mapq_table"):
                out[k] = [v]
            else:
                out[k].append(v)
        flank.append(p.hap_var_flank if p.hap_var_flank is not None else np.empty(0, np.int32))
    return PackedBatch(hap_var_flank=np.concatenate(flank), **{k: np.concatenate(v) for k, v in out.items()})


def generate_ragged(n_windows, seed=0x5EED4, max_reads=400, max_extra=40, max_indel=12, fix_reads=0, fix_read_len=0, fix_haps=0, trimmed=True, trimmed_every=5,
                    extra_mean=9.0) -> PackedBatch:
    """Windows in the shapes the reference's own pipeline produces (fixed seed; bench.py's `ragged` leg, tests):

    * reference haplotype = [minRef - 60, maxRef + 60] around the window's candidates (python/makeWindows.py:72-75): 121 bp for a
      single-position candidate, longer when the candidates span a deletion or are clustered — here 121 + extra, extra = 0 for
      45 % of the windows, else geometric (mean 9) capped at 40;
    * 2..12 candidate haplotypes per window (getHaplotypes, DInDel.cpp:1526-1645), each the reference with one insertion or
      deletion of 1..max_indel bases (max_indel drawn per window from 1..12), so a window's haplotypes have DIFFERENT lengths;
    * 20..max_reads reads per window (log-uniform), one read length per window out of 36 / 76 / 100 / 150 bp (every fifth
      window with trimmed reads of mixed lengths), Phred 2..41 base qualities, nine mapping qualities.
    max_extra / max_indel narrow the haplotype lengths (diagnostics: max_extra=0, max_indel=5 keeps every haplotype on one lane tiling).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    for i in range(n_windows):
        extra = 0 if rng.random() < 0.45 else int(min(max_extra, rng.geometric(1.0 / extra_mean)))
        H = int(rng.integers(2, 13))
        R = int(np.exp(rng.uniform(np.log(20.0), np.log(float(max_reads)))))
        L = int(rng.choice([36, 76, 100, 150], p=[0.1, 0.25, 0.45, 0.2]))
        if fix_reads: R = fix_reads                    # diagnostics (tools/ragged_factors.py): which kind of raggedness costs what
        if fix_read_len: L = fix_read_len
        if fix_haps: H = fix_haps
        parts.append(generate(1, H=H, R=R, L=L, hap_len=121 + extra, seed=int(rng.integers(1, 2 ** 31)), max_indel=int(rng.integers(1, max_indel + 1)),
                              vary_read_len=(trimmed and i % trimmed_every == trimmed_every - 1), mixed_quals=True))
    return concat(parts)
