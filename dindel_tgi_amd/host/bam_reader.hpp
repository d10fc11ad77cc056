// bam_reader.hpp — a self-contained BGZF + BAM + BAI reader.
//
// The reference reads alignments through samtools-0.1.x's libbam (bam_open / bam_header_read / bam_index_load / bam_fetch,
// reference MyBam.hpp:74-88, DInDel.cpp:989-993), which is not in the image; this file restates the published formats
// (SAM/BAM specification: BGZF blocks, BAM records, the BAI binning index) and bam_fetch's traversal — bins overlapping
// the region, chunks below the linear index's minimum offset dropped, chunks sorted and merged, records filtered by
// overlap — so that a caller sees the same records in the same order.  zlib is the only dependency.
#ifndef DINDEL_BAM_READER_HPP
#define DINDEL_BAM_READER_HPP
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace dindel {

// BAM flag bits (SAM specification; bam.h BAM_F*)
enum { BAM_FPAIRED = 1, BAM_FPROPER_PAIR = 2, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16, BAM_FMREVERSE = 32, BAM_FREAD1 = 64,
       BAM_FREAD2 = 128, BAM_FSECONDARY = 256, BAM_FQCFAIL = 512, BAM_FDUP = 1024, BAM_FSUPPLEMENTARY = 2048 };
// CIGAR operations
enum { BAM_CMATCH = 0, BAM_CINS = 1, BAM_CDEL = 2, BAM_CREF_SKIP = 3, BAM_CSOFT_CLIP = 4, BAM_CHARD_CLIP = 5, BAM_CPAD = 6, BAM_CEQUAL = 7, BAM_CDIFF = 8 };

struct BamRecord {                       // bam1_t / bam1_core_t with the variable part decoded
    int32_t tid, pos; uint16_t bin; uint8_t qual; uint16_t flag; int32_t l_qseq, mtid, mpos, isize;
    std::string qname;
    std::vector<uint32_t> cigar;         // len << 4 | op
    std::string seq;                     // bam_nt16_rev_table letters ("=ACMGRSVTWYHKDBN")
    std::vector<uint8_t> qualities;      // Phred, one per base
    std::vector<uint8_t> aux;            // raw auxiliary fields
    uint32_t calend() const;             // bam_calend: pos + reference bases the CIGAR consumes (M, D, N, =, X)
    uint32_t endPos() const { return cigar.empty() ? uint32_t(pos + 1) : calend(); }        // Read::getEndPos, Read.hpp:185-188
    const char *auxString(const char tag[2]) const;      // value of a Z-type field, NULL if absent (bam_aux_get)
};

// A BAM record's bytes as the file holds them (without the 4-byte length word): the fixed fields read in place, the variable
// part located, nothing copied.  `d` must hold a record that decodeCore accepted (lengths consistent with `n`).
struct RawBamView {
    const uint8_t *d; size_t n;
    RawBamView(const uint8_t *bytes, size_t len) : d(bytes), n(len) {}
    static uint32_t u32(const uint8_t *p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
    int32_t tid() const { return int32_t(u32(d)); }
    int32_t pos() const { return int32_t(u32(d + 4)); }
    uint32_t nameBytes() const { return d[8]; }                          // with the terminating NUL
    uint8_t mapq() const { return d[9]; }
    uint32_t nCigar() const { return uint32_t(d[12]) | (uint32_t(d[13]) << 8); }
    uint16_t flag() const { return uint16_t(d[14] | (d[15] << 8)); }
    int32_t lSeq() const { return int32_t(u32(d + 16)); }
    int32_t mtid() const { return int32_t(u32(d + 20)); }
    int32_t mpos() const { return int32_t(u32(d + 24)); }
    const char *name() const { return reinterpret_cast<const char *>(d + 32); }
    size_t nameLength() const { return nameBytes() ? nameBytes() - 1 : 0; }
    uint32_t cigar(uint32_t k) const { return u32(d + 32 + nameBytes() + 4 * size_t(k)); }
    const uint8_t *packedBases() const { return d + 32 + nameBytes() + 4 * size_t(nCigar()); }
    char base(int32_t x) const { return "=ACMGRSVTWYHKDBN"[(packedBases()[x >> 1] >> ((~x & 1) << 2)) & 15]; }      // bam_nt16_rev_table
    const uint8_t *qualities() const { return packedBases() + (size_t(lSeq()) + 1) / 2; }
    const uint8_t *aux() const { return qualities() + size_t(lSeq()); }
    size_t auxBytes() const { return size_t((d + n) - aux()); }
};

class BgzfReader {
public:
    BgzfReader();
    ~BgzfReader();
    void open(const std::string &path);                  // throws std::string
    bool read(void *dst, size_t n);                      // false at end of file (or a short read)
    void seek(uint64_t voffset);
    uint64_t tell() const { return (uint64_t(blockAddress) << 16) | uint64_t(offset); }
private:
    // The windows of a run walk along the chromosome and each bam_fetch starts at its 16-kb bin's first chunk, so the same
    // blocks are asked for again and again: the last few inflated blocks are kept (round robin).
    enum { kCachedBlocks = 8 };
    struct CachedBlock { int64_t address, next; int length; std::vector<uint8_t> data; };
    bool loadBlock(int64_t address);
    FILE *f;
    int64_t blockAddress; int blockLength; int64_t nextAddress;
    const uint8_t *block;                                // the current block's bytes (inside cache[])
    int offset;
    CachedBlock cache[kCachedBlocks];
    int cacheNext;
    std::vector<uint8_t> comp;
    void *zs;                                            // z_stream, reset between blocks
    BgzfReader(const BgzfReader &); BgzfReader &operator=(const BgzfReader &);
};

// BGZF output: blocks of at most 0xff00 input bytes, raw deflate, the "BC" extra field, CRC32 + ISIZE, and the empty block
// that marks the end of the file (SAM/BAM specification §4.1) — what bam_open(fn, "wb") ... bam_close of libbam produces.
class BgzfWriter {
public:
    BgzfWriter() : f(NULL) {}
    ~BgzfWriter() { try { close(); } catch (...) {} }
    void open(const std::string &path);                  // throws std::string("Cannot open bamfile PATH for writing!") like DInDel.cpp:678
    void write(const void *src, size_t n);
    void close();                                        // flushes, writes the end-of-file block
private:
    void flushBlock();
    FILE *f;
    std::vector<uint8_t> pending;
    BgzfWriter(const BgzfWriter &); BgzfWriter &operator=(const BgzfWriter &);
};

class BamFile {
public:
    explicit BamFile(const std::string &path);           // opens path and path + ".bai" (or path with .bam -> .bai); throws std::string("Cannot open BAM file.")
    int getTID(const std::string &name) const;           // throws std::string("Cannot find ID!") like MyBam::getTID
    const std::string &headerText() const { return text; }
    const std::string &headerTextAsStored() const { return rawText; }     // the l_text bytes of the file, padding included (bam_header_write writes these)
    // the undecoded record (without its 4-byte length) the traversal is looking at: valid inside a fetch callback / after next()
    const std::vector<uint8_t> &rawRecord() const { return raw; }
    const std::vector<std::string> &targetNames() const { return names; }
    const std::vector<int32_t> &targetLengths() const { return lengths; }
    // bam_get_library: LB of the @RG line whose ID equals the record's RG tag; NULL if the tag or the library is missing
    const char *getLibrary(const BamRecord &b) const;
    // bam_fetch: every record of reference `tid` overlapping [beg, end), in file order.  The callback returns false to stop.
    template <class F> void fetch(int tid, int beg, int end, F callback) { fetchImpl(tid, beg, end, callback, true); }
    // The same traversal handing over records with the fixed fields and the CIGAR only (qname, seq, qualities, aux empty or
    // stale); the callback calls complete(record) on those it wants whole.  Valid only inside the callback.
    template <class F> void fetchCore(int tid, int beg, int end, F callback) { fetchImpl(tid, beg, end, callback, false); }
    void complete(BamRecord &b) const { decodeRest(raw, b); }
    bool next(BamRecord &b);                             // sequential read at the current position (false at end of file)
    std::string fileName;
private:
    // One record, undecoded: fetch() looks at tid / pos / CIGAR only and decodes the rest (name, bases, qualities, tags) of
    // the records it hands to the callback.
    template <class F> void fetchImpl(int tid, int beg, int end, F callback, bool whole);
    bool nextRaw(std::vector<uint8_t> &d);
    static void decodeCore(const std::vector<uint8_t> &d, BamRecord &b);     // fixed fields + CIGAR
    static void decodeRest(const std::vector<uint8_t> &d, BamRecord &b);
    // Two marks left by the previous fetch on a reference, each saying "every record in front of `voffset` ends at or before `beg`" — in
    // a coordinate-sorted file such records cannot overlap a later region that starts at `beg` or behind it, so the next fetch starts its
    // scan at the mark instead of at the head of the bin's chunk.  [0]: the first record that ended behind the fetch's own `beg`;
    // [1]: the first one that ended behind the fetch's `end` (or, if none did, the place behind the last record that started in front of
    // `end`) — the one that serves a caller walking along the chromosome, whose next region starts where this one ended.  (Records the
    // scan did not visit because their bins do not overlap the region lie in front of a mark only if they start in front of `end`, and
    // then they end at or before `beg`.)
    struct Resume { int tid, beg; uint64_t voffset; bool valid; Resume() : tid(-1), beg(0), voffset(0), valid(false) {} } resume[2];
    std::vector<uint8_t> raw;
    struct Chunk { uint64_t beg, end; bool operator<(const Chunk &o) const { return beg < o.beg; } };
    struct RefIndex { std::map<uint32_t, std::vector<Chunk> > bins; std::vector<uint64_t> linear; };
    std::vector<Chunk> chunksFor(int tid, int beg, int end) const;
    void loadIndex(const std::string &path);
    BgzfReader bgzf;
    std::string text, rawText;
    std::vector<std::string> names;
    std::vector<int32_t> lengths;
    std::map<std::string, int> strToTID;
    std::map<std::string, std::string> rg2lib;
    std::vector<RefIndex> index;
};

template <class F> void BamFile::fetchImpl(int tid, int beg, int end, F callback, bool whole)
{
    const std::vector<Chunk> chunks = chunksFor(tid, beg, end);
    // the later of the two marks that hold for this `beg` (see Resume)
    uint64_t from = 0;
    for (int k = 0; k < 2; k++) if (resume[k].valid && resume[k].tid == tid && beg >= resume[k].beg && resume[k].voffset > from) from = resume[k].voffset;
    bool noted = false, notedEnd = false;
    uint64_t afterLast = from;                    // just behind the last record scanned (all of them start in front of `end`)
    struct SetEndMark {                           // however the scan ends: everything in front of `at` ends at or before `end`
        Resume &r; int tid, end; const bool &found; const uint64_t &at;
        ~SetEndMark() { if (!found && at) { r.tid = tid; r.beg = end; r.voffset = at; r.valid = true; } }
    } setEndMark = { resume[1], tid, end, notedEnd, afterLast };
    resume[1].valid = false;
    BamRecord b;
    for (size_t i = 0; i < chunks.size(); i++) {
        if (chunks[i].end <= from) continue;
        bgzf.seek(chunks[i].beg < from ? from : chunks[i].beg);
        while (bgzf.tell() < chunks[i].end) {
            const uint64_t here = bgzf.tell();
            if (!nextRaw(raw)) return;
            decodeCore(raw, b);
            if (b.tid != tid || b.pos >= end) return;                    // past the region: the file is sorted
            const int recordEnd = int(b.endPos());
            if (!notedEnd && recordEnd > end) { resume[1].tid = tid; resume[1].beg = end; resume[1].voffset = here; resume[1].valid = true; notedEnd = true; }
            afterLast = bgzf.tell();
            if (recordEnd > beg) {
                if (!noted) { resume[0].tid = tid; resume[0].beg = beg; resume[0].voffset = here; resume[0].valid = true; noted = true; }
                if (b.pos < end) {                                       // is_overlap
                    if (whole) decodeRest(raw, b);
                    if (!callback(b)) return;
                }
            }
        }
    }
}

// bam_header_write + bam_write1: the header of `like` and then records given as their undecoded bytes
class BamWriter {
public:
    BamWriter(const std::string &path, const BamFile &like);
    void write(const std::vector<uint8_t> &record);      // block_size + the record
    void close() { out.close(); }
private:
    BgzfWriter out;
};

} // namespace dindel
#endif
