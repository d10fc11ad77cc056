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

class BgzfReader {
public:
    BgzfReader() : f(NULL), blockAddress(-1), blockLength(0), offset(0) {}
    ~BgzfReader() { if (f) fclose(f); }
    void open(const std::string &path);                  // throws std::string
    bool read(void *dst, size_t n);                      // false at end of file (or a short read)
    void seek(uint64_t voffset);
    uint64_t tell() const { return (uint64_t(blockAddress) << 16) | uint64_t(offset); }
private:
    bool loadBlock(int64_t address);
    FILE *f;
    int64_t blockAddress; int blockLength; int64_t nextAddress;
    std::vector<uint8_t> block;
    int offset;
    BgzfReader(const BgzfReader &); BgzfReader &operator=(const BgzfReader &);
};

class BamFile {
public:
    explicit BamFile(const std::string &path);           // opens path and path + ".bai" (or path with .bam -> .bai); throws std::string("Cannot open BAM file.")
    int getTID(const std::string &name) const;           // throws std::string("Cannot find ID!") like MyBam::getTID
    const std::string &headerText() const { return text; }
    const std::vector<std::string> &targetNames() const { return names; }
    const std::vector<int32_t> &targetLengths() const { return lengths; }
    // bam_get_library: LB of the @RG line whose ID equals the record's RG tag; NULL if the tag or the library is missing
    const char *getLibrary(const BamRecord &b) const;
    // bam_fetch: every record of reference `tid` overlapping [beg, end), in file order.  The callback returns false to stop.
    template <class F> void fetch(int tid, int beg, int end, F callback);
    bool next(BamRecord &b);                             // sequential read at the current position (false at end of file)
    std::string fileName;
private:
    struct Chunk { uint64_t beg, end; bool operator<(const Chunk &o) const { return beg < o.beg; } };
    struct RefIndex { std::map<uint32_t, std::vector<Chunk> > bins; std::vector<uint64_t> linear; };
    std::vector<Chunk> chunksFor(int tid, int beg, int end) const;
    void loadIndex(const std::string &path);
    BgzfReader bgzf;
    std::string text;
    std::vector<std::string> names;
    std::vector<int32_t> lengths;
    std::map<std::string, int> strToTID;
    std::map<std::string, std::string> rg2lib;
    std::vector<RefIndex> index;
};

template <class F> void BamFile::fetch(int tid, int beg, int end, F callback)
{
    const std::vector<Chunk> chunks = chunksFor(tid, beg, end);
    BamRecord b;
    for (size_t i = 0; i < chunks.size(); i++) {
        bgzf.seek(chunks[i].beg);
        while (bgzf.tell() < chunks[i].end) {
            if (!next(b)) return;
            if (b.tid != tid || b.pos >= end) return;                    // past the region: the file is sorted
            if (int(b.endPos()) > beg && b.pos < end)                    // is_overlap
                if (!callback(b)) return;
        }
    }
}

} // namespace dindel
#endif
