// get_reads.hpp — the window's read selection: DetInDel::getReads restated on the own BAM reader.
//   Read(const bam1_t*, ...)            reference Read.hpp:120-183   Phred -> probabilities, position statistics, library lookup
//   Read::fetchFuncVectorPooled         reference Read.hpp:388-412   duplicates / QC-fail / supplementary records dropped
//   DetInDel::getReads                  reference DInDel.cpp:885-1262 read buffer over consecutive windows, mate pairing, filters,
//                                                                     sort by mapping quality, maxReads cap
#ifndef DINDEL_GET_READS_HPP
#define DINDEL_GET_READS_HPP
#include <string>
#include <vector>
#include "bam_reader.hpp"
#include "dindel_types.hpp"
#include "window_io.hpp"

namespace dindel {

struct ReadSelectionParameters {         // the DetInDel::Parameters fields getReads looks at (CLI defaults: DInDel.cpp:4122-4157)
    ReadSelectionParameters() : maxReads(10000), maxReadLength(500), minReadOverlap(20), mapQualThreshold(0.99), mapUnmappedReads(false), quiet(true), keepRecords(false) {}
    size_t maxReads; size_t maxReadLength; int minReadOverlap; double mapQualThreshold; bool mapUnmappedReads; bool quiet;
    bool keepRecords;                    // Read::record = the read's BAM record (for --outputRealignedBAM)
    std::string filterReadAux;           // --filterReadAux: "+text" keeps, "-text" (or any other first character) drops the reads whose
                                         // auxiliary fields, printed as getAuxData() prints them, contain text (DInDel.cpp:1233-1243)
};

// Read(const bam1_t *b, libraries, poolID, header, overrideLibName) — reference Read.hpp:120-183.  Throws std::string("Phred error.")
// or std::string("Cannot find library: NAME").
Read makeRead(const BamRecord &b, const BamFile &bam, const LibraryCollection &libraries, int poolID, const std::string &overrideLibName = std::string());

// Read::getAuxData — reference Read.hpp:223-256: "\tXX" + "A:c" / "i:number" / "f:number" / "Z:text" per field.  (A field of type B or d
// is not advanced over in the reference, which then reads on through its bytes; here the text ends in front of such a field.)
std::string auxDataString(const BamRecord &b);

// Read::computePositionStatistics — reference Read.hpp:261-306
std::pair<double, double> computePositionStatistics(const BamRecord &b);

// One buffered alignment: the fields the selection looks at, and where the record's bytes lie in the fetcher's arena.  A Read (two
// strings, 8 bytes of probability per base) is built only for the alignments a window selects, straight into the caller's vector.
struct BufferedAlignment {
    double mapQual;                      // Read::phredToProb of the record's MAPQ
    int32_t pos, mpos; uint32_t endPos; uint32_t length;      // bam->core.pos / mpos, Read::getEndPos(), l_qseq
    uint16_t flag; bool mateSameTid; int32_t pool;
    const Library *library;
    uint64_t nameHash;
    size_t bytesAt; uint32_t nBytes;     // the record (RawBamView) inside ReadFetcher::arena
    bool isUnmapped() const { return (flag & BAM_FUNMAP) != 0; }
    bool mateIsUnmapped() const { return (flag & BAM_FMUNMAP) != 0; }
    bool isReverse() const { return (flag & BAM_FREVERSE) != 0; }
    bool isPaired() const { return (flag & BAM_FPAIRED) != 0; }
};

class ReadFetcher {
public:
    ReadFetcher(std::vector<BamFile *> &bams, const LibraryCollection &libs, const ReadSelectionParameters &p)
        : myBams(bams), libraries(libs), params(p), deadBytes(0), oldLeftPos(0), oldRightFetchReadPos(0), resetReadBuffer(true) {}
    // DetInDel::getReads for the next window of chromosome `tid` (windows must come sorted by leftPos, as in the reference).
    // Throws the reference's strings: "Choose a larger width or a smaller minReadOverlap.", "Too many reads in region",
    // "duplicate reads!", "too_few_reads", "above_read_count_threshold"; where the reference ends the process (windows out of order,
    // inconsistent mate positions, an unmapped read with several mates) it throws a FatalError: that is the run's problem, not a window's.
    struct FatalError { std::string message; int exitCode; };
    void getReads(const std::string &tid, uint32_t leftPos, uint32_t rightPos, std::vector<Read> &reads);
    // what detectIndels does around the call (DInDel.cpp:1327-1333, :1401-1408): a new chromosome or a skipped window resets the buffer
    void newChromosome() { resetReadBuffer = true; oldLeftPos = 0; }
    void windowDone(bool skipped, uint32_t leftPos) { resetReadBuffer = skipped; oldLeftPos = leftPos; }
    uint32_t previousLeftPos() const { return oldLeftPos; }
    // the buffer as getReads left it, in order: (qname, pool) — what the reference's readBuffer holds; for tests of the order across pools
    void buffered(std::vector<std::pair<std::string, int> > &out) const
    {
        out.clear();
        for (size_t r = 0; r < readBuffer.size(); r++) { const RawBamView v = bytesOf(readBuffer[r]); out.push_back(std::make_pair(std::string(v.name(), v.nameLength()), int(readBuffer[r].pool))); }
    }
private:
    RawBamView bytesOf(const BufferedAlignment &a) const { return RawBamView(arena.data() + a.bytesAt, a.nBytes); }
    void buildRead(const BufferedAlignment &a, Read &dst) const;
    void compactArena();
    std::vector<BamFile *> &myBams;
    const LibraryCollection &libraries;
    ReadSelectionParameters params;
    std::vector<BufferedAlignment> readBuffer;   // in the reference's buffer order: survivors of the previous windows, then each pool's new records
    std::vector<uint8_t> arena;                   // the buffered records' bytes; dropped records leave holes that compactArena() closes
    size_t deadBytes;
    uint32_t oldLeftPos, oldRightFetchReadPos;
    bool resetReadBuffer;
    // scratch of getReads, kept between calls
    struct Selection { double mapQual; uint32_t idx; int32_t matePos, mateLen; bool flip; };
    std::vector<Selection> sel;
    std::vector<int32_t> nameSlot, nameNext, nameTail, nameCount;
};

} // namespace dindel
#endif
