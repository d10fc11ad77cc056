// get_reads.cpp — see get_reads.hpp.  Line references are to the reference's DInDel.cpp unless a file is named.
#include "get_reads.hpp"
#include <algorithm>
#include <cmath>
#include <iostream>
#include <list>
#include <map>

namespace dindel {

std::pair<double, double> computePositionStatistics(const BamRecord &b)
{
    int32_t pos = 0, mean = 0, totLen = 0;
    const uint32_t refPos = uint32_t(b.pos);
    double var = 0.0;
    if (b.cigar.empty()) return std::pair<double, double>(-1.0, -1.0);
    for (size_t k = 0; k < b.cigar.size(); ++k) {
        const int op = int(b.cigar[k] & 15u);
        const int32_t len = int32_t(b.cigar[k] >> 4);
        if (op == BAM_CMATCH) { mean += len * (pos - totLen); totLen += len; }
        if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CSOFT_CLIP || op == BAM_CHARD_CLIP) pos += len;
    }
    const double dmean = double(mean) / double(totLen);
    pos = 0; totLen = 0;
    for (size_t k = 0; k < b.cigar.size(); ++k) {
        const int op = int(b.cigar[k] & 15u);
        const int32_t len = int32_t(b.cigar[k] >> 4);
        if (op == BAM_CMATCH) { var += double(len) * (double(pos - totLen) - dmean) * (double(pos - totLen) - dmean); totLen += len; }
        if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CSOFT_CLIP || op == BAM_CHARD_CLIP) pos += len;
    }
    var = var / double(totLen);
    return std::pair<double, double>(dmean + double(refPos), var);
}

Read makeRead(const BamRecord &b, const BamFile &bam, const LibraryCollection &libraries, int poolID, const std::string &overrideLibName)
{
    Read r;
    r.mapQual = Read::phredToProb(double(b.qual));                   // Read.hpp:124-129
    r.pos = uint32_t(b.pos);
    r.seq.seq = b.seq;
    r.qual.reserve(b.qualities.size());
    for (size_t x = 0; x < b.qualities.size(); x++) r.qual.push_back(Read::phredToProb(double(b.qualities[x])));     // :139-148
    r.posStat = computePositionStatistics(b);
    r.unmapped = (b.flag & BAM_FUNMAP) != 0; r.paired = (b.flag & BAM_FPAIRED) != 0; r.mateUnmapped = (b.flag & BAM_FMUNMAP) != 0;
    r.reverse = (b.flag & BAM_FREVERSE) != 0; r.mateReverse = (b.flag & BAM_FMREVERSE) != 0; r.mateSameTid = b.tid == b.mtid;
    r.onReverseStrand = r.reverse;                                   // :167
    r.poolID = poolID;
    r.matePos = b.mpos;                                              // :169
    r.mateLen = -1;
    r.qname = b.qname; r.bamPos = b.pos; r.bamMatePos = b.mpos; r.endPos = b.endPos();
    std::string libName;                                             // getLibraryName, :189-201
    if (r.paired) { const char *p = bam.getLibrary(b); libName = p ? std::string(p) : std::string("dindel_default"); }
    else libName = "single_end";
    LibraryCollection::const_iterator it = libraries.find(overrideLibName.empty() ? libName : overrideLibName);
    if (it == libraries.end()) throw std::string("Cannot find library: ").append(libName);
    r.library = &it->second;
    return r;
}

namespace {
bool byMapQualDescending(const Read &r1, const Read &r2) { return r1.mapQual > r2.mapQual; }     // :890-895
}

void ReadFetcher::getReads(const std::string &tid, uint32_t leftPos, uint32_t rightPos, std::vector<Read> &reads)
{
    const bool reset = resetReadBuffer;
    if (leftPos < oldLeftPos) throw std::string("Windows are not sorted!");                  // the reference exits (:899-902)
    reads.clear();
    if (int(rightPos - leftPos) < 3 * params.minReadOverlap) throw std::string("Choose a larger width or a smaller minReadOverlap.");
    const int maxDev = int(libraries.getMaxInsertSize());
    typedef std::map<std::string, std::list<int> > NameIndex;
    NameIndex mapped_name_to_idx, unmapped_name_to_idx;
    NameIndex::const_iterator hash_it;
    int numUnknownLib = 0;
    const int LEFTPAD = 200;
    const uint32_t rightFetchReadPos = rightPos + uint32_t(maxDev);
    const uint32_t rightMostReadPos = rightPos + uint32_t(maxDev);
    uint32_t leftFetchReadPos = leftPos - uint32_t(maxDev) - uint32_t(LEFTPAD);
    const uint32_t leftMostReadPos = leftPos - uint32_t(maxDev) - uint32_t(LEFTPAD);

    if (reset) {                                                                             // :936-941
        readBuffer.clear();
        oldRightFetchReadPos = rightFetchReadPos;
    } else {
        std::vector<Read> kept;                                                              // :942-961
        for (size_t r = 0; r < readBuffer.size(); r++)
            if (!(uint32_t(readBuffer[r].bamPos) < leftMostReadPos)) kept.push_back(readBuffer[r]);
        readBuffer.swap(kept);
        if (leftMostReadPos < oldRightFetchReadPos) leftFetchReadPos = oldRightFetchReadPos;
    }
    int numReads = int(readBuffer.size());
    std::vector<Read> newReads;
    if (leftFetchReadPos <= rightFetchReadPos) {                                             // :981-993
        for (size_t b = 0; b < myBams.size(); b++) {
            BamFile &bam = *myBams[b];
            const int maxNumReads = int(params.maxReads * 100);
            const int pool = int(b);
            bam.fetch(bam.getTID(tid), int(leftFetchReadPos), int(rightFetchReadPos), [&](const BamRecord &rec) -> bool {
                if (!((rec.flag & BAM_FDUP) || (rec.flag & BAM_FQCFAIL) || (rec.flag & 0x800))) {              // Read.hpp:392
                    try {
                        newReads.push_back(makeRead(rec, bam, libraries, pool));
                        numReads++;
                    } catch (std::string &s) {
                        if (s.find("Cannot find library") == std::string::npos) throw;
                        numUnknownLib++;
                        newReads.push_back(makeRead(rec, bam, libraries, pool, "single_end"));
                        numReads++;
                    }
                }
                if (numReads > maxNumReads) throw std::string("Too many reads in region");
                return true;
            });
        }
        oldRightFetchReadPos = rightFetchReadPos;
    }
    for (size_t r = 0; r < newReads.size(); r++)                                             // :998-1004: reads overlapping the
        if (uint32_t(newReads[r].bamPos) >= leftFetchReadPos) readBuffer.push_back(newReads[r]);   // boundary were picked up before
    {                                                                                        // :1027-1044
        std::map<std::string, int> qnameCount;
        for (size_t r = 0; r < readBuffer.size(); r++)
            if (++qnameCount[readBuffer[r].qname] > 2) throw std::string("duplicate reads!");
    }
    newReads.clear();
    const size_t oldNumReads = readBuffer.size();
    reads = readBuffer;                                                                      // :1055-1057
    for (size_t r = 0; r < reads.size(); r++) {                                              // :1063-1072
        if (reads[r].isUnmapped()) unmapped_name_to_idx[reads[r].qname].push_back(int(r));
        else mapped_name_to_idx[reads[r].qname].push_back(int(r));
    }
    int numTIDmismatch = 0, numOrphan = 0, numOrphanUnmapped = 0, numInRegion = 0;
    double minMapQual = params.mapQualThreshold;
    if (minMapQual < 0.0) minMapQual = 0.0;
    for (int r = 0; r < int(reads.size()); r++) {                                            // :1095-1213
        Read &rd = reads[size_t(r)];
        bool filter = false;
        if (rd.size() > params.maxReadLength) filter = true;
        if (rd.getEndPos() < leftMostReadPos || uint32_t(rd.pos) > rightMostReadPos) filter = true;
        const int rpos = int(int32_t(rd.pos));
        if (!rd.isUnmapped()) {
            if (rpos + int(rd.size()) < int(leftPos) + params.minReadOverlap || rpos > int(rightPos) - params.minReadOverlap) {
                filter = true;
            } else if (rd.mateIsUnmapped() == false) {
                if (!rd.mateSameTid) {
                    numTIDmismatch++;
                } else {
                    hash_it = mapped_name_to_idx.find(rd.qname);
                    if (hash_it == mapped_name_to_idx.end()) { numOrphan++; filter = true; }
                    else {
                        if (hash_it->second.size() > 2) std::cerr << "HUH? DUPLICATE READ LABELS???" << std::endl;
                        filter = true;                                                       // :1125-1127 (mateIsUnmapped() == false here)
                        for (std::list<int>::const_iterator li = hash_it->second.begin(); li != hash_it->second.end(); ++li) {
                            const int idx = *li;
                            if (idx != r) {
                                rd.mateLen = int32_t(reads[size_t(idx)].size());
                                rd.matePos = int32_t(reads[size_t(idx)].pos);
                                filter = false;
                                if (rd.matePos != rd.getBAMMatePos()) throw std::string("matepos inconsistency!");   // the reference exits
                            }
                        }
                        if (filter == true) numOrphan++;
                    }
                }
            } else {                                                                         // mate unmapped (:1148-1166)
                rd.matePos = int32_t(rd.pos);
                hash_it = unmapped_name_to_idx.find(rd.qname);
                if (hash_it == unmapped_name_to_idx.end()) filter = true;
                else {
                    filter = true;
                    if (hash_it->second.size() > 2) std::cerr << "HUH? DUPLICATE READ LABELS???" << std::endl;
                    for (std::list<int>::const_iterator li = hash_it->second.begin(); li != hash_it->second.end(); ++li)
                        if (*li != r) { rd.mateLen = int32_t(reads[size_t(*li)].size()); filter = false; }
                }
                if (filter == true) numOrphan++;
            }
            if (filter == false) numInRegion++;
        } else if (params.mapUnmappedReads) {                                                // :1171-1209
            hash_it = mapped_name_to_idx.find(rd.qname);
            if (hash_it == mapped_name_to_idx.end()) { numOrphanUnmapped++; filter = true; }
            else {
                if (hash_it->second.size() != 1) throw std::string("UNMAPPED READ HAS MORE THAN ONE MATE!");      // the reference exits
                const int idx = *hash_it->second.begin();
                const Read &mate = reads[size_t(idx)];
                const int maxInsert = mate.getLibrary().getMaxInsertSize(), minInsert = 0;
                const uint32_t mpos = mate.pos;
                uint32_t range_l, range_r;
                if (mate.isReverse()) { range_l = mpos - uint32_t(maxInsert); range_r = mpos - uint32_t(minInsert); }
                else { range_l = mpos + uint32_t(minInsert); range_r = mpos + uint32_t(maxInsert); }
                if (range_r > leftPos && range_l < rightPos) {
                    numInRegion++;
                    filter = false;
                    rd.mapQual = mate.mapQual;
                    rd.matePos = int32_t(mate.pos);
                    rd.mateLen = int32_t(mate.size());
                    if (rd.isReverse() == mate.isReverse()) { rd.reverseSeq(); rd.complementSeq(); }
                } else filter = true;
            }
        } else filter = true;
        if (filter == true) rd.mapQual = -1.0;                                               // :1210
    }
    int nUnmapped = 0, nMateposError = 0;
    std::sort(reads.begin(), reads.end(), byMapQualDescending);                              // :1218
    std::vector<Read> filteredReads;
    for (size_t max = 0; max < params.maxReads && max < reads.size(); max++) {               // :1219-1227
        if (reads[max].mapQual < minMapQual) break;
        if (reads[max].matePos == -1 && reads[max].isPaired() && !reads[max].mateIsUnmapped()) { nMateposError++; reads[max].matePos = int32_t(reads[max].pos); }
        filteredReads.push_back(reads[max]);
        if (reads[max].isUnmapped()) nUnmapped++;
    }
    filteredReads.swap(reads);
    if (!params.quiet)
        std::cout << "Number of reads: " << reads.size() << " out of " << oldNumReads << " # unmapped reads: " << nUnmapped << " numReadsUnknownLib: " << numUnknownLib
                  << " numChrMismatch: " << numTIDmismatch << " numMappedWithoutMate: " << numOrphan << " numUnmappedWithoutMate: " << numOrphanUnmapped << std::endl;
    if (nMateposError) std::cerr << "The mate position of " << nMateposError << " reads was recorded as -1 in the BAM file" << std::endl;
    (void)numInRegion;
    if (reads.size() < 2) throw std::string("too_few_reads");                                // :1256-1260
    else if (reads.size() >= params.maxReads) throw std::string("above_read_count_threshold");
}

} // namespace dindel
