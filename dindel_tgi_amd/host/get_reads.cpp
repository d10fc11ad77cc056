// get_reads.cpp — see get_reads.hpp.  Line references are to the reference's DInDel.cpp unless a file is named.
#include "get_reads.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <sstream>

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

std::string auxDataString(const BamRecord &b)
{
    std::ostringstream os;
    const std::vector<uint8_t> &a = b.aux;
    size_t i = 0;
    while (i + 3 <= a.size()) {
        const char k0 = char(a[i]), k1 = char(a[i + 1]), type = char(a[i + 2]);
        i += 3;
        os << "\t" << k0 << k1;
        if (type == 'A') { if (i + 1 > a.size()) break; os << "A:" << char(a[i]); i += 1; }
        else if (type == 'C' || type == 'c') { if (i + 1 > a.size()) break; os << "i:" << unsigned(a[i]); i += 1; }          // 'c' too: (int) of a uint8_t
        else if (type == 'S') { if (i + 2 > a.size()) break; os << "i:" << uint16_t(a[i] | (a[i + 1] << 8)); i += 2; }
        else if (type == 's') { if (i + 2 > a.size()) break; os << "i:" << int16_t(uint16_t(a[i] | (a[i + 1] << 8))); i += 2; }
        else if (type == 'I' || type == 'i' || type == 'f') {
            if (i + 4 > a.size()) break;
            const uint32_t v = uint32_t(a[i]) | (uint32_t(a[i + 1]) << 8) | (uint32_t(a[i + 2]) << 16) | (uint32_t(a[i + 3]) << 24);
            if (type == 'I') os << "i:" << v;
            else if (type == 'i') os << "i:" << int32_t(v);
            else { float f; memcpy(&f, &v, 4); os << "f:" << f; }
            i += 4;
        } else if (type == 'Z' || type == 'H') {
            os << type << ":";
            while (i < a.size() && a[i]) os << char(a[i++]);
            i++;
        } else break;
    }
    return os.str();
}

namespace {
// getLibraryName + the collection lookup of the Read constructor (Read.hpp:170-201)
const Library *lookupLibrary(const BamRecord &b, const BamFile &bam, const LibraryCollection &libraries, const std::string &overrideLibName)
{
    std::string libName;
    if (b.flag & BAM_FPAIRED) { const char *p = bam.getLibrary(b); libName = p ? std::string(p) : std::string("dindel_default"); }
    else libName = "single_end";
    LibraryCollection::const_iterator it = libraries.find(overrideLibName.empty() ? libName : overrideLibName);
    if (it == libraries.end()) throw std::string("Cannot find library: ").append(libName);
    return &it->second;
}
}

Read makeRead(const BamRecord &b, const BamFile &bam, const LibraryCollection &libraries, int poolID, const std::string &overrideLibName)
{
    // Phred values are bytes: Read::phredToProb of each, once (0 marks an entry whose conversion throws "Phred error.")
    static const std::vector<double> table = []() {
        std::vector<double> t(256, 0.0);
        for (int q = 0; q < 256; q++) { try { t[size_t(q)] = Read::phredToProb(double(q)); } catch (std::string &) { t[size_t(q)] = 0.0; } }
        return t;
    }();
    struct Conv { static double prob(const std::vector<double> &t, uint8_t q) { const double v = t[q]; return v != 0.0 ? v : Read::phredToProb(double(q)); } };
    Read r;
    r.mapQual = Conv::prob(table, b.qual);                           // Read.hpp:124-129
    r.pos = uint32_t(b.pos);
    r.seq.seq = b.seq;
    r.qual.resize(b.qualities.size());
    for (size_t x = 0; x < b.qualities.size(); x++) r.qual[x] = Conv::prob(table, b.qualities[x]);                    // :139-148
    r.posStat = computePositionStatistics(b);
    r.unmapped = (b.flag & BAM_FUNMAP) != 0; r.paired = (b.flag & BAM_FPAIRED) != 0; r.mateUnmapped = (b.flag & BAM_FMUNMAP) != 0;
    r.reverse = (b.flag & BAM_FREVERSE) != 0; r.mateReverse = (b.flag & BAM_FMREVERSE) != 0; r.mateSameTid = b.tid == b.mtid;
    r.onReverseStrand = r.reverse;                                   // :167
    r.poolID = poolID;
    r.matePos = b.mpos;                                              // :169
    r.mateLen = -1;
    r.qname = b.qname; r.bamPos = b.pos; r.bamMatePos = b.mpos; r.endPos = b.endPos();
    r.library = lookupLibrary(b, bam, libraries, overrideLibName);   // getLibraryName, :189-201
    return r;
}

namespace {
// The selection below works on one small record per buffered read instead of on copies of the reads: the reference copies the
// whole buffer (reads = readBuffer, :1055-1057), edits the copies, sorts them and keeps the head.  The buffer spans
// 2 * maxInsert + 200 bases around the window and only a fraction of it overlaps the window, so here the edits go into
// Selection records, those are sorted, and only the reads that are kept are copied.  std::sort is driven by the comparator's
// answers alone, so sorting the records with the reference's comparator (mapQual descending, :890-895) leaves them in the
// order it leaves the reads in.
struct Selection { double mapQual; uint32_t idx; int32_t matePos, mateLen; bool flip; };
bool byMapQualDescending(const Selection &r1, const Selection &r2) { return r1.mapQual > r2.mapQual; }

uint64_t hashName(const std::string &s)
{
    uint64_t h = 1469598103934665603ull;                                                     // FNV-1a
    for (size_t i = 0; i < s.size(); i++) { h ^= uint8_t(s[i]); h *= 1099511628211ull; }
    return h;
}

// qname -> the buffer indices carrying it, ascending (what the reference keeps in two hash_map<string, list<int>>, :1063-1072)
class NameIndex {
public:
    NameIndex(const std::deque<Read> &reads) : reads_(reads)
    {
        size_t cap = 16;
        while (cap < 2 * reads.size()) cap <<= 1;
        mask_ = cap - 1;
        slot_.assign(cap, -1);
        next_.assign(reads.size(), -1);
        tail_.assign(reads.size(), -1);
        count_.assign(reads.size(), 0);
        hash_.resize(reads.size());
        for (size_t r = 0; r < reads.size(); r++) {
            const uint64_t h = hash_[r] = hashName(reads[r].qname);
            size_t at = size_t(h) & mask_;
            for (;;) {
                const int head = slot_[at];
                if (head < 0) { slot_[at] = int(r); tail_[r] = int(r); count_[r] = 1; break; }
                if (hash_[size_t(head)] == h && reads[size_t(head)].qname == reads[r].qname) {
                    next_[size_t(tail_[size_t(head)])] = int(r); tail_[size_t(head)] = int(r); count_[size_t(head)]++;
                    break;
                }
                at = (at + 1) & mask_;
            }
        }
    }
    bool anyNameMoreThan(int n) const { for (size_t r = 0; r < count_.size(); r++) if (count_[r] > n) return true; return false; }
    // first buffer index with this name (-1: none); walk the others with next()
    int first(const std::string &qname) const
    {
        const uint64_t h = hashName(qname);
        size_t at = size_t(h) & mask_;
        for (;;) {
            const int head = slot_[at];
            if (head < 0) return -1;
            if (hash_[size_t(head)] == h && reads_[size_t(head)].qname == qname) return head;
            at = (at + 1) & mask_;
        }
    }
    int next(int r) const { return next_[size_t(r)]; }
    // calls f(idx) for every read called qname whose isUnmapped() == unmapped, ascending; returns how many there are
    template <class F> int each(const std::string &qname, bool unmapped, F f) const
    {
        int n = 0;
        for (int i = first(qname); i >= 0; i = next(i)) if (reads_[size_t(i)].isUnmapped() == unmapped) { n++; f(i); }
        return n;
    }
private:
    const std::deque<Read> &reads_;
    size_t mask_;
    std::vector<int> slot_, next_, tail_, count_;
    std::vector<uint64_t> hash_;
};
}

void ReadFetcher::getReads(const std::string &tid, uint32_t leftPos, uint32_t rightPos, std::vector<Read> &reads)
{
    const bool reset = resetReadBuffer;
    if (leftPos < oldLeftPos) throw std::string("Windows are not sorted!");                  // the reference exits (:899-902)
    // `reads` comes back holding the selection and nothing else, as after the reference's reads.clear() (:904); what it held
    // on entry is overwritten in place, so a caller that hands the same vector in again (dindel_gpu recycles its batches) spares
    // the allocator one string and one vector per read
    struct Trim {
        std::vector<Read> &v; size_t n;
        ~Trim() { v.resize(n); }
    } out = { reads, 0 };
    if (int(rightPos - leftPos) < 3 * params.minReadOverlap) throw std::string("Choose a larger width or a smaller minReadOverlap.");
    const int maxDev = int(libraries.getMaxInsertSize());
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
        size_t drop = 0;                                                                     // :942-961
        while (drop < readBuffer.size() && uint32_t(readBuffer[drop].bamPos) < leftMostReadPos) drop++;
        bool prefixOnly = true;                                                              // one sorted file: the reads to drop are the oldest ones
        for (size_t r = drop; r < readBuffer.size() && prefixOnly; r++) if (uint32_t(readBuffer[r].bamPos) < leftMostReadPos) prefixOnly = false;
        if (prefixOnly) readBuffer.erase(readBuffer.begin(), readBuffer.begin() + long(drop));
        else {
            size_t keep = 0;
            for (size_t r = 0; r < readBuffer.size(); r++)
                if (!(uint32_t(readBuffer[r].bamPos) < leftMostReadPos)) { if (keep != r) std::swap(readBuffer[keep], readBuffer[r]); keep++; }
            readBuffer.resize(keep);
        }
        if (leftMostReadPos < oldRightFetchReadPos) leftFetchReadPos = oldRightFetchReadPos;
    }
    int numReads = int(readBuffer.size());
    if (leftFetchReadPos <= rightFetchReadPos) {                                             // :981-993
        for (size_t b = 0; b < myBams.size(); b++) {
            BamFile &bam = *myBams[b];
            const int maxNumReads = int(params.maxReads * 100);
            const int pool = int(b);
            bam.fetchCore(bam.getTID(tid), int(leftFetchReadPos), int(rightFetchReadPos), [&](BamRecord &rec) -> bool {
                if (!((rec.flag & BAM_FDUP) || (rec.flag & BAM_FQCFAIL) || (rec.flag & 0x800))) {              // Read.hpp:392
                    // :998-1004: a read starting left of the fetched stretch was picked up by an earlier window; the reference
                    // builds it and drops it afterwards.  Here it is counted and its library looked up (what can throw), not built.
                    const bool wanted = uint32_t(rec.pos) >= leftFetchReadPos;
                    if (wanted || (rec.flag & BAM_FPAIRED)) bam.complete(rec);      // name, bases, qualities; the RG tag of a paired read
                    try {
                        if (wanted) {
                            readBuffer.push_back(makeRead(rec, bam, libraries, pool));
                            if (params.filterReadAux.size() > 1) readBuffer.back().auxData = auxDataString(rec);
                            if (params.keepRecords) readBuffer.back().record = std::make_shared<const std::vector<uint8_t> >(bam.rawRecord());
                        } else lookupLibrary(rec, bam, libraries, std::string());
                        numReads++;
                    } catch (std::string &s) {
                        if (s.find("Cannot find library") == std::string::npos) throw;
                        numUnknownLib++;
                        if (wanted) {
                            readBuffer.push_back(makeRead(rec, bam, libraries, pool, "single_end"));
                            if (params.filterReadAux.size() > 1) readBuffer.back().auxData = auxDataString(rec);
                            if (params.keepRecords) readBuffer.back().record = std::make_shared<const std::vector<uint8_t> >(bam.rawRecord());
                        } else lookupLibrary(rec, bam, libraries, "single_end");
                        numReads++;
                    }
                }
                if (numReads > maxNumReads) throw std::string("Too many reads in region");
                return true;
            });
        }
        oldRightFetchReadPos = rightFetchReadPos;
    }
    const size_t oldNumReads = readBuffer.size();
    const NameIndex names(readBuffer);
    if (names.anyNameMoreThan(2)) throw std::string("duplicate reads!");                     // :1027-1044
    // the reference's two indices (:1063-1072) are one here: a name's reads are walked in buffer order and told apart by
    // isUnmapped()
    const NameIndex &mates = names;
    std::vector<Selection> sel(readBuffer.size());
    for (size_t r = 0; r < sel.size(); r++) {
        sel[r].mapQual = readBuffer[r].mapQual; sel[r].idx = uint32_t(r); sel[r].matePos = readBuffer[r].matePos; sel[r].mateLen = readBuffer[r].mateLen;
        sel[r].flip = false;
    }
    int numTIDmismatch = 0, numOrphan = 0, numOrphanUnmapped = 0, numInRegion = 0;
    double minMapQual = params.mapQualThreshold;
    if (minMapQual < 0.0) minMapQual = 0.0;
    for (int r = 0; r < int(readBuffer.size()); r++) {                                       // :1095-1213
        const Read &rd = readBuffer[size_t(r)];
        Selection &out = sel[size_t(r)];
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
                    filter = true;                                                           // :1125-1127 (mateIsUnmapped() == false here)
                    bool inconsistent = false;
                    const int n = mates.each(rd.qname, false, [&](int idx) {
                        if (idx != r) {
                            out.mateLen = int32_t(readBuffer[size_t(idx)].size());
                            out.matePos = int32_t(readBuffer[size_t(idx)].pos);
                            filter = false;
                            if (out.matePos != rd.getBAMMatePos()) inconsistent = true;
                        }
                    });
                    if (n > 2) std::cerr << "HUH? DUPLICATE READ LABELS???" << std::endl;
                    if (inconsistent) throw std::string("matepos inconsistency!");           // the reference exits
                    if (filter == true) numOrphan++;                                         // (also when the name is not indexed, :1118-1121)
                }
            } else {                                                                         // mate unmapped (:1148-1166)
                out.matePos = int32_t(rd.pos);
                filter = true;
                const int n = mates.each(rd.qname, true, [&](int idx) {
                    if (idx != r) { out.mateLen = int32_t(readBuffer[size_t(idx)].size()); filter = false; }
                });
                if (n > 2) std::cerr << "HUH? DUPLICATE READ LABELS???" << std::endl;
                if (filter == true) numOrphan++;
            }
            if (filter == false) numInRegion++;
        } else if (params.mapUnmappedReads) {                                                // :1171-1209
            int idx = -1;
            const int n = mates.each(rd.qname, false, [&](int i) { if (idx < 0) idx = i; });
            if (n == 0) { numOrphanUnmapped++; filter = true; }
            else {
                if (n != 1) throw std::string("UNMAPPED READ HAS MORE THAN ONE MATE!");      // the reference exits
                const Read &mate = readBuffer[size_t(idx)];
                const int maxInsert = mate.getLibrary().getMaxInsertSize(), minInsert = 0;
                const uint32_t mpos = mate.pos;
                uint32_t range_l, range_r;
                if (mate.isReverse()) { range_l = mpos - uint32_t(maxInsert); range_r = mpos - uint32_t(minInsert); }
                else { range_l = mpos + uint32_t(minInsert); range_r = mpos + uint32_t(maxInsert); }
                if (range_r > leftPos && range_l < rightPos) {
                    numInRegion++;
                    filter = false;
                    out.mapQual = sel[size_t(idx)].mapQual;                                  // the mate's value as edited so far (-1 if it was filtered)
                    out.matePos = int32_t(mate.pos);
                    out.mateLen = int32_t(mate.size());
                    if (rd.isReverse() == mate.isReverse()) out.flip = true;                 // reverse() + complement()
                } else filter = true;
            }
        } else filter = true;
        if (filter == true) out.mapQual = -1.0;                                              // :1210
    }
    int nUnmapped = 0, nMateposError = 0;
    std::sort(sel.begin(), sel.end(), byMapQualDescending);                                  // :1218
    for (size_t max = 0; max < params.maxReads && max < sel.size(); max++) {                 // :1219-1227
        if (sel[max].mapQual < minMapQual) break;
        if (out.n < reads.size()) reads[out.n] = readBuffer[sel[max].idx]; else reads.push_back(readBuffer[sel[max].idx]);
        Read &rd = reads[out.n++];
        rd.mapQual = sel[max].mapQual; rd.matePos = sel[max].matePos; rd.mateLen = sel[max].mateLen;
        if (sel[max].flip) { rd.reverseSeq(); rd.complementSeq(); }
        if (rd.matePos == -1 && rd.isPaired() && !rd.mateIsUnmapped()) { nMateposError++; rd.matePos = int32_t(rd.pos); }
        if (rd.isUnmapped()) nUnmapped++;
    }
    if (params.filterReadAux.size() > 1) {                                                   // :1233-1243, Read::filterReads (Read.hpp:351-367)
        const bool exclude = params.filterReadAux[0] != '+';
        const std::string match = params.filterReadAux.substr(1);
        size_t keep = 0;
        for (size_t r = 0; r < out.n; r++) {
            const bool found = reads[r].auxData.find(match) != std::string::npos;
            if (found != exclude) { if (keep != r) std::swap(reads[keep], reads[r]); keep++; }
        }
        if (!params.quiet) std::cout << "filterAux: " << out.n - keep << " reads were filtered based on match string " << params.filterReadAux << std::endl;
        out.n = keep;
    }
    if (!params.quiet)
        std::cout << "Number of reads: " << out.n << " out of " << oldNumReads << " # unmapped reads: " << nUnmapped << " numReadsUnknownLib: " << numUnknownLib
                  << " numChrMismatch: " << numTIDmismatch << " numMappedWithoutMate: " << numOrphan << " numUnmappedWithoutMate: " << numOrphanUnmapped << std::endl;
    if (nMateposError) std::cerr << "The mate position of " << nMateposError << " reads was recorded as -1 in the BAM file" << std::endl;
    (void)numInRegion;
    if (out.n < 2) throw std::string("too_few_reads");                                       // :1256-1260
    else if (out.n >= params.maxReads) throw std::string("above_read_count_threshold");
}

} // namespace dindel
