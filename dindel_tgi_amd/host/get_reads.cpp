// get_reads.cpp — see get_reads.hpp.  Line references are to the reference's DInDel.cpp unless a file is named.
#include "get_reads.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
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
const std::vector<double> &phredTable()          // Phred values are bytes: Read::phredToProb of each, once (none of the 256 throws)
{
    static const std::vector<double> table = []() {
        std::vector<double> t(256);
        for (int q = 0; q < 256; q++) t[size_t(q)] = Read::phredToProb(double(q));
        return t;
    }();
    return table;
}

// Read::computePositionStatistics on the record's bytes (the CIGAR in place)
std::pair<double, double> positionStatistics(const RawBamView &v)
{
    const uint32_t n = v.nCigar();
    if (n == 0) return std::pair<double, double>(-1.0, -1.0);
    int32_t pos = 0, mean = 0, totLen = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t c = v.cigar(k);
        const int op = int(c & 15u);
        const int32_t len = int32_t(c >> 4);
        if (op == BAM_CMATCH) { mean += len * (pos - totLen); totLen += len; }
        if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CSOFT_CLIP || op == BAM_CHARD_CLIP) pos += len;
    }
    const double dmean = double(mean) / double(totLen);
    double var = 0.0;
    pos = 0; totLen = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t c = v.cigar(k);
        const int op = int(c & 15u);
        const int32_t len = int32_t(c >> 4);
        if (op == BAM_CMATCH) { var += double(len) * (double(pos - totLen) - dmean) * (double(pos - totLen) - dmean); totLen += len; }
        if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CSOFT_CLIP || op == BAM_CHARD_CLIP) pos += len;
    }
    var = var / double(totLen);
    return std::pair<double, double>(dmean + double(uint32_t(v.pos())), var);
}

uint64_t hashBytes(const char *s, size_t n)
{
    uint64_t h = 1469598103934665603ull;                                                     // FNV-1a
    for (size_t i = 0; i < n; i++) { h ^= uint8_t(s[i]); h *= 1099511628211ull; }
    return h;
}

// The reference copies the whole buffer (reads = readBuffer, :1055-1057), edits the copies, sorts them and keeps the head.  The
// buffer spans 2 * maxInsert + 200 bases around the window and only a fraction of it overlaps the window, so here the edits go
// into one small Selection per buffered alignment, those are sorted, and a Read is built only for the ones that are kept.
// std::sort is driven by the comparator's answers alone, so sorting the Selections with the reference's comparator (mapQual
// descending, :890-895) leaves them in the order it leaves the reads in.
struct ByMapQualDescending { template <class S> bool operator()(const S &r1, const S &r2) const { return r1.mapQual > r2.mapQual; } };
}

// The Read the reference's BAM constructor makes of this record (Read.hpp:120-183), into `dst` — every field is assigned, so
// `dst` may be a Read of an earlier window (its strings and vectors keep their storage).
void ReadFetcher::buildRead(const BufferedAlignment &a, Read &r) const
{
    const RawBamView v = bytesOf(a);
    const std::vector<double> &table = phredTable();
    r.mapQual = a.mapQual;
    r.pos = uint32_t(a.pos);
    const size_t L = size_t(a.length);
    r.seq.seq.resize(L);
    r.qual.resize(L);
    if (L) {
        static const char nt16[] = "=ACMGRSVTWYHKDBN";                // bam_nt16_rev_table
        const uint8_t *packed = v.packedBases(), *q = v.qualities();
        char *bases = &r.seq.seq[0];
        double *probs = r.qual.data();
        const double *t = table.data();
        size_t x = 0;
        for (; x + 1 < L; x += 2) { const uint8_t two = packed[x >> 1]; bases[x] = nt16[two >> 4]; bases[x + 1] = nt16[two & 15]; }
        if (x < L) bases[x] = nt16[packed[x >> 1] >> 4];
        for (x = 0; x < L; x++) probs[x] = t[q[x]];
    }
    r.seq.indels.clear(); r.seq.snps.clear(); r.seq.refHpos.clear();
    r.posStat = positionStatistics(v);
    r.unmapped = a.isUnmapped(); r.paired = a.isPaired(); r.mateUnmapped = a.mateIsUnmapped();
    r.reverse = a.isReverse(); r.mateReverse = (a.flag & BAM_FMREVERSE) != 0; r.mateSameTid = a.mateSameTid;
    r.onReverseStrand = r.reverse;
    r.poolID = a.pool;
    r.matePos = a.mpos;
    r.mateLen = -1;
    r.qname.assign(v.name(), v.nameLength());
    r.bamPos = a.pos; r.bamMatePos = a.mpos; r.endPos = a.endPos;
    r.library = a.library;
    if (params.keepRecords) r.record = std::make_shared<const std::vector<uint8_t> >(v.d, v.d + v.n); else r.record.reset();
    if (params.filterReadAux.size() > 1) {
        BamRecord tmp;
        tmp.aux.assign(v.aux(), v.aux() + v.auxBytes());
        r.auxData = auxDataString(tmp);
    } else r.auxData.clear();
}

void ReadFetcher::compactArena()
{
    std::vector<uint8_t> fresh;
    fresh.reserve(arena.size() - deadBytes + (1u << 16));
    for (size_t r = 0; r < readBuffer.size(); r++) {
        BufferedAlignment &a = readBuffer[r];
        const size_t at = fresh.size();
        fresh.insert(fresh.end(), arena.begin() + long(a.bytesAt), arena.begin() + long(a.bytesAt + a.nBytes));
        a.bytesAt = at;
    }
    arena.swap(fresh);
    deadBytes = 0;
}

void ReadFetcher::getReads(const std::string &tid, uint32_t leftPos, uint32_t rightPos, std::vector<Read> &reads)
{
    const bool reset = resetReadBuffer;
    if (leftPos < oldLeftPos) { const FatalError e = { "Windows are not sorted!", 3 }; throw e; }      // the reference exits (:899-902)
    // `reads` comes back holding the selection and nothing else, as after the reference's reads.clear() (:904); what it held
    // on entry is overwritten in place, so a caller that hands the same vector in again (dindel_gpu recycles its batches) spares
    // the allocator two strings and one vector per read
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
        arena.clear();
        deadBytes = 0;
        oldRightFetchReadPos = rightFetchReadPos;
    } else {
        size_t keep = 0;                                                                     // :942-961, order kept
        for (size_t r = 0; r < readBuffer.size(); r++) {
            if (uint32_t(readBuffer[r].pos) < leftMostReadPos) deadBytes += readBuffer[r].nBytes;
            else { if (keep != r) readBuffer[keep] = readBuffer[r]; keep++; }
        }
        readBuffer.resize(keep);
        if (deadBytes > (1u << 20) && 2 * deadBytes > arena.size()) compactArena();
        if (leftMostReadPos < oldRightFetchReadPos) leftFetchReadPos = oldRightFetchReadPos;
    }
    int numReads = int(readBuffer.size());
    if (leftFetchReadPos <= rightFetchReadPos) {                                             // :981-993
        const std::vector<double> &table = phredTable();
        const LibraryCollection::const_iterator se = libraries.find("single_end");
        const Library *singleEnd = se == libraries.end() ? NULL : &se->second;
        for (size_t b = 0; b < myBams.size(); b++) {
            BamFile &bam = *myBams[b];
            const int maxNumReads = int(params.maxReads * 100);
            const int pool = int(b);
            bam.fetchCore(bam.getTID(tid), int(leftFetchReadPos), int(rightFetchReadPos), [&](BamRecord &rec) -> bool {
                if (!((rec.flag & BAM_FDUP) || (rec.flag & BAM_FQCFAIL) || (rec.flag & 0x800))) {              // Read.hpp:392
                    // :998-1004: a read starting left of the fetched stretch was picked up by an earlier window; the reference
                    // builds it and drops it afterwards.  Either way only what can throw is done now: the library lookup (a paired
                    // read's RG tag); a record that is wanted is buffered as its bytes.
                    const bool wanted = uint32_t(rec.pos) >= leftFetchReadPos;
                    if (rec.flag & BAM_FPAIRED) bam.complete(rec);
                    const Library *library = NULL;
                    try {
                        // an unpaired record's library is "single_end" whatever its tags say (Read::getLibraryName, Read.hpp:189-201)
                        library = (!(rec.flag & BAM_FPAIRED) && singleEnd) ? singleEnd : lookupLibrary(rec, bam, libraries, std::string());
                    } catch (std::string &s) {
                        if (s.find("Cannot find library") == std::string::npos) throw;
                        numUnknownLib++;
                        library = lookupLibrary(rec, bam, libraries, "single_end");
                    }
                    numReads++;
                    if (wanted) {
                        const std::vector<uint8_t> &raw = bam.rawRecord();
                        const RawBamView v(raw.data(), raw.size());
                        BufferedAlignment a;
                        a.mapQual = table[rec.qual];                                         // Read.hpp:124-129
                        a.pos = rec.pos; a.mpos = rec.mpos; a.endPos = rec.endPos(); a.length = uint32_t(rec.l_qseq);
                        a.flag = rec.flag; a.mateSameTid = rec.tid == rec.mtid; a.pool = pool;
                        a.library = library;
                        a.nameHash = hashBytes(v.name(), v.nameLength());
                        a.bytesAt = arena.size(); a.nBytes = uint32_t(raw.size());
                        arena.insert(arena.end(), raw.begin(), raw.end());
                        readBuffer.push_back(a);
                    }
                }
                if (numReads > maxNumReads) throw std::string("Too many reads in region");
                return true;
            });
        }
        oldRightFetchReadPos = rightFetchReadPos;
    }
    const size_t oldNumReads = readBuffer.size();
    const size_t N = readBuffer.size();

    // qname -> the buffer indices carrying it, ascending (what the reference keeps in two hash_map<string, list<int>>, :1063-1072, and
    // in the qnameCount of its duplicate check, :1027-1044): one open-addressing table over the name hashes, rebuilt per window — the
    // hashes are stored with the alignments, so this is a pass over N small records.
    size_t cap = 16;
    while (cap < 2 * N) cap <<= 1;
    const size_t mask = cap - 1;
    nameSlot.assign(cap, -1);
    nameNext.assign(N, -1); nameTail.resize(N); nameCount.assign(N, 0);
    auto sameName = [&](size_t x, size_t y) {
        const RawBamView vx = bytesOf(readBuffer[x]), vy = bytesOf(readBuffer[y]);
        return vx.nameLength() == vy.nameLength() && memcmp(vx.name(), vy.name(), vx.nameLength()) == 0;
    };
    bool tooMany = false;
    for (size_t r = 0; r < N; r++) {
        const uint64_t h = readBuffer[r].nameHash;
        for (size_t at = size_t(h) & mask;; at = (at + 1) & mask) {
            const int32_t head = nameSlot[at];
            if (head < 0) { nameSlot[at] = int32_t(r); nameTail[r] = int32_t(r); nameCount[r] = 1; break; }
            if (readBuffer[size_t(head)].nameHash == h && sameName(size_t(head), r)) {
                nameNext[size_t(nameTail[size_t(head)])] = int32_t(r); nameTail[size_t(head)] = int32_t(r);
                if (++nameCount[size_t(head)] > 2) tooMany = true;
                break;
            }
        }
    }
    if (tooMany) throw std::string("duplicate reads!");                                      // :1027-1044
    // calls f(idx) for every buffered alignment named like `r` whose isUnmapped() == unmapped, ascending; returns how many there are
    auto eachNamedLike = [&](size_t r, bool unmapped, const std::function<void(int)> &f) -> int {
        const uint64_t h = readBuffer[r].nameHash;
        int32_t head = -1;
        for (size_t at = size_t(h) & mask;; at = (at + 1) & mask) {
            head = nameSlot[at];
            if (head < 0 || (readBuffer[size_t(head)].nameHash == h && sameName(size_t(head), r))) break;
        }
        int n = 0;
        for (int32_t i = head; i >= 0; i = nameNext[size_t(i)]) if (readBuffer[size_t(i)].isUnmapped() == unmapped) { n++; f(int(i)); }
        return n;
    };

    sel.resize(N);
    for (size_t r = 0; r < N; r++) {
        sel[r].mapQual = readBuffer[r].mapQual; sel[r].idx = uint32_t(r); sel[r].matePos = readBuffer[r].mpos; sel[r].mateLen = -1;
        sel[r].flip = false;
    }
    int numTIDmismatch = 0, numOrphan = 0, numOrphanUnmapped = 0, numInRegion = 0;
    double minMapQual = params.mapQualThreshold;
    if (minMapQual < 0.0) minMapQual = 0.0;
    for (int r = 0; r < int(N); r++) {                                                       // :1095-1213
        const BufferedAlignment &rd = readBuffer[size_t(r)];
        Selection &out = sel[size_t(r)];
        bool filter = false;
        if (size_t(rd.length) > params.maxReadLength) filter = true;
        if (rd.endPos < leftMostReadPos || uint32_t(rd.pos) > rightMostReadPos) filter = true;
        const int rpos = int(rd.pos);
        if (!rd.isUnmapped()) {
            if (rpos + int(rd.length) < int(leftPos) + params.minReadOverlap || rpos > int(rightPos) - params.minReadOverlap) {
                filter = true;
            } else if (rd.mateIsUnmapped() == false) {
                if (!rd.mateSameTid) {
                    numTIDmismatch++;
                } else {
                    filter = true;                                                           // :1125-1127 (mateIsUnmapped() == false here)
                    bool inconsistent = false;
                    const int n = eachNamedLike(size_t(r), false, [&](int idx) {
                        if (idx != r) {
                            out.mateLen = int32_t(readBuffer[size_t(idx)].length);
                            out.matePos = readBuffer[size_t(idx)].pos;
                            filter = false;
                            if (out.matePos != rd.mpos) inconsistent = true;
                        }
                    });
                    if (n > 2) std::cerr << "HUH? DUPLICATE READ LABELS???" << std::endl;
                    if (inconsistent) { const FatalError e = { "matepos inconsistency!", 1 }; throw e; }       // the reference exits (:1134-1137)
                    if (filter == true) numOrphan++;                                         // (also when the name is not indexed, :1118-1121)
                }
            } else {                                                                         // mate unmapped (:1148-1166)
                out.matePos = rd.pos;
                filter = true;
                const int n = eachNamedLike(size_t(r), true, [&](int idx) {
                    if (idx != r) { out.mateLen = int32_t(readBuffer[size_t(idx)].length); filter = false; }
                });
                if (n > 2) std::cerr << "HUH? DUPLICATE READ LABELS???" << std::endl;
                if (filter == true) numOrphan++;
            }
            if (filter == false) numInRegion++;
        } else if (params.mapUnmappedReads) {                                                // :1171-1209
            int idx = -1;
            const int n = eachNamedLike(size_t(r), false, [&](int i) { if (idx < 0) idx = i; });
            if (n == 0) { numOrphanUnmapped++; filter = true; }
            else {
                if (n != 1) { const FatalError e = { "UNMAPPED READ HAS MORE THAN ONE MATE!", 1 }; throw e; }  // the reference exits (:1180-1183)
                const BufferedAlignment &mate = readBuffer[size_t(idx)];
                const int maxInsert = mate.library->getMaxInsertSize(), minInsert = 0;
                const uint32_t mpos = uint32_t(mate.pos);
                uint32_t range_l, range_r;
                if (mate.isReverse()) { range_l = mpos - uint32_t(maxInsert); range_r = mpos - uint32_t(minInsert); }
                else { range_l = mpos + uint32_t(minInsert); range_r = mpos + uint32_t(maxInsert); }
                if (range_r > leftPos && range_l < rightPos) {
                    numInRegion++;
                    filter = false;
                    out.mapQual = sel[size_t(idx)].mapQual;                                  // the mate's value as edited so far (-1 if it was filtered)
                    out.matePos = mate.pos;
                    out.mateLen = int32_t(mate.length);
                    if (rd.isReverse() == mate.isReverse()) out.flip = true;                 // reverse() + complement()
                } else filter = true;
            }
        } else filter = true;
        if (filter == true) out.mapQual = -1.0;                                              // :1210
    }
    int nUnmapped = 0, nMateposError = 0;
    std::sort(sel.begin(), sel.end(), ByMapQualDescending());                               // :1218
    for (size_t max = 0; max < params.maxReads && max < sel.size(); max++) {                 // :1219-1227
        if (sel[max].mapQual < minMapQual) break;
        if (out.n == reads.size()) reads.push_back(Read());
        Read &rd = reads[out.n++];
        buildRead(readBuffer[sel[max].idx], rd);
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
