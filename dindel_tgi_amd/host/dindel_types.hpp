// dindel_types.hpp — host-side value types of the likelihood path (C++ mirror of the reference interface).
//
// Only the fields the path reads or writes are modelled; names follow the reference so
// that call sites read the same:
//   Haplotype   reference Haplotype.hpp:40-312   (seq, indels, snps)
//   Read        reference Read.hpp:31-449        (seq, qual, mapQual, posStat, isUnmapped())
//   AlignedVariant reference Variant.hpp:78-175  (string form, start/end in haplotype and read, isCovered)
//   MLAlignment reference MLAlignment.hpp:28-76  (the per-(read,haplotype) result record)
//   ObservationModelParameters reference ObservationModel.hpp:28-99
// Written from scratch for this library; no libbam / Boost dependency.
#ifndef DINDEL_TYPES_HPP
#define DINDEL_TYPES_HPP
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace dindel {

class AlignedVariant {
public:
    enum Type { INS, DEL, SNP, REF };
    AlignedVariant() : type(REF), length(0), startHap(-1), endHap(-1), startRead(-1), endRead(-1),
                       leftFlankHap(-1), rightFlankHap(-1), leftFlankRead(-1), rightFlankRead(-1), freq(-1.0), addComb(false) {}
    // "+SEQ" insertion, "-SEQ" deletion, "X=>Y" SNP — reference Variant.hpp:43-74
    AlignedVariant(const std::string &s, int _startHap, int _endHap, int _startRead, int _endRead)
        : str(s), startHap(_startHap), endHap(_endHap), startRead(_startRead), endRead(_endRead),
          leftFlankHap(_startHap), rightFlankHap(_endHap), leftFlankRead(_startRead), rightFlankRead(_endRead),   // Variant.hpp:88-94
          freq(-1.0), addComb(false)
    {
        initFromString(s);
    }
    // a candidate variant of the window file: canonical position, prior / frequency, "add combinatorially" — Variant.hpp:100-121
    AlignedVariant(const std::string &s, int canonicalPos, double _freq = -1.0, bool _addComb = false)
        : str(s), startRead(-1), endRead(-1), leftFlankRead(-1), rightFlankRead(-1), freq(_freq), addComb(_addComb)
    {
        initFromString(s);
        startHap = canonicalPos;
        endHap = (type == DEL) ? startHap + length - 1 : startHap;
        leftFlankHap = startHap; rightFlankHap = endHap;
    }
    const std::string &getString() const { return str; }
    const std::string &getSeq() const { return seq; }
    Type getType() const { return type; }
    int size() const { return length; }
    bool isIndel() const { return type == INS || type == DEL; }
    bool isSNP() const { return type == SNP; }
    bool isRef() const { return type == REF; }
    double getFreq() const { return freq; }
    bool getAddComb() const { return addComb; }
    // ordering inside std::set<AlignedVariant> — Variant.hpp:131-134
    bool operator<(const AlignedVariant &v) const { return startHap != v.startHap ? startHap < v.startHap : str < v.str; }
    // "is this the candidate (pos, type, string)?" — Variant.hpp:135-150: SNPs compare the "=>X" part, insertions the string,
    // deletions only the length
    bool isEqual(int pos, int _type, const std::string &_str) const
    {
        if (int(type) != _type || startHap != pos) return false;
        if (_type == SNP) return _str.substr(1, 3) == str.substr(1, 3);
        if (_type == INS) return str == _str;
        if (_type == DEL) return str.size() == _str.size();
        return false;
    }
    int getStartHap() const { return startHap; }
    int getEndHap() const { return endHap; }
    int getStartRead() const { return startRead; }
    int getEndRead() const { return endRead; }
    // flanking coordinates — reference Variant.hpp:148-160 (set by the haplotype-to-reference alignment)
    int getLeftFlankHap() const { return leftFlankHap; }
    int getRightFlankHap() const { return rightFlankHap; }
    int getLeftFlankRead() const { return leftFlankRead; }
    int getRightFlankRead() const { return rightFlankRead; }
    void setFlanking(int lfh, int rfh, int lfr, int rfr) { leftFlankHap = lfh; rightFlankHap = rfh; leftFlankRead = lfr; rightFlankRead = rfr; }
    // reference Variant.hpp:125-128
    bool isCovered(int pad, int firstBase, int lastBase) const
    {
        return firstBase + pad <= startRead && lastBase - pad >= endRead;
    }
private:
    void initFromString(const std::string &s)          // Variant.hpp:43-74
    {
        if (s.size() > 1 && s[0] == '-') { type = DEL; length = int(s.size()) - 1; seq = s.substr(1); }
        else if (s.size() > 1 && s[0] == '+') { type = INS; length = int(s.size()) - 1; seq = s.substr(1); }
        else if (s.size() == 4 && s[1] == '=' && s[2] == '>') { type = SNP; length = 1; seq = s; }
        else if (s == "*REF") { type = REF; length = 1; seq = s; }           // Variant.hpp:62-65
        else throw std::string("Unrecognized variant");
    }
    Type type;
    std::string seq, str;
    int length;
    int startHap, endHap;     // position of the variant in the haplotype the read is aligned to
    int startRead, endRead;   // position of the variant in the read aligned to the haplotype
    int leftFlankHap, rightFlankHap, leftFlankRead, rightFlankRead;
    double freq;
    bool addComb;
};

class MLAlignment {
public:
    static const int INS = -1, DEL = -2, LO = -3, RO = -4;   // reference MLAlignment.hpp:31-34
    MLAlignment()
        : relPos(-1), firstBase(-1), lastBase(-1), ll(0.0), llOn(0.0), llOff(0.0), offHap(false), offHapHMQ(false),
          hl(-1), hr(-1), numIndels(0), numMismatch(0), nBQT(0), nmmBQT(0), mLogBQ(0.0), nMMLeft(0), nMMRight(0) {}
    int relPos;
    int firstBase, lastBase;
    std::map<int, AlignedVariant> indels, snps;
    std::map<int, bool> hapIndelCovered, hapSNPCovered;
    // not in the reference record: filterHaplotypes' per-read coverage test of each haplotype indel (DInDel.cpp:1973-2054),
    // evaluated on the device next to hpos so the host reduction does not have to re-scan hpos
    std::map<int, bool> hapIndelFilterCovered;
    double ll, llOn, llOff;
    bool offHap, offHapHMQ;
    int hl, hr;
    int numIndels, numMismatch;
    int nBQT, nmmBQT;
    double mLogBQ;
    int nMMLeft, nMMRight;
    std::string align;
    std::vector<int> hpos;
    operator double() const { return ll; }
};

class Haplotype {
public:
    Haplotype() {}
    explicit Haplotype(const std::string &s) : seq(s) {}
    std::string seq;
    std::map<int, AlignedVariant> indels, snps;   // variants of this haplotype w.r.t. the reference sequence
    // hap.ml.hpos of the reference (Haplotype.hpp: `MLAlignment ml`, filled by alignHaplotypes): for every haplotype base its
    // offset on the window's reference sequence, or a negative MLAlignment code.  Only getCIGAR reads it; empty = not aligned.
    std::vector<int> refHpos;
    size_t size() const { return seq.size(); }
    const char &operator[](size_t i) const { return seq[i]; }
    // reference Haplotype.hpp:254-270
    int countIndels() const
    {
        int num = 0;
        for (std::map<int, AlignedVariant>::const_iterator it = indels.begin(); it != indels.end(); ++it) if (it->second.isIndel()) num++;
        return num;
    }
    int countSNPs() const
    {
        int num = 0;
        for (std::map<int, AlignedVariant>::const_iterator it = snps.begin(); it != snps.end(); ++it) if (it->second.isSNP() && !it->second.isRef()) num++;
        return num;
    }
};

// Insert-size distribution of a sequencing library — reference Library.hpp:37-140: probs[d] = normalised histogram
// floored at 1e-10 over d < maxins = min(25 * mode, #bins); getProb clamps |x| to maxins-1; the "95th percentile
// probability" is the smallest of the most probable bins that together hold > 95 % of the mass.
class Library {
public:
    Library() : maxins(0), ninetyfifth_pct_prob(0.0) {}
    explicit Library(const std::vector<double> &counts) { calcProb(counts); }
    int getMaxInsertSize() const { return maxins; }
    double getProb(int x) const { if (x < 0) x = -x; if (x >= maxins) x = maxins - 1; return probs[size_t(x)]; }
    double getNinetyFifthPctProb() const { return ninetyfifth_pct_prob; }
    const std::vector<double> &table() const { return probs; }
private:
    void calcProb(const std::vector<double> &counts);
    int maxins;
    double ninetyfifth_pct_prob;
    std::vector<double> probs;
};

class Read {
public:
    Read() : mapQual(0.0), posStat(0.0, 1.0), pos(0), unmapped(false), reverse(false), mateReverse(false),
             paired(false), mateUnmapped(false), mateSameTid(false), matePos(-1), mateLen(-1), library(NULL),
             onReverseStrand(false), bamPos(0), bamMatePos(-1), endPos(1), poolID(-1) {}
    Haplotype seq;                       // read.seq.seq is the base string, as in the reference
    std::vector<double> qual;            // P(base correct) — reference Read.hpp:143-148
    double mapQual;                      // P(mapping correct) — reference Read.hpp:127-131
    std::pair<double, double> posStat;   // .first: mean first-base position — reference Read.hpp:261-306
    uint32_t pos;
    bool unmapped;                       // BAM flag 0x4 (the reference reads bam->core.flag, Read.hpp:200)
    bool reverse, mateReverse;           // BAM flags 0x10 / 0x20 (Read.hpp:201-203)
    size_t size() const { return seq.size(); }
    bool isUnmapped() const { return unmapped; }
    bool isReverse() const { return reverse; }
    bool mateIsReverse() const { return mateReverse; }
    // inputs of the insert-size prior (mapUnmappedReads) — reference Read.hpp:162-204, 258, 426-434
    bool paired, mateUnmapped, mateSameTid;   // BAM flags 0x1 / 0x8; bam->core.tid == bam->core.mtid
    int32_t matePos, mateLen;            // bam->core.mpos; -1 until the mate has been seen (Read.hpp:163)
    const Library *library;
    bool isPaired() const { return paired; }
    bool mateIsUnmapped() const { return mateUnmapped; }
    const Library &getLibrary() const { return *library; }
    void setAllQual(double v) { qual.assign(seq.size(), v); }
    // what the window's read selection (DetInDel::getReads, DInDel.cpp:885-1262) reads from the BAM record behind the read
    std::string qname;                   // bam1_qname
    bool onReverseStrand;                // BAM flag 0x10 at construction (Read.hpp:167); getReads may reverse an unmapped read's sequence, not this
    int32_t bamPos, bamMatePos;          // bam->core.pos, bam->core.mpos
    uint32_t endPos;                     // getEndPos(): bam_calend, or pos + 1 without a CIGAR (Read.hpp:185-188)
    int poolID;
    // the BAM record behind the read (Read::bam), kept only when the realigned BAM is asked for (ReadSelectionParameters::keepRecords)
    std::shared_ptr<const std::vector<uint8_t> > record;
    // Read::getAuxData() of the record (Read.hpp:223-256), kept only when --filterReadAux asks for it
    std::string auxData;
    uint32_t getEndPos() const { return endPos; }
    int32_t getBAMMatePos() const { return bamMatePos; }
    // reverse() / complement() — Read.hpp:209-227 (getReads applies both to an unmapped read on its mate's strand)
    void complementSeq()
    {
        for (size_t i = 0; i < seq.seq.size(); i++) {
            char &n = seq.seq[i];
            if (n == 'A') n = 'T'; else if (n == 'T') n = 'A'; else if (n == 'C') n = 'G'; else if (n == 'G') n = 'C';
        }
    }
    void reverseSeq() { seq.seq = std::string(seq.seq.rbegin(), seq.seq.rend()); }
    // Phred -> probability exactly as the BAM constructor does — reference Read.hpp:127-131, 143-148
    static double phredToProb(double phred);
};

// reference ObservationModel.hpp:28-99 (fields the path reads)
class ObservationModelParameters {
public:
    ObservationModelParameters() { setDefaultValues(); }
    void setDefaultValues()
    {
        pError = 1e-4; pMut = 1e-4; maxLengthIndel = 10; maxLengthDel = maxLengthIndel; mapQualThreshold = 100.0;
        pFirstgLO = 0.01; checkBaseQualThreshold = 0.95; bMid = -1; forceReadOnHaplotype = false;
        mapUnmappedReads = false; padCover = 5; maxMismatch = 1; capMapQualFast = 40.0;
    }
    // what main() installs from the CLI defaults — reference DInDel.cpp:3937-3949, 4122-4157
    void setCLIDefaultValues()
    {
        setDefaultValues();
        pError = 5e-4; pMut = 1e-5; maxLengthIndel = 5; maxLengthDel = 5; padCover = 2; maxMismatch = 2; capMapQualFast = 45.0;
    }
    double pError, pMut, mapQualThreshold, pFirstgLO, checkBaseQualThreshold, capMapQualFast;
    int maxLengthIndel, maxLengthDel, bMid, padCover, maxMismatch;
    bool forceReadOnHaplotype, mapUnmappedReads;
};

} // namespace dindel
#endif
