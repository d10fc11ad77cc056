// realigned_bam.cpp — see realigned_bam.hpp.  Line references are to the reference's DInDel.cpp.
#include "realigned_bam.hpp"
#include <algorithm>
#include <cmath>
#include <sstream>

namespace dindel {

namespace {
struct PairLik { double ll; int h1, h2; };
bool byLikDescending(const PairLik &a, const PairLik &b) { return a.ll > b.ll; }             // :3826-3831
}

std::pair<int, int> maxLikelihoodPair(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks,
                                      int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params)
{
    std::vector<PairLik> likPairs;                                                           // :3767-3824 (the counters of HapPairLik are not read here)
    const size_t lh = haps.size();
    for (size_t hp = 0; hp < lh; hp++) for (size_t hm = hp; hm < lh; hm++) {
        double ll = 0.0;
        for (size_t r = 0; r < reads.size(); r++) ll += (addLogs(liks.ll(hp, r), liks.ll(hm, r)) + log(.5));
        ll += getHaplotypePrior(haps[hp], haps[hm], leftPos, candidateVariants, params);     // usePrior = true at :591
        PairLik p = { ll, int(hp), int(hm) };
        likPairs.push_back(p);
    }
    // the reference sorts its HapPairLik records with std::sort and this comparator; the order of equal likelihoods is the
    // algorithm's, which depends on the comparisons alone — the same here
    std::sort(likPairs.begin(), likPairs.end(), byLikDescending);
    size_t midx = likPairs.size();                                                           // getMaxHap, :292-300
    double maxll = -HUGE_VAL;
    for (size_t idx = 0; idx < likPairs.size(); idx++) if (likPairs[idx].ll > maxll) { maxll = likPairs[idx].ll; midx = idx; }
    if (midx == likPairs.size()) throw std::string("no haplotype pair with a finite likelihood");   // the reference reads an uninitialised index
    return std::pair<int, int>(likPairs[midx].h1, likPairs[midx].h2);
}

void realignedCigars(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks, std::pair<int, int> pair,
                     int refSeqPos, std::vector<CIGAR> &cigars)
{
    cigars.assign(reads.size(), CIGAR());
    const size_t h1 = size_t(pair.first), h2 = size_t(pair.second);
    for (size_t r = 0; r < reads.size(); r++) {                                              // :596-611
        size_t hmax = h1;
        if (fabs(liks.ll(h1, r) - liks.ll(h2, r)) < 1e-8) {
            if (haps[h1].countIndels() < haps[h2].countIndels()) hmax = h1; else hmax = h2;
        } else {
            if (liks.ll(h1, r) > liks.ll(h2, r)) hmax = h1; else hmax = h2;
        }
        const MLAlignment ml = liks.get(hmax, r);
        cigars[r] = getCIGAR(haps[hmax].refHpos, haps[hmax].size(), ml, reads[r].size(), refSeqPos);
    }
}

std::string realignedBAMFileName(const std::string &prefix, int index, const std::string &tid, uint32_t leftPos, uint32_t rightPos, int minReadOverlap)
{
    std::stringstream os;                                                                    // :614-618
    os << index << "_" << tid << "_" << leftPos + uint32_t(minReadOverlap) << "_" << rightPos - uint32_t(minReadOverlap) << ".bam";
    return std::string(prefix).append(".ra.").append(os.str());
}

void writeRealignedBAMFile(const std::string &fileName, const std::vector<CIGAR> &cigars, const std::vector<Read> &reads, const std::vector<int> &onHap,
                           const BamFile &header)
{
    if (cigars.size() != reads.size()) throw std::string("Problem with the cigars.");        // :672
    BamWriter out(fileName, header);                                                         // bam_open + bam_header_write, :674-680
    std::vector<uint8_t> nb;
    for (size_t r = 0; r < reads.size(); r++) {
        if (!reads[r].record) throw std::string("Read has no BAM record.");
        const std::vector<uint8_t> &b = *reads[r].record;
        if (onHap[r]) {                                                                      // :686-718
            const uint32_t l_qname = b[8];
            const uint32_t old_ncig = uint32_t(b[12]) | (uint32_t(b[13]) << 8);
            const uint32_t new_ncig = uint32_t(cigars[r].size());
            nb.assign(b.begin(), b.begin() + 32 + long(l_qname));                            // the fixed fields and the name
            for (uint32_t n = 0; n < new_ncig; n++) {
                const uint32_t v = (uint32_t(cigars[r][n].second) << 4) | uint32_t(cigars[r][n].first);     // length << BAM_CIGAR_SHIFT | operation
                nb.push_back(uint8_t(v)); nb.push_back(uint8_t(v >> 8)); nb.push_back(uint8_t(v >> 16)); nb.push_back(uint8_t(v >> 24));
            }
            nb.insert(nb.end(), b.begin() + 32 + long(l_qname) + 4 * long(old_ncig), b.end());   // bases, qualities, tags
            nb[12] = uint8_t(new_ncig); nb[13] = uint8_t(new_ncig >> 8);                       // core.n_cigar (16 bits)
            const int32_t pos = int32_t(cigars[r].refPos);                                   // core.pos, :712
            const int32_t mpos = int32_t(uint32_t(b[24]) | (uint32_t(b[25]) << 8) | (uint32_t(b[26]) << 16) | (uint32_t(b[27]) << 24));
            const int32_t isize = pos - mpos;                                                // :714
            for (int k = 0; k < 4; k++) { nb[4 + size_t(k)] = uint8_t(uint32_t(pos) >> (8 * k)); nb[28 + size_t(k)] = uint8_t(uint32_t(isize) >> (8 * k)); }
            out.write(nb);
        } else out.write(b);                                                                 // :720
    }
    out.close();
}

} // namespace dindel
