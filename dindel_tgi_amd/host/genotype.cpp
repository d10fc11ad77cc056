// genotype.cpp — see genotype.hpp.
#include "genotype.hpp"
#include <cmath>
#include <cstdint>
#include <string>

namespace dindel {

// Two cases are answered without calling exp and log, with the value those calls give:
//   diff == 0 (equal finite arguments: every (h, h) pair of diploidGLF, and reads that do not tell two haplotypes apart):
//       exp(0) is 1 exactly, so the sum is l + log(2.0) — log(2.0) as this libm returns it, evaluated once;
//   diff < -36.75: exp(diff) < 2^-53, 1.0 + exp(diff) rounds to 1.0, log(1.0) is +0: the sum is l + 0.0.
// diploidGLF evaluates this 7,200 times per window of 8 haplotypes x 200 reads (tests/test_ref_bits.py checks it against the
// reference's function on equal, far-apart, infinite and NaN arguments).
double addLogs(const double l1, const double l2)
{
    static const volatile double one = 1.0;
    static const double logTwo = log(1.0 + one);          // at run time: the library's value, not the compiler's
    if (l1 > l2) {
        double diff = l2 - l1;
        if (diff < -36.75) return l1 + 0.0;
        return l1 + log(1.0 + exp(diff));
    } else {
        double diff = l1 - l2;
        if (diff == 0.0) return l2 + logTwo;
        if (diff < -36.75) return l2 + 0.0;
        return l2 + log(1.0 + exp(diff));
    }
}

PairPosteriorResult diploidPairPosteriors(int nh, const std::vector<double> &pair_sum, const std::vector<double> &prior,
                                          const std::vector<int> &filtered, const std::vector<int> &hap_num_candidate_indels)
{
    PairPosteriorResult R;
    R.pairs_posterior.assign(size_t(nh) * nh, 0.0);
    R.max_indel_pair[0] = R.max_indel_pair[1] = -1;
    R.max_noindel_pair[0] = R.max_noindel_pair[1] = -1;
    R.max_ll_indel = -HUGE_VAL;
    R.max_ll_noindel = -HUGE_VAL;
    for (int h1 = 0; h1 < nh; h1++) if (filtered[h1] == 0) for (int h2 = h1; h2 < nh; h2++) if (filtered[h2] == 0) {
        const double pp = pair_sum[h1 * nh + h2] + prior[h1 * nh + h2];              // :3091
        R.pairs_posterior[h1 * nh + h2] = pp;
        if (pp > R.max_ll_indel && (hap_num_candidate_indels[h1] > 0 || hap_num_candidate_indels[h2] > 0)) {   // :3103
            R.max_ll_indel = pp; R.max_indel_pair[0] = h1; R.max_indel_pair[1] = h2;
        }
        if (pp > R.max_ll_noindel && (hap_num_candidate_indels[h1] == 0 && hap_num_candidate_indels[h2] == 0)) { // :3108
            R.max_ll_noindel = pp; R.max_noindel_pair[0] = h1; R.max_noindel_pair[1] = h2;
        }
    }
    R.ll_ref = R.max_ll_noindel;
    R.qual = -10.0 * (R.ll_ref - addLogs(R.max_ll_indel, R.ll_ref)) / log(10.0);         // :3118
    if (R.max_indel_pair[0] == -1 || R.max_indel_pair[1] == -1) throw std::string("Could not find indel allele");   // :3121
    return R;
}

// DetInDel::filterHaplotypes on eager records (the view-based form the window loop uses is in diploid_glf.cpp).  Reads are numbered, so
// the reference's sets of read indices are bit sets here; a variant's coverage over the haplotypes that are kept is the OR of its rows.
void filterHaplotypes(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                      const std::vector<std::vector<MLAlignment> > &liks, std::vector<int> &filtered,
                      std::map<VariantKey, VariantCoverage> &varCoverage, bool doFilter)
{
    const size_t nh = haps.size(), nr = reads.size(), words = (nr + 63) / 64;
    typedef std::vector<uint64_t> Bits;
    Bits reverse(words, 0);                                                  // the strand a read counts for (:1982-1987)
    for (size_t r = 0; r < nr; r++) {
        const bool rev = reads[r].isUnmapped() ? !reads[r].mateIsReverse() : reads[r].isReverse();
        if (rev) reverse[r >> 6] |= uint64_t(1) << (r & 63);
    }
    struct Row { VariantKey key; size_t hap; Bits covering; };
    std::vector<Row> rows;                                                   // one per (haplotype, variant) the walk got to
    filtered.assign(nh, 0);
    for (size_t h = 0; h < nh; h++) {
        bool everyIndelCovered = true;
        for (std::map<int, AlignedVariant>::const_iterator it = haps[h].indels.begin(); it != haps[h].indels.end() && everyIndelCovered; ++it) {
            Row row;
            row.key = VariantKey(it->first, it->second.getString());
            row.hap = h;
            row.covering.assign(words, 0);
            if (it->second.isIndel()) {
                bool any = false;
                for (size_t r = 0; r < nr; r++) {
                    const MLAlignment &ml = liks[h][r];
                    if (ml.offHapHMQ || ml.numIndels != 0) continue;             // the reads selected for this haplotype (:1951-1955)
                    const std::map<int, bool>::const_iterator f = ml.hapIndelFilterCovered.find(it->first);
                    if (f != ml.hapIndelFilterCovered.end() && f->second) { row.covering[r >> 6] |= uint64_t(1) << (r & 63); any = true; }
                }
                if (!any) everyIndelCovered = false;                             // :2060-2063: the walk over this haplotype ends here
            }
            rows.push_back(row);                                                 // (a variant without covering reads still gets its entry)
        }
        if (doFilter && !everyIndelCovered) filtered[h] = 1;                     // :2068-2073
    }
    varCoverage.clear();
    std::map<VariantKey, std::pair<Bits, Bits> > united;                         // forward, reverse reads over the haplotypes kept (:2088-2097)
    for (size_t k = 0; k < rows.size(); k++) {
        std::pair<Bits, Bits> &u = united[rows[k].key];
        if (u.first.empty()) { u.first.assign(words, 0); u.second.assign(words, 0); }
        if (filtered[rows[k].hap] == 1) continue;
        for (size_t w = 0; w < words; w++) { u.first[w] |= rows[k].covering[w] & ~reverse[w]; u.second[w] |= rows[k].covering[w] & reverse[w]; }
    }
    for (std::map<VariantKey, std::pair<Bits, Bits> >::const_iterator it = united.begin(); it != united.end(); ++it) {
        int nf = 0, nrv = 0;
        for (size_t w = 0; w < words; w++) { nf += __builtin_popcountll(it->second.first[w]); nrv += __builtin_popcountll(it->second.second[w]); }
        varCoverage[it->first] = VariantCoverage(nf, nrv);
    }
}

} // namespace dindel
