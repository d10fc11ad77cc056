// genotype.cpp — see genotype.hpp.
#include "genotype.hpp"
#include <cmath>
#include <set>
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

void filterHaplotypes(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                      const std::vector<std::vector<MLAlignment> > &liks, std::vector<int> &filtered,
                      std::map<VariantKey, VariantCoverage> &varCoverage, bool doFilter)
{
    const int numHaps = int(haps.size());
    filtered = std::vector<int>(haps.size(), 0);
    varCoverage.clear();
    std::map<VariantKey, std::vector<std::set<int> > > hVarCoverage;
    for (int h = 0; h < numHaps; h++) {
        std::set<int> selReads;                                              // :1951-1955
        for (size_t r = 0; r < reads.size(); r++)
            if (!liks[h][r].offHapHMQ && liks[h][r].numIndels == 0) selReads.insert(int(r));
        bool allCovered = true;
        for (std::map<int, AlignedVariant>::const_iterator it = haps[h].indels.begin(); it != haps[h].indels.end(); ++it) {
            const AlignedVariant &av = it->second;
            const VariantKey pav(it->first, av.getString());
            if (hVarCoverage.find(pav) == hVarCoverage.end()) hVarCoverage[pav] = std::vector<std::set<int> >(haps.size() * 2);
            if (av.getType() == AlignedVariant::INS || av.getType() == AlignedVariant::DEL) {
                bool covered = false;
                for (std::set<int>::const_iterator rt = selReads.begin(); rt != selReads.end(); ++rt) {
                    const int r = *rt;
                    int strand = 0;                                          // :1982-1987
                    if (reads[r].isUnmapped()) { if (!reads[r].mateIsReverse()) strand = 1; }
                    else { if (reads[r].isReverse()) strand = 1; }
                    std::map<int, bool>::const_iterator f = liks[h][r].hapIndelFilterCovered.find(it->first);
                    if (f != liks[h][r].hapIndelFilterCovered.end() && f->second) {
                        hVarCoverage[pav][h + strand * numHaps].insert(r);
                        covered = true;
                    }
                }
                if (!covered) { allCovered = false; break; }                 // :2060-2063
            }
        }
        if (doFilter && !allCovered) filtered[h] = 1;                        // :2068-2073
    }
    for (std::map<VariantKey, std::vector<std::set<int> > >::const_iterator it = hVarCoverage.begin(); it != hVarCoverage.end(); ++it) {
        std::set<int> rf, rr;                                                // :2088-2097
        for (int h = 0; h < numHaps; h++) if (filtered[h] != 1) {
            rf.insert(it->second[h].begin(), it->second[h].end());
            rr.insert(it->second[h + numHaps].begin(), it->second[h + numHaps].end());
        }
        varCoverage[it->first] = VariantCoverage(int(rf.size()), int(rr.size()));
    }
}

} // namespace dindel
