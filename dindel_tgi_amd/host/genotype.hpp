// genotype.hpp — host side of "next" row N1: the MAP haplotype-pair step of DetInDel::diploidGLF
// (reference DInDel.cpp:3062-3120) on top of the device read-sums (dd_pair_sums, genotype_kernel.hip).
#ifndef DINDEL_GENOTYPE_HPP
#define DINDEL_GENOTYPE_HPP
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "dindel_types.hpp"

namespace dindel {

struct PairPosteriorResult {
    std::vector<double> pairs_posterior;     // [nh*nh], entries h1<=h2 of unfiltered pairs
    int max_indel_pair[2], max_noindel_pair[2];
    double max_ll_indel, max_ll_noindel;
    double ll_ref, qual;                     // qual = -10 (ll_ref - addLogs(max_ll_indel, ll_ref)) / ln 10   (:3118)
};

// addLogs — reference Utils.hpp:29-38
double addLogs(double l1, double l2);

// pair_sum[h1*nh+h2] = sum_r log(.5)+addLogs(rl[r,h1],rl[r,h2]) (from the device); prior[h1*nh+h2] =
// getHaplotypePrior(h1,h2) (:3064-3068); filtered[h] from filterHaplotypes; hap_num_candidate_indels[h] (:2984-2993).
// Throws std::string("Could not find indel allele") like :3121.
PairPosteriorResult diploidPairPosteriors(int nh, const std::vector<double> &pair_sum, const std::vector<double> &prior,
                                          const std::vector<int> &filtered, const std::vector<int> &hap_num_candidate_indels);

// DetInDel::filterHaplotypes (reference DInDel.cpp:1932-2100) on top of the device's per-read coverage flags
// (MLAlignment::hapIndelFilterCovered): which haplotypes have every own indel covered by at least one selected read,
// and per variant the number of covering forward / reverse reads over the unfiltered haplotypes.
struct VariantCoverage { int nf, nr; VariantCoverage(int f = 0, int r = 0) : nf(f), nr(r) {} };   // DInDel.hpp:53-66
typedef std::pair<int, std::string> VariantKey;   // (position in the haplotype map, variant string)
void filterHaplotypes(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                      const std::vector<std::vector<MLAlignment> > &liks, std::vector<int> &filtered,
                      std::map<VariantKey, VariantCoverage> &varCoverage, bool doFilter);

} // namespace dindel
#endif
