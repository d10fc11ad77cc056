// genotype.hpp — host side of "next" row N1: the MAP haplotype-pair step of DetInDel::diploidGLF
// (reference DInDel.cpp:3062-3120) on top of the device read-sums (dd_pair_sums, genotype_kernel.hip).
#ifndef DINDEL_GENOTYPE_HPP
#define DINDEL_GENOTYPE_HPP
#include <vector>

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

} // namespace dindel
#endif
