// genotype.cpp — see genotype.hpp.
#include "genotype.hpp"
#include <cmath>
#include <string>

namespace dindel {

double addLogs(const double l1, const double l2)
{
    if (l1 > l2) {
        double diff = l2 - l1;
        return l1 + log(1.0 + exp(diff));
    } else {
        double diff = l1 - l2;
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

} // namespace dindel
