// compute_likelihoods.hpp — C++ host adapter with the reference's operator interface for the hot path.
//
//   void DetInDel::computeLikelihoods(const vector<Haplotype>&, const vector<Read>&,
//                                     vector<vector<MLAlignment> >& liks, uint32_t leftPos, uint32_t rightPos,
//                                     vector<int>& onHap);                 reference DInDel.hpp:136, DInDel.cpp:1707-1739
//
// Same argument meaning, same outputs (liks[hidx][r], onHap[r]) and the same error behaviour:
//   * haplotype shorter than maxLengthDel  -> throws std::string("hapSize error.")   (ObservationModelFB.cpp:47)
//   * NaN / Inf log-likelihood             -> throws std::string("Nan detected")     (DInDel.cpp:1732-1735)
//   * log-likelihood > 0.1                 -> "Likelihood>0" on stderr, exit(1)      (DInDel.cpp:1722-1731)
//   * window shape outside the kernel limits (haplotype > 766 bp, read > 1024 bp, empty sequence; no reference
//     counterpart) -> throws std::string("window outside the GPU kernel limits ...") for THAT window only
//                                             (setThrowOnPositiveLikelihood(true) turns the exit into a throw)
// All arithmetic runs on the GPU through the C ABI (include/dindel_hmm.h); this class only packs the
// windows, calls dd_compute_likelihoods and rebuilds the MLAlignment records (variant strings from hpos).
#ifndef DINDEL_COMPUTE_LIKELIHOODS_HPP
#define DINDEL_COMPUTE_LIKELIHOODS_HPP
#include <cstdint>
#include <vector>
#include "dindel_types.hpp"

namespace dindel {

struct WindowJob {                      // one call of the reference's computeLikelihoods
    const std::vector<Haplotype> *haps;
    const std::vector<Read> *reads;
    uint32_t leftPos, rightPos;
    std::vector<std::vector<MLAlignment> > *liks;   // OUT
    std::vector<int> *onHap;                        // OUT
    std::string error;                              // OUT: empty, or the string the reference would have thrown
};

class LikelihoodEngine {
public:
    explicit LikelihoodEngine(const ObservationModelParameters &obsParams, int device = 0)
        : params(obsParams), device_(device), throwOnPositive_(false), hostThreads_(0) {}

    ObservationModelParameters params;   // the reference passes this->params.obsParams implicitly (DInDel.cpp:1718)

    // drop-in for one window
    void computeLikelihoods(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                            std::vector<std::vector<MLAlignment> > &liks, uint32_t leftPos, uint32_t rightPos,
                            std::vector<int> &onHap);

    // the batched form a GPU wants: prepare-N -> one launch -> reduce-N in order.  A window that would have
    // thrown gets its message in WindowJob::error (the caller turns it into the "skipped" row, DInDel.cpp:1361-1408).
    void computeLikelihoodsBatch(std::vector<WindowJob> &jobs);

    // DetInDel::computeLikelihoodsFaster (reference DInDel.cpp:1790-1833): the --faster model, ObservationModelS.
    // Throws std::string("hapSize error.") (Faster.cpp:47) or std::string("HapHash string too short")
    // (Haplotype.hpp:341, read shorter than the 4-mer); no likelihood checks, as in the reference.
    void computeLikelihoodsFaster(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                                  std::vector<std::vector<MLAlignment> > &liks, uint32_t leftPos, uint32_t rightPos,
                                  std::vector<int> &onHap);
    void computeLikelihoodsFasterBatch(std::vector<WindowJob> &jobs);

    void setThrowOnPositiveLikelihood(bool v) { throwOnPositive_ = v; }
    // host threads rebuilding the MLAlignment records of a batch (0 = min(16, hardware threads))
    void setHostThreads(int n) { hostThreads_ = n; }

    // wall time of the last batch call's three stages (tools/host_adapter_bench.cpp)
    double lastPackSeconds = 0.0, lastDeviceSeconds = 0.0, lastUnpackSeconds = 0.0;

    // ObservationModelFBMax::reportVariants (ObservationModelFB.cpp:1351-1475) from the device's hpos: fills
    // hpos, indels, snps, align, firstBase/lastBase, counters and the covered maps.  Exposed for tests.
    static void rebuildAlignment(const Haplotype &hap, const Read &read, const int16_t *hpos,
                                 const ObservationModelParameters &p, MLAlignment &ml);

    // ObservationModelS::reportVariants (Faster.cpp:579-681) from the device's hpos.  The position of an insertion is
    // one past the last on-haplotype base before it (Faster.cpp:552-571); bases the model parks on the last haplotype
    // base because they lie right of the haplotype are indistinguishable from it in hpos and count as that base here.
    static void rebuildAlignmentFaster(const Haplotype &hap, const Read &read, const int16_t *hpos,
                                       const ObservationModelParameters &p, MLAlignment &ml);

private:
    void runBatch(std::vector<WindowJob> &jobs, bool faster);
    int device_;
    bool throwOnPositive_;
    int hostThreads_;
};

} // namespace dindel
#endif
