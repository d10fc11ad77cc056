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
//                                             (setThrowOnPositiveLikelihood(true) turns the exit into a throw)
//   * window shape outside the kernel limits (haplotype > 766 bp, read > 1024 bp, empty sequence; no reference
//     counterpart) -> throws std::string("window outside the GPU kernel limits ...") for THAT window only
// All arithmetic runs on the GPU through the C ABI (include/dindel_hmm.h); this class only packs the
// windows, calls dd_compute_likelihoods and hands the records back.
//
// Two ways to receive a batch's records:
//   * eager  — WindowJob::liks / onHap point at the reference's containers: every MLAlignment (five std::map's, two
//              strings, a vector) is rebuilt.  The literal drop-in; the host spends ~0.1 us per read base on it.
//   * lazy   — WindowJob::liks == NULL: the job's WindowLikelihoods view exposes every scalar of every record straight from
//              the batch's result block (what the reductions of DInDel.cpp:2933-3660 read: ll, offHap, numIndels, the QC
//              counters, the covered flags) and builds a full MLAlignment only for the pairs a consumer asks for (get()).
#ifndef DINDEL_COMPUTE_LIKELIHOODS_HPP
#define DINDEL_COMPUTE_LIKELIHOODS_HPP
#include <cstdint>
#include <memory>
#include <vector>
#include "dindel_types.hpp"

namespace dindel {

class LikelihoodEngine;
struct BatchBlock;                       // flat result arrays of one batch call (compute_likelihoods.cpp)
struct PackScratch;                      // flat input arrays of one batch call

// Read-only view of one window's records inside a batch's result block.  Cheap to copy; keeps the block alive.
class WindowLikelihoods {
public:
    WindowLikelihoods() : w_(-1), haps_(NULL), reads_(NULL), leftPos_(0), eng_(NULL) {}
    bool valid() const { return w_ >= 0 && bool(blk_); }
    size_t numHaps() const;
    size_t numReads() const;
    // the scalars of liks[h][r] (MLAlignment.hpp:35-75)
    double ll(size_t h, size_t r) const;
    double llOn(size_t h, size_t r) const;
    double llOff(size_t h, size_t r) const;
    double mLogBQ(size_t h, size_t r) const;
    bool offHap(size_t h, size_t r) const;
    bool offHapHMQ(size_t h, size_t r) const;
    int numIndels(size_t h, size_t r) const;      // == liks[h][r].indels.size() for the main model (one key per event)
    int indelCount(size_t h, size_t r) const;     // liks[h][r].indels.size() for either model (the --faster model leaves numIndels 0
                                                  // and may key several events at one position): what DInDel.cpp:3529 reads
    int numMismatch(size_t h, size_t r) const;
    int nBQT(size_t h, size_t r) const;
    int nmmBQT(size_t h, size_t r) const;
    int nMMLeft(size_t h, size_t r) const;
    int nMMRight(size_t h, size_t r) const;
    int firstBase(size_t h, size_t r) const;
    int lastBase(size_t h, size_t r) const;
    int onHap(size_t r) const;                    // DInDel.cpp:1720
    // liks[h][r].hapIndelCovered[key] / hapSNPCovered[key] (false when the haplotype has no such variant, like the
    // reference's find()-then-test at DInDel.cpp:3158, :3539) and the filterHaplotypes coverage flag of a haplotype indel
    bool hapIndelCovered(size_t h, size_t r, int key) const;
    bool hapSNPCovered(size_t h, size_t r, int key) const;
    bool hapIndelFilterCovered(size_t h, size_t r, int key) const;
    // the same flags by slot, for loops over the reads of one (haplotype, variant): slot = varSlot(h, key, snp) once,
    // then coveredAt / filterCoveredAt per read (slot < 0: false)
    int varSlot(size_t h, int key, bool snp) const;
    bool coveredAt(size_t h, size_t r, int slot) const;
    bool filterCoveredAt(size_t h, size_t r, int slot) const;
    // One haplotype's scalars as plain rows indexed by read (x[r] == x(h, r)), and its covered flags: flag of variant slot s for read r
    // is vcov[r * nv + s] (fcov likewise).  For loops over the reads; valid as long as the view is.
    struct Rows {
        const double *ll, *llOn, *llOff, *mLogBQ;
        const uint8_t *offHap, *offHapHMQ, *vcov, *fcov;
        const int16_t *numIndels, *numMismatch, *nBQT, *nmmBQT, *nMMLeft, *nMMRight, *firstBase, *lastBase;
        int nv;
        bool covered(size_t r, int slot) const { return slot >= 0 && vcov[r * size_t(nv) + size_t(slot)] != 0; }
        bool filterCovered(size_t r, int slot) const { return slot >= 0 && fcov[r * size_t(nv) + size_t(slot)] != 0; }
    };
    Rows rows(size_t h) const;
    // the full record of one pair, built on demand (variant maps, align string, hpos).  If the batch was run without
    // alignments (setKeepAlignments(false)) the window is recomputed once through the engine and cached.
    MLAlignment get(size_t h, size_t r) const;
    // all records, as the eager path fills them
    void toLiks(std::vector<std::vector<MLAlignment> > &liks, std::vector<int> &onHap) const;

private:
    friend class LikelihoodEngine;
    int64_t pair(size_t h, size_t r) const;
    std::shared_ptr<const BatchBlock> blk_;
    mutable std::shared_ptr<const BatchBlock> full_;   // recomputed one-window block (no-alignment batches)
    int w_;
    const std::vector<Haplotype> *haps_;
    const std::vector<Read> *reads_;
    uint32_t leftPos_;
    LikelihoodEngine *eng_;
};

struct WindowJob {                      // one call of the reference's computeLikelihoods
    WindowJob() : haps(NULL), reads(NULL), leftPos(0), rightPos(0), liks(NULL), onHap(NULL) {}
    const std::vector<Haplotype> *haps;
    const std::vector<Read> *reads;
    uint32_t leftPos, rightPos;
    std::vector<std::vector<MLAlignment> > *liks;   // OUT (eager); NULL = lazy: use `result`
    std::vector<int> *onHap;                        // OUT (eager); may be NULL
    WindowLikelihoods result;                       // OUT: always set (unless the window has an error)
    std::string error;                              // OUT: empty, or the string the reference would have thrown
};

class LikelihoodEngine {
public:
    explicit LikelihoodEngine(const ObservationModelParameters &obsParams, int device = 0)
        : params(obsParams), device_(device), throwOnPositive_(false), hostThreads_(0), keepAlignments_(true) {}

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
    // host threads packing the windows / rebuilding eager records (0 = min(16, hardware threads))
    void setHostThreads(int n) { hostThreads_ = n; }
    // false: the per-base alignments (hpos, 45 % of the result bytes) stay on the device side of the call; lazy views
    // recompute a window on the first get().  Scalars, covered flags and onHap are always delivered.
    void setKeepAlignments(bool v) { keepAlignments_ = v; }
    // optional: have the calling thread's device cache (arena, staging mirror, streams) of this engine's device made now, for batches of
    // about `pairs` (haplotype, read) pairs — e.g. while the first batch is still being prepared.  Throws like the batch calls.
    void warmUp(size_t pairs);

    // wall time of the last batch call's three stages (tools/host_adapter_bench.cpp, bench.py)
    double lastPackSeconds = 0.0, lastDeviceSeconds = 0.0, lastUnpackSeconds = 0.0;

    // ObservationModelFBMax::reportVariants (ObservationModelFB.cpp:1351-1475) from the device's hpos: fills
    // hpos, indels, snps, align, firstBase/lastBase, counters and the covered maps.  Exposed for tests.
    static void rebuildAlignment(const Haplotype &hap, const Read &read, const int16_t *hpos,
                                 const ObservationModelParameters &p, MLAlignment &ml);

    // ObservationModelS::reportVariants (Faster.cpp:579-681) from the device's hpos (inserted bases carry the key the
    // model assigned them, Faster.cpp:552-571, so overhanging bases parked on the last haplotype base do not confuse it).
    static void rebuildAlignmentFaster(const Haplotype &hap, const Read &read, const int16_t *hpos,
                                       const ObservationModelParameters &p, MLAlignment &ml);

private:
    friend class WindowLikelihoods;
    void runBatch(std::vector<WindowJob> &jobs, bool faster);
    int device_;
    bool throwOnPositive_;
    int hostThreads_;
    bool keepAlignments_;
    std::vector<std::shared_ptr<BatchBlock> > spare_;   // result blocks of earlier calls; one nobody references any more is reused (warm pages)
    unsigned spareNext_ = 0;
    std::shared_ptr<PackScratch> scratch_;   // the packed inputs' buffers, reused between calls
};

} // namespace dindel
#endif
