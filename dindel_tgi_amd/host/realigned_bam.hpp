// realigned_bam.hpp — "next" row N4, the output half: --outputRealignedBAM of the diploid analysis.
//   the haplotype pair and, per read, the haplotype whose alignment is written     reference DInDel.cpp:589-612
//   computePairLikelihoods (usePrior) + getMaxHap                                  reference DInDel.cpp:3765-3833, :290-318
//   writeRealignedBAMFile                                                          reference DInDel.cpp:670-725
// getCIGAR itself is host/cigar.cpp.  The BAM is written by the own BGZF writer (bam_reader.hpp); libbam is not in the image.
#ifndef DINDEL_REALIGNED_BAM_HPP
#define DINDEL_REALIGNED_BAM_HPP
#include <string>
#include <vector>
#include "bam_reader.hpp"
#include "cigar.hpp"
#include "compute_likelihoods.hpp"
#include "diploid_glf.hpp"

namespace dindel {

// The most likely haplotype pair with priors (the head of computePairLikelihoods' sorted list, which is what getMaxHap returns).
// All pairs take part: filterHaplotypes plays no role here.  Throws std::string if no pair has a finite likelihood.
std::pair<int, int> maxLikelihoodPair(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks,
                                      int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params);

// One CIGAR per read against haplotype h1 or h2 of the pair, whichever explains the read better (within 1e-8: the one with fewer
// indels, h2 on a tie) — DInDel.cpp:596-611.  Needs Haplotype::refHpos and the reads' alignments (a batch run with alignments).
// Throws getCIGAR's strings.
void realignedCigars(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks, std::pair<int, int> pair,
                     int refSeqPos, std::vector<CIGAR> &cigars);

// The name the reference gives the window's file: PREFIX.ra.INDEX_TID_(leftPos+minReadOverlap)_(rightPos-minReadOverlap).bam (:614-618)
std::string realignedBAMFileName(const std::string &prefix, int index, const std::string &tid, uint32_t leftPos, uint32_t rightPos, int minReadOverlap);

// writeRealignedBAMFile: every read of the window in `reads` order; a read placed on a haplotype (onHap) gets its new CIGAR,
// position = cigar.refPos and isize = refPos - mpos (bin and everything else as in the original record); the others are copied.
// Throws "Problem with the cigars.", "Cannot open bamfile ... for writing!"; a read without its record: "Read has no BAM record."
void writeRealignedBAMFile(const std::string &fileName, const std::vector<CIGAR> &cigars, const std::vector<Read> &reads, const std::vector<int> &onHap,
                           const BamFile &header);

} // namespace dindel
#endif
