// cigar.hpp — "next" row N4 (second half): the CIGAR of a read realigned through a candidate haplotype.
//
// Mirror of DetInDel::getCIGAR (reference DInDel.cpp:728-882): the read's hpos (position of every read base on the
// haplotype, from the likelihood path) is composed with the haplotype's own alignment to the reference sequence
// (Haplotype::ml.hpos, produced by the off-path alignHaplotypes step) and run-length encoded into BAM operations.
// The BAM record is written by realigned_bam.cpp.
#ifndef DINDEL_CIGAR_HPP
#define DINDEL_CIGAR_HPP
#include <utility>
#include <vector>
#include "dindel_types.hpp"

namespace dindel {

enum { CIG_MATCH = 0, CIG_INS = 1, CIG_DEL = 2, CIG_SOFT_CLIP = 4 };   // BAM_CMATCH, BAM_CINS, BAM_CDEL, BAM_CSOFT_CLIP

struct CIGAR : public std::vector<std::pair<int, int> > {   // (operation, length) — reference DInDel.hpp:168-173
    typedef std::pair<int, int> CIGOp;
    int refPos;                                              // reference position of the first aligned base; -1 if none
    CIGAR() : refPos(-1) {}
};

// hapRefPos = hap.ml.hpos (one entry per haplotype base: reference offset, or a negative MLAlignment code).
// Throws the reference's strings: "Haplotype has not been aligned!", "Read is not properly aligned!", "Error(n)!",
// "How is this possible? (1)".
CIGAR getCIGAR(const std::vector<int> &hapRefPos, size_t hapSize, const MLAlignment &ml, size_t readSize, int refSeqStart);

} // namespace dindel
#endif
