// cigar.cpp — see cigar.hpp.
#include "cigar.hpp"
#include <string>

namespace dindel {

CIGAR getCIGAR(const std::vector<int> &hapRefPos, size_t hapSize, const MLAlignment &ml, size_t readSize, int refSeqStart)
{
    if (hapRefPos.size() != hapSize) throw std::string("Haplotype has not been aligned!");     // DInDel.cpp:730
    if (ml.hpos.size() != readSize) throw std::string("Read is not properly aligned!");        // :731
    const int n = int(readSize);
    // position of every read base on the reference: through the haplotype if the base sits on it (:745-748)
    std::vector<int> onRef(size_t(n > 0 ? n : 0));
    for (int b = 0; b < n; b++) onRef[size_t(b)] = ml.hpos[size_t(b)] >= 0 ? hapRefPos[size_t(ml.hpos[size_t(b)])] : ml.hpos[size_t(b)];

    CIGAR cig;
    int last = n - 1;                                   // last base with a reference position (:772-774)
    while (last >= 0 && onRef[size_t(last)] < 0) last--;
    if (last < 0) {                                     // nothing aligned: the whole read is soft-clipped (:776-780)
        cig.push_back(CIGAR::CIGOp(CIG_SOFT_CLIP, n));
        return cig;
    }
    int b = 0;                                          // leading bases without a reference position are clipped (:788-790)
    while (onRef[size_t(b)] < 0) b++;
    if (b > 0) cig.push_back(CIGAR::CIGOp(CIG_SOFT_CLIP, b));
    int anchor = onRef[size_t(b)];                      // reference position of the last base that was aligned to it
    cig.refPos = refSeqStart + anchor;

    int op = CIG_MATCH, len = 1;                        // the run being built
    for (; b < last; b++) {
        const int here = onRef[size_t(b)], next = onRef[size_t(b + 1)];
        if (next == MLAlignment::INS) {
            if (here == MLAlignment::INS) {             // the insertion goes on (:806-809)
                if (op != CIG_INS) throw std::string("Error(1)!");
                len++;
            } else if (here >= 0) {                     // reference -> insertion (:811-821)
                if (op != CIG_MATCH) throw std::string("Error(2)!");
                cig.push_back(CIGAR::CIGOp(CIG_MATCH, len));
                op = CIG_INS; len = 1;
                anchor = here;
            } else throw std::string("How is this possible? (1)");
        } else if (here >= 0 && next >= 0 && next - here == 1) {          // consecutive reference bases (:825-832)
            if (op != CIG_MATCH) throw std::string("Error(3)!");
            len++;
            anchor = next;
        } else if (here >= 0 && next >= 0 && next - here > 1) {           // reference bases skipped: deletion (:833-845)
            if (op != CIG_MATCH) throw std::string("Error(4)!");
            cig.push_back(CIGAR::CIGOp(CIG_MATCH, len));
            cig.push_back(CIGAR::CIGOp(CIG_DEL, next - here - 1));
            op = CIG_MATCH; len = 1;
            anchor = next;
        } else if (here == MLAlignment::INS && next - anchor == 1) {      // insertion -> next reference base (:846-854)
            cig.push_back(CIGAR::CIGOp(CIG_INS, len));
            op = CIG_MATCH; len = 1;
            anchor = next;
        } else if (here == MLAlignment::INS && next - anchor > 1) {       // insertion followed by a deletion (:855-865)
            cig.push_back(CIGAR::CIGOp(CIG_INS, len));
            cig.push_back(CIGAR::CIGOp(CIG_DEL, next - anchor - 1));
            op = CIG_MATCH; len = 1;
            anchor = next;
        }
        // any other pair of codes leaves the run untouched, as in the reference (no final else there)
    }
    cig.push_back(CIGAR::CIGOp(op, len));               // (:870)
    if (n - 1 - last > 0) cig.push_back(CIGAR::CIGOp(CIG_SOFT_CLIP, n - 1 - last));   // trailing clip (:873-875)
    return cig;
}

} // namespace dindel
