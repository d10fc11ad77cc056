// glf_to_vcf.hpp — `.glf.txt` -> VCF 4.0, the last step of the diploid workflow.
//
// Restates python/mergeOutputDiploid.py of the reference (a Python-2 script that cannot run in a Python-3-only image):
//   getVCFString             :35-154   REF / ALT construction, SNP-only rows shifted by one, hp / q filters, INFO, GT:GQ
//   processDiploidGLFFile    :158-231  which .glf.txt rows become calls (msg ok, dip.map, candidate, int(float(qual)) >= 1)
//   mergeOutput              :240-317  header lines, chromosome order 1..22, X, Y, then the others
// with its helpers python/utils/Fasta.py (:33-53 indexed FASTA access), AnalyzeSequence.py (HomopolymerLength),
// Variant.py (:3-27 variant-string classes) and FileUtils.py (FileWithHeader: space-separated table with a header line).
#ifndef DINDEL_GLF_TO_VCF_HPP
#define DINDEL_GLF_TO_VCF_HPP
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace dindel {

// python/utils/Fasta.py: FastaIndex (.fai: name, length, offset, bases per line, bytes per line) + Fasta.get
class IndexedFasta {
public:
    explicit IndexedFasta(const std::string &fname);       // opens fname and fname + ".fai"; throws std::string
    ~IndexedFasta();
    // `len` bases from 1-based position pos1 of sequence tid, newlines skipped (Fasta.py:39-53).  Throws std::string("KeyError")
    // for an unknown sequence; stops early at end of file (the script would spin there).
    std::string get(const std::string &tid, long pos1based, int len);
    bool has(const std::string &tid) const { return ft.find(tid) != ft.end(); }
    std::vector<std::string> names() const;                // in .fai order
private:
    struct Target { long len, offset, blen, llen; };
    std::map<std::string, Target> ft;
    std::vector<std::string> order;
    FILE *fa;
    IndexedFasta(const IndexedFasta &); IndexedFasta &operator=(const IndexedFasta &);
};

// AnalyzeSequence.HomopolymerLength: run length around seq[pos]; the leftward scan stops at index 1 (range(pos-1, 0, -1)),
// as in the script
int homopolymerLength(const std::string &seq, int pos);

// Variant.py:3-27
struct VariantString {
    enum Type { DEL, INS, SNP, REF } type;
    std::string seq; int length;
    explicit VariantString(const std::string &s);          // throws std::string("Unrecognized variant: ...")
};

// int(float(x)) of the script: parse as double, truncate towards zero.  Throws std::string on a cell that is not a number.
long intOfFloat(const std::string &cell);

struct GlfCall {                                           // the dict `glf` of processDiploidGLFFile (:190-217)
    std::string chr; long pos; long qual; std::vector<std::string> nref_all;
    long num_cover_forward, num_cover_reverse, num_cover_forward_old, num_cover_reverse_old;
    std::string num_hap_reads, genotype;
};

// getVCFString (:35-154): the VCF data line and the position it is reported at
std::pair<std::string, long> getVCFString(const GlfCall &glf, IndexedFasta &fa, int maxHPLen = 10, int filterQual = 0,
                                          const std::vector<std::string> &addFilters = std::vector<std::string>());

typedef std::map<std::string, std::map<long, std::vector<std::string> > > CallsByChrom;
// processDiploidGLFFile (:158-231): adds the calls of one .glf.txt to `variants`; returns the number of skipped windows
int processDiploidGLFFile(const std::string &glfFile, CallsByChrom &variants, IndexedFasta &fa, int maxHPLen = 10, int filterQual = 20);

// mergeOutput (:240-317): list file of .glf.txt paths -> VCF.  Chromosomes 1..22, X, Y first; the script appends the
// remaining ones in Python-2 dict order, which is not reproducible — here they follow in lexicographic order.
void mergeOutput(const std::string &glfFilesFile, const std::string &sampleID, const std::string &refFile, int maxHPLen,
                 const std::string &vcfFile, int filterQual = 20);

} // namespace dindel
#endif
