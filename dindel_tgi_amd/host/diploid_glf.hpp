// diploid_glf.hpp — the "reduce" half of the window loop: DetInDel::diploidGLF on the records of one window
// (reference DInDel.cpp:2933-3660), producing the window's .glf.txt lines.  Works on the lazy WindowLikelihoods view of a
// batch (compute_likelihoods.hpp): only scalars and covered flags are read, no MLAlignment is materialised.
//   filterHaplotypes                      reference DInDel.cpp:1932-2100 (per-read coverage tests come from the device: var_fcov)
//   getHaplotypePrior / getPairPrior      reference DInDel.cpp:1835-1930
//   MAP haplotype pair, qual, dip.map lines   reference DInDel.cpp:3062-3302
//   per-position genotype likelihoods, "dip" lines   reference DInDel.cpp:3305-3660
#ifndef DINDEL_DIPLOID_GLF_HPP
#define DINDEL_DIPLOID_GLF_HPP
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "compute_likelihoods.hpp"
#include "genotype.hpp"
#include "glf_output.hpp"
#include "window_io.hpp"

namespace dindel {

struct DiploidParameters {                // DetInDel::Parameters fields diploidGLF looks at (defaults: DInDel.cpp:4122-4125, :3975)
    DiploidParameters() : priorSNP(1.0 / 1000.0), priorIndel(1.0 / 10000.0), filterHaplotypes(false), outputGLF(true), quiet(true) {}
    double priorSNP, priorIndel; bool filterHaplotypes, outputGLF, quiet;
};

typedef std::pair<int, AlignedVariant> PAV;

// DetInDel::filterHaplotypes on the view (the per-read test of DInDel.cpp:1973-2054 is the device's var_fcov flag)
void filterHaplotypes(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks,
                      std::vector<int> &filtered, std::map<PAV, VariantCoverage> &varCoverage, bool doFilter);

double getPairPrior(const AlignedVariant &av1, const AlignedVariant &av2, int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params);
double getHaplotypePrior(const Haplotype &h1, const Haplotype &h2, int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params);

// DetInDel::diploidGLF.  Writes the window's lines to glfData; throws the reference's strings ("Could not find indel allele",
// "genotyping error") — the caller turns them into the skipped-window line.
void diploidGLF(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks, uint32_t candPos,
                uint32_t leftPos, uint32_t rightPos, OutputData &glfData, int index, const std::string &tid,
                const AlignedCandidates &candidateVariants, const DiploidParameters &params, const std::string &program = "dip");

} // namespace dindel
#endif
