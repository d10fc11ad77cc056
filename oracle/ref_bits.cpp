// ref_bits.cpp — the parts of the reference that DO build in this image, compiled from where they lie.
//
// The hot-path translation units (ObservationModelFB.cpp, Faster.cpp) need samtools' bam.h and Boost, which the image
// lacks, so there is no reference build of the path itself.  Three small headers on or next to the path use
// the standard library only and compile as they are:
//   ReadIndelErrorModel.hpp  getViterbiHPError    — the homopolymer indel-error model behind logProbError[] (A3)
//   Utils.hpp                addLogs              — the genotype read-sum term (N1)
//   Variant.hpp              AlignedVariant       — string forms, isCovered (hapIndelCovered / hapSNPCovered, A10)
//   ObservationModel.hpp     ObservationModelParameters::setDefaultValues — the struct defaults behind dd_params
//   MLAlignment.hpp          the hpos codes INS / DEL / LO / RO and what the constructor zeroes (A11)
// This file only includes them (path given by -I on the command line, see Makefile target _ref) and exports C wrappers;
// the library goes to oracle/_ref/ and is used by tests/test_ref_bits.py to check the restatement and the host tables
// against the reference's own code.  TEST INFRASTRUCTURE ONLY.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
using namespace std;          // ReadIndelErrorModel.hpp relies on the including file for <vector> and the using-directive
#include "ReadIndelErrorModel.hpp"
#include "Utils.hpp"
#include "Variant.hpp"
#include "MLAlignment.hpp"
#include "ObservationModel.hpp"

extern "C" {

double ref_hp_error(int hpLen)
{
    ReadIndelErrorModel m;
    return m.getViterbiHPError(hpLen);
}

double ref_add_logs(double l1, double l2) { return addLogs(l1, l2); }

// returns isCovered; writes type (0 INS, 1 DEL, 2 SNP, 3 REF as in Variant::Type), length, and the sequence
int ref_aligned_variant(const char *str, int startHap, int endHap, int startRead, int endRead, int pad, int firstBase, int lastBase,
                        int *type, int *length, char *seq, int cap)
{
    *type = -9; *length = -9; seq[0] = 0;
    try {
    AlignedVariant av(string(str), startHap, endHap, startRead, endRead);
    *type = int(av.getType());
    *length = av.size();
    strncpy(seq, av.getSeq().c_str(), size_t(cap - 1));
    seq[cap - 1] = 0;
    return av.isCovered(pad, firstBase, lastBase) ? 1 : 0;
    } catch (string &) { return -1; }      // "Unrecognized variant" (Variant.hpp:68)
}


// ObservationModelParameters() — ObservationModel.hpp:31-64: d = {pError, pMut, pFirstgLO, mapQualThreshold, checkBaseQualThreshold,
// capMapQualFast}, i = {maxLengthDel, maxLengthIndel, padCover, bMid, forceReadOnHaplotype, mapUnmappedReads, maxMismatch}
void ref_obs_params_defaults(double *d, int *i)
{
    ObservationModelParameters p;
    d[0] = p.pError; d[1] = p.pMut; d[2] = p.pFirstgLO; d[3] = p.mapQualThreshold; d[4] = p.checkBaseQualThreshold; d[5] = p.capMapQualFast;
    i[0] = p.maxLengthDel; i[1] = p.maxLengthIndel; i[2] = p.padCover; i[3] = p.bMid; i[4] = p.forceReadOnHaplotype; i[5] = p.mapUnmappedReads;
    i[6] = p.maxMismatch;
}

// MLAlignment codes and constructor values — MLAlignment.hpp:31-46
void ref_mlalignment(int *codes, double *d, int *i)
{
    MLAlignment ml;
    codes[0] = MLAlignment::INS; codes[1] = MLAlignment::DEL; codes[2] = MLAlignment::LO; codes[3] = MLAlignment::RO;
    d[0] = ml.ll; d[1] = ml.llOn; d[2] = ml.llOff;
    i[0] = ml.offHap; i[1] = ml.offHapHMQ; i[2] = ml.numIndels; i[3] = ml.numMismatch; i[4] = ml.relPos;
}

}
