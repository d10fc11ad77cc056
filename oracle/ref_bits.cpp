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
//   VariantFile.hpp          AlignedCandidates, VariantFile::getLineVector — the realignment-window file parser (N2; round 4)
// This file only includes them (path given by -I on the command line, see Makefile target _ref) and exports C wrappers;
// the library goes to oracle/_ref/ and is used by tests/test_ref_bits.py to check the restatement and the host tables
// against the reference's own code.  TEST INFRASTRUCTURE ONLY.
#include <cmath>
#include <cstring>
#include <stdint.h>
#include <sstream>
#include <string>
#include <vector>
using namespace std;          // ReadIndelErrorModel.hpp relies on the including file for <vector> and the using-directive
#include "ReadIndelErrorModel.hpp"
#include "Utils.hpp"
#include "Variant.hpp"
#include "MLAlignment.hpp"
#include "ObservationModel.hpp"
#include "VariantFile.hpp"

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

// The window file as the reference's own parser reads it (VariantFile.hpp:188-289): one JSON entry per getLineVector() call of the
// loop `while (!vf.eof())` (DInDel.cpp:1310-1322) — the candidates (position, string, end, type, length, sequence, prior, add-combinatorially)
// or "skipped" for an empty result; a throw ends the list and is reported ("Cannot read left boundary of region." reaches the caller of the
// reference's loop too, DInDel.cpp:1316).  Written in the format of dindel_tgi_amd/host/host_capi.cpp:ddh_window_lines_json, which does the
// same with this repository's parser: tests/test_ref_bits.py compares the two texts.
int ref_window_lines_json(const char *path, int oneBased, char *out, int cap)
{
    ostringstream os;
    os.precision(17);
    os << "{\"calls\":[";
    string thrown;
    bool threw = false;
    try {
        VariantFile vf(path);
        bool first = true;
        while (!vf.eof()) {
            AlignedCandidates c = vf.getLineVector(oneBased != 0);
            os << (first ? "" : ",");
            first = false;
            if (c.variants.empty()) { os << "\"skipped\""; continue; }
            os << "{\"tid\":\"" << c.tid << "\",\"leftPos\":" << c.leftPos << ",\"rightPos\":" << c.rightPos << ",\"centerPos\":" << c.centerPos << ",\"variants\":[";
            for (size_t i = 0; i < c.variants.size(); i++) {
                const AlignedVariant &v = c.variants[i];
                os << (i ? "," : "") << "[" << v.getStartHap() << ",\"" << v.getString() << "\"," << v.getEndHap() << "," << int(v.getType()) << "," << v.size() << ",\"" << v.getSeq()
                   << "\"," << v.getFreq() << "," << int(v.getAddComb()) << "]";
            }
            os << "]}";
        }
    } catch (string &e) { threw = true; thrown = e; }
    os << "]";
    if (threw) os << ",\"throw\":\"" << thrown << "\"";
    os << "}";
    const string t = os.str();
    if (int(t.size()) + 1 > cap) return -int(t.size()) - 1;
    memcpy(out, t.c_str(), t.size() + 1);
    return int(t.size());
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
