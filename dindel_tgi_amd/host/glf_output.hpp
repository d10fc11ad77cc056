// glf_output.hpp — the `.glf.txt` output surface of `dindel --analysis indels --doDiploid`.
//
// The reference writes one space-separated table per run: a header line with the column labels and one line per call or
// skipped window; every cell defaults to "NA" and is formatted with the default ostream operator<< (doubles: 6 significant
// digits, "%g" style).  Mirrors
//   OutputData / OutputData::Line                     reference OutputData.hpp:28-114
//   DetInDelParameters::makeGLFOutputData()           reference DInDel.hpp:262-276 (the columns of .glf.txt)
//   the "skipped window" line                         reference DInDel.cpp:1361-1401
//   the dip.map line of DetInDel::diploidGLF          reference DInDel.cpp:3277-3301
//   the per-position "dip" line                       reference DInDel.cpp:3616-3650
// Written from scratch; no Boost / StringHash dependency.
#ifndef DINDEL_GLF_OUTPUT_HPP
#define DINDEL_GLF_OUTPUT_HPP
#include <cstdint>
#include <cstdio>
#include <map>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>

namespace dindel {

class OutputData {
public:
    explicit OutputData(std::ostream &o) : out(&o), numLines(0) {}
    OutputData &operator()(const std::string &label)
    {
        if (labelToColumn.find(label) != labelToColumn.end()) throw std::string("Duplicate label ").append(label);   // :38
        labelToColumn[label] = int(labels.size());
        labels.push_back(label);
        return *this;
    }
    std::string headerString() const
    {
        std::string s;
        for (size_t x = 0; x < labels.size(); x++) { if (x) s += ' '; s += labels[x]; }
        return s;
    }
    template <class T> void outputLine(T x) { *out << x << std::endl; }

    class Line {
    public:
        explicit Line(const OutputData &od) : lineData(od.labelToColumn.size(), "NA"), labelToColumnPtr(&od.labelToColumn) {}
        std::string get(const std::string &columnLabel) const
        {
            std::map<std::string, int>::const_iterator it = labelToColumnPtr->find(columnLabel);
            if (it == labelToColumnPtr->end()) throw std::string("Column label ").append(columnLabel).append(" not found!");
            return lineData[size_t(it->second)];
        }
        // A cell holds what `stringstream << x` gives with default formatting (the reference makes one stringstream per cell,
        // :82-84).  The common types are formatted directly — constructing a stream copies the global locale, whose reference
        // count every reducing thread would be hammering ~30 times per line: integers as decimal digits, doubles as "%.6g"
        // (what num_put does for the default float format and precision), strings as they are.
        template <class T> Line &set(const std::string &columnLabel, T x)
        {
            static thread_local std::ostringstream os;
            os.str(std::string());
            os.clear();
            os << x;
            cell(columnLabel) = os.str();
            return *this;
        }
        Line &set(const std::string &columnLabel, double x)
        {
            char buf[64];
            snprintf(buf, sizeof(buf), "%.6g", x);
            cell(columnLabel) = buf;
            return *this;
        }
        Line &set(const std::string &columnLabel, int x) { cell(columnLabel) = std::to_string(x); return *this; }
        Line &set(const std::string &columnLabel, unsigned x) { cell(columnLabel) = std::to_string(x); return *this; }
        Line &set(const std::string &columnLabel, long x) { cell(columnLabel) = std::to_string(x); return *this; }
        Line &set(const std::string &columnLabel, unsigned long x) { cell(columnLabel) = std::to_string(x); return *this; }
        Line &set(const std::string &columnLabel, const std::string &x) { cell(columnLabel) = x; return *this; }
        Line &set(const std::string &columnLabel, const char *x) { cell(columnLabel) = x; return *this; }
        std::string toString() const
        {
            std::string s;
            for (size_t x = 0; x < lineData.size(); x++) { if (x) s += ' '; s += lineData[x]; }
            return s;
        }
        std::vector<std::string> lineData;
    private:
        std::string &cell(const std::string &columnLabel)
        {
            std::map<std::string, int>::const_iterator it = labelToColumnPtr->find(columnLabel);
            if (it == labelToColumnPtr->end()) throw std::string("Column label ").append(columnLabel).append(" not found!");
            return lineData[size_t(it->second)];
        }
        const std::map<std::string, int> *labelToColumnPtr;
    };

    void output(const Line &line)
    {
        numLines++;
        *out << line.toString() << std::endl;
    }
    int lines() const { return numLines; }
    std::ostream *out;

private:
    std::map<std::string, int> labelToColumn;
    std::vector<std::string> labels;
    int numLines;
};

// DetInDelParameters::makeGLFOutputData — reference DInDel.hpp:262-276
inline OutputData makeGLFOutputData(std::ostream &out)
{
    OutputData oData(out);
    oData("msg")("index");
    oData("analysis_type");
    oData("tid")("lpos")("rpos")("center_position")("realigned_position")("was_candidate_in_window");
    oData("ref_all")("nref_all")("num_reads");
    oData("post_prob_variant")("qual")("est_freq")("logZ")("hapfreqs");
    oData("indidx")("msq")("numOffAll")("num_indel")("num_cover_forward")("num_cover_reverse")("num_unmapped_realigned");
    oData("var_coverage_forward")("var_coverage_reverse");
    oData("nBQT")("nmmBQT")("mLogBQ")("nMMLeft")("nMMRight");
    oData("glf");
    return oData;
}

// DetInDelParameters::makeOutputData — reference DInDel.hpp:245-260 (the per-run table the diploid analysis leaves empty)
inline OutputData makeOutputData(std::ostream &out)
{
    OutputData oData(out);
    oData("msg")("index");
    oData("analysis_type");
    oData("tid")("lpos")("rpos")("center_position")("realigned_position");
    oData("ref_all")("num_reads")("num_hqreads");
    oData("post_prob_variant")("est_freq")("was_candidate_in_window");
    oData("num_mapped_to_first")("num_mapped_to_second");
    oData("num_off_hap")("loglik_hap_pair")("loglik_next_hap_pair");
    oData("first_var_cover_forward")("first_var_cover_reverse")("second_var_cover_forward")("second_var_cover_reverse");
    oData("first_called_all")("second_called_all")("loglik_called_genotype")("loglik_ref_ref")("alt_genotypes");
    return oData;
}

// The message of a window the caller skipped: "error_" + what was thrown, blanks turned into '_' — reference DInDel.cpp:1366-1368
inline std::string skippedMessage(std::string thrown)
{
    for (size_t x = 0; x < thrown.size(); x++) if (thrown[x] == ' ') thrown[x] = '_';
    return std::string("error_").append(thrown);
}

// the line of a skipped window — reference DInDel.cpp:1389-1395
inline OutputData::Line skippedWindowLine(const OutputData &glfData, const std::string &message, int index, const std::string &tid,
                                          uint32_t leftPos, uint32_t rightPos)
{
    OutputData::Line gline(glfData);
    gline.set("msg", message);
    gline.set("index", index);
    gline.set("tid", tid);
    gline.set("lpos", leftPos);
    gline.set("rpos", rightPos);
    return gline;
}

// values of one "dip.map" call line (one per variant position of the MAP haplotype pair) — reference DInDel.cpp:3277-3301
struct DipMapCall {
    int index; std::string tid; uint32_t leftPos, rightPos, candPos; int realignedPos /* it->first + leftPos */; int was_candidate;
    double qual; std::string nref_all; size_t num_reads; double msq; int numf, numr, vc_f, vc_r, numUnmappedRealigned;
    std::string genotype; double genoqual;
};
inline OutputData::Line dipMapLine(const OutputData &glfData, const DipMapCall &c)
{
    char genoqual[64];
    snprintf(genoqual, sizeof(genoqual), "%.6g", c.genoqual);       // glfs << genotype << ":" << genoqual, :3270
    OutputData::Line line(glfData);
    line.set("msg", "ok");
    line.set("index", c.index);
    line.set("tid", c.tid);
    line.set("analysis_type", std::string("dip.map"));
    line.set("indidx", 0);
    line.set("lpos", c.leftPos);
    line.set("rpos", c.rightPos);
    line.set("center_position", c.candPos);
    line.set("realigned_position", c.realignedPos);
    line.set("was_candidate_in_window", c.was_candidate);
    line.set("qual", c.qual);
    line.set("nref_all", c.nref_all);
    line.set("num_reads", c.num_reads);
    line.set("msq", c.msq);
    line.set("num_cover_forward", c.numf);
    line.set("num_cover_reverse", c.numr);
    line.set("var_coverage_forward", c.vc_f);
    line.set("var_coverage_reverse", c.vc_r);
    line.set("num_unmapped_realigned", c.numUnmappedRealigned);
    line.set("glf", c.genotype + ":" + genoqual);
    return line;
}

// values of one per-position "dip" line — reference DInDel.cpp:3616-3650
struct DipPositionRow {
    int index; std::string tid, program; uint32_t leftPos, rightPos, candPos; int realignedPos; int has_variants_in_window; double logZ;
    int nBQT, nmmBQT; double mLogBQ /* sum over the reads; written as mLogBQ / nBQT */; int nMMLeft, nMMRight; std::string nref_all;
    size_t num_reads; double msq; int numOffAll, num_indel, nf, nr; std::string var_coverage_forward, var_coverage_reverse, glf;
    int numUnmappedRealigned;
};
inline OutputData::Line dipPositionLine(const OutputData &glfData, const DipPositionRow &c)
{
    OutputData::Line line(glfData);
    line.set("msg", "ok");
    line.set("index", c.index);
    line.set("tid", c.tid);
    line.set("analysis_type", c.program);
    line.set("indidx", 0);
    line.set("lpos", c.leftPos);
    line.set("rpos", c.rightPos);
    line.set("center_position", c.candPos);
    line.set("realigned_position", c.realignedPos);
    line.set("was_candidate_in_window", c.has_variants_in_window);
    line.set("logZ", c.logZ);
    line.set("nBQT", c.nBQT);
    line.set("nmmBQT", c.nmmBQT);
    line.set("mLogBQ", c.mLogBQ / double(c.nBQT));                    // :3635 (nan / inf when nBQT == 0, printed as the stream prints them)
    line.set("nMMLeft", c.nMMLeft);
    line.set("nMMRight", c.nMMRight);
    line.set("nref_all", c.nref_all);
    line.set("num_reads", c.num_reads);
    line.set("msq", c.msq);
    line.set("numOffAll", c.numOffAll);
    line.set("num_indel", c.num_indel);
    line.set("num_cover_forward", c.nf);
    line.set("num_cover_reverse", c.nr);
    line.set("var_coverage_forward", c.var_coverage_forward);
    line.set("var_coverage_reverse", c.var_coverage_reverse);
    line.set("glf", c.glf);
    line.set("num_unmapped_realigned", c.numUnmappedRealigned);
    return line;
}

} // namespace dindel
#endif
