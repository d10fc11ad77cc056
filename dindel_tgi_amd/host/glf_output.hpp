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
    // a line already in its final text (GlfText below), without the trailing newline
    void outputText(const std::string &text)
    {
        numLines++;
        out->write(text.data(), std::streamsize(text.size()));
        out->put('\n');
        out->flush();
    }
    // true when the columns are exactly makeGLFOutputData's, in that order: the window loop then writes its two kinds of lines
    // through GlfText (one pass over the 32 cells) instead of through a Line (a label lookup and a string per cell)
    bool hasGLFColumns() const { return glfColumns; }
    void noteColumns(const char *const *expected, size_t n)
    {
        glfColumns = labels.size() == n;
        for (size_t i = 0; glfColumns && i < n; i++) glfColumns = labels[i] == expected[i];
    }
    int lines() const { return numLines; }
    std::ostream *out;

private:
    std::map<std::string, int> labelToColumn;
    std::vector<std::string> labels;
    int numLines;
    bool glfColumns = false;
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
    static const char *const columns[32] = {"msg", "index", "analysis_type", "tid", "lpos", "rpos", "center_position", "realigned_position", "was_candidate_in_window",
        "ref_all", "nref_all", "num_reads", "post_prob_variant", "qual", "est_freq", "logZ", "hapfreqs", "indidx", "msq", "numOffAll", "num_indel", "num_cover_forward",
        "num_cover_reverse", "num_unmapped_realigned", "var_coverage_forward", "var_coverage_reverse", "nBQT", "nmmBQT", "mLogBQ", "nMMLeft", "nMMRight", "glf"};
    oData.noteColumns(columns, 32);
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

// The same two lines as text, cell after cell in the column order of makeGLFOutputData (OutputData::hasGLFColumns()): what
// dipMapLine(...).toString() / dipPositionLine(...).toString() give — an unset cell is "NA", integers are decimal digits, doubles
// "%.6g" — without a Line object (32 strings and ~25 label lookups per line; eight lines per window add up to a fifth of the
// reduce stage).  tests/test_glf_vcf_cpu.py holds both forms against each other.
// "%.6g" of a double — what `ostream << x` prints with the default format — without going through printf for the common case.
// The six significant digits are taken from |x| scaled by an exact power of ten (10^k is exact in fp64 for k <= 22, and a single
// multiplication or division by it is off by at most half an ulp); they are trusted only when the scaled value lies further from a
// rounding boundary (.5) than that error could ever reach, otherwise — and for anything outside 1e-5 <= |x| < 1e15, zero, inf, nan —
// snprintf decides.  tests/test_glf_vcf_cpu.py compares it with snprintf on random, boundary and tie values.
inline int formatG6(double x, char *out)
{
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const double a = x < 0 ? -x : x;
    if (!(a >= 1e-5 && a < 1e15)) return snprintf(out, 32, "%.6g", x);
    int e = 0;                                             // decimal exponent of the leading digit: 10^e <= a < 10^(e+1)
    if (a >= 1.0) { while (e < 22 && a >= p10[e + 1]) e++; }
    else { e = -1; while (a * p10[-e] < 1.0) e--; }        // a * 10^-e in [1, 10) (the product may round across: checked below)
    const int shift = 5 - e;                               // scale to six digits in front of the point
    const double y = shift >= 0 ? a * p10[shift] : a / p10[-shift];
    double r = double((long long)(y + 0.5));
    const double frac = y - (r - 0.5);                     // distance above the lower boundary r - 0.5, in [0, 1)
    if (!(y >= 99999.5 && y < 999999.5) || frac < 1e-4 || frac > 1.0 - 1e-4 || r >= 1e6 || r < 1e5) return snprintf(out, 32, "%.6g", x);
    long long digits = (long long)r;                       // 100000 .. 999999
    char d[6];
    for (int i = 5; i >= 0; i--) { d[i] = char('0' + digits % 10); digits /= 10; }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') nd--;               // %g strips trailing zeros
    char *o = out;
    if (x < 0) *o++ = '-';
    if (e < -4 || e >= 6) {                                // d.ddddde+XX
        *o++ = d[0];
        if (nd > 1) { *o++ = '.'; for (int i = 1; i < nd; i++) *o++ = d[i]; }
        *o++ = 'e'; *o++ = e < 0 ? '-' : '+';
        const int ae = e < 0 ? -e : e;
        *o++ = char('0' + ae / 10); *o++ = char('0' + ae % 10);
    } else if (e >= 0) {                                   // e + 1 digits, a point, the rest
        for (int i = 0; i <= e; i++) *o++ = i < nd ? d[i] : '0';
        if (nd > e + 1) { *o++ = '.'; for (int i = e + 1; i < nd; i++) *o++ = d[i]; }
    } else {                                               // 0.000ddd
        *o++ = '0'; *o++ = '.';
        for (int i = -1; i > e; i--) *o++ = '0';
        for (int i = 0; i < nd; i++) *o++ = d[i];
    }
    *o = 0;
    return int(o - out);
}

class GlfText {
public:
    explicit GlfText(std::string &buffer) : s(buffer), first(true) { s.clear(); }
    GlfText &na() { sep(); s += "NA"; return *this; }
    GlfText &cell(const std::string &x) { sep(); s += x; return *this; }
    GlfText &cell(const char *x) { sep(); s += x; return *this; }
    GlfText &cell(double x) { char b[64]; const int n = formatG6(x, b); sep(); s.append(b, size_t(n)); return *this; }
    GlfText &cell(long long x) { char b[32]; const int n = snprintf(b, sizeof(b), "%lld", x); sep(); s.append(b, size_t(n)); return *this; }
    GlfText &cell(int x) { return cell((long long)x); }
    GlfText &cell(unsigned x) { return cell((long long)x); }
    GlfText &cell(unsigned long x) { char b[32]; const int n = snprintf(b, sizeof(b), "%lu", x); sep(); s.append(b, size_t(n)); return *this; }
private:
    void sep() { if (!first) s += ' '; first = false; }
    std::string &s; bool first;
};

inline void dipMapText(std::string &out, const DipMapCall &c)
{
    char genoqual[64];
    formatG6(c.genoqual, genoqual);
    GlfText t(out);
    t.cell("ok").cell(c.index).cell("dip.map").cell(c.tid).cell(c.leftPos).cell(c.rightPos).cell(c.candPos).cell(c.realignedPos).cell(c.was_candidate);
    t.na().cell(c.nref_all).cell((unsigned long)c.num_reads).na().cell(c.qual).na().na().na().cell(0).cell(c.msq).na().na().cell(c.numf).cell(c.numr);
    t.cell(c.numUnmappedRealigned).cell(c.vc_f).cell(c.vc_r).na().na().na().na().na();
    t.cell(c.genotype + ":" + genoqual);
}

inline void dipPositionText(std::string &out, const DipPositionRow &c)
{
    GlfText t(out);
    t.cell("ok").cell(c.index).cell(c.program).cell(c.tid).cell(c.leftPos).cell(c.rightPos).cell(c.candPos).cell(c.realignedPos).cell(c.has_variants_in_window);
    t.na().cell(c.nref_all).cell((unsigned long)c.num_reads).na().na().na().cell(c.logZ).na().cell(0).cell(c.msq).cell(c.numOffAll).cell(c.num_indel).cell(c.nf).cell(c.nr);
    t.cell(c.numUnmappedRealigned).cell(c.var_coverage_forward).cell(c.var_coverage_reverse).cell(c.nBQT).cell(c.nmmBQT).cell(c.mLogBQ / double(c.nBQT));
    t.cell(c.nMMLeft).cell(c.nMMRight).cell(c.glf);
}

} // namespace dindel
#endif
