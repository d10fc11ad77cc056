// glf_to_vcf.cpp — see glf_to_vcf.hpp.  Line references are to the reference's python/mergeOutputDiploid.py unless a file is named.
#include "glf_to_vcf.hpp"
#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>

namespace dindel {

namespace {

std::vector<std::string> split(const std::string &s, char sep)            // str.split(sep)
{
    std::vector<std::string> out;
    size_t start = 0;
    for (;;) {
        const size_t p = s.find(sep, start);
        if (p == std::string::npos) { out.push_back(s.substr(start)); break; }
        out.push_back(s.substr(start, p - start));
        start = p + 1;
    }
    return out;
}

std::vector<std::string> splitWhitespace(const std::string &s)            // str.split()
{
    std::vector<std::string> out;
    std::istringstream is(s);
    std::string w;
    while (is >> w) out.push_back(w);
    return out;
}

std::string rstrip(std::string s, const char *chars)
{
    while (!s.empty() && strchr(chars, s[s.size() - 1])) s.erase(s.size() - 1);
    return s;
}

long toInt(const std::string &cell, const char *what)                     // int(str): an integer literal, surrounding blanks allowed
{
    const char *p = cell.c_str();
    char *end = NULL;
    errno = 0;
    const long v = strtol(p, &end, 10);
    while (end && (*end == ' ' || *end == '\t')) end++;
    if (end == p || *end != 0 || errno) throw std::string("invalid literal for int() in column ").append(what).append(": '").append(cell).append("'");
    return v;
}

std::string join(const std::vector<std::string> &v, const char *sep)
{
    std::string s;
    for (size_t i = 0; i < v.size(); i++) { if (i) s += sep; s += v[i]; }
    return s;
}

// FileUtils.FileWithHeader opened for reading: header labels and one dict per line
// FileUtils.FileWithHeader: a space-separated table with a header line; readline() gives the row as label -> cell (FileUtils.py:107-125:
// nothing at end of file or at an empty line).  The row is kept as the cells in column order with the labels' positions looked up once
// (the script builds a dict per line: 29 cells x 480,000 lines for a 60,000-window run).
class TableRow {
public:
    const std::string &operator[](const char *key) const
    {
        std::map<std::string, size_t>::const_iterator it = index->find(key);
        if (it == index->end()) throw std::string("KeyError: ").append(key);
        return cells[it->second];
    }
    std::vector<std::string> cells;
    const std::map<std::string, size_t> *index;
};

class TableReader {
public:
    explicit TableReader(const std::string &fname) : f(fname.c_str()), name(fname)
    {
        if (!f.is_open()) throw std::string("Cannot open file ").append(fname);
        std::string header;
        std::getline(f, header);
        labels = split(rstrip(header, " \t\r\n\v\f"), ' ');               // FileUtils.py:55,57
        for (size_t i = 0; i < labels.size(); i++) index[labels[i]] = i;    // a repeated label: the last column wins, as in the dict
    }
    bool readline(TableRow &res)
    {
        if (!std::getline(f, rline)) return false;
        size_t end = rline.size();
        while (end > 0 && rline[end - 1] == '\n') end--;                   // rstrip("\n").rstrip(" ")
        while (end > 0 && rline[end - 1] == ' ') end--;
        res.index = &index;
        size_t n = 0, start = 0;
        for (;;) {                                                         // str.split(' '): every single blank separates
            size_t sp = rline.find(' ', start);
            if (sp == std::string::npos || sp >= end) sp = end;
            if (n == res.cells.size()) res.cells.push_back(std::string());
            res.cells[n++].assign(rline, start, sp - start);
            if (sp >= end) break;
            start = sp + 1;
        }
        res.cells.resize(n);
        if (n == 1 && res.cells[0].empty()) return false;
        if (n != labels.size())
            throw std::string("Line in file ").append(name).append(" does not have the correct number of labels");
        return true;
    }
private:
    std::ifstream f;
    std::string name, rline;
    std::vector<std::string> labels;
    std::map<std::string, size_t> index;
};

const std::string &col(const TableRow &dat, const char *key) { return dat[key]; }

} // namespace

// ---------------- python/utils/Fasta.py ----------------
IndexedFasta::IndexedFasta(const std::string &fname) : fa(NULL)
{
    std::ifstream fi((fname + ".fai").c_str());
    if (!fi.is_open()) throw std::string("Cannot open file ").append(fname).append(".fai");
    std::string line;
    while (std::getline(fi, line)) {
        const std::vector<std::string> dat = splitWhitespace(line);
        if (dat.size() == 5) {                                            // Fasta.py:18-25
            Target t;
            t.len = toInt(dat[1], "fai"); t.offset = toInt(dat[2], "fai"); t.blen = toInt(dat[3], "fai"); t.llen = toInt(dat[4], "fai");
            if (ft.find(dat[0]) == ft.end()) order.push_back(dat[0]);
            ft[dat[0]] = t;
        }
    }
    fa = fopen(fname.c_str(), "rb");
    if (!fa) throw std::string("Cannot open file ").append(fname);
}

IndexedFasta::~IndexedFasta() { if (fa) fclose(fa); }

std::vector<std::string> IndexedFasta::names() const { return order; }

std::string IndexedFasta::get(const std::string &tid, long pos1based, int len)
{
    const long pos = pos1based - 1;                                       // Fasta.py:40
    std::map<std::string, Target>::const_iterator it = ft.find(tid);
    if (it == ft.end()) throw std::string("KeyError");                    // Fasta.py:43-45
    const Target &idx = it->second;
    if (idx.blen <= 0) throw std::string("fai: zero bases per line");
    // Python 2 `/` and `%` on ints floor; positions left of the sequence start are not expected here
    long q = pos / idx.blen, r = pos % idx.blen;
    if (r < 0) { r += idx.blen; q -= 1; }
    const long fpos = idx.offset + q * idx.llen + r;                      // Fasta.py:46
    std::string seq;
    if (fpos < 0 || fseek(fa, fpos, SEEK_SET) != 0) return seq;
    while (int(seq.size()) < len) {                                       // Fasta.py:49-53
        const int c = fgetc(fa);
        if (c == EOF) break;
        if (c != '\n') seq += char(c);
    }
    return seq;
}

// python/utils/AnalyzeSequence.py
int homopolymerLength(const std::string &seq, int pos)
{
    int hp_len = 1;
    for (int i = pos + 1; i < int(seq.size()); i++) {
        if (seq[size_t(i)] == seq[size_t(i - 1)]) hp_len++; else break;
    }
    for (int i = pos - 1; i > 0; i--) {                                   // range(pos-1, 0, -1): index 0 is never looked at
        if (i + 1 < int(seq.size()) && seq[size_t(i)] == seq[size_t(i + 1)]) hp_len++; else break;
    }
    return hp_len;
}

// python/utils/Variant.py:3-27
VariantString::VariantString(const std::string &s) : type(REF), length(0)
{
    const size_t l = s.size();
    if (l == 0) throw std::string("Unrecognized variant: ");
    if (s[0] == '-' && l > 1) { type = DEL; seq = s.substr(1); length = int(l) - 1; }
    else if (s[0] == '+' && l > 1) { type = INS; seq = s.substr(1); length = int(l) - 1; }
    else if (l == 4 && s[1] == '=' && s[2] == '>') { type = SNP; seq = s.substr(3, 1); length = 1; }
    else if (s[0] == '*' || s.find("REF") != std::string::npos || s.find("ref") != std::string::npos) { type = REF; length = 0; }
    else throw std::string("Unrecognized variant: ").append(s);
}

long intOfFloat(const std::string &cell)
{
    const char *p = cell.c_str();
    char *end = NULL;
    const double v = strtod(p, &end);
    while (end && (*end == ' ' || *end == '\t')) end++;
    if (end == p || *end != 0) throw std::string("could not convert string to float: ").append(cell);
    if (v != v) throw std::string("cannot convert float NaN to integer");
    return long(v);                                                       // int(float): truncation towards zero
}

std::pair<std::string, long> getVCFString(const GlfCall &glf, IndexedFasta &fa, int maxHPLen, int filterQual,
                                          const std::vector<std::string> &addFilters)
{
    std::vector<std::string> filters;
    const long pos = glf.pos;
    const std::string seq = fa.get(glf.chr, pos + 1 - 25, 50);            // :46
    const int hplen = homopolymerLength(seq, 25);                         // :47
    long report_pos = pos;
    int max_del_len = 0;
    {
        const std::set<std::string> scanAlleles(glf.nref_all.begin(), glf.nref_all.end());     // :54
        for (std::set<std::string>::const_iterator it = scanAlleles.begin(); it != scanAlleles.end(); ++it) {
            const VariantString var(*it);
            if (var.type == VariantString::DEL && var.length > max_del_len) max_del_len = var.length;
        }
    }
    const int seqlen = 1 + max_del_len;                                   // :61
    std::string refseq = fa.get(glf.chr, report_pos, seqlen);
    if (refseq.empty()) throw std::string("string index out of range");   // refseq[0] below
    std::vector<std::string> altseqs;
    std::map<std::string, VariantString::Type> altseq_to_type;
    for (size_t g = 0; g < glf.nref_all.size(); g++) {                    // :72-93
        const VariantString vnref(glf.nref_all[g]);
        std::string g_altseq;
        bool isRef = false;
        const size_t rl = refseq.size();
        if (vnref.type == VariantString::DEL) g_altseq = refseq.substr(0, 1) + (size_t(1 + vnref.length) < rl ? refseq.substr(size_t(1 + vnref.length)) : std::string());
        else if (vnref.type == VariantString::INS) g_altseq = refseq.substr(0, 1) + vnref.seq + refseq.substr(1);
        else if (vnref.type == VariantString::SNP) g_altseq = refseq.substr(0, 1) + vnref.seq.substr(0, 1) + (rl > 2 ? refseq.substr(2) : std::string());
        else { g_altseq = refseq; isRef = true; }
        if (!isRef && std::find(altseqs.begin(), altseqs.end(), g_altseq) == altseqs.end()) {
            altseqs.push_back(g_altseq);
            altseq_to_type[g_altseq] = vnref.type;
        }
    }
    const std::vector<std::string> gtd = split(glf.genotype, ':');        // :95-96
    if (gtd.size() < 2) throw std::string("list index out of range");
    std::ostringstream rec_gt;
    rec_gt << gtd[0] << ":" << intOfFloat(gtd[1]);

    bool onlySNPs = true;                                                 // :101-104
    for (size_t i = 0; i < altseqs.size(); i++) if (altseq_to_type[altseqs[i]] != VariantString::SNP) onlySNPs = false;
    if (onlySNPs) {                                                       // :108-116: only SNPs -> the row moves one base to the right
        report_pos += 1;
        refseq = fa.get(glf.chr, report_pos, 1);
        for (size_t i = 0; i < altseqs.size(); i++) altseqs[i] = altseqs[i].substr(altseqs[i].empty() ? 0 : 1);
    }
    if (hplen > maxHPLen) { std::ostringstream o; o << "hp" << maxHPLen; filters.push_back(o.str()); }      // :119-120
    if (glf.qual < filterQual) { std::ostringstream o; o << "q" << filterQual; filters.push_back(o.str()); } // :122-123
    for (size_t i = 0; i < altseqs.size(); i++) if (altseqs[i].find('D') != std::string::npos) altseqs[i] = "<DEL>";   // :127-135
    filters.insert(filters.end(), addFilters.begin(), addFilters.end());
    const std::string filterStr = filters.empty() ? std::string("PASS") : join(filters, ";");
    std::ostringstream info;                                              // :145: NRS / NFS get forward_old / reverse_old in this order
    info << "DP=" << toInt(glf.num_hap_reads, "num_reads") << ";NF=" << glf.num_cover_forward << ";NR=" << glf.num_cover_reverse
         << ";NRS=" << glf.num_cover_forward_old << ";NFS=" << glf.num_cover_reverse_old << ";HP=" << hplen;
    std::ostringstream r;                                                 // :152
    r << glf.chr << "\t" << report_pos << "\t.\t" << refseq << "\t" << join(altseqs, ",") << "\t" << glf.qual << "\t" << filterStr << "\t"
      << info.str() << "\tGT:GQ\t" << rec_gt.str();
    return std::make_pair(r.str(), report_pos);
}

int processDiploidGLFFile(const std::string &glfFile, CallsByChrom &variants, IndexedFasta &fa, int maxHPLen, int filterQual)
{
    (void)maxHPLen;          // :219 calls getVCFString without maxHPLen: the hp filter always uses the default 10, whatever --maxHPLen says
    TableReader fglf(glfFile);
    int numSkipped = 0;
    TableRow dat;
    while (fglf.readline(dat)) {
        if (col(dat, "msg") != "ok") { numSkipped++; continue; }           // :182-184
        if (col(dat, "analysis_type") != "dip.map") continue;             // :186
        if (col(dat, "was_candidate_in_window") != "1") continue;         // :189
        GlfCall glf;
        glf.chr = col(dat, "tid");
        glf.pos = toInt(col(dat, "realigned_position"), "realigned_position");
        glf.qual = intOfFloat(col(dat, "qual"));                          // :204
        if (double(glf.qual) < 1.0) continue;                             // :206
        glf.nref_all = split(col(dat, "nref_all"), ',');
        if (glf.nref_all.size() == 1 && glf.nref_all[0] == "R=>D") continue;   // :210
        glf.num_cover_forward = toInt(split(col(dat, "var_coverage_forward"), ',')[0], "var_coverage_forward");
        glf.num_cover_reverse = toInt(split(col(dat, "var_coverage_reverse"), ',')[0], "var_coverage_reverse");
        glf.num_cover_forward_old = toInt(col(dat, "num_cover_forward"), "num_cover_forward");
        glf.num_cover_reverse_old = toInt(col(dat, "num_cover_reverse"), "num_cover_reverse");
        glf.num_hap_reads = col(dat, "num_reads");
        glf.genotype = col(dat, "glf");
        const std::pair<std::string, long> v = getVCFString(glf, fa, 10, filterQual);
        variants[glf.chr][v.second].push_back(v.first);
    }
    return numSkipped;
}

void mergeOutput(const std::string &glfFilesFile, const std::string &sampleID, const std::string &refFile, int maxHPLen,
                 const std::string &vcfFile, int filterQual)
{
    std::ifstream fg(glfFilesFile.c_str());
    if (!fg.is_open()) throw std::string("Cannot open file ").append(glfFilesFile);
    std::vector<std::string> okFiles;
    std::string line;
    int lineidx = 0;
    while (std::getline(fg, line)) {
        lineidx++;
        const std::vector<std::string> dat = splitWhitespace(rstrip(line, "\n"));
        if (dat.empty()) throw std::string("list index out of range");    // dat[0] of an empty line (:256)
        if (dat.size() > 1) std::cerr << "WARNING: additional columns in line " << lineidx << " of file " << glfFilesFile << " were ignored\n";
        if (!std::ifstream(dat[0].c_str()).is_open()) {
            std::cerr << "File " << dat[0] << " does not exist\n. Aborting.\n";
            throw std::string("File does not exist: ").append(dat[0]);
        }
        okFiles.push_back(dat[0]);
    }
    std::cout << "Number of non-empty GLF files: " << okFiles.size() << "\n";
    std::ofstream fv(vcfFile.c_str());
    if (!fv.is_open()) throw std::string("Cannot open file ").append(vcfFile).append(" for writing.");
    fv << "##fileformat=VCFv4.0\n";                                       // :271-286
    fv << "##source=Dindel\n";
    fv << "##reference=" << refFile << "\n";
    fv << "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total number of reads in haplotype window\">\n";
    fv << "##INFO=<ID=HP,Number=1,Type=Integer,Description=\"Reference homopolymer tract length\">\n";
    fv << "##INFO=<ID=NF,Number=1,Type=Integer,Description=\"Number of reads covering non-ref variant on forward strand\">\n";
    fv << "##INFO=<ID=NR,Number=1,Type=Integer,Description=\"Number of reads covering non-ref variant on reverse strand\">\n";
    fv << "##INFO=<ID=NFS,Number=1,Type=Integer,Description=\"Number of reads covering non-ref variant site on forward strand\">\n";
    fv << "##INFO=<ID=NRS,Number=1,Type=Integer,Description=\"Number of reads covering non-ref variant site on reverse strand\">\n";
    fv << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
    fv << "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n";
    fv << "##ALT=<ID=DEL,Description=\"Deletion\">\n";
    fv << "##FILTER=<ID=q" << filterQual << ",Description=\"Quality below " << filterQual << "\">\n";
    fv << "##FILTER=<ID=hp" << maxHPLen << ",Description=\"Reference homopolymer length was longer than " << maxHPLen << "\">\n";
    fv << "##FILTER=<ID=fr0,Description=\"Non-ref allele is not covered by at least one read on both strands\">\n";
    fv << "##FILTER=<ID=wv,Description=\"Other indel in window had higher likelihood\">\n";
    fv << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" << sampleID << "\n";

    IndexedFasta fa(refFile);
    CallsByChrom variants;
    for (size_t i = 0; i < okFiles.size(); i++) {
        std::cout << "Calling variants from GLF file " << okFiles[i] << "\n";
        processDiploidGLFFile(okFiles[i], variants, fa, maxHPLen, filterQual);
    }
    std::vector<std::string> this_chr;                                    // :292-300
    for (int v = 1; v < 23; v++) { std::ostringstream o; o << v; this_chr.push_back(o.str()); }
    this_chr.push_back("X");
    this_chr.push_back("Y");
    for (CallsByChrom::const_iterator it = variants.begin(); it != variants.end(); ++it)
        if (std::find(this_chr.begin(), this_chr.end(), it->first) == this_chr.end()) this_chr.push_back(it->first);
    for (size_t c = 0; c < this_chr.size(); c++) {                        // :302-306
        CallsByChrom::const_iterator it = variants.find(this_chr[c]);
        if (it == variants.end()) continue;
        for (std::map<long, std::vector<std::string> >::const_iterator pt = it->second.begin(); pt != it->second.end(); ++pt)
            for (size_t k = 0; k < pt->second.size(); k++) fv << pt->second[k] << "\n";
    }
}

} // namespace dindel
