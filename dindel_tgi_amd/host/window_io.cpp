// window_io.cpp — see window_io.hpp.
#include "window_io.hpp"
#include <cstdlib>
#include <cmath>
#include <iostream>
#include <sstream>

namespace dindel {

namespace {
template <class T> bool from_string(T &t, const std::string &s)      // reference Utils.hpp:40-48
{
    std::istringstream iss(s);
    return !(iss >> std::dec >> t).fail();
}
}

VariantFile::VariantFile(const std::string &fileName) : index(0)
{
    fin.open(fileName.c_str());
    if (!fin.is_open()) throw std::string("Cannot open variant file ").append(fileName);
}

AlignedCandidates VariantFile::getLineVector(bool isOneBased)
{
    const AlignedCandidates aligned_empty;
    uint32_t pos;
    int leftPos, rightPos;
    std::string tid, line;
    std::getline(fin, line);
    if (line.empty()) return aligned_empty;
    std::istringstream is(line);
    index++;
    if (!is.eof()) is >> tid; else return aligned_empty;
    if (!is.eof()) {
        std::string str;
        is >> str;
        if (!from_string<int>(leftPos, str)) throw std::string("Cannot read left boundary of region.");
    } else return aligned_empty;
    if (!is.eof()) {
        std::string str;
        is >> str;
        if (!from_string<int>(rightPos, str)) throw std::string("Cannot read left boundary of region.");    // (sic) :212
    } else return aligned_empty;
    std::vector<AlignedVariant> variants;
    try {
        while (!is.eof()) {
            std::string pvf_str;
            if (!is.eof()) is >> pvf_str;
            if (pvf_str.empty()) break;
            if (pvf_str[0] == '#' || pvf_str[0] == '%') break;
            std::vector<std::string> els;                          // split at ';' or ',' (:233-241)
            int lastpos = 0;
            for (int x = 0; x < int(pvf_str.size()); x++) {
                if ((pvf_str[size_t(x)] == ';' || pvf_str[size_t(x)] == ',') && x - lastpos > 0) {
                    els.push_back(pvf_str.substr(size_t(lastpos), size_t(x - lastpos)));
                    lastpos = x + 1;
                }
            }
            els.push_back(pvf_str.substr(size_t(lastpos), pvf_str.size() - size_t(lastpos)));
            if (els.size() < 2) {
                std::cerr << "Error reading line in variantfile!\n";
            } else {
                double freq = -1.0;
                bool addComb = false;
                if (!from_string<uint32_t>(pos, els[0])) throw std::string("Cannot read position");
                if (isOneBased) pos--;
                const std::string &col = els[1];
                if (col.size() == 0 || (col[0] != '-' && col[0] != '+' && col[0] != 'A' && col[0] != 'C' && col[0] != 'G' && col[0] != 'T' && col[0] != 'R'))
                    throw std::string("Unrecognized variant");
                if (els.size() > 2 && !from_string<double>(freq, els[2])) throw std::string("Cannot read prior/frequency");
                if (els.size() > 3) {
                    int addc;
                    if (!from_string<int>(addc, els[3])) throw std::string("Cannot add_combinatorial");
                    if (addc) addComb = true;
                }
                AlignedVariant variant(col, int(pos), freq, addComb);
                if (variant.getSeq().size() != 0) variants.push_back(variant);
            }
        }
    } catch (std::string &err) {
        std::cerr << "Could not parse variants in line " << index << " in variants file." << std::endl;
        std::cerr << "Error: " << err << std::endl;
        return aligned_empty;
    }
    if (variants.size() == 0) {
        std::cerr << "Could not parse any variants in line: " << index << " SKIPPING." << std::endl;
        return aligned_empty;
    }
    return AlignedCandidates(tid, variants, leftPos, rightPos);
}

LibraryCollection::LibraryCollection()
{
    (*this)["single_end"] = Library(std::vector<double>(2000, 1.0));         // Library(0), Library.hpp:44-50
}

void LibraryCollection::addFromFile(const std::string &fileName)
{
    std::ifstream fin(fileName.c_str());
    if (!fin.is_open()) throw std::string("Cannot open variant file ").append(fileName);       // (sic) Library.hpp:147
    int numLibs = 0, numLines = 0, prev = -1;
    std::vector<double> counts;
    std::string libName;
    while (!fin.eof()) {
        std::string line;
        std::getline(fin, line);
        numLines++;
        if (line.empty()) break;
        std::istringstream is(line);
        std::string isize_str, count_str;
        int isize = -1;
        double count = -1;
        is >> isize_str;
        if (isize_str == "#LIB") {
            if (counts.size() > 0 && !libName.empty()) {
                if (find(libName) != end()) throw std::string("Library error");                   // duplicate library IDs
                (*this)[libName] = Library(counts);
                numLibs++;
                counts.clear();
                prev = -1;
            }
            std::string label;
            is >> label;
            libName = label;
            if (label.empty()) throw std::string("Cannot read library name ");
            continue;
        }
        if (!from_string<int>(isize, isize_str)) std::cerr << "Error reading from library file" << std::endl;
        is >> count_str;
        if (!from_string<double>(count, count_str)) std::cerr << "Error reading from library file" << std::endl;
        if (isize != prev + 1) throw std::string("Library error.");                              // insert sizes must be consecutive
        if (count < 0) throw std::string("Library error.");
        counts.push_back(count);
        prev = isize;
    }
    if (find(libName) != end()) throw std::string("Library error");
    (*this)[libName] = Library(counts);
    numLibs++;
    if (numLibs == 0) std::cerr << "Could not find any libraries. Are the headers specified correctly?" << std::endl;
}

double LibraryCollection::getMaxInsertSize() const
{
    double max = -HUGE_VAL;
    for (const_iterator it = begin(); it != end(); ++it)
        if (it->second.getMaxInsertSize() > max) max = it->second.getMaxInsertSize();
    return max;
}

HaplotypeFixture::HaplotypeFixture(const std::string &fileName)
{
    std::ifstream fin(fileName.c_str());
    if (!fin.is_open()) throw std::string("Cannot open haplotype file ").append(fileName);
    std::string line;
    WindowHaplotypes *cur = NULL;
    int lineNo = 0;
    std::vector<std::pair<const char *, size_t> > tok;                 // the line's blank-separated fields
    while (std::getline(fin, line)) {
        lineNo++;
        if (line.empty() || line[0] == '#') continue;
        tok.clear();
        for (size_t i = 0; i < line.size();) {
            while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) i++;
            const size_t st = i;
            while (i < line.size() && !(line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) i++;
            if (i > st) tok.push_back(std::make_pair(line.data() + st, i - st));
        }
        if (tok.empty()) continue;
        auto bad = [&](const char *what) {
            std::ostringstream where;
            where << what << " in line " << lineNo << " of " << fileName;
            return where.str();
        };
        auto integer = [&](size_t k, const char *what) -> long {      // field k as a whole decimal number
            if (k >= tok.size()) throw bad(what);
            const std::string t(tok[k].first, tok[k].second);
            char *end = NULL;
            const long v = strtol(t.c_str(), &end, 10);
            if (end == t.c_str() || *end) throw bad(what);
            return v;
        };
        const std::string tag(tok[0].first, tok[0].second);
        if (tag == "W") {
            WindowHaplotypes w;
            w.index = int(integer(1, "Cannot read window record"));
            w.leftPos = uint32_t(integer(2, "Cannot read window record")); w.rightPos = uint32_t(integer(3, "Cannot read window record"));
            cur = &(windows[w.index] = w);
        } else if (tag == "H") {
            if (!cur || tok.size() < 2) throw bad("Cannot read haplotype record");
            cur->haps.push_back(Haplotype(std::string(tok[1].first, tok[1].second)));
        } else if (tag == "V") {
            const char *what = "Cannot read variant record";
            if (!cur || cur->haps.empty() || tok.size() < 12) throw bad(what);
            const std::string kind(tok[1].first, tok[1].second), str(tok[3].first, tok[3].second);
            const int key = int(integer(2, what));
            int v[8];
            for (size_t k = 0; k < 8; k++) v[k] = int(integer(4 + k, what));
            AlignedVariant av(str, v[0], v[1], v[2], v[3]);
            av.setFlanking(v[4], v[5], v[6], v[7]);
            if (kind == "I") cur->haps.back().indels[key] = av;
            else if (kind == "S") cur->haps.back().snps[key] = av;
            else throw bad("Variant record must say I or S");
        } else throw bad("Unknown record");
    }
}

const WindowHaplotypes *HaplotypeFixture::find(int index) const
{
    std::map<int, WindowHaplotypes>::const_iterator it = windows.find(index);
    return it == windows.end() ? NULL : &it->second;
}

} // namespace dindel
