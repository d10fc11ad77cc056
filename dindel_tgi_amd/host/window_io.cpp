// window_io.cpp — see window_io.hpp.
#include "window_io.hpp"
#include <algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
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

namespace {
// the records of one stretch of the fixture file (whole windows: the stretch starts at a W line or at the head of the file)
void parseFixtureStretch(const char *p, const char *end, int lineNo, const std::string &fileName, std::vector<WindowHaplotypes> &out)
{
    WindowHaplotypes *cur = NULL;
    std::vector<std::pair<const char *, size_t> > tok;                 // the line's blank-separated fields
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', size_t(end - p)));
        const char *le = nl ? nl : end;
        lineNo++;
        const char *q = p;
        p = nl ? nl + 1 : end;
        if (q == le || *q == '#') continue;
        tok.clear();
        while (q < le) {
            while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
            const char *st = q;
            while (q < le && !(*q == ' ' || *q == '\t' || *q == '\r')) q++;
            if (q > st) tok.push_back(std::make_pair(st, size_t(q - st)));
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
            char *stop = NULL;
            const long v = strtol(t.c_str(), &stop, 10);
            if (stop == t.c_str() || *stop) throw bad(what);
            return v;
        };
        const std::string tag(tok[0].first, tok[0].second);
        if (tag == "W") {
            WindowHaplotypes w;
            w.index = int(integer(1, "Cannot read window record"));
            w.leftPos = uint32_t(integer(2, "Cannot read window record")); w.rightPos = uint32_t(integer(3, "Cannot read window record"));
            out.push_back(w);
            cur = &out.back();
        } else if (tag == "H") {
            if (!cur || tok.size() < 2) throw bad("Cannot read haplotype record");
            cur->haps.push_back(Haplotype(std::string(tok[1].first, tok[1].second)));
        } else if (tag == "A") {                                       // the haplotype's own alignment to the reference: one number per base
            if (!cur || cur->haps.empty()) throw bad("Cannot read haplotype alignment record");
            std::vector<int> &v = cur->haps.back().refHpos;
            v.clear();
            for (size_t k = 1; k < tok.size(); k++) v.push_back(int(integer(k, "Cannot read haplotype alignment record")));
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
}

HaplotypeFixture::HaplotypeFixture(const std::string &fileName) : fileName_(fileName), text_(NULL), size_(0), mapped_(false)
{
    const int fd = open(fileName.c_str(), O_RDONLY);
    if (fd < 0) throw std::string("Cannot open haplotype file ").append(fileName);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::string("Cannot open haplotype file ").append(fileName); }
    size_ = size_t(st.st_size);
    if (size_ > 0) {
        void *m = mmap(NULL, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) { text_ = static_cast<const char *>(m); mapped_ = true; }
        else {                                                     // a file that cannot be mapped (a pipe): read it
            char *buf = static_cast<char *>(malloc(size_));
            size_t got = 0;
            while (buf && got < size_) { const ssize_t k = read(fd, buf + got, size_ - got); if (k <= 0) break; got += size_t(k); }
            if (!buf || got != size_) { free(buf); close(fd); throw std::string("Cannot read haplotype file ").append(fileName); }
            text_ = buf;
        }
    }
    close(fd);
    // where the W records start, and how many lines lie in front of each: ranges of the file scanned side by side
    struct Mark { size_t at; int linesBefore; };
    unsigned hw = std::thread::hardware_concurrency();
    const size_t workers = size_ < (1u << 20) ? 1 : std::min<size_t>(8, hw ? hw : 1);
    std::vector<std::vector<Mark> > found(workers);
    std::vector<int> linesIn(workers, 0);
    auto scan = [&](size_t k) {
        const size_t lo = size_ * k / workers, hi = size_ * (k + 1) / workers;
        int lines = 0;
        size_t p = lo;
        if (p > 0 && text_[p - 1] != '\n') {                       // the line that straddles `lo` belongs to the range before
            const void *nl = memchr(text_ + p, '\n', hi - p);
            if (!nl) { linesIn[k] = 0; return; }
            p = size_t(static_cast<const char *>(nl) - text_) + 1;
            lines = 1;                                             // that newline is counted here: it lies in [lo, hi)
        }
        while (p < hi) {
            if (text_[p] == 'W' && p + 1 < size_ && (text_[p + 1] == ' ' || text_[p + 1] == '\t')) { Mark m = { p, lines }; found[k].push_back(m); }
            const void *nl = memchr(text_ + p, '\n', hi - p);
            if (!nl) break;
            p = size_t(static_cast<const char *>(nl) - text_) + 1;
            lines++;
        }
        linesIn[k] = lines;
    };
    {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < workers; k++) pool.push_back(std::thread(scan, k));
        scan(0);
        for (size_t k = 0; k < pool.size(); k++) pool[k].join();
    }
    std::vector<Mark> marks;
    int base = 0;
    for (size_t k = 0; k < workers; k++) {
        for (size_t i = 0; i < found[k].size(); i++) { Mark m = { found[k][i].at, base + found[k][i].linesBefore }; marks.push_back(m); }
        base += linesIn[k];
    }
    // records in front of the first window (comments; anything else is an error, reported now)
    {
        std::vector<WindowHaplotypes> none;
        parseFixtureStretch(text_, text_ + (marks.empty() ? size_ : marks[0].at), 0, fileName_, none);
    }
    std::vector<WindowHaplotypes> head;
    for (size_t i = 0; i < marks.size(); i++) {
        const char *q = text_ + marks[i].at;
        const char *lineEnd = static_cast<const char *>(memchr(q, '\n', size_ - marks[i].at));
        if (!lineEnd) lineEnd = text_ + size_;
        head.clear();
        parseFixtureStretch(q, lineEnd, marks[i].linesBefore, fileName_, head);               // the W line itself: a malformed one is reported now
        Entry e;
        e.begin = marks[i].at; e.end = i + 1 < marks.size() ? marks[i + 1].at : size_; e.lineBase = marks[i].linesBefore; e.state = 0;
        std::map<int, Entry>::iterator it = windows.find(head[0].index);
        if (it == windows.end()) windows.insert(std::make_pair(head[0].index, e));
        else { it->second.begin = e.begin; it->second.end = e.end; it->second.lineBase = e.lineBase; }     // the later record wins
    }
}

HaplotypeFixture::~HaplotypeFixture()
{
    if (text_) { if (mapped_) munmap(const_cast<char *>(text_), size_); else free(const_cast<char *>(text_)); }
}

const WindowHaplotypes *HaplotypeFixture::find(int index) const
{
    std::map<int, Entry>::const_iterator it = windows.find(index);
    if (it == windows.end()) return NULL;
    const Entry &e = it->second;
    std::lock_guard<std::mutex> lk(locks_[size_t(unsigned(index)) % 64]);
    if (e.state == 0) {
        std::vector<WindowHaplotypes> one;
        try {
            parseFixtureStretch(text_ + e.begin, text_ + e.end, e.lineBase, fileName_, one);
            if (one.size() != 1) throw std::string("Cannot read window record of ").append(fileName_);
            e.win.index = one[0].index; e.win.leftPos = one[0].leftPos; e.win.rightPos = one[0].rightPos;
            e.win.haps.swap(one[0].haps);
            e.state = 1;
        } catch (std::string &msg) { e.error = msg; e.state = 2; }
    }
    if (e.state == 2) { Error err; err.message = e.error; throw err; }
    return &e.win;
}

void HaplotypeFixture::release(int index) const
{
    std::map<int, Entry>::const_iterator it = windows.find(index);
    if (it == windows.end()) return;
    const Entry &e = it->second;
    std::lock_guard<std::mutex> lk(locks_[size_t(unsigned(index)) % 64]);
    if (e.state == 1) { std::vector<Haplotype>().swap(e.win.haps); e.state = 0; }
}

} // namespace dindel
