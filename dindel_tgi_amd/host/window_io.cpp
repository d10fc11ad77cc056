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

// What `std::istringstream(s) >> std::dec >> t` accepts is the reference's definition of a number (Utils.hpp:40-48: a prefix that parses
// is enough, "12abc" is 12), so the conversion itself stays with the stream classes; everything around it is this file's own scanner.
template <class T> bool numberFrom(const std::string &s, T &t)
{
    std::istringstream in(s);
    in >> std::dec >> t;
    return !in.fail();
}

// Blank-separated words of one text line, with the one property of formatted stream input the reference's callers depend on:
// `drained` turns true when a word (or the search for one) ran into the end of the line — a line that ends in a blank is NOT drained after
// its last word, and asking again yields an empty word.
class Words {
public:
    explicit Words(const std::string &line) : p_(line.data()), e_(line.data() + line.size()), drained_(line.empty()) {}
    bool drained() const { return drained_; }
    std::string next()
    {
        while (p_ < e_ && isspace((unsigned char)*p_)) ++p_;
        const char *b = p_;
        while (p_ < e_ && !isspace((unsigned char)*p_)) ++p_;
        if (p_ == e_) drained_ = true;
        return std::string(b, p_);
    }
private:
    const char *p_, *e_;
    bool drained_;
};

// "pos,variant[,prior[,addCombinatorially]]" — fields end at ',' or ';', except that a separator met while the field is still empty does
// not end anything and stays in front of the field ("7,,+A" is {"7", ",+A"}).
void candidateFields(const std::string &word, std::vector<std::string> &fields)
{
    fields.clear();
    std::string cur;
    for (size_t i = 0; i < word.size(); i++) {
        const char c = word[i];
        if ((c == ',' || c == ';') && !cur.empty()) { fields.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    fields.push_back(cur);
}

bool variantLeadChar(char c) { return c && strchr("-+ACGTR", c) != NULL; }

// One candidate of a window line -> `out` (false: its variant string parsed to nothing and is dropped).  Throws the reference's messages.
bool candidateFrom(const std::vector<std::string> &f, bool isOneBased, AlignedVariant &out)
{
    uint32_t pos = 0;
    if (!numberFrom(f[0], pos)) throw std::string("Cannot read position");
    if (isOneBased) pos--;                                       // to zero-based
    if (!variantLeadChar(f[1].empty() ? '\0' : f[1][0])) throw std::string("Unrecognized variant");
    double prior = -1.0;
    if (f.size() > 2 && !numberFrom(f[2], prior)) throw std::string("Cannot read prior/frequency");
    int combine = 0;
    if (f.size() > 3 && !numberFrom(f[3], combine)) throw std::string("Cannot add_combinatorial");
    out = AlignedVariant(f[1], int(pos), prior, combine != 0);
    return out.getSeq().size() != 0;
}

} // namespace

VariantFile::VariantFile(const std::string &fileName) : index(0)
{
    fin.open(fileName.c_str());
    if (!fin.is_open()) throw std::string("Cannot open variant file ").append(fileName);
}

// Window file (format of reference VariantFile.hpp:188-289): `tid leftPos rightPos candidate ...`, a word starting with '#' or '%' ends
// the line.  Outcomes, as there: an AlignedCandidates without variants for an empty line, a line that ends before its third word, a
// candidate that does not parse (reported on stderr with the line number) or no candidate at all; a std::string thrown for a boundary
// that is not a number.
AlignedCandidates VariantFile::getLineVector(bool isOneBased)
{
    std::string line;
    std::getline(fin, line);
    if (line.empty()) return AlignedCandidates();
    index++;
    Words words(line);
    const std::string tid = words.next();
    if (words.drained()) return AlignedCandidates();
    int bounds[2];
    for (int k = 0; k < 2; k++) {
        if (!numberFrom(words.next(), bounds[k])) throw std::string("Cannot read left boundary of region.");   // the same text for both
        if (k == 0 && words.drained()) return AlignedCandidates();
    }
    std::vector<AlignedVariant> found;
    std::vector<std::string> fields;
    std::string problem;
    while (!words.drained() && problem.empty()) {
        const std::string word = words.next();
        if (word.empty() || word[0] == '#' || word[0] == '%') break;
        candidateFields(word, fields);
        if (fields.size() < 2) { std::cerr << "Error reading line in variantfile!\n"; continue; }
        try {
            AlignedVariant v;
            if (candidateFrom(fields, isOneBased, v)) found.push_back(v);
        } catch (const std::string &message) { problem = message; }
    }
    if (!problem.empty()) {
        std::cerr << "Could not parse variants in line " << index << " in variants file." << std::endl << "Error: " << problem << std::endl;
        return AlignedCandidates();
    }
    if (found.empty()) {
        std::cerr << "Could not parse any variants in line: " << index << " SKIPPING." << std::endl;
        return AlignedCandidates();
    }
    return AlignedCandidates(tid, found, bounds[0], bounds[1]);
}

LibraryCollection::LibraryCollection()
{
    (*this)["single_end"] = Library(std::vector<double>(2000, 1.0));         // Library(0), Library.hpp:44-50
}

// Library file (format of reference Library.hpp:143-241): sections `#LIB <name>` followed by `<insertSize> <count>` rows, insert sizes
// counting up from 0 without gaps; the first empty line ends the file.  A section is kept when the next header (or the end) is reached;
// a header that follows no rows only renames the section, and rows in front of the first header run on into the first named section —
// both as the reference's reader behaves.  Throws "Library error" (a name used twice), "Library error." (gap in the insert sizes,
// negative count), "Cannot read library name ".
void LibraryCollection::addFromFile(const std::string &fileName)
{
    std::ifstream file(fileName.c_str(), std::ios::in | std::ios::binary);
    if (!file.is_open()) throw std::string("Cannot open variant file ").append(fileName);       // (sic) the reference's text
    std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());

    std::string name;
    std::vector<double> counts;
    auto keep = [&]() {
        if (count(name)) { std::cerr << "Duplicate library IDs: " << name << std::endl; throw std::string("Library error"); }
        (*this)[name] = Library(counts);
    };
    for (size_t at = 0; at < text.size();) {
        size_t nl = text.find('\n', at);
        if (nl == std::string::npos) nl = text.size();
        const std::string line = text.substr(at, nl - at);
        at = nl + 1;
        if (line.empty()) break;
        Words words(line);
        const std::string first = words.next();
        if (first == "#LIB") {
            if (!counts.empty() && !name.empty()) { keep(); counts.clear(); }
            name = words.next();
            if (name.empty()) throw std::string("Cannot read library name ");
            continue;
        }
        int insertSize = -1;
        double n = -1;
        if (!numberFrom(first, insertSize) | !numberFrom(words.next(), n)) std::cerr << "Error reading from library file" << std::endl;
        if (insertSize != int(counts.size())) { std::cerr << "Library insert sizes must be consecutive" << std::endl; throw std::string("Library error."); }
        if (n < 0) { std::cerr << "Library insert size count is negative.." << std::endl; throw std::string("Library error."); }
        counts.push_back(n);
    }
    keep();                                                                   // the last section, whatever it holds
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
        auto integer = [&](size_t k, const char *what) -> long {      // field k as a whole decimal number ([+-]digits, nothing else)
            if (k >= tok.size()) throw bad(what);
            const char *c = tok[k].first, *e = c + tok[k].second;
            const bool neg = *c == '-';
            if (*c == '-' || *c == '+') c++;
            if (c == e || e - c > 18) throw bad(what);
            long v = 0;
            for (; c < e; c++) {
                const unsigned d = unsigned(*c) - unsigned('0');
                if (d > 9) throw bad(what);
                v = v * 10 + long(d);
            }
            return neg ? -v : v;
        };
        const char tag = tok[0].second == 1 ? tok[0].first[0] : '?';
        if (tag == 'W') {
            WindowHaplotypes w;
            w.index = int(integer(1, "Cannot read window record"));
            w.leftPos = uint32_t(integer(2, "Cannot read window record")); w.rightPos = uint32_t(integer(3, "Cannot read window record"));
            out.push_back(w);
            cur = &out.back();
        } else if (tag == 'H') {
            if (!cur || tok.size() < 2) throw bad("Cannot read haplotype record");
            cur->haps.push_back(Haplotype(std::string(tok[1].first, tok[1].second)));
        } else if (tag == 'A') {                                       // the haplotype's own alignment to the reference: one number per base
            if (!cur || cur->haps.empty()) throw bad("Cannot read haplotype alignment record");
            std::vector<int> &v = cur->haps.back().refHpos;
            v.resize(tok.size() - 1);
            for (size_t k = 1; k < tok.size(); k++) v[k - 1] = int(integer(k, "Cannot read haplotype alignment record"));
        } else if (tag == 'V') {
            const char *what = "Cannot read variant record";
            if (!cur || cur->haps.empty() || tok.size() < 12) throw bad(what);
            const char kind = tok[1].second == 1 ? tok[1].first[0] : '?';
            const std::string str(tok[3].first, tok[3].second);
            const int key = int(integer(2, what));
            int v[8];
            for (size_t k = 0; k < 8; k++) v[k] = int(integer(4 + k, what));
            AlignedVariant av(str, v[0], v[1], v[2], v[3]);
            av.setFlanking(v[4], v[5], v[6], v[7]);
            if (kind == 'I') cur->haps.back().indels[key] = av;
            else if (kind == 'S') cur->haps.back().snps[key] = av;
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
