// window_io.hpp — the text inputs of the window loop: the realignment-window ("variants") file, the
// library insert-size file and — in place of the reference's candidate-haplotype construction, which is out of this
// repository's scope — a haplotype fixture file.
//   AlignedCandidates, VariantFile::getLineVector     reference VariantFile.hpp:38-70, :188-289
//   LibraryCollection (addFromFile, getMaxInsertSize)  reference Library.hpp:135-253
#ifndef DINDEL_WINDOW_IO_HPP
#define DINDEL_WINDOW_IO_HPP
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "dindel_types.hpp"

namespace dindel {

// one realignment window with its candidate variants — reference VariantFile.hpp:38-70
class AlignedCandidates {
public:
    AlignedCandidates() : centerPos(0), leftPos(0), rightPos(0) {}
    AlignedCandidates(const std::string &_tid, const std::vector<AlignedVariant> &_variants, int _leftPos, int _rightPos)
        : variants(_variants), tid(_tid), leftPos(_leftPos), rightPos(_rightPos) { centerPos = leftPos + (rightPos - leftPos) / 2; }
    std::vector<AlignedVariant> variants;
    std::string tid;
    int centerPos, leftPos, rightPos;
    const AlignedVariant *findVariant(int pos, int type, const std::string &str) const
    {
        for (size_t x = 0; x < variants.size(); x++) if (variants[x].isEqual(pos, type, str)) return &variants[x];
        return NULL;
    }
};

// the window file: "tid leftPos rightPos pos,variant[,prior[,addCombinatorially]] ..." per line — VariantFile.hpp:188-289
class VariantFile {
public:
    explicit VariantFile(const std::string &fileName);     // throws std::string("Cannot open variant file ...")
    bool eof() { return fin.eof(); }
    // the next window; an AlignedCandidates without variants for an empty or unparsable line (the caller skips those).
    // Throws std::string("Cannot read left boundary of region.") like the reference.
    AlignedCandidates getLineVector(bool isOneBased = false);
private:
    std::ifstream fin;
    int index;
};

// reference Library.hpp:135-253: name -> Library; "single_end" (a flat 2000-bin library) always exists
class LibraryCollection : public std::map<std::string, Library> {
public:
    LibraryCollection();
    void addFromFile(const std::string &fileName);        // "#LIB name" headers followed by "insertSize count" lines
    double getMaxInsertSize() const;
};

// Candidate haplotypes of one window as DetInDel::getHaplotypes would leave them (DInDel.cpp:1526-1645): the block bounds it
// settled on and, per haplotype, the sequence and its variants against the reference (Haplotype::indels / ::snps with read /
// flank coordinates).  Text format, one record per line:
//   W <index> <leftPos> <rightPos>
//   H <sequence>
//   V I|S <key> <string> <startHap> <endHap> <startRead> <endRead> <leftFlankHap> <rightFlankHap> <leftFlankRead> <rightFlankRead>
//   A <refpos of base 0> <refpos of base 1> ...      optional, after H: the haplotype's alignment to the window's reference
//                                                    sequence (hap.ml.hpos; -1 = inserted base), needed by --outputRealignedBAM
struct WindowHaplotypes {
    int index; uint32_t leftPos, rightPos;
    std::vector<Haplotype> haps;
};
// The file is indexed when it is opened (where each W record starts; a later record of an index replaces an earlier one) and a window's
// records are parsed the first time it is asked for — by the thread that asks, so the workers that prepare windows side by side share
// the parsing and a run does not wait for a 300-MB file to be read through before its first window.  A malformed record therefore
// surfaces at find() of its window, as a HaplotypeFixture::Error (not a std::string: it is not one window's problem but the run's).
class HaplotypeFixture {
public:
    struct Error { std::string message; };
    explicit HaplotypeFixture(const std::string &fileName);   // throws std::string: cannot open; malformed records in front of the first W
    ~HaplotypeFixture();
    const WindowHaplotypes *find(int index) const;        // NULL: no haplotypes given for that window.  Thread-safe.
    // the caller is done with what find(index) returned: the parsed records are dropped (a later find() parses them again), so that a
    // long run holds the windows in flight only
    void release(int index) const;
private:
    struct Entry { size_t begin, end; int lineBase; mutable int state; mutable WindowHaplotypes win; mutable std::string error; };
    std::string fileName_;
    const char *text_; size_t size_; bool mapped_;
    std::map<int, Entry> windows;
    mutable std::mutex locks_[64];
    HaplotypeFixture(const HaplotypeFixture &); HaplotypeFixture &operator=(const HaplotypeFixture &);
};

} // namespace dindel
#endif
