// compute_likelihoods.cpp — see compute_likelihoods.hpp.
#include "compute_likelihoods.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include "../../include/dindel_hmm.h"

namespace dindel {

double Read::phredToProb(double phred)
{   // reference Read.hpp:127-131 / 143-148
    double q = (1.0 - pow(10.0, -phred / 10.0));
    if (q < 0.0 || q > 1.0 || std::isnan(q) || std::isinf(q)) throw std::string("Phred error.");
    if (q < 1e-16) q = 1e-16;
    if (q > 1.0 - 1e-16) q = 1.0 - 1e-16;
    return q;
}

// reference Library.hpp:78-128
void Library::calcProb(const std::vector<double> &counts)
{
    int max_isize = 2000;
    double max_count = -1;
    for (int s = 0; s < int(counts.size()); s++)
        if (counts[size_t(s)] >= max_count) { max_count = double(int(counts[size_t(s)])); max_isize = s; }   // max_count is an int there
    maxins = 25 * max_isize;
    if (maxins > int(counts.size())) maxins = int(counts.size());
    probs.assign(size_t(maxins), 0.0);
    double z = 0.0;
    for (int d = 0; d < maxins; d++) { probs[size_t(d)] = counts[size_t(d)]; z += probs[size_t(d)]; }
    for (int d = 0; d < maxins; d++) {
        probs[size_t(d)] /= z;
        if (probs[size_t(d)] < 1e-10) probs[size_t(d)] = 1e-10;
    }
    std::vector<double> sorted(probs);
    std::sort(sorted.begin(), sorted.end());
    double sum = 0.0;
    ninetyfifth_pct_prob = sorted.empty() ? 0.0 : sorted.back();
    for (int x = int(sorted.size()) - 1; x > 0; x--) {
        sum += sorted[size_t(x)];
        if (sum > 0.95) { ninetyfifth_pct_prob = sorted[size_t(x)]; break; }
    }
}

static dd_params to_abi(const ObservationModelParameters &o)
{
    dd_params p;
    p.pError = o.pError; p.pMut = o.pMut; p.pFirstgLO = o.pFirstgLO; p.mapQualThreshold = o.mapQualThreshold;
    p.checkBaseQualThreshold = o.checkBaseQualThreshold; p.maxLengthDel = o.maxLengthDel; p.padCover = o.padCover;
    p.bMid = o.bMid; p.forceReadOnHaplotype = o.forceReadOnHaplotype ? 1 : 0; p.mapUnmappedReads = o.mapUnmappedReads ? 1 : 0;
    p.maxMismatch = o.maxMismatch;
    p.capMapQualFast = o.capMapQualFast;
    return p;
}

// Rebuilds what reportVariants derives from mapState (reference ObservationModelFB.cpp:1351-1475), from
// hpos: >=0 haplotype index, -3 LO, -4 RO, inserted base = DD_HPOS_INS_KEY0 - pos as the library writes it (pos = the x of
// the inserted state numS+x, which keys ml.indels, :1380).  A bare MLAlignment::INS (-1: an array in the reference's own
// coding) is accepted too; its key is then taken from the neighbouring entries — (previous on-haplotype base)+1 or the
// next on-haplotype base — which is what the model's transitions imply whenever the run has such a neighbour.
void LikelihoodEngine::rebuildAlignment(const Haplotype &hap, const Read &read, const int16_t *hp,
                                        const ObservationModelParameters &p, MLAlignment &ml)
{
    const int L = int(read.size()), Hs = int(hap.size());
    ml.align = std::string(Hs, 'R');
    ml.indels.clear(); ml.snps.clear(); ml.hapIndelCovered.clear(); ml.hapSNPCovered.clear();
    ml.hpos.assign(L, 0);
    ml.firstBase = -1; ml.lastBase = -1;
    ml.nBQT = 0; ml.nmmBQT = 0; ml.mLogBQ = 0.0; ml.nMMRight = 0; ml.nMMLeft = 0; ml.numIndels = 0; ml.numMismatch = 0;
    for (int b = 0; b < L; b++) ml.hpos[b] = DD_HPOS_IS_INS(hp[b]) ? int(MLAlignment::INS) : int(hp[b]);
    int b = 0;
    while (b < L) {
        const int h = hp[b];
        if (DD_HPOS_IS_INS(h)) {
            // inserted state numS+x: x-1 is the last haplotype base the read consumed before the run (an insertion
            // is entered from "on base x", ObservationModelFB.cpp:1823-1826 / 1746-1749), so pos = x (:1380)
            int rpos = b, len = 0;
            while (b < L && DD_HPOS_IS_INS(hp[b])) { b++; len++; }
            int pos;
            if (h != MLAlignment::INS) pos = DD_HPOS_INS_POS(h);  // the key the device recorded
            else if (rpos > 0 && hp[rpos - 1] >= 0) pos = hp[rpos - 1] + 1;
            else if (b < L && hp[b] >= 0) pos = hp[b];           // run starts the read: next on-haplotype base
            else pos = Hs;                                       // bare code, run followed by RO: inserted at the last base
            std::string seq = read.seq.seq.substr(rpos, len);
            ml.indels[pos] = AlignedVariant(std::string("+").append(seq), pos, pos, rpos, b - 1);
            ml.numIndels++;
            continue;
        }
        if (h >= 0) {
            if (ml.firstBase == -1) ml.firstBase = h; else if (h < ml.firstBase) ml.firstBase = h;
            if (ml.lastBase == -1) ml.lastBase = h; else if (h > ml.lastBase) ml.lastBase = h;
            if (read.qual[b] > p.checkBaseQualThreshold) {
                ml.nBQT++;
                ml.mLogBQ += log10(1.0 - read.qual[b]);
            }
            if (read.seq[b] != hap.seq[h]) {
                std::string snp;
                snp += hap.seq[h];
                snp.append("=>");
                snp += read.seq[b];
                if (read.qual[b] > p.checkBaseQualThreshold) ml.nmmBQT++;
                if (b < 6) ml.nMMLeft++;
                if (b > L - 6) ml.nMMRight++;
                if (read.qual[b] > 0.95) ml.numMismatch++;
                ml.snps[h] = AlignedVariant(snp, h, h, b, b);
                ml.align[h] = read.seq[b];
            }
            if (b < L - 1) {
                // next state on a base (or right of the haplotype) more than one position further: deletion (:1437-1453)
                const int nh = hp[b + 1];
                int ns = -1;                       // haplotype-state index (s = h+1); RO = Hs+1
                if (nh >= 0) ns = nh + 1;
                else if (nh == MLAlignment::RO) ns = Hs + 1;
                else if (nh == MLAlignment::LO) ns = 0;
                const int s = h + 1;
                if (ns >= 0 && ns - s > 1) {
                    const int pos = s;
                    const int len = ns - s - 1;
                    for (int y = pos; y < len + pos && y < Hs; y++) ml.align[y] = 'D';
                    std::string seq = hap.seq.substr(pos, len);
                    ml.indels[pos] = AlignedVariant(std::string("-").append(seq), pos, pos + len - 1, b, b + 1);
                    ml.numIndels++;
                }
            }
        }
        b++;
    }
    for (std::map<int, AlignedVariant>::const_iterator it = hap.indels.begin(); it != hap.indels.end(); ++it)
        ml.hapIndelCovered[it->first] = it->second.isCovered(p.padCover, ml.firstBase, ml.lastBase);
    for (std::map<int, AlignedVariant>::const_iterator it = hap.snps.begin(); it != hap.snps.end(); ++it)
        ml.hapSNPCovered[it->first] = it->second.isCovered(p.padCover, ml.firstBase, ml.lastBase);
}

void LikelihoodEngine::rebuildAlignmentFaster(const Haplotype &hap, const Read &read, const int16_t *hp,
                                              const ObservationModelParameters &p, MLAlignment &ml)
{
    const int L = int(read.size()), Hs = int(hap.size());
    ml.align = std::string(Hs, 'R');
    ml.indels.clear(); ml.snps.clear(); ml.hapIndelCovered.clear(); ml.hapSNPCovered.clear();
    ml.hpos.assign(size_t(L), 0);
    for (int i = 0; i < L; i++) ml.hpos[size_t(i)] = DD_HPOS_IS_INS(hp[i]) ? int(MLAlignment::INS) : int(hp[i]);
    ml.firstBase = -1; ml.lastBase = -1;
    int lhp = 1;                                       // Faster.cpp:553 (only used for a bare INS code)
    int b = 0;
    while (b < L) {
        const int h = hp[b];
        if (DD_HPOS_IS_INS(h)) {                       // Faster.cpp:606-621
            const int rpos = b;
            int len = 0;
            while (b < L && DD_HPOS_IS_INS(hp[b])) { b++; len++; }
            const int pos = (h != MLAlignment::INS) ? DD_HPOS_INS_POS(h) : lhp;   // the key the device recorded (lhp of :556-566)
            ml.indels[pos] = AlignedVariant(std::string("+").append(read.seq.seq.substr(rpos, len)), pos, pos, rpos, b - 1);
            continue;
        }
        if (h >= 0) {
            lhp = h + 1;
            if (ml.firstBase == -1) ml.firstBase = h; else if (h < ml.firstBase) ml.firstBase = h;
            if (ml.lastBase == -1) ml.lastBase = h; else if (h > ml.lastBase) ml.lastBase = h;
            if (read.seq[b] != hap.seq[h]) {           // :631-643
                std::string snp;
                snp += hap.seq[h];
                snp.append("=>");
                snp += read.seq[b];
                ml.snps[h] = AlignedVariant(snp, h, h, b, b);
                ml.align[h] = read.seq[b];
            }
            if (b < L - 1 && hp[b + 1] >= 0 && hp[b + 1] - h > 1) {     // :645-661: next state on a later base
                const int pos = h + 1, len = hp[b + 1] - h - 1;
                for (int y = pos; y < pos + len && y < Hs; y++) ml.align[y] = 'D';
                ml.indels[pos] = AlignedVariant(std::string("-").append(hap.seq.substr(pos, len)), pos, pos + len - 1, b, b + 1);
            }
        }
        b++;
    }
    for (std::map<int, AlignedVariant>::const_iterator it = hap.indels.begin(); it != hap.indels.end(); ++it)
        ml.hapIndelCovered[it->first] = it->second.isCovered(p.padCover, ml.firstBase, ml.lastBase);
    for (std::map<int, AlignedVariant>::const_iterator it = hap.snps.begin(); it != hap.snps.end(); ++it)
        ml.hapSNPCovered[it->first] = it->second.isCovered(p.padCover, ml.firstBase, ml.lastBase);
}

// ------------------------------------------------------------------------------------------------------------------------
// Flat buffers of a batch.  Result arrays are not value-initialised (4 GB of zero-fill per 10,000 windows would cost more
// than the kernels) and live in page-locked memory when the device gives it (dd_host_alloc): the D2H copies then run at
// link speed and overlap the kernels.
template <class T> struct RawBuf {
    T *p; size_t cap; bool pinned;
    RawBuf() : p(NULL), cap(0), pinned(false) {}
    ~RawBuf() { release(); }
    void release() { if (p) { if (pinned) dd_host_free(p); else free(p); } p = NULL; cap = 0; }
    T *reserve(size_t n)
    {
        if (n <= cap) return p;
        release();
        const size_t want = n + n / 8 + 64;
        p = static_cast<T *>(dd_host_alloc(want * sizeof(T)));
        pinned = p != NULL;
        if (!p) p = static_cast<T *>(malloc(want * sizeof(T)));
        if (!p) throw std::string("out of memory");
        cap = want;
        return p;
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
private:
    RawBuf(const RawBuf &); RawBuf &operator=(const RawBuf &);
};

struct BatchBlock {
    int W;
    bool faster, has_hpos;
    ObservationModelParameters params;
    std::vector<int32_t> win_hap_off, win_read_off, hap_var_off, read_seq_off;
    std::vector<int64_t> pair_off, hpos_off, vc_off;
    RawBuf<double> ll, llOn, llOff, mLogBQ;
    RawBuf<uint8_t> offHap, offHapHMQ, vcov, fcov;
    RawBuf<int16_t> numIndels, numMismatch, nBQT, nmmBQT, nMMLeft, nMMRight, firstBase, lastBase, hpos;
    RawBuf<int32_t> status;
    BatchBlock() : W(0), faster(false), has_hpos(false) {}
};

struct PackScratch {
    std::vector<int32_t> hap_seq_off, hap_var, hap_var_flank;
    std::vector<uint32_t> win_hap_start, read_start;
    std::vector<char> hap_seq;
    RawBuf<char> read_seq;               // the two big inputs (one byte per read base each) sit in page-locked memory: their
    RawBuf<uint8_t> read_qidx;           // H2D copy then runs at link speed instead of through the driver's staging buffers
    std::vector<uint8_t> read_mqidx, read_flags, read_lib;
    std::vector<int32_t> read_mate_pos, read_mate_len;
};

namespace {

// value -> small index, shared by the packing threads: a lock-free read of the per-thread cache, the shared table under a mutex
// on a miss (at most 256 distinct values per batch: BAM qualities are integer Phred)
class ValueTable {
public:
    std::vector<double> tab;
    int intern(double v)
    {
        std::lock_guard<std::mutex> g(m_);
        for (size_t i = 0; i < tab.size(); i++)
            if (memcmp(&tab[i], &v, sizeof(double)) == 0) return int(i);
        if (tab.size() >= 256) return -1;
        tab.push_back(v);
        return int(tab.size()) - 1;
    }
private:
    std::mutex m_;
};

struct ValueCache {                      // per thread: open addressing on the bit pattern
    uint64_t key[512]; int16_t val[512];
    ValueCache() { for (int i = 0; i < 512; i++) val[i] = -1; }
    int get(double v, ValueTable &T)
    {
        uint64_t k;
        memcpy(&k, &v, 8);
        unsigned h = unsigned((k * 0x9E3779B97F4A7C15ull) >> 55);
        for (;;) {
            if (val[h] < 0) {
                const int idx = T.intern(v);
                if (idx < 0) return -1;
                key[h] = k; val[h] = int16_t(idx);
                return idx;
            }
            if (key[h] == k) return val[h];
            h = (h + 1) & 511;
        }
    }
};

unsigned pick_threads(int hostThreads, int64_t units)
{
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr > 16) nthr = 16;
    if (nthr < 1) nthr = 1;
    if (hostThreads > 0) nthr = unsigned(hostThreads);
    if (int64_t(nthr) > units) nthr = unsigned(units > 0 ? units : 1);
    return nthr;
}

template <class F> void parallel_windows(int W, unsigned nthr, int grain, F f)
{
    if (nthr <= 1 || W <= grain) { for (int w = 0; w < W; w++) f(w); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthr; t++)
        pool.push_back(std::thread([&]() {
            for (int w0 = next.fetch_add(grain); w0 < W; w0 = next.fetch_add(grain))
                for (int w = w0; w < W && w < w0 + grain; w++) f(w);
        }));
    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
}

// one record from the block (what the reference's loop body leaves in liks[h][r], DInDel.cpp:1717-1719)
void fill_record(const BatchBlock &B, int w, size_t h, size_t r, const Haplotype &Hh, const Read &Rd, MLAlignment &ml)
{
    const size_t Rn = size_t(B.win_read_off[w + 1] - B.win_read_off[w]);
    const int r0 = B.win_read_off[w];
    const int g = B.win_hap_off[w] + int(h);
    const int nv = B.hap_var_off[g + 1] - B.hap_var_off[g];
    const int64_t p = B.pair_off[w] + int64_t(h) * int64_t(Rn) + int64_t(r);
    const int64_t SL = int64_t(B.read_seq_off[B.win_read_off[w + 1]]) - B.read_seq_off[r0];
    const int L = int(Rd.size()), Hs = int(Hh.size());
    const int16_t *hp = B.hpos.p + B.hpos_off[w] + int64_t(h) * SL + (B.read_seq_off[r0 + r] - B.read_seq_off[r0]);
    const int64_t vb = B.vc_off[w] + int64_t(B.hap_var_off[g] - B.hap_var_off[B.win_hap_off[w]]) * int64_t(Rn) + int64_t(r) * nv;
    // Per pair the device already delivers every counter of the record; the variant maps and the `align` string need the
    // per-base walk (rebuildAlignment) only when the read shows an indel or differs from the haplotype segment it sits on.
    bool plain = false;                       // gap-free, mismatch-free placement: nothing to list
    if (!B.faster && B.numIndels[p] == 0) {
        if (B.firstBase[p] < 0) plain = true;   // no base on the haplotype
        else {
            int b0 = 0;
            while (b0 < L && hp[b0] < 0) b0++;
            const int n = B.lastBase[p] - B.firstBase[p] + 1;
            plain = b0 + n <= L && hp[b0] == B.firstBase[p] &&
                    memcmp(Rd.seq.seq.data() + b0, Hh.seq.data() + B.firstBase[p], size_t(n)) == 0;
        }
    }
    if (plain) {
        ml.align = std::string(size_t(Hs), 'R');
        ml.indels.clear(); ml.snps.clear(); ml.hapIndelCovered.clear(); ml.hapSNPCovered.clear();
        ml.hpos.assign(hp, hp + L);              // no inserted base in a plain placement: the codes are the reference's
        ml.firstBase = B.firstBase[p]; ml.lastBase = B.lastBase[p];
        ml.numIndels = 0; ml.numMismatch = B.numMismatch[p]; ml.nBQT = B.nBQT[p]; ml.nmmBQT = B.nmmBQT[p];
        ml.nMMLeft = B.nMMLeft[p]; ml.nMMRight = B.nMMRight[p];
        int i = 0;
        for (std::map<int, AlignedVariant>::const_iterator it = Hh.indels.begin(); it != Hh.indels.end(); ++it, ++i)
            ml.hapIndelCovered[it->first] = B.vcov[vb + i] != 0;
        for (std::map<int, AlignedVariant>::const_iterator it = Hh.snps.begin(); it != Hh.snps.end(); ++it, ++i)
            ml.hapSNPCovered[it->first] = B.vcov[vb + i] != 0;
    } else if (B.faster) LikelihoodEngine::rebuildAlignmentFaster(Hh, Rd, hp, B.params, ml);
    else LikelihoodEngine::rebuildAlignment(Hh, Rd, hp, B.params, ml);
    ml.ll = B.ll[p]; ml.llOn = B.llOn[p]; ml.llOff = B.llOff[p];
    ml.offHap = B.offHap[p] != 0; ml.offHapHMQ = B.offHapHMQ[p] != 0;
    if (!B.faster) ml.mLogBQ = B.mLogBQ[p];           // the device's serial sum (same order as the reference)
    ml.hapIndelFilterCovered.clear();
    {   // per haplotype-indel coverage flags of filterHaplotypes, in hap.indels map order (first nvI of the hap's list)
        int i = 0;
        for (std::map<int, AlignedVariant>::const_iterator it = Hh.indels.begin(); it != Hh.indels.end(); ++it, ++i)
            ml.hapIndelFilterCovered[it->first] = B.fcov[vb + i] != 0;
    }
}

} // namespace

// ---------------- WindowLikelihoods: the lazy view ----------------
size_t WindowLikelihoods::numHaps() const { return size_t(blk_->win_hap_off[w_ + 1] - blk_->win_hap_off[w_]); }
size_t WindowLikelihoods::numReads() const { return size_t(blk_->win_read_off[w_ + 1] - blk_->win_read_off[w_]); }
int64_t WindowLikelihoods::pair(size_t h, size_t r) const { return blk_->pair_off[w_] + int64_t(h) * int64_t(numReads()) + int64_t(r); }
double WindowLikelihoods::ll(size_t h, size_t r) const { return blk_->ll[pair(h, r)]; }
double WindowLikelihoods::llOn(size_t h, size_t r) const { return blk_->llOn[pair(h, r)]; }
double WindowLikelihoods::llOff(size_t h, size_t r) const { return blk_->llOff[pair(h, r)]; }
double WindowLikelihoods::mLogBQ(size_t h, size_t r) const { return blk_->mLogBQ[pair(h, r)]; }
bool WindowLikelihoods::offHap(size_t h, size_t r) const { return blk_->offHap[pair(h, r)] != 0; }
bool WindowLikelihoods::offHapHMQ(size_t h, size_t r) const { return blk_->offHapHMQ[pair(h, r)] != 0; }
int WindowLikelihoods::numIndels(size_t h, size_t r) const { return blk_->numIndels[pair(h, r)]; }
int WindowLikelihoods::indelCount(size_t h, size_t r) const
{
    const BatchBlock &B = *blk_;
    if (!B.faster) return B.numIndels[pair(h, r)];
    if (!B.has_hpos) return int(get(h, r).indels.size());
    // --faster: the size of the record's indel map needs the walk over hpos (rebuildAlignmentFaster), but not the record:
    // the keys that walk would enter — an insertion run's recorded position, a deletion's first base — counted once each
    const int r0 = B.win_read_off[w_];
    const int64_t SL = int64_t(B.read_seq_off[B.win_read_off[w_ + 1]]) - B.read_seq_off[r0];
    const int16_t *hp = B.hpos.p + B.hpos_off[w_] + int64_t(h) * SL + (B.read_seq_off[size_t(r0) + r] - B.read_seq_off[r0]);
    const int L = int((*reads_)[r].size());
    int keys[64], nk = 0, lhp = 1, b = 0;
    bool spilled = false;
    while (b < L && !spilled) {
        const int c = hp[b];
        int key = -1;
        if (DD_HPOS_IS_INS(c)) {
            while (b < L && DD_HPOS_IS_INS(hp[b])) b++;
            key = (c != MLAlignment::INS) ? DD_HPOS_INS_POS(c) : lhp;
        } else {
            if (c >= 0) { lhp = c + 1; if (b < L - 1 && hp[b + 1] >= 0 && hp[b + 1] - c > 1) key = c + 1; }
            b++;
        }
        if (key >= 0) {
            bool seen = false;
            for (int k = 0; k < nk; k++) if (keys[k] == key) seen = true;
            if (!seen) { if (nk == 64) spilled = true; else keys[nk++] = key; }
        }
    }
    return spilled ? int(get(h, r).indels.size()) : nk;
}
int WindowLikelihoods::numMismatch(size_t h, size_t r) const { return blk_->numMismatch[pair(h, r)]; }
int WindowLikelihoods::nBQT(size_t h, size_t r) const { return blk_->nBQT[pair(h, r)]; }
int WindowLikelihoods::nmmBQT(size_t h, size_t r) const { return blk_->nmmBQT[pair(h, r)]; }
int WindowLikelihoods::nMMLeft(size_t h, size_t r) const { return blk_->nMMLeft[pair(h, r)]; }
int WindowLikelihoods::nMMRight(size_t h, size_t r) const { return blk_->nMMRight[pair(h, r)]; }
int WindowLikelihoods::firstBase(size_t h, size_t r) const { return blk_->firstBase[pair(h, r)]; }
int WindowLikelihoods::lastBase(size_t h, size_t r) const { return blk_->lastBase[pair(h, r)]; }
// onHap[r] = 1 if the read lies on some haplotype under the artificial mapping quality (DInDel.cpp:1720: !liks[hidx][r].offHapHMQ for any hidx).
// Taken from the offHapHMQ flags here instead of from the device's dd_onhap_kernel: one kernel (and one kernel boundary, ~0.15 ms on
// this part) less per batch, for a value only the realigned-BAM step reads.
int WindowLikelihoods::onHap(size_t r) const
{
    const size_t nh = numHaps();
    for (size_t h = 0; h < nh; h++) if (!blk_->offHapHMQ[pair(h, r)]) return 1;
    return 0;
}

// slot of the haplotype's variant `key` in the per-pair flag list: hap.indels in map order, then hap.snps (-1: no such variant)
int WindowLikelihoods::varSlot(size_t h, int key, bool snp) const
{
    const Haplotype &H = (*haps_)[h];
    const std::map<int, AlignedVariant> &m = snp ? H.snps : H.indels;
    std::map<int, AlignedVariant>::const_iterator it = m.find(key);
    if (it == m.end()) return -1;
    return int(std::distance(m.begin(), it)) + (snp ? int(H.indels.size()) : 0);
}

static inline int64_t var_base(const BatchBlock &B, int w, size_t h, size_t r)
{
    const int g = B.win_hap_off[w] + int(h);
    const int64_t Rn = B.win_read_off[w + 1] - B.win_read_off[w];
    return B.vc_off[w] + int64_t(B.hap_var_off[g] - B.hap_var_off[B.win_hap_off[w]]) * Rn + int64_t(r) * (B.hap_var_off[g + 1] - B.hap_var_off[g]);
}

bool WindowLikelihoods::hapIndelCovered(size_t h, size_t r, int key) const
{
    const int s = varSlot(h, key, false);
    return s >= 0 && blk_->vcov[var_base(*blk_, w_, h, r) + s] != 0;
}
bool WindowLikelihoods::hapSNPCovered(size_t h, size_t r, int key) const
{
    const int s = varSlot(h, key, true);
    return s >= 0 && blk_->vcov[var_base(*blk_, w_, h, r) + s] != 0;
}
bool WindowLikelihoods::hapIndelFilterCovered(size_t h, size_t r, int key) const
{
    const int s = varSlot(h, key, false);
    return s >= 0 && blk_->fcov[var_base(*blk_, w_, h, r) + s] != 0;
}

bool WindowLikelihoods::coveredAt(size_t h, size_t r, int slot) const { return slot >= 0 && blk_->vcov[var_base(*blk_, w_, h, r) + slot] != 0; }
bool WindowLikelihoods::filterCoveredAt(size_t h, size_t r, int slot) const { return slot >= 0 && blk_->fcov[var_base(*blk_, w_, h, r) + slot] != 0; }

WindowLikelihoods::Rows WindowLikelihoods::rows(size_t h) const
{
    const BatchBlock &B = *blk_;
    const int64_t p = pair(h, 0);
    const int g = B.win_hap_off[w_] + int(h);
    Rows R;
    R.ll = B.ll.p + p; R.llOn = B.llOn.p + p; R.llOff = B.llOff.p + p; R.mLogBQ = B.mLogBQ.p + p;
    R.offHap = B.offHap.p + p; R.offHapHMQ = B.offHapHMQ.p + p;
    R.numIndels = B.numIndels.p + p; R.numMismatch = B.numMismatch.p + p; R.nBQT = B.nBQT.p + p; R.nmmBQT = B.nmmBQT.p + p;
    R.nMMLeft = B.nMMLeft.p + p; R.nMMRight = B.nMMRight.p + p; R.firstBase = B.firstBase.p + p; R.lastBase = B.lastBase.p + p;
    R.nv = B.hap_var_off[g + 1] - B.hap_var_off[g];
    const int64_t vb = var_base(B, w_, h, 0);
    R.vcov = B.vcov.p + vb; R.fcov = B.fcov.p + vb;
    return R;
}

MLAlignment WindowLikelihoods::get(size_t h, size_t r) const
{
    MLAlignment ml;
    if (blk_->has_hpos) { fill_record(*blk_, w_, h, r, (*haps_)[h], (*reads_)[r], ml); return ml; }
    if (!full_) {                          // the batch kept no alignments: this window once more, with them
        if (!eng_) throw std::string("WindowLikelihoods::get: batch was run without alignments and the engine is gone");
        std::vector<WindowJob> one(1);
        one[0].haps = haps_; one[0].reads = reads_; one[0].leftPos = leftPos_; one[0].rightPos = leftPos_ + 1;
        const bool keep = eng_->keepAlignments_;
        eng_->keepAlignments_ = true;
        try { eng_->runBatch(one, blk_->faster); } catch (...) { eng_->keepAlignments_ = keep; throw; }
        eng_->keepAlignments_ = keep;
        if (!one[0].error.empty()) throw one[0].error;
        full_ = one[0].result.blk_;
    }
    fill_record(*full_, 0, h, r, (*haps_)[h], (*reads_)[r], ml);
    return ml;
}

void WindowLikelihoods::toLiks(std::vector<std::vector<MLAlignment> > &liks, std::vector<int> &onHapV) const
{
    const size_t H = numHaps(), Rn = numReads();
    onHapV = std::vector<int>(Rn, 0);
    liks = std::vector<std::vector<MLAlignment> >(H, std::vector<MLAlignment>(Rn));
    for (size_t h = 0; h < H; h++)
        for (size_t r = 0; r < Rn; r++) liks[h][r] = get(h, r);
    for (size_t r = 0; r < Rn; r++) onHapV[r] = onHap(r);
}

// ---------------- LikelihoodEngine ----------------
void LikelihoodEngine::computeLikelihoodsFaster(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                                                std::vector<std::vector<MLAlignment> > &liks, uint32_t leftPos,
                                                uint32_t rightPos, std::vector<int> &onHap)
{
    std::vector<WindowJob> jobs(1);
    jobs[0].haps = &haps; jobs[0].reads = &reads; jobs[0].leftPos = leftPos; jobs[0].rightPos = rightPos;
    jobs[0].liks = &liks; jobs[0].onHap = &onHap;
    runBatch(jobs, true);
    if (!jobs[0].error.empty()) throw jobs[0].error;
}

void LikelihoodEngine::warmUp(size_t pairs)
{
    // per pair: ~25 input bytes (a read of 100 bases is shared by its window's haplotypes) and ~270 result bytes if everything is staged
    const size_t bytes = pairs * 300 + (size_t(32) << 20);
    if (dd_reserve_cache(device_, bytes, bytes <= (size_t(64) << 20) ? bytes : 0) != DD_SUCCESS) throw std::string("dd_reserve_cache: ") + dd_last_error();
    // two result blocks with their per-pair arrays at that size, made now (page-locking 90 MB takes 40-100 ms): a pipelined caller
    // would otherwise grow its blocks step by step under its first batches.  The per-base and per-variant arrays follow the data.
    const size_t np = pairs + 1;
    while (spare_.size() < 2) {
        std::shared_ptr<BatchBlock> blk = std::make_shared<BatchBlock>();
        BatchBlock &B = *blk;
        B.ll.reserve(np); B.llOn.reserve(np); B.llOff.reserve(np); B.mLogBQ.reserve(np);
        B.offHap.reserve(np); B.offHapHMQ.reserve(np); B.status.reserve(np);
        B.numIndels.reserve(np); B.numMismatch.reserve(np); B.nBQT.reserve(np); B.nmmBQT.reserve(np);
        B.nMMLeft.reserve(np); B.nMMRight.reserve(np); B.firstBase.reserve(np); B.lastBase.reserve(np);
        spare_.push_back(blk);
    }
}

void LikelihoodEngine::computeLikelihoodsFasterBatch(std::vector<WindowJob> &jobs) { runBatch(jobs, true); }

void LikelihoodEngine::computeLikelihoodsBatch(std::vector<WindowJob> &jobs) { runBatch(jobs, false); }

void LikelihoodEngine::computeLikelihoods(const std::vector<Haplotype> &haps, const std::vector<Read> &reads,
                                          std::vector<std::vector<MLAlignment> > &liks, uint32_t leftPos,
                                          uint32_t rightPos, std::vector<int> &onHap)
{
    std::vector<WindowJob> jobs(1);
    jobs[0].haps = &haps; jobs[0].reads = &reads; jobs[0].leftPos = leftPos; jobs[0].rightPos = rightPos;
    jobs[0].liks = &liks; jobs[0].onHap = &onHap;
    computeLikelihoodsBatch(jobs);
    if (!jobs[0].error.empty()) throw jobs[0].error;
}

void LikelihoodEngine::runBatch(std::vector<WindowJob> &jobs, bool faster)
{
    const std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();
    const int W = int(jobs.size());
    // The result block of an earlier call is reused when no view refers to it any more (its pages are mapped and, when
    // pinned, registered with the device already); otherwise a new one is made and the old one lives on with its views.
    for (int w = 0; w < W; w++) jobs[w].result = WindowLikelihoods();      // views of an earlier call held by these jobs
    // A pipelined caller (dindel_gpu: batch k+1 computes while batch k reduces) keeps two or three blocks alive in turn,
    // so a few are remembered.
    std::shared_ptr<BatchBlock> blk;
    for (size_t i = 0; i < spare_.size() && !blk; i++) if (spare_[i].use_count() == 1) blk = spare_[i];
    if (!blk) {
        blk = std::make_shared<BatchBlock>();
        if (spare_.size() < 4) spare_.push_back(blk);
        else spare_[size_t(spareNext_++ % 4)] = blk;
    }
    if (!scratch_) scratch_ = std::make_shared<PackScratch>();
    BatchBlock &B = *blk;
    PackScratch &S = *scratch_;
    B.W = W; B.faster = faster; B.params = params;
    bool any_eager = false;
    for (int w = 0; w < W; w++) if (jobs[w].liks) any_eager = true;
    B.has_hpos = keepAlignments_ || any_eager;

    // ---- pack (CSR), pass 1: sizes and offsets.  Per window in parallel (its haplotypes and reads are scattered objects: the pass is
    // cache misses), one serial prefix sum over the windows in between ----
    const bool with_mates = params.mapUnmappedReads && !faster;
    const unsigned nthr1 = pick_threads(hostThreads_, W);
    std::vector<int64_t> whb(size_t(W) + 1, 0), wrb(size_t(W) + 1, 0), wvar(size_t(W) + 1, 0);       // per window: haplotype bases, read bases, variants
    B.win_hap_off.assign(size_t(W) + 1, 0); B.win_read_off.assign(size_t(W) + 1, 0);
    std::atomic<int> bad_read(0);
    parallel_windows(W, nthr1, 16, [&](int w) {
        WindowJob &J = jobs[w];
        J.error.clear();
        int64_t hb = 0, rb = 0, nv = 0;
        for (size_t h = 0; h < J.haps->size(); h++) { const Haplotype &H = (*J.haps)[h]; hb += int64_t(H.seq.size()); nv += int64_t(H.indels.size() + H.snps.size()); }
        for (size_t r = 0; r < J.reads->size(); r++) {
            const Read &R = (*J.reads)[r];
            if (R.qual.size() != R.size()) bad_read.store(1);
            rb += int64_t(R.size());
        }
        whb[size_t(w) + 1] = hb; wrb[size_t(w) + 1] = rb; wvar[size_t(w) + 1] = nv;
        B.win_hap_off[size_t(w) + 1] = int32_t(J.haps->size()); B.win_read_off[size_t(w) + 1] = int32_t(J.reads->size());
    });
    if (bad_read.load()) throw std::string("Read: qual and seq differ in length");
    for (int w = 0; w < W; w++) {
        whb[size_t(w) + 1] += whb[size_t(w)]; wrb[size_t(w) + 1] += wrb[size_t(w)]; wvar[size_t(w) + 1] += wvar[size_t(w)];
        B.win_hap_off[size_t(w) + 1] += B.win_hap_off[size_t(w)]; B.win_read_off[size_t(w) + 1] += B.win_read_off[size_t(w)];
        if (whb[size_t(w) + 1] > 0x7fffffffLL || wrb[size_t(w) + 1] > 0x7fffffffLL)
            throw std::string("batch too large: more than 2^31 haplotype or read bases (split the windows over several calls)");
    }
    const int64_t n_hap_bases = whb[size_t(W)], n_read_bases = wrb[size_t(W)];
    B.hap_var_off.resize(size_t(B.win_hap_off[size_t(W)]) + 1); S.hap_seq_off.resize(size_t(B.win_hap_off[size_t(W)]) + 1);
    B.read_seq_off.resize(size_t(B.win_read_off[size_t(W)]) + 1);
    B.hap_var_off[0] = 0; S.hap_seq_off[0] = 0; B.read_seq_off[0] = 0;
    parallel_windows(W, nthr1, 16, [&](int w) {
        const WindowJob &J = jobs[w];
        int64_t hb = whb[size_t(w)], rb = wrb[size_t(w)], nv = wvar[size_t(w)];
        const size_t g0 = size_t(B.win_hap_off[size_t(w)]), q0 = size_t(B.win_read_off[size_t(w)]);
        for (size_t h = 0; h < J.haps->size(); h++) {
            const Haplotype &H = (*J.haps)[h];
            hb += int64_t(H.seq.size()); nv += int64_t(H.indels.size() + H.snps.size());
            S.hap_seq_off[g0 + h + 1] = int32_t(hb);
            B.hap_var_off[g0 + h + 1] = int32_t(nv);
        }
        for (size_t r = 0; r < J.reads->size(); r++) { rb += int64_t((*J.reads)[r].size()); B.read_seq_off[q0 + r + 1] = int32_t(rb); }
    });
    std::vector<int32_t> lib_off(1, 0);
    std::vector<double> lib_prob, lib_p95;
    if (with_mates) {                                    // libraries in order of first appearance (serial: the order defines the indices)
        std::map<const Library *, int> lib_index;
        S.read_lib.clear();
        for (int w = 0; w < W; w++) {
            const WindowJob &J = jobs[w];
            for (size_t r = 0; r < J.reads->size(); r++) {
                const Read &R = (*J.reads)[r];
                int li = 0;
                if (R.isPaired()) {                      // the reference dereferences the library of paired reads only (:279-289)
                    if (!R.library) throw std::string("Cannot find library: ");            // Read.hpp:176
                    std::map<const Library *, int>::iterator lt = lib_index.find(R.library);
                    if (lt == lib_index.end()) {
                        if (lib_index.size() >= 256) throw std::string("more than 256 libraries in one batch");
                        lt = lib_index.insert(std::make_pair(R.library, int(lib_index.size()))).first;
                        lib_prob.insert(lib_prob.end(), R.library->table().begin(), R.library->table().end());
                        lib_off.push_back(int32_t(lib_prob.size()));
                        lib_p95.push_back(R.library->getNinetyFifthPctProb());
                    }
                    li = lt->second;
                }
                S.read_lib.push_back(uint8_t(li));
            }
        }
    }
    const size_t n_haps = size_t(B.win_hap_off[W]), n_reads = size_t(B.win_read_off[W]), n_var = size_t(B.hap_var_off[n_haps]);
    S.win_hap_start.resize(size_t(W)); S.hap_seq.resize(size_t(n_hap_bases) + 1); S.read_seq.reserve(size_t(n_read_bases) + 1);
    S.read_qidx.reserve(size_t(n_read_bases) + 1); S.read_mqidx.resize(n_reads + 1); S.read_start.resize(n_reads + 1);
    S.read_flags.resize(n_reads + 1); S.hap_var.resize(2 * n_var + 1); S.hap_var_flank.resize(3 * n_var + 1);
    if (with_mates) { S.read_mate_pos.resize(n_reads + 1); S.read_mate_len.resize(n_reads + 1); }

    const std::chrono::steady_clock::time_point t_pass1 = std::chrono::steady_clock::now();
    // ---- pass 2: the bytes, windows in parallel ----
    ValueTable qtab, mqtab;
    std::atomic<int> overflow(0);
    const unsigned nthr = pick_threads(hostThreads_, W);
    {
        auto pack_window = [&](int w, ValueCache &qc, ValueCache &mc) {
            const WindowJob &J = jobs[w];
            S.win_hap_start[size_t(w)] = J.leftPos;
            for (size_t h = 0; h < J.haps->size(); h++) {
                const Haplotype &H = (*J.haps)[h];
                const int g = B.win_hap_off[w] + int(h);
                memcpy(S.hap_seq.data() + S.hap_seq_off[size_t(g)], H.seq.data(), H.seq.size());
                int32_t *hv = S.hap_var.data() + 2 * size_t(B.hap_var_off[g]);
                int32_t *hf = S.hap_var_flank.data() + 3 * size_t(B.hap_var_off[g]);
                for (std::map<int, AlignedVariant>::const_iterator it = H.indels.begin(); it != H.indels.end(); ++it) {
                    *hv++ = it->second.getStartRead(); *hv++ = it->second.getEndRead();
                    *hf++ = it->second.getLeftFlankRead(); *hf++ = it->second.getRightFlankRead();
                    *hf++ = it->second.getType() == AlignedVariant::DEL ? 1 : it->second.getType() == AlignedVariant::INS ? 2 : 0;
                }
                for (std::map<int, AlignedVariant>::const_iterator it = H.snps.begin(); it != H.snps.end(); ++it) {
                    *hv++ = it->second.getStartRead(); *hv++ = it->second.getEndRead();
                    *hf++ = 0; *hf++ = 0; *hf++ = 0;
                }
            }
            for (size_t r = 0; r < J.reads->size(); r++) {
                const Read &R = (*J.reads)[r];
                const size_t q = size_t(B.win_read_off[w]) + r;
                const size_t so = size_t(B.read_seq_off[q]);
                memcpy(S.read_seq.p + so, R.seq.seq.data(), R.size());
                uint8_t *qi = S.read_qidx.p + so;
                double lastq = -1.0;
                int lastidx = 0;
                for (size_t b = 0; b < R.qual.size(); b++) {
                    if (R.qual[b] != lastq) {                 // runs of equal qualities skip the lookup
                        const int idx = qc.get(R.qual[b], qtab);
                        if (idx < 0) { overflow.store(1); return; }
                        lastq = R.qual[b]; lastidx = idx;
                    }
                    qi[b] = uint8_t(lastidx);
                }
                const int mi = mc.get(R.mapQual, mqtab);
                if (mi < 0) { overflow.store(2); return; }
                S.read_mqidx[q] = uint8_t(mi);
                S.read_start[q] = uint32_t(R.posStat.first);      // uint32_t(read.posStat.first), ObservationModelFB.cpp:52
                S.read_flags[q] = uint8_t((R.isUnmapped() ? DD_READ_UNMAPPED : 0) | (R.isPaired() ? DD_READ_PAIRED : 0) |
                                          (R.mateIsUnmapped() ? DD_READ_MATE_UNMAPPED : 0) | (R.mateIsReverse() ? DD_READ_MATE_REVERSE : 0) |
                                          (R.mateSameTid ? DD_READ_MATE_SAME_TID : 0));
                if (with_mates) {
                    S.read_mate_pos[q] = R.matePos;
                    S.read_mate_len[q] = R.isPaired() ? R.mateLen : -1;
                }
            }
        };
        if (nthr <= 1 || W < 8) {
            ValueCache qc, mc;
            for (int w = 0; w < W; w++) pack_window(w, qc, mc);
        } else {
            std::atomic<int> next(0);
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nthr; t++)
                pool.push_back(std::thread([&]() {
                    ValueCache qc, mc;
                    for (int w0 = next.fetch_add(8); w0 < W; w0 = next.fetch_add(8))
                        for (int w = w0; w < W && w < w0 + 8; w++) pack_window(w, qc, mc);
                }));
            for (size_t t = 0; t < pool.size(); t++) pool[t].join();
        }
        if (overflow.load() == 1) throw std::string("more than 256 distinct base qualities in one batch");
        if (overflow.load() == 2) throw std::string("more than 256 distinct mapping qualities in one batch");
    }
    dd_batch Bt;
    memset(&Bt, 0, sizeof(Bt));
    Bt.n_windows = W;
    Bt.win_hap_off = B.win_hap_off.data(); Bt.win_read_off = B.win_read_off.data(); Bt.win_hap_start = S.win_hap_start.data();
    Bt.hap_seq_off = S.hap_seq_off.data(); Bt.hap_seq = S.hap_seq.data(); Bt.hap_var_off = B.hap_var_off.data();
    Bt.hap_var = n_var ? S.hap_var.data() : NULL;
    Bt.hap_var_flank = n_var ? S.hap_var_flank.data() : NULL;
    Bt.read_seq_off = B.read_seq_off.data(); Bt.read_seq = S.read_seq.p; Bt.read_qidx = S.read_qidx.p;
    Bt.read_mqidx = S.read_mqidx.data(); Bt.read_start = S.read_start.data(); Bt.read_flags = S.read_flags.data();
    double one_q = 0.5;
    Bt.n_qual = int(qtab.tab.size()); Bt.qual_table = qtab.tab.empty() ? &one_q : qtab.tab.data();
    Bt.n_mapq = int(mqtab.tab.size()); Bt.mapq_table = mqtab.tab.empty() ? &one_q : mqtab.tab.data();
    if (with_mates) {
        if (lib_p95.empty()) { lib_prob.push_back(1.0); lib_off.push_back(1); lib_p95.push_back(1.0); }   // no paired read: placeholder table
        Bt.read_mate_pos = S.read_mate_pos.data(); Bt.read_mate_len = S.read_mate_len.data(); Bt.read_lib = S.read_lib.data();
        Bt.n_libs = int(lib_p95.size()); Bt.lib_off = lib_off.data(); Bt.lib_prob = lib_prob.data(); Bt.lib_p95 = lib_p95.data();
    }

    dd_sizes sz;
    if (dd_batch_sizes(&Bt, &sz) != DD_SUCCESS) throw std::string(dd_last_error());
    B.pair_off.resize(size_t(W) + 1); B.hpos_off.resize(size_t(W) + 1); B.vc_off.resize(size_t(W) + 1);
    dd_batch_offsets(&Bt, B.pair_off.data(), B.hpos_off.data(), B.vc_off.data());

    const size_t np = size_t(sz.n_pairs) + 1;
    dd_result Rz;
    memset(&Rz, 0, sizeof(Rz));
    Rz.ll = B.ll.reserve(np); Rz.llOn = B.llOn.reserve(np); Rz.llOff = B.llOff.reserve(np); Rz.mLogBQ = B.mLogBQ.reserve(np);
    Rz.offHap = B.offHap.reserve(np); Rz.offHapHMQ = B.offHapHMQ.reserve(np); Rz.status = B.status.reserve(np);
    Rz.numIndels = B.numIndels.reserve(np); Rz.numMismatch = B.numMismatch.reserve(np); Rz.nBQT = B.nBQT.reserve(np);
    Rz.nmmBQT = B.nmmBQT.reserve(np); Rz.nMMLeft = B.nMMLeft.reserve(np); Rz.nMMRight = B.nMMRight.reserve(np);
    Rz.firstBase = B.firstBase.reserve(np); Rz.lastBase = B.lastBase.reserve(np);
    Rz.onHap = NULL;                                    // derived from offHapHMQ on the host (WindowLikelihoods::onHap)
    Rz.var_fcov = B.fcov.reserve(size_t(sz.var_cov_len) + 1); Rz.var_covered = B.vcov.reserve(size_t(sz.var_cov_len) + 1);
    if (B.has_hpos) Rz.hpos = B.hpos.reserve(size_t(sz.hpos_len) + 1);
    dd_params P = to_abi(params);
    if (faster) P.mapUnmappedReads = 0;                 // ObservationModelS has no insert-size prior
    const std::chrono::steady_clock::time_point t_packed = std::chrono::steady_clock::now();
    if (getenv("DD_TIMING"))
        fprintf(stderr, "pack_timing: windows=%d pass1=%.2fms pass2+alloc=%.2fms\n", W, std::chrono::duration<double, std::milli>(t_pass1 - t_start).count(),
                std::chrono::duration<double, std::milli>(t_packed - t_pass1).count());
    if (sz.n_pairs > 0) {
        const int rc = faster ? dd_compute_likelihoods_faster(&P, &Bt, &Rz, device_) : dd_compute_likelihoods(&P, &Bt, &Rz, device_);
        if (rc != DD_SUCCESS) throw std::string(faster ? "dd_compute_likelihoods_faster: " : "dd_compute_likelihoods: ") + dd_last_error();
    }

    const std::chrono::steady_clock::time_point t_device = std::chrono::steady_clock::now();
    // ---- per window: the error the reference's loop would have stopped at (first pair in liks[h][r] order that is not OK),
    // the lazy view, and — for jobs that bring the reference's containers — every record ----
    const std::shared_ptr<const BatchBlock> cblk = blk;
    auto finish_window = [&](int w) {
        WindowJob &J = jobs[w];
        const size_t H = J.haps->size(), Rn = J.reads->size();
        const bool eager = J.liks != NULL;
        if (J.onHap) *J.onHap = std::vector<int>(Rn, 0);                                                        // DInDel.cpp:1710
        if (eager) *J.liks = std::vector<std::vector<MLAlignment> >(H, std::vector<MLAlignment>(Rn));           // DInDel.cpp:1714
        for (size_t h = 0; h < H && J.error.empty(); h++) {
            for (size_t r = 0; r < Rn; r++) {
                const int64_t p = B.pair_off[w] + int64_t(h) * int64_t(Rn) + int64_t(r);
                const int st = B.status[size_t(p)];
                if (st == DD_PAIR_HAPSIZE) { J.error = "hapSize error."; break; }   // ObservationModelFB.cpp:47, Faster.cpp:47
                if (st == DD_PAIR_UNSUPPORTED) {      // this window only; the caller skips it like a window that threw (DInDel.cpp:1369-1374)
                    J.error = "window outside the GPU kernel limits (haplotype > 766 bp, read > 1024 bp or an empty sequence)";
                    break;
                }
                if (faster && st != DD_PAIR_OK) { J.error = "HapHash string too short"; break; }   // Haplotype.hpp:341
                if (eager) fill_record(B, w, h, r, (*J.haps)[h], (*J.reads)[r], (*J.liks)[h][r]);
                if (J.onHap && !B.offHapHMQ[size_t(p)]) (*J.onHap)[r] = 1;                // DInDel.cpp:1720
                if (faster) continue;                                                     // computeLikelihoodsFaster has no ll checks
                if (st == DD_PAIR_LLPOS) {                                                // DInDel.cpp:1722-1731
                    if (throwOnPositive_) { J.error = "Likelihood>0"; break; }
                    std::cout << "hidx: " << h << " r: " << r << std::endl;
                    std::cerr << "Likelihood>0" << std::endl;
                    exit(1);
                }
                if (st == DD_PAIR_NAN) {                                                  // DInDel.cpp:1732-1735
                    std::cout << "NAN/Inf error" << std::endl;
                    J.error = "Nan detected";
                    break;
                }
            }
        }
        if (J.error.empty()) {
            J.result.blk_ = cblk; J.result.w_ = w; J.result.haps_ = J.haps; J.result.reads_ = J.reads;
            J.result.leftPos_ = J.leftPos; J.result.eng_ = this;
        }
    };
    parallel_windows(W, pick_threads(hostThreads_, W), 4, finish_window);
    const std::chrono::steady_clock::time_point t_end = std::chrono::steady_clock::now();
    lastPackSeconds = std::chrono::duration<double>(t_packed - t_start).count();
    lastDeviceSeconds = std::chrono::duration<double>(t_device - t_packed).count();
    lastUnpackSeconds = std::chrono::duration<double>(t_end - t_device).count();
}

} // namespace dindel
