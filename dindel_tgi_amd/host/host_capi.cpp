// host_capi.cpp — small extern "C" hooks so the pytest suite can drive the C++ host adapter
// (LikelihoodEngine) through ctypes.  Not part of the drop-in boundary.
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <sstream>
#include <string>
#include "compute_likelihoods.hpp"
#include "genotype.hpp"
#include "cigar.hpp"

using namespace dindel;

static void json_ml(std::ostringstream &os, const MLAlignment &ml)
{
    os.precision(17);
    os << "{\"ll\":" << ml.ll << ",\"llOn\":" << ml.llOn << ",\"llOff\":" << ml.llOff << ",\"mLogBQ\":" << ml.mLogBQ
       << ",\"offHap\":" << int(ml.offHap) << ",\"offHapHMQ\":" << int(ml.offHapHMQ) << ",\"numIndels\":" << ml.numIndels
       << ",\"numMismatch\":" << ml.numMismatch << ",\"nBQT\":" << ml.nBQT << ",\"nmmBQT\":" << ml.nmmBQT
       << ",\"nMMLeft\":" << ml.nMMLeft << ",\"nMMRight\":" << ml.nMMRight << ",\"firstBase\":" << ml.firstBase
       << ",\"lastBase\":" << ml.lastBase << ",\"align\":\"" << ml.align << "\",\"hpos\":[";
    for (size_t i = 0; i < ml.hpos.size(); i++) os << (i ? "," : "") << ml.hpos[i];
    os << "],\"indels\":[";
    bool first = true;
    for (std::map<int, AlignedVariant>::const_iterator it = ml.indels.begin(); it != ml.indels.end(); ++it, first = false)
        os << (first ? "" : ",") << "[" << it->first << ",\"" << it->second.getString() << "\"," << it->second.getStartHap() << ","
           << it->second.getEndHap() << "," << it->second.getStartRead() << "," << it->second.getEndRead() << "]";
    os << "],\"snps\":[";
    first = true;
    for (std::map<int, AlignedVariant>::const_iterator it = ml.snps.begin(); it != ml.snps.end(); ++it, first = false)
        os << (first ? "" : ",") << "[" << it->first << ",\"" << it->second.getString() << "\"]";
    os << "],\"hapIndelCovered\":[";
    first = true;
    for (std::map<int, bool>::const_iterator it = ml.hapIndelCovered.begin(); it != ml.hapIndelCovered.end(); ++it, first = false)
        os << (first ? "" : ",") << "[" << it->first << "," << int(it->second) << "]";
    os << "],\"hapIndelFilterCovered\":[";
    first = true;
    for (std::map<int, bool>::const_iterator it = ml.hapIndelFilterCovered.begin(); it != ml.hapIndelFilterCovered.end(); ++it, first = false)
        os << (first ? "" : ",") << "[" << it->first << "," << int(it->second) << "]";
    os << "]}";
}

static ObservationModelParameters make_params(const double *pd, const int *pi)
{
    ObservationModelParameters p;
    p.pError = pd[0]; p.pMut = pd[1]; p.pFirstgLO = pd[2]; p.mapQualThreshold = pd[3]; p.checkBaseQualThreshold = pd[4];
    p.capMapQualFast = pd[5];
    p.maxLengthDel = p.maxLengthIndel = pi[0]; p.padCover = pi[1]; p.bMid = pi[2];
    return p;
}

static int emit(const std::string &s, char *out, int cap)
{
    if (int(s.size()) + 1 > cap) return -int(s.size()) - 1;
    memcpy(out, s.c_str(), s.size() + 1);
    return int(s.size());
}

extern "C" {

// CPU only: reportVariants rebuilt from an hpos vector (no device involved)
int ddh_rebuild_json(const char *hap, const char *read, const double *qual, const short *hpos, int L,
                     const double *pd, const int *pi, const int *hap_indels /* n x {key,startRead,endRead} */, int n_hap_indels,
                     char *out, int cap)
{
    const bool faster = (n_hap_indels & 0x10000) != 0;       // bit 16: ObservationModelS::reportVariants instead
    n_hap_indels &= 0xffff;
    try {
        Haplotype H((std::string(hap)));
        for (int i = 0; i < n_hap_indels; i++)
            H.indels[hap_indels[3 * i]] = AlignedVariant("-A", hap_indels[3 * i], hap_indels[3 * i], hap_indels[3 * i + 1], hap_indels[3 * i + 2]);
        Read R;
        R.seq.seq = read;
        R.qual.assign(qual, qual + L);
        MLAlignment ml;
        if (faster) LikelihoodEngine::rebuildAlignmentFaster(H, R, hpos, make_params(pd, pi), ml);
        else LikelihoodEngine::rebuildAlignment(H, R, hpos, make_params(pd, pi), ml);
        std::ostringstream os;
        json_ml(os, ml);
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}

// GPU: one window through LikelihoodEngine::computeLikelihoods.  haps / reads are '\n'-joined strings,
// quals one double per read base (concatenated), mapq / start / unmapped per read.
// mate: NULL, or 4 ints per read {bit0 paired | bit1 mateUnmapped | bit2 mateReverse | bit3 sameTid, matePos, mateLen, library index};
// libraries as insert-size histograms (lib_counts concatenated, lib_sizes[n_libs]) -> dindel::Library; switches mapUnmappedReads on
static int compute_window_json(bool faster, const char *haps_nl, const char *reads_nl, const double *quals, const double *mapq,
                               const double *pos_first, const int *unmapped, unsigned leftPos, const double *pd, const int *pi,
                               int device, char *out, int cap, const int *mate = NULL, const double *lib_counts = NULL,
                               const int *lib_sizes = NULL, int n_libs = 0)
{
    try {
        std::vector<Haplotype> haps;
        std::vector<Read> reads;
        std::istringstream hs(haps_nl), rs(reads_nl);
        std::string line;
        while (std::getline(hs, line)) if (!line.empty()) haps.push_back(Haplotype(line));
        size_t qoff = 0, ri = 0;
        while (std::getline(rs, line)) {
            if (line.empty()) continue;
            Read R;
            R.seq.seq = line;
            R.qual.assign(quals + qoff, quals + qoff + line.size());
            qoff += line.size();
            R.mapQual = mapq[ri];
            R.posStat.first = pos_first[ri];
            R.unmapped = unmapped[ri] != 0;
            reads.push_back(R);
            ri++;
        }
        std::vector<Library> libs;
        if (mate) {
            size_t o = 0;
            for (int i = 0; i < n_libs; i++) { libs.push_back(Library(std::vector<double>(lib_counts + o, lib_counts + o + lib_sizes[i]))); o += size_t(lib_sizes[i]); }
            for (size_t r = 0; r < reads.size(); r++) {
                const int *m = mate + 4 * r;
                reads[r].paired = (m[0] & 1) != 0; reads[r].mateUnmapped = (m[0] & 2) != 0; reads[r].mateReverse = (m[0] & 4) != 0;
                reads[r].mateSameTid = (m[0] & 8) != 0; reads[r].matePos = m[1]; reads[r].mateLen = m[2];
                reads[r].library = (m[3] >= 0 && m[3] < n_libs) ? &libs[size_t(m[3])] : NULL;
            }
        }
        ObservationModelParameters prm = make_params(pd, pi);
        prm.mapUnmappedReads = mate != NULL;
        LikelihoodEngine eng(prm, device);
        eng.setThrowOnPositiveLikelihood(true);
        std::vector<std::vector<MLAlignment> > liks;
        std::vector<int> onHap;
        if (faster) eng.computeLikelihoodsFaster(haps, reads, liks, leftPos, leftPos + 1, onHap);
        else eng.computeLikelihoods(haps, reads, liks, leftPos, leftPos + 1, onHap);
        std::ostringstream os;
        os << "{\"onHap\":[";
        for (size_t i = 0; i < onHap.size(); i++) os << (i ? "," : "") << onHap[i];
        os << "],\"liks\":[";
        for (size_t h = 0; h < liks.size(); h++) {
            os << (h ? "," : "") << "[";
            for (size_t r = 0; r < liks[h].size(); r++) { if (r) os << ","; json_ml(os, liks[h][r]); }
            os << "]";
        }
        os << "]}";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}

int ddh_compute_window_json(const char *haps_nl, const char *reads_nl, const double *quals, const double *mapq,
                            const double *pos_first, const int *unmapped, unsigned leftPos, const double *pd, const int *pi,
                            int device, char *out, int cap)
{
    return compute_window_json(false, haps_nl, reads_nl, quals, mapq, pos_first, unmapped, leftPos, pd, pi, device, out, cap);
}

// computeLikelihoods with the insert-size prior (mapUnmappedReads; the reference's --libFile run)
int ddh_compute_window_mates_json(const char *haps_nl, const char *reads_nl, const double *quals, const double *mapq,
                                  const double *pos_first, const int *unmapped, unsigned leftPos, const double *pd, const int *pi,
                                  int device, const int *mate, const double *lib_counts, const int *lib_sizes, int n_libs, char *out, int cap)
{
    return compute_window_json(false, haps_nl, reads_nl, quals, mapq, pos_first, unmapped, leftPos, pd, pi, device, out, cap,
                               mate, lib_counts, lib_sizes, n_libs);
}

// the same through LikelihoodEngine::computeLikelihoodsFaster (--faster model)
int ddh_compute_window_faster_json(const char *haps_nl, const char *reads_nl, const double *quals, const double *mapq,
                                   const double *pos_first, const int *unmapped, unsigned leftPos, const double *pd, const int *pi,
                                   int device, char *out, int cap)
{
    return compute_window_json(true, haps_nl, reads_nl, quals, mapq, pos_first, unmapped, leftPos, pd, pi, device, out, cap);
}

// AlignedVariant mirror next to the reference's class (tests/test_ref_bits.py): returns isCovered, writes type / length / seq
int ddh_aligned_variant(const char *str, int startHap, int endHap, int startRead, int endRead, int pad, int firstBase, int lastBase,
                        int *type, int *length, char *seq, int cap)
{
    *type = -9; *length = -9; seq[0] = 0;
    try {
        AlignedVariant av(std::string(str), startHap, endHap, startRead, endRead);
        *type = av.getType() == AlignedVariant::INS ? 0 : av.getType() == AlignedVariant::DEL ? 1 : av.getType() == AlignedVariant::SNP ? 2 : 3;
        *length = av.size();
        strncpy(seq, av.getSeq().c_str(), size_t(cap - 1));
        seq[cap - 1] = 0;
        return av.isCovered(pad, firstBase, lastBase) ? 1 : 0;
    } catch (std::string &) {
        return -1;
    }
}

double ddh_add_logs(double l1, double l2) { return addLogs(l1, l2); }

// N4: getCIGAR.  out_ops receives (op, len) pairs; returns the number of operations, or -(1+k) for the k-th throw string of
// {"Haplotype has not been aligned!", "Read is not properly aligned!", "Error(1)!", "Error(2)!", "Error(3)!", "Error(4)!", "How is this possible? (1)"}
int ddh_get_cigar(const int *hap_ref_pos, int hap_size, const short *hpos, int read_size, int ref_seq_start, int *out_ops, int cap, int *ref_pos)
{
    static const char *msgs[] = {"Haplotype has not been aligned!", "Read is not properly aligned!", "Error(1)!", "Error(2)!", "Error(3)!",
                                 "Error(4)!", "How is this possible? (1)"};
    try {
        MLAlignment ml;
        ml.hpos.assign(hpos, hpos + read_size);
        CIGAR c = getCIGAR(std::vector<int>(hap_ref_pos, hap_ref_pos + hap_size), size_t(hap_size), ml, size_t(read_size), ref_seq_start);
        if (int(c.size()) * 2 > cap) return -100;
        for (size_t i = 0; i < c.size(); i++) { out_ops[2 * i] = c[i].first; out_ops[2 * i + 1] = c[i].second; }
        *ref_pos = c.refPos;
        return int(c.size());
    } catch (std::string &e) {
        for (int k = 0; k < 7; k++) if (e == msgs[k]) return -(1 + k);
        return -99;
    }
}

// N1 host step: returns {max_indel_pair, max_noindel_pair, max_ll_indel, max_ll_noindel, qual} or -1 on the
// reference's "Could not find indel allele" throw
int ddh_pair_posteriors(int nh, const double *pair_sum, const double *prior, const int *filtered, const int *ncand,
                        double *posterior_out, int *pairs_out, double *vals_out)
{
    try {
        std::vector<double> ps(pair_sum, pair_sum + nh * nh), pr(prior, prior + nh * nh);
        std::vector<int> f(filtered, filtered + nh), nc(ncand, ncand + nh);
        PairPosteriorResult R = diploidPairPosteriors(nh, ps, pr, f, nc);
        for (int i = 0; i < nh * nh; i++) posterior_out[i] = R.pairs_posterior[i];
        pairs_out[0] = R.max_indel_pair[0]; pairs_out[1] = R.max_indel_pair[1];
        pairs_out[2] = R.max_noindel_pair[0]; pairs_out[3] = R.max_noindel_pair[1];
        vals_out[0] = R.max_ll_indel; vals_out[1] = R.max_ll_noindel; vals_out[2] = R.qual;
        return 0;
    } catch (std::string &e) {
        return -1;
    }
}

// GPU: one window with annotated haplotype indels through computeLikelihoods + filterHaplotypes.
// hap_vars: per haplotype n then n x {key, kind(1 DEL,2 INS), leftFlankRead, rightFlankRead}; reads flags: bit0 unmapped, bit1 reverse, bit2 mate reverse
int ddh_filter_window_json(const char *haps_nl, const int *hap_vars, const char *reads_nl, const double *quals, const double *mapq,
                           const double *pos_first, const int *rflags, unsigned leftPos, const double *pd, const int *pi, int maxMismatch,
                           int doFilter, int device, char *out, int cap)
{
    try {
        std::vector<Haplotype> haps;
        std::vector<Read> reads;
        std::istringstream hs(haps_nl), rs(reads_nl);
        std::string line;
        const int *hv = hap_vars;
        while (std::getline(hs, line)) {
            if (line.empty()) continue;
            Haplotype H(line);
            const int n = *hv++;
            for (int i = 0; i < n; i++, hv += 4) {
                AlignedVariant av(hv[1] == 1 ? "-A" : "+A", hv[0], hv[0], hv[2], hv[3]);
                av.setFlanking(hv[0], hv[0], hv[2], hv[3]);
                H.indels[hv[0]] = av;
            }
            haps.push_back(H);
        }
        size_t qoff = 0, ri = 0;
        while (std::getline(rs, line)) {
            if (line.empty()) continue;
            Read R;
            R.seq.seq = line;
            R.qual.assign(quals + qoff, quals + qoff + line.size());
            qoff += line.size();
            R.mapQual = mapq[ri]; R.posStat.first = pos_first[ri];
            R.unmapped = (rflags[ri] & 1) != 0; R.reverse = (rflags[ri] & 2) != 0; R.mateReverse = (rflags[ri] & 4) != 0;
            reads.push_back(R);
            ri++;
        }
        ObservationModelParameters P = make_params(pd, pi);
        P.maxMismatch = maxMismatch;
        LikelihoodEngine eng(P, device);
        std::vector<std::vector<MLAlignment> > liks;
        std::vector<int> onHap, filtered;
        const bool faster = (doFilter & 2) != 0;          // bit 1: use the --faster model for the likelihoods
        if (faster) eng.computeLikelihoodsFaster(haps, reads, liks, leftPos, leftPos + 1, onHap);
        else eng.computeLikelihoods(haps, reads, liks, leftPos, leftPos + 1, onHap);
        std::map<VariantKey, VariantCoverage> cov;
        filterHaplotypes(haps, reads, liks, filtered, cov, (doFilter & 1) != 0);
        std::ostringstream os;
        os << "{\"filtered\":[";
        for (size_t i = 0; i < filtered.size(); i++) os << (i ? "," : "") << filtered[i];
        os << "],\"coverage\":[";
        bool first = true;
        for (std::map<VariantKey, VariantCoverage>::const_iterator it = cov.begin(); it != cov.end(); ++it, first = false)
            os << (first ? "" : ",") << "[" << it->first.first << ",\"" << it->first.second << "\"," << it->second.nf << "," << it->second.nr << "]";
        os << "],\"flags\":[";
        for (size_t h = 0; h < liks.size(); h++) {
            os << (h ? "," : "") << "[";
            for (size_t r = 0; r < liks[h].size(); r++) {
                os << (r ? "," : "") << "[";
                bool f2 = true;
                for (std::map<int, bool>::const_iterator it = liks[h][r].hapIndelFilterCovered.begin(); it != liks[h][r].hapIndelFilterCovered.end(); ++it, f2 = false)
                    os << (f2 ? "" : ",") << int(it->second);
                os << "]";
            }
            os << "]";
        }
        os << "]}";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}



// ---- batch hooks: LikelihoodEngine::computeLikelihoodsBatch over several windows ----
namespace {
struct HookWindows {
    std::vector<std::vector<Haplotype> > haps;
    std::vector<std::vector<Read> > reads;
    std::vector<uint32_t> left;
};

// windows as flat arrays: n_haps[W], n_reads[W], '\n'-joined haplotypes and reads (an empty line = an empty sequence),
// one quality per read base, mapq / start / unmapped per read, leftPos per window
void parse_windows(int W, const int *n_haps, const int *n_reads, const char *haps_nl, const char *reads_nl, const double *quals,
                   const double *mapq, const double *pos_first, const int *unmapped, const unsigned *leftPos, HookWindows &out)
{
    out.haps.resize(size_t(W)); out.reads.resize(size_t(W)); out.left.assign(leftPos, leftPos + W);
    const char *hp = haps_nl, *rp = reads_nl;
    size_t qoff = 0, ri = 0;
    for (int w = 0; w < W; w++) {
        for (int h = 0; h < n_haps[w]; h++) {
            const char *e = strchr(hp, '\n');
            const size_t n = e ? size_t(e - hp) : strlen(hp);
            out.haps[size_t(w)].push_back(Haplotype(std::string(hp, n)));
            hp += n + (e ? 1 : 0);
        }
        for (int r = 0; r < n_reads[w]; r++, ri++) {
            const char *e = strchr(rp, '\n');
            const size_t n = e ? size_t(e - rp) : strlen(rp);
            Read R;
            R.seq.seq = std::string(rp, n);
            rp += n + (e ? 1 : 0);
            R.qual.assign(quals + qoff, quals + qoff + n);
            qoff += n;
            R.mapQual = mapq[ri]; R.posStat.first = pos_first[ri]; R.unmapped = unmapped[ri] != 0;
            out.reads[size_t(w)].push_back(R);
        }
    }
}
} // namespace

// Runs the windows twice through computeLikelihoodsBatch — once eager (the reference's containers), once lazy (views; with
// flags bit1 also without alignments kept) — and reports, per window, the error string and the lazy view's scalars, plus the
// number of pairs whose lazy view (every scalar accessor and the full record from get()) differs from the eager record.
// flags: bit0 = --faster model, bit1 = setKeepAlignments(false) for the lazy run.
int ddh_batch_json(int W, const int *n_haps, const int *n_reads, const char *haps_nl, const char *reads_nl, const double *quals,
                   const double *mapq, const double *pos_first, const int *unmapped, const unsigned *leftPos, const double *pd,
                   const int *pi, int flags, int device, char *out, int cap)
{
    try {
        HookWindows Wn;
        parse_windows(W, n_haps, n_reads, haps_nl, reads_nl, quals, mapq, pos_first, unmapped, leftPos, Wn);
        const bool faster = (flags & 1) != 0;
        LikelihoodEngine eng(make_params(pd, pi), device);
        eng.setThrowOnPositiveLikelihood(true);
        std::vector<std::vector<std::vector<MLAlignment> > > liks(static_cast<size_t>(W));
        std::vector<std::vector<int> > onHap(static_cast<size_t>(W));
        std::vector<WindowJob> eager(static_cast<size_t>(W)), lazy(static_cast<size_t>(W));
        for (int w = 0; w < W; w++) {
            eager[size_t(w)].haps = lazy[size_t(w)].haps = &Wn.haps[size_t(w)];
            eager[size_t(w)].reads = lazy[size_t(w)].reads = &Wn.reads[size_t(w)];
            eager[size_t(w)].leftPos = lazy[size_t(w)].leftPos = Wn.left[size_t(w)];
            eager[size_t(w)].rightPos = lazy[size_t(w)].rightPos = Wn.left[size_t(w)] + 1;
            eager[size_t(w)].liks = &liks[size_t(w)]; eager[size_t(w)].onHap = &onHap[size_t(w)];
        }
        if (faster) eng.computeLikelihoodsFasterBatch(eager); else eng.computeLikelihoodsBatch(eager);
        if (flags & 2) eng.setKeepAlignments(false);
        if (faster) eng.computeLikelihoodsFasterBatch(lazy); else eng.computeLikelihoodsBatch(lazy);
        long mismatch = 0;
        std::ostringstream os;
        os.precision(17);
        os << "{\"windows\":[";
        for (int w = 0; w < W; w++) {
            const WindowJob &E = eager[size_t(w)], &Lz = lazy[size_t(w)];
            os << (w ? "," : "") << "{\"error\":\"" << Lz.error << "\"";
            if (E.error != Lz.error) mismatch++;
            if (Lz.error.empty()) {
                const WindowLikelihoods &V = Lz.result;
                if (!V.valid() || V.numHaps() != Wn.haps[size_t(w)].size() || V.numReads() != Wn.reads[size_t(w)].size()) { mismatch++; os << "}"; continue; }
                os << ",\"ll\":[";
                for (size_t h = 0; h < V.numHaps(); h++)
                    for (size_t r = 0; r < V.numReads(); r++) {
                        os << ((h || r) ? "," : "") << V.ll(h, r);
                        const MLAlignment &m = liks[size_t(w)][h][r];
                        if (V.ll(h, r) != m.ll || V.llOn(h, r) != m.llOn || V.llOff(h, r) != m.llOff || V.offHap(h, r) != m.offHap ||
                            V.offHapHMQ(h, r) != m.offHapHMQ || V.firstBase(h, r) != m.firstBase || V.lastBase(h, r) != m.lastBase)
                            mismatch++;
                        if (!faster && (V.mLogBQ(h, r) != m.mLogBQ || V.numIndels(h, r) != m.numIndels || V.numMismatch(h, r) != m.numMismatch ||
                                        V.nBQT(h, r) != m.nBQT || V.nmmBQT(h, r) != m.nmmBQT || V.nMMLeft(h, r) != m.nMMLeft || V.nMMRight(h, r) != m.nMMRight))
                            mismatch++;
                        if (!faster && V.numIndels(h, r) != int(m.indels.size())) mismatch++;      // what DInDel.cpp:3529 reads
                        if (V.indelCount(h, r) != int(m.indels.size())) mismatch++;                // the view's shortcut to the same number
                        std::ostringstream a, b;
                        json_ml(a, m);
                        json_ml(b, V.get(h, r));
                        if (a.str() != b.str()) mismatch++;
                    }
                os << "],\"onHap\":[";
                for (size_t r = 0; r < V.numReads(); r++) {
                    os << (r ? "," : "") << V.onHap(r);
                    if (V.onHap(r) != onHap[size_t(w)][r]) mismatch++;
                }
                os << "]";
            }
            os << "}";
        }
        os << "],\"mismatch\":" << mismatch << "}";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}

// bench.py's end-to-end leg: W synthetic windows of the BASELINE shape (H haplotypes of ~HL bp = a random reference plus 1-3-bp
// indel variants, R reads of L bp sampled from them with 0.1 % substitutions, Q30, mapping quality Phred 40) as the
// reference's own objects (vector<Haplotype>, vector<Read>), through LikelihoodEngine::computeLikelihoodsBatch.
// mode: bit0 = eager records (every MLAlignment rebuilt), bit1 = without alignments (lazy only), bit2 = --faster model.
// out[0..5] = best wall seconds of `reps` calls after one warm-up call, its pack / device / finish split, the ll of window 0
// pair (0,0), the number of windows with an error.
int ddh_bench_batch(int W, int H, int R, int L, int HL, unsigned long long seed, int mode, int device, int reps, double *out)
{
    try {
        uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ull;
        auto rnd = [&s]() -> uint64_t { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };   // xorshift64
        const char *acgt = "ACGT";
        std::vector<std::vector<Haplotype> > haps(static_cast<size_t>(W));
        std::vector<std::vector<Read> > reads(static_cast<size_t>(W));
        for (int w = 0; w < W; w++) {
            std::string ref(size_t(HL), 'A');
            for (int i = 0; i < HL; i++) ref[size_t(i)] = acgt[rnd() & 3];
            haps[size_t(w)].push_back(Haplotype(ref));
            for (int h = 1; h < H; h++) {
                const size_t pos = size_t(HL / 2 - 10 + int(rnd() % 20)), ln = 1 + size_t(rnd() % 3);
                haps[size_t(w)].push_back(Haplotype((rnd() & 1) ? ref.substr(0, pos) + ref.substr(pos + ln)
                                                                : ref.substr(0, pos) + std::string(ln, acgt[rnd() & 3]) + ref.substr(pos)));
            }
            reads[size_t(w)].resize(size_t(R));
            for (int r = 0; r < R; r++) {
                const std::string &src = haps[size_t(w)][rnd() % uint64_t(H)].seq;
                const int off = int(rnd() % src.size()) - L / 2;
                Read &rd = reads[size_t(w)][size_t(r)];
                rd.seq.seq.resize(size_t(L));
                for (int i = 0; i < L; i++) {
                    const int j = off + i;
                    rd.seq.seq[size_t(i)] = (j >= 0 && j < int(src.size()) && (rnd() % 1000)) ? src[size_t(j)] : acgt[rnd() & 3];
                }
                rd.qual.assign(size_t(L), 0.999);
                rd.mapQual = 0.9999;
                rd.posStat.first = 1000.0 + off;
            }
        }
        ObservationModelParameters P;
        P.setCLIDefaultValues();
        LikelihoodEngine eng(P, device);
        const bool eager = (mode & 1) != 0, faster = (mode & 4) != 0;
        if (mode & 2) eng.setKeepAlignments(false);
        std::vector<std::vector<std::vector<MLAlignment> > > liks(eager ? size_t(W) : 0);
        std::vector<std::vector<int> > onHap(eager ? size_t(W) : 0);
        std::vector<WindowJob> jobs(static_cast<size_t>(W));
        double best = 1e300;
        for (int rep = 0; rep <= reps; rep++) {
            for (int w = 0; w < W; w++) {
                WindowJob &J = jobs[size_t(w)];
                J.haps = &haps[size_t(w)]; J.reads = &reads[size_t(w)]; J.leftPos = 1000; J.rightPos = 1000 + uint32_t(HL);
                if (eager) { J.liks = &liks[size_t(w)]; J.onHap = &onHap[size_t(w)]; }
            }
            const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            if (faster) eng.computeLikelihoodsFasterBatch(jobs); else eng.computeLikelihoodsBatch(jobs);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0 && dt < best) {
                best = dt;
                out[0] = dt; out[1] = eng.lastPackSeconds; out[2] = eng.lastDeviceSeconds; out[3] = eng.lastUnpackSeconds;
            }
        }
        int nerr = 0;
        for (int w = 0; w < W; w++) nerr += !jobs[size_t(w)].error.empty();
        out[4] = (W > 0 && jobs[0].error.empty()) ? jobs[0].result.ll(0, 0) : 0.0;
        out[5] = nerr;
        return 0;
    } catch (std::string &e) {
        fprintf(stderr, "ddh_bench_batch: %s\n", e.c_str());
        return -1;
    }
}

} // extern "C"

// ---- N3: the .glf.txt output surface (glf_output.hpp) ----
#include <fstream>
#include "glf_output.hpp"

extern "C" {

// how the table prints a double: default ostream formatting (6 significant digits), as OutputData::Line::set does
// formatG6 (glf_output.hpp) against snprintf("%.6g") on `n` values: log-uniform magnitudes over 1e-7 .. 1e17 with both signs, values
// sitting on and one ulp either side of six-digit rounding boundaries (d.ddddd5 x 10^k), powers of ten, and small integers.  Returns
// the number of values whose text differs (0 is the test) and writes the first difference to `out`.
int ddh_format_g6_check(unsigned long long seed, int n, char *out, int cap)
{
    unsigned long long st = seed ? seed : 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    int bad = 0;
    std::string firstBad;
    auto one = [&](double x) {
        char a[64], b[64];
        formatG6(x, a);
        snprintf(b, sizeof(b), "%.6g", x);
        if (strcmp(a, b) != 0) { if (!bad) { char m[200]; snprintf(m, sizeof(m), "%.17g: %s vs %s", x, a, b); firstBad = m; } bad++; }
    };
    for (int i = 0; i < n; i++) {
        const double u = double(rnd() >> 11) / 9007199254740992.0, v = double(rnd() >> 11) / 9007199254740992.0;
        const double mag = pow(10.0, -7.0 + 24.0 * u) * (1.0 + v);
        one(mag); one(-mag);
        // a six-digit boundary: (d + 0.5) x 10^k with d a 6-digit integer, and its neighbours
        const long long d = 100000 + (long long)(rnd() % 900000);
        const int k = int(rnd() % 22) - 11;
        const double tie = (double(d) + 0.5) * pow(10.0, double(k));
        one(tie); one(nextafter(tie, 0.0)); one(nextafter(tie, 1e300)); one(-tie);
        one(double(d) * pow(10.0, double(k)));
        one(double((long long)(rnd() % 2000000)) - 1000000.0);
    }
    const double odd[] = {0.0, -0.0, 1.0, 10.0, 100000.0, 999999.0, 999999.5, 999999.4999999999, 1e6, 1e-4, 9.9999949999e-5, 9.9999950001e-5, 1e-5, 1e15, 123456.5, 0.1, 1.0 / 3.0,
                          HUGE_VAL, -HUGE_VAL, NAN, 5e-324, 1.7976931348623157e308};
    for (size_t i = 0; i < sizeof(odd) / sizeof(odd[0]); i++) one(odd[i]);
    emit(bad ? firstBad : std::string("ok"), out, cap);
    return bad;
}

int ddh_format_double(double x, char *out, int cap)
{
    std::ostringstream os;
    OutputData od(os);
    od("v");
    OutputData::Line line(od);
    line.set("v", x);
    std::stringstream ref;                          // what the reference's Line::set does (OutputData.hpp:82-84)
    ref << x;
    if (ref.str() != line.get("v")) return emit(std::string("MISMATCH ") + line.get("v") + " vs " + ref.str(), out, cap);
    const long asLong = long(x);
    const unsigned asUnsigned = unsigned(asLong);
    std::stringstream r1, r2, r3, r4;
    r1 << asLong; r2 << asUnsigned; r3 << size_t(asUnsigned); r4 << int(asLong);
    OutputData::Line l2(od);
    if (l2.set("v", asLong).get("v") != r1.str() || l2.set("v", asUnsigned).get("v") != r2.str() || l2.set("v", size_t(asUnsigned)).get("v") != r3.str() ||
        l2.set("v", int(asLong)).get("v") != r4.str())
        return emit("MISMATCH integers", out, cap);
    return emit(line.get("v"), out, cap);
}

// Writes a small .glf.txt through the C++ API: header, a skipped-window line for `thrown`, one dip.map line and one per-position
// "dip" line from the numbers given (vals: qual, msq, genoqual, logZ, mLogBQ sum).  Returns the number of lines written.
int ddh_glf_demo(const char *path, const char *thrown, const double *vals)
{
    try {
        std::ofstream f(path);
        OutputData glf = makeGLFOutputData(f);
        glf.outputLine(glf.headerString());
        glf.output(skippedWindowLine(glf, skippedMessage(thrown), 7, "20", 1000123u, 1000243u));
        DipMapCall c;
        c.index = 8; c.tid = "20"; c.leftPos = 2000000u; c.rightPos = 2000120u; c.candPos = 2000060u; c.realignedPos = 2000058; c.was_candidate = 1;
        c.qual = vals[0]; c.nref_all = "-AC"; c.num_reads = 173; c.msq = vals[1]; c.numf = 11; c.numr = 9; c.vc_f = 12; c.vc_r = 10;
        c.numUnmappedRealigned = 2; c.genotype = "0/1"; c.genoqual = vals[2];
        glf.output(dipMapLine(glf, c));
        DipPositionRow d;
        d.index = 8; d.tid = "20"; d.program = "dip"; d.leftPos = 2000000u; d.rightPos = 2000120u; d.candPos = 2000060u; d.realignedPos = 2000058;
        d.has_variants_in_window = 1; d.logZ = vals[3]; d.nBQT = 15000; d.nmmBQT = 37; d.mLogBQ = vals[4]; d.nMMLeft = 3; d.nMMRight = 1;
        d.nref_all = "-AC"; d.num_reads = 173; d.msq = vals[1]; d.numOffAll = 4; d.num_indel = 21; d.nf = 11; d.nr = 9;
        d.var_coverage_forward = "12"; d.var_coverage_reverse = "10"; d.glf = "0/0:-310.5,0/1:-250.25,1/1:-400"; d.numUnmappedRealigned = 2;
        glf.output(dipPositionLine(glf, d));
        // the same two lines through the text writers the window loop uses (GlfText): lines 5 and 6 of the file
        if (!glf.hasGLFColumns()) return -2;
        std::string text;
        dipMapText(text, c); glf.outputText(text);
        dipPositionText(text, d); glf.outputText(text);
        return glf.lines();
    } catch (std::string &e) {
        return -1;
    }
}

} // extern "C"

// ---- N2: readers and the window's read selection ----
#include "bam_reader.hpp"
#include "get_reads.hpp"
#include "fast_inflate.hpp"
#include "realigned_bam.hpp"
#include "window_io.hpp"

extern "C" {

// every record bam_fetch would hand over for (tid name, [beg, end)), as JSON; with beg < 0 the whole file sequentially
int ddh_bam_fetch_json(const char *path, const char *tid, int beg, int end, char *out, int cap)
{
    try {
        BamFile bam(path);
        std::ostringstream os;
        os << "{\"targets\":[";
        for (size_t i = 0; i < bam.targetNames().size(); i++) os << (i ? "," : "") << "[\"" << bam.targetNames()[i] << "\"," << bam.targetLengths()[i] << "]";
        os << "],\"records\":[";
        bool first = true;
        auto emitRec = [&](const BamRecord &b) -> bool {
            os << (first ? "" : ",") << "{\"qname\":\"" << b.qname << "\",\"tid\":" << b.tid << ",\"pos\":" << b.pos << ",\"flag\":" << b.flag << ",\"mapq\":" << int(b.qual)
               << ",\"mtid\":" << b.mtid << ",\"mpos\":" << b.mpos << ",\"isize\":" << b.isize << ",\"end\":" << b.endPos() << ",\"seq\":\"" << b.seq << "\",\"qual\":[";
            for (size_t i = 0; i < b.qualities.size(); i++) os << (i ? "," : "") << int(b.qualities[i]);
            os << "],\"cigar\":[";
            for (size_t i = 0; i < b.cigar.size(); i++) os << (i ? "," : "") << b.cigar[i];
            const char *lib = bam.getLibrary(b);
            os << "],\"lib\":" << (lib ? std::string("\"") + lib + "\"" : std::string("null")) << "}";
            first = false;
            return true;
        };
        if (beg < 0) { BamRecord b; while (bam.next(b)) emitRec(b); }
        else bam.fetch(bam.getTID(tid), beg, end, emitRec);
        os << "]}";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}

// the names bam_fetch hands over for each of n regions (reg[2k], reg[2k+1]), all fetched on ONE handle, as a JSON list of lists
int ddh_bam_fetch_seq_json(const char *path, const char *tid, const int *reg, int n, char *out, int cap)
{
    try {
        BamFile bam(path);
        const int t = bam.getTID(tid);
        std::ostringstream os;
        os << "[";
        for (int k = 0; k < n; k++) {
            os << (k ? "," : "") << "[";
            bool first = true;
            bam.fetch(t, reg[2 * k], reg[2 * k + 1], [&](const BamRecord &b) -> bool { os << (first ? "" : ",") << "\"" << b.qname << "\""; first = false; return true; });
            os << "]";
        }
        os << "]";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}

// fastInflate (fast_inflate.cpp) on a raw DEFLATE stream: 1 = decoded and equal to `want`, 0 = declined (zlib would take over),
// -1 = decoded something else
int ddh_fast_inflate_check(const unsigned char *comp, int clen, const unsigned char *want, int n)
{
    std::vector<unsigned char> out(size_t(n) + 8, 0xAB);
    if (!fastInflate(comp, size_t(clen), out.data(), size_t(n))) return 0;
    return memcmp(out.data(), want, size_t(n)) == 0 ? 1 : -1;
}

unsigned ddh_fast_crc32(unsigned crc, const unsigned char *buf, int n) { return fastCrc32(crc, buf, size_t(n)); }

// the haplotype fixture as the driver sees it: for the windows asked for, [index, leftPos, rightPos, [[seq, [[kind, key, string, startHap,
// endHap, startRead, endRead, leftFlankHap, rightFlankHap, leftFlankRead, rightFlankRead], ...]], ...]] (null: no such window)
int ddh_fixture_json(const char *path, const int *indices, int n, char *out, int cap)
{
    try {
        HaplotypeFixture fx(path);
        std::ostringstream os;
        os << "[";
        for (int k = 0; k < n; k++) {
            const WindowHaplotypes *w = fx.find(indices[k]);
            os << (k ? "," : "");
            if (!w) { os << "null"; continue; }
            os << "[" << w->index << "," << w->leftPos << "," << w->rightPos << ",[";
            for (size_t h = 0; h < w->haps.size(); h++) {
                os << (h ? "," : "") << "[\"" << w->haps[h].seq << "\",[";
                bool first = true;
                for (int pass = 0; pass < 2; pass++) {
                    const std::map<int, AlignedVariant> &m = pass ? w->haps[h].snps : w->haps[h].indels;
                    for (std::map<int, AlignedVariant>::const_iterator it = m.begin(); it != m.end(); ++it) {
                        const AlignedVariant &v = it->second;
                        os << (first ? "" : ",") << "[\"" << (pass ? "S" : "I") << "\"," << it->first << ",\"" << v.getString() << "\"," << v.getStartHap() << "," << v.getEndHap()
                           << "," << v.getStartRead() << "," << v.getEndRead() << "," << v.getLeftFlankHap() << "," << v.getRightFlankHap() << "," << v.getLeftFlankRead() << ","
                           << v.getRightFlankRead() << "]";
                        first = false;
                    }
                }
                os << "]";
                if (!w->haps[h].refHpos.empty()) {                  // the A record (hap.ml.hpos), when the file carries one
                    os << ",[";
                    for (size_t b = 0; b < w->haps[h].refHpos.size(); b++) os << (b ? "," : "") << w->haps[h].refHpos[b];
                    os << "]";
                }
                os << "]";
            }
            os << "]]";
        }
        os << "]";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    } catch (HaplotypeFixture::Error &e) {
        return emit(std::string("{\"throw\":\"") + e.message + "\"}", out, cap);
    }
}

// writeRealignedBAMFile on the records bam_fetch hands over for (tid, [beg, end)), in that order: read k gets the CIGAR
// (op, len) pairs cig[cigOff[k] .. cigOff[k + 1]) with refPos[k] if onHap[k], else it is copied.  Returns the number of reads, or < 0.
int ddh_write_realigned(const char *inBam, const char *tid, int beg, int end, const char *outBam, const int *onHap, const int *cig, const int *cigOff,
                        const int *refPos, int n, char *err, int cap)
{
    try {
        BamFile bam(inBam);
        LibraryCollection libs;
        std::vector<Read> reads;
        bam.fetch(bam.getTID(tid), beg, end, [&](const BamRecord &b) -> bool {
            reads.push_back(makeRead(b, bam, libs, 0, "single_end"));
            reads.back().record = std::make_shared<const std::vector<uint8_t> >(bam.rawRecord());
            return true;
        });
        if (int(reads.size()) != n) { emit("read count differs", err, cap); return -int(reads.size()) - 1000; }
        std::vector<CIGAR> cigars(reads.size());
        std::vector<int> on(onHap, onHap + n);
        for (int k = 0; k < n; k++) {
            for (int i = cigOff[k]; i < cigOff[k + 1]; i++) cigars[size_t(k)].push_back(CIGAR::CIGOp(cig[2 * i], cig[2 * i + 1]));
            cigars[size_t(k)].refPos = refPos[k];
        }
        writeRealignedBAMFile(outBam, cigars, reads, on, bam);
        return n;
    } catch (std::string &e) {
        emit(e, err, cap);
        return -1;
    }
}

// the window file and the library file as the driver sees them
int ddh_parse_inputs_json(const char *varFile, int oneBased, const char *libFile, char *out, int cap)
{
    try {
        std::ostringstream os;
        os.precision(17);
        os << "{\"windows\":[";
        if (varFile && *varFile) {
            VariantFile vf(varFile);
            bool first = true;
            while (!vf.eof()) {
                AlignedCandidates c = vf.getLineVector(oneBased != 0);
                if (c.variants.empty()) continue;
                os << (first ? "" : ",") << "{\"tid\":\"" << c.tid << "\",\"leftPos\":" << c.leftPos << ",\"rightPos\":" << c.rightPos << ",\"centerPos\":" << c.centerPos << ",\"variants\":[";
                for (size_t i = 0; i < c.variants.size(); i++)
                    os << (i ? "," : "") << "[" << c.variants[i].getStartHap() << ",\"" << c.variants[i].getString() << "\"," << c.variants[i].getEndHap() << "," << c.variants[i].getFreq()
                       << "," << int(c.variants[i].getAddComb()) << "]";
                os << "]}";
                first = false;
            }
        }
        os << "],\"libraries\":{";
        LibraryCollection libs;
        if (libFile && *libFile) libs.addFromFile(libFile);
        bool first = true;
        for (LibraryCollection::const_iterator it = libs.begin(); it != libs.end(); ++it, first = false)
            os << (first ? "" : ",") << "\"" << it->first << "\":[" << it->second.getMaxInsertSize() << "," << it->second.getNinetyFifthPctProb() << "," << it->second.getProb(0) << ","
               << it->second.getProb(it->second.getMaxInsertSize() / 2) << "]";
        os << "},\"maxInsertSize\":" << libs.getMaxInsertSize() << "}";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    }
}

// The window file call by call: what every getLineVector() of the loop `while (!vf.eof())` returned (the candidates, or "skipped"), and the
// string that ended the loop if one was thrown — the format of oracle/ref_bits.cpp:ref_window_lines_json, which runs the reference's own
// VariantFile.hpp over the same file (tests/test_ref_bits.py).
int ddh_window_lines_json(const char *varFile, int oneBased, char *out, int cap)
{
    std::ostringstream os;
    os.precision(17);
    os << "{\"calls\":[";
    std::string thrown;
    bool threw = false;
    try {
        VariantFile vf(varFile);
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
    } catch (std::string &e) { threw = true; thrown = e; }
    os << "]";
    if (threw) os << ",\"throw\":\"" << thrown << "\"";
    os << "}";
    return emit(os.str(), out, cap);
}

// DetInDel::getReads over consecutive windows (win: n x {leftPos, rightPos}); prm: {maxReads, maxReadLength, minReadOverlap,
// mapUnmappedReads}; per window the selected reads in order, or the string thrown
int ddh_get_reads_aux_json(const char *bamPath, const char *libFile, const char *tid, const int *win, int n, const int *prm, double mapQualThreshold,
                           const char *filterReadAux, char *out, int cap);
int ddh_get_reads_json(const char *bamPath, const char *libFile, const char *tid, const int *win, int n, const int *prm, double mapQualThreshold,
                       char *out, int cap)
{
    return ddh_get_reads_aux_json(bamPath, libFile, tid, win, n, prm, mapQualThreshold, "", out, cap);
}

// the same with --filterReadAux
int ddh_get_reads_pools_json(const char *bamPaths, const char *libFile, const char *tid, const int *win, int n, const int *prm, double mapQualThreshold,
                             const char *filterReadAux, int withBuffer, char *out, int cap);
int ddh_get_reads_aux_json(const char *bamPath, const char *libFile, const char *tid, const int *win, int n, const int *prm, double mapQualThreshold,
                           const char *filterReadAux, char *out, int cap)
{
    return ddh_get_reads_pools_json(bamPath, libFile, tid, win, n, prm, mapQualThreshold, filterReadAux, 0, out, cap);
}

// several BAM files as pools of one buffer (bamPaths: one path per line); withBuffer: every window also reports the buffer's order,
// [qname, pool] per buffered alignment
int ddh_get_reads_pools_json(const char *bamPaths, const char *libFile, const char *tid, const int *win, int n, const int *prm, double mapQualThreshold,
                             const char *filterReadAux, int withBuffer, char *out, int cap)
{
    try {
        std::vector<std::unique_ptr<BamFile> > handles;
        std::vector<BamFile *> bams;
        {
            std::istringstream list(bamPaths);
            std::string path;
            while (std::getline(list, path)) if (!path.empty()) { handles.push_back(std::unique_ptr<BamFile>(new BamFile(path))); bams.push_back(handles.back().get()); }
        }
        LibraryCollection libs;
        if (libFile && *libFile) libs.addFromFile(libFile);
        ReadSelectionParameters p;
        p.maxReads = size_t(prm[0]); p.maxReadLength = size_t(prm[1]); p.minReadOverlap = prm[2]; p.mapUnmappedReads = prm[3] != 0; p.mapQualThreshold = mapQualThreshold;
        p.filterReadAux = filterReadAux ? filterReadAux : "";
        ReadFetcher f(bams, libs, p);
        std::ostringstream os;
        os.precision(17);
        os << "[";
        std::vector<std::pair<std::string, int> > buffered;
        for (int w = 0; w < n; w++) {
            std::vector<Read> reads;
            os << (w ? "," : "");
            bool skipped = false;
            try {
                f.getReads(tid, uint32_t(win[2 * w]), uint32_t(win[2 * w + 1]), reads);
                os << "{\"reads\":[";
                for (size_t r = 0; r < reads.size(); r++)
                    os << (r ? "," : "") << "[\"" << reads[r].qname << "\"," << int32_t(reads[r].pos) << "," << reads[r].mapQual << "," << reads[r].matePos << "," << reads[r].mateLen << ","
                       << int(reads[r].isUnmapped()) << ",\"" << reads[r].seq.seq << "\"," << reads[r].posStat.first << (withBuffer ? "," : "") << (withBuffer ? std::to_string(reads[r].poolID) : std::string()) << "]";
                os << "]";
            } catch (std::string &e) {
                os << "{\"throw\":\"" << e << "\"";
                skipped = true;
            }
            if (withBuffer) {
                f.buffered(buffered);
                os << ",\"buffer\":[";
                for (size_t r = 0; r < buffered.size(); r++) os << (r ? "," : "") << "[\"" << buffered[r].first << "\"," << buffered[r].second << "]";
                os << "]";
            }
            os << "}";
            f.windowDone(skipped, uint32_t(win[2 * w]));
        }
        os << "]";
        return emit(os.str(), out, cap);
    } catch (std::string &e) {
        return emit(std::string("{\"throw\":\"") + e + "\"}", out, cap);
    } catch (ReadFetcher::FatalError &e) {                 // where the reference calls exit(): the whole run ends, not one window
        return emit(std::string("{\"fatal\":\"") + e.message + "\",\"exitCode\":" + std::to_string(e.exitCode) + "}", out, cap);
    }
}

} // extern "C"
