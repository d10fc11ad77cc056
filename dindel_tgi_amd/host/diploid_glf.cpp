// diploid_glf.cpp — see diploid_glf.hpp.  Line references are to the reference's DInDel.cpp.
//
// The reference's function keeps its intermediate results in maps and sets keyed by (position, variant) pairs, strings and
// integer sets, built afresh for every window and every variant position.  The numbers it prints depend on three things
// only: WHICH terms enter a sum, in WHICH ORDER they are added (fp64 addition is not associative), and the iteration order
// of its ordered containers.  This file computes the same sums in the same orders on flat tables:
//   * the window's distinct variants are numbered once, in the order std::set<pair<int, AlignedVariant> > walks them
//     (position key, then AlignedVariant::operator<: startHap, string); everything a container was keyed by — coverage
//     counts, candidate look-ups, prior terms, the variant of a haplotype at a position — is an array over those numbers;
//   * read sets are bit sets (the reads of a window are numbered 0 .. nr-1 and every set was walked in ascending order);
//   * a haplotype pair's prior is the sum, in set order, over the sorted union of the two haplotypes' variant lists, whose
//     log terms were taken once per variant;
//   * the genotype table of a variant position is an array over (smaller, larger) variant numbers, walked in the order the
//     reference's map<set<int>, double> would be.
#include "diploid_glf.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace dindel {

namespace {

// ---- the window's variants, numbered --------------------------------------------------------------------------------------------
struct VariantOrder {                                       // std::pair<int, AlignedVariant>::operator< / AlignedVariant::operator<
    static bool lessAV(const AlignedVariant &a, const AlignedVariant &b) { return a.getStartHap() != b.getStartHap() ? a.getStartHap() < b.getStartHap() : a.getString() < b.getString(); }
    static bool sameAV(const AlignedVariant &a, const AlignedVariant &b) { return a.getStartHap() == b.getStartHap() && a.getString() == b.getString(); }
    bool operator()(const std::pair<int, const AlignedVariant *> &a, const std::pair<int, const AlignedVariant *> &b) const
    {
        return a.first != b.first ? a.first < b.first : lessAV(*a.second, *b.second);
    }
};

// what DetInDel::getHaplotypePrior / getPairPrior add for one variant (:1837-1927): log of the candidate's prior if the window
// file lists the variant with one, else log of the default for its kind
struct CandidateTerms {
    const AlignedCandidates &cand; int leftPos; double logSNP, logIndel;
    CandidateTerms(const AlignedCandidates &c, int lp, const DiploidParameters &p) : cand(c), leftPos(lp), logSNP(log(p.priorSNP)), logIndel(log(p.priorIndel)) {}
    const AlignedVariant *find(const AlignedVariant &v) const { return cand.findVariant(v.getStartHap() + leftPos, v.getType(), v.getString()); }
    double haplotypeTerm(const AlignedVariant &v) const     // getHaplotypePrior: every variant, SNPs included, falls back to priorIndel (:1899-1908)
    {
        const AlignedVariant *c = find(v);
        return (c == NULL || c->getFreq() < 0.0) ? logIndel : log(c->getFreq());
    }
    double siteTerm(const AlignedVariant &v) const          // getPairPrior: the fall-back follows the variant's kind; a reference allele adds 0 (:1842-1853)
    {
        const int type = v.getType();
        const double lnf = type == AlignedVariant::SNP ? logSNP : (type == AlignedVariant::DEL || type == AlignedVariant::INS) ? logIndel : 0.0;
        const AlignedVariant *c = find(v);
        return (c == NULL || c->getFreq() < 0.0) ? lnf : log(c->getFreq());
    }
};

typedef std::vector<std::pair<const AlignedVariant *, double> > TermList;

// the sum over std::set<AlignedVariant>{a's, b's} in set order, given each side sorted and free of duplicates
double sumOverUnion(const TermList &a, const TermList &b, double ll)
{
    size_t i = 0, j = 0;
    while (i < a.size() || j < b.size()) {
        if (j == b.size() || (i < a.size() && VariantOrder::lessAV(*a[i].first, *b[j].first))) ll += a[i++].second;
        else if (i == a.size() || VariantOrder::lessAV(*b[j].first, *a[i].first)) ll += b[j++].second;
        else { ll += a[i].second; i++; j++; }
    }
    return ll;
}

bool termLess(const TermList::value_type &x, const TermList::value_type &y) { return VariantOrder::lessAV(*x.first, *y.first); }
bool termSame(const TermList::value_type &x, const TermList::value_type &y) { return VariantOrder::sameAV(*x.first, *y.first); }

// one haplotype's contribution to getHaplotypePrior: its indel-map variants other than reference alleles and SNPs, and its
// SNP-map variants other than reference alleles and "=>D", each as std::set<AlignedVariant> would hold them
void haplotypeTermLists(const Haplotype &h, const CandidateTerms &terms, TermList &indels, TermList &snps)
{
    typedef std::map<int, AlignedVariant>::const_iterator It;
    indels.clear(); snps.clear();
    for (It it = h.indels.begin(); it != h.indels.end(); ++it) {
        const std::string &s = it->second.getString();
        if (s.find("*REF") == std::string::npos && s.find("=>") == std::string::npos) indels.push_back(std::make_pair(&it->second, 0.0));
    }
    for (It it = h.snps.begin(); it != h.snps.end(); ++it) {
        const std::string &s = it->second.getString();
        if (s.find("*REF") == std::string::npos && s.find("=>D") == std::string::npos) snps.push_back(std::make_pair(&it->second, 0.0));
    }
    TermList *lists[2] = {&indels, &snps};
    for (int k = 0; k < 2; k++) {
        TermList &L = *lists[k];
        std::sort(L.begin(), L.end(), termLess);
        L.erase(std::unique(L.begin(), L.end(), termSame), L.end());
        for (size_t i = 0; i < L.size(); i++) L[i].second = terms.haplotypeTerm(*L[i].first);
    }
}

// ---- read sets --------------------------------------------------------------------------------------------------------------------
struct BitRows {                                            // rows of nr bits
    size_t words; std::vector<uint64_t> bits;
    explicit BitRows(size_t nr) : words((nr + 63) / 64) {}
    size_t addRow() { bits.resize(bits.size() + words, 0); return bits.size() / words - 1; }
    uint64_t *row(size_t i) { return &bits[i * words]; }
    static int count(const uint64_t *r, size_t words) { int n = 0; for (size_t w = 0; w < words; w++) n += __builtin_popcountll(r[w]); return n; }
};

// every entry of every haplotype's indels map, and the window's distinct variants in std::set<PAV> order
struct WindowVariants {
    struct Entry { int key; const AlignedVariant *av; int id; };        // id: number of the entry's distinct variant
    std::vector<std::vector<Entry> > ofHap;                             // in map order
    std::vector<std::pair<int, const AlignedVariant *> > distinct;
    static bool samePAV(const std::pair<int, const AlignedVariant *> &a, const std::pair<int, const AlignedVariant *> &b)
    {
        return a.first == b.first && VariantOrder::sameAV(*a.second, *b.second);
    }
    explicit WindowVariants(const std::vector<Haplotype> &haps) : ofHap(haps.size())
    {
        typedef std::map<int, AlignedVariant>::const_iterator It;
        size_t total = 0;
        for (size_t h = 0; h < haps.size(); h++) { ofHap[h].reserve(haps[h].indels.size()); total += haps[h].indels.size(); }
        distinct.reserve(total);
        for (size_t h = 0; h < haps.size(); h++)
            for (It it = haps[h].indels.begin(); it != haps[h].indels.end(); ++it) {
                Entry e = { it->first, &it->second, -1 };
                ofHap[h].push_back(e);
                distinct.push_back(std::make_pair(it->first, &it->second));
            }
        std::sort(distinct.begin(), distinct.end(), VariantOrder());
        distinct.erase(std::unique(distinct.begin(), distinct.end(), samePAV), distinct.end());
        for (size_t h = 0; h < haps.size(); h++) {          // a haplotype's entries ascend by key, so their numbers are found in one walk
            size_t at = 0;
            for (size_t i = 0; i < ofHap[h].size(); i++) {
                const std::pair<int, const AlignedVariant *> probe(ofHap[h][i].key, ofHap[h][i].av);
                while (at < distinct.size() && VariantOrder()(distinct[at], probe)) at++;
                ofHap[h][i].id = int(at);
            }
        }
    }
    int idOf(int key, const AlignedVariant &av) const
    {
        const std::pair<int, const AlignedVariant *> probe(key, &av);
        const size_t at = size_t(std::lower_bound(distinct.begin(), distinct.end(), probe, VariantOrder()) - distinct.begin());
        return (at < distinct.size() && samePAV(distinct[at], probe)) ? int(at) : -1;
    }
    const Entry *entryAt(size_t h, int key) const           // haps[h].indels.find(key)
    {
        const std::vector<Entry> &v = ofHap[h];
        size_t lo = 0, hi = v.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (v[mid].key < key) lo = mid + 1; else hi = mid; }
        return (lo < v.size() && v[lo].key == key) ? &v[lo] : NULL;
    }
};

// DetInDel::filterHaplotypes (:1932-2100) on the numbered variants: filtered[h], and per variant the reads of either strand that
// cover it on a haplotype that is kept
struct Coverage {
    std::vector<int> filtered, nf, nr;                      // per haplotype; per distinct variant
};

void coverageOf(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks, const WindowVariants &vars, bool doFilter,
                Coverage &out)
{
    const size_t nh = haps.size(), nr = reads.size();
    BitRows rows(nr);
    const size_t words = rows.words;
    std::vector<uint64_t> reverse(words, 0), selected(words);
    for (size_t r = 0; r < nr; r++) {                                                        // the strand a read counts for (:1982-1987)
        const bool rev = reads[r].isUnmapped() ? !reads[r].mateIsReverse() : reads[r].isReverse();
        if (rev) reverse[r >> 6] |= uint64_t(1) << (r & 63);
    }
    struct Covered { int id; size_t hap, row; };
    std::vector<Covered> found;
    out.filtered.assign(nh, 0);
    for (size_t h = 0; h < nh; h++) {
        const WindowLikelihoods::Rows row = liks.rows(h);
        std::fill(selected.begin(), selected.end(), 0);                                      // reads on the haplotype without an indel of their own (:1951-1955)
        for (size_t r = 0; r < nr; r++) if (!row.offHapHMQ[r] && row.numIndels[r] == 0) selected[r >> 6] |= uint64_t(1) << (r & 63);
        bool allCovered = true;
        const std::vector<WindowVariants::Entry> &entries = vars.ofHap[h];
        for (size_t i = 0; i < entries.size(); i++) {
            if (!entries[i].av->isIndel()) continue;
            const int slot = liks.varSlot(h, entries[i].key, false);
            const size_t at = rows.addRow();
            uint64_t *bits = rows.row(at);
            bool any = false;
            for (size_t w = 0; w < words; w++) {
                uint64_t m = selected[w], got = 0;
                while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    if (row.filterCovered(w * 64 + size_t(b), slot)) got |= uint64_t(1) << b;       // the device's test (:1989-2054)
                }
                bits[w] = got;
                any = any || got != 0;
            }
            if (!any) { allCovered = false; break; }                                         // :2060-2063: the haplotype's later variants are not looked at
            Covered c = { entries[i].id, h, at };
            found.push_back(c);
        }
        if (doFilter && !allCovered) out.filtered[h] = 1;                                    // :2068-2073
    }
    const size_t nd = vars.distinct.size();
    std::vector<uint64_t> fwd(nd * words, 0), rev(nd * words, 0);                            // :2088-2097: the union over the haplotypes that are kept
    for (size_t k = 0; k < found.size(); k++) if (out.filtered[found[k].hap] != 1) {
        const uint64_t *bits = rows.row(found[k].row);
        for (size_t w = 0; w < words; w++) { fwd[size_t(found[k].id) * words + w] |= bits[w] & ~reverse[w]; rev[size_t(found[k].id) * words + w] |= bits[w] & reverse[w]; }
    }
    out.nf.assign(nd, 0); out.nr.assign(nd, 0);
    for (size_t v = 0; v < nd; v++) { out.nf[v] = BitRows::count(&fwd[v * words], words); out.nr[v] = BitRows::count(&rev[v * words], words); }
}

// DINDEL_REDUCE_TIMING=1: seconds per section of diploidGLF, summed over all calls and threads, printed at exit (profiling aid)
struct SectionClock {
    static const int N = 8;
    static std::atomic<long long> total[N];
    static bool enabled() { static const bool on = getenv("DINDEL_REDUCE_TIMING") != NULL; return on; }
    static void report()
    {
        static const char *name[N] = {"variants + coverage", "ll table", "variant tables", "haplotype priors", "pair sums", "dip.map lines", "position lines", ""};
        for (int i = 0; i < 7; i++) fprintf(stderr, "reduce_timing: %-20s %.3f s\n", name[i], double(total[i].load()) * 1e-9);
    }
    std::chrono::steady_clock::time_point t;
    bool on;
    SectionClock() : on(enabled())
    {
        if (on) { static const int once = atexit(report); (void)once; t = std::chrono::steady_clock::now(); }
    }
    void mark(int section)
    {
        if (!on) return;
        const std::chrono::steady_clock::time_point n = std::chrono::steady_clock::now();
        total[section] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count();
        t = n;
    }
};
std::atomic<long long> SectionClock::total[SectionClock::N];

// sums[p] += terms[r * n + p] for r = 0 .. nr-1, every p: each pair's additions in read order (fp64 + is not associative), the pairs
// side by side — four to a vector register where the CPU has AVX2 (lane-wise adds: the same sums)
void addRowsScalar(double *sums, const double *terms, size_t n, size_t nr)
{
    for (size_t r = 0; r < nr; r++) { const double *t = terms + r * n; for (size_t p = 0; p < n; p++) sums[p] += t[p]; }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) void addRowsAvx2(double *sums, const double *terms, size_t n, size_t nr)
{
    size_t p = 0;
    for (; p + 12 <= n; p += 12) {
        __m256d a0 = _mm256_loadu_pd(sums + p), a1 = _mm256_loadu_pd(sums + p + 4), a2 = _mm256_loadu_pd(sums + p + 8);
        const double *t = terms + p;
        for (size_t r = 0; r < nr; r++, t += n) {
            a0 = _mm256_add_pd(a0, _mm256_loadu_pd(t)); a1 = _mm256_add_pd(a1, _mm256_loadu_pd(t + 4)); a2 = _mm256_add_pd(a2, _mm256_loadu_pd(t + 8));
        }
        _mm256_storeu_pd(sums + p, a0); _mm256_storeu_pd(sums + p + 4, a1); _mm256_storeu_pd(sums + p + 8, a2);
    }
    for (; p + 4 <= n; p += 4) {
        __m256d a0 = _mm256_loadu_pd(sums + p);
        const double *t = terms + p;
        for (size_t r = 0; r < nr; r++, t += n) a0 = _mm256_add_pd(a0, _mm256_loadu_pd(t));
        _mm256_storeu_pd(sums + p, a0);
    }
    for (; p < n; p++) { double a = sums[p]; const double *t = terms + p; for (size_t r = 0; r < nr; r++, t += n) a += *t; sums[p] = a; }
}
#endif
void addRows(double *sums, const double *terms, size_t n, size_t nr)
{
#if defined(__x86_64__)
    static const bool avx2 = (__builtin_cpu_init(), __builtin_cpu_supports("avx2"));
    if (avx2) { addRowsAvx2(sums, terms, n, nr); return; }
#endif
    addRowsScalar(sums, terms, n, nr);
}

void appendInt(std::string &s, int v) { char b[16]; const int n = snprintf(b, sizeof(b), "%d", v); s.append(b, size_t(n)); }

} // namespace

// ---- the reference's helpers under their own names (realigned_bam.cpp and callers of the header use them) -------------------------------
void filterHaplotypes(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks,
                      std::vector<int> &filtered, std::map<PAV, VariantCoverage> &varCoverage, bool doFilter)
{
    const WindowVariants vars(haps);
    Coverage cov;
    coverageOf(haps, reads, liks, vars, doFilter, cov);
    filtered = cov.filtered;
    varCoverage.clear();
    // the reference's map holds an entry for every variant it walked past; variants behind an uncovered one of every haplotype that
    // carries them have none there and read as (0, 0) when asked for — here every variant has its (possibly zero) entry
    for (size_t v = 0; v < vars.distinct.size(); v++) varCoverage[PAV(vars.distinct[v].first, *vars.distinct[v].second)] = VariantCoverage(cov.nf[v], cov.nr[v]);
}

double getPairPrior(const AlignedVariant &av1, const AlignedVariant &av2, int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params)
{
    const CandidateTerms terms(candidateVariants, leftPos, params);                          // :1837-1854: over std::set{av1, av2}
    if (VariantOrder::sameAV(av1, av2)) return 0.0 + terms.siteTerm(av1);
    const bool firstIs1 = VariantOrder::lessAV(av1, av2);
    return (0.0 + terms.siteTerm(firstIs1 ? av1 : av2)) + terms.siteTerm(firstIs1 ? av2 : av1);
}

double getHaplotypePrior(const Haplotype &h1, const Haplotype &h2, int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params)
{
    const CandidateTerms terms(candidateVariants, leftPos, params);                          // :1857-1927
    TermList i1, s1, i2, s2;
    haplotypeTermLists(h1, terms, i1, s1);
    haplotypeTermLists(h2, terms, i2, s2);
    return sumOverUnion(s1, s2, sumOverUnion(i1, i2, 0.0));
}

// ---- DetInDel::diploidGLF (:2933-3660) ---------------------------------------------------------------------------------------------------
void diploidGLF(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks, uint32_t candPos,
                uint32_t leftPos, uint32_t rightPos, OutputData &glfData, int index, const std::string &tid,
                const AlignedCandidates &candidateVariants, const DiploidParameters &params, const std::string &program)
{
    SectionClock clk;
    const size_t nh = haps.size(), nr = reads.size();
    const WindowVariants vars(haps);
    Coverage cov;
    coverageOf(haps, reads, liks, vars, params.filterHaplotypes, cov);                       // filterHaplotypes, :2941
    const std::vector<int> &filtered = cov.filtered;
    clk.mark(0);

    std::vector<double> rl(nh * nr);                                                         // read-major log-likelihoods (:2943-2961)
    for (size_t h = 0; h < nh; h++) {
        const double *llh = liks.rows(h).ll;
        for (size_t r = 0; r < nr; r++) rl[r * nh + h] = llh[r];
    }
    clk.mark(1);

    // ---- per distinct variant: is it one of the window file's candidates; does it count as a variant of the window (:2982-3016) ----
    const CandidateTerms terms(candidateVariants, int(leftPos), params);
    const size_t nd = vars.distinct.size();
    std::vector<char> isCandidate(nd);
    std::vector<int> variantId(1, -1);                      // number (1-based, allVariants' order) -> distinct id
    std::vector<int> posKey;                                // the variant positions, ascending (allVariantsByPos)
    for (size_t v = 0; v < nd; v++) {
        const AlignedVariant &av = *vars.distinct[v].second;
        isCandidate[v] = terms.find(av) != NULL;
        if (!av.isRef() && !(av.isSNP() && av.getString()[3] == 'D')) {
            variantId.push_back(int(v));
            if (posKey.empty() || posKey.back() != vars.distinct[v].first) posKey.push_back(vars.distinct[v].first);
        }
    }
    const size_t nv = variantId.size() - 1, numVarPos = posKey.size();
    std::vector<int> hapCandidateIndels(nh, 0);             // hap_num_candidate_indels
    for (size_t h = 0; h < nh; h++) {
        bool anyIndel = false;
        int nc = 0;
        for (size_t i = 0; i < vars.ofHap[h].size(); i++) { anyIndel = anyIndel || vars.ofHap[h][i].av->isIndel(); nc += isCandidate[size_t(vars.ofHap[h][i].id)] ? 1 : 0; }
        if (anyIndel) hapCandidateIndels[h] = nc;           // counted over every entry, but only for haplotypes with an indel (:2984-2993)
    }
    // hapVar[h][p]: number of the window variant haplotype h carries at position p — the LAST of the position's variants whose string
    // equals the haplotype's (:3036-3050 assigns in ascending order of the variants)
    std::vector<int> hapVar(nh * numVarPos, 0);
    for (size_t n = 1; n <= nv; n++) {
        const int key = vars.distinct[size_t(variantId[n])].first;
        const size_t p = size_t(std::lower_bound(posKey.begin(), posKey.end(), key) - posKey.begin());
        const std::string &str = vars.distinct[size_t(variantId[n])].second->getString();
        for (size_t h = 0; h < nh; h++) {
            const WindowVariants::Entry *e = vars.entryAt(h, key);
            if (e && e->av->getString() == str) hapVar[h * numVarPos + p] = int(n);
        }
    }
    std::vector<double> mqOf(nr);                                                            // -10 log10(1 - mapQual)
    for (size_t r = 0; r < nr; r++) mqOf[r] = -10 * log10(1.0 - reads[r].mapQual);
    clk.mark(2);

    // ---- haplotype-pair priors (:3068-3075) ----
    std::vector<double> prior(nh * nh, 0.0), pairs_posterior(nh * nh, 0.0);
    {
        std::vector<TermList> indelTerms(nh), snpTerms(nh);
        for (size_t h = 0; h < nh; h++) haplotypeTermLists(haps[h], terms, indelTerms[h], snpTerms[h]);
        for (size_t h1 = 0; h1 < nh; h1++)
            for (size_t h2 = h1; h2 < nh; h2++) prior[h1 * nh + h2] = sumOverUnion(snpTerms[h1], snpTerms[h2], sumOverUnion(indelTerms[h1], indelTerms[h2], 0.0));
    }
    clk.mark(3);

    // ---- read sums per haplotype pair, MAP pairs, qual (:3083-3121) ----
    // log(0.5) + addLogs(rl[r][h1], rl[r][h2]) per pair and read: the reference evaluates it here and again for every variant position
    // (:3372-3376); the values are kept (read-major: the pairs of one read side by side) and every sum is redone in the reference's
    // order, read after read — all pairs at once, so that the additions of different pairs overlap.
    std::vector<size_t> pairH1, pairH2;
    for (size_t h1 = 0; h1 < nh; h1++) if (filtered[h1] == 0) for (size_t h2 = h1; h2 < nh; h2++) if (filtered[h2] == 0) { pairH1.push_back(h1); pairH2.push_back(h2); }
    const size_t nPairs = pairH1.size();
    const bool keepTerms = nPairs * nr <= (size_t(1) << 23);                                  // at most 64 MB; beyond that they are recomputed
    std::vector<double> pairTerm(keepTerms ? nPairs * nr : nPairs), pairSum(nPairs, 0.0);
    const double logHalf = log(0.5);
    // A read that does not reach the place where two haplotypes differ has the same log-likelihood on both, bit for bit (the same
    // path, the same terms in the same order), so a read's nh values fall into a few classes of equal values; addLogs is a function
    // of the two values alone, so it is evaluated once per pair of classes and looked up per pair of haplotypes (exp + log are
    // what this stage spends its time in: 36 pairs x 200 reads per window of 8 haplotypes).
    std::vector<int> classOf(nh);
    std::vector<double> classValue(nh), classTerm(nh * nh);
    auto addTermsOfAllReads = [&](double *sums, bool compute) {
        for (size_t r = 0; r < nr; r++) {
            double *term = keepTerms ? &pairTerm[r * nPairs] : &pairTerm[0];
            if (compute) {
                const double *rlr = &rl[r * nh];
                size_t nc = 0;
                for (size_t h = 0; h < nh; h++) {
                    size_t c = 0;
                    while (c < nc && memcmp(&classValue[c], &rlr[h], sizeof(double)) != 0) c++;
                    if (c == nc) classValue[nc++] = rlr[h];
                    classOf[h] = int(c);
                }
                for (size_t c1 = 0; c1 < nc; c1++)
                    for (size_t c2 = c1; c2 < nc; c2++) classTerm[c1 * nh + c2] = classTerm[c2 * nh + c1] = logHalf + addLogs(classValue[c1], classValue[c2]);
                for (size_t p = 0; p < nPairs; p++) term[p] = classTerm[size_t(classOf[pairH1[p]]) * nh + size_t(classOf[pairH2[p]])];
            }
            if (!keepTerms) for (size_t p = 0; p < nPairs; p++) sums[p] += term[p];
        }
        if (keepTerms) addRows(sums, pairTerm.data(), nPairs, nr);
    };
    addTermsOfAllReads(pairSum.data(), true);
    int mapIndelPair[2] = {-1, -1};
    double max_ll_indel = -HUGE_VAL, max_ll_noindel = -HUGE_VAL;
    for (size_t p = 0; p < nPairs; p++) {
        const size_t h1 = pairH1[p], h2 = pairH2[p];
        const double pp = pairs_posterior[h1 * nh + h2] = pairSum[p] + prior[h1 * nh + h2];
        const bool withIndel = hapCandidateIndels[h1] > 0 || hapCandidateIndels[h2] > 0;
        if (withIndel && pp > max_ll_indel) { max_ll_indel = pp; mapIndelPair[0] = int(h1); mapIndelPair[1] = int(h2); }
        if (!withIndel && pp > max_ll_noindel) max_ll_noindel = pp;
    }
    clk.mark(4);
    const double ll_ref = max_ll_noindel;
    const double qual = -10.0 * (ll_ref - addLogs(max_ll_indel, ll_ref)) / log(10.0);          // :3118
    if (!params.quiet) std::cout << "ll_ref: " << ll_ref << " max_ll_indel: " << max_ll_indel << " qual: " << qual << std::endl;
    if (mapIndelPair[0] == -1 || mapIndelPair[1] == -1) throw std::string("Could not find indel allele");   // :3121

    static thread_local std::string text;
    {   // ---- one dip.map line per variant position of the MAP pair (:3122-3302) ----
        const size_t hx1 = size_t(mapIndelPair[0]), hx2 = size_t(mapIndelPair[1]);
        int numUnmappedRealigned = 0;
        for (size_t r = 0; r < nr; r++)
            if (reads[r].isUnmapped() && (liks.offHap(hx1, r) == false || liks.offHap(hx2, r) == false)) numUnmappedRealigned++;
        // the positions where either haplotype carries something else than a plain reference allele, each with its alleles in set order
        std::vector<std::pair<int, const AlignedVariant *> > sites;
        for (int i = 0; i < 2; i++) {
            const std::vector<WindowVariants::Entry> &E = vars.ofHap[i ? hx2 : hx1];
            for (size_t k = 0; k < E.size(); k++)
                if (!E[k].av->isRef() || (E[k].av->isSNP() && E[k].av->getString()[3] == 'D')) sites.push_back(std::make_pair(E[k].key, E[k].av));
        }
        std::sort(sites.begin(), sites.end(), VariantOrder());
        static const std::string refAllele("*REF");
        std::vector<const std::string *> alleleOf(nh);
        for (size_t s0 = 0; s0 < sites.size();) {
            size_t s1 = s0;                                                                  // [s0, s1): the alleles at this position (duplicates adjacent)
            while (s1 < sites.size() && sites[s1].first == sites[s0].first) s1++;
            const int pos = sites[s0].first;
            double msq = 0;
            int numf = 0, numr = 0, n = 0;
            const int m = (hx1 == hx2) ? 1 : 2;
            for (int i = 0; i < m; i++) {                                                    // reads covering the indel on its haplotype (:3150-3175)
                const size_t h = i ? hx2 : hx1;
                const WindowVariants::Entry *e = vars.entryAt(h, pos);
                if (e && e->av->isIndel()) {
                    const int slot = liks.varSlot(h, pos, false);
                    for (size_t r = 0; r < nr; r++)
                        if (liks.coveredAt(h, r, slot)) {                                    // liks[h][r].hapIndelCovered[pos]
                            if (reads[r].onReverseStrand) numr++; else numf++;
                            msq += mqOf[r] * mqOf[r];
                            n++;
                        }
                }
            }
            msq = n != 0 ? sqrt(msq / double(n)) : 0.0;
            const int firstId = vars.idOf(pos, *sites[s0].second), lastId = vars.idOf(pos, *sites[s1 - 1].second);
            int was_candidate = isCandidate[size_t(firstId)] ? 1 : 0;
            int vc_f = cov.nf[size_t(firstId)], vc_r = cov.nr[size_t(firstId)];
            const WindowVariants::Entry *e1 = vars.entryAt(hx1, pos), *e2 = vars.entryAt(hx2, pos);
            const bool a1_ref = !(e1 && !e1->av->isRef()), a2_ref = !(e2 && !e2->av->isRef());
            const std::string &a1 = a1_ref ? refAllele : e1->av->getString(), &a2 = a2_ref ? refAllele : e2->av->getString();
            if (a1_ref && a2_ref) throw std::string("genotyping error");
            std::string genotype, nref_all;
            if (a1 == a2) { genotype = "1/1"; nref_all = a1; }
            else if (a1_ref) { genotype = "0/1"; nref_all = a2; }
            else if (a2_ref) { genotype = "0/1"; nref_all = a1; }
            else {
                nref_all = a1 + ',' + a2;
                genotype = "1/2";
                if (isCandidate[size_t(lastId)]) was_candidate = 1;                          // the second allele counts too (:3226-3230)
                vc_f += cov.nf[size_t(lastId)];
                vc_r += cov.nr[size_t(lastId)];
            }
            // genotype quality: the best pair that says something else at this position (:3238-3265)
            for (size_t h = 0; h < nh; h++) {
                const WindowVariants::Entry *e = vars.entryAt(h, pos);
                alleleOf[h] = (e == NULL || e->av->isRef()) ? &refAllele : &e->av->getString();
            }
            double max_ll_altgeno = -HUGE_VAL;
            for (size_t p = 0; p < nPairs; p++) {
                const size_t h1 = pairH1[p], h2 = pairH2[p];
                if ((h1 == hx1 && h2 == hx2) || (h2 == hx1 && h1 == hx2)) continue;
                const std::string &x = *alleleOf[h1], &y = *alleleOf[h2];                    // {x, y} != {a1, a2} as sets of strings
                const bool same = (x == a1 || x == a2) && (y == a1 || y == a2) && (a1 == x || a1 == y) && (a2 == x || a2 == y);
                if (!same && max_ll_altgeno < pairs_posterior[h1 * nh + h2]) max_ll_altgeno = pairs_posterior[h1 * nh + h2];
            }
            DipMapCall c;
            c.index = index; c.tid = tid; c.leftPos = leftPos; c.rightPos = rightPos; c.candPos = candPos; c.realignedPos = pos + int(leftPos);
            c.was_candidate = was_candidate; c.qual = qual; c.nref_all = nref_all; c.num_reads = nr; c.msq = msq; c.numf = numf; c.numr = numr;
            c.vc_f = vc_f; c.vc_r = vc_r; c.numUnmappedRealigned = numUnmappedRealigned; c.genotype = genotype;
            c.genoqual = -10.0 * (max_ll_altgeno - addLogs(max_ll_indel, max_ll_altgeno)) / log(10.0);
            if (glfData.hasGLFColumns()) { dipMapText(text, c); glfData.outputText(text); }
            else glfData.output(dipMapLine(glfData, c));
            s0 = s1;
        }
    }
    clk.mark(5);

    // ---- per variant position: genotype likelihoods over the haplotype pairs, coverage and QC sums (:3305-3660) ----
    const AlignedVariant plainRef("*REF", -1);
    std::vector<double> siteTermOf(nv + 1), sitePrior((nv + 1) * (nv + 1));                  // getPairPrior of the variants numbered (v1, v2), on first use
    std::vector<char> sitePriorKnown((nv + 1) * (nv + 1), 0);
    for (size_t n = 0; n <= nv; n++) siteTermOf[n] = terms.siteTerm(n ? *vars.distinct[size_t(variantId[n])].second : plainRef);
    std::vector<int> indelCountOf(nh * nr, -1);
    std::vector<double> genLik((nv + 1) * (nv + 2));                                         // [smaller * (nv + 2) + (larger + 1)], slot 0: both alleles the same
    std::vector<char> genSeen((nv + 1) * (nv + 2));
    std::vector<int> genKeys, vcfIdx(nv + 1), slotOf(nh);
    std::string alleles, covF, covR, glf;
    for (size_t p0 = 0; p0 < numVarPos; p0++) {
        const int pos = posKey[p0];
        int has_variants_in_window = 0;
        for (size_t n = 1; n <= nv && !has_variants_in_window; n++)
            if (vars.distinct[size_t(variantId[n])].first == pos && isCandidate[size_t(variantId[n])]) has_variants_in_window = 1;
        for (size_t p = 0; p < nPairs; p++) {                                                // :3347-3389, the starting values: the site's prior taken out again
            const size_t h1 = pairH1[p], h2 = pairH2[p];
            const size_t v1 = size_t(hapVar[h1 * numVarPos + p0]), v2 = size_t(hapVar[h2 * numVarPos + p0]);
            const size_t at = v1 * (nv + 1) + v2;
            if (!sitePriorKnown[at]) {
                const AlignedVariant &A = v1 ? *vars.distinct[size_t(variantId[v1])].second : plainRef, &B = v2 ? *vars.distinct[size_t(variantId[v2])].second : plainRef;
                if (VariantOrder::sameAV(A, B)) sitePrior[at] = 0.0 + siteTermOf[v1];
                else if (VariantOrder::lessAV(A, B)) sitePrior[at] = (0.0 + siteTermOf[v1]) + siteTermOf[v2];
                else sitePrior[at] = (0.0 + siteTermOf[v2]) + siteTermOf[v1];
                sitePriorKnown[at] = 1;
            }
            pairSum[p] = prior[h1 * nh + h2] - sitePrior[at];
        }
        addTermsOfAllReads(pairSum.data(), !keepTerms);
        genKeys.clear();
        std::fill(genSeen.begin(), genSeen.end(), 0);
        double maxll = -HUGE_VAL;
        size_t hx1 = 0, hx2 = 0;
        for (size_t p = 0; p < nPairs; p++) {
            const size_t h1 = pairH1[p], h2 = pairH2[p];
            const int v1 = hapVar[h1 * numVarPos + p0], v2 = hapVar[h2 * numVarPos + p0];
            const size_t key = size_t(std::min(v1, v2)) * (nv + 2) + size_t(v1 == v2 ? 0 : std::max(v1, v2) + 1);   // orders like set<int>{v1, v2}: {v} before {v, w}
            const double ll = pairSum[p];
            if (!genSeen[key]) { genSeen[key] = 1; genLik[key] = ll; genKeys.push_back(int(key)); }
            else genLik[key] = addLogs(genLik[key], ll);
            if (ll > maxll) { maxll = ll; hx1 = h1; hx2 = h2; }
        }
        std::sort(genKeys.begin(), genKeys.end());
        const WindowLikelihoods::Rows row1 = liks.rows(hx1), row2 = liks.rows(hx2);
        int numUnmappedRealigned = 0;                                                        // :3395-3402
        for (size_t r = 0; r < nr; r++)
            if (reads[r].isUnmapped() && (row1.offHap[r] == 0 || row2.offHap[r] == 0)) numUnmappedRealigned++;
        double allmsq = 0.0, msq = 0.0, mLogBQ = 0.0;                                        // :3492-3560
        int numMappedIndels = 0, nBQT = 0, nmmBQT = 0, nMMLeft = 0, nMMRight = 0, numOffBoth = 0, nf = 0, nrv = 0, n = 0;
        std::fill(slotOf.begin(), slotOf.end(), -2);                                         // the position's slot in a haplotype's flag list, on first use
        for (size_t r = 0; r < nr; r++) {
            const double mq = mqOf[r];
            allmsq += (mq * mq);
            if (row1.offHap[r] && row2.offHap[r]) numOffBoth++;
            const bool first = row1.ll[r] >= row2.ll[r];
            const size_t h = first ? hx1 : hx2;                                              // the read's better haplotype of the pair
            const WindowLikelihoods::Rows &row = first ? row1 : row2;
            int &known = indelCountOf[h * nr + r];                                           // liks[h][r].indels.size(), :3529: asked for at every position
            if (known < 0) known = liks.indelCount(h, r);
            numMappedIndels += known;
            nBQT += row.nBQT[r];
            nmmBQT += row.nmmBQT[r];
            mLogBQ += row.mLogBQ[r];
            if (row.nMMLeft[r] >= 2) nMMLeft++;
            if (row.nMMRight[r] >= 2) nMMRight++;
            if (slotOf[h] == -2) {                                                           // :3536-3541
                slotOf[h] = -1;
                const WindowVariants::Entry *e = vars.entryAt(h, pos);
                if (e && e->av->isIndel()) slotOf[h] = liks.varSlot(h, pos, false);          // liks[h][r].hapIndelCovered[pos]
                else if (e && e->av->isSNP()) slotOf[h] = liks.varSlot(h, pos, true);        // liks[h][r].hapSNPCovered[pos]
            }
            if (row.covered(r, slotOf[h])) {
                if (reads[r].onReverseStrand) nrv++; else nf++;
                msq += mq * mq;
                n++;
            }
        }
        msq = n != 0 ? sqrt(msq / double(n)) : 0.0;
        allmsq = nr != 0 ? sqrt(allmsq / double(nr)) : 0;
        // the position's alleles numbered as the VCF will number them: in order of first appearance over the haplotypes (:3572-3595)
        std::fill(vcfIdx.begin(), vcfIdx.end(), -1);
        vcfIdx[0] = 0;
        int nidx = 1;
        alleles.clear(); covF.clear(); covR.clear(); glf.clear();
        for (size_t h = 0; h < nh; h++) {
            const int v = hapVar[h * numVarPos + p0];
            if (v != 0 && vcfIdx[size_t(v)] < 0) {
                if (nidx > 1) { alleles += ','; covF += ','; covR += ','; }
                vcfIdx[size_t(v)] = nidx++;
                alleles += vars.distinct[size_t(variantId[size_t(v)])].second->getString();
                appendInt(covF, cov.nf[size_t(variantId[size_t(v)])]);
                appendInt(covR, cov.nr[size_t(variantId[size_t(v)])]);
            }
        }
        for (size_t k = 0; k < genKeys.size(); k++) {                                        // :3600-3613
            const size_t a = size_t(genKeys[k]) / (nv + 2), b = size_t(genKeys[k]) % (nv + 2);
            char cell[64];
            if (k) glf += ',';
            appendInt(glf, vcfIdx[a]); glf += '/'; appendInt(glf, vcfIdx[b == 0 ? a : b - 1]); glf += ':';
            glf.append(cell, size_t(formatG6(genLik[size_t(genKeys[k])], cell)));
        }
        if (params.outputGLF) {
            DipPositionRow d;
            d.index = index; d.tid = tid; d.program = program; d.leftPos = leftPos; d.rightPos = rightPos; d.candPos = candPos; d.realignedPos = pos + int(leftPos);
            d.has_variants_in_window = has_variants_in_window; d.logZ = maxll; d.nBQT = nBQT; d.nmmBQT = nmmBQT; d.mLogBQ = mLogBQ; d.nMMLeft = nMMLeft;
            d.nMMRight = nMMRight; d.nref_all = alleles; d.num_reads = nr; d.msq = allmsq; d.numOffAll = numOffBoth; d.num_indel = numMappedIndels;
            d.nf = nf; d.nr = nrv; d.var_coverage_forward = covF; d.var_coverage_reverse = covR; d.glf = glf;
            d.numUnmappedRealigned = numUnmappedRealigned;
            if (glfData.hasGLFColumns()) { dipPositionText(text, d); glfData.outputText(text); }
            else glfData.output(dipPositionLine(glfData, d));
        }
    }
    clk.mark(6);
}

} // namespace dindel
