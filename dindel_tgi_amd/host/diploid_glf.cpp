// diploid_glf.cpp — see diploid_glf.hpp.  Line references are to the reference's DInDel.cpp.
#include "diploid_glf.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <iostream>
#include <set>
#include <sstream>

namespace dindel {

void filterHaplotypes(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks,
                      std::vector<int> &filtered, std::map<PAV, VariantCoverage> &varCoverage, bool doFilter)
{
    const int numHaps = int(haps.size());
    filtered = std::vector<int>(haps.size(), 0);
    varCoverage.clear();
    typedef std::map<int, AlignedVariant>::const_iterator It;
    // per variant, per (haplotype, strand): the reads covering it (:1940-1944).  Reads are visited in ascending order, so the
    // reference's set<int> is a plain ascending list here.
    std::map<PAV, std::vector<std::vector<int> > > hVarCoverage;
    std::vector<int> selReads, strandOf(reads.size(), 0);
    for (size_t r = 0; r < reads.size(); r++) {                                              // :1982-1987
        if (reads[r].isUnmapped()) { if (!reads[r].mateIsReverse()) strandOf[r] = 1; }
        else { if (reads[r].isReverse()) strandOf[r] = 1; }
    }
    for (int h = 0; h < numHaps; h++) {
        const WindowLikelihoods::Rows row = liks.rows(size_t(h));
        selReads.clear();                                                                    // :1951-1955
        for (size_t r = 0; r < reads.size(); r++)
            if (!row.offHapHMQ[r] && row.numIndels[r] == 0) selReads.push_back(int(r));
        bool allCovered = true;
        for (It it = haps[size_t(h)].indels.begin(); it != haps[size_t(h)].indels.end(); ++it) {
            const AlignedVariant &av = it->second;
            const PAV pav(it->first, av);
            std::map<PAV, std::vector<std::vector<int> > >::iterator cv = hVarCoverage.find(pav);
            if (cv == hVarCoverage.end()) cv = hVarCoverage.insert(std::make_pair(pav, std::vector<std::vector<int> >(haps.size() * 2))).first;
            if (av.getType() == AlignedVariant::INS || av.getType() == AlignedVariant::DEL) {
                bool covered = false;
                const int slot = liks.varSlot(size_t(h), it->first, false);
                for (size_t k = 0; k < selReads.size(); k++) {
                    const int r = selReads[k];
                    if (row.filterCovered(size_t(r), slot)) {                                // the device's test (:1989-2054)
                        cv->second[size_t(h + strandOf[size_t(r)] * numHaps)].push_back(r);
                        covered = true;
                    }
                }
                if (!covered) { allCovered = false; break; }                                 // :2060-2063
            }
        }
        if (doFilter && !allCovered) filtered[size_t(h)] = 1;                                // :2068-2073
    }
    std::vector<char> seenF(reads.size()), seenR(reads.size());
    for (std::map<PAV, std::vector<std::vector<int> > >::const_iterator it = hVarCoverage.begin(); it != hVarCoverage.end(); ++it) {
        std::fill(seenF.begin(), seenF.end(), 0);                                            // :2088-2097: the union over the haplotypes kept
        std::fill(seenR.begin(), seenR.end(), 0);
        int nf = 0, nr = 0;
        for (int h = 0; h < numHaps; h++) if (filtered[size_t(h)] != 1) {
            const std::vector<int> &f = it->second[size_t(h)], &rv = it->second[size_t(h + numHaps)];
            for (size_t k = 0; k < f.size(); k++) if (!seenF[size_t(f[k])]) { seenF[size_t(f[k])] = 1; nf++; }
            for (size_t k = 0; k < rv.size(); k++) if (!seenR[size_t(rv[k])]) { seenR[size_t(rv[k])] = 1; nr++; }
        }
        varCoverage[it->first] = VariantCoverage(nf, nr);
    }
}

double getPairPrior(const AlignedVariant &av1, const AlignedVariant &av2, int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params)
{
    std::set<AlignedVariant> vars;                                                           // :1837-1854
    vars.insert(av1); vars.insert(av2);
    double ll = 0.0;
    for (std::set<AlignedVariant>::const_iterator vt = vars.begin(); vt != vars.end(); ++vt) {
        const AlignedVariant &avar = *vt;
        double lnf = 0.0;
        const int type = avar.getType();
        const AlignedVariant *av = candidateVariants.findVariant(avar.getStartHap() + leftPos, avar.getType(), avar.getString());
        if (type == AlignedVariant::SNP) lnf = log(params.priorSNP);
        else if (type == AlignedVariant::DEL || type == AlignedVariant::INS) lnf = log(params.priorIndel);
        if (av == NULL) ll += lnf;
        else { const double prior = av->getFreq(); if (prior < 0.0) ll += lnf; else ll += log(prior); }
    }
    return ll;
}

double getHaplotypePrior(const Haplotype &h1, const Haplotype &h2, int leftPos, const AlignedCandidates &candidateVariants, const DiploidParameters &params)
{
    double ll = 0.0;                                                                         // :1857-1927
    typedef std::map<int, AlignedVariant>::const_iterator AVIt;
    std::set<AlignedVariant> indels, snps;
    const Haplotype *hs[2] = {&h1, &h2};
    for (int i = 0; i < 2; i++)
        for (AVIt it = hs[i]->indels.begin(); it != hs[i]->indels.end(); ++it)
            if (it->second.getString().find("*REF") == std::string::npos && it->second.getString().find("=>") == std::string::npos) indels.insert(it->second);
    for (int i = 0; i < 2; i++)
        for (AVIt it = hs[i]->snps.begin(); it != hs[i]->snps.end(); ++it)
            if (it->second.getString().find("*REF") == std::string::npos && it->second.getString().find("=>D") == std::string::npos) snps.insert(it->second);
    const std::set<AlignedVariant> *sets[2] = {&indels, &snps};
    for (int s = 0; s < 2; s++)                                                              // SNPs get priorIndel too, as written (:1899-1908)
        for (std::set<AlignedVariant>::const_iterator vt = sets[s]->begin(); vt != sets[s]->end(); ++vt) {
            const AlignedVariant *av = candidateVariants.findVariant(vt->getStartHap() + leftPos, vt->getType(), vt->getString());
            if (av == NULL) ll += log(params.priorIndel);
            else { const double prior = av->getFreq(); if (prior < 0.0) ll += log(params.priorIndel); else ll += log(prior); }
        }
    return ll;
}

namespace {
// DINDEL_REDUCE_TIMING=1: seconds per section of diploidGLF, summed over all calls and threads, printed at exit (profiling aid)
struct SectionClock {
    static const int N = 8;
    static std::atomic<long long> total[N];
    static bool enabled() { static const bool on = getenv("DINDEL_REDUCE_TIMING") != NULL; return on; }
    static void report()
    {
        static const char *name[N] = {"filterHaplotypes", "ll table", "variant tables", "haplotype priors", "pair sums", "dip.map lines", "position lines", ""};
        for (int i = 0; i < 7; i++) fprintf(stderr, "reduce_timing: %-18s %.3f s\n", name[i], double(total[i].load()) * 1e-9);
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
}

void diploidGLF(const std::vector<Haplotype> &haps, const std::vector<Read> &reads, const WindowLikelihoods &liks, uint32_t candPos,
                uint32_t leftPos, uint32_t rightPos, OutputData &glfData, int index, const std::string &tid,
                const AlignedCandidates &candidateVariants, const DiploidParameters &params, const std::string &program)
{
    SectionClock clk;
    const size_t nh = haps.size(), nr = reads.size();
    std::vector<int> filtered(nh, 0);
    std::map<PAV, VariantCoverage> varCoverage;
    filterHaplotypes(haps, reads, liks, filtered, varCoverage, params.filterHaplotypes);        // :2941
    clk.mark(0);

    std::vector<double> rl(nh * nr, 0.0);                                                    // :2943-2961
    {
        for (size_t h = 0; h < nh; h++) {
            const double *llh = liks.rows(h).ll;
            for (size_t r = 0; r < nr; r++) rl[r * nh + h] = llh[r];
        }
    }
    clk.mark(1);
    const int VARSNP = 1, VARINDEL = 2;
    std::set<PAV> allVariants;
    std::map<int, std::set<PAV> > allVariantsByPos;
    typedef std::map<int, AlignedVariant>::const_iterator It;
    typedef std::map<int, std::set<PAV> >::const_iterator PIt;
    std::vector<int> hap_num_indels(nh, 0), hap_num_candidate_indels(nh, 0), hap_num_snps(nh, 0);
    for (size_t th = 0; th < nh; th++) {                                                     // :2982-3016
        const Haplotype &hap = haps[th];
        hap_num_indels[th] = hap.countIndels();
        hap_num_snps[th] = hap.countSNPs();
        if (hap_num_indels[th] != 0) {
            int nc = 0;
            for (It it = hap.indels.begin(); it != hap.indels.end(); ++it) {
                const AlignedVariant &avar = it->second;
                if (candidateVariants.findVariant(avar.getStartHap() + int(leftPos), avar.getType(), avar.getString()) != NULL) nc += 1;
            }
            hap_num_candidate_indels[th] = nc;
        }
        for (It it = hap.indels.begin(); it != hap.indels.end(); ++it)
            if (!it->second.isRef() && !(it->second.isSNP() && it->second.getString()[3] == 'D')) {
                allVariants.insert(PAV(it->first, it->second));
                allVariantsByPos[it->first].insert(PAV(it->first, it->second));
            }
    }
    std::map<int, int> posToPosIdx;                                                          // :3018-3026
    {
        int idx = 0;
        for (PIt pit = allVariantsByPos.begin(); pit != allVariantsByPos.end(); ++pit) posToPosIdx[pit->first] = idx++;
    }
    const int numVarPos = int(allVariantsByPos.size());
    const int nv = int(allVariants.size());
    std::vector<int> hapVar(nh * size_t(numVarPos), 0), varType(size_t(nv) + 1);
    std::vector<PAV> variants(size_t(nv) + 1);
    {
        int idx = 1;                                                                         // :3036-3050
        for (std::set<PAV>::const_iterator pt = allVariants.begin(); pt != allVariants.end(); ++pt, ++idx) {
            const PAV &pav = *pt;
            varType[size_t(idx)] = pav.second.isIndel() ? VARINDEL : VARSNP;
            const int posIdx = posToPosIdx[pav.first];
            for (size_t h = 0; h < nh; h++) {
                It it = haps[h].indels.find(pav.first);
                if (it != haps[h].indels.end() && it->second.getString() == pav.second.getString()) hapVar[h * size_t(numVarPos) + size_t(posIdx)] = idx;
            }
            variants[size_t(idx)] = pav;
        }
    }
    const size_t numReadIdx = nr;                                                            // readidx of the reference: every read, ascending
    std::vector<double> mqOf(nr);                                                            // -10 log10(1 - mapQual), used thrice per read and variant position
    for (size_t r = 0; r < nr; r++) mqOf[r] = -10 * log10(1.0 - reads[r].mapQual);

    clk.mark(2);
    std::vector<double> prior(nh * nh, 0.0), pairs_posterior(nh * nh, 0);                    // :3068-3075
    for (size_t h1 = 0; h1 < nh; h1++)
        for (size_t h2 = h1; h2 < nh; h2++) prior[h1 * nh + h2] = getHaplotypePrior(haps[h1], haps[h2], int(leftPos), candidateVariants, params);

    clk.mark(3);
    std::vector<int> max_indel_pair(2, -1), max_noindel_pair(2, -1);
    double max_ll_indel = -HUGE_VAL, max_ll_noindel = -HUGE_VAL;
    // log(0.5) + addLogs(rl[r][h1], rl[r][h2]) per haplotype pair and read: the reference evaluates it here and again for every
    // variant position (:3372-3376); the values are kept (read-major: the pairs of one read side by side) and every sum is redone
    // in the reference's order, read after read — all pairs at once, so that the additions of different pairs overlap instead of
    // each pair waiting through its own chain of 200 dependent additions.
    std::vector<size_t> pairH1, pairH2;                                                      // the pairs that take part, in the reference's loop order
    for (size_t h1 = 0; h1 < nh; h1++) if (filtered[h1] == 0) for (size_t h2 = h1; h2 < nh; h2++) if (filtered[h2] == 0) { pairH1.push_back(h1); pairH2.push_back(h2); }
    const size_t nPairs = pairH1.size();
    const bool keepTerms = nPairs * nr <= (size_t(1) << 23);                                  // at most 64 MB; beyond that they are recomputed
    std::vector<double> pairTerm(keepTerms ? nPairs * nr : nPairs), pairSum(nPairs);
    const double logHalf = log(0.5);
    // sums[p] += term(p, r) for r = 0 .. nr-1, starting from what sums[] holds; stores the terms when they are kept
    auto addTermsOfAllReads = [&](double *sums, bool compute) {
        for (size_t r = 0; r < nr; r++) {
            double *term = keepTerms ? &pairTerm[r * nPairs] : &pairTerm[0];
            if (compute) {
                const double *rlr = &rl[r * nh];
                for (size_t p = 0; p < nPairs; p++) term[p] = logHalf + addLogs(rlr[pairH1[p]], rlr[pairH2[p]]);
            }
            for (size_t p = 0; p < nPairs; p++) sums[p] += term[p];
        }
    };
    std::fill(pairSum.begin(), pairSum.end(), 0.0);                                          // :3083-3114
    addTermsOfAllReads(pairSum.data(), true);
    for (size_t p = 0; p < nPairs; p++) {
        const size_t h1 = pairH1[p], h2 = pairH2[p];
        pairs_posterior[h1 * nh + h2] = pairSum[p] + prior[h1 * nh + h2];
        const double pp = pairs_posterior[h1 * nh + h2];
        if (pp > max_ll_indel && (hap_num_candidate_indels[h1] > 0 || hap_num_candidate_indels[h2] > 0)) { max_ll_indel = pp; max_indel_pair[0] = int(h1); max_indel_pair[1] = int(h2); }
        if (pp > max_ll_noindel && (hap_num_candidate_indels[h1] == 0 && hap_num_candidate_indels[h2] == 0)) { max_ll_noindel = pp; max_noindel_pair[0] = int(h1); max_noindel_pair[1] = int(h2); }
    }
    clk.mark(4);
    const double ll_ref = max_ll_noindel;
    const double qual = -10.0 * (ll_ref - addLogs(max_ll_indel, ll_ref)) / log(10.0);          // :3118
    if (!params.quiet) std::cout << "ll_ref: " << ll_ref << " max_ll_indel: " << max_ll_indel << " qual: " << qual << std::endl;
    if (max_indel_pair[0] == -1 || max_indel_pair[1] == -1) throw std::string("Could not find indel allele");   // :3121

    {   // ---- map-based variant calls: one dip.map line per variant position of the MAP pair (:3122-3302) ----
        int numUnmappedRealigned = 0;
        const size_t hx1 = size_t(max_indel_pair[0]), hx2 = size_t(max_indel_pair[1]);
        for (size_t r = 0; r < nr; r++)
            if (reads[r].isUnmapped() && (liks.offHap(hx1, r) == false || liks.offHap(hx2, r) == false)) numUnmappedRealigned++;
        std::map<int, std::set<AlignedVariant> > indels;
        for (int i = 0; i < 2; i++) {
            const Haplotype &hap = haps[size_t(max_indel_pair[size_t(i)])];
            for (It it = hap.indels.begin(); it != hap.indels.end(); ++it)
                if (!it->second.isRef() || (it->second.isSNP() && it->second.getString()[3] == 'D')) indels[it->first].insert(it->second);
        }
        for (std::map<int, std::set<AlignedVariant> >::const_iterator it = indels.begin(); it != indels.end(); ++it) {
            double msq = 0;
            int numf = 0, numr = 0, n = 0;
            const int m = (max_indel_pair[0] == max_indel_pair[1]) ? 1 : 2;
            for (int i = 0; i < m; i++) {
                const size_t h = size_t(max_indel_pair[size_t(i)]);
                It iter = haps[h].indels.find(it->first);
                if (iter != haps[h].indels.end() && iter->second.isIndel()) {
                    const int slot = liks.varSlot(h, it->first, false);
                    for (size_t r = 0; r < nr; r++) {
                        bool nft = false, nrt = false;
                        if (liks.coveredAt(h, r, slot)) {                                    // liks[h][r].hapIndelCovered[pos], :3158-3159
                            if (reads[r].onReverseStrand) nrt = true; else nft = true;
                            const double mq = mqOf[r];
                            msq += mq * mq;
                            n++;
                        }
                        if (nft) numf++;
                        if (nrt) numr++;
                    }
                }
            }
            if (n != 0) msq = sqrt(msq / double(n)); else msq = 0.0;
            int was_candidate = 0;
            const std::set<AlignedVariant> &alleles = it->second;
            std::string genotype, nref_all;
            std::set<std::string> all_genotype;
            int vc_f = 0, vc_r = 0;
            {
                const AlignedVariant &avar = *alleles.begin();
                if (candidateVariants.findVariant(avar.getStartHap() + int(leftPos), avar.getType(), avar.getString()) != NULL) was_candidate = 1;
                vc_f += varCoverage[PAV(it->first, avar)].nf;
                vc_r += varCoverage[PAV(it->first, avar)].nr;
            }
            std::string a1 = "*REF", a2 = "*REF";
            bool a1_ref = true, a2_ref = true;
            It ita1 = haps[hx1].indels.find(it->first), ita2 = haps[hx2].indels.find(it->first);
            if (ita1 != haps[hx1].indels.end() && !ita1->second.isRef()) { a1 = ita1->second.getString(); a1_ref = false; }
            if (ita2 != haps[hx2].indels.end() && !ita2->second.isRef()) { a2 = ita2->second.getString(); a2_ref = false; }
            all_genotype.insert(a1);
            all_genotype.insert(a2);
            if (a1_ref && a2_ref) throw std::string("genotyping error");
            if (a1 == a2) { genotype = "1/1"; nref_all = a1; }
            else if (a1_ref) { genotype = "0/1"; nref_all = a2; }
            else if (a2_ref) { genotype = "0/1"; nref_all = a1; }
            else {
                nref_all = a1 + ',' + a2;
                genotype = "1/2";
                const AlignedVariant &avar = *alleles.rbegin();
                if (candidateVariants.findVariant(avar.getStartHap() + int(leftPos), avar.getType(), avar.getString()) != NULL) was_candidate = 1;
                vc_f += varCoverage[PAV(it->first, avar)].nf;
                vc_r += varCoverage[PAV(it->first, avar)].nr;
            }
            double max_ll_altgeno = -HUGE_VAL;                                               // genotype quality (:3238-3265)
            static const std::string refAllele("*REF");
            std::vector<const std::string *> alleleOf(nh);                                   // each haplotype's allele at this position
            for (size_t h = 0; h < nh; h++) {
                It it2 = haps[h].indels.find(it->first);
                alleleOf[h] = (it2 == haps[h].indels.end() || it2->second.isRef()) ? &refAllele : &it2->second.getString();
            }
            for (size_t h1 = 0; h1 < nh; h1++) if (filtered[h1] == 0) for (size_t h2 = h1; h2 < nh; h2++) if (filtered[h2] == 0) {
                if (!((h1 == hx1 && h2 == hx2) || (h2 == hx1 && h1 == hx2))) {
                    // set<string>{x, y} != set<string>{a1, a2} of the reference
                    const std::string &x = *alleleOf[h1], &y = *alleleOf[h2];
                    const bool same = (x == a1 || x == a2) && (y == a1 || y == a2) && (a1 == x || a1 == y) && (a2 == x || a2 == y);
                    if (!same && max_ll_altgeno < pairs_posterior[h1 * nh + h2]) max_ll_altgeno = pairs_posterior[h1 * nh + h2];
                }
            }
            const double genoqual = -10.0 * (max_ll_altgeno - addLogs(max_ll_indel, max_ll_altgeno)) / log(10.0);
            DipMapCall c;
            c.index = index; c.tid = tid; c.leftPos = leftPos; c.rightPos = rightPos; c.candPos = candPos; c.realignedPos = it->first + int(leftPos);
            c.was_candidate = was_candidate; c.qual = qual; c.nref_all = nref_all; c.num_reads = numReadIdx; c.msq = msq; c.numf = numf; c.numr = numr;
            c.vc_f = vc_f; c.vc_r = vc_r; c.numUnmappedRealigned = numUnmappedRealigned; c.genotype = genotype; c.genoqual = genoqual;
            glfData.output(dipMapLine(glfData, c));
        }
    }

    clk.mark(5);
    std::vector<int> indelCountOf(nh * nr, -1);
    // ---- per variant position: genotype likelihoods over the haplotype pairs, coverage and QC sums (:3305-3660) ----
    for (PIt it = allVariantsByPos.begin(); it != allVariantsByPos.end(); ++it) {
        int has_variants_in_window = 0;
        for (std::set<PAV>::const_iterator pt = it->second.begin(); pt != it->second.end(); ++pt) {
            const AlignedVariant &avar = pt->second;
            if (candidateVariants.findVariant(avar.getStartHap() + int(leftPos), avar.getType(), avar.getString()) != NULL) { has_variants_in_window = 1; break; }
        }
        int nf = 0, nrv = 0;
        const int pos = it->first;
        const int posIdx = posToPosIdx[pos];
        double msq = 0.0;
        int n = 0;
        // the reference keys the genotype likelihoods by set<int>{v1, v2}; (smaller, larger) — a single-element set as (v, INT_MIN) —
        // sorts the same way, and the map is walked in that order below
        typedef std::pair<int, int> IntGenotype;
        std::map<IntGenotype, double> genLiks;
        std::map<std::pair<int, int>, double> pairPriorAt;
        double maxll = -HUGE_VAL;
        size_t hx1 = 0, hx2 = 0;
        for (size_t p = 0; p < nPairs; p++) {                                                // :3347-3389, the starting values
            const size_t h1 = pairH1[p], h2 = pairH2[p];
            const int v1 = hapVar[h1 * size_t(numVarPos) + size_t(posIdx)], v2 = hapVar[h2 * size_t(numVarPos) + size_t(posIdx)];
            double logPriorPos;                                                              // a function of (v1, v2) alone at this position
            std::map<std::pair<int, int>, double>::const_iterator known = pairPriorAt.find(std::make_pair(v1, v2));
            if (known != pairPriorAt.end()) logPriorPos = known->second;
            else {
                const AlignedVariant av1 = v1 ? variants[size_t(v1)].second : AlignedVariant("*REF", -1);
                const AlignedVariant av2 = v2 ? variants[size_t(v2)].second : AlignedVariant("*REF", -1);
                logPriorPos = pairPriorAt[std::make_pair(v1, v2)] = getPairPrior(av1, av2, int(leftPos), candidateVariants, params);
            }
            pairSum[p] = prior[h1 * nh + h2] - logPriorPos;            // the site's prior taken out again: a likelihood
        }
        addTermsOfAllReads(pairSum.data(), !keepTerms);
        for (size_t p = 0; p < nPairs; p++) {
            const size_t h1 = pairH1[p], h2 = pairH2[p];
            const int v1 = hapVar[h1 * size_t(numVarPos) + size_t(posIdx)], v2 = hapVar[h2 * size_t(numVarPos) + size_t(posIdx)];
            const IntGenotype genotype(std::min(v1, v2), v1 == v2 ? INT_MIN : std::max(v1, v2));
            const double ll = pairSum[p];
            std::map<IntGenotype, double>::iterator igit = genLiks.find(genotype);
            if (igit == genLiks.end()) genLiks[genotype] = ll; else igit->second = addLogs(igit->second, ll);
            if (ll > maxll) { maxll = ll; hx1 = h1; hx2 = h2; }
        }
        const WindowLikelihoods::Rows row1 = liks.rows(hx1), row2 = liks.rows(hx2);
        int numUnmappedRealigned = 0;                                                        // :3395-3402
        for (size_t r = 0; r < nr; r++)
            if (reads[r].isUnmapped() && (row1.offHap[r] == 0 || row2.offHap[r] == 0)) numUnmappedRealigned++;
        double allmsq = 0.0;                                                                 // :3492-3560
        int numMappedIndels = 0, nBQT = 0, nmmBQT = 0, nMMLeft = 0, nMMRight = 0, numOffBoth = 0;
        double mLogBQ = 0.0;
        std::vector<int> slotOf(nh, -2);                                                     // the position's slot in each haplotype's flag list, on first use
        for (size_t r = 0; r < numReadIdx; r++) {
            const double mq = mqOf[r];
            allmsq += (mq * mq);
            if (row1.offHap[r] && row2.offHap[r]) numOffBoth++;
            const bool first = row1.ll[r] >= row2.ll[r];
            const size_t h = first ? hx1 : hx2;                                              // the read's better haplotype of the pair
            const WindowLikelihoods::Rows &row = first ? row1 : row2;
            bool nrt = false, nft = false, covered = false;
            int &known = indelCountOf[h * nr + r];                                           // liks[h][r].indels.size(), :3529: the same
            if (known < 0) known = liks.indelCount(h, r);                                    // pair is asked for at every variant position
            numMappedIndels += known;
            nBQT += row.nBQT[r];
            nmmBQT += row.nmmBQT[r];
            mLogBQ += row.mLogBQ[r];
            if (row.nMMLeft[r] >= 2) nMMLeft++;
            if (row.nMMRight[r] >= 2) nMMRight++;
            if (slotOf[h] == -2) {                                                           // :3536-3541
                slotOf[h] = -1;
                It hit = haps[h].indels.find(pos);
                if (hit != haps[h].indels.end()) {
                    if (hit->second.isIndel()) slotOf[h] = liks.varSlot(h, pos, false);       // liks[h][r].hapIndelCovered[pos]
                    else if (hit->second.isSNP()) slotOf[h] = liks.varSlot(h, pos, true);    // liks[h][r].hapSNPCovered[pos]
                }
            }
            covered = row.covered(r, slotOf[h]);
            if (covered) {
                if (reads[r].onReverseStrand) nrt = true; else nft = true;
                const double mq2 = mqOf[r];
                msq += mq2 * mq2;
                n++;
            }
            if (nft) nf++;
            if (nrt) nrv++;
        }
        if (n != 0) msq = sqrt(msq / double(n)); else msq = 0.0;
        allmsq = (numReadIdx != 0) ? sqrt(allmsq / double(numReadIdx)) : 0;
        std::map<int, int> toVCFidx;                                                         // :3572-3595
        int nidx = 1;
        toVCFidx[0] = 0;
        static thread_local std::ostringstream oAlleles, oCovForward, oCovReverse, o;        // reused: see OutputData::Line::set
        oAlleles.str(std::string()); oCovForward.str(std::string()); oCovReverse.str(std::string()); o.str(std::string());
        int first = 1;
        for (size_t h = 0; h < nh; h++) {
            const int v = hapVar[h * size_t(numVarPos) + size_t(posIdx)];
            if (v != 0 && toVCFidx.find(v) == toVCFidx.end()) {
                toVCFidx[v] = nidx++;
                const std::string str = (first == 1) ? std::string("") : std::string(",");
                oAlleles << str << variants[size_t(v)].second.getString();
                oCovForward << str << varCoverage[variants[size_t(v)]].nf;
                oCovReverse << str << varCoverage[variants[size_t(v)]].nr;
                first = 0;
            }
        }
        // :3600-3613
        first = 1;
        for (std::map<IntGenotype, double>::iterator git = genLiks.begin(); git != genLiks.end(); ++git) {
            const int a1 = toVCFidx[git->first.first], a2 = toVCFidx[git->first.second == INT_MIN ? git->first.first : git->first.second];
            o << ((first == 1) ? "" : ",") << a1 << "/" << a2 << ":" << git->second;
            first = 0;
        }
        if (params.outputGLF) {
            DipPositionRow d;
            d.index = index; d.tid = tid; d.program = program; d.leftPos = leftPos; d.rightPos = rightPos; d.candPos = candPos; d.realignedPos = pos + int(leftPos);
            d.has_variants_in_window = has_variants_in_window; d.logZ = maxll; d.nBQT = nBQT; d.nmmBQT = nmmBQT; d.mLogBQ = mLogBQ; d.nMMLeft = nMMLeft;
            d.nMMRight = nMMRight; d.nref_all = oAlleles.str(); d.num_reads = numReadIdx; d.msq = allmsq; d.numOffAll = numOffBoth; d.num_indel = numMappedIndels;
            d.nf = nf; d.nr = nrv; d.var_coverage_forward = oCovForward.str(); d.var_coverage_reverse = oCovReverse.str(); d.glf = o.str();
            d.numUnmappedRealigned = numUnmappedRealigned;
            glfData.output(dipPositionLine(glfData, d));
        }
    }
    clk.mark(6);
}

} // namespace dindel
