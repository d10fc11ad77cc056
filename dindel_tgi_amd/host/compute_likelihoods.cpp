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
    // ---- pack (CSR) ----
    std::vector<int32_t> win_hap_off(1, 0), win_read_off(1, 0), hap_seq_off(1, 0), hap_var_off(1, 0), hap_var, hap_var_flank, read_seq_off(1, 0);
    std::vector<uint32_t> win_hap_start, read_start;
    std::string hap_seq, read_seq;
    std::vector<uint8_t> read_qidx, read_mqidx, read_flags, read_lib;
    std::vector<int32_t> read_mate_pos, read_mate_len, lib_off(1, 0);
    std::vector<double> lib_prob, lib_p95;
    std::map<const Library *, int> lib_index;
    const bool with_mates = params.mapUnmappedReads && !faster;
    std::map<double, int> qmap, mqmap;
    std::vector<double> qtab, mqtab;
    for (int w = 0; w < W; w++) {
        WindowJob &J = jobs[w];
        J.error.clear();
        win_hap_start.push_back(J.leftPos);
        for (size_t h = 0; h < J.haps->size(); h++) {
            const Haplotype &H = (*J.haps)[h];
            hap_seq += H.seq;
            hap_seq_off.push_back(int32_t(hap_seq.size()));
            for (std::map<int, AlignedVariant>::const_iterator it = H.indels.begin(); it != H.indels.end(); ++it) {
                hap_var.push_back(it->second.getStartRead()); hap_var.push_back(it->second.getEndRead());
                hap_var_flank.push_back(it->second.getLeftFlankRead()); hap_var_flank.push_back(it->second.getRightFlankRead());
                hap_var_flank.push_back(it->second.getType() == AlignedVariant::DEL ? 1 : it->second.getType() == AlignedVariant::INS ? 2 : 0);
            }
            for (std::map<int, AlignedVariant>::const_iterator it = H.snps.begin(); it != H.snps.end(); ++it) {
                hap_var.push_back(it->second.getStartRead()); hap_var.push_back(it->second.getEndRead());
                hap_var_flank.push_back(0); hap_var_flank.push_back(0); hap_var_flank.push_back(0);
            }
            hap_var_off.push_back(int32_t(hap_var.size() / 2));
        }
        win_hap_off.push_back(win_hap_off.back() + int32_t(J.haps->size()));
        for (size_t r = 0; r < J.reads->size(); r++) {
            const Read &R = (*J.reads)[r];
            if (R.qual.size() != R.size()) throw std::string("Read: qual and seq differ in length");
            read_seq += R.seq.seq;
            read_seq_off.push_back(int32_t(read_seq.size()));
            double lastq = -1.0;
            int lastidx = 0;
            for (size_t b = 0; b < R.qual.size(); b++) {
                if (R.qual[b] != lastq) {                 // runs of equal qualities skip the map
                    std::map<double, int>::iterator it = qmap.find(R.qual[b]);
                    if (it == qmap.end()) { it = qmap.insert(std::make_pair(R.qual[b], int(qtab.size()))).first; qtab.push_back(R.qual[b]); }
                    if (it->second > 255) throw std::string("more than 256 distinct base qualities in one batch");
                    lastq = R.qual[b]; lastidx = it->second;
                }
                read_qidx.push_back(uint8_t(lastidx));
            }
            std::map<double, int>::iterator it = mqmap.find(R.mapQual);
            if (it == mqmap.end()) { it = mqmap.insert(std::make_pair(R.mapQual, int(mqtab.size()))).first; mqtab.push_back(R.mapQual); }
            if (it->second > 255) throw std::string("more than 256 distinct mapping qualities in one batch");
            read_mqidx.push_back(uint8_t(it->second));
            read_start.push_back(uint32_t(R.posStat.first));      // uint32_t(read.posStat.first), ObservationModelFB.cpp:52
            read_flags.push_back(uint8_t((R.isUnmapped() ? DD_READ_UNMAPPED : 0) | (R.isPaired() ? DD_READ_PAIRED : 0) |
                                         (R.mateIsUnmapped() ? DD_READ_MATE_UNMAPPED : 0) | (R.mateIsReverse() ? DD_READ_MATE_REVERSE : 0) |
                                         (R.mateSameTid ? DD_READ_MATE_SAME_TID : 0)));
            if (with_mates) {
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
                read_lib.push_back(uint8_t(li));
                read_mate_pos.push_back(R.matePos);
                read_mate_len.push_back(R.isPaired() ? R.mateLen : -1);
            }
        }
        win_read_off.push_back(win_read_off.back() + int32_t(J.reads->size()));
    }
    dd_batch B;
    memset(&B, 0, sizeof(B));
    B.n_windows = W;
    B.win_hap_off = win_hap_off.data(); B.win_read_off = win_read_off.data(); B.win_hap_start = win_hap_start.data();
    B.hap_seq_off = hap_seq_off.data(); B.hap_seq = hap_seq.data(); B.hap_var_off = hap_var_off.data();
    B.hap_var = hap_var.empty() ? NULL : hap_var.data();
    B.hap_var_flank = hap_var_flank.empty() ? NULL : hap_var_flank.data();
    B.read_seq_off = read_seq_off.data(); B.read_seq = read_seq.data(); B.read_qidx = read_qidx.data();
    B.read_mqidx = read_mqidx.data(); B.read_start = read_start.data(); B.read_flags = read_flags.data();
    B.n_qual = int(qtab.size()); B.qual_table = qtab.data(); B.n_mapq = int(mqtab.size()); B.mapq_table = mqtab.data();
    if (with_mates) {
        if (lib_p95.empty()) { lib_prob.push_back(1.0); lib_off.push_back(1); lib_p95.push_back(1.0); }   // no paired read: placeholder table
        B.read_mate_pos = read_mate_pos.data(); B.read_mate_len = read_mate_len.data(); B.read_lib = read_lib.data();
        B.n_libs = int(lib_p95.size()); B.lib_off = lib_off.data(); B.lib_prob = lib_prob.data(); B.lib_p95 = lib_p95.data();
    }

    dd_sizes sz;
    if (dd_batch_sizes(&B, &sz) != DD_SUCCESS) throw std::string(dd_last_error());
    std::vector<int64_t> pair_off(W + 1), hpos_off(W + 1), vc_off(W + 1);
    dd_batch_offsets(&B, pair_off.data(), hpos_off.data(), vc_off.data());

    std::vector<double> ll(sz.n_pairs), llOn(sz.n_pairs), llOff(sz.n_pairs), mLogBQ(sz.n_pairs);
    std::vector<uint8_t> offHap(sz.n_pairs), offHapHMQ(sz.n_pairs), onHapV(sz.n_reads ? sz.n_reads : 1);
    std::vector<int16_t> hpos(sz.hpos_len ? sz.hpos_len : 1);
    std::vector<int16_t> numIndels(sz.n_pairs), numMismatch(sz.n_pairs), nBQT(sz.n_pairs), nmmBQT(sz.n_pairs), nMMLeft(sz.n_pairs),
        nMMRight(sz.n_pairs), firstBase(sz.n_pairs), lastBase(sz.n_pairs);
    std::vector<int32_t> status(sz.n_pairs);
    std::vector<uint8_t> fcov(sz.var_cov_len ? sz.var_cov_len : 1), vcov(sz.var_cov_len ? sz.var_cov_len : 1);
    dd_result Rz;
    memset(&Rz, 0, sizeof(Rz));
    Rz.ll = ll.data(); Rz.llOn = llOn.data(); Rz.llOff = llOff.data(); Rz.mLogBQ = mLogBQ.data();
    Rz.offHap = offHap.data(); Rz.offHapHMQ = offHapHMQ.data(); Rz.hpos = hpos.data(); Rz.status = status.data();
    Rz.numIndels = numIndels.data(); Rz.numMismatch = numMismatch.data(); Rz.nBQT = nBQT.data(); Rz.nmmBQT = nmmBQT.data();
    Rz.nMMLeft = nMMLeft.data(); Rz.nMMRight = nMMRight.data(); Rz.firstBase = firstBase.data(); Rz.lastBase = lastBase.data();
    Rz.onHap = onHapV.data();
    Rz.var_fcov = fcov.data(); Rz.var_covered = vcov.data();
    dd_params P = to_abi(params);
    if (faster) P.mapUnmappedReads = 0;                 // ObservationModelS has no insert-size prior
    const std::chrono::steady_clock::time_point t_packed = std::chrono::steady_clock::now();
    if (sz.n_pairs > 0) {
        const int rc = faster ? dd_compute_likelihoods_faster(&P, &B, &Rz, device_) : dd_compute_likelihoods(&P, &B, &Rz, device_);
        if (rc != DD_SUCCESS) throw std::string(faster ? "dd_compute_likelihoods_faster: " : "dd_compute_likelihoods: ") + dd_last_error();
    }

    const std::chrono::steady_clock::time_point t_device = std::chrono::steady_clock::now();
    // ---- unpack into liks[hidx][r] / onHap: windows are independent, so a few host threads share them ----
    // Per pair the device already delivers every counter of the record; the variant maps and the `align` string need the
    // per-base walk (rebuildAlignment) only when the read shows an indel or differs from the haplotype segment it sits on.
    auto unpack_window = [&](int w) {
        WindowJob &J = jobs[w];
        const size_t H = J.haps->size(), Rn = J.reads->size();
        *J.onHap = std::vector<int>(Rn, 0);                                              // DInDel.cpp:1710
        *J.liks = std::vector<std::vector<MLAlignment> >(H, std::vector<MLAlignment>(Rn)); // DInDel.cpp:1714
        const int r0 = win_read_off[w];
        const int64_t SL = int64_t(read_seq_off[win_read_off[w + 1]]) - read_seq_off[r0];
        for (size_t h = 0; h < H && J.error.empty(); h++) {
            const Haplotype &Hh = (*J.haps)[h];
            const int Hs = int(Hh.size());
            const int g = win_hap_off[w] + int(h);
            const int nv = hap_var_off[g + 1] - hap_var_off[g];
            for (size_t r = 0; r < Rn; r++) {
                const int64_t p = pair_off[w] + int64_t(h) * int64_t(Rn) + int64_t(r);
                if (status[p] == DD_PAIR_HAPSIZE) { J.error = "hapSize error."; break; }   // ObservationModelFB.cpp:47, Faster.cpp:47
                if (status[p] == DD_PAIR_UNSUPPORTED) {      // this window only; the caller skips it like a window that threw (DInDel.cpp:1369-1374)
                    J.error = "window outside the GPU kernel limits (haplotype > 766 bp, read > 1024 bp or an empty sequence)";
                    break;
                }
                if (faster && status[p] != DD_PAIR_OK) { J.error = "HapHash string too short"; break; }   // Haplotype.hpp:341
                MLAlignment &ml = (*J.liks)[h][r];
                const Read &Rd = (*J.reads)[r];
                const int L = int(Rd.size());
                const int16_t *hp = hpos.data() + hpos_off[w] + int64_t(h) * SL + (read_seq_off[r0 + r] - read_seq_off[r0]);
                const int64_t vb = vc_off[w] + int64_t(hap_var_off[g] - hap_var_off[win_hap_off[w]]) * int64_t(Rn) + int64_t(r) * nv;
                bool plain = false;                       // gap-free, mismatch-free placement: nothing to list
                if (!faster && numIndels[p] == 0) {
                    if (firstBase[p] < 0) plain = true;   // no base on the haplotype
                    else {
                        int b0 = 0;
                        while (b0 < L && hp[b0] < 0) b0++;
                        const int n = lastBase[p] - firstBase[p] + 1;
                        plain = b0 + n <= L && hp[b0] == firstBase[p] &&
                                memcmp(Rd.seq.seq.data() + b0, Hh.seq.data() + firstBase[p], size_t(n)) == 0;
                    }
                }
                if (plain) {
                    ml.align = std::string(size_t(Hs), 'R');
                    ml.hpos.assign(hp, hp + L);
                    ml.firstBase = firstBase[p]; ml.lastBase = lastBase[p];
                    ml.numIndels = 0; ml.numMismatch = numMismatch[p]; ml.nBQT = nBQT[p]; ml.nmmBQT = nmmBQT[p];
                    ml.nMMLeft = nMMLeft[p]; ml.nMMRight = nMMRight[p];
                    int i = 0;
                    for (std::map<int, AlignedVariant>::const_iterator it = Hh.indels.begin(); it != Hh.indels.end(); ++it, ++i)
                        ml.hapIndelCovered[it->first] = vcov[vb + i] != 0;
                    for (std::map<int, AlignedVariant>::const_iterator it = Hh.snps.begin(); it != Hh.snps.end(); ++it, ++i)
                        ml.hapSNPCovered[it->first] = vcov[vb + i] != 0;
                } else if (faster) rebuildAlignmentFaster(Hh, Rd, hp, params, ml);
                else rebuildAlignment(Hh, Rd, hp, params, ml);
                ml.ll = ll[p]; ml.llOn = llOn[p]; ml.llOff = llOff[p];
                ml.offHap = offHap[p] != 0; ml.offHapHMQ = offHapHMQ[p] != 0;
                if (!faster) ml.mLogBQ = mLogBQ[p];           // the device's serial sum (same order as the reference)
                {   // per haplotype-indel coverage flags of filterHaplotypes, in hap.indels map order (first nvI of the hap's list)
                    int i = 0;
                    for (std::map<int, AlignedVariant>::const_iterator it = Hh.indels.begin(); it != Hh.indels.end(); ++it, ++i)
                        ml.hapIndelFilterCovered[it->first] = fcov[vb + i] != 0;
                }
                if (!ml.offHapHMQ) (*J.onHap)[r] = 1;                                     // DInDel.cpp:1720
                if (faster) continue;                                                     // computeLikelihoodsFaster has no ll checks
                if (status[p] == DD_PAIR_LLPOS) {                                         // DInDel.cpp:1722-1731
                    if (throwOnPositive_) { J.error = "Likelihood>0"; break; }
                    std::cout << "hidx: " << h << " r: " << r << std::endl;
                    std::cerr << "Likelihood>0" << std::endl;
                    exit(1);
                }
                if (status[p] == DD_PAIR_NAN) {                                           // DInDel.cpp:1732-1735
                    std::cout << "NAN/Inf error" << std::endl;
                    J.error = "Nan detected";
                    break;
                }
            }
        }
    };
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr > 16) nthr = 16;
    if (nthr < 1) nthr = 1;
    if (hostThreads_ > 0) nthr = unsigned(hostThreads_);
    if (nthr > unsigned(W)) nthr = unsigned(W > 0 ? W : 1);
    if (nthr <= 1) {
        for (int w = 0; w < W; w++) unpack_window(w);
    } else {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthr; t++)
            pool.push_back(std::thread([&]() { for (int w = next++; w < W; w = next++) unpack_window(w); }));
        for (size_t t = 0; t < pool.size(); t++) pool[t].join();
    }
    const std::chrono::steady_clock::time_point t_end = std::chrono::steady_clock::now();
    lastPackSeconds = std::chrono::duration<double>(t_packed - t_start).count();
    lastDeviceSeconds = std::chrono::duration<double>(t_device - t_packed).count();
    lastUnpackSeconds = std::chrono::duration<double>(t_end - t_device).count();
}

} // namespace dindel
