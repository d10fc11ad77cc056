// faster_kernel.hip — the secondary "--faster" model of the path: ObservationModelS.
//
// Replaces, for a batch of windows, DetInDel::computeLikelihoodsFaster (reference DInDel.cpp:1790-1833): per
// (haplotype, read) pair HapHash k-mer voting for <= 15 candidate diagonals (Faster.cpp:131-189, Haplotype.hpp:315-384)
// and a Viterbi over S <= 16 relative-position states x {not inserted, inserted} run from both read ends to bMid
// (Faster.cpp:253-576), then mapState / hpos (:552-571, :579-681).  Quirks kept on purpose: the last k-mer of the
// haplotype is never hashed, non-ACGT hashes as 'A', `hp>=0 || hp<hlen` makes offHap / offHapHMQ always false, states
// right of the haplotype map to hap base hlen-1.
//
// One workgroup = one haplotype (its 4-mer keys live in registers of every wave), one wavefront = one pair; lane t < S
// is the "on diagonal t" state, lane S+t the "inserted at diagonal t" state.  The reference pushes candidates source by
// source; here every target pulls its candidates in the same source order, so the EPS=1e-7 hysteresis sees the same
// sequence.  No transcendental on the device (tables from dd_build_tables).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hmm_kernel.h"

namespace ddk {

#define FAST_EPS 1e-7
#define FNEG_INF (-__builtin_huge_val())

__device__ __forceinline__ int fast_map_char(unsigned char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }

__device__ __forceinline__ double bcast(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) dd_faster_kernel(const KernelArgs P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwav = blockDim.x >> 6;
    const double *T = P.tables;
    const double l1mE = T[TC_FAST + 0], lE = T[TC_FAST + 1], NIf = T[TC_FAST + 2], hqOn = T[TC_FAST + 3], hqOff = T[TC_FAST + 4];
    const double IIf = -0.25;
    // wave-private LDS: freq[F] ints, rd eq/uq [Lmax][2] doubles, read bytes [Lmax], bt [Lmax][32] bytes, st [Lmax] shorts
    unsigned char *wb = smem + (size_t)wave * P.lds_wave_bytes;
    int *freq = reinterpret_cast<int *>(wb + P.lds_off_A);
    double *rdE = reinterpret_cast<double *>(wb + P.lds_off_rdE);
    unsigned char *rdB = wb + P.lds_off_rdC;
    unsigned char *bt = wb + P.lds_off_bt;
    int16_t *st = reinterpret_cast<int16_t *>(wb + P.lds_off_ms);

    for (int item = P.item_begin + blockIdx.x; item < P.n_items; item += gridDim.x) {
        const int g = item / P.n_split, split = item - g * P.n_split;
        const int w = P.hap_window[g], h0 = P.win_hap_off[w];
        const int r0 = P.win_read_off[w], r1 = P.win_read_off[w + 1], R = r1 - r0;
        const int hs_off = P.hap_seq_off[g], hlen = P.hap_seq_off[g + 1] - hs_off;
        const uint32_t hapStart = P.win_hap_start[w];
        const char *hap = P.hap_seq + hs_off;
        const int64_t pair_base = P.win_pair_off[w] + (int64_t)(g - h0) * R;
        const int rs_base = P.read_seq_off[r0];
        const int64_t SL = (int64_t)P.read_seq_off[r1] - rs_base;
        const int64_t hpos_base = P.win_hpos_off[w] + (int64_t)(g - h0) * SL;
        const int nv = P.hap_var_off ? (P.hap_var_off[g + 1] - P.hap_var_off[g]) : 0;
        const bool hap_ok = P.maxLengthDel <= hlen;                 // maxLengthIndel (Faster.cpp:47)

        // 4-mer keys of the haplotype: positions lane, lane+64, ... (x < hlen-4: the last k-mer is not hashed)
        constexpr int HK = (DD_MAX_HAP_LEN + 63) / 64;
        int hkey[HK];
#pragma unroll
        for (int j = 0; j < HK; j++) {
            const int hx = lane + 64 * j;
            int key = -1;
            if (hx < hlen - 4) {
                key = 0;
                for (int y = 0; y < 4; y++) key |= fast_map_char((unsigned char)hap[hx + y]) << (2 * y);
            }
            hkey[j] = key;
        }

        for (int ri = split * nwav + wave; ri < R; ri += P.n_split * nwav) {
            const int rr = r0 + ri;
            const int64_t pair = pair_base + ri;
            const int so = P.read_seq_off[rr], L = P.read_seq_off[rr + 1] - so;
            if (!hap_ok || L < 4) {
                if (lane == 0) {
                    P.out.status[pair] = hap_ok ? DD_PAIR_NAN : DD_PAIR_HAPSIZE;
                    P.out.ll[pair] = 0.0;
                    if (P.out.offHapHMQ) P.out.offHapHMQ[pair] = 1;      // the reference throws here: never counts as on-haplotype
                }
                if (nv > 0) {
                    const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv;
                    for (int i = lane; i < nv; i += 64) {
                        if (P.out.var_covered) P.out.var_covered[vb + i] = 0;
                        if (P.out.var_fcov) P.out.var_fcov[vb + i] = 0;
                    }
                }
                continue;
            }
            // bMid — ObservationModelS::computeBMid (Faster.cpp:60-88)
            int bMid;
            {
                const uint32_t hapEnd = hapStart + (uint32_t)hlen, mReadStart = P.read_start[rr];
                const uint32_t readEnd = mReadStart + (uint32_t)L - 1u;
                if (mReadStart > hapEnd) bMid = 0;
                else if (readEnd < hapStart) bMid = L - 1;
                else {
                    const uint32_t olStart = (hapStart > mReadStart) ? hapStart : mReadStart;
                    const uint32_t olEnd = (hapEnd > readEnd) ? readEnd : hapEnd;
                    bMid = ((int)olEnd - (int)olStart) / 2 + (int)olStart - (int)mReadStart;
                }
                if (bMid < 0) bMid = 0;
                if (bMid >= L) bMid = L - 1;
            }
            // stage the read + clear the vote histogram (index rpfb + L, rpfb in [-(L-4), hlen-5])
            const int F = L + hlen;
            for (int b = lane; b < L; b += 64) {
                const int qi = P.read_qidx[so + b];
                rdB[b] = (unsigned char)P.read_seq[so + b];
                rdE[2 * b] = T[T_QUAL + 4 * qi];
                rdE[2 * b + 1] = T[T_QUAL + 4 * qi + 1];
            }
            for (int i = lane; i < F; i += 64) freq[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // AlignHash (:131-189): every read 4-mer votes for the diagonals of equal haplotype 4-mers
            {
                int key = 0;
                for (int y = 0; y < 3; y++) key |= fast_map_char(rdB[y]) << (2 * (y + 1));
                for (int x = 0; x <= L - 4; x++) {
                    key = (key >> 2) | (fast_map_char(rdB[x + 3]) << 6);          // HapHash::pushBack (Haplotype.hpp:349)
#pragma unroll
                    for (int j = 0; j < HK; j++) {
                        if (64 * j >= hlen - 4) break;
                        if (hkey[j] == key) atomicAdd(&freq[lane + 64 * j - x + L], 1);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // top 15 diagonals: frequency descending, ties by ascending relative position (:159-181)
            int myrel = 0x7fffffff;      // lane i < S holds candidate i (unsorted)
            int S = 0;
            for (int round = 0; round < 15; round++) {
                unsigned best = 0;
                for (int i = lane; i < F; i += 64) {
                    const int f = freq[i];
                    if (f > 0) {
                        const unsigned k2 = ((unsigned)f << 16) | (unsigned)(0xFFFF - i);
                        best = k2 > best ? k2 : best;
                    }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const unsigned o = __shfl_xor(best, off);
                    best = o > best ? o : best;
                }
                if (best == 0) break;
                const int idx = 0xFFFF - (int)(best & 0xFFFFu);
                if (lane == S) myrel = idx - L;
                if (lane == 0) freq[idx] = 0;
                S++;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (lane == S) myrel = -L;                   // relPos.push_back(-readLen) (:263)
            S = __builtin_amdgcn_readfirstlane(S + 1);
            // sort ascending (values are distinct): rank = number of smaller values
            int rank = 0;
            for (int i = 0; i < S; i++) {
                const int v = __builtin_amdgcn_readlane(myrel, i);
                rank += (lane < S && v < myrel) ? 1 : 0;
            }
            int *srt = freq;                              // reuse
            if (lane < S) srt[rank] = myrel;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int Tn = 2 * S;
            const bool isIns = lane >= S && lane < Tn;
            const bool isNo = lane < S;
            const int myd = isNo ? lane : (isIns ? lane - S : 0);     // diagonal index of this lane's state
            const int relMine = srt[myd];

            // ---------------- SStateHMM (:253-576) ----------------
            const int mqi = P.read_mqidx[rr];
            const double lOn = T[T_MAPQF + 2 * mqi], lOff = T[T_MAPQF + 2 * mqi + 1];
            double prev = 0.0;            // message of the neighbouring read base for this lane's state (0 at the read ends)
            auto obs_of = [&](int r, int rel) -> double {           // obs[r][s] (:286-296)
                const int hp = rel + r;
                if (hp >= 0 && hp < hlen) return ((char)rdB[r] == hap[hp]) ? rdE[2 * r] : rdE[2 * r + 1];
                return rdE[2 * r];
            };
            const bool act = lane < Tn;
            // from left to bMid (:373-416)
            for (int r = 0; r < bMid; r++) {
                const double LM = rdE[2 * r];
                const double pvMine = obs_of(r, relMine) + prev;                 // meaningful on non-inserted lanes
                const double prevIofMine = __shfl(prev, isNo ? lane + S : lane); // alpha[r-1][cs+S] seen from lane cs
                double cur = -1000.0;
                int bp = 0;
                for (int cs = 0; cs < S; cs++) {
                    const double pv = bcast(pvMine, cs);
                    const double pI = bcast(prevIofMine, cs);
                    const int relc = __builtin_amdgcn_readlane(relMine, cs);
                    const double d = fabs((double)(relc - relMine));
                    const double trI = (d - 1.0) * IIf;
                    if (isNo) {
                        if (cs <= lane) {
                            const double nvv = pv + ((cs != lane) ? trI + lE : l1mE);
                            if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs; }
                        } else if (relc - r >= relMine) {
                            const double nvv = ((LM + trI) + lE) + pI;
                            if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs + S; }
                        }
                    } else if (cs == myd) {
                        double nvv = pv + NIf;
                        if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs; }
                        nvv = (LM + IIf) + pI;
                        if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs + S; }
                    }
                }
                if (act) bt[r * 32 + lane] = (unsigned char)bp;
                prev = cur;
            }
            const double leftMsg = prev;      // alpha[bMid-1] (0 if bMid == 0)
            // from right to bMid (:422-466)
            prev = 0.0;
            for (int r = L - 1; r > bMid; r--) {
                const double LM = rdE[2 * r];
                const double obMine = obs_of(r, relMine);
                const double pvMine = obMine + prev;
                const double prevIofMine = __shfl(prev, isNo ? lane + S : lane);
                double cur = -1000.0;
                int bp = 0;
                for (int cs = 0; cs < S; cs++) {
                    const double pv = bcast(pvMine, cs);
                    const double pI = bcast(prevIofMine, cs);
                    const double ob = bcast(obMine, cs), pN = bcast(prev, cs);
                    const int relc = __builtin_amdgcn_readlane(relMine, cs);
                    const double d = fabs((double)(relc - relMine));
                    const double trI = (d - 1.0) * IIf;
                    if (isNo) {
                        if (lane <= cs) {
                            const double nvv = pv + ((cs != lane) ? trI + lE : l1mE);
                            if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs; }
                        }
                        if (cs == lane) {
                            const double nvv = (LM + lE) + pI;                  // (:436-438)
                            if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs + S; }
                        }
                    } else if (cs < myd) {
                        if (relc > relMine - r) {
                            const double nvv = ((ob + NIf) + trI) + pN;          // (:456-459)
                            if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs; }
                        }
                    } else if (cs == myd) {
                        const double nvv = (LM + IIf) + pI;
                        if (nvv > cur + FAST_EPS) { cur = nvv; bp = cs + S; }
                    }
                }
                if (act) bt[r * 32 + lane] = (unsigned char)bp;
                prev = cur;
            }
            const double rightMsg = prev;     // alpha[bMid+1] (0 if bMid == L-1)
            // join at bMid (:469-538): plain '>' maxima, state order x = ins*S + y
            double vR = FNEG_INF, vH = FNEG_INF;
            if (act) {
                const int hp = relMine + bMid;
                const bool on = hp >= 0 && hp < hlen;
                const double pins = isNo ? l1mE : lE;
                const double obsv = isNo ? obs_of(bMid, relMine) : rdE[2 * bMid];
                double a = obsv + ((on ? lOn : lOff) + pins);
                if (bMid < L - 1) a += rightMsg;
                if (bMid > 0) a += leftMsg;
                vR = a;
                double hv = obsv + ((on ? hqOn : hqOff) + pins);
                if (bMid < L - 1) hv += rightMsg;
                if (bMid > 0) hv += leftMsg;
                vH = hv;
            }
            double ll = FNEG_INF;
            int xH = 0;
            {
                double mh = FNEG_INF;
                for (int x = 0; x < Tn; x++) {                    // sequential '>' scan = first maximum
                    const double a = bcast(vR, x), hv = bcast(vH, x);
                    if (a > ll) ll = a;
                    if (hv > mh) { mh = hv; xH = x; }
                }
            }
            // backtrack (:540-548) and mapState (:552-571), wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                int s = xH;
                if (lane == 0) st[bMid] = (int16_t)s;
                for (int b = bMid; b > 0; b--) { s = bt[(b - 1) * 32 + s]; if (lane == 0) st[b - 1] = (int16_t)s; }
                s = xH;
                for (int b = bMid; b < L - 1; b++) { s = bt[(b + 1) * 32 + s]; if (lane == 0) st[b + 1] = (int16_t)s; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                int lhp = 1;
                for (int r = 0; r < L; r++) {
                    const int s = st[r];
                    int m;
                    if (s < S) {
                        const int hp = srt[s] + r;
                        if (hp >= 0 && hp < hlen) { m = hp + 1; lhp = hp + 1; }
                        else if (hp < 0) m = 0; else m = hlen;
                    } else m = hlen + 2 + lhp;
                    if (lane == 0) st[r] = (int16_t)m;             // st now holds mapState
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // reportVariants (:579-681): hpos, firstBase / lastBase
            const int numS = hlen + 2;
            int firstB = 0x7fffffff, lastB = -1;
            int16_t *hp_out = P.out.hpos ? P.out.hpos + hpos_base + (so - rs_base) : nullptr;
            for (int b = lane; b < L; b += 64) {
                const int s = st[b];
                const int xm = s % numS;
                int hp;
                if (xm > 0 && xm <= hlen) {
                    if (s >= numS) hp = DD_HPOS_INS;
                    else { hp = s - 1; firstB = hp < firstB ? hp : firstB; lastB = hp > lastB ? hp : lastB; }
                } else hp = (xm == 0) ? DD_HPOS_LO : DD_HPOS_RO;
                if (hp_out) hp_out[b] = (int16_t)hp;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int f = __shfl_xor(firstB, off), l2 = __shfl_xor(lastB, off);
                firstB = f < firstB ? f : firstB;
                lastB = l2 > lastB ? l2 : lastB;
            }
            if (firstB == 0x7fffffff) firstB = -1;
            if (P.out.var_covered && nv > 0) {
                const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv;
                for (int i = lane; i < nv; i += 64) {
                    const int sR = P.hap_var[2 * (P.hap_var_off[g] + i)], eR = P.hap_var[2 * (P.hap_var_off[g] + i) + 1];
                    P.out.var_covered[vb + i] = (firstB + P.padCover <= sR && lastB - P.padCover >= eR) ? 1 : 0;
                }
            }
            // DetInDel::filterHaplotypes' per-read test (DInDel.cpp:1951-2054): this model leaves numIndels = 0 and
            // offHapHMQ = false, so every read is selected, and its hpos may skip or repeat haplotype bases: the
            // covered set is marked base by base in LDS.
            if (P.out.var_fcov && P.hap_var_flank && nv > 0) {
                const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv;
                for (int i = 0; i < nv; i++) {
                    const int32_t *fl = P.hap_var_flank + 3 * (size_t)(P.hap_var_off[g] + i);
                    const int left = fl[0] - P.padCover, right = fl[1] + P.padCover, kind = fl[2];
                    int cov = 0;
                    if (kind != 0 && right >= left) {
                        for (int x = lane; x < hlen; x += 64) freq[x] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        int nmm = 0;
                        bool anyIns = false;
                        for (int b0 = 0; b0 < L; b0 += 64) {
                            const int b = b0 + lane;
                            bool mm = false;
                            if (b < L) {
                                const int s2 = st[b];
                                anyIns = anyIns || s2 >= numS;
                                if (s2 >= 1 && s2 <= hlen) {
                                    const int hb = s2 - 1;
                                    if (hb >= left && hb <= right) {
                                        freq[hb] = 1;
                                        const char hc = hap[hb];
                                        mm = (char)rdB[b] != hc && (kind == 2 || hc != 'N');      // 'N' exempt for DEL (:1992)
                                    }
                                }
                            }
                            nmm += __popcll(__ballot(mm));
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        int csize = 0;
                        const int lo = left > 0 ? left : 0, hi = right < hlen - 1 ? right : hlen - 1;
                        for (int x0 = lo; x0 <= hi; x0 += 64) {
                            const int x = x0 + lane;
                            csize += __popcll(__ballot(x <= hi && freq[x] != 0));
                        }
                        // hpos of an inserted base is the sentinel -1, which the reference's set also collects (:1989-1991)
                        if (left == DD_HPOS_INS && __ballot(anyIns) != 0) csize++;
                        cov = (csize >= right - left + 1 && nmm <= P.maxMismatch) ? 1 : 0;
                    }
                    if (lane == 0) P.out.var_fcov[vb + i] = (uint8_t)cov;
                }
            }
            if (lane == 0) {
                P.out.ll[pair] = ll;
                P.out.status[pair] = DD_PAIR_OK;             // computeLikelihoodsFaster has no ll checks
                if (P.out.llOn) P.out.llOn[pair] = 0.0;
                if (P.out.llOff) P.out.llOff[pair] = 0.0;
                if (P.out.mLogBQ) P.out.mLogBQ[pair] = 0.0;
                if (P.out.offHap) P.out.offHap[pair] = 0;       // always false (:491)
                if (P.out.offHapHMQ) P.out.offHapHMQ[pair] = 0; // always false (:529)
                if (P.out.numIndels) P.out.numIndels[pair] = 0;
                if (P.out.numMismatch) P.out.numMismatch[pair] = 0;
                if (P.out.nBQT) P.out.nBQT[pair] = 0;
                if (P.out.nmmBQT) P.out.nmmBQT[pair] = 0;
                if (P.out.nMMLeft) P.out.nMMLeft[pair] = 0;
                if (P.out.nMMRight) P.out.nMMRight[pair] = 0;
                if (P.out.firstBase) P.out.firstBase[pair] = (int16_t)firstB;
                if (P.out.lastBase) P.out.lastBase[pair] = (int16_t)lastB;
            }
        }
    }
}

hipError_t launch_faster(const KernelArgs &A, unsigned grid, int waves, size_t lds, hipStream_t st)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dd_faster_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dd_faster_kernel, dim3(grid), dim3(waves * 64), lds, st, A);
    return hipGetLastError();
}

} // namespace ddk
