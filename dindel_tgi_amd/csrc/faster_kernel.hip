// faster_kernel.hip — the secondary "--faster" model of the path: ObservationModelS.
//
// Replaces, for a batch of windows, DetInDel::computeLikelihoodsFaster (reference DInDel.cpp:1790-1833): per
// (haplotype, read) pair HapHash k-mer voting for <= 15 candidate diagonals (Faster.cpp:131-189, Haplotype.hpp:315-384)
// and a Viterbi over S <= 16 relative-position states x {not inserted, inserted} run from both read ends to bMid
// (Faster.cpp:253-576), then mapState / hpos (:552-571, :579-681).  Quirks kept on purpose: the last k-mer of the
// haplotype is never hashed, non-ACGT hashes as 'A', `hp>=0 || hp<hlen` makes offHap / offHapHMQ always false, states
// right of the haplotype map to hap base hlen-1.
//
// Mapping: one workgroup = one haplotype (its bytes and its 4-mer bucket index sit in block-shared LDS), one wavefront =
// FOUR pairs at a time: 16 lanes per pair, lane d owns diagonal d — both its "on diagonal" and its "inserted at diagonal"
// state.  Per read base every lane publishes {value, inserted value} of its diagonal in a 16-byte LDS slot and pulls the
// <= 16 slots of its pair in source order; the reference pushes candidates source by source, so pulling in the same
// order shows the EPS = 1e-7 hysteresis the same sequence per target.  Transition terms per (source, own diagonal) are
// loop constants in registers.  No transcendental on the device (tables from dd_build_tables).
//
// Per-pair LDS ("pair area", KernelArgs::lds_wave_bytes bytes each, 4 per wavefront):
//   freq  int[(L+hlen+1)/2]  vote histogram, two 16-bit bins per word; after the selection the same bytes hold
//                            st i16[L]: state path, then mapState
//   rd    u16[L]             read base | quality index << 8
//   bt    u8[L][16]          back-pointers of diagonal d's two states in one byte (encoding at bt_left / bt_right)
//   bc    double2[16]        the per-base exchange slots (later: coverage bitmap);  srt int[16]: sorted relative positions
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <type_traits>
#include "hmm_kernel.h"

namespace ddk {

#define FAST_EPS 1e-7
#define FAST_CHUNK 256   /* reads ordered at a time (capi.cpp sizes the LDS for it) */
#define FNEG_INF (-__builtin_huge_val())

__device__ __forceinline__ int fast_map_char(unsigned c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }

// LDS traffic between lanes of one wavefront: DS operations of a wave execute in order; this only pins the compiler.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform maximum of a value that is uniform inside each 16-lane group
__device__ __forceinline__ int gmax4(int v)
{
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// bMid — ObservationModelS::computeBMid (Faster.cpp:60-88)
__device__ __forceinline__ int fast_bmid(uint32_t hapStart, int hlen, uint32_t mReadStart, int L)
{
    const uint32_t hapEnd = hapStart + (uint32_t)hlen;
    const uint32_t readEnd = mReadStart + (uint32_t)L - 1u;
    int bMid;
    if (mReadStart > hapEnd) bMid = 0;
    else if (readEnd < hapStart) bMid = L - 1;
    else {
        const uint32_t olStart = (hapStart > mReadStart) ? hapStart : mReadStart;
        const uint32_t olEnd = (hapEnd > readEnd) ? readEnd : hapEnd;
        bMid = ((int)olEnd - (int)olStart) / 2 + (int)olStart - (int)mReadStart;
    }
    if (bMid < 0) bMid = 0;
    if (bMid >= L) bMid = L - 1;
    return bMid;
}

// `if (nv > cur + EPS) { cur = nv; bp = code; }` (Faster.cpp:383 and every other update of the model)
#define FOLD(cur, bp, nvv, code, ok)                         \
    do {                                                     \
        const double nv__ = (nvv);                           \
        const bool t__ = (ok) && nv__ > (cur) + FAST_EPS;    \
        (cur) = t__ ? nv__ : (cur);                          \
        (bp) = t__ ? (code) : (bp);                          \
    } while (0)

// 2 waves per SIMD: the 16-source loops want ~200 VGPRs; one more resident wave costs spills inside them, one fewer
// leaves the LDS round trips of a read base exposed (measured 277 -> 160 ms at 4000 windows going from 1 to 2).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) dd_faster_kernel(const KernelArgs P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwav = blockDim.x >> 6;
    const double *T = P.tables;
    const double l1mE = T[TC_FAST + 0], lE = T[TC_FAST + 1], NIf = T[TC_FAST + 2], hqOn = T[TC_FAST + 3], hqOff = T[TC_FAST + 4];
    const double IIf = -0.25;
    // block-shared
    double2 *qt = reinterpret_cast<double2 *>(smem + P.lds_off_E);          // {log match, log mismatch} per quality index
    unsigned char *shHap = smem + P.lds_off_N;
    uint16_t *bk = reinterpret_cast<uint16_t *>(smem + P.lds_off_Q);        // bucket starts [257]
    uint16_t *hpl = reinterpret_cast<uint16_t *>(smem + P.lds_off_C);       // haplotype positions grouped by 4-mer
    int *cnt = reinterpret_cast<int *>(smem + P.lds_off_Y);                 // [256] build scratch
    uint32_t *skey = reinterpret_cast<uint32_t *>(smem + P.lds_off_rdE);    // [FAST_CHUNK] sort keys of a chunk of reads
    uint16_t *sord = reinterpret_cast<uint16_t *>(skey + FAST_CHUNK);       // [FAST_CHUNK] chunk-local read index by rank
    // pair area of this 16-lane group
    const int ngrp = P.fast_groups;                                          // 4 unless a pair area is too large for LDS
    const int nar = ngrp < 4 ? ngrp + 1 : 4;                                 // idle groups share a dummy area (index ngrp)
    unsigned char *pa = smem + P.lds_shared_bytes + (size_t)(wave * nar + (grp < ngrp ? grp : ngrp)) * P.lds_wave_bytes;
    int *freq = reinterpret_cast<int *>(pa + P.lds_off_A);
    uint16_t *rd = reinterpret_cast<uint16_t *>(pa + P.lds_off_rdC);
    unsigned char *bt = pa + P.lds_off_bt;
    int16_t *st = reinterpret_cast<int16_t *>(pa + P.lds_off_A);     // aliases freq: the histogram is dead once the diagonals are chosen
    double2 *bc = reinterpret_cast<double2 *>(pa + P.lds_off_I);
    int *srt = reinterpret_cast<int *>(pa + P.lds_off_rdQ);
    int *bm = reinterpret_cast<int *>(pa + P.lds_off_I);               // coverage bitmap (<= 24 words) over the idle exchange slots

    for (int i = tid; i < P.n_qual; i += blockDim.x) qt[i] = make_double2(T[T_QUAL + 4 * i], T[T_QUAL + 4 * i + 1]);

    for (int item = P.item_begin + xcd_contiguous_block_id(P.n_items - P.item_begin); item < P.n_items; item += gridDim.x) {
        const int g = item / P.n_split, split = item - g * P.n_split;
        const int w = P.hap_window[g], h0 = P.win_hap_off[w];
        const int r0 = P.win_read_off[w], r1 = P.win_read_off[w + 1], R = r1 - r0;
        if (P.win_skip && P.win_skip[w]) {
            // window outside the kernel limits (dd_screen_windows): its pairs are only marked, nothing of it is read
            const int64_t pb = P.win_pair_off[w] + (int64_t)(g - h0) * R;
            for (int ri = split * (int)blockDim.x + tid; ri < R; ri += P.n_split * (int)blockDim.x) {
                P.out.status[pb + ri] = DD_PAIR_UNSUPPORTED;
                P.out.ll[pb + ri] = 0.0;
                if (P.out.offHap) P.out.offHap[pb + ri] = 1;
                if (P.out.offHapHMQ) P.out.offHapHMQ[pb + ri] = 1;
            }
            continue;
        }
        const int hs_off = P.hap_seq_off[g], hlen = P.hap_seq_off[g + 1] - hs_off;
        const uint32_t hapStart = P.win_hap_start[w];
        const char *hap = P.hap_seq + hs_off;
        const int64_t pair_base = P.win_pair_off[w] + (int64_t)(g - h0) * R;
        const int rs_base = P.read_seq_off[r0];
        const int64_t SL = (int64_t)P.read_seq_off[r1] - rs_base;
        const int64_t hpos_base = P.win_hpos_off[w] + (int64_t)(g - h0) * SL;
        const int nv = P.hap_var_off ? (P.hap_var_off[g + 1] - P.hap_var_off[g]) : 0;
        const bool hap_ok = P.maxLengthDel <= hlen;                 // maxLengthIndel (Faster.cpp:47)
        const int numS = hlen + 2;

        // ---- HapHash (Haplotype.hpp:378-381): positions x < hlen-4 bucketed by their 4-mer ----
        __syncthreads();                                            // the previous item's readers are done
        for (int i = tid; i < hlen; i += blockDim.x) shHap[i] = (unsigned char)hap[i];
        for (int i = tid; i < 256; i += blockDim.x) cnt[i] = 0;
        __syncthreads();
        for (int hx = tid; hx < hlen - 4; hx += blockDim.x) {
            int key = 0;
            for (int y = 0; y < 4; y++) key |= fast_map_char(shHap[hx + y]) << (2 * y);
            atomicAdd(&cnt[key], 1);
        }
        __syncthreads();
        if (wave == 0) {                                            // exclusive prefix over the 256 buckets
            int c4[4], sum = 0;
            for (int j = 0; j < 4; j++) { c4[j] = cnt[4 * lane + j]; sum += c4[j]; }
            int incl = sum;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            int run = incl - sum;
            for (int j = 0; j < 4; j++) { bk[4 * lane + j] = (uint16_t)run; cnt[4 * lane + j] = run; run += c4[j]; }
            if (lane == 63) bk[256] = (uint16_t)run;
        }
        __syncthreads();
        for (int hx = tid; hx < hlen - 4; hx += blockDim.x) {
            int key = 0;
            for (int y = 0; y < 4; y++) key |= fast_map_char(shHap[hx + y]) << (2 * y);
            hpl[atomicAdd(&cnt[key], 1)] = (uint16_t)hx;
        }
        __syncthreads();

        // The four pairs of a wavefront advance base by base together, so they should need the same number of bases on
        // either side of bMid: order each chunk of the window's reads by (bMid, bases right of it) and hand consecutive
        // ranks to a wavefront.  (Results do not depend on the grouping; synthetic windows: 126 -> 100 rows per four pairs.)
        for (int cb = 0; cb < R; cb += FAST_CHUNK) {
        const int nch = (R - cb < FAST_CHUNK) ? R - cb : FAST_CHUNK;
        __syncthreads();
        for (int t = tid; t < nch; t += blockDim.x) {
            const int rr = r0 + cb + t;
            const int L = P.read_seq_off[rr + 1] - P.read_seq_off[rr];
            skey[t] = ((uint32_t)fast_bmid(hapStart, hlen, P.read_start[rr], L) << 11) | (uint32_t)(L > 0 ? ((L - 1) & 2047) : 0);
            // Outputs this model leaves at MLAlignment's defaults, written here coalesced over the chunk's pairs instead of
            // pair by pair from the group leaders (which go in bMid order).  A pair the reference throws for is finished here.
            const int64_t pair = pair_base + cb + t;
            const bool good = hap_ok && L >= 4;
            if (P.out.llOn) P.out.llOn[pair] = 0.0;
            if (P.out.llOff) P.out.llOff[pair] = 0.0;
            if (P.out.mLogBQ) P.out.mLogBQ[pair] = 0.0;
            if (P.out.offHap) P.out.offHap[pair] = 0;                          // always false (:491)
            if (P.out.offHapHMQ) P.out.offHapHMQ[pair] = good ? 0 : 1;         // always false (:529); a thrown pair never counts as on-haplotype
            if (P.out.numIndels) P.out.numIndels[pair] = 0;
            if (P.out.numMismatch) P.out.numMismatch[pair] = 0;
            if (P.out.nBQT) P.out.nBQT[pair] = 0;
            if (P.out.nmmBQT) P.out.nmmBQT[pair] = 0;
            if (P.out.nMMLeft) P.out.nMMLeft[pair] = 0;
            if (P.out.nMMRight) P.out.nMMRight[pair] = 0;
            if (!good) {
                P.out.status[pair] = hap_ok ? DD_PAIR_NAN : DD_PAIR_HAPSIZE;
                P.out.ll[pair] = 0.0;
                if (nv > 0) {
                    const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)(cb + t) * nv;
                    for (int i = 0; i < nv; i++) {
                        if (P.out.var_covered) P.out.var_covered[vb + i] = 0;
                        if (P.out.var_fcov) P.out.var_fcov[vb + i] = 0;
                    }
                }
            }
        }
        __syncthreads();
        for (int t = tid; t < nch; t += blockDim.x) {
            const uint32_t k = skey[t];
            int rank = 0;
            for (int j = 0; j < nch; j++) {
                const uint32_t kj = skey[j];
                rank += (kj < k || (kj == k && j < t)) ? 1 : 0;
            }
            sord[rank] = (uint16_t)t;
        }
        __syncthreads();
        for (int rb = (split * nwav + wave) * ngrp; rb < nch; rb += P.n_split * nwav * ngrp) {
            const bool valid = rb + grp < nch && grp < ngrp;
            const int ri = cb + (valid ? (int)sord[rb + grp] : 0);
            const int rr = r0 + (valid ? ri : 0);
            const int64_t pair = pair_base + ri;
            const int so = P.read_seq_off[rr];
            const int Lraw = P.read_seq_off[rr + 1] - so;
            const bool good = valid && hap_ok && Lraw >= 4;
            const int L = good ? Lraw : 0;
            if (__ballot(good) == 0) continue;

            const int bMid = good ? fast_bmid(hapStart, hlen, P.read_start[rr], L) : 0;
            // stage the read, clear the vote histogram (bin index rpfb + L, rpfb in [-(L-4), hlen-5])
            const int F = L + hlen;
            for (int b = l16; b < L; b += 16)
                rd[b] = (uint16_t)((unsigned char)P.read_seq[so + b] | ((unsigned)P.read_qidx[so + b] << 8));
            for (int i = l16; i < (F + 1) / 2; i += 16) freq[i] = 0;
            wave_sync();
            // AlignHash (Faster.cpp:131-189): every read 4-mer votes for the diagonals of the equal haplotype 4-mers
            for (int x = l16; x <= L - 4; x += 16) {
                int key = 0;
                for (int y = 0; y < 4; y++) key |= fast_map_char(rd[x + y] & 0xFF) << (2 * y);
                const int e = bk[key + 1];
                for (int p = bk[key]; p < e; p++) {
                    const int idx = (int)hpl[p] - x + L;
                    atomicAdd(&freq[idx >> 1], (idx & 1) ? 0x10000 : 1);
                }
            }
            wave_sync();
            // top 15 diagonals: frequency descending, ties by ascending relative position (:159-181)
            int myrel = 0x7fffffff;      // lane i < S of the group holds candidate i (unsorted)
            int S = 0;
            bool done = !good;
            for (int round = 0; round < 15; round++) {
                unsigned best = 0;
                if (!done)
                    for (int i = l16; i < (F + 1) / 2; i += 16) {
                        const unsigned v = (unsigned)freq[i];
                        if (v) {
                            const unsigned f0 = v & 0xFFFFu, f1 = v >> 16;
                            const unsigned k0 = f0 ? ((f0 << 16) | (unsigned)(0xFFFF - 2 * i)) : 0u;
                            const unsigned k1 = f1 ? ((f1 << 16) | (unsigned)(0xFFFF - (2 * i + 1))) : 0u;
                            const unsigned k = k0 > k1 ? k0 : k1;
                            best = k > best ? k : best;
                        }
                    }
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) {
                    const unsigned o = __shfl_xor(best, off, 16);
                    best = o > best ? o : best;
                }
                done = done || best == 0;
                if (!done) {
                    const int idx = 0xFFFF - (int)(best & 0xFFFFu);
                    if (l16 == S) myrel = idx - L;
                    if (l16 == 0) freq[idx >> 1] &= (idx & 1) ? 0x0000FFFF : 0xFFFF0000;
                    S++;
                }
                if (__ballot(!done) == 0) break;
                wave_sync();
            }
            if (good) {
                if (l16 == S) myrel = -L;                // relPos.push_back(-readLen) (:263)
                S++;
            }
            // sort ascending (values are distinct): rank = number of smaller values
            srt[l16] = myrel;
            wave_sync();
            int rank = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) rank += (srt[i] < myrel) ? 1 : 0;
            wave_sync();
            srt[l16 < S ? rank : l16] = (l16 < S) ? myrel : 0;
            wave_sync();
            const int relMine = srt[l16];
            const bool act = good && l16 < S;
            const int Smax = gmax4(S);
            const int code0 = l16, code1 = l16 | 16;
            const int wI = 32 - __clz(l16 + 1);          // bits of the inserted-state field of bt_right: values 0..own+1

            // ---------------- SStateHMM (:253-576) ----------------
            auto emis = [&](int r, double &LM, double &ob) {            // logMatch[r] and obs[r][own diagonal] (:286-296)
                const unsigned v = rd[r];
                const double2 q = qt[v >> 8];
                const int hp = relMine + r;
                LM = q.x;
                ob = (hp >= 0 && hp < hlen && (unsigned)shHap[hp] != (v & 0xFFu)) ? q.y : q.x;
            };
            // Transition terms between every source diagonal cs and this lane's diagonal (:339-352) are loop constants.
            // A term that does not apply to this lane (wrong side of the diagonal order) is -inf, so the candidate it
            // produces is -inf and can never pass `nv > cur + EPS`: no lane masks in the inner loops.
            // Sources beyond the pair's S publish -inf, so the source loops may run to the next multiple of 4 of the
            // wave's largest S: four fully unrolled instances, no per-source bounds checks.
            double aN = 0.0, aI = 0.0;                  // previous base's values of this diagonal (0 at the read end)
            double leftN = 0.0, leftI = 0.0;
            auto passes = [&](auto nsc) {
                constexpr int NS = decltype(nsc)::value;
                int lv = l16;                               // opaque copy: keeps the per-source selects below from being hoisted
                asm volatile("" : "+v"(lv));                // out of the read loop as 2 x 16 spilled lane constants
            // from left to bMid (:373-416)
            {
                double tA[NS], tB[NS];
                int dL[NS];
#pragma unroll
                for (int cs = 0; cs < NS; cs++) {
                    const int df = srt[cs] - relMine;
                    const double trI = (fabs((double)df) - 1.0) * IIf;
                    tA[cs] = (cs < lv) ? trI + lE : (cs == lv ? l1mE : FNEG_INF);   // on-diagonal source cs <= own (:380-384)
                    tB[cs] = (cs > lv) ? trI : FNEG_INF;                              // inserted source cs > own (:404-411)
                    dL[cs] = (cs > lv) ? df : 0x7fffffff;                             // its condition relPos[cs]-r >= relPos[ns]
                    asm volatile("" : "+v"(tA[cs]), "+v"(tB[cs]), "+v"(dL[cs]));           // keep them as plain register constants
                }
                const int rows = gmax4(bMid);
                double LMn = 0.0, obn = 0.0;
                if (act && bMid > 0) emis(0, LMn, obn);
                for (int r = 0; r < rows; r++) {
                    const bool rowact = act && r < bMid;
                    const double LM = LMn, ob = obn;
                    const double pvOwn = ob + aN;
                    bc[l16] = rowact ? make_double2(pvOwn, aI) : make_double2(FNEG_INF, FNEG_INF);
                    wave_sync();
                    if (act && r + 1 < bMid) emis(r + 1, LMn, obn);
                    double curN = -1000.0, curI = -1000.0;
                    int bpN = 0, bpI = 32;                                      // untouched = the reference's bt 0: on-diagonal state of diagonal 0
#pragma unroll
                    for (int cs = 0; cs < NS; cs++) {
                        {
                            const double2 s = bc[cs];
                            const double vA = s.x + tA[cs];
                            const double vB = ((LM + tB[cs]) + lE) + s.y;
                            FOLD(curN, bpN, fmax(vA, vB), cs, dL[cs] >= r);
                        }
                    }
                    FOLD(curI, bpI, pvOwn + NIf, code0, true);                   // (:387-391)
                    FOLD(curI, bpI, (LM + IIf) + aI, code1, true);               // (:396-400)
                    if (rowact) {
                        bt[r * 16 + l16] = (unsigned char)(bpN | (bpI & 48));   // bt_left: bits 0-3 source diagonal of the on-diagonal state (a source above the own diagonal is its inserted state); bit 4: the inserted state came from itself, bit 5: it was never set
                        aN = curN; aI = curI;
                    }
                    wave_sync();
                }
            }
            leftN = aN; leftI = aI;                     // alpha[bMid-1] (0 if bMid == 0)
            // from right to bMid (:422-466)
            aN = 0.0; aI = 0.0;
            {
                double tD[NS], tA[NS];
                int dR[NS];
#pragma unroll
                for (int cs = 0; cs < NS; cs++) {
                    const int df = srt[cs] - relMine;
                    const double trI = (fabs((double)df) - 1.0) * IIf;
                    tD[cs] = (cs < lv) ? trI : FNEG_INF;                  // into the inserted state of a higher diagonal (:453-461)
                    tA[cs] = (cs > lv) ? trI + lE : FNEG_INF;             // on-diagonal source cs > own (:427-431); own: below
                    dR[cs] = df;                                           // condition relPos[cs] > relPos[ns]-r
                    asm volatile("" : "+v"(tD[cs]), "+v"(tA[cs]), "+v"(dR[cs]));
                }
                const int rows = gmax4(L - 1 - bMid);
                double LMn = 0.0, obn = 0.0;
                if (act && bMid < L - 1) emis(L - 1, LMn, obn);
                for (int k = 0; k < rows; k++) {
                    const int r = L - 1 - k;
                    const bool rowact = act && r > bMid;
                    const double LM = LMn, ob = obn;
                    bc[l16] = rowact ? make_double2(ob, aN) : make_double2(FNEG_INF, 0.0);
                    wave_sync();
                    if (act && r - 1 > bMid) emis(r - 1, LMn, obn);
                    double curN = -1000.0, curI = -1000.0;
                    int bpN = -1, bpI = -1;                                     // untouched = the reference's bt 0
                    FOLD(curN, bpN, (ob + aN) + l1mE, code0, true);              // own diagonal (:427-431)
                    FOLD(curN, bpN, (LM + lE) + aI, code1, true);                // (:436-438)
#pragma unroll
                    for (int cs = 0; cs < NS; cs++) {
                        {
                            const double2 s = bc[cs];
                            const double vD = ((s.x + NIf) + tD[cs]) + s.y;
                            const double vA = (s.x + s.y) + tA[cs];
                            FOLD(curI, bpI, vD, cs, dR[cs] > -r);
                            FOLD(curN, bpN, vA, cs, true);
                        }
                    }
                    FOLD(curI, bpI, (LM + IIf) + aI, code1, true);               // (:443-447)
                    if (rowact) {
                        // bt_right: two variable-width fields (widths depend on the diagonal, 8 bits in total at most):
                        // low wI bits, inserted state: 0 never set, 1 itself, 2+cs on-diagonal source cs < own;
                        // the rest, on-diagonal state: 0 never set, 1 own inserted state, 2+(cs-own) on-diagonal source cs >= own
                        const int iIdx = bpI < 0 ? 0 : ((bpI & 16) ? 1 : bpI + 2);
                        const int nIdx = bpN < 0 ? 0 : ((bpN & 16) ? 1 : bpN - l16 + 2);
                        bt[r * 16 + l16] = (unsigned char)(iIdx | (nIdx << wI));
                        aN = curN; aI = curI;
                    }
                    wave_sync();
                }
            }
            };
            if (Smax <= 4) passes(std::integral_constant<int, 4>());
            else if (Smax <= 8) passes(std::integral_constant<int, 8>());
            else if (Smax <= 12) passes(std::integral_constant<int, 12>());
            else passes(std::integral_constant<int, 16>());
            // join at bMid (:469-538): plain '>' maxima over x = ins*S + y
            double ll = FNEG_INF;
            int xH = 0;
            {
                double vN = FNEG_INF, vI = FNEG_INF, hN = FNEG_INF, hI = FNEG_INF;
                if (act) {
                    const int mqi = P.read_mqidx[rr];
                    const double lOn = T[T_MAPQF + 2 * mqi], lOff = T[T_MAPQF + 2 * mqi + 1];
                    double LM, ob;
                    emis(bMid, LM, ob);
                    const int hp = relMine + bMid;
                    const bool on = hp >= 0 && hp < hlen;
                    const bool hasR = bMid < L - 1, hasL = bMid > 0;
                    vN = ob + ((on ? lOn : lOff) + l1mE);
                    vI = LM + ((on ? lOn : lOff) + lE);
                    hN = ob + ((on ? hqOn : hqOff) + l1mE);
                    hI = LM + ((on ? hqOn : hqOff) + lE);
                    if (hasR) { vN += aN; vI += aI; hN += aN; hI += aI; }
                    if (hasL) { vN += leftN; vI += leftI; hN += leftN; hI += leftI; }
                }
                ll = vN > vI ? vN : vI;
                double mh = hI > hN ? hI : hN;                       // first maximum: the on-diagonal state wins a tie
                int mx = hI > hN ? 16 + l16 : l16;                   // order key: ins*16 + diagonal (same order as ins*S + y)
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) {
                    const double oll = __shfl_xor(ll, off, 16), omh = __shfl_xor(mh, off, 16);
                    const int omx = __shfl_xor(mx, off, 16);
                    ll = oll > ll ? oll : ll;
                    const bool tk = omh > mh || (omh == mh && omx < mx);
                    mh = tk ? omh : mh;
                    mx = tk ? omx : mx;
                }
                xH = (mh == FNEG_INF) ? 0 : mx;                      // nothing exceeded -inf: xmax stays 0 (:508)
            }
            // backtrack (:540-548); every lane of the group walks the same path.  code = diagonal | 16 if inserted.
            wave_sync();                                             // st aliases the histogram: all its readers are done
            if (l16 == 0 && good) st[bMid] = (int16_t)xH;
            {
                const int rows = gmax4(bMid);
                int code = xH;
                for (int k = 0; k < rows; k++) {
                    const int b = bMid - k;
                    if (good && b > 0) {
                        const int d = code & 15, wv = bt[(b - 1) * 16 + d];
                        if (code & 16) code = (wv & 32) ? 0 : (d | (wv & 16));
                        else { const int c = wv & 15; code = c | (c > d ? 16 : 0); }
                        if (l16 == 0) st[b - 1] = (int16_t)code;
                    }
                }
                const int rows2 = gmax4(L - 1 - bMid);
                code = xH;
                for (int k = 0; k < rows2; k++) {
                    const int b = bMid + k;
                    if (good && b < L - 1) {
                        const int d = code & 15, wv = bt[(b + 1) * 16 + d];
                        const int wd = 32 - __clz(d + 1);
                        if (code & 16) { const int i = wv & ((1 << wd) - 1); code = i == 0 ? 0 : (i == 1 ? (d | 16) : i - 2); }
                        else { const int n = wv >> wd; code = n == 0 ? 0 : (n == 1 ? (d | 16) : n - 2 + d); }
                        if (l16 == 0) st[b + 1] = (int16_t)code;
                    }
                }
            }
            wave_sync();
            // mapState (:552-571)
            {
                const int rows = gmax4(L);
                int lhp = 1;
                for (int r = 0; r < rows; r++) {
                    if (r < L) {
                        const int c = st[r];
                        int m;
                        if (!(c & 16)) {
                            const int hp = srt[c & 15] + r;
                            if (hp >= 0 && hp < hlen) { m = hp + 1; lhp = hp + 1; }
                            else if (hp < 0) m = 0; else m = hlen;
                        } else m = hlen + 2 + lhp;
                        if (l16 == 0) st[r] = (int16_t)m;
                    }
                }
            }
            wave_sync();
            // reportVariants (:579-681): hpos, firstBase / lastBase
            int firstB = 0x7fffffff, lastB = -1;
            int16_t *hp_out = (P.out.hpos && good) ? P.out.hpos + hpos_base + (so - rs_base) : nullptr;
            for (int b = l16; b < L; b += 16) {
                const int s = st[b];
                const int xm = s % numS;
                int hp;
                if (xm > 0 && xm <= hlen) {
                    if (s >= numS) hp = DD_HPOS_INS_KEY0 - xm;     // inserted base, carrying its key (pos = lhp, :556-566, :608)
                    else { hp = s - 1; firstB = hp < firstB ? hp : firstB; lastB = hp > lastB ? hp : lastB; }
                } else hp = (xm == 0) ? DD_HPOS_LO : DD_HPOS_RO;
                if (hp_out) hp_out[b] = (int16_t)hp;
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const int f = __shfl_xor(firstB, off, 16), l2 = __shfl_xor(lastB, off, 16);
                firstB = f < firstB ? f : firstB;
                lastB = l2 > lastB ? l2 : lastB;
            }
            if (firstB == 0x7fffffff) firstB = -1;
            const int64_t vb = (nv > 0) ? P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv : 0;
            if (P.out.var_covered && nv > 0 && good) {
                for (int i = l16; i < nv; i += 16) {
                    const int sR = P.hap_var[2 * (P.hap_var_off[g] + i)], eR = P.hap_var[2 * (P.hap_var_off[g] + i) + 1];
                    P.out.var_covered[vb + i] = (firstB + P.padCover <= sR && lastB - P.padCover >= eR) ? 1 : 0;
                }
            }
            // DetInDel::filterHaplotypes' per-read test (DInDel.cpp:1951-2054): this model leaves numIndels = 0 and
            // offHapHMQ = false, so every read is selected, and its hpos may skip or repeat haplotype bases: the
            // covered set is marked base by base (one bit per haplotype base).  Sentinel hpos values (< 0) never cover anything:
            // an interval reaching below haplotype base 0 is never covered (the reference indexes the sequence with them there).
            if (P.out.var_fcov && P.hap_var_flank && nv > 0) {
                for (int i = 0; i < nv; i++) {
                    const int32_t *fl = P.hap_var_flank + 3 * (size_t)(P.hap_var_off[g] + i);
                    const int left = fl[0] - P.padCover, right = fl[1] + P.padCover, kind = fl[2];
                    int cov = 0;
                    if (kind != 0 && right >= left) {
                        wave_sync();
                        for (int x = l16; x < (hlen + 31) / 32; x += 16) bm[x] = 0;
                        wave_sync();
                        int nmm = 0;
                        for (int b = l16; b < L; b += 16) {
                            const int s2 = st[b];
                            if (s2 >= 1 && s2 <= hlen) {
                                const int hb = s2 - 1;
                                if (hb >= left && hb <= right) {
                                    atomicOr(&bm[hb >> 5], 1 << (hb & 31));
                                    const unsigned hc = shHap[hb];
                                    nmm += ((rd[b] & 0xFFu) != hc && (kind == 2 || hc != 'N')) ? 1 : 0;   // 'N' exempt for DEL (:1992)
                                }
                            }
                        }
                        wave_sync();
                        int csize = 0;
                        const int lo = left > 0 ? left : 0, hi = right < hlen - 1 ? right : hlen - 1;
                        for (int x = lo + l16; x <= hi; x += 16) csize += (bm[x >> 5] >> (x & 31)) & 1;
#pragma unroll
                        for (int off = 8; off >= 1; off >>= 1) {
                            nmm += __shfl_xor(nmm, off, 16);
                            csize += __shfl_xor(csize, off, 16);
                        }
                        cov = (csize >= right - left + 1 && nmm <= P.maxMismatch) ? 1 : 0;
                    }
                    if (l16 == 0 && good) P.out.var_fcov[vb + i] = (uint8_t)cov;
                }
            }
            if (l16 == 0 && good) {
                P.out.ll[pair] = ll;
                P.out.status[pair] = DD_PAIR_OK;             // computeLikelihoodsFaster has no ll checks
                if (P.out.firstBase) P.out.firstBase[pair] = (int16_t)firstB;
                if (P.out.lastBase) P.out.lastBase[pair] = (int16_t)lastB;
            }
            wave_sync();
        }
        }   // chunk of reads
    }
}

hipError_t launch_faster(const KernelArgs &A, unsigned grid, int waves, size_t lds, hipStream_t st)
{
    // raised once per device to the CU's 160 KiB (see launch_one in hmm_kernel.hip: a per-launch value races between host threads)
    static std::atomic<unsigned> raised(0u);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned bit = 1u << (dev & 31);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dd_faster_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised.fetch_or(bit, std::memory_order_release);
    }
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dd_faster_kernel, dim3(grid), dim3(waves * 64), lds, st, A);
    return hipGetLastError();
}

} // namespace ddk
