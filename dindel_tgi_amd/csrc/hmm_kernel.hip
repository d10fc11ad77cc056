// hmm_kernel.hip — gfx950 (CDNA4) kernels for Dindel's read x candidate-haplotype HMM likelihood.
//
// Replaces, for a whole batch of windows, the double loop of DetInDel::computeLikelihoods
// (reference DInDel.cpp:1715-1737) and everything ObservationModelFBMaxErr::calcLikelihood does per
// (haplotype, read) pair (reference ObservationModelFB.cpp:1057-1165, 1351-1475, 1641-1829).
//
// Mapping to the machine (NOT a translation of the reference's per-pair heap arrays):
//   * one workgroup = one candidate haplotype (x a slice of its window's reads); the haplotype's
//     state codes and per-position indel-error logs are built once in LDS and turned into per-lane
//     register constants, then reused for every read the workgroup's wavefronts process;
//   * one wavefront = one (read, haplotype) pair at a time; lane l owns the K consecutive hap
//     positions x = l*K .. l*K+K-1 (both the "on base x" and the "inserted at x" state), so a
//     64-wide wave covers numS = Hs+2 <= 64*K positions and one read base (one HMM slice) is one
//     sweep of straight-line fp64 add / compare / select code;
//   * neighbour states (the <= D = maxLengthDel+1 jump candidates) come from a wave-private LDS row;
//     single buffer, no barrier: DS operations of one wave execute in order;
//   * back-pointers are 4 (5 when D > 7) bits per (read base, position), packed per lane into one word of
//     wave-private LDS — or of a per-wave HBM scratch tile in the GBT builds; the scalarised traceback walks
//     them, and hpos[] / QC counters are produced lane-parallel over read bases;
//   * no transcendental is evaluated on the device: every log() the reference takes is tabulated on
//     the host with libm (dd_build_tables, capi.cpp) so results cannot differ from glibc's;
//   * bases are compared as symbol ids from a 256-entry table (dd_build_symbol_lut): any byte, as in the reference.
//
// Arithmetic: fp64, every sum in the reference's written term order (fp64 + is not associative and
// updateMax has a 1e-10 hysteresis, ObservationModelFB.cpp:877-888).  No MFMA: this is a max-plus
// recurrence, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "hmm_kernel.h"

// which pieces this translation unit carries (see the instantiation block at the end of the file)
#if defined(DD_ONLY_K) || !defined(DD_INST_D) || DD_INST_D == 0 || DD_INST_D == 6
#define DD_INST_COMMON 1
#endif

namespace ddk {

#define DD_EPS 1e-10

// Diagnostic build only (-DDD_STAMPS): per-phase s_memtime shares, summed into P.dbg[phase].  Never
// compiled into the product library; the stamps go to a buffer of their own and feed no output.
#ifdef DD_STAMPS
// One s_memtime per stamp, accumulated per wave in registers and flushed with one atomic per phase when the wave retires
// (an atomic per stamp cost more than the phases it was timing).
#define STAMP(i)                                                                  \
    do {                                                                          \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();               \
        _acc[i] += _t - _tprev;                                                   \
        _tprev = _t;                                                              \
    } while (0)
#define STAMP_INIT unsigned long long _acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long _tprev = __builtin_amdgcn_s_memtime()
#define STAMP_FLUSH                                                               \
    do {                                                                          \
        if (lane == 0 && P.dbg)                                                   \
            for (int _i = 0; _i < 8; _i++) atomicAdd(&P.dbg[_i], _acc[_i]);       \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_INIT do { } while (0)
#define STAMP_FLUSH do { } while (0)
#endif
#define NEG_INF (-__builtin_huge_val())

// Symbols.  The reference compares bases as characters (hap[y]==nuc, ObservationModelFB.cpp:246; read.seq[b]!=hap.seq[s-1],
// :1410) and treats only a haplotype 'N' as a wildcard, so any byte may occur on either side (IUPAC codes, soft-masked
// lower case).  Bytes are mapped to symbol ids through a 256-entry table (dd_build_symbol_lut): A,C,G,T -> 0..3, N -> 4,
// every other byte present in a haplotype of the batch -> 5..30, every remaining byte -> 31 (a read-only symbol: equal to no
// haplotype symbol).  Equal ids <=> equal characters wherever a haplotype is involved.  Without a table: A,C,G,T,N and 31.
#define DD_SYM_N 4
#define DD_SYM_PAD 255   /* padded state: matches nothing */
__device__ __forceinline__ int builtin_symbol(unsigned ch)
{
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'N' ? DD_SYM_N : 31;
}

// hap[y]=='N' || hap[y]==nuc  (ObservationModelFB.cpp:246): state symbol 4 = 'N' or LO/RO (always eq) matches every read
// symbol, any other state symbol only itself — kept per position as a 32-bit mask over read symbols (mOwn)

// exact ObservationModelFBMax::updateMax (ObservationModelFB.cpp:877-888) for the few special states
__device__ __forceinline__ void update_max(double &dest, int &idx, int &code, double v, int newIdx, int newCode)
{
    if (v > dest + DD_EPS) { dest = v; idx = newIdx; code = newCode; }
    else if (v >= dest && v <= dest + 1e-5 && idx > newIdx) { dest = v; idx = newIdx; code = newCode; }
}

// max of two non-NaN doubles as ONE v_max_f64 (the generic fmax lowering may add canonicalising self-max
// instructions in IEEE mode)
__device__ __forceinline__ double dmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// G pairs per wavefront (G = 1 or 2): a pair's states live on a group of W = 64 / G consecutive lanes.  The cross-lane steps of a pair
// stay inside its group: xor-shuffles with offsets below W, the group's W bits of a ballot.  Control flow may diverge BETWEEN the two
// groups of a wavefront (different reads), never inside one: every lane of a group holds the same per-read values.
template <int G>
__device__ __forceinline__ double group_max(double v)
{
#pragma unroll
    for (int off = 32 / G; off >= 1; off >>= 1) {
        const double o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}
// the group's bits of a ballot, lane 0 of the group in bit 0 (G = 1: the ballot itself)
template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool p, int grp)
{
    const unsigned long long b = __ballot(p);
    if constexpr (G == 1) return b; else return (b >> (grp * 32)) & 0xffffffffull;
}

// arg max over one HMM slice with the reference's scan semantics (ObservationModelFB.cpp:1096-1102,
// 1110-1114): states visited in index order, a state replaces the incumbent only if it beats it by
// more than EPS.  If exactly one state lies within 3e-10 of the true maximum M, that scan provably ends
// on it (every earlier incumbent is <= M-3e-10 < M-EPS, nothing later exceeds M+EPS), so the parallel
// max is exact.  Otherwise (near-ties, e.g. repeats) the scan is replayed verbatim from LDS.
template <int K, int G>
__device__ __forceinline__ void slice_argmax(const double (&vA)[K], const double (&vI)[K], int x0, int numS, int grp,
                                             double *ldsA, double *ldsI, double &best, int &idx)
{
    double m = NEG_INF;
#pragma unroll
    for (int k = 0; k < K; k++) {
        m = vA[k] > m ? vA[k] : m;
        m = vI[k] > m ? vI[k] : m;
    }
    m = group_max<G>(m);
    const double thr = m - 3e-10;
    int cnt = 0, cand = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const unsigned long long ba = group_ballot<G>(vA[k] >= thr, grp), bi = group_ballot<G>(vI[k] >= thr, grp);
        cnt += __popcll(ba) + __popcll(bi);
        if (ba) cand = (__ffsll((long long)ba) - 1) * K + k;
        if (bi) cand = numS + (__ffsll((long long)bi) - 1) * K + k;
    }
    if (cnt == 1) { best = m; idx = cand; return; }
#pragma unroll
    for (int k = 0; k < K; k++) { ldsA[x0 + k] = vA[k]; ldsI[x0 + k] = vI[k]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    best = NEG_INF;
    idx = 0;
    for (int s = 0; s < 2 * numS; s++) {
        const double v = s < numS ? ldsA[s] : ldsI[s - numS];
        if (v > best + DD_EPS) { best = v; idx = s; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Back-pointer packing: per position CB bits of transition choice (0 = from the inserted state, y = jump
// length; LO/RO use their own small codes) + 1 bit for the inserted state's choice.  D <= 7 -> 4 bits per
// position, 5 up to D = 15, 7 on the D = 32 build (K <= 9 there: 63 bits).  One lane's K positions of one slice go into one 1/2/4/8-byte word.
// a pair of a window that dd_screen_windows flagged: DD_PAIR_UNSUPPORTED, ll = 0, "off the haplotype"
__device__ __forceinline__ void mark_unsupported(const dd_result &o, int64_t pair)
{
    o.status[pair] = DD_PAIR_UNSUPPORTED;
    o.ll[pair] = 0.0;
    if (o.offHap) o.offHap[pair] = 1;
    if (o.offHapHMQ) o.offHapHMQ[pair] = 1;
}

#ifndef DD_LEAN_RULE
#define DD_LEAN_RULE(K, D, GBT, G) (((GBT) || (G) > 1) && ((D) > 7 || (K) >= 3))
#endif
template <int K, int D> struct BtPack {
    static constexpr int CB = (D <= 7) ? 3 : (D <= 15 ? 4 : 6);       // D = 32 build (maxLengthDel 12..31): choices 0..32
    static constexpr int PB = CB + 1;
    static constexpr int BITS = K * PB;
    static constexpr int BYTES = BITS <= 8 ? 1 : BITS <= 16 ? 2 : BITS <= 32 ? 4 : 8;
};
template <int BYTES> struct BtWord;
template <> struct BtWord<1> { typedef uint8_t type; };
template <> struct BtWord<2> { typedef uint16_t type; };
template <> struct BtWord<4> { typedef uint32_t type; };
template <> struct BtWord<8> { typedef uint64_t type; };

// Occupancy target (waves per SIMD) the register allocator is held to.  Measured (tools/ab_point.py): K<=2, D<=7
// +20 % going from 2 to 3 waves/SIMD (the VALU is the bound and two waves cannot keep it issuing); K=2, D=11
// needs ~236 VGPRs unspilled: the HBM-scratch build still gains 14 % at 3 waves with spills, the LDS build is
// LDS-limited to 2 waves/SIMD anyway and loses 7 % to the spills, so it stays at 2.  K = 3 (round 3, profiles/r03/k3_waves_ab.txt): the
// D = 6 scratch build at 3 waves/SIMD (168 VGPRs, a few spills outside the sweeps) gains 10-12 % (231.7 -> 207.2 ms per 3,000 windows of
// 160-bp haplotypes); the D = 11 one loses 14 % to its spills and stays at 2, as does K = 4.
#ifndef DD_MIN_WAVES_PER_SIMD
#define DD_MIN_WAVES_PER_SIMD(K, D, GBT, G) ((D) > 12 ? ((K) <= 3 ? 3 : (K) <= 6 ? 2 : 1) : (K) <= 2 ? (((D) <= 7 || (GBT)) ? 3 : 2) : ((K) == 3 && (D) <= 7 && (GBT)) ? 3 : ((((K) <= 5 || (G) == 2) && (GBT)) ? 2 : 1))
#endif
// GBT = back-pointers in a per-wave HBM scratch tile instead of LDS: for read length x haplotype length
// combinations whose tile would leave a CU with too few wavefronts (or not fit its 160 KiB at all).  The
// tile is written once per read base (coalesced, one word per lane) and read back by the scalar traceback;
// a few tens of MB for the whole chip, so it lives in L2 / Infinity Cache.
// FOLD: the right->middle pass runs its LO / RO end states inside the generic candidate code (see foldRO in the body); built
// for the LDS builds with K <= 2, D = 6 only (the other builds lose more to the extra stores than the blocks cost them).
// OCC: waves per SIMD the register allocator is held to when it is not the rule above (0 = the rule).  One build uses it: K = 3 at D = 6
// with scratch back-pointers also exists at 2 waves per SIMD, for reads so long (> ~250 bp) that LDS keeps fewer than 12 waves on
// the CU anyway — there the 3-wave build's spills cost 9 % and buy nothing (profiles/r03/plan_check.jsonl).
// G: pairs a wavefront works on at a time.  G = 2 (round 4): two reads of the haplotype side by side on the two 32-lane halves, K
// positions per lane of a half, so a pair costs K / 2 lane-positions per read base instead of ceil((Hs + 2) / 64): haplotypes of
// 127..158 bp run as K = 5 halves (2.5 instead of 3), 63..94 bp as K = 3 halves (1.5 instead of 2), <= 30 bp as K = 1 halves (capi.cpp kHapClasses).
// The two reads advance base by base together (their trip counts are padded to the longer one, the shorter one's lanes masked off), so
// the workgroup first orders the window's reads by (length, bMid) and a wavefront takes two consecutive ranks.  Every
// per-read quantity that the G = 1 build keeps on the scalar unit is a per-lane value here, equal across the lanes of a half.
template <int K, int D, bool GBT, bool FOLD = false, int OCC = 0, int G = 1>
__global__ void __launch_bounds__(DD_WAVES * 64, OCC ? OCC : DD_MIN_WAVES_PER_SIMD(K, D, GBT, G)) dd_hmm_kernel(const KernelArgs P)
{
    static_assert(G == 1 || (G == 2 && !FOLD), "two pairs per wavefront: 32-lane halves, end states in their one-lane blocks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wlane = tid & 63;                    // lane of the wavefront
    constexpr int W = 64 / G;                      // lanes per pair
    const int grp = (G == 1) ? 0 : (wlane >> 5);   // which of the wavefront's pairs this lane works for
    const int lane = (G == 1) ? wlane : (wlane & (W - 1));   // lane within the pair's group: owner of positions lane*K .. lane*K+K-1
    // wave index is uniform across the 64 lanes: tell the compiler, so that everything per read (offsets,
    // lengths, bMid, loop counters, the traceback chain) lives on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthr = blockDim.x, nwav = blockDim.x >> 6;   // 1..4 waves (host picks what fits LDS best)
    const int NP = W * K;
    constexpr bool foldLO = FOLD;                   // (host: only for haplotypes with numS <= NP - 1)
    const int Dr = P.D;                             // real D (== D unless the generic D=12 build is used)
    const double *T = P.tables;
    const double lLL = T[TC_LLL], lFL = T[TC_LFL], II = T[TC_II], NI = T[TC_NI], NN = T[TC_NN];

    // ---------------- shared (per haplotype) region ----------------
    unsigned char *sc = smem;                                   // [NP+16] state symbols
    unsigned char *shLut = smem + P.lds_off_L;                  // [256] byte -> symbol id
    double *shE = reinterpret_cast<double *>(smem + P.lds_off_E);  // [NP+D+2] logProbError per state
    double *shN = reinterpret_cast<double *>(smem + P.lds_off_N);  // [NP+D+2] logProbNoError per state
    double *shQ = reinterpret_cast<double *>(smem + P.lds_off_Q);  // [n_qual][4] eq, uq, log10(1-q), q
    // ---------------- wave-private region ----------------
    unsigned char *wave_base = smem + P.lds_shared_bytes + (size_t)wave * P.lds_wave_bytes;
    unsigned char *wbase = (G == 1) ? wave_base : wave_base + (size_t)grp * P.lds_group_bytes;   // the pair's own rows (G = 1: the wavefront's)
    // one HMM slice as the neighbours see it: {value, emission log of that state for the slice's read base}
    // K interleaved arrays (state s lives in array s % K at index s / K, PADQ pad entries either side) so that every
    // wave-wide ds_read_b128 / ds_write_b128 touches consecutive 16-byte slots: a flat [state] layout puts lanes l and
    // l+8 on the same banks for K = 2 (2-way conflict on every neighbour read; SQ_LDS_BANK_CONFLICT was 41 % of LDS cycles)
    constexpr int PADQ = (D + K - 1) / K;
    constexpr int AQ = W + 2 * PADQ;
    double2 *rowA = reinterpret_cast<double2 *>(wbase + P.lds_off_A);     // [K][AQ]
    double *rowI = reinterpret_cast<double *>(wbase + P.lds_off_I);       // [1 + NP + 1]   state s -> rowI[1+s]
    double *rdE = reinterpret_cast<double *>(wbase + P.lds_off_rdE);      // [Lmax][2]  eq, uq per read base
    unsigned char *rdC = wbase + P.lds_off_rdC;                            // [Lmax] read base code 0..5
    unsigned char *rdQ = wbase + P.lds_off_rdQ;                            // [Lmax] quality index
    int16_t *ms = reinterpret_cast<int16_t *>(wbase + P.lds_off_ms);      // [Lmax] MAP state per base
    // LEAN (HBM-scratch builds with D > 7 or K >= 3): the two [K][D] per-lane constant arrays (88 VGPRs at K = 2, D = 11) do not fit next
    // to the slice window at 3 waves/SIMD; the Inc constants come from a block-shared LDS table instead (LDS is idle in
    // this build) and the Dec jump penalties are formed on the fly from E[x] and a broadcast (y-1)*II.
    constexpr bool LEAN = DD_LEAN_RULE(K, D, GBT, G);
    constexpr bool BIGD = D > 12;                   // the D = 32 build (maxLengthDel 12..31): jump candidates in a run-time loop over y
    static_assert(!BIGD || LEAN, "the D = 32 build is a scratch build (its Dec penalties come from shY)");
    // SLIM (K >= 3 or two pairs per wavefront): the builds that sit at their register budget keep less alive across the sweeps —
    // the per-position constants of a pass are (re)loaded from the haplotype's LDS tables right in front of that pass instead of once
    // per haplotype (their loads cannot be hoisted: the index goes through an opaque register), beta[bMid] waits in a wave-private LDS
    // row while the left->middle pass runs (4 K registers; scratch builds only: behind the wave's back-pointer tile, one coalesced
    // 8-byte store / load per position per read — an LDS row for it cost the K = 5 builds two of their eight waves per CU), and with two pairs per wavefront the per-read addresses are formed again
    // after the sweeps instead of being carried through them.  (Round 4: the K = 5 half-wave build spilled 59 registers, 5 scratch
    // accesses inside the right->middle sweep, without this.)
    // Which builds: the ones measured to gain or hold (tools/plan_check.py against round 3, profiles/r04): K = 3, K = 5, K = 4 at D <= 7 and every
    // half-wave build; K = 4 at D > 7 and K >= 6 (1 wave per SIMD, registers to spare) lost 3-7 % to the reloads and keep the round-3 form.
    constexpr bool SLIM = (G > 1) || K == 3 || K == 5 || (K == 4 && D <= 7);
    constexpr bool STASH = SLIM && GBT;            // beta[bMid] parked behind the wave's back-pointer tile in the HBM scratch: [2 K][64] doubles
    double *shC = reinterpret_cast<double *>(smem + P.lds_off_C);  // LEAN: [K*D][W] lp_y(src)+Nn[src]
    double *shY = reinterpret_cast<double *>(smem + P.lds_off_Y);  // LEAN: [D] (y-1)*II
    typedef BtPack<K, D> BP;
    typedef typename BtWord<BP::BYTES>::type btword_t;
    typedef typename BtWord<(BP::BYTES == 8 ? 8 : 4)>::type btacc_t;   // register type the word is assembled in
    btword_t *bt;                                                          // [Lmax][64] packed back-pointers
    double *beS = nullptr;
    if constexpr (GBT) {
        unsigned char *tile = reinterpret_cast<unsigned char *>(P.bt_scratch) + (size_t)(blockIdx.x * nwav + wave) * (size_t)P.bt_wave_bytes;
        bt = reinterpret_cast<btword_t *>(tile);
        beS = reinterpret_cast<double *>(tile + (size_t)P.bt_rows * 64 * sizeof(btword_t));
    } else
        bt = reinterpret_cast<btword_t *>(wave_base + P.lds_off_bt);
    // G = 2: the workgroup's order of a chunk of the window's reads (sort keys, then read index by rank)
    // the workgroup's work counter: its wavefronts take the haplotype's reads (G = 2: pairs of ranks) one after the other as they become
    // free, so that reads of different lengths, reads of another length class and a read count that is no multiple of the wave count
    // do not leave a wavefront waiting at the next haplotype's barrier while another still has reads queued (round 4)
    int *wq = reinterpret_cast<int *>(smem + P.lds_off_W);
    uint32_t *skey = reinterpret_cast<uint32_t *>(smem + P.lds_off_S);     // [DD_HALF_CHUNK]
    uint16_t *sord = reinterpret_cast<uint16_t *>(skey + DD_HALF_CHUNK);   // [DD_HALF_CHUNK]

    for (int i = tid; i < 4 * P.n_qual; i += nthr) shQ[i] = T[T_QUAL + i];
    for (int i = tid; i < 256; i += nthr) shLut[i] = P.sym_lut ? P.sym_lut[i] : (unsigned char)builtin_symbol((unsigned)i);
    // pads of the wave-private rows: written once, never touched again
    if (lane == 0) { rowI[0] = NEG_INF; rowI[1 + NP] = NEG_INF; }
    STAMP_INIT;

    // ======================= loop over this workgroup's (haplotype, read-slice) items =======================
    // Static (uniform batches): workgroup b takes items b, b + gridDim, ... of an XCD-contiguous numbering.  Dynamic (ragged launches,
    // P.work_counter != NULL): a persistent grid whose workgroups take the next item from a counter in device memory when they are
    // done with the last one — items of a ragged launch differ 20-fold in work (reads per window x read length), and neither the
    // in-order dispatch of a one-shot grid round-robin over the XCDs nor a fixed stride keeps 256 CUs busy to the end then.
    for (int item = P.item_begin + xcd_contiguous_block_id(P.n_items - P.item_begin);; item += gridDim.x) {
    __syncthreads();                               // every wave is done with the previous haplotype's LDS tables
    if (tid == 0) *wq = 0;                         // (the set-up's barriers lie between this and the first pull)
    if (P.work_counter) {
        if (tid == 0) wq[1] = P.item_begin + atomicAdd(P.work_counter, 1);
        __syncthreads();
        item = wq[1];
    }
    if (item >= P.n_items) break;
    const int gi = item / P.n_split;
    const int split = item - gi * P.n_split;
    const int g = P.hap_list ? P.hap_list[gi] : gi;   // global haplotype index

    const int w = P.hap_window[g];
    const int h0 = P.win_hap_off[w];
    const int r0 = P.win_read_off[w], r1 = P.win_read_off[w + 1];
    const int R = r1 - r0;
    if (P.win_skip && P.win_skip[w]) {
        // window outside the kernel limits (dd_screen_windows): its pairs are only marked, nothing of it is read
        const int64_t pb = P.win_pair_off[w] + (int64_t)(g - h0) * R;
        for (int ri = split * nthr + tid; ri < R; ri += P.n_split * nthr) mark_unsupported(P.out, pb + ri);
        continue;
    }
    // workgroups this haplotype's reads are dealt to: the launch's n_split, or — ragged batches — as many as give a wavefront
    // about P.reads_per_wave reads (the other workgroups of the haplotype have nothing to do and leave before the set-up)
    int nsp = P.n_split;
    if (P.reads_per_wave > 0) {
        nsp = ((R + G - 1) / G + nwav * P.reads_per_wave - 1) / (nwav * P.reads_per_wave);
        nsp = nsp < 1 ? 1 : (nsp > P.n_split ? P.n_split : nsp);
        if (split >= nsp) continue;
    }
    const int hs_off = P.hap_seq_off[g];
    const int Hs = P.hap_seq_off[g + 1] - hs_off;
    const int numS = Hs + 2, RO = Hs + 1;
    const uint32_t hapStart = P.win_hap_start[w];
    const bool hap_ok = (P.maxLengthDel <= Hs);     // else "hapSize error." (ObservationModelFB.cpp:47)

    // ---- per-haplotype setup: state codes + homopolymer indel-error logs (setupTransitionProbs :1675-1703)
    for (int s = tid; s < NP + 16; s += nthr) {
        int code = DD_SYM_PAD;                       // padded state: matches nothing
        if (s == 0 || s == RO) code = DD_SYM_N;      // LO / RO: emission is always eq (:237-241)
        else if (s < RO) code = shLut[(unsigned char)P.hap_seq[hs_off + s - 1]];
        sc[s] = (unsigned char)code;
    }
    for (int s = tid; s < NP + D + 2; s += nthr) {
        shE[s] = T[TC_EDEF];
        shN[s] = T[TC_NDEF];
    }
    __syncthreads();
    if (tid == 0) { shE[1] = T[T_HP + 2 * 1]; shN[1] = T[T_HP + 2 * 1 + 1]; }
    __syncthreads();
    for (int b = 1 + tid; b < Hs; b += nthr) {
        const char *hp = P.hap_seq + hs_off;
        if (hp[b] != hp[b - 1]) {
            int len = 1;
            while (b - 1 - len >= 0 && hp[b - 1 - len] == hp[b - 1]) len++;
            int li = len < DD_HP_TABLE - 1 ? len : DD_HP_TABLE - 1;
            shE[b] = T[T_HP + 2 * li];
            shN[b] = T[T_HP + 2 * li + 1];
        }
    }
    __syncthreads();
    if (tid == 0 && Hs >= 1) {
        const char *hp = P.hap_seq + hs_off;
        int len = 1;
        while (Hs - 1 - len >= 0 && hp[Hs - 1 - len] == hp[Hs - 1]) len++;
        int li = len < DD_HP_TABLE - 1 ? len : DD_HP_TABLE - 1;
        shE[Hs - 1] = T[T_HP + 2 * li];              // index hapSize-1, as the reference writes it (:1702)
        shN[Hs - 1] = T[T_HP + 2 * li + 1];
    }
    __syncthreads();

    if constexpr (LEAN) {
        if constexpr (D <= 12)                     // (the D = 32 build forms these on the fly: its table would take 16 KB x K of LDS)
        for (int i = tid; i < K * D * W; i += nthr) {
            const int l = i & (W - 1), ky = i / W, k = ky / D, y = ky - k * D + 1;
            const int src = l * K + k + y;
            double c = NEG_INF;
            if (src <= RO && y <= Dr) {
                const double Es = shE[src], Ns = shN[src];
                const double lp = (y == 1) ? Ns : Es + (double)(y - 1) * II;
                c = lp + Ns;
            }
            shC[i] = c;
        }
        for (int i = tid; i < D; i += nthr) shY[i] = (i + 1 > Dr) ? NEG_INF : (double)i * II;   // y = i+1; jumps past the real D are off
        __syncthreads();
        // the end states riding in the generic code (see foldLO / foldRO below): their candidate constants
        if (foldLO) {
            for (int y = 1 + tid; y <= D; y += nthr) {
                shC[((K - 1) * D + y - 1) * 64 + 63] = y == 2 ? lLL : (y == 3 ? lFL : NEG_INF);
                if (RO - (RO / K) * K == K - 1 && numS <= NP - 3) shC[((K - 1) * D + y - 1) * 64 + RO / K] = y == 2 ? 0.0 : NEG_INF;
            }
            __syncthreads();
        }
    }
    // ---- per-lane register constants for this haplotype (states past RO are switched off with -inf) ----
    const int x0 = lane * K;
    const int laneRO = RO / K, kRO = RO - laneRO * K;
    double lpn[K], eIn[K], eInc[K], niDec[K];
    uint32_t mOwn[K];                               // bit c: this state's emission is eq for read column c
    auto load_dec_constants = [&](int xb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int x = xb + k;
            const bool valid = x < numS;
            // Loaded for every lane, then selected (x < NP lies inside the tables): NO branch over part of the wavefront here.  hipcc (ROCm 7.2)
            // put register-allocator spills of values that live across such a branch IN FRONT of the exec restore of its join block in three
            // builds (K = 11 / D = 6 and K = 7 / D = 12 scratch, K = 7 / D = 12 LDS): the lanes that sat the branch out lost the LDS address of
            // their constants from a workgroup's second item on (tools/check_exec_spills.py, tests/test_gpu_persistent_rounds.py).
            double tn = shN[x], te = shE[x];
            asm volatile("" : "+v"(tn), "+v"(te));  // (opaque: the loads are not sunk back under the condition)
            lpn[k] = valid ? tn : NEG_INF;          // Dec: logProbNoError[x]
            eIn[k] = valid ? te : NEG_INF;          // Dec: logProbError[x]   (insertion-open into x)
            niDec[k] = (x == 0) ? NEG_INF : NI;     // Dec: no "inserted -> on base" edge into LO (:1823 starts at x=1)
        }
    };
    auto load_inc_constants = [&](int xb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < K; k++) {                // Inc: logProbError[x+1]   (x + 1 <= NP lies inside the table)
            double te = shE[xb + k + 1];
            asm volatile("" : "+v"(te));
            eInc[k] = (xb + k + 1 <= RO) ? te : NEG_INF;
        }
    };
    if constexpr (!SLIM) { load_dec_constants(x0); load_inc_constants(x0); }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int x = x0 + k;
        const bool valid = x < numS;
        // bit c set <=> code_match(state symbol, c): 'N' / LO / RO match every column, another symbol its own, a pad none
        const unsigned scode = sc[x];
        unsigned m = scode == DD_SYM_N ? 0xffffffffu : (scode < 32u ? 1u << scode : 0u);
        asm volatile("" : "+v"(m));                 // (as above: every lane loads and forms its mask, a select drops it: no branch)
        mOwn[k] = valid ? m : 0u;
    }
    const double Nn_RO = shN[RO], E_RO = shE[RO], E_Hs = shE[Hs], E_1 = shE[1];
    // Right->middle pass: the two end states ride in the generic candidate code instead of running as one-lane blocks of their
    // own (13 + 8 fp64 and 7 + 5 32-bit wave instructions per read base, a fifth of that pass's sweep).
    //   * LO moves from position 0 to the idle position NP-1 (lane 63's last slot), whose jump candidates read the pad slots
    //     NP, NP+1, NP+2.  Those carry what the LO recurrence (:1720-1722, :1746-1749, :1762) needs, so that the generic forms
    //     (c_y + v) + ov, (eq + in) + eInc and (ov + v) + NI evaluate the reference's sums term by term:
    //       slot NP   {beta[LO], eq}            y = 1 switched off (c = -inf); read by the inserted state:  (eq + beta[LO]) + NI
    //       slot NP+1 {eq + beta[LO], NN}       y = 2, c = lLL:   ((eq + beta[LO]) + lLL) + NN             -> stays LO      (idx 0)
    //       slot NP+2 {ov1 + beta[1], NN}       y = 3, c = lFL:   ((obs[1] + beta[1]) + lFL) + NN          -> base 1        (idx 1)
    //       insertion edge with eInc = E[1]:    (eq + beta[numS]) + E[1]                                     -> numS          (idx numS)
    //     (x + y is commutative and the association is kept; y >= 2 and the insertion edge use the same "> best + EPS" rule the LO
    //     block used; y = 1 leaves best = -inf, which the first real candidate always beats.)
    //   * RO, when it sits in its lane's last slot (kRO == K-1: its candidates then come from LDS, not from the lane's own
    //     registers), gets {beta[RO], eq} into slot RO+1 and {eq + beta[RO], Nn[RO]} into slot RO+2, c_2 = 0, eInc = 0:
    //       y = 2: (0 + (eq + beta[RO])) + Nn[RO]   -> stays RO;   insertion edge: (eq + beta[numS+RO]) + 0   -> numS+RO   (:1741-1742, :1750)
    //   LO needs position NP-1 idle (numS <= NP-1: the host launches the FOLD build only for length classes with 64 K >= Hs + 3,
    //   capi.cpp); RO's two slots must not reach LO's position, so RO is folded only when numS <= NP-3 as well — otherwise its
    //   one-lane block runs as in the other builds.
    //   What it buys (profiles/r03/fold_ab.txt, fold_publish_ab.txt): the sweep's loop body goes from 214 to 188 instructions (fp64
    //   89 -> 70) and the kernel gained 2.3-3.3 % at K <= 2 / D = 6 with the first form of the stores in front of the slice exchange
    //   (two masked blocks, a dozen moves, spilled-SGPR addresses: 245.8 -> 240.3 ms per 6,000 windows of configs[1]); with the one
    //   masked block below (197 instructions) another 2.1 % (-> 235.9 ms).  It now also pays for K = 2 at D = 11 with LDS back-pointers
    //   (+1.4 %) and for the K = 2 scratch build at D = 6 (long reads, +3-4 %); the D = 11 scratch build still loses 4.5 % and K >= 3
    //   would need state 1's pair from another slot than the lane's last, so those keep the one-lane blocks (capi.cpp picks).
    const bool foldRO = foldLO && kRO == K - 1 && numS <= NP - 3;
    constexpr int k1 = 1 % K;                       // state 1 lives in lane 1 / K, slot k1
    const int lane1 = 1 / K;
    if (foldLO && lane == 63) eInc[K - 1] = E_1;
    if (foldRO && lane == laneRO) eInc[K - 1] = 0.0;
    // where this lane stores for its end state: lane 63 (LO) slots NP / NP+1, lane laneRO (RO, when folded) slots RO+1 / RO+2;
    // the owner of state 1 stores obs[1] + beta[1] into slot NP+2
    const bool isEnd = foldLO && (lane == 63 || (foldRO && lane == laneRO));
    double *rowAd = reinterpret_cast<double *>(rowA);
    double *endA = rowAd + 2 * (lane == 63 ? (0 % K) * AQ + PADQ + 64 + 0 / K : 0 * AQ + PADQ + laneRO + 1);
    double *endV = rowAd + 2 * (lane == 63 ? (1 % K) * AQ + PADQ + 64 + 1 / K : (1 % K) * AQ + PADQ + laneRO + 1 + 1 / K);
    double *oneV = rowAd + 2 * ((2 % K) * AQ + PADQ + 64 + 2 / K);
    const double endC = (lane == 63) ? NN : Nn_RO;
    // One masked block per read base stores all three: the owners of LO / RO and the owner of state 1 run the same four stores with
    // their own addresses and constants (state 1's owner has no first pair to give: it goes to the unused pad slot NP+3, which lane 63
    // reads under c = -inf).  LO's idle position takes the emission of a state that matches every column, like RO's own, so the
    // value that goes with beta is the lane's ov[K-1] in all three.
    int pubFlag = (isEnd || (foldLO && lane == lane1)) ? 1 : 0;
    asm volatile("" : "+v"(pubFlag));               // one opaque per-lane flag: one compare per read base instead of the three conditions re-derived
    double *pubA = isEnd ? endA : rowAd + 2 * ((3 % K) * AQ + PADQ + 64 + 3 / K);
    double *pubV = isEnd ? endV : oneV;
    const double pubC = isEnd ? endC : NN;
    const unsigned mLast = (foldLO && lane == 63) ? 0xffffffffu : mOwn[K - 1];
    static_assert(!FOLD || (K <= 2 && k1 == K - 1), "the folded publish block takes state 1 from its lane's last slot");

    const int64_t pair_base = P.win_pair_off[w] + (int64_t)(g - h0) * R;
    const int rs_base = P.read_seq_off[r0];
    const int64_t SL = (int64_t)P.read_seq_off[r1] - rs_base;
    const int64_t hpos_base = P.win_hpos_off[w] + (int64_t)(g - h0) * SL;
    const int nv = P.hap_var_off ? (P.hap_var_off[g + 1] - P.hap_var_off[g]) : 0;

    STAMP(0);
    // ---- bMid: ObservationModelFB::Init (ObservationModelFB.cpp:51-99) ----
    auto bmid_of = [&](int r, int L) -> int {
        int bMid;
        const uint32_t hapEnd = hapStart + (uint32_t)Hs;
        const uint32_t mReadStart = P.read_start[r];
        const uint32_t readEnd = mReadStart + (uint32_t)L - 1u;
        if ((P.read_flags[r] & 1) || mReadStart > hapEnd || readEnd < hapStart) {
            bMid = L / 2;
        } else {
            const uint32_t olStart = (hapStart > mReadStart) ? hapStart : mReadStart;
            const uint32_t olEnd = (hapEnd > readEnd) ? readEnd : hapEnd;
            const int mid = ((int)olEnd - (int)olStart) / 2 + (int)olStart;
            bMid = mid - (int)mReadStart;
        }
        if (P.bMid != -1) bMid = P.bMid;
        if (bMid < 0) bMid = 0;
        if (bMid >= L) bMid = L - 1;
        return bMid;
    };
    // a pair the reference throws "hapSize error." for (ObservationModelFB.cpp:47): status, no coverage flags
    auto mark_hapsize = [&](int ri, int first, int step) {
        const int64_t pair = pair_base + ri;
        if (first == 0) {
            P.out.status[pair] = DD_PAIR_HAPSIZE;
            P.out.ll[pair] = 0.0;
        }
        if (nv > 0) {
            const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv;
            for (int i = first; i < nv; i += step) {
                if (P.out.var_covered) P.out.var_covered[vb + i] = 0;
                if (P.out.var_fcov) P.out.var_fcov[vb + i] = 0;
            }
        }
    };
    // ======================= loop over this wave's reads =======================
    // The haplotype's workgroups share its reads in contiguous blocks; inside a workgroup the wavefronts pull the next read from a counter
    // in LDS when they are free.  G = 2: the window's reads are taken in chunks of DD_HALF_CHUNK; every workgroup of the haplotype orders
    // the chunk's reads of this launch's length class by (length bucket, bMid) — the two pairs of a wavefront run max(L1-1-bMid1, L2-1-bMid2) +
    // max(bMid1, bMid2) sweeps, so they should agree in both — and a pull takes two consecutive ranks.  Results do not depend on the order.
    const int nChunks = (G == 1) ? 1 : (R + DD_HALF_CHUNK - 1) / DD_HALF_CHUNK;
    for (int chunk0 = 0, ci = 0; ci < nChunks; ci++, chunk0 += DD_HALF_CHUNK) {
    int nch = R;                                   // reads this pass hands out (G = 2: the chunk's reads of this length class)
    if constexpr (G > 1) {
        const int nraw = (R - chunk0 < DD_HALF_CHUNK) ? R - chunk0 : DD_HALF_CHUNK;
        __syncthreads();                           // the previous chunk's order is no longer read
        if (tid == 0) *wq = 0;                     // every wave has left the previous chunk's queue; two barriers before the first pull
        for (int t = tid; t < nraw; t += nthr) {
            const int rr = r0 + chunk0 + t;
            const int L = P.read_seq_off[rr + 1] - P.read_seq_off[rr];
            uint32_t key = 0xffffffffu;            // not of this launch's length class: behind every read that is
            if (L >= P.len_min && L <= P.len_max) {
                if (!hap_ok) { if (split == 0) mark_hapsize(chunk0 + t, 0, 1); }
                else {
                    // Pairs of consecutive ranks run max(nInc1, nInc2) + max(nDec1, nDec2) sweeps (nDec = bMid, nInc = L - 1 - bMid): order by
                    // length in buckets of 8 bases, inside a bucket by bMid — ascending in even buckets, descending in odd ones, so that the
                    // ranks either side of a bucket boundary are close in bMid too.  Reads of one length: plain bMid order.
                    const uint32_t bucket = (uint32_t)(L - 1) >> 3, bm = (uint32_t)bmid_of(rr, L);
                    key = (bucket << 11) | ((bucket & 1u) ? 1023u - bm : bm);
                }
            }
            skey[t] = key;
        }
        __syncthreads();
        for (int t = tid; t < nraw; t += nthr) {
            const uint32_t k = skey[t];
            int rank = 0;
            for (int j = 0; j < nraw; j++) {
                const uint32_t kj = skey[j];
                rank += (kj < k || (kj == k && j < t)) ? 1 : 0;
            }
            sord[rank] = (uint16_t)t;
        }
        __syncthreads();
        int lo = 0, hi = nraw;                     // reads that take part = ranks in front of the first 0xffffffff key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (skey[sord[mid]] != 0xffffffffu) lo = mid + 1; else hi = mid;
        }
        nch = __builtin_amdgcn_readfirstlane(lo);
    }
    // this workgroup's share: a contiguous block of the reads (G = 2: of the pairs of consecutive ranks)
    const int nUnits = (nch + G - 1) / G, perSplit = (nUnits + nsp - 1) / nsp;
    const int unitLo = split * perSplit, unitHi = (unitLo + perSplit < nUnits) ? unitLo + perSplit : nUnits;
    for (;;) {
        int unit = 0;
        if (wlane == 0) unit = atomicAdd(wq, 1);
        unit = unitLo + __builtin_amdgcn_readfirstlane(unit);
        if (unit >= unitHi) break;
        const int rb = unit * G;
        // G = 2: the second pair of the last wavefront may be missing: its lanes run the code with an empty read and store nothing
        const bool live = (G == 1) || (rb + grp < nch);
        // (G = 2: ri, L and bMid are what a lane carries through the sweeps; r, pair and so are formed again behind them)
        int ri = (G == 1) ? rb : chunk0 + (live ? (int)sord[rb + grp] : (int)sord[rb]);
        int r = r0 + ri;
        int64_t pair = pair_base + ri;
        int so = P.read_seq_off[r];
        const int Lread = P.read_seq_off[r + 1] - so;
        const int L = live ? Lread : 0;

        if constexpr (G == 1) {
            if (L < P.len_min || L > P.len_max) continue;      // another length-class launch owns this read
            if (!hap_ok) { mark_hapsize(ri, lane, 64); continue; }
        }

        const int bMid = live ? bmid_of(r, L) : 0;
        // trip counts of the two passes; G = 2: the wavefront runs the longer of its two pairs' counts, the other pair's lanes wait masked
        const int nInc = L - 1 - bMid, nDec = bMid;
        int nIncW = nInc, nDecW = nDec;
        if constexpr (G > 1) {
            const int oi = __shfl_xor(nInc, 32), od = __shfl_xor(nDec, 32);
            nIncW = oi > nInc ? oi : nInc;
            nDecW = od > nDec ? od : nDec;
            nIncW = __builtin_amdgcn_readfirstlane(nIncW);
            nDecW = __builtin_amdgcn_readfirstlane(nDecW);
        }

        // ---- stage the read: base codes + emission logs (setupReadObservationPotentials :220-252) ----
        for (int b = lane; b < L; b += W) {
            const int qi = P.read_qidx[so + b];
            rdC[b] = shLut[(unsigned char)P.read_seq[so + b]];
            rdQ[b] = (unsigned char)qi;
            rdE[2 * b] = shQ[4 * qi];
            rdE[2 * b + 1] = shQ[4 * qi + 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // pads of the slice arrays (the join's near-tie replay may have used the buffer as scratch)
        for (int i = lane; i < 2 * PADQ * K; i += W) {
            const int kk = i / (2 * PADQ), j = i - kk * (2 * PADQ);
            rowA[kk * AQ + (j < PADQ ? j : W + j)] = make_double2(NEG_INF, 0.0);
        }
        STAMP(1);   // bMid + staging
        double a[K], in[K];           // current slice: "on base x" and "inserted at x"
        // ================= right -> middle: passMessageTwoInc for b = L-1..bMid+1 (:1576-1578, :1715-1773)
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool valid = (x0 + k) < numS;
            a[k] = valid ? 0.0 : NEG_INF;      // beta[L-1][*] = 0
            in[k] = valid ? 0.0 : NEG_INF;
        }
        if constexpr (SLIM) {
            int xb = x0;
            asm volatile("" : "+v"(xb));       // opaque: the loads below stay inside this read's code, in front of the pass that uses them
            load_inc_constants(xb);
        }
        {
            double cInc[LEAN ? 1 : K][LEAN ? 1 : D];   // lp_y(src)+Nn[src] for src = x+y  (:1730-1735)
            if constexpr (!LEAN) {
#pragma unroll
                for (int k = 0; k < K; k++) {
#pragma unroll
                    for (int y = 1; y <= D; y++) {
                        const int src = x0 + k + y;
                        const double Es = shE[src], Ns = shN[src];
                        const double lp = (y == 1) ? Ns : Es + (double)(y - 1) * II;
                        cInc[k][y - 1] = (src <= RO && y <= Dr) ? lp + Ns : NEG_INF;
                    }
                }
                if (foldLO) {
#pragma unroll
                    for (int y = 1; y <= D; y++) {
                        if (lane == 63) cInc[K - 1][y - 1] = y == 2 ? lLL : (y == 3 ? lFL : NEG_INF);
                        if (foldRO && lane == laneRO) cInc[K - 1][y - 1] = y == 2 ? 0.0 : NEG_INF;
                    }
                }
            }
            if (foldLO && lane == 63) { a[K - 1] = 0.0; in[K - 1] = 0.0; }      // beta[L-1][LO] = beta[L-1][numS] = 0, at LO's position in this pass
            auto cinc = [&](int k, int y) -> double {       // y = 1..D
                if constexpr (LEAN && D > 12) {                 // D = 32 build: from the haplotype's tables, every read base (correct, not fast)
                    const int src = x0 + k + y;
                    if (src > RO || y > Dr) return NEG_INF;
                    const double Es = shE[src], Ns = shN[src];
                    const double lp = (y == 1) ? Ns : Es + (double)(y - 1) * II;
                    return lp + Ns;
                } else if constexpr (LEAN) return shC[(k * D + y - 1) * W + lane]; else return cInc[k][y - 1];
            };
            // G = 2: both pairs run the longer pair's trip count, the lanes of the pair that is done wait masked.  (One loop for both forms,
            // no lambda for the body: with the body in a lambda the K = 2 / D = 11 scratch build spilled 38 registers instead of 6.)
            for (int it = 0, b = L - 1; (G == 1) ? (b > bMid) : (it < nIncW); it++, b--) {
                if (G == 1 || it < nInc) {
                const double eq = rdE[2 * b], uq = rdE[2 * b + 1];
                const int col = rdC[b];
                // (D = 32 build: only the lane's own positions and the one behind them are staged; the jump candidates read their source from
                // the LDS slice inside a run-time loop — 2 x (D + K) doubles per lane was 700-1,100 spilled registers per kernel)
                constexpr int NVI = BIGD ? K + 1 : D + K;
                double v[NVI], ov[NVI];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const unsigned mk = (foldLO && k == K - 1) ? mLast : mOwn[k];
                    ov[k] = ((mk >> col) & 1u) ? eq : uq;
                    v[k] = a[k];
                    rowA[k * AQ + PADQ + lane] = make_double2(a[k], ov[k]);
                }
                if constexpr (foldLO) {                            // what the end states' generic candidates read (after the owners' own slots)
                    // 8-byte stores (RO's two slots are ordinary positions whose owner has just stored its own pair there: both halves are
                    // rewritten)
                    if (pubFlag) { pubA[0] = a[K - 1]; pubA[1] = ov[K - 1]; pubV[0] = ov[K - 1] + a[K - 1]; pubV[1] = pubC; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int j = 0; j < NVI - K; j++) {                // states x0+K .. x0+K+D-1
                    const double2 t = rowA[((K + j) % K) * AQ + PADQ + lane + (K + j) / K];   // state x0+K+j
                    v[K + j] = t.x;
                    ov[K + j] = t.y;
                }
                double na[K], ni[K];
                btacc_t btb[K];
                double bestY[BIGD ? K : 1];
                btacc_t chY[BIGD ? K : 1];
                if constexpr (BIGD) {
                    // the jump candidates y = 1..Dr of every position, y in a run-time loop (same order per position as the unrolled form:
                    // y ascending, "take iff > best + EPS"; candidates beyond the real D are not formed instead of being -inf)
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        bestY[k] = (cinc(k, 1) + v[k + 1]) + ov[k + 1];
                        chY[k] = (btacc_t)1;
                    }
#pragma unroll 1
                    for (int y = 2; y <= Dr; y++) {
                        const double yII = (double)(y - 1) * II;
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            const int j = k + y, src = x0 + j;
                            const double2 t = rowA[(j % K) * AQ + PADQ + lane + j / K];       // state x0+k+y
                            double c = NEG_INF;
                            if (src <= RO) { const double Ns = shN[src]; c = (shE[src] + yII) + Ns; }
                            const double val = (c + t.x) + t.y;
                            const bool take = val > bestY[k] + DD_EPS;
                            bestY[k] = take ? val : bestY[k];
                            chY[k] = take ? (btacc_t)y : chY[k];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < K; k++) {
                    double best;
                    const int sh = (D <= 7) ? k * BP::PB : 0;
                    btacc_t ch;
                    if constexpr (BIGD) { best = bestY[k]; ch = chY[k]; }
                    else {
                    best = (cinc(k, 1) + v[k + 1]) + ov[k + 1];   // lp+lpn+beta+obs (:1735), y = 1
                    ch = (btacc_t)1 << sh;
#pragma unroll
                    for (int y = 2; y <= D; y++) {
                        const double val = (cinc(k, y) + v[k + y]) + ov[k + y];
                        const bool take = val > best + DD_EPS;     // newIdx > destIdx: branch 1 only
                        best = take ? val : best;
                        ch = take ? ((btacc_t)y << sh) : ch;
                    }
                    }
                    {
                        const double val = (eq + in[k]) + eInc[k]; // to inserted state numS+x (:1746-1749)
                        const bool take = val > best + DD_EPS;
                        best = take ? val : best;
                        ch = take ? (btacc_t)0 : ch;
                    }
                    na[k] = best;
                    const double d = (eq + in[k]) + II;            // (:1754-1758)
                    const double val = (ov[k + 1] + v[k + 1]) + NI;   // src = x+1 (:1763-1767)
                    const bool take = val >= d;
                    ni[k] = dmax(d, val);
                    btb[k] = ch | (take ? ((btacc_t)(1u << BP::CB) << sh) : (btacc_t)0);
                }
                if constexpr (!foldLO) if (lane == 0) {             // x = 0 (:1720-1722, :1746-1749, :1762)
                    double best = ((eq + a[0]) + lLL) + NN;         // idx 0
                    unsigned code = 0;
                    const double c2 = ((ov[1] + v[1]) + lFL) + NN;  // idx 1
                    const bool t2 = c2 > best + DD_EPS;
                    best = t2 ? c2 : best;
                    code = t2 ? 1u : code;
                    const double c3 = (eq + in[0]) + E_1;           // idx numS
                    const bool t3 = c3 > best + DD_EPS;
                    best = t3 ? c3 : best;
                    code = t3 ? 2u : code;
                    na[0] = best;
                    const double d = (eq + in[0]) + II;
                    const double val = (eq + a[0]) + NI;
                    const bool take = val >= d;
                    ni[0] = dmax(d, val);
                    btb[0] = (btacc_t)(code | (take ? (1u << BP::CB) : 0u));
                }
                if (!foldRO && lane == laneRO) {                    // x = RO (:1741-1742, :1750, :1763-1767)
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        if (k == kRO) {
                            double best = (eq + a[k]) + Nn_RO;      // idx RO
                            const double c2 = eq + in[k];           // idx numS+RO
                            const bool t2 = c2 > best + DD_EPS;
                            best = t2 ? c2 : best;
                            na[k] = best;
                            const double d = (eq + in[k]) + II;
                            const double val = (eq + a[k]) + NI;
                            const bool take = val >= d;
                            ni[k] = dmax(d, val);
                            btb[k] = (btacc_t)((t2 ? 1u : 0u) | (take ? (1u << BP::CB) : 0u)) << ((D <= 7) ? k * BP::PB : 0);
                        }
                    }
                }
                btword_t word = 0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    a[k] = na[k];
                    in[k] = ni[k];
                    word |= (D <= 7) ? (btword_t)btb[k] : (btword_t)((btword_t)btb[k] << (k * BP::PB));
                }
                bt[b * 64 + wlane] = word;                          // btb[b-1] stored at row b
                }
            }
        }
        STAMP(3);   // Inc passes
        double be_a[K], be_i[K];           // beta[bMid]
#pragma unroll
        for (int k = 0; k < K; k++) {
            be_a[k] = a[k]; be_i[k] = in[k];
            if constexpr (STASH) { beS[(2 * k) * 64 + wlane] = a[k]; beS[(2 * k + 1) * 64 + wlane] = in[k]; }   // parked until the join (the same lane writes and reads)
        }
        if (foldLO) {                      // LO's beta back to position 0, where the join and the left->middle pass have it
            const double tA = __shfl(a[K - 1], 63), tI = __shfl(in[K - 1], 63);
            if (lane == 0) { be_a[0] = tA; be_i[0] = tI; }
            if (lane == 63) { be_a[K - 1] = NEG_INF; be_i[K - 1] = NEG_INF; }
        }

        // The left->middle chain of RO / numS+RO is a sink: nothing else in that pass reads it, and it only matters
        // if a state with prior -100 wins the join.  Every alpha, beta, emission and transition term is <= 0, so those
        // two join values are <= -100; if the maximum over the other states exceeds -99 under both priors, neither the
        // EPS scan nor the traceback can touch them and the pass without RO (about 20 % fewer instructions per read
        // base) is exact.  Otherwise (very unlikely reads) the pass is redone with RO evaluated.
        double ll, llHMQ, llOff, llOn;
        int mapRMQ, mapHMQ;
        for (int pass = 0; pass < 2; pass++) {
        const bool with_ro = (pass == 1) || P.always_ro;
        // ================= left -> middle: passMessageTwoDec for b = 1..bMid (:1573-1575, :1775-1829)
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool valid = (x0 + k) < numS;
            a[k] = valid ? 0.0 : NEG_INF;      // alpha[0][*] = 0 (:335-338)
            in[k] = valid ? 0.0 : NEG_INF;
        }
        if constexpr (SLIM) {
            int xb = x0;
            asm volatile("" : "+v"(xb));
            load_dec_constants(xb);
        }
        {
            double lpDec[LEAN ? 1 : K][LEAN ? 1 : D];   // y=1: Nn[x]; y>=2: E[x]+(y-1)*II   (:1786-1791)
            if constexpr (!LEAN) {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    lpDec[k][0] = lpn[k];
#pragma unroll
                    for (int y = 2; y <= D; y++) lpDec[k][y - 1] = (y > Dr) ? NEG_INF : eIn[k] + (double)(y - 1) * II;
                }
            }
            for (int b = 1; (G == 1) ? (b <= bMid) : (b <= nDecW); b++) {
                if (G == 1 || b <= nDec) {
                const double eq = rdE[2 * (b - 1)], uq = rdE[2 * (b - 1) + 1];
                const int col = rdC[b - 1];
                // (DW = how many states below the lane's own are staged: D, or — D = 32 build — the one state the y = 1 edge and RO's block need;
                // the array index of "state x0 + k - y" is DW + k - y)
                constexpr int DW = BIGD ? 1 : D;
                double v[DW + K], ov[DW + K];
                // publish slice b-1 (value + this state's emission for read base b-1) for the neighbours
#pragma unroll
                for (int k = 0; k < K; k++) {
                    ov[DW + k] = ((mOwn[k] >> col) & 1u) ? eq : uq;
                    v[DW + k] = a[k];
                    rowA[k * AQ + PADQ + lane] = make_double2(a[k], ov[DW + k]);
                    rowI[1 + x0 + k] = in[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int j = 0; j < DW; j++) {                     // states x0-DW .. x0-1
                    const int off = j - DW, kk = ((off % K) + K) % K, q = (off - kk) / K;     // state x0-DW+j = (lane+q)*K + kk
                    const double2 t = rowA[kk * AQ + PADQ + lane + q];
                    v[j] = t.x;
                    ov[j] = t.y;
                }
                const double im1 = rowI[x0];                        // I[x0-1]
                double na[K], ni[K];
                btacc_t btb[K];
                double bestY[BIGD ? K : 1];
                btacc_t chY[BIGD ? K : 1];
                if constexpr (BIGD) {
                    // jump candidates in a run-time loop over y (see the right->middle pass): source state x0 + k - y from the LDS slice
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        bestY[k] = ((ov[DW + k - 1] + lpn[k]) + v[DW + k - 1]) + lpn[k];
                        chY[k] = (btacc_t)1;
                    }
#pragma unroll 1
                    for (int y = 2; y <= Dr; y++) {
                        const double yII = shY[y - 1];              // (y-1)*II
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            const int jj = k - y + K * PADQ;        // >= 0: y <= D <= K * PADQ
                            const double2 t = rowA[(jj % K) * AQ + lane + jj / K];            // state x0+k-y (pads hold -inf below state 0)
                            const double lp = eIn[k] + yII;
                            const double val = ((t.y + lp) + t.x) + lpn[k];
                            const bool take = val >= bestY[k];
                            bestY[k] = dmax(bestY[k], val);
                            chY[k] = take ? (btacc_t)y : chY[k];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < K; k++) {
                    double best;
                    const int sh = (D <= 7) ? k * BP::PB : 0;     // field of this position in the packed word (D<=7: codes are
                                                                  // pre-shifted; larger codes would stop being inline constants)
                    btacc_t ch;
                    if constexpr (BIGD) { best = bestY[k]; ch = chY[k]; }
                    else {
                    best = ((ov[DW + k - 1] + lpn[k]) + v[DW + k - 1]) + lpn[k];         // (:1793), y = 1: lp = lpn
                    ch = (btacc_t)1 << sh;
#pragma unroll
                    for (int y = 2; y <= D; y++) {
                        double lp;
                        if constexpr (LEAN) lp = eIn[k] + shY[y - 1]; else lp = lpDec[k][y - 1];
                        const double val = ((ov[DW + k - y] + lp) + v[DW + k - y]) + lpn[k];
                        const bool take = val >= best;             // newIdx < destIdx: either branch of updateMax
                        best = dmax(best, val);
                        ch = take ? ((btacc_t)y << sh) : ch;
                    }
                    }
                    {
                        const double ip = (k == 0) ? im1 : in[k > 0 ? k - 1 : 0];
                        const double val = (eq + ip) + eIn[k];     // from inserted state numS+x-1 (:1807-1811)
                        const bool take = val > best + DD_EPS;     // newIdx > destIdx: branch 1 only
                        best = take ? val : best;
                        ch = take ? (btacc_t)0 : ch;
                    }
                    na[k] = best;
                    const double d = (eq + in[k]) + II;            // stay inserted (:1816-1820)
                    const double val = (ov[DW + k] + a[k]) + niDec[k];   // open insertion after x (:1823-1826)
                    const bool take = val >= d;
                    ni[k] = dmax(d, val);
                    btb[k] = ch | (take ? ((btacc_t)(1u << BP::CB) << sh) : (btacc_t)0);
                }
                if (lane == 0) {                                    // x = 0 (:1798-1799)
                    na[0] = (eq + a[0]) + NN;
                    btb[0] = 0;
                }
                if (with_ro && lane == laneRO) {                    // x = RO (:1780-1782, :1804-1805)
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        if (k == kRO) {
                            const double aHs = v[DW + k - 1], oHs = ov[DW + k - 1];
                            const double iHs = (k == 0) ? im1 : in[k > 0 ? k - 1 : 0];
                            // candidate order and indices: RO, Hs (< RO), numS+RO (largest), numS+Hs
                            double best = ((eq + a[k]) + lLL) + NN;
                            unsigned code = 0;
                            const double c2 = ((oHs + aHs) + lFL) + NN;
                            const bool t2 = c2 >= best;            // smaller index: either branch
                            best = dmax(best, c2);
                            code = t2 ? 1u : code;
                            const double c3 = ((eq + in[k]) + lLL) + E_RO;
                            const bool t3 = c3 > best + DD_EPS;    // larger index: branch 1 only
                            best = t3 ? c3 : best;
                            code = t3 ? 2u : code;
                            const double c4 = ((eq + iHs) + lFL) + E_Hs;
                            const bool t4 = (c4 > best + DD_EPS) || (t3 && c4 >= best);   // numS+Hs < numS+RO only
                            best = t4 ? c4 : best;
                            code = t4 ? 3u : code;
                            na[k] = best;
                            btb[k] = (btb[k] & ((btacc_t)(1u << BP::CB) << ((D <= 7) ? k * BP::PB : 0))) | ((btacc_t)code << ((D <= 7) ? k * BP::PB : 0));
                        }
                    }
                }
                btword_t word = 0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    a[k] = na[k];
                    in[k] = ni[k];
                    word |= (D <= 7) ? (btword_t)btb[k] : (btword_t)((btword_t)btb[k] << (k * BP::PB));
                }
                bt[b * 64 + wlane] = word;
                }
            }
        }
        STAMP(2);   // Dec passes
        // ================= join at bMid: calcLikelihoodFromLastSlice (:1075-1144) + computeBMidPrior (:268-305)
        {
            const double eq = rdE[2 * bMid], uq = rdE[2 * bMid + 1];
            const int col = rdC[bMid];
            if constexpr (G > 1) { asm volatile("" : "+v"(ri)); r = r0 + ri; }
            const int mqi = P.read_mqidx[r];
            const double prOff0 = T[T_MAPQ + 4 * mqi + 0], prOff1 = T[T_MAPQ + 4 * mqi + 1];
            const double prOn0 = T[T_MAPQ + 4 * mqi + 2], prOn1 = T[T_MAPQ + 4 * mqi + 3];
            const double hqOff0 = T[TC_HMQ + 0], hqOff1 = T[TC_HMQ + 1], hqOn0 = T[TC_HMQ + 2], hqOn1 = T[TC_HMQ + 3];
            const double roPrior = with_ro ? -100.0 : NEG_INF;     // prior[RO] = -100 (:299); -inf while RO is not evaluated
            double vA[K], vI[K], hA[K], hI[K];
            double *scanA = reinterpret_cast<double *>(rowA), *scanI = scanA + NP;   // near-tie replay scratch (the slice buffer is free now)
            double on = NEG_INF;
            // Insert-size prior of a paired read whose mate is mapped on the same chromosome (mapUnmappedReads,
            // computeBMidPrior :279-292): pinsert[x] = log P_library(|distance between the read start implied by "base bMid
            // sits on haplotype base x" and the mate|), pinsert[LO] = log of the library's 95th-percentile probability.
            // prior[on, x] = (pinsert[x] + log(1-pOff)) + logpIns;  prior[LO] = (log(pOff) + logpIns) + pinsert[LO]  (:296-303)
            bool usePin = false;
            int pinBase = 0, pinMax = 1, pinD0 = 0;
            if (P.read_mate_pos) {
                const int fl = P.read_flags[r], mlen = P.read_mate_len[r];
                usePin = (fl & DD_READ_PAIRED) && !(fl & DD_READ_MATE_UNMAPPED) && mlen != -1 && (fl & DD_READ_MATE_SAME_TID);
                if (usePin) {
                    const int lib = P.read_lib[r];
                    pinBase = P.lib_off[lib];
                    pinMax = P.lib_off[lib + 1] - pinBase;
                    // all int arithmetic, as in the reference (hapStart, bMid, readSize are ints there)
                    pinD0 = (fl & DD_READ_MATE_REVERSE) ? (int)hapStart - bMid - (P.read_mate_pos[r] + mlen)
                                                        : (int)hapStart + L - bMid - P.read_mate_pos[r];
                }
            }
            const double lpOffR = T[T_MAPQ2 + 2 * mqi], lpOnR = T[T_MAPQ2 + 2 * mqi + 1];   // log(pOff), log(1-pOff)
            const double lpOffH = T[TC_PINS + 1], lpOnH = T[TC_PINS + 2];
            const double lIns0 = T[TC_PINS + 0], lIns1 = T[TC_IN];                          // logpIns for i = 0, 1 (:297)
            const double pin0 = usePin ? P.lib_log95[P.read_lib[r]] : 0.0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int x = x0 + k;
                const double o = ((mOwn[k] >> col) & 1u) ? eq : uq;
                const double bA = STASH ? beS[(2 * k) * 64 + wlane] : be_a[k], bI = STASH ? beS[(2 * k + 1) * 64 + wlane] : be_i[k];
                const double baseA = (a[k] + o) + bA;                   // alpha + obs + beta (:1098)
                const double baseI = (in[k] + eq) + bI;
                double pOn0 = prOn0, pOn1 = prOn1, qOn0 = hqOn0, qOn1 = hqOn1, pOf0 = prOff0, pOf1 = prOff1, qOf0 = hqOff0, qOf1 = hqOff1;
                if (usePin) {
                    int d = pinD0 + x;
                    d = d < 0 ? -d : d;                                 // Library::getProb (Library.hpp:60-64)
                    d = d >= pinMax ? pinMax - 1 : d;
                    const double pin = (x >= 1 && x <= Hs) ? P.lib_logprob[pinBase + d] : 0.0;
                    pOn0 = (pin + lpOnR) + lIns0; pOn1 = (pin + lpOnR) + lIns1;
                    qOn0 = (pin + lpOnH) + lIns0; qOn1 = (pin + lpOnH) + lIns1;
                    pOf0 = (lpOffR + lIns0) + pin0; pOf1 = (lpOffR + lIns1) + pin0;
                    qOf0 = (lpOffH + lIns0) + pin0; qOf1 = (lpOffH + lIns1) + pin0;
                }
                vA[k] = baseA + ((x == 0) ? pOf0 : (x == RO ? roPrior : pOn0));    // read's mapping quality
                vI[k] = baseI + ((x == 0) ? pOf1 : (x == RO ? roPrior : pOn1));
                hA[k] = baseA + ((x == 0) ? qOf0 : (x == RO ? roPrior : qOn0));    // mapQual = 1-1e-10 (:1093)
                hI[k] = baseI + ((x == 0) ? qOf1 : (x == RO ? roPrior : qOn1));
                if (x >= 1 && x <= Hs) {                                   // (:1106-1107) plain max
                    on = vA[k] > on ? vA[k] : on;
                    on = vI[k] > on ? vI[k] : on;
                }
            }
            llOn = group_max<G>(on);
            llOff = vA[0] > vI[0] ? vA[0] : vI[0];                          // states 0 and numS (:1104-1105); lane 0 only
            slice_argmax<K, G>(vA, vI, x0, numS, grp, scanA, scanI, ll, mapRMQ);
            slice_argmax<K, G>(hA, hI, x0, numS, grp, scanA, scanI, llHMQ, mapHMQ);
        }
        if constexpr (G == 1) {
            if (with_ro || (ll > -99.0 && llHMQ > -99.0)) break;
        } else {
            // the pass with RO evaluated is the reference's computation for any pair: if one of the two pairs needs it, both redo
            if (with_ro || __ballot(live && !(ll > -99.0 && llHMQ > -99.0)) == 0ull) break;
        }
        }
        STAMP(4);   // join
        if constexpr (G > 1) {
            asm volatile("" : "+v"(ri));
            r = r0 + ri; pair = pair_base + ri; so = P.read_seq_off[r];
        }
        const int xR = mapRMQ % numS, xH = mapHMQ % numS;
        const bool offHap = (xR == 0 || xR == RO);
        const bool offHapHMQ = (xH == 0 || xH == RO);

        // ================= traceback: computeMAPState (:1148-1165).  The two chains (towards base 0 through
        // btf, towards base L-1 through btb) are independent: step them in the same iteration so their LDS
        // round trips overlap.  Every lane walks redundantly (wave-uniform); lane 0 records.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (GBT) {
            // The tile was written by all 64 lanes and is now read through another lane's address — by the SAME wave on
            // the same CU.  Workgroup scope is enough (the CU's vector L1 is write-through and coherent for its own
            // stores); it waits for the stores (vmcnt) without the L2 write-back an agent-scope release costs on a
            // multi-XCD part.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        {
            // A MAP path is mostly RUNS: the read walks down a diagonal of the haplotype (every step "from the base before",
            // ch == 1), or sits left / right of it (LO, RO: "stay").  A run is recognised 64 rows at a time: lane j fetches
            // the back-pointer field the path would use j steps ahead IF the run continued (row b -/+ j, position x -/+ j),
            // a ballot gives the run length, and the 64 states are stored lane-parallel.  Only the events between runs (an
            // indel, entering or leaving the haplotype) take the scalar, one-row-at-a-time decode — a typical read needs 2-4
            // of those instead of ~L dependent steps (the serial walk was 19 % of the kernel's wave time).
            constexpr unsigned chmask = (1u << BP::CB) - 1u, fmask = (1u << BP::PB) - 1u;
            auto fieldAt = [&](int row, int x) -> unsigned {               // per-lane row / position: the PB bits of that state
                const int src = x / K, sh = (x - src * K) * BP::PB;
                return (unsigned)(bt[row * 64 + grp * W + src] >> sh) & fmask;
            };
            // G = 1: the walk is wave-uniform and runs on the scalar unit; G = 2: every lane of a group holds its pair's walk
            const int s0 = (G == 1) ? __builtin_amdgcn_readfirstlane(mapHMQ) : mapHMQ;
            auto group_lane_value = [&](unsigned v, int l) -> unsigned {      // v of the group's lane l (l uniform in the group)
                if constexpr (G == 1) return (unsigned)__builtin_amdgcn_readlane((int)v, l);
                else return (unsigned)__shfl((int)v, grp * W + l);
            };
            auto group_uniform = [&](unsigned v) -> unsigned {                // v is equal across the group's lanes
                if constexpr (G == 1) return (unsigned)__builtin_amdgcn_readfirstlane((int)v); else return v;
            };
            constexpr unsigned long long fullRun = (G == 1) ? ~0ull : 0xffffffffull;
            if (lane == 0) ms[bMid] = (int16_t)s0;
            // ---- towards base 0: mapState[b-1] = btf[b][mapState[b]] ----
            {
                int ins = s0 >= numS ? 1 : 0, x = s0 - (ins ? numS : 0), b = bMid;
                while (b > 0) {
                    unsigned f;
                    if (!ins && x == 0) {                                  // LO is absorbing in this direction (:1798-1799)
                        for (int j = lane; j < b; j += W) ms[j] = 0;
                        break;
                    }
                    if (!ins && x != RO) {                                 // diagonal run: rows b, b-1, ...; positions x, x-1, ...
                        const int maxrun = b < W ? b : W;
                        const bool valid = lane < maxrun && x - lane >= 1;
                        f = valid ? fieldAt(b - lane, x - lane) : 0u;
                        const unsigned long long m = group_ballot<G>(valid && (f & chmask) == 1u, grp);
                        const int run = (m == fullRun) ? W : __builtin_ctzll(~m);
                        if (lane < run) ms[b - 1 - lane] = (int16_t)(x - 1 - lane);
                        b -= run; x -= run;
                        if (b == 0) break;
                        if (run == maxrun || x == 0) continue;             // next 64 rows / reached LO
                        f = group_lane_value(f, run);                      // the row that ended the run: decode it below
                    } else {
                        f = group_uniform(fieldAt(b, x));
                    }
                    // one row, any state.  on base x: ch = jump length y (from x-y), 0 = from the inserted state numS+x-1;
                    // RO: 0 RO, 1 Hs, 2 numS+RO, 3 numS+Hs.  inserted at x: bit set = entered from "on base x", else stays
                    const int ch = (int)(f & chmask), ib = (int)(f >> BP::CB);
                    const int z = ch == 0 ? 1 : 0;
                    int gx = x - ch - z, gi = z;
                    if (x == RO) { gx = RO - (ch & 1); gi = ch >> 1; }
                    if (ins) { gx = x; gi = ib ^ 1; }
                    x = gx; ins = gi;
                    if (lane == 0) ms[b - 1] = (int16_t)(ins ? numS + x : x);
                    b--;
                }
            }
            // ---- towards base L-1: mapState[b+1] = btb[b][mapState[b]], stored at row b+1 ----
            {
                int ins = s0 >= numS ? 1 : 0, x = s0 - (ins ? numS : 0), b = bMid;
                while (b < L - 1) {
                    unsigned f;
                    // where this pass kept the state's back-pointer field, and the code that means "stays": LO folded into the generic
                    // code sits at position NP-1 and stays with y = 2 (as does a folded RO); the one-lane blocks use code 0
                    const int xf = (foldLO && x == 0) ? NP - 1 : x;
                    const unsigned stayCode = (x == 0 ? foldLO : foldRO) ? 2u : 0u;
                    if (!ins) {
                        const int left = L - 1 - b, maxrun = left < W ? left : W;
                        const bool stay = (x == 0 || x == RO);             // LO / RO
                        const bool valid = lane < maxrun && (stay || x + lane <= Hs);
                        f = valid ? fieldAt(b + 1 + lane, stay ? xf : x + lane) : 0u;
                        const unsigned long long m = group_ballot<G>(valid && (f & chmask) == (stay ? stayCode : 1u), grp);
                        const int run = (m == fullRun) ? W : __builtin_ctzll(~m);
                        if (lane < run) ms[b + 1 + lane] = (int16_t)(stay ? x : x + 1 + lane);
                        b += run;
                        if (!stay) x += run;
                        if (b >= L - 1) break;
                        if (run == maxrun) continue;
                        if (!stay && x > Hs) continue;                     // the diagonal ran into RO: next round handles it as a stay run
                        f = group_lane_value(f, run);
                    } else {
                        f = group_uniform(fieldAt(b + 1, xf));
                    }
                    // on base x: ch = jump length y (to x+y), 0 = to the inserted state numS+x; LO: 0 LO, 1 base 1, 2 numS;
                    // RO: 0 RO, else numS+RO.  inserted at x: bit set = leaves to min(x+1, RO) (LO's stays LO), else stays.
                    // Folded end states carry generic codes: 0 = to the inserted state, 2 = stays, LO's 3 = base 1.
                    const int ch = (int)(f & chmask), ib = (int)(f >> BP::CB);
                    const int z = ch == 0 ? 1 : 0;
                    int gx = x + ch, gi = z;
                    if (x == RO) { gx = RO; gi = foldRO ? z : (z ^ 1); }
                    if (x == 0) {
                        if (foldLO) { gx = ch == 3 ? 1 : 0; gi = z; }
                        else { gx = ch & 1; gi = ch >> 1; }
                    }
                    if (ins) {
                        int lx = x + 1 > RO ? RO : x + 1;
                        if (x == 0) lx = 0;
                        gx = ib ? lx : x;
                        gi = ib ^ 1;
                    }
                    x = gx; ins = gi;
                    if (lane == 0) ms[b + 1] = (int16_t)(ins ? numS + x : x);
                    b++;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        STAMP(5);   // traceback
        // ================= reportVariants (:1351-1475): hpos + QC counters, lane-parallel over read bases ====
        int nIndel = 0, nMis = 0, nBQT = 0, nmmBQT = 0, nMML = 0, nMMR = 0;     // wave-uniform (ballot counts)
        int firstB = 0x7fffffff, lastB = -1;                                   // per lane, reduced below
        double mLogBQ = 0.0;
        const double thr = T[TC_BQT];
        int16_t *hp_out = P.out.hpos ? P.out.hpos + hpos_base + (so - rs_base) : nullptr;
        for (int b0 = 0; b0 < L; b0 += W) {
            const int b = b0 + lane;
            double cb = 0.0;
            bool pIndel = false, pDel = false, pMis = false, pBQT = false, pmmBQT = false, pL = false, pR = false;
            if (b < L) {
                const int s = ms[b];
                const bool ins = s >= numS;
                const int x = ins ? s - numS : s;
                int hp;
                if (x == 0) hp = DD_HPOS_LO;
                else if (x == RO) hp = DD_HPOS_RO;
                else if (ins) {
                    hp = DD_HPOS_INS_KEY0 - x;                                  // inserted base, carrying its key (pos = x, :1380)
                    pIndel = (b == 0 || ms[b - 1] < numS);                      // start of an insertion run (:1379-1394)
                } else {
                    hp = x - 1;
                    firstB = hp < firstB ? hp : firstB;
                    lastB = hp > lastB ? hp : lastB;
                    const int qi = rdQ[b];
                    const double q = shQ[4 * qi + 3];
                    const bool hiq = q > thr;
                    if (hiq) { pBQT = true; cb = shQ[4 * qi + 2]; }             // (:1404-1407)
                    if ((int)rdC[b] != (int)sc[x]) {                            // read.seq[b]!=hap.seq[s-1] (:1410)
                        pmmBQT = hiq;
                        pL = b < 6;
                        pR = b > L - 6;
                        pMis = q > 0.95;
                    }
                    if (b < L - 1) {
                        const int ns = ms[b + 1];
                        pDel = (ns < numS && ns - s > 1);                       // deletion (:1437-1453)
                    }
                }
                if (hp_out) hp_out[b] = (int16_t)hp;
            }
            nIndel += __popcll(group_ballot<G>(pIndel, grp)) + __popcll(group_ballot<G>(pDel, grp));
            nMis += __popcll(group_ballot<G>(pMis, grp));
            nmmBQT += __popcll(group_ballot<G>(pmmBQT, grp));
            nMML += __popcll(group_ballot<G>(pL, grp));
            nMMR += __popcll(group_ballot<G>(pR, grp));
            const unsigned long long bq = group_ballot<G>(pBQT, grp);
            nBQT += __popcll(bq);
            // mLogBQ: the reference adds log10(1-q) base by base in read order (:1404-1407); fp64 + is not
            // associative, so the sum runs serially in that order (adding +0.0 for skipped bases is exact).  The terms
            // go through the (now idle) emission staging buffer: every lane reads term i from the same LDS address
            // (a broadcast), so a term costs one v_add_f64 — no cross-lane moves.
            if (bq) {
                double *cbuf = rdE;                       // [2 * Lmax] doubles: lane < L - b0 <= Lmax fits
                if (b < L) cbuf[lane] = cb;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int first = __ffsll((long long)bq) - 1, last = 63 - __clzll((long long)bq);   // +0.0 terms outside change nothing
                int i = first;
                for (; i + 8 <= last + 1; i += 8) {
                    double t[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) t[j] = cbuf[i + j];
#pragma unroll
                    for (int j = 0; j < 8; j++) mLogBQ += t[j];
                }
                for (; i <= last; i++) mLogBQ += cbuf[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
#pragma unroll
        for (int off = 32 / G; off >= 1; off >>= 1) {
            const int f = __shfl_xor(firstB, off), l2 = __shfl_xor(lastB, off);
            firstB = f < firstB ? f : firstB;
            lastB = l2 > lastB ? l2 : lastB;
        }
        if (firstB == 0x7fffffff) firstB = -1;
        STAMP(6);   // hpos + counters + mLogBQ

        // hapIndelCovered / hapSNPCovered (:1465-1472; AlignedVariant::isCovered Variant.hpp:125-128)
        const int nvl = live ? nv : 0;                 // (G = 2: a missing second pair stores nothing)
        if (P.out.var_covered && nv > 0) {
            const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv;
            for (int i = lane; i < nvl; i += W) {
                const int sR = P.hap_var[2 * (P.hap_var_off[g] + i)];
                const int eR = P.hap_var[2 * (P.hap_var_off[g] + i) + 1];
                P.out.var_covered[vb + i] = (firstB + P.padCover <= sR && lastB - P.padCover >= eR) ? 1 : 0;
            }
        }

        // filterHaplotypes' per-read coverage test of the haplotype's own indels (next row N1; reference
        // DInDel.cpp:1951-2062): a selected read (!offHapHMQ && numIndels==0) has consecutive hpos values, so the set of
        // haplotype bases it covers inside [left,right] is the overlap with [firstBase,lastBase].
        if (P.out.var_fcov && P.hap_var_flank && nv > 0) {
            const int64_t vb = P.win_varcov_off[w] + (int64_t)(P.hap_var_off[g] - P.hap_var_off[h0]) * R + (int64_t)ri * nv;
            const bool sel = !offHapHMQ && nIndel == 0;
            for (int i = 0; i < nvl; i++) {
                const int32_t *fl = P.hap_var_flank + 3 * (size_t)(P.hap_var_off[g] + i);
                const int left = fl[0] - P.padCover, right = fl[1] + P.padCover, kind = fl[2];
                int cov = 0;
                if (sel && kind != 0) {
                    int nmm = 0;
                    for (int b0 = 0; b0 < L; b0 += W) {
                        const int b = b0 + lane;
                        bool mm = false;
                        if (b < L) {
                            const int s = ms[b];
                            if (s >= 1 && s <= Hs) {                          // on a haplotype base (no inserted states here)
                                const int hb = s - 1;
                                const int hc = sc[s];
                                mm = hb >= left && hb <= right && (int)rdC[b] != hc && (kind == 2 || hc != DD_SYM_N);   // 'N' exempt for DEL (:1992)
                            }
                        }
                        nmm += __popcll(group_ballot<G>(mm, grp));
                    }
                    const int lo = firstB > left ? firstB : left, hi = lastB < right ? lastB : right;
                    const int csize = (firstB >= 0 && hi >= lo) ? hi - lo + 1 : 0;
                    cov = (csize >= right - left + 1 && nmm <= P.maxMismatch) ? 1 : 0;
                }
                if (lane == 0) P.out.var_fcov[vb + i] = (uint8_t)cov;
            }
        }

        if (lane == 0 && live) {
            int status = DD_PAIR_OK;
            if (ll > 0.1) status = DD_PAIR_LLPOS;                         // DInDel.cpp:1722
            else if (ll != ll || ll == NEG_INF || ll == -NEG_INF) status = DD_PAIR_NAN;   // DInDel.cpp:1732
            P.out.ll[pair] = ll;
            P.out.status[pair] = status;
            if (P.out.llOn) P.out.llOn[pair] = llOn;
            if (P.out.llOff) P.out.llOff[pair] = llOff;
            if (P.out.mLogBQ) P.out.mLogBQ[pair] = mLogBQ;
            if (P.out.offHap) P.out.offHap[pair] = offHap ? 1 : 0;
            if (P.out.offHapHMQ) P.out.offHapHMQ[pair] = offHapHMQ ? 1 : 0;
            if (P.out.numIndels) P.out.numIndels[pair] = (int16_t)nIndel;
            if (P.out.numMismatch) P.out.numMismatch[pair] = (int16_t)nMis;
            if (P.out.nBQT) P.out.nBQT[pair] = (int16_t)nBQT;
            if (P.out.nmmBQT) P.out.nmmBQT[pair] = (int16_t)nmmBQT;
            if (P.out.nMMLeft) P.out.nMMLeft[pair] = (int16_t)nMML;
            if (P.out.nMMRight) P.out.nMMRight[pair] = (int16_t)nMMR;
            if (P.out.firstBase) P.out.firstBase[pair] = (int16_t)firstB;
            if (P.out.lastBase) P.out.lastBase[pair] = (int16_t)lastB;
        }
        STAMP(7);   // mLogBQ + coverage + scalar outputs
        // the next read reuses rdE/rdC/ms/bt of this wave: all of this pair's LDS reads precede (in program
        // order, same wave) the next pair's LDS writes, and DS ops of one wave execute in order.
    }
    }   // chunks of the window's reads (G = 1: one pass)
    }   // item loop
    STAMP_FLUSH;
}

#ifdef DD_INST_COMMON
// onHap[r] = 1 iff any haplotype of the window has !offHapHMQ for read r (DInDel.cpp:1710, 1720)
__global__ void dd_onhap_kernel(const KernelArgs P)
{
    const int r = P.read_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.read_end) return;
    // window of read r: binary search in win_read_off
    int lo = 0, hi = P.n_windows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (P.win_read_off[mid] <= r) lo = mid; else hi = mid;
    }
    const int w = lo;
    const int H = P.win_hap_off[w + 1] - P.win_hap_off[w];
    const int r0 = P.win_read_off[w];
    const int R = P.win_read_off[w + 1] - r0;
    const int64_t base = P.win_pair_off[w] + (r - r0);
    int on = 0;
    for (int h = 0; h < H; h++) {
        const int64_t p = base + (int64_t)h * R;
        const int st = P.out.status[p];
        if (st != DD_PAIR_HAPSIZE && st != DD_PAIR_UNSUPPORTED && !P.out.offHapHMQ[p]) on = 1;
    }
    P.out.onHap[r] = (uint8_t)on;
}
#endif

template <int K, int D, bool GBT, bool FOLD = false, int OCC = 0, int G = 1>
static hipError_t launch_one(const KernelArgs &A, dim3 grid, int waves, size_t lds, hipStream_t st)
{
    // The cap on dynamic LDS is a property of the function, not of a launch: it is raised ONCE per template instance (and
    // per device) to the CU's 160 KiB, so that host threads launching the same instance with different tile sizes cannot
    // lower it under each other between the set and the launch.
    static std::atomic<unsigned> raised(0u);             // bit d: done on device d
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned bit = 1u << (dev & 31);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dd_hmm_kernel<K, D, GBT, FOLD, OCC, G>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised.fetch_or(bit, std::memory_order_release);
    }
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipLaunchKernelGGL((dd_hmm_kernel<K, D, GBT, FOLD, OCC, G>), grid, dim3(waves * 64), lds, st, A);
    return hipGetLastError();
}

#ifdef DD_ONLY_K   // diagnostic builds: a single (K, D) compiles in seconds
hipError_t launch_hmm(int K, int Dt, bool gbt, int build, const KernelArgs &A, unsigned grid, int waves, size_t lds, hipStream_t st)
{
    const bool fold = (build & DD_BUILD_FOLD) != 0;
    if (K != DD_ONLY_K || Dt != DD_ONLY_D) return hipErrorInvalidValue;
    if (build & DD_BUILD_HALF)
        return gbt ? launch_one<DD_ONLY_K, DD_ONLY_D, true, false, 0, 2>(A, dim3(grid), waves, lds, st)
                   : launch_one<DD_ONLY_K, DD_ONLY_D, false, false, 0, 2>(A, dim3(grid), waves, lds, st);
    if constexpr (DD_ONLY_K <= 2) { if (fold && !gbt) return launch_one<DD_ONLY_K, DD_ONLY_D, false, true>(A, dim3(grid), waves, lds, st); }
    else if (fold) return hipErrorInvalidValue;
    return gbt ? launch_one<DD_ONLY_K, DD_ONLY_D, true>(A, dim3(grid), waves, lds, st)
               : launch_one<DD_ONLY_K, DD_ONLY_D, false>(A, dim3(grid), waves, lds, st);
}
#else
// The product library instantiates K = 1..12 positions per lane x the D builds 6 / 11 / 12 x {LDS, HBM-scratch} back-pointers:
// 72 kernels, plus the FOLD variants: K = 1 and K = 2 LDS builds at D = 6, the K = 2 scratch build at D = 6, the K = 2 LDS build at D = 11.  The file is compiled once per D build (-DDD_INST_D=6|11|12, see the Makefile) so that they build in parallel;
// the D = 6 unit also carries the dispatcher and the small kernels.
template <int D, bool GBT>
static hipError_t launch_k(int K, const KernelArgs &A, dim3 grid, int waves, size_t lds, hipStream_t st)
{
    switch (K) {
    case 1: return launch_one<1, D, GBT>(A, grid, waves, lds, st);
    case 2: return launch_one<2, D, GBT>(A, grid, waves, lds, st);
    case 3: return launch_one<3, D, GBT>(A, grid, waves, lds, st);
    case 4: return launch_one<4, D, GBT>(A, grid, waves, lds, st);
    case 5: return launch_one<5, D, GBT>(A, grid, waves, lds, st);
    case 6: return launch_one<6, D, GBT>(A, grid, waves, lds, st);
    case 7: return launch_one<7, D, GBT>(A, grid, waves, lds, st);
    case 8: return launch_one<8, D, GBT>(A, grid, waves, lds, st);
    case 9: return launch_one<9, D, GBT>(A, grid, waves, lds, st);
    case 10: return launch_one<10, D, GBT>(A, grid, waves, lds, st);
    case 11: return launch_one<11, D, GBT>(A, grid, waves, lds, st);
    case 12: return launch_one<12, D, GBT>(A, grid, waves, lds, st);
    default: return hipErrorInvalidValue;
    }
}

// half-wave builds (two pairs per wavefront): K = 1, 3, 5 positions per lane of a 32-lane half
template <int D, bool GBT>
static hipError_t launch_half(int K, const KernelArgs &A, dim3 grid, int waves, size_t lds, hipStream_t st)
{
    switch (K) {
    case 1: return launch_one<1, D, GBT, false, 0, 2>(A, grid, waves, lds, st);
    case 3: return launch_one<3, D, GBT, false, 0, 2>(A, grid, waves, lds, st);
    case 5: return launch_one<5, D, GBT, false, 0, 2>(A, grid, waves, lds, st);
    default: return hipErrorInvalidValue;
    }
}

#ifndef DD_INST_D
#define DD_INST_D 0            // 0: every D build in this unit
#endif
hipError_t launch_hmm_d6(int K, bool gbt, int build, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st);
hipError_t launch_hmm_d11(int K, bool gbt, int build, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st);
hipError_t launch_hmm_d12(int K, bool gbt, int build, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st);
hipError_t launch_hmm_d32(int K, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st);
#if DD_INST_D == 0 || DD_INST_D == 6
hipError_t launch_hmm_d6(int K, bool gbt, int build, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st)
{
    const bool fold = (build & DD_BUILD_FOLD) != 0;
    if (build & DD_BUILD_HALF) return gbt ? launch_half<6, true>(K, A, g, waves, lds, st) : launch_half<6, false>(K, A, g, waves, lds, st);
    if (gbt && K == 3 && (build & DD_BUILD_TWO_WAVES)) return launch_one<3, 6, true, false, 2>(A, g, waves, lds, st);
    if (fold && !gbt && K == 1) return launch_one<1, 6, false, true>(A, g, waves, lds, st);
    if (fold && !gbt && K == 2) return launch_one<2, 6, false, true>(A, g, waves, lds, st);
    if (fold && gbt && K == 2) return launch_one<2, 6, true, true>(A, g, waves, lds, st);
    return gbt ? launch_k<6, true>(K, A, g, waves, lds, st) : launch_k<6, false>(K, A, g, waves, lds, st);
}
#endif
#if DD_INST_D == 0 || DD_INST_D == 11
hipError_t launch_hmm_d11(int K, bool gbt, int build, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st)
{
    const bool fold = (build & DD_BUILD_FOLD) != 0;
    if (build & DD_BUILD_HALF) return gbt ? launch_half<11, true>(K, A, g, waves, lds, st) : launch_half<11, false>(K, A, g, waves, lds, st);
    if (fold && !gbt && K == 2) return launch_one<2, 11, false, true>(A, g, waves, lds, st);
    return gbt ? launch_k<11, true>(K, A, g, waves, lds, st) : launch_k<11, false>(K, A, g, waves, lds, st);
}
#endif
#if DD_INST_D == 0 || DD_INST_D == 12
hipError_t launch_hmm_d12(int K, bool gbt, int build, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st)
{
    if (build & DD_BUILD_HALF) return gbt ? launch_half<12, true>(K, A, g, waves, lds, st) : launch_half<12, false>(K, A, g, waves, lds, st);
    return gbt ? launch_k<12, true>(K, A, g, waves, lds, st) : launch_k<12, false>(K, A, g, waves, lds, st);
}
#endif

#if DD_INST_D == 0 || DD_INST_D == 32
// maxLengthDel 12..31 (the reference takes any --maxLengthIndel, DInDel.cpp:4157): one D = 32 build per K = 1..9 (7 bits of back-pointer per
// position: 63 bits), whole wavefronts, scratch back-pointers, register-lean form with the jump candidates in a run-time loop (no spilled register), 3 / 2 / 1 waves per SIMD for K <= 3 / <= 6 / <= 9.
hipError_t launch_hmm_d32(int K, const KernelArgs &A, dim3 g, int waves, size_t lds, hipStream_t st)
{
    switch (K) {
    case 1: return launch_one<1, 32, true>(A, g, waves, lds, st);
    case 2: return launch_one<2, 32, true>(A, g, waves, lds, st);
    case 3: return launch_one<3, 32, true>(A, g, waves, lds, st);
    case 4: return launch_one<4, 32, true>(A, g, waves, lds, st);
    case 5: return launch_one<5, 32, true>(A, g, waves, lds, st);
    case 6: return launch_one<6, 32, true>(A, g, waves, lds, st);
    case 7: return launch_one<7, 32, true>(A, g, waves, lds, st);
    case 8: return launch_one<8, 32, true>(A, g, waves, lds, st);
    case 9: return launch_one<9, 32, true>(A, g, waves, lds, st);
    default: return hipErrorInvalidValue;
    }
}
#endif

#ifdef DD_INST_COMMON
hipError_t launch_hmm(int K, int Dt, bool gbt, int build, const KernelArgs &A, unsigned grid, int waves, size_t lds, hipStream_t st)
{
    if (Dt == 32) return (gbt && !(build & (DD_BUILD_FOLD | DD_BUILD_HALF))) ? launch_hmm_d32(K, A, dim3(grid), waves, lds, st) : hipErrorInvalidValue;
    switch (Dt) {
    case 6: return launch_hmm_d6(K, gbt, build, A, dim3(grid), waves, lds, st);
    case 11: return launch_hmm_d11(K, gbt, build, A, dim3(grid), waves, lds, st);
    case 12: return launch_hmm_d12(K, gbt, build, A, dim3(grid), waves, lds, st);
    default: return hipErrorInvalidValue;
    }
}
#endif
#endif

#ifdef DD_INST_COMMON
hipError_t launch_onhap(const KernelArgs &A, hipStream_t st)
{
    const int n = A.read_end - A.read_begin;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(dd_onhap_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A);
    return hipGetLastError();
}
#endif

} // namespace ddk
