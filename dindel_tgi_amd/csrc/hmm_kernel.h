// hmm_kernel.h — kernel argument block, table layout and launchers shared by hmm_kernel.hip and capi.cpp.
#ifndef DD_HMM_KERNEL_H
#define DD_HMM_KERNEL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dindel_hmm.h"

#define DD_WAVES 4   /* max wavefronts per workgroup: one per SIMD of a CU, one pair (two in the half-wave builds) in flight per wave */
#define DD_HALF_CHUNK 256   /* half-wave builds (two pairs per wavefront): reads of a window ordered at a time (LDS keys, capi.cpp sizes them) */

/* layout of the host-built table block (dd_build_tables) */
enum {
    TC_LLL = 0,   /* log(1-pFirstgLO)            ObservationModelFB.cpp:1643 */
    TC_LFL = 1,   /* log(pFirstgLO)              :1644 */
    TC_II = 2,    /* logpInsgIns = -0.5          :1664 */
    TC_NI = 3,    /* log(1-exp(logpInsgIns))     :1665 */
    TC_IN = 4,    /* log(pError)                 :1666 */
    TC_NN = 5,    /* log(1-pError)               :1667 */
    TC_EDEF = 6,  /* log(1e-5)                   :1678 */
    TC_NDEF = 7,  /* log(1-1e-5)                 :1679 */
    TC_BQT = 8,   /* checkBaseQualThreshold */
    TC_HMQ = 12,  /* 4 doubles: bMid prior for mapQual = 1-1e-10 (:1093): off/noins, off/ins, on/noins, on/ins */
    TC_PINS = 21, /* insert-size prior path (mapUnmappedReads): log(1-exp(log pError)), then for mapQual = 1-1e-10: log(pOff), log(1-pOff) */
    TC_FAST = 16, /* --faster model (Faster.cpp:300-352): log(1-pError), log(pError), log(1-exp(-0.25)), log(1-1e-10), log(1e-10) */
    T_QUAL = 32,                          /* 4 per quality: eq, uq (:232-234), log10(1-q) (:1406), q */
    T_MAPQ = T_QUAL + 4 * DD_MAX_QUAL_TABLE, /* 4 per mapping quality: prior off/noins, off/ins, on/noins, on/ins (:296-303) */
    T_HP = T_MAPQ + 4 * DD_MAX_QUAL_TABLE,   /* 2 per run length: log(perr(len)), log(1-perr(len)) (ReadIndelErrorModel.hpp:36-50) */
    T_MAPQF = T_HP + 2 * DD_HP_TABLE,         /* --faster: 2 per mapping quality: log(1-pOffFirst), log(pOffFirst) with capMapQualFast (Faster.cpp:117-124) */
    T_MAPQ2 = T_MAPQF + 2 * DD_MAX_QUAL_TABLE, /* insert-size prior path: 2 per mapping quality: log(pOffFirst), log(1-pOffFirst) (:296-303) */
    T_END = T_MAPQ2 + 2 * DD_MAX_QUAL_TABLE
};

namespace ddk {

struct KernelArgs {
    /* batch (device pointers) */
    int32_t n_windows, n_haps, n_reads;
    const int32_t *win_hap_off, *win_read_off;
    const uint32_t *win_hap_start;
    const int32_t *hap_seq_off;
    const char *hap_seq;
    const int32_t *hap_var_off, *hap_var, *hap_var_flank;
    const int32_t *read_seq_off;
    const char *read_seq;
    const uint8_t *read_qidx, *read_mqidx;
    const uint32_t *read_start;
    const uint8_t *read_flags;
    const int32_t *hap_window;
    const int64_t *win_pair_off, *win_hpos_off, *win_varcov_off;
    const double *tables;
    const uint8_t *sym_lut;              /* byte -> symbol id (dd_build_symbol_lut); NULL = A,C,G,T,N only */
    const uint8_t *win_skip;             /* per window: 1 = shape outside the kernel limits, pairs only get DD_PAIR_UNSUPPORTED; NULL = none */
    /* mapUnmappedReads: mate arrays + library log tables; read_mate_pos == NULL switches the insert-size prior off */
    const int32_t *read_mate_pos, *read_mate_len; const uint8_t *read_lib;
    const int32_t *lib_off; const double *lib_logprob, *lib_log95;
    dd_result out;
    /* params */
    int32_t D, maxLengthDel, padCover, bMid, maxMismatch;
    int32_t always_ro;                   /* 1: never skip the RO sink chain speculatively (diagnostics / A-B) */
    /* launch geometry */
    int32_t n_split, n_items;            /* items = haplotypes x read slices; workgroups stride over them */
    int32_t reads_per_wave;              /* main kernel, ragged batches: > 0 = a haplotype uses only ceil(its reads / (waves x this)) of its n_split slices */
    int32_t item_begin;                  /* this launch covers items [item_begin, n_items) (chunked host path) */
    int32_t *work_counter;               /* main kernel: NULL = items by fixed stride; else a zeroed counter in device memory the workgroups draw items from */
    int32_t read_begin, read_end;        /* reads covered by this launch (onHap kernel) */
    const int32_t *hap_list;             /* length-class launches: haplotype of item i is hap_list[i / n_split]; NULL = identity */
    int32_t len_min, len_max;            /* length-class launches: only reads with len_min <= L <= len_max */
    int32_t run_onhap;
    int32_t fast_groups;                 /* --faster kernel: pairs a wavefront works on at a time (4, 2 or 1; LDS-limited) */
    void *bt_scratch; int32_t bt_rows;   /* GBT builds: per-wave back-pointer tiles in HBM, rows = max read length */
    unsigned long long *dbg;   /* diagnostic builds only (DD_STAMPS); NULL otherwise */
    /* LDS layout (bytes) */
    uint32_t lds_off_L, lds_off_E, lds_off_N, lds_off_Q, lds_off_C, lds_off_Y, lds_shared_bytes, lds_wave_bytes;
    int32_t n_qual;
    uint32_t lds_off_A, lds_off_I, lds_off_rdE, lds_off_rdC, lds_off_rdQ, lds_off_ms, lds_off_bt;
    /* half-wave builds: bytes of one pair's rows inside a wavefront's region (rows of pair q at q * lds_group_bytes; the back-pointer tile
     * is the wavefront's), and the block-shared sort keys */
    uint32_t lds_group_bytes, lds_off_S;
    uint32_t lds_off_W;                  /* the workgroup's work counter (one int) */
    uint32_t bt_wave_bytes;              /* GBT builds: bytes of a wavefront's scratch region = back-pointer tile (bt_rows x 64 words) + beta[bMid] stash */
};

/* Workgroups are dealt round-robin to the 8 XCDs (workgroup b -> XCD b % 8), each with its own L2.  The haplotypes of a
 * window re-read the same reads, so consecutive items should land on ONE XCD: XCD x takes the x-th contiguous eighth of
 * the grid's index range.  Bijection on [0, gridDim.x).  Only for one-shot grids (one item per workgroup): a persistent grid
 * whose last round is partial would leave that round's items on the first XCDs alone. */
__device__ __forceinline__ int xcd_contiguous_block_id(int n_items_in_launch)
{
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    if (G < n_items_in_launch) return b;
    const int x = b & 7, q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

/* build: which variant of the (K, D, back-pointer) build — DD_BUILD_FOLD: the right->middle pass carries the LO / RO end states in the
 * generic candidate code (K <= 2; every haplotype of the launch must leave position 64 K - 1 idle: 64 K >= Hs + 3);
 * DD_BUILD_TWO_WAVES: the K = 3 / D = 6 scratch build compiled for 2 waves per SIMD instead of 3 */
#define DD_BUILD_FOLD 1
#define DD_BUILD_TWO_WAVES 2
#define DD_BUILD_HALF 4       /* two pairs per wavefront on 32-lane halves (K positions per lane of a half): K = 1, 3, 5 */
hipError_t launch_hmm(int K, int Dt, bool gbt, int build, const KernelArgs &A, unsigned grid, int waves, size_t lds_bytes, hipStream_t st);
hipError_t launch_onhap(const KernelArgs &A, hipStream_t st);
hipError_t launch_faster(const KernelArgs &A, unsigned grid, int waves, size_t lds_bytes, hipStream_t st);   /* faster_kernel.hip */

/* N1: per-window haplotype-pair read sums (genotype_kernel.hip) */
struct PairSumArgs {
    int32_t n_windows;
    int64_t n_slots;                 /* sum over windows of H_w^2 */
    const int32_t *win_hap_off, *win_read_off;
    const int64_t *win_pair_off, *win_hh_off;
    const double *ll;                /* per pair, liks[h][r] order */
    double *out;                     /* [n_slots]: window w, slot h1*H_w+h2 (h1<=h2) */
};
hipError_t launch_pair_sums(const PairSumArgs &A, hipStream_t st);

/* N1: MAP haplotype pairs + qual per window (genotype_kernel.hip) */
struct MapPairArgs {
    int32_t n_windows;
    const int32_t *win_hap_off;
    const int64_t *win_hh_off;
    const double *pair_sum, *prior;  /* [n_slots] */
    const uint8_t *filtered;         /* [n_haps] */
    const int32_t *ncand;            /* [n_haps] hap_num_candidate_indels */
    double *posterior;               /* [n_slots] or NULL */
    int32_t *pairs;                  /* [4*n_windows] */
    double *vals;                    /* [3*n_windows] */
};
hipError_t launch_map_pairs(const MapPairArgs &A, hipStream_t st);

} // namespace ddk
#endif
