/*
 * dindel_hmm.h — C ABI of the MI355X-native read x candidate-haplotype HMM likelihood path.
 *
 * This is the drop-in boundary for ONE hot path of genome/dindel-tgi:
 *
 *     void DetInDel::computeLikelihoods(const vector<Haplotype>& haps, const vector<Read>& reads,
 *                                       vector<vector<MLAlignment> >& liks,
 *                                       uint32_t leftPos, uint32_t rightPos, vector<int>& onHap);
 *                                                  (reference: DInDel.hpp:136, DInDel.cpp:1707-1739)
 *
 * which, per (haplotype, read) pair, constructs an ObservationModelFBMaxErr and calls
 * calcLikelihood() (reference: ObservationModelFB.cpp:1631-1829, 1057-1165, 1351-1475).
 * Everything is plain pointers and sizes; no STL, no torch types.  One call processes a BATCH of
 * windows (the reference processes one window per call; a GPU needs many to fill 256 CUs).
 *
 * Two ways in:
 *   (1) dd_compute_likelihoods()  — host pointers in, host pointers out.  The library owns device
 *       memory, H2D/D2H and the launch.  This is what the reference's C++ host would bind.
 *   (2) dd_workspace_bytes() / dd_build_tables() / dd_launch_device() — every buffer is a DEVICE
 *       pointer owned by the caller (e.g. torch tensors), launch on a caller-supplied hipStream_t.
 *       bench.py and the multi-GPU driver use this so inputs are resident in HBM when timing starts.
 *
 * There is NO CPU fallback behind this ABI: without a HIP device every compute entry point returns
 * DD_ERR_NO_DEVICE.  The CPU restatement lives in oracle/ and is test infrastructure only.
 */
#ifndef DINDEL_HMM_H
#define DINDEL_HMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_ABI_VERSION 12

/* hpos[] sentinel values — reference: MLAlignment.hpp:31-34 */
#define DD_HPOS_INS (-1)
#define DD_HPOS_DEL (-2)
#define DD_HPOS_LO  (-3)
#define DD_HPOS_RO  (-4)
/* hpos[] AS WRITTEN BY THIS LIBRARY carries, for an inserted read base, the haplotype position the reference keys the
 * insertion at (`pos`, ObservationModelFB.cpp:1380 / Faster.cpp:608 — the x of the inserted state numS+x) instead of the
 * bare MLAlignment::INS: the value is DD_HPOS_INS_KEY0 - pos (pos >= 1, so <= -17).  MLAlignment::hpos only says "-1", and
 * the key cannot always be recovered from the neighbouring entries (a read that is inserted as a whole; in the --faster model
 * an insertion after bases parked on the last haplotype base), yet ml.indels is keyed by it.  A consumer that wants the
 * reference's array maps DD_HPOS_IS_INS(v) to DD_HPOS_INS (the C++ adapter does). */
#define DD_HPOS_INS_KEY0    (-16)
#define DD_HPOS_IS_INS(v)   ((v) == DD_HPOS_INS || (v) < DD_HPOS_INS_KEY0)
#define DD_HPOS_INS_POS(v)  (DD_HPOS_INS_KEY0 - (v))

/* return codes of the entry points */
#define DD_SUCCESS            0
#define DD_ERR_NO_DEVICE     -1   /* no HIP device / HIP runtime error; message via dd_last_error() */
#define DD_ERR_INVALID       -2   /* malformed batch (offsets not monotone, null pointer, ...)       */
#define DD_ERR_UNSUPPORTED   -3   /* shape or option outside what the kernels cover (see limits)     */
#define DD_ERR_HIP           -4

/* per-pair status, mapped by the host adapter to the reference's throw strings / exit:
 *   1 -> throw string("hapSize error.")   ObservationModelFB.cpp:47
 *   2 -> throw string("Nan detected")     DInDel.cpp:1732-1735
 *   3 -> "Likelihood>0", exit(1)          DInDel.cpp:1722-1731                                   */
#define DD_PAIR_OK            0
#define DD_PAIR_HAPSIZE       1
#define DD_PAIR_NAN           2
#define DD_PAIR_LLPOS         3
#define DD_PAIR_UNSUPPORTED   4   /* the pair's WINDOW has a shape outside the kernel limits below (or an empty read / haplotype): none of
                                     its pairs is computed (ll = 0, offHap = offHapHMQ = 1, onHap = 0, the other outputs unwritten); every
                                     other window of the batch is processed normally.  The reference has no such limit: the C++ adapter
                                     reports the window like one whose likelihood step threw (skipped row, DInDel.cpp:1369-1374). */

/* kernel limits (screened per window on the host before launch: dd_screen_windows) */
#define DD_MAX_HAP_LEN      766   /* numS = Hs+2 <= 64 lanes x 12 positions                         */
#define DD_MAX_READ_LEN    1024
#define DD_MAX_LENGTH_DEL    31   /* D = maxLengthDel + 1 <= 32.  0..11 run the specialised D = 6 / 11 / 12 builds; 12..31 one D = 32 build (correct, not tuned;
                                     haplotypes up to 574 bp there) */
#define DD_MAX_QUAL_TABLE   256
#define DD_HP_TABLE          64   /* homopolymer run lengths >= 52 are all capped at 0.99            */

/* Mirrors the fields of ObservationModelParameters the path reads — ObservationModel.hpp:28-99. */
typedef struct dd_params {
    double  pError;                  /* probability of a read indel                                  */
    double  pMut;                    /* probability of a mutation in the read                        */
    double  pFirstgLO;               /* P(first on-haplotype base | left of haplotype) = 0.01        */
    double  mapQualThreshold;        /* phred cap on the read mapping quality ("capMapQualThreshold") */
    double  checkBaseQualThreshold;  /* 0.95: base-quality threshold for nBQT / nmmBQT / mLogBQ      */
    int32_t maxLengthDel;            /* = maxLengthIndel; numT = maxLengthDel+2                      */
    int32_t padCover;                /* "flankRefSeq"                                                */
    int32_t bMid;                    /* -1 = compute from overlap (ObservationModelFB.cpp:96)        */
    int32_t forceReadOnHaplotype;    /* ObservationModelFB.cpp:307-316                               */
    int32_t mapUnmappedReads;        /* 1: add the library insert-size prior of paired reads at the join (ObservationModelFB.cpp:279-292;
                                        the CLI sets it with --libFile, DInDel.cpp:4268-4272); needs the mate / library arrays of dd_batch */
    int32_t maxMismatch;             /* "flankMaxMismatch": used only by the filterHaplotypes coverage flags */
    double  capMapQualFast;          /* phred cap on the mapping quality in the --faster model (Faster.cpp:117); 40 / CLI 45 */
} dd_params;

/* ObservationModelParameters::setDefaultValues() — ObservationModel.hpp:39-64 */
void dd_params_struct_defaults(dd_params *p);
/* the set main() installs from the CLI defaults — DInDel.cpp:3937-3949, 4122-4157 */
void dd_params_cli_defaults(dd_params *p);

/*
 * One batch = n_windows windows, CSR-packed.  Window w owns haplotypes [win_hap_off[w], win_hap_off[w+1])
 * and reads [win_read_off[w], win_read_off[w+1]).  Pairs are produced hap-major then read, like
 * liks[hidx][r]:   pair(w,h,r) = win_pair_off[w] + h * R_w + r .
 *
 * Base qualities and mapping qualities are passed as one-byte indices into small tables of the
 * probabilities the reference stores in Read::qual / Read::mapQual (Read.hpp:127-148): BAM gives
 * integer phred, so at most 94 / 256 distinct values occur; the host adapter dedups arbitrary
 * doubles.  The logs of those probabilities are taken ON THE HOST with libm (dd_build_tables) so
 * that the device does only add / compare / select and cannot differ from glibc's log().
 */
typedef struct dd_batch {
    int32_t        n_windows;
    const int32_t *win_hap_off;    /* [n_windows+1]                                                 */
    const int32_t *win_read_off;   /* [n_windows+1]                                                 */
    const uint32_t*win_hap_start;  /* [n_windows]   leftPos = hapStart of every model in the window  */

    const int32_t *hap_seq_off;    /* [n_haps+1]    into hap_seq                                    */
    const char    *hap_seq;        /* haplotype bases, any byte; 'N' matches everything              */
    const int32_t *hap_var_off;    /* [n_haps+1]    into hap_var (units: variants); may be NULL      */
    const int32_t *hap_var;        /* per variant of the HAPLOTYPE: {startRead, endRead}
                                      (AlignedVariant, Variant.hpp:125-128): hap.indels in map order
                                      then hap.snps in map order — ObservationModelFB.cpp:1465-1472  */

    const int32_t *read_seq_off;   /* [n_reads+1]   into read_seq / read_qidx                       */
    const char    *read_seq;       /* read bases                                                     */
    const uint8_t *read_qidx;      /* per base: index into qual_table                                */
    const uint8_t *read_mqidx;     /* [n_reads]     index into mapq_table                            */
    const uint32_t*read_start;     /* [n_reads]     uint32_t(read.posStat.first)                     */
    const uint8_t *read_flags;     /* [n_reads]     DD_READ_* bits; bit0 = read.isUnmapped() (BAM flag 0x4) */

    int32_t        n_qual;  const double *qual_table;  /* P(base correct), Read.hpp:143-148          */
    int32_t        n_mapq;  const double *mapq_table;  /* read.mapQual,    Read.hpp:127-131          */

    const int32_t *hap_var_flank;  /* optional (NULL = none), per variant of hap_var, 3 ints:
                                      {getLeftFlankRead(), getRightFlankRead(), kind}, kind 1 = DEL, 2 = INS,
                                      0 = neither (SNP entries): inputs of filterHaplotypes, DInDel.cpp:1973-1976 */

    /* Only read when dd_params.mapUnmappedReads != 0 (NULL otherwise): what computeBMidPrior takes from the mate and the
     * read's library (ObservationModelFB.cpp:279-292, Read.hpp:162-204,426, Library.hpp:60-66). */
    const int32_t *read_mate_pos;  /* [n_reads]     read.matePos                                     */
    const int32_t *read_mate_len;  /* [n_reads]     read.mateLen, -1 = unknown (no prior for that read) */
    const uint8_t *read_lib;       /* [n_reads]     index of read.getLibrary() in the tables below   */
    int32_t        n_libs;
    const int32_t *lib_off;        /* [n_libs+1]    into lib_prob; a library has maxins = lib_off[i+1]-lib_off[i] >= 1 entries */
    const double  *lib_prob;       /* Library::getProb(x) for x = 0..maxins-1 (normalised, floored at 1e-10) */
    const double  *lib_p95;        /* [n_libs]      Library::getNinetyFifthPctProb()                 */
} dd_batch;

/* read_flags bits (BAM flag bits the path looks at: Read.hpp:201-204, ObservationModelFB.cpp:56,279-281) */
#define DD_READ_UNMAPPED       1   /* read.isUnmapped()                                    */
#define DD_READ_PAIRED         2   /* read.isPaired()                                      */
#define DD_READ_MATE_UNMAPPED  4   /* read.mateIsUnmapped()                                */
#define DD_READ_MATE_REVERSE   8   /* read.mateIsReverse()                                 */
#define DD_READ_MATE_SAME_TID 16   /* bam->core.tid == bam->core.mtid                      */

/* Per (window,hap,read) pair, in pair order.  Any output pointer may be NULL (then not written),
 * except ll and status. Mirrors MLAlignment (MLAlignment.hpp:28-76). */
typedef struct dd_result {
    double  *ll, *llOn, *llOff, *mLogBQ;
    uint8_t *offHap, *offHapHMQ;
    int16_t *numIndels, *numMismatch, *nBQT, *nmmBQT, *nMMLeft, *nMMRight, *firstBase, *lastBase;
    int16_t *hpos;        /* L entries per pair, at  hpos_off(w) + h*SL_w + (read_seq_off[r]-read_seq_off[r0(w)]),
                             SL_w = total read bases of window w;  >=0 hap index, -3 LO, -4 RO,
                             inserted base: DD_HPOS_INS_KEY0 - pos (see DD_HPOS_INS_POS) */
    uint8_t *var_covered; /* per (pair, variant of that hap): hapIndelCovered / hapSNPCovered        */
    int32_t *status;      /* DD_PAIR_*                                                               */
    uint8_t *onHap;       /* [n_reads] 1 iff any hap of the window has !offHapHMQ (DInDel.cpp:1720)  */
    uint8_t *var_fcov;    /* next row N1, same indexing as var_covered: 1 iff the read counts as covering the
                             haplotype's indel in DetInDel::filterHaplotypes (DInDel.cpp:1951-2062): read selected
                             (!offHapHMQ && numIndels==0), every haplotype base of [leftFlank-padCover,
                             rightFlank+padCover] aligned to, at most maxMismatch mismatches there.  Needs
                             dd_batch.hap_var_flank.  Two undefined behaviours of the reference loop are not reproduced:
                             it reads hpos[L], one past the end; and when leftFlank-padCover < 0 it would count the
                             negative hpos sentinels as covered positions and index the haplotype sequence with them —
                             here such an interval is simply never covered.  All pairs of a DD_PAIR_HAPSIZE haplotype get 0. */
} dd_result;

/* sizes derived from a batch (host-side, O(n_windows + n_haps)) */
typedef struct dd_sizes {
    int64_t n_haps, n_reads, n_pairs;
    int64_t hap_bases, read_bases;
    int64_t hpos_len;        /* total int16 entries of dd_result.hpos                                */
    int64_t var_cov_len;     /* total bytes of dd_result.var_covered                                 */
    int64_t cells;           /* sum over pairs of L * Hs  (the metric's "cell")                      */
    int32_t max_hap_len, max_read_len;
} dd_sizes;

int dd_batch_sizes(const dd_batch *b, dd_sizes *out);

/* Per-window shape screen: win_skip[w] = 1 iff window w holds a haplotype longer than DD_MAX_HAP_LEN, a read longer than
 * DD_MAX_READ_LEN, an empty haplotype / read, or a haplotype byte that dd_build_symbol_lut could not number (more than 26
 * distinct non-ACGTN values in the batch); its pairs get DD_PAIR_UNSUPPORTED instead of failing the whole batch.
 * max_len_out[2] (may be NULL) = longest haplotype / read among the windows that pass.  Returns the number of skipped
 * windows (>= 0) or a DD_ERR_* code.  dd_compute_likelihoods does this by itself; callers of dd_launch_device upload
 * win_skip (dd_device_batch.win_skip) and plan with the returned maxima. */
int dd_screen_windows(const dd_batch *b, uint8_t *win_skip, int32_t max_len_out[2]);

/* Offsets a consumer needs to index the ragged outputs; arrays sized [n_windows+1]. */
int dd_batch_offsets(const dd_batch *b, int64_t *win_pair_off, int64_t *win_hpos_off, int64_t *win_varcov_off);

/* ---- (1) host-pointer entry point --------------------------------------------------------- */
/* device >= 0: HIP device ordinal.  Synchronous.  Replaces the body of computeLikelihoods for a
 * batch of windows. */
int dd_compute_likelihoods(const dd_params *p, const dd_batch *b, dd_result *r, int device);
/* Several devices of ONE process (the reference is one process, DInDel.cpp:4074): the batch is cut into n_devices contiguous
 * window blocks balanced by cells (sum over a window's pairs of L * Hs) — dd_partition_windows — and block i
 * runs on devices[i] under its own host thread, device arena and streams; every block writes straight into the caller's
 * arrays at its windows' offsets, so the result is the single-device result bit for bit and no collective is needed.
 * A device may be named more than once (its blocks then run concurrently on it).  Returns the first non-zero code of a block. */
int dd_compute_likelihoods_multi(const dd_params *p, const dd_batch *b, dd_result *r, const int *devices, int n_devices);
/* bounds[n_parts+1]: block i = windows [bounds[i], bounds[i+1]) */
int dd_partition_windows(const dd_batch *b, int n_parts, int32_t *bounds);

/* The host-pointer entry points keep a device arena, a pinned staging mirror and two streams per host thread between
 * calls (one window per call would otherwise be all allocation overhead); this frees them. */
void dd_release_cache(void);
/* Makes the calling thread's cache on `device` at least this big now (device arena, pinned staging mirror) and creates its streams:
 * a caller that knows work is coming can pay the allocations while its first batch is still being prepared.  Optional. */
int dd_reserve_cache(int device, size_t device_bytes, size_t pinned_bytes);
/* Page-locked host memory for the arrays of dd_batch / dd_result: with it the H2D / D2H copies of dd_compute_likelihoods run
 * at full link speed and truly overlap the kernels (pageable memory is staged by the runtime).  Optional: any host pointer
 * works.  dd_host_alloc returns NULL when there is no device or the allocation fails.  Output arrays of dd_compute_likelihoods that
 * lie in such memory are written by the kernels directly (see dd_last_direct_outputs). */
void *dd_host_alloc(size_t bytes);
void dd_host_free(void *p);

/* ---- (2) device-pointer entry points ------------------------------------------------------ */
/* Table block built on the host with libm (emission logs per quality, bMid priors per mapping
 * quality, homopolymer indel-error logs, transition constants).  Returns number of doubles written
 * (<= DD_TABLE_DOUBLES); `out` is host memory that the caller copies to the device. */
#define DD_TABLE_DOUBLES (32 + 4*DD_MAX_QUAL_TABLE + 4*DD_MAX_QUAL_TABLE + 2*DD_HP_TABLE + 2*DD_MAX_QUAL_TABLE + 2*DD_MAX_QUAL_TABLE + 64)
int dd_build_tables(const dd_params *p, const double *qual_table, int n_qual,
                    const double *mapq_table, int n_mapq, double *out);

/* Bases are compared as characters, as in the reference (hap[y]==nuc || hap[y]=='N', ObservationModelFB.cpp:246): any
 * byte may occur in reads and haplotypes (IUPAC codes, soft-masked lower case).  This builds the byte -> symbol-id table
 * the main kernel uses (A,C,G,T -> 0..3, N -> 4, other bytes present in a haplotype of the batch -> 5..30, the rest -> 31);
 * out[256].  With more than 26 distinct non-ACGTN haplotype bytes in one batch the values beyond the 26th (in byte order) get no id:
 * dd_screen_windows flags the windows that hold one (DD_PAIR_UNSUPPORTED for those windows, not an error of the call). */
int dd_build_symbol_lut(const dd_batch *b, uint8_t *out);

/* log(Library::getProb(x)) for every entry of dd_batch.lib_prob (logprob_out[lib_off[n_libs]]) and
 * log(getNinetyFifthPctProb()) per library (log95_out[n_libs]), taken with libm on the host. */
int dd_build_library_tables(const dd_batch *b, double *logprob_out, double *log95_out);

/* Host-side derived index arrays the kernels need, sized by the caller:
 * hap_window[n_haps], win_pair_off[n_windows+1], win_hpos_off[n_windows+1], win_varcov_off[n_windows+1] */
int dd_build_index(const dd_batch *b, int32_t *hap_window, int64_t *win_pair_off,
                   int64_t *win_hpos_off, int64_t *win_varcov_off);

/* Ragged batches: every launch covers one (lane tiling of the haplotypes, read class), so one long haplotype or read does not put the
 * whole batch on the slower build.  The haplotype classes are the lane tilings (<= 30, 62, 94, 126, 158, 190, 222, 254, 318, ... 766 bp).
 * Read classes of a tiling: 0 = the reads of windows whose longest read (up to 160 bp) is <= T, 1 = the reads up to 160 bp of the other
 * windows, 2 = reads longer than 160 bp; T = the longest read whose back-pointer tile still fits LDS at full occupancy for that tiling and
 * these params (no class 0 / 1 distinction when p is NULL).  A haplotype is listed in a launch only if its window has reads for it.
 * dd_compute_likelihoods does this by itself; for dd_launch_device the caller builds the summary once on the host
 * (dd_build_length_classes), uploads hap_class_list and hands both over in dd_device_batch. */
#define DD_N_HAP_CLASSES 16
#define DD_N_READ_CLASSES 3
typedef struct dd_launch_class {
    int32_t list_off, list_len;         /* hap_class_list[list_off .. list_off + list_len): the launch's haplotypes, ascending            */
    int32_t hap_class;                  /* lane tiling (index of the haplotype-length class)                                              */
    int32_t max_hap_len;                /* longest haplotype of the list                                                                  */
    int32_t min_read_len, max_read_len; /* shortest admissible read / longest read present                                                */
    int32_t max_window_reads, avg_window_reads;   /* reads of the interval per window of the list: most, mean                             */
    int32_t avg_read_len;               /* mean length of those reads                                                                      */
} dd_launch_class;
typedef struct dd_length_classes {
    int32_t n_launches;                 /* non-empty (tiling, interval) combinations                                                      */
    int32_t list_len;                   /* entries of hap_class_list in use (<= n_haps * DD_N_READ_CLASSES)                               */
    dd_launch_class launch[DD_N_HAP_CLASSES * DD_N_READ_CLASSES];
} dd_length_classes;
/* hap_class_list[n_haps * DD_N_READ_CLASSES] (host memory; copy list_len entries to the device).
 * win_skip (may be NULL = none): dd_screen_windows' flags; haplotypes of skipped windows ride in the first launch without counting
 * towards its maxima (the kernel only marks their pairs), and their reads do not count. */
int dd_build_length_classes(const dd_batch *b, const uint8_t *win_skip, const dd_params *p, int32_t *hap_class_list, dd_length_classes *out);

typedef struct dd_device_batch {   /* all DEVICE pointers; same meaning as dd_batch */
    int32_t n_windows, n_haps, n_reads;
    int32_t max_hap_len, max_read_len;
    int32_t max_window_reads;       /* most reads a window has (0 = not known: every haplotype's reads are then dealt to the same number of workgroups) */
    const int32_t *win_hap_off, *win_read_off; const uint32_t *win_hap_start;
    const int32_t *hap_seq_off; const char *hap_seq; const int32_t *hap_var_off, *hap_var;
    const int32_t *read_seq_off; const char *read_seq; const uint8_t *read_qidx, *read_mqidx;
    const uint32_t *read_start; const uint8_t *read_flags;
    const int32_t *hap_window; const int64_t *win_pair_off, *win_hpos_off, *win_varcov_off;
    const double  *tables;          /* DD_TABLE_DOUBLES doubles from dd_build_tables               */
    int32_t n_qual, n_mapq;
    const int32_t *hap_var_flank;   /* optional, see dd_batch */
    const uint8_t *sym_lut;         /* optional: 256 bytes from dd_build_symbol_lut; NULL = haplotypes hold A,C,G,T,N only */
    /* mapUnmappedReads only (NULL otherwise): device copies of dd_batch's mate arrays and the log tables of
     * dd_build_library_tables */
    const int32_t *read_mate_pos, *read_mate_len; const uint8_t *read_lib;
    const int32_t *lib_off; const double *lib_logprob, *lib_log95;
    /* optional, both or neither: per-class launches for ragged batches (main model only) */
    const int32_t *hap_class_list;          /* DEVICE copy of dd_build_length_classes' list */
    const dd_length_classes *classes;       /* HOST pointer */
    const uint8_t *win_skip;                /* optional: DEVICE copy of dd_screen_windows' flags [n_windows]; NULL = every window is
                                               within the limits (max_hap_len / max_read_len then cover the whole batch) */
} dd_device_batch;

/* bytes of device scratch dd_launch_device needs for this shape: the back-pointer tiles of the HBM-scratch builds behind a 256-byte header whose
 * first word is the item counter a launch zeroes (hipMemsetAsync on `stream`) and its workgroups draw from — never 0, and never shared by two
 * launches that may run at the same time (one workspace per stream) */
size_t dd_workspace_bytes(const dd_params *p, const dd_device_batch *b);

/* Enqueue the whole path for the batch on `stream` (a hipStream_t, may be NULL = default stream).
 * Result pointers are DEVICE pointers.  Asynchronous; no allocation, no synchronisation. */
int dd_launch_device(const dd_params *p, const dd_device_batch *b, const dd_result *r,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---- row A13: the --faster model ---------------------------------------------------------- */
/* Same batch in, same result layout out, but each pair is scored by ObservationModelS(hap, read, hapStart,
 * params).align(HapHash(4, hap)) — DetInDel::computeLikelihoodsFaster, reference DInDel.cpp:1790-1833
 * (Faster.cpp:42-681, Haplotype.hpp:315-384).  That model fills ll, hpos, firstBase, lastBase and the coverage
 * flags; offHap / offHapHMQ are always 0 (Faster.cpp:491,:529); llOn, llOff, numIndels, numMismatch stay 0 as MLAlignment's
 * constructor sets them (MLAlignment.hpp:35-46), and mLogBQ, nBQT, nmmBQT, nMMLeft, nMMRight — which that constructor leaves
 * uninitialised and this model never assigns — are reported as 0.  Params used: pError, pMut, maxLengthDel (size check only), padCover,
 * capMapQualFast, maxMismatch.  status: DD_PAIR_HAPSIZE (`throw string("hapSize error.")`, Faster.cpp:47) or
 * DD_PAIR_NAN for a read shorter than the 4-mer (`throw string("HapHash string too short")`, Haplotype.hpp:341). */
int dd_compute_likelihoods_faster(const dd_params *p, const dd_batch *b, dd_result *r, int device);
int dd_compute_likelihoods_faster_multi(const dd_params *p, const dd_batch *b, dd_result *r, const int *devices, int n_devices);
int dd_launch_device_faster(const dd_params *p, const dd_device_batch *b, const dd_result *r, void *stream);

/* ---- N1 (next row): read sums of the diploid genotype reduction --------------------------- */
/* S[w][h1*H_w+h2] (h1<=h2) = sum over the window's reads, in order, of log(0.5)+addLogs(ll[h1][r], ll[h2][r])
 * — the inner loop of DetInDel::diploidGLF, reference DInDel.cpp:3085-3091 (and :3372-3374); addLogs is
 * reference Utils.hpp:29-38.  `ll` is the per-pair array dd_launch_device / dd_compute_likelihoods produced.
 * win_hh_off[n_windows+1] = prefix sums of H_w^2 (dd_pair_sum_offsets).  exp/log run on the device:
 * agreement with glibc is ~1e-15 relative, not bit-for-bit. */
int dd_pair_sum_offsets(const dd_batch *b, int64_t *win_hh_off);
int dd_pair_sums_device(const dd_device_batch *b, const int64_t *win_hh_off_dev, int64_t n_slots,
                        const double *ll_dev, double *out_dev, void *stream);
int dd_pair_sums(const dd_batch *b, const double *ll_host, double *out_host, int device);

/* MAP haplotype pairs of the same function (reference DInDel.cpp:3073-3118), per window, on the device:
 *   posterior[h1*H+h2] = S[h1,h2] + prior[h1,h2] for h1<=h2 with filtered[h1]==filtered[h2]==0 (0 elsewhere)     (:3091)
 *   pairs[4w+0..1] = first strict maximum among pairs with ncand[h1]>0 || ncand[h2]>0  (max_indel_pair, -1 if none) (:3103)
 *   pairs[4w+2..3] = the same among pairs with no candidate indel (max_noindel_pair)                                (:3108)
 *   vals[3w+0..2]  = max_ll_indel, max_ll_noindel (= ll_ref), qual = -10 (ll_ref - addLogs(max_ll_indel, ll_ref)) / ln 10 (:3116-3118)
 * prior = DetInDel::getHaplotypePrior per pair (:3064-3068, host data), filtered[n_haps] from filterHaplotypes (:2941),
 * ncand[n_haps] = hap_num_candidate_indels (:2984-2993).  The caller raises the reference's
 * `throw string("Could not find indel allele")` (:3121) when pairs[4w] == -1.
 * dd_map_pairs: host pointers; runs the read sums (from `ll`) and this step in one go; pair_sum_out / posterior_out may be NULL. */
int dd_map_pairs_device(const dd_device_batch *b, const int64_t *win_hh_off_dev, const double *pair_sum_dev,
                        const double *prior_dev, const uint8_t *filtered_dev, const int32_t *ncand_dev,
                        double *posterior_dev, int32_t *pairs_dev, double *vals_dev, void *stream);
int dd_map_pairs(const dd_batch *b, const double *ll_host, const double *prior_host, const uint8_t *filtered_host,
                 const int32_t *ncand_host, double *pair_sum_out, double *posterior_out, int32_t *pairs_out,
                 double *vals_out, int device);

/* Launch plan the library would use for the main kernel on a batch with these maxima (no device needed):
 * out = {K positions per lane, D build (6 / 11 / 12), 1 if back-pointers go to HBM scratch else 0, waves per workgroup,
 *        read split of a haplotype, LDS bytes per workgroup, scratch bytes (low 31 bits, in KiB), waves per CU the plan expects,
 *        pairs per wavefront (1, or 2: the builds that run two reads side by side on 32-lane halves), 0}.
 * avg_reads = reads per window, n_haps = haplotypes the launch covers. */
int dd_plan_info(const dd_params *p, int max_hap_len, int max_read_len, int n_qual, int avg_reads, int n_haps, int32_t out[10]);

/* name of the dominant kernel as rocprofv3 reports it, and launch geometry of the last launch */
const char *dd_kernel_name(void);
/* geometry of the last dd_launch_device on this host thread's library instance:
 * {K positions/lane, D build, waves/workgroup, LDS bytes/workgroup, grid, read split, LDS bytes/wave, shared LDS bytes} */
void dd_last_launch(int32_t out[8]);
/* Every main-model kernel launch of the last dd_launch_device / dd_compute_likelihoods call on this host thread (a ragged batch is one
 * launch per non-empty (haplotype class, read class)): up to max_records records of DD_LAUNCH_LOG_FIELDS int32 each are written to out,
 * the number of launches is returned.  Record: {K positions per lane, pairs per wavefront (1, or 2 on 32-lane halves), D build,
 * 1 = back-pointers in HBM scratch, 1 = FOLD build, waves per workgroup, LDS bytes per workgroup, grid, read split, haplotypes covered,
 * longest haplotype of the class, shortest admissible read, longest read of the class, waves per CU the plan expects, occupancy variant,
 * duration in microseconds (-1 unless the process runs with DD_LAUNCH_TIMING=1: diagnostics, the read-out synchronises),
 * 1 = persistent grid drawing its items from a counter (ragged launches), reads per wavefront a haplotype's workgroups are sized for (0 = the
 * launch's read split for every haplotype)}. */
#define DD_LAUNCH_LOG_FIELDS 18
int dd_launch_log(int32_t *out, int max_records);
/* How many output arrays of the last dd_compute_likelihoods / _faster call on this host thread were written by the kernels directly
 * into the caller's memory: arrays that lie in page-locked, device-addressable host memory (dd_host_alloc, hipHostMalloc) are
 * not staged in HBM and not copied afterwards (status and offHapHMQ excepted).  0 for pageable memory. */
int dd_last_direct_outputs(void);
const char *dd_last_error(void);
int dd_abi_version(void);
int dd_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* DINDEL_HMM_H */
