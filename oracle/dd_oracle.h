/* dd_oracle.h — interface of the CPU restatement (TEST INFRASTRUCTURE ONLY; see dd_oracle.c header). */
#ifndef DD_ORACLE_H
#define DD_ORACLE_H
#include "../include/dindel_hmm.h"
#ifdef __cplusplus
extern "C" {
#endif

#define DDO_MAX_VAR 1056   /* > DD_MAX_READ_LEN: a read cannot show more variants than it has bases */

typedef struct ddo_out {
    double ll, llOn, llOff, mLogBQ;
    int32_t offHap, offHapHMQ, numIndels, numMismatch, nBQT, nmmBQT, nMMLeft, nMMRight, firstBase, lastBase;
    int32_t bMid, status;
    /* variants the read shows relative to the haplotype, in read order (reportVariants :1375-1463) */
    int32_t n_indel, indel_pos[DDO_MAX_VAR], indel_len[DDO_MAX_VAR] /* >0 ins, <0 del */, indel_rpos[DDO_MAX_VAR];
    int32_t n_snp, snp_pos[DDO_MAX_VAR], snp_rpos[DDO_MAX_VAR];
} ddo_out;

/* what computeBMidPrior takes from the mate and the library when mapUnmappedReads is on (ObservationModelFB.cpp:279-292) */
typedef struct ddo_mate {
    int paired, mate_unmapped, mate_reverse, same_tid;   /* read.isPaired(), mateIsUnmapped(), mateIsReverse(), tid == mtid */
    int32_t mate_pos, mate_len;                          /* read.matePos, read.mateLen (-1 unknown) */
    const double *lib_prob; int maxins; double p95;      /* Library::probs, maxins, ninetyfifth_pct_prob */
} ddo_mate;

/* one (haplotype, read) pair: ObservationModelFBMaxErr(hap, read, hapStart, params).calcLikelihood() */
int ddo_pair(const char *hap, int Hs, const char *readseq, const double *qual, int L,
             double mapQual, uint32_t readStartU32, uint32_t hapStart, int unmapped,
             const dd_params *P, ddo_out *out, int *hpos /* [L] */);

/* the same with the insert-size prior inputs (mate == NULL: none) */
int ddo_pair_mate(const char *hap, int Hs, const char *readseq, const double *qual, int L,
                  double mapQual, uint32_t readStartU32, uint32_t hapStart, int unmapped,
                  const dd_params *P, const ddo_mate *mate, ddo_out *out, int *hpos);

/* sibling model ObservationModelFBMax (not on the production path): only to check its three SURVEY §8(c) values */
int ddo_pair_fbmax(const char *hap, int Hs, const char *readseq, const double *qual, int L,
                   double mapQual, uint32_t readStartU32, uint32_t hapStart, int unmapped,
                   const dd_params *P, ddo_out *out, int *hpos);

/* a batch in the product's flat layout; windows [first_window, first_window+n_win) (n_win<0: all) */
int ddo_batch(const dd_params *P, const dd_batch *B, dd_result *R, int nthreads, int64_t first_window, int64_t n_win);

/* secondary path: ObservationModelS(hap, read, hapStart, params).align(HapHash(4, hap)) — Faster.cpp, Haplotype.hpp:315-384 */
int ddo_pair_fast(const char *hap, int hlen, const char *readseq, const double *qual, int rlen,
                  double mapQual, uint32_t readStartU32, uint32_t hapStart, const dd_params *P, ddo_out *out, int *hpos);

/* the same for a batch (DetInDel::computeLikelihoodsFaster, DInDel.cpp:1790-1833) */
int ddo_batch_fast(const dd_params *P, const dd_batch *B, dd_result *R, int nthreads, int64_t first_window, int64_t n_win);

/* N1: S[w][h1*H+h2] = sum_r log(0.5)+addLogs(ll[h1][r], ll[h2][r]) (reference DInDel.cpp:3085-3091) */
int ddo_pair_sums(const dd_batch *B, const double *ll, double *out);

/* pieces exported for the check against oracle/_ref (reference headers compiled as they are) */
double ddo_hp_error(int hpLen);
double ddo_add_logs(double l1, double l2);

#ifdef __cplusplus
}
#endif
#endif
