// genotype_kernel.hip — "next" row N1: the read-sum at the heart of the diploid genotype reduction.
//
// For every window and every unordered pair of candidate haplotypes (h1 <= h2) the reference accumulates
//     ll = 0;  for r in reads (in order):  ll += log(0.5) + addLogs(rl[r,h1], rl[r,h2]);
// (DetInDel::diploidGLF, reference DInDel.cpp:3085-3091 and again at :3372-3374), with
//     addLogs(l1,l2) = max + log(1 + exp(min - max))                          (reference Utils.hpp:29-38)
// and rl[r,h] = liks[h][r].ll — exactly the array the likelihood kernel leaves in HBM.  Everything around it
// (priors from the candidate file, haplotype filtering, argmax over pairs, qual) stays on the host
// (dindel_tgi_amd/host/genotype.cpp).
//
// One wavefront per (window, h1, h2): lanes evaluate the per-read terms in parallel, then the terms are added
// serially in read order (v_readlane chain) because the reference's sum is sequential and fp64 + is not
// associative.  exp/log come from the device math library and may differ from glibc in the last ulp, so this
// kernel is checked against the CPU within 1e-12 relative, not bit-for-bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hmm_kernel.h"

namespace ddk {

__device__ __forceinline__ double add_logs(double l1, double l2)
{
    if (l1 > l2) { const double diff = l2 - l1; return l1 + log(1.0 + exp(diff)); }
    const double diff = l1 - l2;
    return l2 + log(1.0 + exp(diff));
}

__global__ void __launch_bounds__(256) dd_pair_sum_kernel(PairSumArgs P)
{
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (slot >= P.n_slots) return;
    // window of this slot: binary search in the prefix of H_w^2
    int lo = 0, hi = P.n_windows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (P.win_hh_off[mid] <= slot) lo = mid; else hi = mid;
    }
    const int w = lo;
    const int H = P.win_hap_off[w + 1] - P.win_hap_off[w];
    const int R = P.win_read_off[w + 1] - P.win_read_off[w];
    const int idx = (int)(slot - P.win_hh_off[w]);
    const int h1 = idx / H, h2 = idx - h1 * H;
    if (h2 < h1) { if (lane == 0) P.out[slot] = 0.0; return; }     // lower triangle is never read (:3073)
    const double *l1 = P.ll + P.win_pair_off[w] + (int64_t)h1 * R;
    const double *l2 = P.ll + P.win_pair_off[w] + (int64_t)h2 * R;
    const double log5 = log(0.5);
    double sum = 0.0;
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        double t = 0.0;
        if (r < R) t = log5 + add_logs(l1[r], l2[r]);
        const int lo32 = __double2loint(t), hi32 = __double2hiint(t);
        const int nb = (R - r0) < 64 ? (R - r0) : 64;
        for (int i = 0; i < nb; i++)
            sum += __hiloint2double(__builtin_amdgcn_readlane(hi32, i), __builtin_amdgcn_readlane(lo32, i));
    }
    if (lane == 0) P.out[slot] = sum;
}

// MAP haplotype pairs of DetInDel::diploidGLF (reference DInDel.cpp:3073-3118): per window
//     pairs_posterior[h1,h2] = S[h1,h2] + prior[h1,h2]            for unfiltered h1 <= h2                    (:3091)
//     (max_ll_indel, pair) = first strict maximum over pairs with a candidate indel on either haplotype      (:3103)
//     (max_ll_noindel, pair) = the same over pairs with none                                                 (:3108)
//     qual = -10 (ll_ref - addLogs(max_ll_indel, ll_ref)) / ln 10,  ll_ref = max_ll_noindel                  (:3116-3118)
// One wavefront per window; the reference scans (h1,h2) in row-major order with `>`, i.e. takes the FIRST maximum:
// lanes keep (value, slot) of their own ascending slots and the reduction prefers the smaller slot on equal values.
__global__ void __launch_bounds__(256) dd_map_pair_kernel(MapPairArgs P)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= P.n_windows) return;
    const int h0 = P.win_hap_off[w], H = P.win_hap_off[w + 1] - h0;
    const int64_t base = P.win_hh_off[w];
    double bi = -__builtin_huge_val(), bn = -__builtin_huge_val();
    int ii = 0x7fffffff, in_ = 0x7fffffff;
    for (int idx = lane; idx < H * H; idx += 64) {
        const int h1 = idx / H, h2 = idx - h1 * H;
        double pp = 0.0;
        if (h2 >= h1 && !P.filtered[h0 + h1] && !P.filtered[h0 + h2]) {
            pp = P.pair_sum[base + idx] + P.prior[base + idx];
            const bool cand = P.ncand[h0 + h1] > 0 || P.ncand[h0 + h2] > 0;
            if (cand && pp > bi) { bi = pp; ii = idx; }
            if (!cand && pp > bn) { bn = pp; in_ = idx; }
        }
        if (P.posterior) P.posterior[base + idx] = pp;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double obi = __shfl_xor(bi, off), obn = __shfl_xor(bn, off);
        const int oii = __shfl_xor(ii, off), oin = __shfl_xor(in_, off);
        if (obi > bi || (obi == bi && oii < ii)) { bi = obi; ii = oii; }
        if (obn > bn || (obn == bn && oin < in_)) { bn = obn; in_ = oin; }
    }
    if (lane == 0) {
        const bool hi = ii != 0x7fffffff && bi > -__builtin_huge_val(), hn = in_ != 0x7fffffff && bn > -__builtin_huge_val();
        P.pairs[4 * w + 0] = hi ? ii / H : -1;
        P.pairs[4 * w + 1] = hi ? ii % H : -1;
        P.pairs[4 * w + 2] = hn ? in_ / H : -1;
        P.pairs[4 * w + 3] = hn ? in_ % H : -1;
        const double mi = hi ? bi : -__builtin_huge_val(), mn = hn ? bn : -__builtin_huge_val();
        P.vals[3 * w + 0] = mi;
        P.vals[3 * w + 1] = mn;
        P.vals[3 * w + 2] = -10.0 * (mn - add_logs(mi, mn)) / log(10.0);
    }
}

hipError_t launch_map_pairs(const MapPairArgs &A, hipStream_t st)
{
    if (A.n_windows <= 0) return hipSuccess;
    hipLaunchKernelGGL(dd_map_pair_kernel, dim3((unsigned)((A.n_windows + 3) / 4)), dim3(256), 0, st, A);
    return hipGetLastError();
}

hipError_t launch_pair_sums(const PairSumArgs &A, hipStream_t st)
{
    if (A.n_slots <= 0) return hipSuccess;
    const int64_t blocks = (A.n_slots + 3) / 4;
    hipLaunchKernelGGL(dd_pair_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A);
    return hipGetLastError();
}

} // namespace ddk
