// capi.cpp — the C ABI of include/dindel_hmm.h: host-side tables, validation, device staging, launch.
//
// Host work here is O(bases) bookkeeping only (offsets, validation, log tables per distinct quality).
// All likelihood arithmetic happens in hmm_kernel.hip; there is no CPU path for it in this library.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>
#include "hmm_kernel.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(DD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

// ReadIndelErrorModel::getViterbiHPError — reference ReadIndelErrorModel.hpp:36-50
double hp_error(int hpLen)
{
    static const double base[10] = {2.9e-5, 2.9e-5, 2.9e-5, 2.9e-5, 4.3e-5, 1.1e-4, 2.4e-4, 5.7e-4, 1.0e-3, 1.4e-3};
    int len = hpLen < 1 ? 1 : hpLen;
    double pbe = (len <= 10) ? base[len - 1] : base[9] + 4.3e-4 * double(len - 10);
    pbe *= double(hpLen);
    if (pbe > 0.99) pbe = 0.99;
    return pbe;
}

// Lane tilings, by haplotype length.  numS = Hs + 2 states go over the 64 lanes of a wavefront, K positions per lane (every K = 1..12 is
// instantiated, so no shape pays for more than 63 positions it does not have) — or, round 4, over the 32 lanes of HALF a wavefront, two
// pairs side by side (G = 2), where 32 K is the tighter fit: 127..158 bp run as K = 5 halves (2.5 lane-positions per pair instead of 3),
// 63..94 bp as K = 3 halves (1.5 instead of 2), <= 30 bp as K = 1 halves (0.5 instead of 1).
// Measured against the whole-wavefront tilings (tools/tiling_sweep.py, profiles/r04/tiling_sweep*.jsonl; 8 x 200 reads of 100 bp): <= 30 bp
// x 1.8, 63..94 bp + 1-7 %, 127..158 bp + 14-18 % (+ 4-9 % in ragged batches whose reads differ in length inside a window: the two reads
// of a wavefront run the longer one's trip counts).  31..62 bp as K = 2 halves were level with K = 1 on a whole wavefront (its FOLD build)
// and 191..222 bp as K = 7 halves gained 4-10 % on uniform windows of 100-bp reads but LOST 26-33 % on ragged ones with 150-bp reads (their
// LDS rows leave a CU 6 wavefronts, profiles/r04/wide_sample_ab.jsonl): both classes stay on a whole wavefront (the 191..222-bp class keeps
// its own launch: K = 4 like its neighbour).
struct HapClassDef { int bound, G, K; };     // haplotypes up to `bound` bp: G pairs per wavefront, K positions per lane
const HapClassDef kHapClasses[DD_N_HAP_CLASSES] = {
    {30, 2, 1}, {62, 1, 1}, {94, 2, 3}, {126, 1, 2}, {158, 2, 5}, {190, 1, 3}, {222, 1, 4}, {254, 1, 4},
    {318, 1, 5}, {382, 1, 6}, {446, 1, 7}, {510, 1, 8}, {574, 1, 9}, {638, 1, 10}, {702, 1, 11}, {DD_MAX_HAP_LEN, 1, 12}};
bool half_wave_off() { return getenv("DD_NO_HALF") != nullptr; }   // A/B and tests: whole-wavefront tilings only
int hap_class_of(int hap_len)
{
    for (int c = 0; c < DD_N_HAP_CLASSES; c++)
        if (hap_len <= kHapClasses[c].bound) return c;
    return -1;
}
// tiling for a launch whose longest haplotype is max_hap_len: false if it is too long
bool pick_tiling(int max_hap_len, int Dt, int &G, int &K)
{
    const int c = hap_class_of(max_hap_len);
    if (c < 0) return false;
    G = kHapClasses[c].G; K = kHapClasses[c].K;
    // (the half tilings in use gain at the D = 6, 11 and 12 builds: 9-16 % at maxLengthDel 10 / 11; the D = 32 build exists for whole
    // wavefronts only, up to K = 9 — 574 bp — because a position's back-pointer takes 7 bits there)
    if (G > 1 && (half_wave_off() || Dt > 12)) { G = 1; K = (kHapClasses[c].bound + 2 + 63) / 64; }
    if (Dt > 12 && K > 9) return false;
    return true;
}

int pick_Dt(int D)
{
    // every build switches candidates y > D off with -inf constants, so a smaller D runs on the next larger build
    // (maxLengthDel 0..4 on the D=6 build: 2.4e11 -> 4.0e11 cells/s at configs[1]; 6..9 on the D=11 build)
    if (D <= 6) return 6;
    if (D <= 11) return 11;
    if (D <= 12) return 12;
    return 32;                                    // maxLengthDel 12..31: the one build for the values beyond the reference's defaults (5 / 10)
}

uint32_t up16(uint32_t v) { return (v + 15u) & ~15u; }

// LDS carve-up for (K, Dt, Lmax); returns total dynamic LDS bytes per workgroup
size_t lds_layout(int K, int Dt, int Lmax, int n_qual, int waves, bool gbt, int G, ddk::KernelArgs &A)
{
    const uint32_t W = 64u / (uint32_t)G;        // lanes per pair
    const uint32_t NP = W * K;
    uint32_t o = 0;
    o = up16(NP + 16);
    A.lds_off_L = o;  o += 256;                  // byte -> symbol id table
    A.lds_off_E = o;  o += up16((NP + Dt + 2) * 8);
    A.lds_off_N = o;  o += up16((NP + Dt + 2) * 8);
    A.lds_off_Q = o;  o += up16((uint32_t)n_qual * 32);
    A.n_qual = n_qual;
    A.lds_off_C = o;  A.lds_off_Y = o;
    if ((gbt || G > 1) && (Dt > 7 || K >= 3)) {  // LEAN build: block-shared Inc constants + (y-1)*II (hmm_kernel.hip)
        if (Dt <= 12) o += up16((uint32_t)K * Dt * W * 8u);   // (the D = 32 build forms these constants on the fly)
        A.lds_off_Y = o;  o += up16((uint32_t)Dt * 8u);
    }
    A.lds_off_W = o;  o += 16;                   // the workgroup's work counter
    A.lds_off_S = o;
    if (G > 1) o += up16(DD_HALF_CHUNK * 6u);    // half-wave builds: sort keys (u32) + order (u16) of a chunk of the window's reads
    A.lds_shared_bytes = o;
    uint32_t wv = 0;
    A.lds_off_A = wv;   wv += up16((uint32_t)K * (W + 2u * (uint32_t)((Dt + K - 1) / K)) * 16u);   // K arrays of {value, emission} + pads
    A.lds_off_I = wv;   wv += up16((NP + 2) * 8);
    A.lds_off_rdE = wv; wv += up16(Lmax * 16);
    A.lds_off_rdC = wv; wv += up16(Lmax);
    A.lds_off_rdQ = wv; wv += up16(Lmax);
    A.lds_off_ms = wv;  wv += up16(Lmax * 2);
    A.lds_group_bytes = wv;                      // the rows above exist once per pair of the wavefront; the back-pointer tile is the wavefront's
    wv *= (uint32_t)G;
    {   // packed back-pointers: one word of K*(CB+1) bits per lane per read base (BtPack in hmm_kernel.hip)
        const uint32_t bits = (uint32_t)K * ((Dt <= 7) ? 4u : (Dt <= 15 ? 5u : 7u));
        const uint32_t bytes = bits <= 8 ? 1 : bits <= 16 ? 2 : bits <= 32 ? 4 : 8;
        A.lds_off_bt = wv;
        if (!gbt) wv += up16((uint32_t)Lmax * 64u * bytes);   // GBT builds keep the tile in HBM scratch
    }
    A.lds_wave_bytes = wv;
    return (size_t)A.lds_shared_bytes + (size_t)waves * wv;
}

// waves per CU the register file allows for each K (kernel-resource-usage of the shipped builds)
int reg_limited_waves_per_cu(int K, int Dt, bool gbt, int G = 1)
{
    (void)G;
    if (Dt > 12) return K <= 3 ? 12 : (K <= 6 ? 8 : 4);   // the D = 32 build keeps a lane's own positions only: 125-144 registers up to K = 3, <= 244 up to K = 6
    if (const char *e = getenv("DD_REG_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 16) return v; }   // A/B builds with another occupancy
    if (K <= 2) return (Dt <= 7 || gbt) ? 12 : 8;
    if (K == 3) return (gbt && Dt <= 7) ? 12 : ((gbt || Dt <= 7) ? 8 : 4);      // round 3: the D = 6 scratch build is held to 168 VGPRs (3 waves/SIMD)
    if (K == 4) return gbt ? 8 : 4;
    if (K == 5 || G == 2) return gbt ? 8 : 4;    // round 4: the K = 5 scratch build held to 2 waves per SIMD gains 34 % over 1 (profiles/r04/occupancy_ab.txt)
    return 4;
}

uint32_t bt_word_bytes(int K, int Dt)
{
    const uint32_t bits = (uint32_t)K * ((Dt <= 7) ? 4u : (Dt <= 15 ? 5u : 7u));
    return bits <= 8 ? 1 : bits <= 16 ? 2 : bits <= 32 ? 4 : 8;
}

// the workspace starts with the item counter of the dynamic (ragged) launches; the back-pointer tiles follow
static const size_t DD_WS_HEADER = 256;

// bytes of one wavefront's region of the HBM scratch: its back-pointer tile + (K >= 3 or half-wave builds) the [2 K][64] doubles where
// beta[bMid] waits for the join (hmm_kernel.hip STASH)
size_t scratch_wave_bytes(int K, int Dt, int G, int max_read_len)
{
    const bool slim = G > 1 || K == 3 || K == 5 || (K == 4 && Dt <= 7);       // hmm_kernel.hip SLIM
    return (size_t)max_read_len * 64u * bt_word_bytes(K, Dt) + (slim ? (size_t)2 * K * 64 * 8 : 0);
}

// Launch plan: LDS-resident back-pointers when that keeps the CU as full as the registers allow, otherwise
// the HBM-scratch build (GBT) with a persistent grid (one scratch tile per resident wave).
struct Plan {
    int K, Dt, waves, waves_per_cu;
    int G = 1;               // pairs per wavefront (2: the half-wave builds)
    bool gbt;
    bool two_waves = false;  // K = 3 / D = 6 scratch build: the variant compiled for 2 waves per SIMD (LDS keeps fewer than 12 waves on the CU anyway)
    size_t lds, scratch_bytes;
    unsigned grid_cap;       // 0 = one workgroup per item
};

int make_plan(const dd_params *p, int max_hap_len, int max_read_len, int n_qual, Plan &pl, ddk::KernelArgs &A)
{
    pl.Dt = pick_Dt(p->maxLengthDel + 1);
    if (!pick_tiling(max_hap_len, pl.Dt, pl.G, pl.K))
        return fail(DD_ERR_UNSUPPORTED, pl.Dt > 12 ? "haplotype longer than 574 bp with maxLengthDel > 11" : "haplotype too long");
    int best[2] = {0, 0}, bw[2] = {0, 0}, cap[2] = {0, 0};
    for (int gbt = 0; gbt < 2; gbt++) {
        const int reg_cap = reg_limited_waves_per_cu(pl.K, pl.Dt, gbt != 0, pl.G);
        cap[gbt] = reg_cap;
        for (int wv = DD_WAVES; wv >= 1; wv--) {
            ddk::KernelArgs tmp = A;
            const size_t l = lds_layout(pl.K, pl.Dt, max_read_len, n_qual, wv, gbt != 0, pl.G, tmp);
            if (l > 160u * 1024u) continue;
            int total = (int)((160u * 1024u) / l) * wv;
            if (total > reg_cap) total = reg_cap;
            if (total > best[gbt]) { best[gbt] = total; bw[gbt] = wv; }
        }
    }
    if (pl.Dt > 12) best[0] = 0;                   // the D = 32 build keeps its back-pointers in the HBM scratch only
    if (best[0] == 0 && best[1] == 0)
        return fail(DD_ERR_UNSUPPORTED, "read length x haplotype length does not fit the LDS row buffers");
    // K = 3 / D = 6 scratch: the 3-waves-per-SIMD build only where LDS lets 12 waves stay (reads up to ~250 bp); beyond, the build
    // for 2 waves per SIMD (no spills) with the geometry that fills 8: 9 waves of the spilling build lost 9 % to it at 400-bp reads
    pl.two_waves = false;
    if (pl.G == 1 && pl.K == 3 && pl.Dt <= 7 && best[1] > 0 && best[1] < 12) {
        pl.two_waves = true;
        best[1] = 0; bw[1] = 0; cap[1] = 8;
        for (int wv = DD_WAVES; wv >= 1; wv--) {
            ddk::KernelArgs tmp = A;
            const size_t l = lds_layout(pl.K, pl.Dt, max_read_len, n_qual, wv, true, pl.G, tmp);
            if (l > 160u * 1024u) continue;
            int total = (int)((160u * 1024u) / l) * wv;
            if (total > 8) total = 8;
            if (total > best[1]) { best[1] = total; bw[1] = wv; }
        }
    }
    // HBM scratch costs a coalesced row fetch per 8 traceback steps and (D=11) block-shared constants; the LDS tile
    // costs occupancy.  Measured over six shapes (tools/ab_point.py with DD_FORCE_GBT=0/1): the LDS build wins
    // whenever its tile still lets the CU hold as many waves as its registers allow, the scratch build wins
    // (5-80 %) once LDS caps it below that (tools/plan_check.py grid: at 10 of 12 waves the scratch build is already
    // 14 % ahead).
    // For K >= 3 the scratch build is also the register-lean one (block-shared constants) and wins at every
    // shape measured (+26 ... +41 % in round 2; +10 ... +30 % on the round-3 grid) — except K = 3 at D = 6 with reads short
    // enough (<= 90 bp) for the LDS tile to keep the 8 waves its registers allow: there the LDS build is 4-10 % ahead of the
    // scratch build (profiles/r03/plan_check.jsonl, k3_lds_vs_scratch.jsonl), so that case follows the K <= 2 rule.
    // Round 4 (profiles/r04/plan_check.jsonl): K = 4 at D = 6 with reads up to ~80 bp is 17-18 % faster on the LDS build too (at 100 bp it loses 20 %).
    // End of round 4 (profiles/r04/plan_check.jsonl): with the item counter on every multi-round launch the scratch builds gained 10-18 % and the LDS builds
    // 1 %, and the exceptions above lost: K = 4 / D = 6 with reads <= 80 bp ran 35-45 % BEHIND on the LDS build, K = 3 / D = 6 with short reads 4-12 %, and
    // K = 2 on the D = 11 build 5-8 % at every read length the LDS tile fits.  Now: scratch for every K >= 3 and for K = 2 above D = 7.
    const bool lean_only = pl.K >= 3 || (pl.K == 2 && pl.Dt > 7);
    pl.gbt = best[0] == 0 || (lean_only && best[1] > 0) || best[0] < cap[0];
    if (const char *f = getenv("DD_FORCE_GBT")) {                  // A/B only
        if (f[0] == '1' && best[1] > 0) pl.gbt = true;
        if (f[0] == '0' && best[0] > 0) pl.gbt = false;
    }
    pl.waves = bw[pl.gbt ? 1 : 0];
    pl.waves_per_cu = best[pl.gbt ? 1 : 0];
    pl.lds = lds_layout(pl.K, pl.Dt, max_read_len, n_qual, pl.waves, pl.gbt, pl.G, A);
    pl.grid_cap = 0;
    pl.scratch_bytes = 0;
    if (pl.gbt) {
        const int blocks_per_cu = (pl.waves_per_cu + pl.waves - 1) / pl.waves;
        pl.grid_cap = 256u * (unsigned)blocks_per_cu;
        pl.scratch_bytes = (size_t)pl.grid_cap * pl.waves * scratch_wave_bytes(pl.K, pl.Dt, pl.G, max_read_len);
    }
    return DD_SUCCESS;
}

int check_params(const dd_params *p)
{
    if (!p) return fail(DD_ERR_INVALID, "null params");
    if (p->forceReadOnHaplotype) return fail(DD_ERR_UNSUPPORTED, "forceReadOnHaplotype is not on the production path (DInDel.cpp:1446 only)");
    if (p->maxLengthDel < 0 || p->maxLengthDel > DD_MAX_LENGTH_DEL) return fail(DD_ERR_UNSUPPORTED, "maxLengthDel outside [0,31]");
    if (!(p->pError > 0.0 && p->pError < 1.0)) return fail(DD_ERR_INVALID, "pError outside (0,1)");
    return DD_SUCCESS;
}

// Per-host-thread cache kept between calls of the host-pointer entry points: one device arena, a pinned host
// mirror for small batches and two streams.  The literal drop-in use (one window per call) is dominated by
// allocation / copy-call overheads otherwise (1.5 ms per call with ~30 hipMalloc + ~35 hipMemcpy).
struct DeviceCtx {
    int device = -1;
    unsigned char *arena = nullptr;  size_t arena_cap = 0;
    unsigned char *pinned = nullptr; size_t pinned_cap = 0;
    hipStream_t s[2] = {nullptr, nullptr};
    void release()
    {
        if (device < 0) return;
        (void)hipSetDevice(device);
        if (arena) (void)hipFree(arena);
        if (pinned) (void)hipHostFree(pinned);
        for (int i = 0; i < 2; i++) if (s[i]) (void)hipStreamDestroy(s[i]);
        arena = pinned = nullptr; arena_cap = pinned_cap = 0; s[0] = s[1] = nullptr; device = -1;
    }
    int reserve(int dev, size_t dev_bytes, size_t pinned_bytes)
    {
        if (device != dev) { release(); device = dev; }
        hipError_t e;
        if (dev_bytes > arena_cap) {
            if (arena) (void)hipFree(arena);
            arena = nullptr; arena_cap = 0;
            const size_t want = dev_bytes + dev_bytes / 8 + (1u << 20);
            void *p = nullptr;
            if ((e = hipMalloc(&p, want)) != hipSuccess) return fail(DD_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
            arena = static_cast<unsigned char *>(p); arena_cap = want;
        }
        if (pinned_bytes > pinned_cap) {
            if (pinned) (void)hipHostFree(pinned);
            pinned = nullptr; pinned_cap = 0;
            const size_t want = pinned_bytes + pinned_bytes / 4 + (1u << 16);
            void *p = nullptr;
            if ((e = hipHostMalloc(&p, want, hipHostMallocDefault)) != hipSuccess) return fail(DD_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
            pinned = static_cast<unsigned char *>(p); pinned_cap = want;
        }
        for (int i = 0; i < 2; i++)
            if (!s[i] && (e = hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)) != hipSuccess)
                return fail(DD_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        return DD_SUCCESS;
    }
};
// A host thread that ends gives its arena, pinned mirror and streams back (short-lived worker threads would otherwise
// leak them without bound).  Once exit() has begun the HIP runtime may already be unloading: then the cache is left to the
// process teardown.  The flag is raised by an atexit handler registered when the first cache is created, i.e. after the HIP
// runtime registered its own, so it runs before them.
std::atomic<bool> g_process_exiting(false);
void note_process_exit() { g_process_exiting.store(true); }
struct CtxHolder {
    DeviceCtx c;
    CtxHolder() { static const int once = atexit(note_process_exit); (void)once; }
    ~CtxHolder() { if (!g_process_exiting.load()) c.release(); }
};
thread_local CtxHolder g_ctx;

// Bump allocator over the cached arena.  In `staged` mode uploads are memcpy'd into the pinned mirror at the same
// offsets and shipped with ONE hipMemcpyAsync (flush_uploads); otherwise each upload is its own (large) copy.
struct DevBuf {
    DeviceCtx &C;
    size_t used = 0;
    bool staged = false;
    explicit DevBuf(DeviceCtx &c) : C(c) {}
    static size_t align(size_t v) { return (v + 255u) & ~size_t(255u); }
    template <class T> int alloc(T **out, size_t n)
    {
        const size_t off = align(used), bytes = (n ? n : 1) * sizeof(T);
        if (off + bytes > C.arena_cap) return fail(DD_ERR_HIP, "internal: device arena under-sized");
        used = off + bytes;
        *out = reinterpret_cast<T *>(C.arena + off);
        return DD_SUCCESS;
    }
    template <class T> int upload(const T **out, const T *src, size_t n)
    {
        T *d = nullptr;
        int rc = alloc(&d, n);
        if (rc) return rc;
        if (n) {
            if (staged) {
                memcpy(C.pinned + (reinterpret_cast<unsigned char *>(d) - C.arena), src, n * sizeof(T));
            } else {
                hipError_t e = hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice);
                if (e != hipSuccess) return fail(DD_ERR_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
            }
        }
        *out = d;
        return DD_SUCCESS;
    }
    int flush_uploads(hipStream_t st)
    {
        if (!staged || !used) return DD_SUCCESS;
        hipError_t e = hipMemcpyAsync(C.arena, C.pinned, used, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(DD_ERR_HIP, std::string("hipMemcpyAsync H2D: ") + hipGetErrorString(e));
        return DD_SUCCESS;
    }
};

} // namespace

static thread_local int32_t g_last_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per host thread, like the cache and the error string
static thread_local char g_kernel_name[64] = "dd_hmm_kernel";
static thread_local int g_last_direct = 0;     // output arrays the last host-pointer call on this thread let the kernels write in place
static thread_local int g_last_fold = 0;       // the last main-model launch used the FOLD build (hmm_kernel.hip)
static thread_local int g_last_G = 1;          // ... pairs per wavefront of that launch
static thread_local int g_last_occ = 0;        // ... the build variant compiled for that many waves per SIMD (0 = the build's usual occupancy)

// Launch log of the last dd_launch_device / dd_compute_likelihoods call on this host thread (dd_launch_log): one record per main-model
// kernel launch.  With DD_LAUNCH_TIMING=1 (diagnostics) every launch is bracketed by HIP events and its duration is filled in when the
// log is read (dd_launch_log synchronises on them).
struct LaunchRec {
    int32_t v[DD_LAUNCH_LOG_FIELDS];
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
static thread_local std::vector<LaunchRec> g_launch_log;
static void launch_log_clear()
{
    for (auto &r : g_launch_log) {
        if (r.e0) (void)hipEventDestroy(r.e0);
        if (r.e1) (void)hipEventDestroy(r.e1);
    }
    g_launch_log.clear();
}

#ifdef DD_STAMPS
static unsigned long long *g_dbg = nullptr;
extern "C" void dd_debug_set_stamp_buffer(void *p) { g_dbg = static_cast<unsigned long long *>(p); }
#endif

extern "C" {

int dd_abi_version(void) { return DD_ABI_VERSION; }
const char *dd_last_error(void) { return g_err.c_str(); }
const char *dd_kernel_name(void)
{   // the template instance this host thread launched last, as rocprofv3 prints it (inside "void ddk::...(ddk::KernelArgs)")
    const int K = g_last_launch[0], D = g_last_launch[1];
    if (K > 0 && D > 0) snprintf(g_kernel_name, sizeof(g_kernel_name), "dd_hmm_kernel<%d, %d, %s, %s, %d, %d>", K, D % 100, D >= 100 ? "true" : "false", g_last_fold ? "true" : "false", g_last_occ, g_last_G);
    else if (K > 0) snprintf(g_kernel_name, sizeof(g_kernel_name), "dd_faster_kernel");
    return g_kernel_name;
}

int dd_last_direct_outputs(void) { return g_last_direct; }

void dd_last_launch(int32_t out[8])
{   // K, D build, waves per workgroup, LDS bytes per workgroup, grid, read split, LDS per wave, shared LDS
    for (int i = 0; i < 8; i++) out[i] = g_last_launch[i];
}

int dd_launch_log(int32_t *out, int max_records)
{   // see include/dindel_hmm.h
    const int n = (int)g_launch_log.size();
    for (int i = 0; i < n && i < max_records && out; i++) {
        LaunchRec &r = g_launch_log[(size_t)i];
        if (r.e0 && r.e1) {
            float ms = 0.f;
            if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) r.v[15] = (int32_t)(ms * 1000.f + 0.5f);
        }
        memcpy(out + (size_t)i * DD_LAUNCH_LOG_FIELDS, r.v, sizeof(r.v));
    }
    return n;
}

void dd_release_cache(void)
{   // frees this host thread's cached device arena, pinned mirror and streams
    g_ctx.c.release();
}

int dd_reserve_cache(int device, size_t device_bytes, size_t pinned_bytes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DD_ERR_NO_DEVICE, "no HIP device: the likelihood path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(DD_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return g_ctx.c.reserve(device, device_bytes, pinned_bytes);
}

void *dd_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void dd_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int dd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void dd_params_struct_defaults(dd_params *p)
{   // ObservationModelParameters::setDefaultValues — reference ObservationModel.hpp:39-64
    p->pError = 1e-4; p->pMut = 1e-4; p->pFirstgLO = 0.01; p->mapQualThreshold = 100.0;
    p->checkBaseQualThreshold = 0.95; p->maxLengthDel = 10; p->padCover = 5; p->bMid = -1;
    p->forceReadOnHaplotype = 0; p->mapUnmappedReads = 0; p->maxMismatch = 1; p->capMapQualFast = 40.0;
}

void dd_params_cli_defaults(dd_params *p)
{   // what main() installs — reference DInDel.cpp:3937-3949 with the option defaults at :4122-4157
    dd_params_struct_defaults(p);
    p->pError = 5e-4; p->pMut = 1e-5; p->maxLengthDel = 5; p->mapQualThreshold = 100.0; p->padCover = 2; p->maxMismatch = 2; p->capMapQualFast = 45.0;
}

int dd_batch_sizes(const dd_batch *b, dd_sizes *out)
{
    if (!b || !out) return fail(DD_ERR_INVALID, "null argument");
    if (b->n_windows < 0 || !b->win_hap_off || !b->win_read_off || !b->hap_seq_off || !b->read_seq_off)
        return fail(DD_ERR_INVALID, "null offset array");
    memset(out, 0, sizeof(*out));
    const int W = b->n_windows;
    out->n_haps = b->win_hap_off[W];
    out->n_reads = b->win_read_off[W];
    if (b->win_hap_off[0] != 0 || b->win_read_off[0] != 0 || b->hap_seq_off[0] != 0 || b->read_seq_off[0] != 0)
        return fail(DD_ERR_INVALID, "offset arrays must start at 0");
    for (int64_t h = 0; h < out->n_haps; h++) {
        int len = b->hap_seq_off[h + 1] - b->hap_seq_off[h];
        if (len < 0) return fail(DD_ERR_INVALID, "hap_seq_off not monotone");
        if (len > out->max_hap_len) out->max_hap_len = len;
    }
    for (int64_t r = 0; r < out->n_reads; r++) {
        int len = b->read_seq_off[r + 1] - b->read_seq_off[r];
        if (len < 0) return fail(DD_ERR_INVALID, "read_seq_off not monotone");
        if (len > out->max_read_len) out->max_read_len = len;
    }
    out->hap_bases = b->hap_seq_off[out->n_haps];
    out->read_bases = b->read_seq_off[out->n_reads];
    for (int w = 0; w < W; w++) {
        int64_t H = b->win_hap_off[w + 1] - b->win_hap_off[w];
        int64_t R = b->win_read_off[w + 1] - b->win_read_off[w];
        if (H < 0 || R < 0) return fail(DD_ERR_INVALID, "window offsets not monotone");
        int64_t SL = b->read_seq_off[b->win_read_off[w + 1]] - b->read_seq_off[b->win_read_off[w]];
        int64_t SH = b->hap_seq_off[b->win_hap_off[w + 1]] - b->hap_seq_off[b->win_hap_off[w]];
        int64_t nv = b->hap_var_off ? (b->hap_var_off[b->win_hap_off[w + 1]] - b->hap_var_off[b->win_hap_off[w]]) : 0;
        out->n_pairs += H * R;
        out->hpos_len += H * SL;
        out->var_cov_len += nv * R;
        out->cells += SH * SL;
    }
    return DD_SUCCESS;
}

// byte -> symbol id (dd_build_symbol_lut's rule).  Returns how many distinct haplotype bytes were left without an id (the 27th and later
// non-ACGTN values in byte order; they map to 31 like a byte no haplotype holds), or a DD_ERR_* code.
static int assign_symbols(const dd_batch *b, uint8_t *out)
{
    if (!b || !out || !b->hap_seq_off || !b->win_hap_off) return fail(DD_ERR_INVALID, "null argument");
    for (int i = 0; i < 256; i++) out[i] = 31;                    // read-only symbol: equal to no haplotype symbol
    out[(unsigned char)'A'] = 0; out[(unsigned char)'C'] = 1; out[(unsigned char)'G'] = 2; out[(unsigned char)'T'] = 3;
    out[(unsigned char)'N'] = 4;
    bool seen[256] = {false};
    const int64_t n_haps = b->n_windows > 0 ? b->win_hap_off[b->n_windows] : 0;
    const int64_t nb = n_haps > 0 ? b->hap_seq_off[n_haps] : 0;
    if (nb > 0 && !b->hap_seq) return fail(DD_ERR_INVALID, "null input array");
    for (int64_t i = 0; i < nb; i++) seen[(unsigned char)b->hap_seq[i]] = true;
    int next = 5, left = 0;
    for (int c = 0; c < 256; c++) {
        if (!seen[c] || c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') continue;
        if (next > 30) { left++; continue; }
        out[c] = (uint8_t)next++;
    }
    return left;
}

// The screen proper.  sym_lut (may be NULL: lengths only) is assign_symbols' table; with_symbols: windows whose haplotypes hold a byte
// that got no id are skipped too (main model only: the --faster kernel compares the bytes themselves).
static int screen_windows(const dd_batch *b, uint8_t *win_skip, int32_t max_len_out[2], const uint8_t *sym_lut, bool with_symbols)
{
    int n_skip = 0, mh = 0, mr = 0;
    for (int w = 0; w < b->n_windows; w++) {
        int wh = 0, wr = 0;
        bool bad = false;
        for (int h = b->win_hap_off[w]; h < b->win_hap_off[w + 1]; h++) {
            const int len = b->hap_seq_off[h + 1] - b->hap_seq_off[h];
            if (len < 1 || len > DD_MAX_HAP_LEN) bad = true;
            if (len > wh) wh = len;
        }
        for (int q = b->win_read_off[w]; q < b->win_read_off[w + 1]; q++) {
            const int len = b->read_seq_off[q + 1] - b->read_seq_off[q];
            if (len < 1 || len > DD_MAX_READ_LEN) bad = true;
            if (len > wr) wr = len;
        }
        if (with_symbols && !bad) {
            const int64_t from = b->hap_seq_off[b->win_hap_off[w]], to = b->hap_seq_off[b->win_hap_off[w + 1]];
            for (int64_t i = from; i < to && !bad; i++) {
                const unsigned char c = (unsigned char)b->hap_seq[i];
                if (sym_lut[c] == 31) bad = true;          // a haplotype byte is never 31 unless it was left without an id
            }
        }
        win_skip[w] = bad ? 1 : 0;
        if (bad) { n_skip++; continue; }
        if (b->win_hap_off[w + 1] > b->win_hap_off[w] && b->win_read_off[w + 1] > b->win_read_off[w]) {   // windows with pairs
            if (wh > mh) mh = wh;
            if (wr > mr) mr = wr;
        }
    }
    if (max_len_out) { max_len_out[0] = mh; max_len_out[1] = mr; }
    return n_skip;
}

int dd_screen_windows(const dd_batch *b, uint8_t *win_skip, int32_t max_len_out[2])
{
    if (!b || !win_skip) return fail(DD_ERR_INVALID, "null argument");
    if (b->n_windows < 0 || !b->win_hap_off || !b->win_read_off || !b->hap_seq_off || !b->read_seq_off)
        return fail(DD_ERR_INVALID, "null offset array");
    uint8_t lut[256];
    const int left = assign_symbols(b, lut);
    if (left < 0) return left;
    return screen_windows(b, win_skip, max_len_out, lut, left > 0);
}

int dd_batch_offsets(const dd_batch *b, int64_t *win_pair_off, int64_t *win_hpos_off, int64_t *win_varcov_off)
{
    if (!b) return fail(DD_ERR_INVALID, "null batch");
    int64_t p = 0, hp = 0, vc = 0;
    for (int w = 0; w < b->n_windows; w++) {
        if (win_pair_off) win_pair_off[w] = p;
        if (win_hpos_off) win_hpos_off[w] = hp;
        if (win_varcov_off) win_varcov_off[w] = vc;
        int64_t H = b->win_hap_off[w + 1] - b->win_hap_off[w];
        int64_t R = b->win_read_off[w + 1] - b->win_read_off[w];
        int64_t SL = b->read_seq_off[b->win_read_off[w + 1]] - b->read_seq_off[b->win_read_off[w]];
        int64_t nv = b->hap_var_off ? (b->hap_var_off[b->win_hap_off[w + 1]] - b->hap_var_off[b->win_hap_off[w]]) : 0;
        p += H * R; hp += H * SL; vc += nv * R;
    }
    if (win_pair_off) win_pair_off[b->n_windows] = p;
    if (win_hpos_off) win_hpos_off[b->n_windows] = hp;
    if (win_varcov_off) win_varcov_off[b->n_windows] = vc;
    return DD_SUCCESS;
}

int dd_build_index(const dd_batch *b, int32_t *hap_window, int64_t *win_pair_off, int64_t *win_hpos_off, int64_t *win_varcov_off)
{
    int rc = dd_batch_offsets(b, win_pair_off, win_hpos_off, win_varcov_off);
    if (rc) return rc;
    if (hap_window)
        for (int w = 0; w < b->n_windows; w++)
            for (int h = b->win_hap_off[w]; h < b->win_hap_off[w + 1]; h++) hap_window[h] = w;
    return DD_SUCCESS;
}

int dd_build_library_tables(const dd_batch *b, double *logprob_out, double *log95_out)
{
    if (!b || !logprob_out || !log95_out) return fail(DD_ERR_INVALID, "null argument");
    if (b->n_libs < 1 || b->n_libs > 256 || !b->lib_off || !b->lib_prob || !b->lib_p95)
        return fail(DD_ERR_INVALID, "mapUnmappedReads needs 1..256 libraries (lib_off, lib_prob, lib_p95)");
    if (b->lib_off[0] != 0) return fail(DD_ERR_INVALID, "lib_off[0] must be 0");
    for (int i = 0; i < b->n_libs; i++) {
        if (b->lib_off[i + 1] - b->lib_off[i] < 1) return fail(DD_ERR_INVALID, "empty library table");
        if (!(b->lib_p95[i] > 0.0)) return fail(DD_ERR_INVALID, "library probabilities must be positive");
        log95_out[i] = log(b->lib_p95[i]);                                   // ObservationModelFB.cpp:289
    }
    for (int i = 0; i < b->lib_off[b->n_libs]; i++) {
        if (!(b->lib_prob[i] > 0.0)) return fail(DD_ERR_INVALID, "library probabilities must be positive");
        logprob_out[i] = log(b->lib_prob[i]);                                // :285, :287
    }
    return DD_SUCCESS;
}

int dd_build_symbol_lut(const dd_batch *b, uint8_t *out)
{
    const int left = assign_symbols(b, out);
    return left < 0 ? left : DD_SUCCESS;        // bytes left without an id: their windows are dd_screen_windows' business
}

int dd_build_tables(const dd_params *p, const double *qual_table, int n_qual, const double *mapq_table, int n_mapq, double *out)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n_qual < 0 || n_qual > DD_MAX_QUAL_TABLE || n_mapq < 0 || n_mapq > DD_MAX_QUAL_TABLE)
        return fail(DD_ERR_INVALID, "quality table larger than 256 entries");
    for (int i = 0; i < DD_TABLE_DOUBLES; i++) out[i] = 0.0;
    // ObservationModelFBMaxErr::setupTransitionProbs — reference ObservationModelFB.cpp:1643-1667
    const double logpInsgIns = -.5;
    const double logpInsgNoIns = log(p->pError);
    out[TC_LLL] = log(1.0 - p->pFirstgLO);
    out[TC_LFL] = log(p->pFirstgLO);
    out[TC_II] = logpInsgIns;
    out[TC_NI] = log(1.0 - exp(logpInsgIns));
    out[TC_IN] = logpInsgNoIns;
    out[TC_NN] = log(1 - p->pError);
    out[TC_EDEF] = log(1e-5);           // :1678
    out[TC_NDEF] = log(1 - 1e-5);       // :1679
    out[TC_BQT] = p->checkBaseQualThreshold;
    // emissions — setupReadObservationPotentials, reference ObservationModelFB.cpp:226-234
    for (int i = 0; i < n_qual; i++) {
        const double rq = qual_table[i];
        const double pr = rq * (1.0 - p->pMut);
        out[T_QUAL + 4 * i + 0] = log(.25 + .75 * pr);
        out[T_QUAL + 4 * i + 1] = log(.75 + 1e-10 - .75 * pr);
        out[T_QUAL + 4 * i + 2] = log10(1.0 - rq);      // mLogBQ term, :1406
        out[T_QUAL + 4 * i + 3] = rq;
    }
    // bMid prior — computeBMidPrior, reference ObservationModelFB.cpp:268-303 with pinsert == 0
    for (int i = -1; i < n_mapq; i++) {
        const double mapQual = (i < 0) ? (1.0 - 1e-10) : mapq_table[i];   // i<0: the "HMQ" prior of :1093
        double mq = 1.0 - mapQual;
        if (-10.0 * log10(mq) > p->mapQualThreshold) mq = pow(10.0, -p->mapQualThreshold / 10.0);
        const double pOffFirst = mq;
        const double pinsert = 0.0;
        double *dst = (i < 0) ? &out[TC_HMQ] : &out[T_MAPQ + 4 * i];
        for (int k = 0; k < 2; k++) {
            const double logpIns = (k == 1) ? logpInsgNoIns : log(1.0 - exp(logpInsgNoIns));
            dst[0 + k] = log(pOffFirst) + logpIns + pinsert;          // prior[i*numS+0]
            dst[2 + k] = pinsert + log((1.0 - pOffFirst)) + logpIns;  // prior[i*numS+x], 1<=x<=hapSize
        }
    }
    // insert-size prior path (mapUnmappedReads): the separate terms of computeBMidPrior — :272-276, :296-303
    out[TC_PINS + 0] = log(1.0 - exp(logpInsgNoIns));
    for (int i = -1; i < n_mapq; i++) {
        const double mapQual = (i < 0) ? (1.0 - 1e-10) : mapq_table[i];
        double mq = 1.0 - mapQual;
        if (-10.0 * log10(mq) > p->mapQualThreshold) mq = pow(10.0, -p->mapQualThreshold / 10.0);
        double *dst = (i < 0) ? &out[TC_PINS + 1] : &out[T_MAPQ2 + 2 * i];
        dst[0] = log(mq);
        dst[1] = log((1.0 - mq));
    }
    // --faster model: ObservationModelS::setupReadLikelihoods / SStateHMM constants — reference Faster.cpp:117-124, :300-352
    out[TC_FAST + 0] = log(1.0 - p->pError);
    out[TC_FAST + 1] = log(p->pError);
    out[TC_FAST + 2] = log(1 - exp(-0.25));
    out[TC_FAST + 3] = log(1.0 - 1e-10);
    out[TC_FAST + 4] = log(1e-10);
    for (int i = 0; i < n_mapq; i++) {
        double mq = 1.0 - mapq_table[i];
        if (-10.0 * log10(mq) > p->capMapQualFast) mq = pow(10.0, -p->capMapQualFast / 10.0);
        out[T_MAPQF + 2 * i] = log(1.0 - mq);
        out[T_MAPQF + 2 * i + 1] = log(mq);
    }
    // homopolymer indel-error logs — reference ObservationModelFB.cpp:1683-1703
    for (int len = 0; len < DD_HP_TABLE; len++) {
        const double perr = hp_error(len < 1 ? 1 : len);
        out[T_HP + 2 * len] = log(perr);
        out[T_HP + 2 * len + 1] = log(1.0 - perr);
    }
    return T_END;
}

// ---- launch classes of a ragged batch ----
// A launch = (lane tiling of the haplotypes, read class).  Reads up to 160 bp and longer ones are separate launches, as in round 3; the
// reads up to 160 bp of a tiling are cut once more where its launch plan changes — BY WINDOW: windows whose longest read still lets the
// back-pointer tile sit in LDS at full occupancy (<= T, from make_plan: ~115 bp at K = 2) are one launch, the other windows (all their
// reads up to 160 bp, short ones included) another — so windows of short reads keep the faster LDS build when a batch also holds 150-bp
// windows (one "<= 160 bp" launch put the 100-bp reads on the scratch build, 18 % slower), and no window pays the haplotype set-up in both
// (cutting by READ made the trimmed-read windows straddle the cut: 3.70e11 against 3.98e11 cells/s without the cut on bench.py's ragged
// batch).  A haplotype is listed in a launch only if its window has reads for it.
static int lds_read_threshold(const dd_params *p, int max_hap_len, int n_qual)
{
    if (!p || check_params(p) != DD_SUCCESS) return 0;
    int lo = 0, hi = 160;                       // largest L in [1, 160] whose plan keeps the back-pointers in LDS (0: none)
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        Plan pl;
        ddk::KernelArgs A;
        memset(&A, 0, sizeof(A));
        if (make_plan(p, max_hap_len, mid, n_qual > 0 ? n_qual : 1, pl, A) == DD_SUCCESS && !pl.gbt) lo = mid; else hi = mid - 1;
    }
    return lo;
}

static int build_launch_classes(const dd_batch *b, const uint8_t *win_skip, const dd_params *p, int32_t *list, dd_length_classes *out)
{
    dd_sizes sz;
    int rc = dd_batch_sizes(b, &sz);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    const int W = b->n_windows;
    // pass 0: the folded builds (end states inside the generic candidate code, K <= 2 on the D = 6 build: +4.5 %) need every haplotype of their
    // launch to leave the last position idle — 64 K >= Hs + 3, i.e. up to 61 / 125 bp of the 62 / 126 the tiling holds.  One 126-bp haplotype
    // among thousands of shorter ones switched the fold off for all of them (the ragged leg's K = 2 launches).  When the haplotypes of exactly
    // the tiling's full length are few, they run apart (promote[c]): launches of their own, same tiling, not folded; the others fold.
    bool promote[DD_N_HAP_CLASSES] = {false};
    if (p && check_params(p) == DD_SUCCESS && pick_Dt(p->maxLengthDel + 1) == 6 && !getenv("DD_NO_FOLD") && !getenv("DD_NO_PROMOTE")) {
        int64_t n_fold[DD_N_HAP_CLASSES] = {0}, n_edge[DD_N_HAP_CLASSES] = {0};
        for (int w = 0; w < W; w++) {
            if (win_skip && win_skip[w]) continue;
            for (int64_t h = b->win_hap_off[w]; h < b->win_hap_off[w + 1]; h++) {
                const int len = b->hap_seq_off[h + 1] - b->hap_seq_off[h];
                const int c = hap_class_of(len < 1 ? 1 : (len > DD_MAX_HAP_LEN ? DD_MAX_HAP_LEN : len));
                if (len == kHapClasses[c].bound) n_edge[c]++; else n_fold[c]++;
            }
        }
        int used = 0, moved = 0;
        for (int c = 0; c < DD_N_HAP_CLASSES; c++) used += (n_fold[c] + n_edge[c]) > 0 ? 1 : 0;
        for (int c = 0; c < DD_N_HAP_CLASSES; c++) {
            int G0 = 1, K0 = 1;
            if (!pick_tiling(kHapClasses[c].bound, 6, G0, K0)) continue;
            if (G0 != 1 || K0 > 2 || n_edge[c] == 0 || n_fold[c] == 0) continue;   // only the tilings that have a folded build
            // the full-length haplotypes get launches of their own (same tiling, not folded: they cost what they cost before); worth it when the
            // launch they leave behind is the bigger part — a small launch fills the chip badly
            promote[c] = n_edge[c] * 4 < n_fold[c] && (used + moved + 1) * DD_N_READ_CLASSES <= DD_N_HAP_CLASSES * DD_N_READ_CLASSES;
            moved += promote[c] ? 1 : 0;
        }
    }
    // internal class ids: 0 .. N-1 the tilings, N + c the full-length haplotypes of tiling c when they run apart
    constexpr int NC = 2 * DD_N_HAP_CLASSES;
    auto cls = [&](int len) { const int c = hap_class_of(len); return (promote[c] && len == kHapClasses[c].bound) ? DD_N_HAP_CLASSES + c : c; };
    // pass 1: haplotype class maxima (the read thresholds depend on the class' longest haplotype)
    int hmax[NC] = {0};
    bool any_skipped = false;
    for (int w = 0; w < W; w++) {
        if (win_skip && win_skip[w]) { any_skipped = any_skipped || b->win_hap_off[w + 1] > b->win_hap_off[w]; continue; }
        for (int64_t h = b->win_hap_off[w]; h < b->win_hap_off[w + 1]; h++) {
            const int len = b->hap_seq_off[h + 1] - b->hap_seq_off[h];
            if (len > DD_MAX_HAP_LEN) return fail(DD_ERR_UNSUPPORTED, "haplotype longer than 766 in a window that is not flagged in win_skip");
            const int c = cls(len < 1 ? 1 : len);
            if (len > hmax[c]) hmax[c] = len;
        }
    }
    int bound[NC][DD_N_READ_CLASSES];                        // upper read length of each interval of each tiling
    const bool one_read_class = getenv("DD_LENGTH_CLASSES") && !strcmp(getenv("DD_LENGTH_CLASSES"), "k");   // A/B: haplotype classes only
    for (int c = 0; c < NC; c++) {
        int T = hmax[c] > 0 ? lds_read_threshold(p, hmax[c], b->n_qual) : 0;
        if (getenv("DD_READ_BOUND")) T = atoi(getenv("DD_READ_BOUND"));                                    // A/B only
        if (T < 1 || T >= 160) T = 0;
        bound[c][0] = one_read_class ? DD_MAX_READ_LEN : (T ? T : 160);
        bound[c][1] = one_read_class ? DD_MAX_READ_LEN : 160;
        bound[c][2] = DD_MAX_READ_LEN;
    }
    auto read_class = [&](int c, int len) { return len <= bound[c][0] ? 0 : (len <= bound[c][1] ? 1 : 2); };
    // pass 2: per window, which (tiling, interval) launches its haplotypes take part in
    struct Acc { std::vector<int32_t> haps; int max_hap = 0, max_read = 0, max_reads = 0; int64_t sum_reads = 0, n_win = 0, sum_len = 0; };
    std::vector<Acc> acc((size_t)NC * DD_N_READ_CLASSES);
    std::vector<int32_t> skipped;                             // haplotypes of skipped windows: marked by the first launch
    for (int w = 0; w < W; w++) {
        const int64_t h0 = b->win_hap_off[w], h1 = b->win_hap_off[w + 1], q0 = b->win_read_off[w], q1 = b->win_read_off[w + 1];
        if (win_skip && win_skip[w]) { for (int64_t h = h0; h < h1; h++) skipped.push_back((int32_t)h); continue; }
        if (h1 <= h0 || q1 <= q0) continue;
        unsigned seen = 0;                                    // tilings of this window already handled
        for (int64_t h = h0; h < h1; h++) {
            const int hl = b->hap_seq_off[h + 1] - b->hap_seq_off[h];
            const int c = cls(hl < 1 ? 1 : hl);
            if (seen & (1u << c)) continue;
            seen |= 1u << c;
            int cnt[DD_N_READ_CLASSES] = {0}, mx[DD_N_READ_CLASSES] = {0};
            int64_t sl[DD_N_READ_CLASSES] = {0};
            int mx160 = 0;                                        // the window's longest read up to 160 bp decides between classes 0 and 1
            for (int64_t q = q0; q < q1; q++) {
                const int len = b->read_seq_off[q + 1] - b->read_seq_off[q];
                if (len >= 1 && len <= bound[c][1] && len > mx160) mx160 = len;
            }
            const int k160 = read_class(c, mx160 > 0 ? mx160 : 1);
            for (int64_t q = q0; q < q1; q++) {
                const int len = b->read_seq_off[q + 1] - b->read_seq_off[q];
                if (len < 1) continue;
                const int k = len <= bound[c][1] ? k160 : 2;
                cnt[k]++;
                sl[k] += len;
                if (len > mx[k]) mx[k] = len;
            }
            for (int k = 0; k < DD_N_READ_CLASSES; k++) {
                if (!cnt[k]) continue;
                Acc &a = acc[(size_t)c * DD_N_READ_CLASSES + k];
                for (int64_t g = h; g < h1; g++) {
                    const int gl = b->hap_seq_off[g + 1] - b->hap_seq_off[g];
                    if (cls(gl < 1 ? 1 : gl) != c) continue;
                    a.haps.push_back((int32_t)g);
                    if (gl > a.max_hap) a.max_hap = gl;
                }
                if (mx[k] > a.max_read) a.max_read = mx[k];
                if (cnt[k] > a.max_reads) a.max_reads = cnt[k];
                a.sum_reads += cnt[k];
                a.sum_len += sl[k];
                a.n_win++;
            }
        }
    }
    int32_t off = 0;
    auto emit = [&](int c, int k, Acc &a, const std::vector<int32_t> *extra) {
        dd_launch_class &L = out->launch[out->n_launches++];
        L.list_off = off;
        if (extra && !extra->empty()) {                        // merge (both ascending) so that the list stays sorted
            std::vector<int32_t> m(a.haps.size() + extra->size());
            std::merge(a.haps.begin(), a.haps.end(), extra->begin(), extra->end(), m.begin());
            a.haps.swap(m);
        }
        L.list_len = (int32_t)a.haps.size();
        if (list) memcpy(list + off, a.haps.data(), a.haps.size() * sizeof(int32_t));
        off += L.list_len;
        L.hap_class = c % DD_N_HAP_CLASSES;
        L.max_hap_len = a.max_hap > 0 ? a.max_hap : 1;
        L.min_read_len = k == 2 ? bound[c][1] + 1 : 1;       // (class 1 = the windows with a read beyond T: all their reads up to 160 bp)
        L.max_read_len = a.max_read > 0 ? a.max_read : 1;
        L.max_window_reads = a.max_reads;
        L.avg_window_reads = a.n_win ? (int32_t)((a.sum_reads + a.n_win - 1) / a.n_win) : 0;
        L.avg_read_len = a.sum_reads ? (int32_t)(a.sum_len / a.sum_reads) : 0;
    };
    bool first = true;
    for (int c = 0; c < NC; c++)
        for (int k = 0; k < DD_N_READ_CLASSES; k++) {
            Acc &a = acc[(size_t)c * DD_N_READ_CLASSES + k];
            if (a.haps.empty()) continue;
            if (out->n_launches >= DD_N_HAP_CLASSES * DD_N_READ_CLASSES) return fail(DD_ERR_UNSUPPORTED, "more launch classes than dd_length_classes holds");
            emit(c, k, a, first ? &skipped : nullptr);
            first = false;
        }
    if (first && !skipped.empty()) {                           // nothing but skipped windows: one launch that only marks their pairs
        Acc a;
        emit(0, 0, a, &skipped);
    }
    out->list_len = off;
    return DD_SUCCESS;
}

int dd_build_length_classes(const dd_batch *b, const uint8_t *win_skip, const dd_params *p, int32_t *hap_class_list, dd_length_classes *out)
{
    if (!b || !hap_class_list || !out) return fail(DD_ERR_INVALID, "null argument");
    return build_launch_classes(b, win_skip, p, hap_class_list, out);
}

size_t dd_workspace_bytes(const dd_params *p, const dd_device_batch *b)
{
    if (!p || !b || check_params(p) != DD_SUCCESS) return 0;
    if (b->max_hap_len < 1 || b->max_read_len < 1) return 0;
    Plan pl;
    ddk::KernelArgs A;
    memset(&A, 0, sizeof(A));
    if (make_plan(p, b->max_hap_len, b->max_read_len, b->n_qual, pl, A) != DD_SUCCESS) return 0;
    size_t bytes = pl.scratch_bytes;
    if (b->classes && b->hap_class_list)          // per-class launches: the largest scratch any of them needs
        for (int i = 0; i < b->classes->n_launches; i++) {
            ddk::KernelArgs A2;
            memset(&A2, 0, sizeof(A2));
            if (make_plan(p, b->classes->launch[i].max_hap_len, b->classes->launch[i].max_read_len, b->n_qual, pl, A2) == DD_SUCCESS &&
                pl.scratch_bytes > bytes)
                bytes = pl.scratch_bytes;
        }
    return bytes + DD_WS_HEADER;                  // the work counter of the ragged launches in front of the back-pointer tiles
}

// Enqueue the path for haplotypes [hap_begin, hap_end) and reads [read_begin, read_end) of the batch (a
// contiguous block of windows); hap_end < 0 means the whole batch.
// One (haplotype-length class, read-length class) of a ragged batch: the launch plan (K, LDS tile) is made for
// the class' own maxima instead of the batch-wide ones.
struct LenClass {
    const int32_t *hap_list = nullptr;   // device: haplotype indices of the class (sorted); nullptr = all haplotypes
    int list_begin = 0, list_end = 0;    // range of hap_list this launch covers
    int max_hap_len = 0, max_read_len = 0, min_read_len = 1;
    int max_window_reads = 0, avg_window_reads = 0;   // reads of the class per window of the list (0 = not known)
    int avg_read_len = 0;                             // mean length of the class' reads (0 = not known)
    bool run_onhap = true;
};

// which per-pair model an entry point runs
enum Model { MODEL_FBMAXERR = 0 /* ObservationModelFBMaxErr, computeLikelihoods */, MODEL_S = 1 /* ObservationModelS, computeLikelihoodsFaster */ };

// Waves per workgroup for windows with `units` units of work per haplotype (reads for the main kernel, groups of pairs for
// the --faster one): a workgroup's waves take the units round-robin, so with few units some waves idle in the last round
// while the workgroup's LDS stays allocated.  Keep `maxw` unless a smaller workgroup uses its waves > 10 % better
// (tools/coverage_sweep.py: 2 reads per window ran at half the rate with 4-wave workgroups).
// per_cu: waves the build keeps on a CU — workgroups of w waves leave per_cu % w of them unused (three-wave workgroups of a build
// held to 8 waves: 6 resident; the --faster kernel ran 10-read windows at 3.7e11 instead of 4.6e11 that way)
static int waves_for_reads(int64_t units, int maxw, int per_cu)
{
    if (units < 1) units = 1;
    if (per_cu < maxw) per_cu = maxw;
    auto use = [&](int w) {
        return (double)units / (double)(((units + w - 1) / w) * w) * (double)((per_cu / w) * w) / (double)per_cu;
    };
    int best = maxw;
    double bestu = use(maxw);
    for (int w = maxw - 1; w >= 1; w--) {
        const double u = use(w);
        if (u > bestu * 1.10) { best = w; bestu = u; }
    }
    return best;
}

// Read split of a haplotype over workgroups.  One workgroup per haplotype is the cheapest (the per-haplotype setup is done
// once), but a small batch has too few haplotypes to fill the chip, a split that leaves the workgroup's `waves` waves a
// ragged number of rounds wastes wave slots (tools/batch_size_sweep.py), and a grid that is only a few times what the chip
// holds at once (`resident` workgroups) ends with a partly filled last round.  Among the splits from the one that yields
// `min_blocks` workgroups up to 8 x that, take the one with the best product of wave-slot use, round fill and setup
// amortisation (the per-haplotype tables cost about a quarter of one read's work per wave); within 1 % the smaller split
// wins.  Measured against the rule without the round term (DD_SPLIT_NO_ROUNDS=1, profiles/r03/split_ab.txt): 128 windows
// +4.6 %, 512 windows +1.8 %, the other sizes within 0.5 % — workgroups do not finish in lock step, so the rounds matter
// less than the count suggests.  resident == 0: the rounds are not modelled (the --faster kernel's callers).
static int64_t pick_split(int64_t n_haps, int64_t units, int waves, int64_t min_blocks, int64_t resident = 0)
{
    if (units < 1) units = 1;
    if (n_haps < 1) n_haps = 1;
    const int64_t max_split = (units + waves - 1) / waves;
    int64_t need = (min_blocks + n_haps - 1) / n_haps;
    if (need < 1) need = 1;
    if (need > max_split) need = max_split;
    if (resident <= 0) {
        int64_t best = need;
        double bestu = -1.0;
        for (int64_t sp = need; sp <= max_split; sp++) {
            const int64_t slots = sp * waves;
            const double u = (double)units / (double)(slots * ((units + slots - 1) / slots));
            if (u >= 0.9) return sp;
            if (u > bestu) { bestu = u; best = sp; }
        }
        return best;
    }
    int64_t best = need;
    double beste = -1.0;
    const int64_t last = std::min<int64_t>(max_split, need * 8);
    for (int64_t sp = need; sp <= last; sp++) {
        const int64_t slots = sp * waves;
        const double per_wave = (double)((units + slots - 1) / slots);          // reads of the busiest wave: the workgroup's duration
        const double use = (double)units / ((double)slots * per_wave);
        const int64_t n = n_haps * sp;
        const double fill = (double)n / (double)(((n + resident - 1) / resident) * resident);
        const double e = use * fill * per_wave / (per_wave + 0.25);
        if (e > beste * 1.01) { beste = e; best = sp; }
    }
    return best;
}

// --faster model LDS: block-shared haplotype index + per-pair areas (layout in faster_kernel.hip's header)
static size_t lds_layout_fast(int max_hap_len, int max_read_len, int n_qual, int &waves, int &groups, ddk::KernelArgs &A)
{
    uint32_t off = 0;
    A.lds_off_E = off; off = up16(off + 16u * (uint32_t)(n_qual > 0 ? n_qual : 1));
    A.lds_off_N = off; off = up16(off + (uint32_t)max_hap_len);
    A.lds_off_Q = off; off = up16(off + 2u * 257u);
    A.lds_off_C = off; off = up16(off + 2u * (uint32_t)max_hap_len);
    A.lds_off_Y = off; off = up16(off + 4u * 256u);
    A.lds_off_rdE = off; off = up16(off + 6u * 256u);          // FAST_CHUNK sort keys (u32) + ranks (u16)
    A.lds_shared_bytes = off;
    uint32_t po = 0;
    A.lds_off_A = po;   po = up16(po + 4u * (uint32_t)((max_hap_len + max_read_len + 1) / 2 + 1));
    A.lds_off_rdC = po; po = up16(po + 2u * (uint32_t)max_read_len);
    A.lds_off_bt = po;  po = up16(po + 16u * (uint32_t)max_read_len);
    A.lds_off_ms = A.lds_off_A;                  // the state path reuses the histogram bytes
    A.lds_off_I = po;   po = up16(po + 256u);
    A.lds_off_rdQ = po; po = up16(po + 64u);
    A.lds_wave_bytes = po;                       // bytes per PAIR area
    const size_t cap = (size_t)160 * 1024;
    // pair areas per wavefront: one per concurrent pair, plus a dummy for the idle 16-lane groups when fewer than 4 fit
    auto areas = [](int gq) { return gq < 4 ? gq + 1 : 4; };
    groups = 4;
    if (const char *e = getenv("DD_FAST_GROUPS")) {             // tests: exercise the fewer-pairs-per-wavefront geometry
        const int gq = atoi(e);
        if (gq == 1 || gq == 2) groups = gq;
    }
    while (groups > 1 && (size_t)off + (size_t)areas(groups) * po > cap) groups >>= 1;
    // waves per workgroup: whatever keeps the most wavefronts resident per CU (ties: more waves share one haplotype index)
    int best_w = 1, best_res = 0;
    for (int wv = DD_WAVES; wv >= 1; wv--) {
        const size_t bytes = (size_t)off + (size_t)wv * areas(groups) * po;
        if (bytes > cap) continue;
        int blocks = (int)(cap / bytes);
        int res = blocks * wv;
        if (res > 8) res = 8;                    // the kernel is built for 2 waves per SIMD
        if (res > best_res) { best_res = res; best_w = wv; }
    }
    waves = best_w;
    if (const char *e = getenv("DD_FAST_WAVES")) {              // A/B only
        const int wv = atoi(e);
        if (wv >= 1 && wv <= DD_WAVES && (size_t)off + (size_t)wv * areas(groups) * po <= cap) waves = wv;
    }
    return (size_t)off + (size_t)waves * areas(groups) * po;
}

static int launch_fast(const dd_params *p, const dd_device_batch *b, ddk::KernelArgs &A, void *stream, int hap_begin, int hap_end,
                       int read_begin, int read_end, bool want_onhap)
{
    (void)p;
    int waves = DD_WAVES, groups = 4;
    size_t lds = lds_layout_fast(b->max_hap_len, b->max_read_len, b->n_qual, waves, groups, A);
    if (lds > (size_t)160 * 1024) return fail(DD_ERR_UNSUPPORTED, "shape exceeds the LDS tile of the --faster kernel");
    A.n_qual = b->n_qual;
    A.fast_groups = groups;
    int64_t target_blocks = 1024;                 // workgroups wanted before haplotypes are split (A/B: DD_FAST_TARGET_BLOCKS)
    if (const char *e = getenv("DD_FAST_TARGET_BLOCKS")) { const long v = atol(e); if (v >= 1) target_blocks = v; }   // A/B only
    int64_t avg_reads = (b->n_reads + b->n_windows - 1) / (b->n_windows > 0 ? b->n_windows : 1);
    {   // thin windows: no more wavefronts per workgroup than the windows have groups of `groups` reads
        const int w2 = waves_for_reads((avg_reads + groups - 1) / groups, waves, 8);
        if (w2 != waves) {
            waves = w2;
            lds = (size_t)A.lds_shared_bytes + (size_t)waves * (groups < 4 ? groups + 1 : 4) * A.lds_wave_bytes;
        }
    }
    const int64_t split = pick_split(hap_end - hap_begin, (avg_reads + groups - 1) / groups, waves, target_blocks);
    if ((int64_t)b->n_haps * split > 0x7fffffffLL) return fail(DD_ERR_UNSUPPORTED, "batch too large for one launch");
    A.n_split = (int32_t)split;
    A.item_begin = (int32_t)(hap_begin * split);
    A.n_items = (int32_t)(hap_end * split);
    A.read_begin = read_begin; A.read_end = read_end;
    A.hap_list = nullptr; A.len_min = 0; A.len_max = 0x7fffffff;
    int64_t grid = (int64_t)(hap_end - hap_begin) * split;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grid > 0) {
        g_last_launch[0] = groups; g_last_launch[1] = 0; g_last_launch[2] = waves; g_last_launch[3] = (int32_t)lds;
        g_last_launch[4] = (int32_t)grid; g_last_launch[5] = (int32_t)split; g_last_launch[6] = (int32_t)A.lds_wave_bytes; g_last_launch[7] = 0;
        HIP_TRY(ddk::launch_faster(A, (unsigned)grid, waves, lds, st));
    }
    if (want_onhap) HIP_TRY(ddk::launch_onhap(A, st));
    return DD_SUCCESS;
}

static thread_local bool tl_overlapping_chunks = false;   // set by the host-pointer path while it alternates chunk launches between two streams
static int launch_range(Model model, const dd_params *p, const dd_device_batch *b, const dd_result *r, void *workspace, size_t workspace_bytes,
                        void *stream, int hap_begin, int hap_end, int read_begin, int read_end, const LenClass *lc = nullptr)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (!b || !r || !r->ll || !r->status) return fail(DD_ERR_INVALID, "ll and status outputs are required");
    if (b->n_haps <= 0 || b->n_reads <= 0) return DD_SUCCESS;
    if (b->max_hap_len < 1 || b->max_hap_len > DD_MAX_HAP_LEN) return fail(DD_ERR_UNSUPPORTED, "haplotype length outside [1,766]");
    if (b->max_read_len < 1 || b->max_read_len > DD_MAX_READ_LEN) return fail(DD_ERR_UNSUPPORTED, "read length outside [1,1024]");
    const int D = p->maxLengthDel + 1;
    ddk::KernelArgs A;
    memset(&A, 0, sizeof(A));
    A.n_windows = b->n_windows; A.n_haps = b->n_haps; A.n_reads = b->n_reads;
    A.win_hap_off = b->win_hap_off; A.win_read_off = b->win_read_off; A.win_hap_start = b->win_hap_start;
    A.hap_seq_off = b->hap_seq_off; A.hap_seq = b->hap_seq; A.hap_var_off = b->hap_var_off; A.hap_var = b->hap_var; A.hap_var_flank = b->hap_var_flank;
    A.read_seq_off = b->read_seq_off; A.read_seq = b->read_seq; A.read_qidx = b->read_qidx; A.read_mqidx = b->read_mqidx;
    A.read_start = b->read_start; A.read_flags = b->read_flags;
    A.hap_window = b->hap_window; A.win_pair_off = b->win_pair_off; A.win_hpos_off = b->win_hpos_off;
    A.win_varcov_off = b->win_varcov_off; A.tables = b->tables; A.sym_lut = b->sym_lut; A.win_skip = b->win_skip;
    if (p->mapUnmappedReads && model == MODEL_FBMAXERR) {   // the --faster model has no insert-size prior
        if (!b->read_mate_pos || !b->read_mate_len || !b->read_lib || !b->lib_off || !b->lib_logprob || !b->lib_log95)
            return fail(DD_ERR_INVALID, "mapUnmappedReads needs the mate arrays and the library log tables");
        A.read_mate_pos = b->read_mate_pos; A.read_mate_len = b->read_mate_len; A.read_lib = b->read_lib;
        A.lib_off = b->lib_off; A.lib_logprob = b->lib_logprob; A.lib_log95 = b->lib_log95;
    }
    A.out = *r;
#ifdef DD_STAMPS
    A.dbg = g_dbg;
#endif
    A.always_ro = getenv("DD_ALWAYS_RO") ? 1 : 0;
    A.D = D; A.maxLengthDel = p->maxLengthDel; A.padCover = p->padCover; A.bMid = p->bMid; A.maxMismatch = p->maxMismatch;
    if (model == MODEL_S) {
        if (hap_end < 0) { hap_begin = 0; hap_end = b->n_haps; read_begin = 0; read_end = b->n_reads; }
        return launch_fast(p, b, A, stream, hap_begin, hap_end, read_begin, read_end, r->onHap && r->offHapHMQ);
    }
    Plan pl;
    const int cls_hap = lc ? lc->max_hap_len : b->max_hap_len, cls_read = lc ? lc->max_read_len : b->max_read_len;
    rc = make_plan(p, cls_hap, cls_read, b->n_qual, pl, A);
    if (rc) return rc;
    A.hap_list = lc ? lc->hap_list : nullptr;
    A.len_min = lc ? lc->min_read_len : 0;
    A.len_max = lc ? lc->max_read_len : 0x7fffffff;
    const int K = pl.K, Dt = pl.Dt;
    int waves = pl.waves;
    size_t lds = pl.lds;
    // thin windows: a wavefront works on one read at a time, so a workgroup never needs more waves than the windows have
    // reads (tools/coverage_sweep.py: 2 reads per window ran at half the rate with idle waves in every workgroup)
    const int64_t avg_reads_w = (lc && lc->avg_window_reads > 0) ? lc->avg_window_reads : (b->n_reads + b->n_windows - 1) / (b->n_windows > 0 ? b->n_windows : 1);
    {
        const int w2 = waves_for_reads((avg_reads_w + pl.G - 1) / pl.G, waves, pl.waves_per_cu);   // a wavefront takes G reads at a time
        if (w2 != waves) {
            waves = w2;
            lds = lds_layout(K, Dt, cls_read, b->n_qual, waves, pl.gbt, pl.G, A);
        }
    }
    if (pl.gbt) {
        if (!workspace || workspace_bytes < pl.scratch_bytes + DD_WS_HEADER)
            return fail(DD_ERR_INVALID, "workspace too small for this shape: allocate dd_workspace_bytes() bytes");
        A.bt_scratch = static_cast<unsigned char *>(workspace) + DD_WS_HEADER;
        A.bt_rows = cls_read;
        A.bt_wave_bytes = (uint32_t)scratch_wave_bytes(K, Dt, pl.G, cls_read);
    }
    // enough workgroups to fill 256 CUs several times over, but keep >= 1 read per wave
    int64_t target_blocks = 4096;
    if (const char *e = getenv("DD_TARGET_BLOCKS")) { const long v = atol(e); if (v >= 1) target_blocks = v; }   // A/B only
    int64_t avg_reads = avg_reads_w;
    if (hap_end < 0) { hap_begin = 0; hap_end = b->n_haps; read_begin = 0; read_end = b->n_reads; }
    if (lc && lc->hap_list) { hap_begin = lc->list_begin; hap_end = lc->list_end; }   // positions in the class list
    // the split is chosen for the haplotypes THIS launch covers: a rare length class or a small window block must still
    // spread over the chip
    // workgroups the chip holds at once (one-shot grids; a persistent GBT grid is capped to that number below anyway)
    int64_t resident = 256 * (int64_t)std::max(1, std::min((int)((160u * 1024u) / (lds ? lds : 1)), pl.waves_per_cu / waves));
    if (getenv("DD_SPLIT_NO_ROUNDS")) resident = 0;                            // A/B only: the rule before round 3
    int64_t split = pick_split(hap_end - hap_begin, (avg_reads + pl.G - 1) / pl.G, waves, target_blocks, resident);
    // Ragged windows: `split` suits a window with the average number of reads; a haplotype whose window has more gets proportionally more
    // workgroups (the kernel derives its own count from reads_per_wave), so that a 400-read window among 100-read ones does not end the
    // launch with one workgroup still at work; when the spread is wide the work of a wavefront is also capped (DD_READS_PER_WAVE, A/B).
    A.reads_per_wave = 0;
    const int max_reads = (lc && lc->max_window_reads > 0) ? lc->max_window_reads : b->max_window_reads;
    if (max_reads > 0 && max_reads * 4 > avg_reads * 5 && !getenv("DD_UNIFORM_SPLIT")) {
        const int64_t units = (avg_reads + pl.G - 1) / pl.G, max_units = (max_reads + pl.G - 1) / pl.G;
        int64_t rpw = (units + waves * split - 1) / (waves * split);
        int cap_rpw = (Dt > 7 || K >= 3) ? 24 : 12;
        if (const char *e = getenv("DD_READS_PER_WAVE")) { const int v = atoi(e); if (v >= 1) cap_rpw = v; }
        if (rpw > cap_rpw) rpw = cap_rpw;
        if (rpw < 1) rpw = 1;
        A.reads_per_wave = (int32_t)rpw;
        split = (max_units + waves * rpw - 1) / (waves * rpw);
        if (split < 1) split = 1;
    }
    A.n_split = (int32_t)split;
    if ((int64_t)b->n_haps * split > 0x7fffffffLL) return fail(DD_ERR_UNSUPPORTED, "batch too large for one launch");
    A.item_begin = (int32_t)(hap_begin * split);
    A.n_items = (int32_t)(hap_end * split);
    A.read_begin = read_begin; A.read_end = read_end;
    int64_t grid = (int64_t)(hap_end - hap_begin) * split;
    if (grid <= 0) {
        if (r->onHap && r->offHapHMQ && lc && lc->run_onhap) HIP_TRY(ddk::launch_onhap(A, static_cast<hipStream_t>(stream)));
        return DD_SUCCESS;
    }
    const int64_t n_launch_items = grid;
    if (pl.grid_cap) {
        // the scratch holds grid_cap x pl.waves back-pointer tiles; smaller workgroups (thin windows) may be more numerous
        const int64_t cap = (int64_t)pl.grid_cap * pl.waves / waves;
        if (grid > cap) grid = cap;
    }
    // More items than the chip holds workgroups: a persistent grid that draws its items from a counter.  Built for the ragged launches (items that
    // differ 20-fold in work); at the end of round 4 it turned out to be worth as much on UNIFORM batches wherever the grid was persistent
    // already — every HBM-scratch build ran a fixed stride, and the items of a uniform batch still differ by their reads' bMid: 3.42 -> 3.85e11 cells/s
    // at 130 bp, 3.34 -> 3.88e11 at 80 bp, +10-18 % on every K >= 3 tiling (profiles/r04/item_counter_uniform_ab.txt) — and +0.1-1.5 % on the
    // one-shot LDS grids (the headline build: +0.9 %), whose XCD-contiguous numbering (one L2 per window's haplotypes) it gives up.
    hipStream_t st = static_cast<hipStream_t>(stream);
    A.work_counter = nullptr;
    {
        // (Except: the chunks of the host-pointer path alternate between two streams, and a one-shot grid lets the next chunk's workgroups move in
        // while this one's drain; a chip-filling persistent grid holds its slots to the end — `dd_compute_likelihoods` with pageable pointers lost 7 %
        // that way, 23.7 -> 22.0 k windows/s at configs[1].  There the LDS builds keep their one-shot grids unless the batch is ragged.)
        const bool spread = A.reads_per_wave > 0 || (lc && lc->avg_read_len > 0 && lc->max_read_len * 4 > lc->avg_read_len * 5);
        const char *e = getenv("DD_DYNAMIC");                                 // A/B: 0 = never (one-shot LDS grids, fixed stride on scratch builds)
        const bool dynamic = e ? (e[0] == '1') : (pl.gbt || spread || !tl_overlapping_chunks);
        if (dynamic && workspace && workspace_bytes >= DD_WS_HEADER && resident > 0 && n_launch_items > resident) {
            A.work_counter = static_cast<int32_t *>(workspace);
            HIP_TRY(hipMemsetAsync(workspace, 0, 4, st));
            if (grid > resident) grid = resident;
        }
    }
    g_last_launch[0] = K; g_last_launch[1] = Dt + (pl.gbt ? 100 : 0); g_last_launch[2] = waves; g_last_launch[3] = (int32_t)lds;
    g_last_launch[4] = (int32_t)grid; g_last_launch[5] = (int32_t)split; g_last_launch[6] = (int32_t)A.lds_wave_bytes;
    g_last_launch[7] = (int32_t)A.lds_shared_bytes;
    // the build with the end states folded into the generic candidate code (hmm_kernel.hip, FOLD): K <= 2 at D build 6 with LDS
    // back-pointers, K = 2 at D build 6 with scratch back-pointers, K = 2 at D build 11 with LDS back-pointers — and every haplotype
    // of this launch leaves position 64 K - 1 idle (numS <= 64 K - 1)
    const bool fold_build = pl.gbt ? (K == 2 && Dt == 6) : ((K <= 2 && Dt == 6) || (K == 2 && Dt == 11));     // the builds measured to gain from it
    const bool fold = pl.G == 1 && fold_build && 64 * K >= cls_hap + 3 && !getenv("DD_NO_FOLD");
    g_last_fold = fold ? 1 : 0;
    g_last_occ = (pl.gbt && pl.two_waves) ? 2 : 0;
    const int build = (fold ? DD_BUILD_FOLD : 0) | ((pl.gbt && pl.two_waves) ? DD_BUILD_TWO_WAVES : 0) | (pl.G == 2 ? DD_BUILD_HALF : 0);
    g_last_G = pl.G;
    LaunchRec rec;
    {
        const int32_t v[DD_LAUNCH_LOG_FIELDS] = {K, pl.G, Dt, pl.gbt ? 1 : 0, fold ? 1 : 0, waves, (int32_t)lds, (int32_t)grid, (int32_t)split,
                                                 hap_end - hap_begin, cls_hap, lc ? lc->min_read_len : 1, cls_read, pl.waves_per_cu, g_last_occ, -1,
                                                 A.work_counter ? 1 : 0, A.reads_per_wave};
        memcpy(rec.v, v, sizeof(v));
    }
    static const bool timing = getenv("DD_LAUNCH_TIMING") != nullptr;
    if (timing) { HIP_TRY(hipEventCreate(&rec.e0)); HIP_TRY(hipEventCreate(&rec.e1)); HIP_TRY(hipEventRecord(rec.e0, st)); }
    HIP_TRY(ddk::launch_hmm(K, Dt, pl.gbt, build, A, (unsigned)grid, waves, lds, st));
    if (timing) HIP_TRY(hipEventRecord(rec.e1, st));
    g_launch_log.push_back(rec);
    if (r->onHap && r->offHapHMQ && (!lc || lc->run_onhap)) HIP_TRY(ddk::launch_onhap(A, st));
    return DD_SUCCESS;
}

int dd_plan_info(const dd_params *p, int max_hap_len, int max_read_len, int n_qual, int avg_reads, int n_haps, int32_t out[10])
{
    int rc = check_params(p);
    if (rc) return rc;
    if (!out) return fail(DD_ERR_INVALID, "null argument");
    if (max_hap_len < 1 || max_hap_len > DD_MAX_HAP_LEN) return fail(DD_ERR_UNSUPPORTED, "haplotype length outside [1,766]");
    if (max_read_len < 1 || max_read_len > DD_MAX_READ_LEN) return fail(DD_ERR_UNSUPPORTED, "read length outside [1,1024]");
    Plan pl;
    ddk::KernelArgs A;
    memset(&A, 0, sizeof(A));
    if ((rc = make_plan(p, max_hap_len, max_read_len, n_qual, pl, A))) return rc;
    const int waves = waves_for_reads((avg_reads + pl.G - 1) / pl.G, pl.waves, pl.waves_per_cu);
    out[8] = pl.G; out[9] = 0;
    avg_reads = (avg_reads + pl.G - 1) / pl.G;   // units of work per haplotype: a wavefront takes G reads at a time
    out[0] = pl.K; out[1] = pl.Dt; out[2] = pl.gbt ? 1 : 0; out[3] = waves;
    out[5] = (int32_t)lds_layout(pl.K, pl.Dt, max_read_len, n_qual, waves, pl.gbt, pl.G, A);
    out[4] = (int32_t)pick_split(n_haps, avg_reads, waves, 4096,
                                 256 * (int64_t)std::max(1, std::min((int)((160u * 1024u) / (out[5] > 0 ? (unsigned)out[5] : 1u)), pl.waves_per_cu / waves)));
    out[6] = (int32_t)((pl.scratch_bytes >> 10) & 0x7fffffff);
    out[7] = pl.waves_per_cu;
    return DD_SUCCESS;
}

int dd_launch_device(const dd_params *p, const dd_device_batch *b, const dd_result *r, void *workspace, size_t workspace_bytes, void *stream)
{
    launch_log_clear();
    if (b && b->classes && b->hap_class_list && b->classes->n_launches > 1) {
        // ragged batch: one launch per (lane tiling, read-length interval) that has work, onHap once at the end
        const dd_length_classes *C = b->classes;
        for (int i = 0; i < C->n_launches; i++) {
            const dd_launch_class &L = C->launch[i];
            LenClass lc;
            lc.hap_list = b->hap_class_list + L.list_off;
            lc.list_begin = 0;
            lc.list_end = L.list_len;
            lc.max_hap_len = L.max_hap_len;
            lc.min_read_len = L.min_read_len;
            lc.max_read_len = L.max_read_len;
            lc.max_window_reads = L.max_window_reads;
            lc.avg_window_reads = L.avg_window_reads;
            lc.avg_read_len = L.avg_read_len;
            lc.run_onhap = (i == C->n_launches - 1);
            const int rc = launch_range(MODEL_FBMAXERR, p, b, r, workspace, workspace_bytes, stream, 0, b->n_haps, 0, b->n_reads, &lc);
            if (rc) return rc;
        }
        return DD_SUCCESS;
    }
    return launch_range(MODEL_FBMAXERR, p, b, r, workspace, workspace_bytes, stream, 0, -1, 0, 0);
}

int dd_launch_device_faster(const dd_params *p, const dd_device_batch *b, const dd_result *r, void *stream)
{
    return launch_range(MODEL_S, p, b, r, nullptr, 0, stream, 0, -1, 0, 0);
}

int dd_pair_sum_offsets(const dd_batch *b, int64_t *win_hh_off)
{
    if (!b || !win_hh_off) return fail(DD_ERR_INVALID, "null argument");
    int64_t o = 0;
    for (int w = 0; w < b->n_windows; w++) {
        win_hh_off[w] = o;
        const int64_t H = b->win_hap_off[w + 1] - b->win_hap_off[w];
        o += H * H;
    }
    win_hh_off[b->n_windows] = o;
    return DD_SUCCESS;
}

int dd_pair_sums_device(const dd_device_batch *b, const int64_t *win_hh_off_dev, int64_t n_slots,
                        const double *ll_dev, double *out_dev, void *stream)
{
    if (!b || !win_hh_off_dev || !ll_dev || !out_dev) return fail(DD_ERR_INVALID, "null argument");
    ddk::PairSumArgs A;
    A.n_windows = b->n_windows; A.n_slots = n_slots;
    A.win_hap_off = b->win_hap_off; A.win_read_off = b->win_read_off; A.win_pair_off = b->win_pair_off;
    A.win_hh_off = win_hh_off_dev; A.ll = ll_dev; A.out = out_dev;
    if ((n_slots + 3) / 4 > 0x7fffffffLL) return fail(DD_ERR_UNSUPPORTED, "batch too large for one launch");
    HIP_TRY(ddk::launch_pair_sums(A, static_cast<hipStream_t>(stream)));
    return DD_SUCCESS;
}

int dd_map_pairs_device(const dd_device_batch *b, const int64_t *win_hh_off_dev, const double *pair_sum_dev, const double *prior_dev,
                        const uint8_t *filtered_dev, const int32_t *ncand_dev, double *posterior_dev, int32_t *pairs_dev,
                        double *vals_dev, void *stream)
{
    if (!b || !win_hh_off_dev || !pair_sum_dev || !prior_dev || !filtered_dev || !ncand_dev || !pairs_dev || !vals_dev)
        return fail(DD_ERR_INVALID, "null argument");
    ddk::MapPairArgs A;
    A.n_windows = b->n_windows; A.win_hap_off = b->win_hap_off; A.win_hh_off = win_hh_off_dev;
    A.pair_sum = pair_sum_dev; A.prior = prior_dev; A.filtered = filtered_dev; A.ncand = ncand_dev;
    A.posterior = posterior_dev; A.pairs = pairs_dev; A.vals = vals_dev;
    HIP_TRY(ddk::launch_map_pairs(A, static_cast<hipStream_t>(stream)));
    return DD_SUCCESS;
}

int dd_map_pairs(const dd_batch *b, const double *ll_host, const double *prior_host, const uint8_t *filtered_host,
                 const int32_t *ncand_host, double *pair_sum_out, double *posterior_out, int32_t *pairs_out, double *vals_out, int device)
{
    dd_sizes sz;
    int rc = dd_batch_sizes(b, &sz);
    if (rc) return rc;
    if (!ll_host || !prior_host || !filtered_host || !ncand_host || !pairs_out || !vals_out) return fail(DD_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DD_ERR_NO_DEVICE, "no HIP device: the genotype reduction has no CPU fallback in this library");
    if (device < 0 || device >= ndev) return fail(DD_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    const int W = b->n_windows;
    if (W <= 0) return DD_SUCCESS;
    std::vector<int64_t> pair_off(W + 1), hh_off(W + 1);
    dd_batch_offsets(b, pair_off.data(), nullptr, nullptr);
    dd_pair_sum_offsets(b, hh_off.data());
    const size_t ns = (size_t)hh_off[W];
    DeviceCtx &ctx = g_ctx.c;
    if ((rc = ctx.reserve(device, (size_t)(2 * (W + 1) * 4 + 2 * (W + 1) * 8 + ((size_t)sz.n_pairs + 3 * ns) * 8 + (size_t)sz.n_haps * 5 + (size_t)W * 40 + 32 * 256), 0))) return rc;
    DevBuf dev(ctx);
    dd_device_batch db;
    memset(&db, 0, sizeof(db));
    db.n_windows = W;
    if ((rc = dev.upload(&db.win_hap_off, b->win_hap_off, (size_t)W + 1))) return rc;
    if ((rc = dev.upload(&db.win_read_off, b->win_read_off, (size_t)W + 1))) return rc;
    if ((rc = dev.upload(&db.win_pair_off, (const int64_t *)pair_off.data(), pair_off.size()))) return rc;
    const int64_t *hh_dev = nullptr;
    const double *ll_dev = nullptr, *prior_dev = nullptr;
    const uint8_t *filt_dev = nullptr;
    const int32_t *nc_dev = nullptr;
    double *sum_dev = nullptr, *post_dev = nullptr, *vals_dev = nullptr;
    int32_t *pairs_dev = nullptr;
    if ((rc = dev.upload(&hh_dev, (const int64_t *)hh_off.data(), hh_off.size()))) return rc;
    if ((rc = dev.upload(&ll_dev, ll_host, (size_t)sz.n_pairs))) return rc;
    if ((rc = dev.upload(&prior_dev, prior_host, ns))) return rc;
    if ((rc = dev.upload(&filt_dev, filtered_host, (size_t)sz.n_haps))) return rc;
    if ((rc = dev.upload(&nc_dev, ncand_host, (size_t)sz.n_haps))) return rc;
    if ((rc = dev.alloc(&sum_dev, ns))) return rc;
    if ((rc = dev.alloc(&post_dev, ns))) return rc;
    if ((rc = dev.alloc(&vals_dev, (size_t)3 * W))) return rc;
    if ((rc = dev.alloc(&pairs_dev, (size_t)4 * W))) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    if ((rc = dd_pair_sums_device(&db, hh_dev, hh_off[W], ll_dev, sum_dev, nullptr))) return rc;
    if ((rc = dd_map_pairs_device(&db, hh_dev, sum_dev, prior_dev, filt_dev, nc_dev, post_dev, pairs_dev, vals_dev, nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (pair_sum_out) HIP_TRY(hipMemcpy(pair_sum_out, sum_dev, ns * sizeof(double), hipMemcpyDeviceToHost));
    if (posterior_out) HIP_TRY(hipMemcpy(posterior_out, post_dev, ns * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pairs_out, pairs_dev, (size_t)4 * W * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(vals_out, vals_dev, (size_t)3 * W * sizeof(double), hipMemcpyDeviceToHost));
    return DD_SUCCESS;
}

int dd_pair_sums(const dd_batch *b, const double *ll_host, double *out_host, int device)
{
    dd_sizes sz;
    int rc = dd_batch_sizes(b, &sz);
    if (rc) return rc;
    if (!ll_host || !out_host) return fail(DD_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DD_ERR_NO_DEVICE, "no HIP device: the genotype read-sum has no CPU fallback in this library");
    if (device < 0 || device >= ndev) return fail(DD_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    const int W = b->n_windows;
    std::vector<int64_t> pair_off(W + 1), hh_off(W + 1);
    dd_batch_offsets(b, pair_off.data(), nullptr, nullptr);
    dd_pair_sum_offsets(b, hh_off.data());
    if (hh_off[W] == 0) return DD_SUCCESS;
    DeviceCtx &ctx = g_ctx.c;
    if ((rc = ctx.reserve(device, (size_t)(2 * (W + 1) * 4 + 2 * (W + 1) * 8 + (sz.n_pairs + hh_off[W]) * 8 + 16 * 256), 0))) return rc;
    DevBuf dev(ctx);
    dd_device_batch db;
    memset(&db, 0, sizeof(db));
    db.n_windows = W;
    if ((rc = dev.upload(&db.win_hap_off, b->win_hap_off, (size_t)W + 1))) return rc;
    if ((rc = dev.upload(&db.win_read_off, b->win_read_off, (size_t)W + 1))) return rc;
    if ((rc = dev.upload(&db.win_pair_off, (const int64_t *)pair_off.data(), pair_off.size()))) return rc;
    const int64_t *hh_dev = nullptr;
    const double *ll_dev = nullptr;
    double *out_dev = nullptr;
    if ((rc = dev.upload(&hh_dev, (const int64_t *)hh_off.data(), hh_off.size()))) return rc;
    if ((rc = dev.upload(&ll_dev, ll_host, (size_t)sz.n_pairs))) return rc;
    if ((rc = dev.alloc(&out_dev, (size_t)hh_off[W]))) return rc;
    rc = dd_pair_sums_device(&db, hh_dev, hh_off[W], ll_dev, out_dev, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_host, out_dev, (size_t)hh_off[W] * sizeof(double), hipMemcpyDeviceToHost));
    return DD_SUCCESS;
}

static int compute_likelihoods_impl(Model model, const dd_params *p, const dd_batch *b, dd_result *r, int device);

int dd_compute_likelihoods(const dd_params *p, const dd_batch *b, dd_result *r, int device)
{
    return compute_likelihoods_impl(MODEL_FBMAXERR, p, b, r, device);
}

int dd_compute_likelihoods_faster(const dd_params *p, const dd_batch *b, dd_result *r, int device)
{
    return compute_likelihoods_impl(MODEL_S, p, b, r, device);
}

namespace {
// true iff some p[i] >= limit.  The arrays checked this way hold one byte per read base (2e8 for configs[1]): a branch-free
// pass the compiler vectorises, cut into pieces for a few threads when it is long.
bool any_at_or_above(const uint8_t *p, size_t n, unsigned limit)
{
    if (limit > 255 || n == 0) return false;
    auto scan = [p, limit](size_t lo, size_t hi) -> unsigned {
        unsigned bad = 0;
        for (size_t i = lo; i < hi; i++) bad |= (unsigned)(p[i] >= limit);
        return bad;
    };
    const size_t piece = (size_t)8 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>(std::min<size_t>(8, hw ? hw : 1), (n + piece - 1) / piece);
    if (nt <= 1) return scan(0, n) != 0;
    std::vector<unsigned> bad(nt, 0);
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; t++) th.emplace_back([&, t]() { bad[t] = scan(n * t / nt, n * (t + 1) / nt); });
    bad[0] = scan(0, n / nt);
    for (auto &x : th) x.join();
    for (size_t t = 0; t < nt; t++) if (bad[t]) return true;
    return false;
}

// DD_TIMING=1: where a host-pointer call spends its time outside the kernels (stderr, one line per call)
struct StageClock {
    bool on; std::chrono::steady_clock::time_point t; std::string line;
    StageClock() : on(getenv("DD_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        const std::chrono::steady_clock::time_point n = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof(buf), " %s=%.2fms", what, std::chrono::duration<double, std::milli>(n - t).count());
        line += buf;
        t = n;
    }
    ~StageClock() { if (on) fprintf(stderr, "dd_timing:%s\n", line.c_str()); }
};
}

static int compute_likelihoods_impl(Model model, const dd_params *p, const dd_batch *b, dd_result *r, int device)
{
    StageClock clk;
    launch_log_clear();
    int rc = check_params(p);
    if (rc) return rc;
    if (!r || !r->ll || !r->status) return fail(DD_ERR_INVALID, "ll and status outputs are required");
    dd_sizes sz;
    rc = dd_batch_sizes(b, &sz);
    if (rc) return rc;
    if (sz.n_pairs == 0) {          // reads without haplotypes: no pair, but onHap[r] is an output per READ (the onHap kernel writes 0 there)
        if (r->onHap && sz.n_reads > 0) memset(r->onHap, 0, (size_t)sz.n_reads * sizeof(*r->onHap));
        return DD_SUCCESS;
    }
    // validate content
    if (!b->hap_seq || !b->read_seq || !b->read_qidx || !b->read_mqidx || !b->read_start || !b->read_flags ||
        !b->win_hap_start || !b->qual_table || !b->mapq_table)
        return fail(DD_ERR_INVALID, "null input array");
    if (b->n_qual < 1 || b->n_qual > DD_MAX_QUAL_TABLE || b->n_mapq < 1 || b->n_mapq > DD_MAX_QUAL_TABLE)
        return fail(DD_ERR_INVALID, "quality tables must hold 1..256 entries");
    // windows whose shape the kernels do not cover are skipped one by one (DD_PAIR_UNSUPPORTED), not the batch
    std::vector<uint8_t> win_skip((size_t)b->n_windows);
    int32_t ok_max[2] = {0, 0};
    if (!b->hap_seq_off || !b->read_seq_off) return fail(DD_ERR_INVALID, "null offset array");
    uint8_t sym_lut[256];
    const int sym_left = assign_symbols(b, sym_lut);     // > 0: more than 26 distinct non-ACGTN haplotype bytes in the batch
    if (sym_left < 0) return sym_left;
    clk.mark("symbols");
    const int n_skip = screen_windows(b, win_skip.data(), ok_max, sym_lut, sym_left > 0 && model == MODEL_FBMAXERR);
    sz.max_hap_len = ok_max[0] > 0 ? ok_max[0] : 1;      // planning maxima: the windows that are computed
    sz.max_read_len = ok_max[1] > 0 ? ok_max[1] : 1;
    clk.mark("screen");
    std::vector<double> lib_logprob, lib_log95;
    if (p->mapUnmappedReads && model == MODEL_FBMAXERR) {
        if (!b->read_mate_pos || !b->read_mate_len || !b->read_lib)
            return fail(DD_ERR_INVALID, "mapUnmappedReads needs read_mate_pos, read_mate_len and read_lib");
        if (b->n_libs < 1 || !b->lib_off) return fail(DD_ERR_INVALID, "mapUnmappedReads needs the library tables");
        lib_logprob.resize((size_t)(b->lib_off[b->n_libs] > 0 ? b->lib_off[b->n_libs] : 1));
        lib_log95.resize((size_t)b->n_libs);
        if ((rc = dd_build_library_tables(b, lib_logprob.data(), lib_log95.data()))) return rc;
        for (int64_t q = 0; q < sz.n_reads; q++)
            if (b->read_lib[q] >= b->n_libs) return fail(DD_ERR_INVALID, "read_lib out of range");
    }
    if (any_at_or_above(b->read_mqidx, (size_t)sz.n_reads, (unsigned)b->n_mapq)) return fail(DD_ERR_INVALID, "read_mqidx out of range");
    if (any_at_or_above(b->read_qidx, (size_t)sz.read_bases, (unsigned)b->n_qual)) return fail(DD_ERR_INVALID, "read_qidx out of range");
    for (int i = 0; i < b->n_qual; i++)
        if (!(b->qual_table[i] >= 0.0 && b->qual_table[i] <= 1.0)) return fail(DD_ERR_INVALID, "base quality outside [0,1]");
    for (int i = 0; i < b->n_mapq; i++)
        if (!(b->mapq_table[i] >= 0.0 && b->mapq_table[i] < 1.0)) return fail(DD_ERR_INVALID, "mapping quality outside [0,1)");

    clk.mark("validate");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DD_ERR_NO_DEVICE, "no HIP device: the likelihood path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(DD_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));

    const int W = b->n_windows;
    std::vector<int32_t> hap_window((size_t)sz.n_haps);
    std::vector<int64_t> pair_off(W + 1), hpos_off(W + 1), vc_off(W + 1);
    dd_build_index(b, hap_window.data(), pair_off.data(), hpos_off.data(), vc_off.data());
    std::vector<double> tables(DD_TABLE_DOUBLES);
    rc = dd_build_tables(p, b->qual_table, b->n_qual, b->mapq_table, b->n_mapq, tables.data());
    if (rc < 0) return rc;

    // ---- host-side planning first (no device work yet): launch classes and scratch size ----
    // Ragged batches: one launch per (lane tiling of the haplotypes, read-length interval) that has work (build_launch_classes) — a single
    // 170-bp haplotype or 250-bp read does not drag every pair of the batch onto the K = 3 / long-read build.
    dd_length_classes lcls;
    std::vector<int32_t> class_list;
    memset(&lcls, 0, sizeof(lcls));
    if (model == MODEL_FBMAXERR && !getenv("DD_NO_LENGTH_CLASSES")) {                      // env: A/B only
        class_list.resize((size_t)sz.n_haps * DD_N_READ_CLASSES + 1);
        if ((rc = build_launch_classes(b, n_skip ? win_skip.data() : nullptr, p, class_list.data(), &lcls))) return rc;
    }
    const int32_t *class_list_dev = nullptr;
    dd_device_batch db;
    memset(&db, 0, sizeof(db));
    db.n_windows = W; db.n_haps = (int32_t)sz.n_haps; db.n_reads = (int32_t)sz.n_reads;
    db.max_hap_len = sz.max_hap_len; db.max_read_len = sz.max_read_len;
    db.n_qual = b->n_qual; db.n_mapq = b->n_mapq;
    for (int w = 0; w < W; w++)
        if (!win_skip[(size_t)w] && b->win_read_off[w + 1] - b->win_read_off[w] > db.max_window_reads) db.max_window_reads = b->win_read_off[w + 1] - b->win_read_off[w];
    size_t ws_bytes = model == MODEL_S ? 0 : dd_workspace_bytes(p, &db);
    const bool single_class = lcls.n_launches <= 1;
    if (!single_class) {
        db.classes = &lcls;
        db.hap_class_list = class_list.data();             // (host pointer: only dd_workspace_bytes' class walk looks at it here)
        const size_t w2 = dd_workspace_bytes(p, &db);
        if (w2 > ws_bytes) ws_bytes = w2;
        db.classes = nullptr; db.hap_class_list = nullptr;
    }

    clk.mark("plan");
    // ---- device arena (cached per host thread) ----
    const size_t np = (size_t)sz.n_pairs;
    const size_t n_var = b->hap_var_off ? (size_t)b->hap_var_off[sz.n_haps] : 0;
    const size_t in_bytes = (size_t)(W + 1) * (4 + 4 + 8 + 8 + 8) + (size_t)W * 4 + (size_t)(sz.n_haps + 1) * 8 + (size_t)sz.hap_bases +
                            (size_t)(sz.n_reads + 1) * 4 + (size_t)sz.read_bases * 2 + (size_t)sz.n_reads * 6 + n_var * 20 +
                            (size_t)sz.n_haps * 4 + (size_t)lcls.list_len * 4 + 64 + DD_TABLE_DOUBLES * 8 + 48 * 256 + (size_t)W + 256 +
                            (lib_log95.empty() ? 0 : (size_t)sz.n_reads * 9 + (lib_logprob.size() + lib_log95.size()) * 8 + (size_t)(b->n_libs + 1) * 4);
    const size_t out_bytes = np * (4 * 8 + 2 + 8 * 2 + 4) + (size_t)sz.hpos_len * 2 + 2 * (size_t)sz.var_cov_len + (size_t)sz.n_reads + 24 * 256;
    const bool staged = in_bytes + out_bytes <= (size_t)64 << 20;    // small batch: one H2D, one D2H through the pinned mirror
    DeviceCtx &ctx = g_ctx.c;
    if ((rc = ctx.reserve(device, in_bytes + out_bytes + 2 * (ws_bytes + 256), staged ? in_bytes + out_bytes : 0))) return rc;
    clk.mark("reserve");
    DevBuf dev(ctx);
    dev.staged = staged;
#define UP(field, n) if ((rc = dev.upload(&db.field, b->field, (size_t)(n)))) return rc
    UP(win_hap_off, W + 1); UP(win_read_off, W + 1); UP(win_hap_start, W);
    UP(hap_seq_off, sz.n_haps + 1); UP(hap_seq, sz.hap_bases);
    UP(read_seq_off, sz.n_reads + 1);
    // The two big inputs (one byte per read base each).  Large batches: only reserved here — each window block's share is copied
    // on the block's own stream right in front of its kernels, so all but the first block's transfer hides behind the kernels of
    // the block before.
    const bool late_reads = !staged && sz.read_bases > 0;
    char *d_read_seq = nullptr; uint8_t *d_read_qidx = nullptr;
    if (late_reads) {
        if ((rc = dev.alloc(&d_read_seq, (size_t)sz.read_bases))) return rc;
        if ((rc = dev.alloc(&d_read_qidx, (size_t)sz.read_bases))) return rc;
        db.read_seq = d_read_seq; db.read_qidx = d_read_qidx;
    } else {
        // Staged (small) batches whose two big inputs the caller keeps in page-locked, device-addressable memory (the C++ adapter packs
        // into dd_host_alloc buffers): the kernels read them in place over the link — each (read, haplotype) pair fetches its read once,
        // ~80 MB per 256-window batch at 8 haplotypes — instead of waiting for a staged copy whose blit kernels share the CUs with the
        // batch that is running (profiles/r03/window_loop_timeline.txt).  DD_ZERO_COPY_IN=0 switches it off (A/B).
        static const bool zero_copy_in = !(getenv("DD_ZERO_COPY_IN") && !strcmp(getenv("DD_ZERO_COPY_IN"), "0"));
        auto mapped_in = [&](const void *host, size_t bytes) -> const void * {
            if (!zero_copy_in || !host || !bytes) return nullptr;
            hipPointerAttribute_t a, e;
            if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
            if (hipPointerGetAttributes(&e, static_cast<const unsigned char *>(host) + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            if (e.type != hipMemoryTypeHost || !e.devicePointer ||
                static_cast<unsigned char *>(e.devicePointer) - static_cast<unsigned char *>(a.devicePointer) != (ptrdiff_t)(bytes - 1)) return nullptr;
            return a.devicePointer;
        };
        const void *ms = mapped_in(b->read_seq, (size_t)sz.read_bases), *mq = mapped_in(b->read_qidx, (size_t)sz.read_bases);
        if (ms && mq) { db.read_seq = static_cast<const char *>(ms); db.read_qidx = static_cast<const uint8_t *>(mq); }
        else { UP(read_seq, sz.read_bases); UP(read_qidx, sz.read_bases); }
    }
    UP(read_mqidx, sz.n_reads); UP(read_start, sz.n_reads); UP(read_flags, sz.n_reads);
#undef UP
    if (b->hap_var_off) {
        if ((rc = dev.upload(&db.hap_var_off, b->hap_var_off, (size_t)sz.n_haps + 1))) return rc;
        if ((rc = dev.upload(&db.hap_var, b->hap_var, 2 * n_var))) return rc;
        if (b->hap_var_flank && (rc = dev.upload(&db.hap_var_flank, b->hap_var_flank, 3 * n_var))) return rc;
    }
    if ((rc = dev.upload(&db.hap_window, (const int32_t *)hap_window.data(), hap_window.size()))) return rc;
    if ((rc = dev.upload(&db.win_pair_off, (const int64_t *)pair_off.data(), pair_off.size()))) return rc;
    if ((rc = dev.upload(&db.win_hpos_off, (const int64_t *)hpos_off.data(), hpos_off.size()))) return rc;
    if ((rc = dev.upload(&db.win_varcov_off, (const int64_t *)vc_off.data(), vc_off.size()))) return rc;
    if ((rc = dev.upload(&db.tables, (const double *)tables.data(), tables.size()))) return rc;
    if ((rc = dev.upload(&db.sym_lut, (const uint8_t *)sym_lut, (size_t)256))) return rc;
    if (n_skip > 0 && (rc = dev.upload(&db.win_skip, (const uint8_t *)win_skip.data(), win_skip.size()))) return rc;
    if (!lib_log95.empty()) {
        if ((rc = dev.upload(&db.read_mate_pos, b->read_mate_pos, (size_t)sz.n_reads))) return rc;
        if ((rc = dev.upload(&db.read_mate_len, b->read_mate_len, (size_t)sz.n_reads))) return rc;
        if ((rc = dev.upload(&db.read_lib, b->read_lib, (size_t)sz.n_reads))) return rc;
        if ((rc = dev.upload(&db.lib_off, b->lib_off, (size_t)b->n_libs + 1))) return rc;
        if ((rc = dev.upload(&db.lib_logprob, (const double *)lib_logprob.data(), lib_logprob.size()))) return rc;
        if ((rc = dev.upload(&db.lib_log95, (const double *)lib_log95.data(), lib_log95.size()))) return rc;
    }
    if (!single_class && (rc = dev.upload(&class_list_dev, (const int32_t *)class_list.data(), (size_t)lcls.list_len))) return rc;
    if ((rc = dev.flush_uploads(ctx.s[0]))) return rc;       // staged mode: the one H2D copy
    if (!staged) HIP_TRY(hipStreamSynchronize(nullptr));     // pageable uploads went through the null stream's DMA
    clk.mark("upload");

    dd_result dr;
    memset(&dr, 0, sizeof(dr));
    const size_t out_begin = DevBuf::align(dev.used);
    // Output arrays the caller keeps in page-locked host memory that this device can address (dd_host_alloc, hipHostMalloc) are
    // written by the kernels themselves, over the link, as the pairs finish: no staging copy in HBM and no copy afterwards.
    // (The runtime carries device -> host copies of this size out with a copy KERNEL: profiles/r02/hostapi_timeline.txt shows
    // 300 ms of such kernels on the CUs beside 410 ms of HMM kernels for configs[1].)  status and offHapHMQ stay in HBM: the
    // onHap kernel reads them back.  DD_ZERO_COPY=0 switches this off (A/B).
    static const bool zero_copy_on = !(getenv("DD_ZERO_COPY") && !strcmp(getenv("DD_ZERO_COPY"), "0"));
    auto mapped = [&](void *host, size_t bytes) -> void * {
        if (!zero_copy_on || !host || !bytes) return nullptr;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
        hipPointerAttribute_t e;                                   // the last byte belongs to the same registered range
        if (hipPointerGetAttributes(&e, static_cast<unsigned char *>(host) + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (e.type != hipMemoryTypeHost || !e.devicePointer ||
            static_cast<unsigned char *>(e.devicePointer) - static_cast<unsigned char *>(a.devicePointer) != (ptrdiff_t)(bytes - 1)) return nullptr;
        return a.devicePointer;
    };
    struct { bool ll, llOn, llOff, mLogBQ, offHap, numIndels, numMismatch, nBQT, nmmBQT, nMMLeft, nMMRight, firstBase, lastBase, hpos, var_covered,
             status, onHap, var_fcov, offHapHMQ; } direct;
    memset(&direct, 0, sizeof(direct));
    int n_direct = 0;
#define OUT(field, n) if (r->field) { \
        void *m = mapped(r->field, (size_t)(n) * sizeof(*r->field)); \
        if (m) { dr.field = static_cast<decltype(dr.field)>(m); direct.field = true; n_direct++; } \
        else if ((rc = dev.alloc(&dr.field, (size_t)(n)))) return rc; }
#define OUT_DEVICE(field, n) if (r->field && (rc = dev.alloc(&dr.field, (size_t)(n)))) return rc
    OUT(ll, np); OUT(llOn, np); OUT(llOff, np); OUT(mLogBQ, np); OUT(offHap, np); OUT(numIndels, np);
    OUT(numMismatch, np); OUT(nBQT, np); OUT(nmmBQT, np); OUT(nMMLeft, np); OUT(nMMRight, np); OUT(firstBase, np);
    OUT(lastBase, np); OUT(hpos, sz.hpos_len); OUT(var_covered, sz.var_cov_len); OUT_DEVICE(status, np); OUT(onHap, sz.n_reads);
    OUT(var_fcov, sz.var_cov_len);
#undef OUT
#undef OUT_DEVICE
    g_last_direct = n_direct;
    if ((r->offHapHMQ || r->onHap) && (rc = dev.alloc(&dr.offHapHMQ, np))) return rc;   // onHap is derived from it
    const size_t out_end = dev.used;
    unsigned char *ws[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++)
        if (ws_bytes && (rc = dev.alloc(&ws[i], ws_bytes))) return rc;
    struct { hipStream_t s[2]; } streams = {{ctx.s[0], ctx.s[1]}};

    // Chunked, double-buffered execution: contiguous window blocks alternate between two streams, and the D2H
    // of block c is issued after the kernel of block c+1 has been enqueued, so the copy engine drains results
    // while the CUs work on the next block (with pageable user memory the copy call blocks the host thread, not
    // the GPU).  Each stream has its own back-pointer scratch.  Small (staged) batches run as one block on stream 0
    // and come back with one D2H through the pinned mirror.
    int n_chunks = staged ? 1 : (int)((sz.n_pairs + 999999) / 1000000);
    if (n_chunks > 64) n_chunks = 64;
    if (n_chunks > W) n_chunks = W;
    if (n_chunks < 1) n_chunks = 1;
    std::vector<int> cw(n_chunks + 1, 0);      // window boundaries with ~equal pair counts
    for (int c = 1; c < n_chunks; c++) {
        const int64_t target = sz.n_pairs * c / n_chunks;
        int w = cw[c - 1];
        while (w < W && pair_off[w] < target) w++;
        cw[c] = w;
    }
    cw[n_chunks] = W;
#define DOWN(field, off, n) if (r->field && (n) && !direct.field) HIP_TRY(hipMemcpyAsync(r->field + (off), dr.field + (off), (size_t)(n) * sizeof(*r->field), hipMemcpyDeviceToHost, st))
    auto download = [&](int c) -> int {
        hipStream_t st = streams.s[c & 1];
        const int w0 = cw[c], w1 = cw[c + 1];
        const int64_t p0 = pair_off[w0], pn = pair_off[w1] - p0;
        DOWN(ll, p0, pn); DOWN(llOn, p0, pn); DOWN(llOff, p0, pn); DOWN(mLogBQ, p0, pn); DOWN(offHap, p0, pn); DOWN(offHapHMQ, p0, pn);
        DOWN(numIndels, p0, pn); DOWN(numMismatch, p0, pn); DOWN(nBQT, p0, pn); DOWN(nmmBQT, p0, pn); DOWN(nMMLeft, p0, pn);
        DOWN(nMMRight, p0, pn); DOWN(firstBase, p0, pn); DOWN(lastBase, p0, pn); DOWN(status, p0, pn);
        DOWN(hpos, hpos_off[w0], hpos_off[w1] - hpos_off[w0]);
        DOWN(var_covered, vc_off[w0], vc_off[w1] - vc_off[w0]);
        DOWN(var_fcov, vc_off[w0], vc_off[w1] - vc_off[w0]);
        DOWN(onHap, b->win_read_off[w0], b->win_read_off[w1] - b->win_read_off[w0]);
        return DD_SUCCESS;
    };
    auto enqueue_and_collect = [&]() -> int {
    struct ChunkFlag { bool prev; explicit ChunkFlag(bool v) : prev(tl_overlapping_chunks) { tl_overlapping_chunks = v; } ~ChunkFlag() { tl_overlapping_chunks = prev; } } chunk_flag(n_chunks > 1);
    for (int c = 0; c < n_chunks; c++) {
        const int w0 = cw[c], w1 = cw[c + 1];
        const int g0 = b->win_hap_off[w0], g1 = b->win_hap_off[w1], q0 = b->win_read_off[w0], q1 = b->win_read_off[w1];
        if (late_reads && q1 > q0) {
            const size_t s0 = (size_t)b->read_seq_off[q0], sn = (size_t)b->read_seq_off[q1] - s0;
            HIP_TRY(hipMemcpyAsync(d_read_seq + s0, b->read_seq + s0, sn, hipMemcpyHostToDevice, streams.s[c & 1]));
            HIP_TRY(hipMemcpyAsync(d_read_qidx + s0, b->read_qidx + s0, sn, hipMemcpyHostToDevice, streams.s[c & 1]));
        }
        if (single_class) {
            rc = launch_range(model, p, &db, &dr, ws[c & 1], ws_bytes, streams.s[c & 1], g0, g1, q0, q1);
            if (rc) return rc;
        } else {
            // every launch class of this window block, then onHap once
            for (int i = 0; i < lcls.n_launches; i++) {
                const dd_launch_class &L = lcls.launch[i];
                const int32_t *hl = class_list.data() + L.list_off;
                LenClass lc;
                lc.hap_list = class_list_dev + L.list_off;
                lc.list_begin = (int)(std::lower_bound(hl, hl + L.list_len, g0) - hl);
                lc.list_end = (int)(std::lower_bound(hl, hl + L.list_len, g1) - hl);
                lc.max_hap_len = L.max_hap_len;
                lc.min_read_len = L.min_read_len; lc.max_read_len = L.max_read_len;
                lc.max_window_reads = L.max_window_reads; lc.avg_window_reads = L.avg_window_reads; lc.avg_read_len = L.avg_read_len;
                lc.run_onhap = (i == lcls.n_launches - 1);
                rc = launch_range(model, p, &db, &dr, ws[c & 1], ws_bytes, streams.s[c & 1], g0, g1, q0, q1, &lc);
                if (rc) return rc;
            }
        }
        if (!staged && c > 0 && (rc = download(c - 1))) return rc;
    }
    if (staged) {
        HIP_TRY(hipMemcpyAsync(ctx.pinned + out_begin, ctx.arena + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, streams.s[0]));
        HIP_TRY(hipStreamSynchronize(streams.s[0]));
#define BACK(field, n) if (r->field && (n) && !direct.field) memcpy(r->field, ctx.pinned + (reinterpret_cast<unsigned char *>(dr.field) - ctx.arena), (size_t)(n) * sizeof(*r->field))
        BACK(ll, np); BACK(llOn, np); BACK(llOff, np); BACK(mLogBQ, np); BACK(offHap, np); BACK(offHapHMQ, np);
        BACK(numIndels, np); BACK(numMismatch, np); BACK(nBQT, np); BACK(nmmBQT, np); BACK(nMMLeft, np); BACK(nMMRight, np);
        BACK(firstBase, np); BACK(lastBase, np); BACK(status, np); BACK(hpos, sz.hpos_len); BACK(var_covered, sz.var_cov_len);
        BACK(onHap, sz.n_reads); BACK(var_fcov, sz.var_cov_len);
#undef BACK
    } else {
        if ((rc = download(n_chunks - 1))) return rc;
        HIP_TRY(hipStreamSynchronize(streams.s[0]));
        HIP_TRY(hipStreamSynchronize(streams.s[1]));
    }
    return DD_SUCCESS;
    };
#undef DOWN
    clk.mark("outputs");
    rc = enqueue_and_collect();
    clk.mark("run");
    if (rc != DD_SUCCESS) {
        // kernels / copies already enqueued keep writing into the caller's buffers and this thread's arena: drain both
        // streams before the error is reported, so that neither is reused while still in flight
        const std::string msg = g_err;
        (void)hipStreamSynchronize(streams.s[0]);
        (void)hipStreamSynchronize(streams.s[1]);
        g_err = msg;
    }
    return rc;
}

// ---------------- several devices of one process ----------------
int dd_partition_windows(const dd_batch *b, int n_parts, int32_t *bounds)
{
    if (!b || !bounds || n_parts < 1) return fail(DD_ERR_INVALID, "null argument");
    if (b->n_windows < 0 || !b->win_hap_off || !b->win_read_off || !b->hap_seq_off || !b->read_seq_off)
        return fail(DD_ERR_INVALID, "null offset array");
    const int W = b->n_windows;
    std::vector<double> cum((size_t)W + 1, 0.0);          // cells before window w (exact in a double far beyond any batch)
    for (int w = 0; w < W; w++) {
        const int64_t SH = (int64_t)b->hap_seq_off[b->win_hap_off[w + 1]] - b->hap_seq_off[b->win_hap_off[w]];
        const int64_t SL = (int64_t)b->read_seq_off[b->win_read_off[w + 1]] - b->read_seq_off[b->win_read_off[w]];
        cum[(size_t)w + 1] = cum[(size_t)w] + (double)SH * (double)SL;
    }
    bounds[0] = 0;
    for (int i = 1; i < n_parts; i++) {
        // first boundary whose prefix reaches i/n of the work, moved one window back when that lands closer
        const double target = cum[(size_t)W] * (double)i / (double)n_parts;
        int w = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        if (w > 0 && target - cum[(size_t)w - 1] < cum[(size_t)w] - target) w--;
        if (w < bounds[i - 1]) w = bounds[i - 1];
        if (w > W) w = W;
        bounds[i] = w;
    }
    bounds[n_parts] = W;
    return DD_SUCCESS;
}

namespace {

// One persistent host thread per block slot: its thread_local device cache (arena, pinned mirror, streams) survives between
// calls, which per-call threads would allocate and free every time.  The threads wait for work for the life of the process.
class SlotWorker {
public:
    SlotWorker() : has_job_(false), done_(true) { th_ = std::thread([this]() { loop(); }); th_.detach(); }
    void submit(std::function<void()> f)
    {
        std::unique_lock<std::mutex> lk(m_);
        job_ = std::move(f); has_job_ = true; done_ = false;
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this]() { return done_; });
    }
private:
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this]() { return has_job_; });
                f = std::move(job_); has_job_ = false;
            }
            f();
            {
                std::unique_lock<std::mutex> lk(m_);
                done_ = true;
                cv_.notify_all();
            }
        }
    }
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool has_job_, done_;
};

std::mutex g_multi_mutex;                       // one multi-device call at a time per process (the slots are shared)
std::vector<SlotWorker *> g_slots;              // never destroyed: the threads outlive static destruction

int compute_multi(Model model, const dd_params *p, const dd_batch *b, dd_result *r, const int *devices, int n)
{
    if (!devices || n < 1) return fail(DD_ERR_INVALID, "dd_compute_likelihoods_multi: no devices");
    if (n == 1) return compute_likelihoods_impl(model, p, b, r, devices[0]);
    int rc = check_params(p);
    if (rc) return rc;
    if (!r || !r->ll || !r->status) return fail(DD_ERR_INVALID, "ll and status outputs are required");
    dd_sizes sz;
    if ((rc = dd_batch_sizes(b, &sz))) return rc;
    if (sz.n_pairs == 0) {          // reads without haplotypes: no pair, but onHap[r] is an output per READ (the onHap kernel writes 0 there)
        if (r->onHap && sz.n_reads > 0) memset(r->onHap, 0, (size_t)sz.n_reads * sizeof(*r->onHap));
        return DD_SUCCESS;
    }
    const int W = b->n_windows;
    std::vector<int32_t> bounds((size_t)n + 1);
    if ((rc = dd_partition_windows(b, n, bounds.data()))) return rc;
    std::vector<int64_t> pair_off((size_t)W + 1), hpos_off((size_t)W + 1), vc_off((size_t)W + 1);
    dd_batch_offsets(b, pair_off.data(), hpos_off.data(), vc_off.data());

    struct Block {
        std::vector<int32_t> win_hap_off, win_read_off, hap_seq_off, read_seq_off, hap_var_off;
        dd_batch b; dd_result r; int rc; std::string err;
    };
    std::vector<std::unique_ptr<Block> > blocks;
    for (int i = 0; i < n; i++) {
        const int w0 = bounds[(size_t)i], w1 = bounds[(size_t)i + 1];
        std::unique_ptr<Block> B(new Block());
        B->rc = DD_SUCCESS;
        const int h0 = b->win_hap_off[w0], h1 = b->win_hap_off[w1], q0 = b->win_read_off[w0], q1 = b->win_read_off[w1];
        const int hs0 = b->hap_seq_off[h0], rs0 = b->read_seq_off[q0];
        const int v0 = b->hap_var_off ? b->hap_var_off[h0] : 0;
        // offset arrays rebased to the block; data arrays are the caller's, shifted
        for (int w = w0; w <= w1; w++) { B->win_hap_off.push_back(b->win_hap_off[w] - h0); B->win_read_off.push_back(b->win_read_off[w] - q0); }
        for (int h = h0; h <= h1; h++) {
            B->hap_seq_off.push_back(b->hap_seq_off[h] - hs0);
            if (b->hap_var_off) B->hap_var_off.push_back(b->hap_var_off[h] - v0);
        }
        for (int q = q0; q <= q1; q++) B->read_seq_off.push_back(b->read_seq_off[q] - rs0);
        dd_batch &s = B->b;
        s = *b;
        s.n_windows = w1 - w0;
        s.win_hap_off = B->win_hap_off.data(); s.win_read_off = B->win_read_off.data();
        s.win_hap_start = b->win_hap_start ? b->win_hap_start + w0 : nullptr;
        s.hap_seq_off = B->hap_seq_off.data(); s.hap_seq = b->hap_seq ? b->hap_seq + hs0 : nullptr;
        s.hap_var_off = b->hap_var_off ? B->hap_var_off.data() : nullptr;
        s.hap_var = b->hap_var ? b->hap_var + 2 * (size_t)v0 : nullptr;
        s.hap_var_flank = b->hap_var_flank ? b->hap_var_flank + 3 * (size_t)v0 : nullptr;
        s.read_seq_off = B->read_seq_off.data();
        s.read_seq = b->read_seq ? b->read_seq + rs0 : nullptr;
        s.read_qidx = b->read_qidx ? b->read_qidx + rs0 : nullptr;
        s.read_mqidx = b->read_mqidx ? b->read_mqidx + q0 : nullptr;
        s.read_start = b->read_start ? b->read_start + q0 : nullptr;
        s.read_flags = b->read_flags ? b->read_flags + q0 : nullptr;
        s.read_mate_pos = b->read_mate_pos ? b->read_mate_pos + q0 : nullptr;
        s.read_mate_len = b->read_mate_len ? b->read_mate_len + q0 : nullptr;
        s.read_lib = b->read_lib ? b->read_lib + q0 : nullptr;
        dd_result &o = B->r;
        memset(&o, 0, sizeof(o));
        const int64_t p0 = pair_off[(size_t)w0];
#define SHIFT(f, off) o.f = r->f ? r->f + (off) : nullptr
        SHIFT(ll, p0); SHIFT(llOn, p0); SHIFT(llOff, p0); SHIFT(mLogBQ, p0); SHIFT(offHap, p0); SHIFT(offHapHMQ, p0);
        SHIFT(numIndels, p0); SHIFT(numMismatch, p0); SHIFT(nBQT, p0); SHIFT(nmmBQT, p0); SHIFT(nMMLeft, p0); SHIFT(nMMRight, p0);
        SHIFT(firstBase, p0); SHIFT(lastBase, p0); SHIFT(status, p0);
        SHIFT(hpos, hpos_off[(size_t)w0]); SHIFT(var_covered, vc_off[(size_t)w0]); SHIFT(var_fcov, vc_off[(size_t)w0]); SHIFT(onHap, q0);
#undef SHIFT
        blocks.push_back(std::move(B));
    }
    std::lock_guard<std::mutex> g(g_multi_mutex);
    while ((int)g_slots.size() < n) g_slots.push_back(new SlotWorker());
    for (int i = 0; i < n; i++) {
        Block *B = blocks[(size_t)i].get();
        const int dev = devices[i];
        g_slots[(size_t)i]->submit([B, model, p, dev]() {
            B->rc = compute_likelihoods_impl(model, p, &B->b, &B->r, dev);
            if (B->rc != DD_SUCCESS) B->err = g_err;          // the worker thread's message
        });
    }
    for (int i = 0; i < n; i++) g_slots[(size_t)i]->wait();
    for (int i = 0; i < n; i++)
        if (blocks[(size_t)i]->rc != DD_SUCCESS)
            return fail(blocks[(size_t)i]->rc, "block " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + blocks[(size_t)i]->err);
    return DD_SUCCESS;
}

} // namespace

int dd_compute_likelihoods_multi(const dd_params *p, const dd_batch *b, dd_result *r, const int *devices, int n_devices)
{
    return compute_multi(MODEL_FBMAXERR, p, b, r, devices, n_devices);
}

int dd_compute_likelihoods_faster_multi(const dd_params *p, const dd_batch *b, dd_result *r, const int *devices, int n_devices)
{
    return compute_multi(MODEL_S, p, b, r, devices, n_devices);
}

} // extern "C"
