// valu_rate.hip — issue cost of the VALU instructions the HMM kernels are made of, on gfx950, at 1..4 waves per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
//
// Each test is a loop of 64 independent instructions of one kind (8 register sets round-robin, so no instruction waits on
// the one before it) run by every wave of a workgroup of 256 x W threads (W waves per SIMD) on every CU; the figure printed
// is SIMD cycles per wave-instruction = elapsed shader cycles (s_memtime) x 1 / (instructions issued per SIMD), i.e. the
// reciprocal throughput one SIMD sustains for that instruction.  Prices the instruction-mix table of profiles/r02.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITER 2000

// LANES < 64: the timed loop runs under an EXEC mask of the wave's first LANES lanes (what the HMM kernel's one-lane LO / RO blocks do)
template <int OP, int LANES = 64>
__global__ void __launch_bounds__(1024) rate_kernel(double *out, unsigned long long *cyc, double seed)
{
    double a[8], b[8];
    unsigned u[8];
    float f[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1) + threadIdx.x * 1e-9; b[i] = -seed * (i + 2); u[i] = threadIdx.x + i; f[i] = float(i) + float(seed); }
    if (OP >= 14 && OP <= 16)
        asm volatile("v_mov_b64 v[100:101], %0\n v_mov_b64 v[108:109], %1\n v_mov_b32 v106, 0\n v_mov_b32 v114, 0" : : "v"(b[0]), "v"(b[1])
                     : "v100", "v101", "v106", "v108", "v109", "v114");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (LANES == 64 || (threadIdx.x & 63) < LANES)
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 4) & 7]) : );
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 9) {      // the hysteresis step of the right->middle pass: add, compare, 3 selects
#define X(i) asm volatile("v_add_f64 %1, %0, %4\n v_cmp_gt_f64 vcc, %3, %1\n v_cndmask_b32 %2, %2, %5, vcc" : "+v"(a[i]), "+v"(b[i]), "+v"(u[i]) : "v"(a[(i + 1) & 7]), "v"(seed), "v"(u[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 13) {
#define X(i) asm volatile("v_cmp_ge_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(a[i]), "+v"(u[i]) : "v"(b[i]), "v"(u[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 14) {     // one candidate of the right->middle pass as the compiler emits it: ONE chain through `best` (v[100:101]); 8 per r
                                       // (fixed physical registers: the selects work on the halves of the 64-bit values)
#define X(i) asm volatile("v_add_f64 v[104:105], %0, %1\n v_add_f64 v[104:105], v[104:105], %2\n v_add_f64 v[102:103], v[100:101], %3\n" \
                          "v_cmp_gt_f64 vcc, v[104:105], v[102:103]\n s_nop 1\n" \
                          "v_cndmask_b32 v101, v101, v105, vcc\n v_cndmask_b32 v100, v100, v104, vcc\n v_cndmask_b32 v106, v106, %4, vcc" \
                          : : "v"(a[1 + (i % 3)]), "v"(b[1 + (i % 5)]), "v"(b[6]), "v"(seed), "v"(u[1 + (i & 3)]) \
                          : "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106");
                REP8(X)
#undef X
            } else if (OP == 15) {     // the same work for TWO positions, the two chains (v[100:101], v[108:109]) interleaved, compare results in SGPR pairs:
                                       // two VALU instructions lie between a compare and the first select that reads it, no s_nop
#define X(i) asm volatile("v_add_f64 v[104:105], %0, %1\n v_add_f64 v[112:113], %0, %2\n v_add_f64 v[104:105], v[104:105], %2\n v_add_f64 v[112:113], v[112:113], %1\n" \
                          "v_add_f64 v[102:103], v[100:101], %3\n v_cmp_gt_f64 s[20:21], v[104:105], v[102:103]\n" \
                          "v_add_f64 v[110:111], v[108:109], %3\n v_cmp_gt_f64 s[22:23], v[112:113], v[110:111]\n" \
                          "v_cndmask_b32 v101, v101, v105, s[20:21]\n v_cndmask_b32 v100, v100, v104, s[20:21]\n v_cndmask_b32 v106, v106, %4, s[20:21]\n" \
                          "v_cndmask_b32 v109, v109, v113, s[22:23]\n v_cndmask_b32 v108, v108, v112, s[22:23]\n v_cndmask_b32 v114, v114, %4, s[22:23]" \
                          : : "v"(a[2 + (i % 3)]), "v"(b[2 + (i % 5)]), "v"(b[7]), "v"(seed), "v"(u[2 + (i & 3)]) \
                          : "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v108", "v109", "v110", "v111", "v112", "v113", "v114");
                REP8(X)
#undef X
            } else if (OP == 16) {     // one candidate of the left->middle pass: three adds, compare, max, one select (chain through v_max_f64 only)
#define X(i) asm volatile("v_add_f64 v[104:105], %0, %1\n v_add_f64 v[104:105], v[104:105], %2\n v_add_f64 v[104:105], v[104:105], %3\n" \
                          "v_cmp_nge_f64 vcc, v[104:105], v[100:101]\n v_max_f64 v[100:101], v[100:101], v[104:105]\n v_cndmask_b32 v106, v106, %4, vcc" \
                          : : "v"(a[1 + (i % 3)]), "v"(b[1 + (i % 5)]), "v"(b[6]), "v"(seed), "v"(u[1 + (i & 3)]) \
                          : "vcc", "v100", "v101", "v104", "v105", "v106");
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + u[i] + f[i];
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot < 1024u * 256u) out[slot] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)
static double *g_out = nullptr;
static unsigned long long *g_cyc = nullptr;

template <int OP, int LANES = 64> static void run(const char *name, int instr_per_x)
{
    for (int w = 1; w <= 4; w++) {   // (round 2's w = 4 fault: OP 9 / OP 13 wrote u[i] through an input-only operand, so the register that also held
                                     //  threadIdx.x drifted and the final store left the buffer; every written register is an output operand now)
        const int threads = 256 * w, blocks = 256;
        if (!g_out) {                                  // once, for the largest launch
            CHECK(hipMalloc(&g_out, sizeof(double) * 1024 * 256));
            CHECK(hipMalloc(&g_cyc, sizeof(unsigned long long) * 256 * 16));
        }
        double *out = g_out;
        unsigned long long *cyc = g_cyc;
        hipLaunchKernelGGL((rate_kernel<OP, LANES>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL((rate_kernel<OP, LANES>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * (threads / 64));
        CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : h) mean += double(v);
        mean /= double(h.size());
        // a SIMD runs w waves; each issues ITER*64*instr_per_x wave-instructions in `mean` cycles
        const double per = mean / (double(ITER) * 64.0 * instr_per_x * w);
        printf("%-34s waves/SIMD=%d  cycles per wave-instruction on one SIMD: %6.2f   (wave-time per instruction %6.2f)\n", name, w, per,
               mean / (double(ITER) * 64.0 * instr_per_x));
        fflush(stdout);
    }
}

int main()
{
    run<0>("v_add_f64", 1);
    run<1>("v_max_f64", 1);
    run<2>("v_cmp_gt_f64", 1);
    run<3>("v_cndmask_b32", 1);
    run<4>("v_add_u32", 1);
    run<5>("v_cmp_gt_u64", 1);
    run<6>("v_mov_b64", 1);
    run<7>("v_add_f32", 1);
    run<8>("v_cmp_gt_u32", 1);
    run<9>("add_f64+cmp_gt_f64+cndmask (3)", 3);
    run<10>("v_lshl_add_u64", 1);
    run<11>("v_min_u32", 1);
    run<12>("v_pk_add_f32", 1);
    run<13>("cmp_ge_f64+cndmask (2)", 2);
    // whole candidates (figures are per INSTRUCTION; x 7 / x 6 / x 6 for one candidate; the s_nop of the first is not counted as one)
    run<14>("Inc candidate, one chain (7)", 7);
    run<15>("Inc candidate x2, interleaved (14)", 14);
    run<16>("Dec candidate (6)", 6);
    // sparse EXEC masks: does a wave64 instruction with few enabled lanes cost less?
    run<0, 32>("v_add_f64, lanes 0-31 enabled", 1);
    run<0, 16>("v_add_f64, lanes 0-15 enabled", 1);
    run<0, 1>("v_add_f64, lane 0 enabled", 1);
    run<9, 1>("add+cmp+cndmask (3), lane 0 enabled", 3);
    run<4, 1>("v_add_u32, lane 0 enabled", 1);
    return 0;
}
