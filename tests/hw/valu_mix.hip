// valu_mix.hip -- how a SIMD prices a MIX of the two VALU issue classes (gfx950, MI355X).
//
// valu_rate.hip showed two classes in unmixed streams: v_max_f64 / v_max_i32 / SDWA / VOP3 forms at ~4.15 clocks per wave64
// instruction per SIMD, plain VOP2 add / and / or / sub at ~2.15 once >= 2 waves share the SIMD.  The alignment cell is 6 of the
// first and 4 of the second.  Question: what does a block of 10 such instructions cost, and does the ORDER inside the block
// (runs of cheap instructions, isolated ones, the cell's own order) change it?  Same method as valu_rate.hip: one workgroup per
// CU, w waves per SIMD, independent destinations, constant sources, HIP-event time x measured shader clock.
//
// Build:  hipcc --offload-arch=gfx950 -O3 tests/hw/valu_mix.hip -o tests/hw/valu_mix
// Run:    tests/hw/valu_mix
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define S(i) "v_max_i32 %" #i ", %16, %17\n\t"
#define F(i) "v_or_b32 %" #i ", %16, %17\n\t"
#define X(i) "v_add_u32_sdwa %" #i ", %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
// the "real classes" blocks: operands 0-15 = r, 16-23 = eight 64-bit pairs, 24 / 25 = a / b, 26 = a 64-bit constant
#define D(j) "v_max_f64 %" #j ", %26, %" #j "\n\t"
#define X2(i) "v_add_u32_sdwa %" #i ", %24, %25 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define F2(i) "v_or_b32 %" #i ", %24, %25\n\t"
#define A2(i) "v_and_b32 %" #i ", %24, %25\n\t"
#define B2(i) "v_sub_u32 %" #i ", %24, %25\n\t"
#define REP12(B) B B B B B B B B B B B B
#define OUTS "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])

enum { P_S, P_F, P_CELL, P_RUN4, P_ISOLATED, P_PAIRS, P_8F2S, P_5F5S, P_CELL64, P_CELL64_RUN, P_COUNT };
static const char* kName[P_COUNT] = {
    "SSSSSSSSSS  (v_max_i32 only)", "FFFFFFFFFF  (v_or_b32 only)", "SSSSFFSSFF  (the cell's order)", "SSSSSSFFFF  (one run of four)",
    "SFSFSFSFSS  (isolated)", "SSFFSSSFFS  (two pairs apart)", "FFFFFFFFSS", "FFFFFSSSSS",
    "cell, real classes: 4 v_max_f64 + 2 SDWA add + or or and sub (cell order)", "the same, the four cheap ones in one run"};

template <int P>
__global__ __launch_bounds__(1024) void k_mix(unsigned long long* __restrict__ out, int iters, int seed) {
    extern __shared__ uint32_t lds_pad[];
    uint32_t a = (uint32_t)(threadIdx.x * 2654435761u + seed), b = (uint32_t)(threadIdx.x ^ 0x9e3779b9u) + seed;
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = a + i;
    double dq[8]; const double dx = __hiloint2double(0x40000000 + (int)(a & 0xffff), (int)b);
#pragma unroll
    for (int i = 0; i < 8; ++i) dq[i] = __hiloint2double(0x40000000 + (int)((r[i] >> 4) & 0xfffff), (int)r[i + 8]);
    unsigned long long t0, t1, q0, q1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(q0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (P == P_S) asm volatile(REP12(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_F) asm volatile(REP12(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_CELL) asm volatile(REP12(S(0) S(1) S(2) S(3) F(4) F(5) S(6) S(7) F(8) F(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_RUN4) asm volatile(REP12(S(0) S(1) S(2) S(3) S(4) S(5) F(6) F(7) F(8) F(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_ISOLATED) asm volatile(REP12(S(0) F(1) S(2) F(3) S(4) F(5) S(6) F(7) S(8) S(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_PAIRS) asm volatile(REP12(S(0) S(1) F(2) F(3) S(4) S(5) S(6) F(7) F(8) S(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_8F2S) asm volatile(REP12(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) S(8) S(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_5F5S) asm volatile(REP12(F(0) F(1) F(2) F(3) F(4) S(5) S(6) S(7) S(8) S(9)) : OUTS : "v"(a), "v"(b));
        else if constexpr (P == P_CELL64)
            asm volatile(REP12(D(16) D(17) X2(0) X2(1) F2(2) F2(3) D(18) D(19) A2(4) B2(5))
                         : OUTS, "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]), "+v"(dq[4]), "+v"(dq[5]), "+v"(dq[6]), "+v"(dq[7]) : "v"(a), "v"(b), "v"(dx));
        else
            asm volatile(REP12(D(16) D(17) X2(0) X2(1) D(18) D(19) F2(2) F2(3) A2(4) B2(5))
                         : OUTS, "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]), "+v"(dq[4]), "+v"(dq[5]), "+v"(dq[6]), "+v"(dq[7]) : "v"(a), "v"(b), "v"(dx));
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1)::"memory");
    uint32_t acc = a ^ b;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= (uint32_t)__double2hiint(dq[i]) ^ (uint32_t)__double2loint(dq[i]);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = q1 - q0; }
    if (acc == 0x12345u && iters < 0) lds_pad[threadIdx.x] = acc;
}

typedef void (*kern_t)(unsigned long long*, int, int);
template <int... I> static void fill_table(kern_t* t, std::integer_sequence<int, I...>) { ((t[I] = k_mix<I>), ...); }

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    kern_t table[P_COUNT];
    fill_table(table, std::make_integer_sequence<int, P_COUNT>());
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, sizeof(unsigned long long) * cus * 2 * 16 * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int waves_per_simd[] = {1, 2, 3, 4, 5, 7, 8};
    printf("%-76s", "clocks per BLOCK OF 10 per SIMD (kernel time x shader clock)");
    for (int w : waves_per_simd) printf("   w=%d", w);
    printf("\n");
    for (int p = 0; p < P_COUNT; ++p) {
        printf("%-76s", kName[p]);
        for (int w : waves_per_simd) {
            const int block = w <= 4 ? 256 * w : (w == 8 ? 1024 : 256 * w);       // 5 and 7 waves per SIMD: one workgroup of 1,280 / 1,792 threads does not exist -> two workgroups below
            int wg_per_cu = 1, threads = 256 * w;
            if (threads > 1024) { wg_per_cu = 2; threads = (w == 8) ? 1024 : 0; }
            if (threads == 0) { printf("      -"); continue; }                   // (5, 7: skipped -- uneven workgroups would not share SIMDs evenly)
            const size_t lds = wg_per_cu == 1 ? 96 * 1024 : 64 * 1024;
            CK(hipFuncSetAttribute((const void*)table[p], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * wg_per_cu, nwaves = grid * threads / 64;
            auto run = [&](int iters) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(table[p], dim3(grid), dim3(threads), lds, 0, d_out, iters, p);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                return (double)ms;
            };
            run(500);
            const double probe_ms = run(2000);
            int iters = (int)(2000.0 * 12.0 / (probe_ms > 0.01 ? probe_ms : 0.01));
            iters = std::max(2000, std::min(iters, 2000000));
            const double ms = run(iters);
            std::vector<unsigned long long> h(2 * nwaves);
            CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 2 * nwaves, hipMemcpyDeviceToHost));
            std::vector<double> freq(nwaves);
            for (int i = 0; i < nwaves; ++i) freq[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 100.0 : 0.0;
            std::sort(freq.begin(), freq.end());
            const double blocks = (double)iters * 12.0 * w * (wg_per_cu == 2 ? 1.0 : 1.0);   // blocks per SIMD: w waves each run iters x 12
            printf(" %6.2f", ms * 1e-3 * freq[nwaves / 2] * 1e6 / blocks);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
