// valu_rate.hip -- VALU issue-rate probe for gfx950 (MI355X).
//
// Question it settles ("What's weak" 3): does a SIMD retire a wave64 integer VALU instruction
// every 4 clocks (16 lanes/clk) or every 2 (32 lanes/clk) once >= 2 waves share it, and does the answer depend
// on the instruction class the alignment kernel is made of (VOP2 / VOP3 / VOPC->SGPR / SDWA / DPP)?
//
// Method: one workgroup per CU (an LDS request > half the CU's LDS keeps a second one out), 4*w waves per
// workgroup = w waves per SIMD, every wave runs ITERS iterations of 16 independent instructions of one class
// (distinct destination registers, constant sources), bracketed by s_memtime.  Reported per class and w:
//   clk/instr/SIMD = median over waves of (cycles elapsed) / (ITERS * 16 * w)
// and the same figure from the kernel's HIP-event time * the measured shader clock, as a cross-check.
// 4.0 means 16 lanes/clk, 2.0 means 32 lanes/clk.
//
// Build:  hipcc --offload-arch=gfx950 -O3 tests/hw/valu_rate.hip -o tests/hw/valu_rate
// Run:    tests/hw/valu_rate [out.json]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define R8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define R8X(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define REP16_1(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
// 128 instructions per loop iteration: a taken branch costs a lone wave ~100 clocks of instruction fetch
#define REP16(F) REP16_1(F) REP16_1(F) REP16_1(F) REP16_1(F) REP16_1(F) REP16_1(F) REP16_1(F) REP16_1(F)

enum {
    OP_FMA_F32, OP_PK_FMA_F32, OP_MAX_I32, OP_MAX3_I32, OP_ADD_U32, OP_AND_B32, OP_BCNT, OP_CNDMASK_VCC, OP_CNDMASK_SGPR,
    OP_CMP_VCC, OP_CMP_SGPR, OP_CMP_SDWA_SGPR, OP_ADD_SDWA, OP_ADDC, OP_CNDMASK_DPP_WAVE, OP_MOV_DPP_WAVE, OP_MOV_DPP_ROW,
    OP_PK_MAX_I16, OP_ADD3_U32, OP_PERM, OP_CNDMASK_VCC_HOISTED, OP_CNDMASK_VCC_E64, OP_MIN_U32, OP_SUB_U32, OP_OR_B32, OP_LSHL_ADD, OP_XOR_B32, OP_MOV_B32, OP_LSHLREV, OP_ASHRREV, OP_MAX_F32, OP_ADD_F32, OP_MUL_U24, OP_MAX_I16, OP_ADD_U16, OP_POPC_MIX, OP_AND_MAX_MIX, OP_MAX_F64, OP_CELL64, OP_CELL, OP_STEP_VCC, OP_STEP_NOVCC, OP_COUNT
};
static const char* kOpName[OP_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_max_i32", "v_max3_i32", "v_add_u32", "v_and_b32", "v_bcnt_u32_b32", "v_cndmask_b32 (vcc)",
    "v_cndmask_b32 (sgpr pair)", "v_cmp_gt_i32 -> vcc", "v_cmp_gt_i32 -> sgpr pair", "v_cmp_eq_u32_sdwa -> sgpr pair",
    "v_add_u32_sdwa", "v_addc_co_u32 (sgpr carry)", "v_cndmask_b32_dpp wave_shr:1", "v_mov_b32_dpp wave_shr:1",
    "v_mov_b32_dpp row_shr:1", "v_pk_max_i16", "v_add3_u32", "v_perm_b32", "v_cndmask_b32 (vcc set outside the loop)", "v_cndmask_b32_e64 (vcc as sgpr operand)", "v_min_u32", "v_sub_u32", "v_or_b32", "v_lshl_add_u32", "v_xor_b32", "v_mov_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_max_f32", "v_add_f32", "v_mul_u32_u24", "v_max_i16", "v_add_u16", "popcount tile mix: 32 v_and then 32 v_bcnt (16 chains of 2)", "alternating v_and / v_max_i32", "v_max_f64", "DP cell as 64-bit lexicographic max (11 instr: 4 v_max_f64)",
    "DP cell (15 instr, pc_nw.hip schedule)", "row step: prologue (s_mov vcc + 10 VALU) + 8 cells", "row step: prologue with vcc set outside + 8 cells"};
// instructions per loop iteration
static int op_instrs(int op) { return op == OP_CELL ? 15 * 8 : op == OP_STEP_VCC || op == OP_STEP_NOVCC ? 15 * 8 + 11 : op == OP_CELL64 ? 11 * 8 : 128; }

template <int OP>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long* __restrict__ out, int iters, int seed) {
    extern __shared__ uint32_t lds_pad[];
    uint32_t a = (uint32_t)(threadIdx.x * 2654435761u + seed), b = (uint32_t)(threadIdx.x ^ 0x9e3779b9u) + seed;
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = a + i;
    unsigned long long m = __builtin_amdgcn_ballot_w64((threadIdx.x & 3) == 0);
    unsigned long long s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = m + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8]; const f2 px = {1.0f, 0.5f}, py = {0.25f, 2.0f};
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{__uint_as_float(r[2 * i] & 0x3fffffffu), __uint_as_float(r[2 * i + 1] & 0x3fffffffu)};
    double dq[8]; const double dx = __hiloint2double(0x40000000 + (int)(a & 0xffff), (int)b);
#pragma unroll
    for (int i = 0; i < 8; ++i) dq[i] = __hiloint2double(0x40000000 + (int)((r[i] >> 4) & 0xfffff), (int)r[i + 8]);
    unsigned long long t0, t1, q0, q1;
    if constexpr (OP == OP_CNDMASK_VCC_HOISTED || OP == OP_CNDMASK_VCC_E64 || OP == OP_STEP_NOVCC)
        asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");          // the loops below contain nothing that writes vcc
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(q0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == OP_FMA_F32) {
#define F(i) "v_fma_f32 %" #i ", %16, %17, %" #i "\n\t"
            asm volatile(REP16(F) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_PK_FMA_F32) {
            // 8 independent 64-bit register pairs (declared outside the loop)
#define F(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
            asm volatile(R8X(F)
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(px), "v"(py));
#undef F
        } else if constexpr (OP == OP_MAX_I32 || OP == OP_ADD_U32 || OP == OP_AND_B32 || OP == OP_BCNT || OP == OP_PK_MAX_I16) {
#define BODY(NAME) asm volatile(REP16(F) : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(a), "v"(b))
            if constexpr (OP == OP_MAX_I32) {
#define F(i) "v_max_i32 %" #i ", %16, %17\n\t"
                BODY();
#undef F
            } else if constexpr (OP == OP_ADD_U32) {
#define F(i) "v_add_u32 %" #i ", %16, %17\n\t"
                BODY();
#undef F
            } else if constexpr (OP == OP_AND_B32) {
#define F(i) "v_and_b32 %" #i ", %16, %17\n\t"
                BODY();
#undef F
            } else if constexpr (OP == OP_BCNT) {
#define F(i) "v_bcnt_u32_b32 %" #i ", %16, %17\n\t"
                BODY();
#undef F
            } else {
#define F(i) "v_pk_max_i16 %" #i ", %16, %17\n\t"
                BODY();
#undef F
            }
        } else if constexpr (OP == OP_MAX3_I32 || OP == OP_ADD3_U32 || OP == OP_PERM) {
            if constexpr (OP == OP_MAX3_I32) {
#define F(i) "v_max3_i32 %" #i ", %16, %17, %16\n\t"
                BODY();
#undef F
            } else if constexpr (OP == OP_ADD3_U32) {
#define F(i) "v_add3_u32 %" #i ", %16, %17, %16\n\t"
                BODY();
#undef F
            } else {
#define F(i) "v_perm_b32 %" #i ", %16, %17, %16\n\t"
                BODY();
#undef F
            }
        } else if constexpr (OP == OP_CNDMASK_VCC) {
#define F(i) "v_cndmask_b32 %" #i ", %16, %17, vcc\n\t"
            asm volatile("s_mov_b64 vcc, %18\n\t" REP16(F) : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(a), "v"(b), "s"(m) : "vcc");
#undef F
        } else if constexpr (OP == OP_CNDMASK_VCC_HOISTED) {
#define F(i) "v_cndmask_b32 %" #i ", %16, %17, vcc\n\t"
            asm volatile(REP16(F) : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_CNDMASK_VCC_E64) {
#define F(i) "v_cndmask_b32_e64 %" #i ", %16, %17, vcc\n\t"
            asm volatile(REP16(F) : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_MIN_U32) {
#define F(i) "v_min_u32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_SUB_U32) {
#define F(i) "v_sub_u32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_OR_B32) {
#define F(i) "v_or_b32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_LSHL_ADD) {
#define F(i) "v_lshl_add_u32 %" #i ", %16, 2, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_XOR_B32) {
#define F(i) "v_xor_b32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_LSHLREV) {
#define F(i) "v_lshlrev_b32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_ASHRREV) {
#define F(i) "v_ashrrev_i32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_MAX_F32) {
#define F(i) "v_max_f32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_ADD_F32) {
#define F(i) "v_add_f32 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_MUL_U24) {
#define F(i) "v_mul_u32_u24 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_MAX_I16) {
#define F(i) "v_max_i16 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_ADD_U16) {
#define F(i) "v_add_u16 %" #i ", %16, %17\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_MOV_B32) {
#define F(i) "v_mov_b32 %" #i ", %16\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_POPC_MIX) {
            // what k_set_popc's inner loop issues per bitmap word: 32 independent v_and, then 16 counters each fed by two
            // back-to-back v_bcnt (the second reads the first's result); twice per iteration = 128 instructions
#define ANDS(o) "v_and_b32 %" #o ", %16, %17\n\t"
#define CNT(o) "v_bcnt_u32_b32 %" #o ", %16, %" #o "\n\tv_bcnt_u32_b32 %" #o ", %17, %" #o "\n\t"
#define HALF REP16_1(ANDS) REP16_1(ANDS) REP16_1(CNT)
            asm volatile(HALF HALF : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(a), "v"(b));
#undef HALF
#undef CNT
#undef ANDS
        } else if constexpr (OP == OP_AND_MAX_MIX) {
#define F(i) "v_and_b32 %" #i ", %16, %17\n\tv_max_i32 %" #i ", %16, %17\n\t"
            asm volatile(REP16_1(F) REP16_1(F) REP16_1(F) REP16_1(F) : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_MAX_F64) {
            // 8 independent 64-bit register pairs
#define F(i) "v_max_f64 %" #i ", %8, %" #i "\n\t"
            asm volatile(R8X(F) : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]), "+v"(dq[4]), "+v"(dq[5]), "+v"(dq[6]), "+v"(dq[7]) : "v"(dx));
#undef F
        } else if constexpr (OP == OP_CELL64) {
            // candidate cell: (score*4 | tag) in the high dword, statistics in the low dword, v_max_f64 as a lexicographic max
            double HoL = dq[0], EL = dq[1];
            uint32_t Dhi = r[0], Dlo = r[1];
            const uint32_t K = 0x10000u, ac = a & 31u, bcn = b, pwn = b ^ a;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                double HoU = dq[2 + (c & 1)], FU = dq[4 + (c & 1)], E, H, T;
                asm volatile("v_max_f64 %0, %1, %2" : "=v"(E) : "v"(HoL), "v"(EL));
                asm volatile("v_max_f64 %0, %1, %0" : "+v"(FU) : "v"(HoU));
                unsigned long long c2; uint32_t Dnhi, Dnlo;
                const uint32_t HoUhi = (uint32_t)__double2hiint(HoU), HoUlo = (uint32_t)__double2loint(HoU);
                asm volatile("v_cmp_eq_u32_sdwa %0, %3, %4 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
                             "v_add_u32_sdwa %1, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                             "v_addc_co_u32 %2, %0, %7, %8, %0"
                             : "=&s"(c2), "=&v"(Dnhi), "=&v"(Dnlo) : "v"(ac), "v"(bcn), "v"(pwn), "v"(HoUhi), "v"(K), "v"(HoUlo));
                E = __hiloint2double(__double2hiint(E) | 1, __double2loint(E));
                FU = __hiloint2double(__double2hiint(FU) | 2, __double2loint(FU));
                const double D = __hiloint2double((int)Dhi, (int)Dlo);
                asm volatile("v_max_f64 %0, %1, %2" : "=v"(T) : "v"(D), "v"(FU));
                asm volatile("v_max_f64 %0, %1, %2" : "=v"(H) : "v"(T), "v"(E));
                HoU = __hiloint2double((__double2hiint(H) & -4) - 40, __double2loint(H));
                dq[2 + (c & 1)] = HoU; dq[4 + (c & 1)] = FU;
                HoL = HoU; EL = E; Dhi = Dnhi; Dlo = Dnlo;
            }
            dq[0] = HoL; dq[1] = EL; r[0] = Dhi; r[1] = Dlo;
        } else if constexpr (OP == OP_CNDMASK_SGPR) {
#define F(i) "v_cndmask_b32 %" #i ", %16, %17, %18\n\t"
            asm volatile(REP16(F) : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(a), "v"(b), "s"(m));
#undef F
        } else if constexpr (OP == OP_CMP_VCC) {
#define F(i) "v_cmp_gt_i32 vcc, %1, %2\n\t"
            asm volatile(REP16(F) "s_mov_b64 %0, vcc\n\t" : "=s"(s[0]) : "v"(a), "v"(b) : "vcc");
#undef F
        } else if constexpr (OP == OP_CMP_SGPR || OP == OP_CMP_SDWA_SGPR) {
            if constexpr (OP == OP_CMP_SGPR) {
#define F(i) "v_cmp_gt_i32 %" #i ", %8, %9\n\t"
                asm volatile(R8X(F)
                             : "=&s"(s[0]), "=&s"(s[1]), "=&s"(s[2]), "=&s"(s[3]), "=&s"(s[4]), "=&s"(s[5]), "=&s"(s[6]), "=&s"(s[7]) : "v"(a), "v"(b));
#undef F
            } else {
#define F(i) "v_cmp_eq_u32_sdwa %" #i ", %8, %9 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t"
                asm volatile(R8X(F)
                             : "=&s"(s[0]), "=&s"(s[1]), "=&s"(s[2]), "=&s"(s[3]), "=&s"(s[4]), "=&s"(s[5]), "=&s"(s[6]), "=&s"(s[7]) : "v"(a), "v"(b));
#undef F
            }
        } else if constexpr (OP == OP_ADD_SDWA) {
#define F(i) "v_add_u32_sdwa %" #i ", %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            BODY();
#undef F
        } else if constexpr (OP == OP_ADDC) {
            // carry in and out through the same SGPR pair, as the cell's stats update does; 8 pairs rotate
#define F(i, j) "v_addc_co_u32 %" #i ", %" #j ", %24, %25, %" #j "\n\t"
            #define A16 F(0, 16) F(1, 17) F(2, 18) F(3, 19) F(4, 20) F(5, 21) F(6, 22) F(7, 23) F(8, 16) F(9, 17) F(10, 18) F(11, 19) F(12, 20) F(13, 21) F(14, 22) F(15, 23)
            asm volatile(A16 A16 A16 A16 A16 A16 A16 A16
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]),
                           "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]), "+s"(s[4]), "+s"(s[5]), "+s"(s[6]), "+s"(s[7])
                         : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_CNDMASK_DPP_WAVE) {
#define F(i) "v_cndmask_b32_dpp %" #i ", %16, %17, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            asm volatile("s_mov_b64 vcc, %18\n\t" REP16(F) : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]) : "v"(a), "v"(b), "s"(m) : "vcc");
#undef F
        } else if constexpr (OP == OP_MOV_DPP_WAVE) {
#define F(i) "v_mov_b32_dpp %" #i ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            asm volatile(REP16(F) : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]) : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_MOV_DPP_ROW) {
#define F(i) "v_mov_b32_dpp %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            asm volatile(REP16(F) : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]) : "v"(a), "v"(b));
#undef F
        } else if constexpr (OP == OP_CELL || OP == OP_STEP_VCC || OP == OP_STEP_NOVCC) {
            // 8 chained DP cells exactly as pc_nw.hip schedules them (PC_CELL_BODY): the E/SE chain runs through the
            // cells, the column state is private to each cell.  The two "row step" classes put pc_nw.hip's step
            // prologue (five wave_shr:1 exchanges under vcc, flag tests, first diagonal term) in front of them.
            int Hol = (int)a, El = (int)b, D = (int)r[0]; uint32_t SHl = r[1], SEl = r[2], SD = r[3];
            const uint32_t K = 0x10000u, bcn = b, pwn = b ^ a;
            uint32_t ac = a & 31u;
            if constexpr (OP != OP_CELL) {
                uint32_t an; unsigned long long rstm, lastm, c2;
                const int v_hb = -22, v_neg = -(1 << 29); const uint32_t v_zero = 0;
#define PROLOGUE(SET_VCC)                                                                                              \
                asm volatile(                                                                                          \
                    "s_nop 1\n\t" SET_VCC                                                                              \
                    "v_cmp_eq_u32_sdwa %[c2], %[a], %[bc0] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"                         \
                    "v_cndmask_b32_dpp %[an], %[a], %[en], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
                    "v_cndmask_b32_dpp %[Hol], %[Hw], %[hb], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                    "v_cndmask_b32_dpp %[El], %[oE], %[neg], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                    "v_cndmask_b32_dpp %[SHl], %[SHw], %[zero], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                    "v_mov_b32_dpp %[SEl], %[oSE] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
                    "v_cmp_lt_u32_sdwa %[rstm], %[K], %[a] src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"                         \
                    "v_cmp_eq_u32_sdwa %[lastm], %[K], %[a] src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"                        \
                    "v_add_u32_sdwa %[D0], %[pw0], %[Hod] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t" \
                    "v_addc_co_u32 %[SD0], %[c2], %[K], %[SHd], %[c2]\n\t"                                              \
                    : [an] "=&v"(an), [Hol] "=&v"(Hol), [El] "=&v"(El), [SHl] "=&v"(SHl), [SEl] "=&v"(SEl), [D0] "=&v"(D),     \
                      [SD0] "=&v"(SD), [rstm] "=&s"(rstm), [lastm] "=&s"(lastm), [c2] "=&s"(c2)                                \
                    : [hm] "s"(m), [a] "v"(ac), [en] "v"(r[5]), [Hw] "v"(r[4]), [hb] "v"(v_hb), [oE] "v"(b), [neg] "v"(v_neg), \
                      [SHw] "v"(r[12]), [zero] "v"(v_zero), [oSE] "v"(r[2]), [K] "v"(K), [bc0] "v"(bcn), [pw0] "v"(pwn),       \
                      [Hod] "v"(r[6]), [SHd] "v"(r[13])                                                                      \
                    : "vcc")
                if constexpr (OP == OP_STEP_VCC) PROLOGUE("s_mov_b64 vcc, %[hm]\n\t");
                else PROLOGUE("");
                s[1] = rstm ^ lastm;
                ac = an & 31u;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                int Hou = (int)r[4 + (c & 3)], Fu = (int)r[8 + (c & 3)]; uint32_t SHu = r[12 + (c & 3)], SFu = r[(c & 3)];
                int E, H, Dn; uint32_t SE, T, SDn; unsigned long long c0, c1, c2, c3, c4;
                asm volatile(
                    "v_cmp_gt_i32 %[c0], %[Hol], %[El]\n\t"
                    "v_cmp_gt_i32 %[c1], %[Hou], %[Fu]\n\t"
                    "v_cmp_eq_u32_sdwa %[c2], %[ac], %[bcn] src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
                    "v_max_i32 %[E], %[Hol], %[El]\n\t"
                    "v_max_i32 %[Fu], %[Hou], %[Fu]\n\t"
                    "v_add_u32_sdwa %[Dn], %[pwn], %[Hou] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                    "v_max3_i32 %[H], %[D], %[E], %[Fu]\n\t"
                    "v_cmp_eq_u32 %[c3], %[H], %[Fu]\n\t"
                    "v_cmp_eq_u32 %[c4], %[H], %[D]\n\t"
                    "v_cndmask_b32 %[SE], %[SEl], %[SHl], %[c0]\n\t"
                    "v_cndmask_b32 %[SFu], %[SFu], %[SHu], %[c1]\n\t"
                    "v_addc_co_u32 %[SDn], %[c2], %[K], %[SHu], %[c2]\n\t"
                    "v_add_u32 %[Hou], -10, %[H]\n\t"
                    "v_cndmask_b32 %[T], %[SE], %[SFu], %[c3]\n\t"
                    "v_cndmask_b32 %[SHu], %[T], %[SD], %[c4]\n\t"
                    : [E] "=&v"(E), [SE] "=&v"(SE), [H] "=&v"(H), [T] "=&v"(T), [Dn] "=&v"(Dn), [SDn] "=&v"(SDn), [Hou] "+v"(Hou),
                      [Fu] "+v"(Fu), [SHu] "+v"(SHu), [SFu] "+v"(SFu), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4)
                    : [D] "v"(D), [SD] "v"(SD), [Hol] "v"(Hol), [El] "v"(El), [SHl] "v"(SHl), [SEl] "v"(SEl), [ac] "v"(ac), [bcn] "v"(bcn),
                      [pwn] "v"(pwn), [K] "v"(K));
                r[4 + (c & 3)] = (uint32_t)Hou; r[8 + (c & 3)] = (uint32_t)Fu; r[12 + (c & 3)] = SHu; r[(c & 3)] = SFu;
                Hol = Hou; El = E; SHl = SHu; SEl = SE; D = Dn; SD = SDn;
            }
            a = (uint32_t)Hol; b = (uint32_t)El;
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1)::"memory");
    uint32_t acc = a ^ b;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= (uint32_t)s[i] ^ __float_as_uint(p[i].x) ^ __float_as_uint(p[i].y) ^ (uint32_t)__double2hiint(dq[i]) ^ (uint32_t)__double2loint(dq[i]);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = q1 - q0; }
    if (acc == 0x12345u && iters < 0) lds_pad[threadIdx.x] = acc;            // keep everything live
}

typedef void (*kern_t)(unsigned long long*, int, int);
template <int OP> static kern_t kern() { return k_rate<OP>; }
template <int... I> static void fill_table(kern_t* t, std::integer_sequence<int, I...>) { ((t[I] = kern<I>()), ...); }

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int clock_khz = 0;
    CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    kern_t table[OP_COUNT];
    fill_table(table, std::make_integer_sequence<int, OP_COUNT>());
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, sizeof(unsigned long long) * cus * 2 * 16 * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int waves_per_simd[] = {1, 2, 4, 8};
    const double target_ms = 12.0;
    std::string json = "{\n  \"device\": \"" + std::string(prop.gcnArchName) + "\", \"compute_units\": " + std::to_string(cus) +
                       ", \"clock_khz_reported\": " + std::to_string(clock_khz) +
                       ",\n  \"unit\": \"shader clocks per wave64 instruction per SIMD (4.0 = 16 lanes/clk, 2.0 = 32 lanes/clk); each run ~12 ms; "
                       "clk_per_instr_per_simd = kernel time (HIP events) x measured shader clock / instructions per SIMD; "
                       "wave_median = the same from the median wave's own s_memtime span (waves of a workgroup do not start together, so it reads low at 4-8 waves); "
                       "shader_mhz = s_memtime ticks per s_memrealtime tick x 100 MHz\",\n  \"classes\": [\n";
    printf("%-52s %7s %7s %7s %7s   | median wave's own span        | shader MHz\n", "class  (clk / wave-instr / SIMD from kernel time)", "w=1", "w=2", "w=4", "w=8");
    for (int op = 0; op < OP_COUNT; ++op) {
        double by_clk[4], by_time[4], mhz[4];
        for (int wi = 0; wi < 4; ++wi) {
            const int w = waves_per_simd[wi];
            const int block = w <= 4 ? 256 * w : 1024;
            const int wg_per_cu = w <= 4 ? 1 : 2;
            const size_t lds = w <= 4 ? 96 * 1024 : 64 * 1024;          // 160 KB per CU: one (two) workgroup(s) fit
            CK(hipFuncSetAttribute((const void*)table[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * wg_per_cu;
            const int nwaves = grid * block / 64;
            auto run = [&](int iters) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(table[op], dim3(grid), dim3(block), lds, 0, d_out, iters, op);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                return (double)ms;
            };
            run(500);                                                   // warm-up (clocks, code)
            const double probe_ms = run(2000);
            int iters = (int)(2000.0 * target_ms / (probe_ms > 0.01 ? probe_ms : 0.01));
            if (iters < 2000) iters = 2000;
            if (iters > 2000000) iters = 2000000;
            const double ms = run(iters);
            std::vector<unsigned long long> h(2 * nwaves);
            CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 2 * nwaves, hipMemcpyDeviceToHost));
            std::vector<double> span(nwaves), freq(nwaves);
            for (int i = 0; i < nwaves; ++i) { span[i] = (double)h[2 * i]; freq[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 100.0 : 0.0; }
            std::sort(span.begin(), span.end()); std::sort(freq.begin(), freq.end());
            const double n = (double)iters * op_instrs(op) * w;
            mhz[wi] = freq[nwaves / 2];
            by_clk[wi] = span[nwaves / 2] / n;
            by_time[wi] = ms * 1e-3 * mhz[wi] * 1e6 / n;
        }
        printf("%-52s %7.2f %7.2f %7.2f %7.2f   | %6.2f %6.2f %6.2f %6.2f   | %5.0f %5.0f %5.0f %5.0f\n", kOpName[op], by_time[0], by_time[1], by_time[2], by_time[3],
               by_clk[0], by_clk[1], by_clk[2], by_clk[3], mhz[0], mhz[1], mhz[2], mhz[3]);
        fflush(stdout);
        char buf[768];
        snprintf(buf, sizeof(buf), "    {\"class\": \"%s\", \"instr_per_iter\": %d, \"clk_per_instr_per_simd\": {\"1\": %.3f, \"2\": %.3f, \"4\": %.3f, \"8\": %.3f}, "
                 "\"wave_median\": {\"1\": %.3f, \"2\": %.3f, \"4\": %.3f, \"8\": %.3f}, \"shader_mhz\": {\"1\": %.0f, \"2\": %.0f, \"4\": %.0f, \"8\": %.0f}}%s\n",
                 kOpName[op], op_instrs(op), by_time[0], by_time[1], by_time[2], by_time[3], by_clk[0], by_clk[1], by_clk[2], by_clk[3],
                 mhz[0], mhz[1], mhz[2], mhz[3], op + 1 < OP_COUNT ? "," : "");
        json += buf;
    }
    json += "  ]\n}\n";
    if (out_path) { FILE* f = fopen(out_path, "w"); if (f) { fputs(json.c_str(), f); fclose(f); } }
    return 0;
}
