// pc_nw_systolic.h -- the systolic alignment kernel (K4) and the launch wrapper its translation units instantiate.
// Included by pc_nw.hip (tie rules 0 and 1, the general kernel, everything host-side) and by pc_nw_rules.hip, which is
// compiled three times (rules 2-3, 4-5, 6-7): 8 rules x (24 widths + 21 with the profile cell) kernels take ~2 1/4 min in
// one hipcc process and ~40 s in four.
#pragma once
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "pc_common.h"
#include "../../include/phamclust_hip.h"

// NCBI BLOSUM62 over ARNDCQEGHILKMFPSTWYVBZX* (SURVEY.md section 8c); codes >= 23 use the '*' row.
static __constant__ int8_t c_b62[24][24] = {
    { 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-2,-1, 0,-4},
    {-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-1, 0,-1,-4},
    {-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3, 3, 0,-1,-4},
    {-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    { 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-3,-3,-2,-4},
    {-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2, 0, 3,-1,-4},
    {-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-1,-2,-1,-4},
    {-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3, 0, 0,-1,-4},
    {-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-3,-3,-1,-4},
    {-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-3,-1,-4},
    {-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2, 0, 1,-1,-4},
    {-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-3,-1,-1,-4},
    {-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-3,-3,-1,-4},
    {-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-2,-1,-2,-4},
    { 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2, 0, 0, 0,-4},
    { 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-1,-1, 0,-4},
    {-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-3,-2,-4},
    {-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-3,-2,-1,-4},
    { 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-3,-2,-1,-4},
    {-2,-1, 3, 4,-3, 0, 1,-1, 0,-3,-4, 0,-3,-3,-2, 0,-1,-4,-3,-3, 4, 1,-1,-4},
    {-1, 0, 0, 1,-3, 3, 4,-2, 0,-3,-3, 1,-1,-3,-1, 0,-1,-3,-2,-2, 1, 4,-1,-4},
    { 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-1,-1,-1,-4},
    {-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4, 1},
};


// ---------------------------------------------------------------------------------
// Systolic kernel (the production path): wavefront-level anti-diagonal sweep.
//
// One workgroup (4 waves) per task = one column gene + up to PC_TASK_ROWS of its row sequences; the
// waves share the column gene's substitution profile in LDS (the profile is 24*lb bytes, so
// sharing it is what keeps 4 waves per SIMD resident for long genes) and otherwise run
// independently (one barrier, after the profile is built).  Within a wave the column
// sequence b (lb residues) is cut into strips of W columns, one strip per lane; G = ceil(lb/W)
// consecutive lanes form a segment, and nseg = min(16, floor(64/G)) segments of the same b
// work side by side on different row sequences.  Lane k of a segment keeps the previous row of its W columns (Ho, F and the
// two stats) in registers and at step t processes row t-k of the segment's row stream:
// what it needs from the left neighbour -- Ho, E, stats of that lane's last column, and
// the row's residue -- arrives by DPP wave_shr:1 from the neighbour's previous step, so
// the active cells at any step form an anti-diagonal and neither LDS nor a barrier is
// involved in the recurrence.  A segment's stream is its row sequences back to back, each
// preceded by a "virtual row -1" (flag RESET: the ordinary recurrence then produces the
// boundary H(-1,j) = -(11+j) with zero stats, because the new alignment's scores start a
// base step above anything the lanes still hold: PC_BASE_STEP below), so the
// pipeline fills once per task, not once per alignment, and nothing is cleared in between.  The stream is staged through LDS
// 32 entries at a time (coalesced residue reads); the head lane of a segment only reads
// one 32-bit entry per step.  The lane holding column lb-1 emits (n_ident, aln_len) when a
// row flagged LAST leaves it.  Substitution scores come from a per-task profile in LDS:
// score bytes 4*(S(r, b_j)+12) and, for the 10-instruction cell, 16-bit statistics increments, per (residue row r,
// column j); a row's offset travels in the stream entry's high half, the lane adds its own column, and the strip is
// read a dword at a time as the cells consume it (layout and bank mapping: at `prof` in the kernel).
//
// Scores are kept with an anti-diagonal bias: every stored H, E, F of cell (i,j) carries
// + (i + j).  Because the extend cost is exactly 1 per step, both extend decrements vanish:
//   E^(i,j) = max(Ho^(i,j-1), E^(i,j-1)),  F^(i,j) = max(Ho^(i-1,j), F^(i-1,j)),  Ho^ = H^ - 10,
//   H^(i,j) = max3(Ho^(i-1,j-1) + S + 12, E^, F^); all candidates of one cell share the bias, so every
// comparison and tie-break is unchanged, and the boundaries become constants (Ho^(i,-1) = Ho^(-1,j) = -22,
// Ho^(-1,-1) = -12).  The DP cell (pc_cell64) is 10 or 11 VALU instructions, column state updated in place, the NEXT
// cell's diagonal term computed from the old column state before it is overwritten (so no register copies), and
// the one VALU-written SGPR pair read >= 2 instructions later (gfx950 needs 2 wait states there; hipcc pads
// nothing inside asm).
// ---------------------------------------------------------------------------------
// Stream entry (u32): byte 0 residue code | byte 1 flags | high half = offset of the code's profile row
// (bytes, so that the row's address is one SDWA add; dwords for W >= 48, whose tables pass 64 KB); each flag is one SDWA compare.
#define PCF_LAST 0x100
#define PCF_RESET 0x200
#define PC_MAX_SEG 16
#define PC_WIN 32                                   // stream entries staged per refill (one segment per 64-lane pass; 64 was measured: no gain)

// ---------------------------------------------------------------------------------
// The DP cell as a LEXICOGRAPHIC MAX on 64-bit words (r02; the r01 cell carried scores and statistics in separate
// registers and needed a compare + select for every statistic it moved: 15 instructions).
//
// Every DP value travels as one 64-bit word  V = (hi, lo):
//   hi = 0x40000000 + 4 * score + tag     (score = the anti-diagonal-biased score; tag in the two low bits)
//   lo = n_ident | n_diag << 16           (the path statistics, as before)
// 0x20000000 <= hi < 0x50000000 for every score this kernel can meet (-inf is -2^27), so V read as an IEEE double is a
// positive NORMAL number, and for positive doubles "greater" is the unsigned order of the 63 bits: v_max_f64 returns,
// bit for bit, the operand with the larger score -- on equal scores the one with the larger tag -- and the statistics
// ride along in the low mantissa bits for free.  v_max_f64 issues at the same 4 clocks as v_max_i32
// (profiles/valu_issue_rate.json).  The tags ARE the tie rules:
//   E = max(Ho_left [tOE], E_left [tE])          ties -> the larger tag: extend (tE > tOE) or open (tOE > tE)
//   F = max(Ho_up   [tOF], F_up   [tF])
//   H = max(D [3], F [tF], E [tE])               DIAG first, then the gap state with the larger tag
// after each max the result is re-tagged for its next use ((hi & ~3) | tag: one v_and_or_b32), and Ho = H - 10 becomes
// (hi & ~3) - 40 + tOF.  11 VALU instructions per cell (4 x v_max_f64, 2 x v_and_or, v_and, v_add, and for the next cell's
// diagonal term v_cmp_eq_sdwa, v_add_sdwa, v_addc), against r01's 15; 10 where the statistics' increment comes from a
// second profile (one more v_add_sdwa instead of v_cmp + v_addc: PC_INC16_MAX_W below).  Six of the eight tie rules are a consistent order of
// (tO, tE, tF); rules 3 and 4 ask for tOE > tE > tF > tOF resp. tOF > tF > tE > tOE, i.e. two different "open" tags, which
// costs them one more v_add per cell.
// ---------------------------------------------------------------------------------
#define PC_HI0 0x40000000                                  // hi of score 0, tag 0
#define PC_S4(x) (PC_HI0 + 4 * (x))                        // hi of (anti-diagonal-biased) score x, tag 0
#define PC_NEG4 PC_S4(-(1 << 27))                          // "-infinity": survives +95 per diagonal step for 65,535 steps

template <int RULE>
struct PcTag {                                             // the TIE-RULE TABLE of the systolic kernel (rule bits: see pc_cell above)
    static_assert(RULE >= 0 && RULE < 8, "tie rule");
    static constexpr bool ins_first = (RULE & 1) != 0, e_opens = (RULE & 2) != 0, f_opens = (RULE & 4) != 0;
    static constexpr bool cyclic = RULE == 3 || RULE == 4;
    //                              rule:   0  1  2  3  4  5  6  7
    static constexpr int kE[8]   =        { 1, 2, 0, 2, 1, 2, 0, 1 };     // tag of a stored E (gap in query, INS)
    static constexpr int kF[8]   =        { 2, 1, 2, 1, 2, 0, 1, 0 };     // tag of a stored F (gap in ref, DEL)
    static constexpr int kOE[8]  =        { 0, 0, 1, 3, 0, 1, 2, 2 };     // tag of Ho where E's max reads it
    static constexpr int kOF[8]  =        { 0, 0, 1, 0, 3, 1, 2, 2 };     // tag of Ho where F's max reads it (the stored one)
    static constexpr int tD = 3, tE = kE[RULE], tF = kF[RULE], tOE = kOE[RULE], tOF = kOF[RULE];
    static_assert((tE > tOE) != e_opens && (tF > tOF) != f_opens && (tE > tF) == ins_first && tE != tF, "tags do not spell the rule");
    static_assert(tE < tD && tF < tD && (cyclic || tOE == tOF), "tags");
};

__device__ __forceinline__ double pc_pack(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ uint32_t pc_hi(double v) { return (uint32_t)__double2hiint(v); }
__device__ __forceinline__ uint32_t pc_lo(double v) { return (uint32_t)__double2loint(v); }
__device__ __forceinline__ double pc_retag(double v, uint32_t tag) { return pc_pack((pc_hi(v) & ~3u) | tag, pc_lo(v)); }
// Re-tag the result of max(x [tag FROM], y [tag TO]) -- its tag is one of the two -- as TO.  Where one of the tags' bits contain the
// other's, that is ONE VOP2 instruction (v_or_b32 / v_and_b32 with an inline constant) instead of the VOP3 v_and_or_b32: the plain
// 32-bit logic and add instructions are the ones a SIMD retires two of per 4 clocks when several waves share it
// (profiles/valu_issue_rate.txt: ~2.2 clocks against 4.2 for v_max_f64 / SDWA / VOP3 forms; ~3.5 in the cell's mix).  Rule 0 (the
// production rule: tags O 0, E 1, F 2): both re-tags are an OR.
#define PC_RETAG_ONE_OP 1
template <int FROM, int TO>
__device__ __forceinline__ double pc_retag_from(double v) {
    if constexpr (PC_RETAG_ONE_OP && (FROM | TO) == TO) return pc_pack(pc_hi(v) | (uint32_t)TO, pc_lo(v));
    else if constexpr (PC_RETAG_ONE_OP && (FROM & TO) == TO) return pc_pack(pc_hi(v) & (~3u | (uint32_t)TO), pc_lo(v));
    else return pc_retag(v, (uint32_t)TO);
}

// One cell.  In: D (this cell's diagonal candidate, tag 3), chain values HoL [tag tOF] and EL [tE], row code ac.
// In/out (in place): column state Hou -> Ho, Fu -> F.  Out: E (chain), and for the next cell Dn = old Hou + score of the
// next cell (+ 3 - tOF, folded into the profile byte) with statistics old Hou's + 0x10000 + (ac == bcn).
// The first block is asm because of its SDWA forms and because v_cmp's SGPR result must not be read by v_addc sooner than
// two instructions later (gfx950; nothing pads inside asm): the two independent v_max_f64 sit in between.
template <int NEXT_COL, int RULE, bool INC16>
__device__ __forceinline__ void pc_cell64(double D, double HoL, double EL, double& Hou, double& Fu, double& E, double& Dn,
                                          uint32_t ac, uint32_t bcn, uint32_t pwn, uint32_t pmn, uint32_t K) {
    using T = PcTag<RULE>;
    constexpr int NEXT_BYTE = NEXT_COL < 0 ? -1 : (NEXT_COL & 3);
    if constexpr (T::cyclic) HoL = pc_pack(pc_hi(HoL) + (uint32_t)(T::tOE - T::tOF), pc_lo(HoL));
    const uint32_t ohi = pc_hi(Hou), olo = pc_lo(Hou);
    uint32_t dn_hi = 0, dn_lo = 0;
    if constexpr (NEXT_COL < 0) {
        asm("v_max_f64 %[E], %[HoL], %[EL]\n\tv_max_f64 %[Fu], %[Hou], %[Fu]" : [E] "=&v"(E), [Fu] "+v"(Fu) : [HoL] "v"(HoL), [EL] "v"(EL), [Hou] "v"(Hou));
    } else if constexpr (INC16) {
        // the statistics' increment comes from the profile too: a 16-bit entry 0x2000 + (row residue == column residue)
#define PC_CELL64_B(SEL, WSEL)                                                                                         \
    asm("v_max_f64 %[E], %[HoL], %[EL]\n\t"                                                                            \
        "v_max_f64 %[Fu], %[Hou], %[Fu]\n\t"                                                                           \
        "v_add_u32_sdwa %[dh], %[pwn], %[ohi] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:DWORD\n\t" \
        "v_add_u32_sdwa %[dl], %[pmn], %[olo] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" WSEL " src1_sel:DWORD"    \
        : [E] "=&v"(E), [Fu] "+v"(Fu), [dh] "=&v"(dn_hi), [dl] "=&v"(dn_lo)                                            \
        : [HoL] "v"(HoL), [EL] "v"(EL), [Hou] "v"(Hou), [ohi] "v"(ohi), [olo] "v"(olo), [pwn] "v"(pwn), [pmn] "v"(pmn))
        if constexpr (NEXT_BYTE == 0) PC_CELL64_B("BYTE_0", "WORD_0");
        else if constexpr (NEXT_BYTE == 1) PC_CELL64_B("BYTE_1", "WORD_1");
        else if constexpr (NEXT_BYTE == 2) PC_CELL64_B("BYTE_2", "WORD_0");
        else PC_CELL64_B("BYTE_3", "WORD_1");
#undef PC_CELL64_B
    } else {
        unsigned long long c2;
#define PC_CELL64_A(SEL)                                                                                               \
    asm("v_cmp_eq_u32_sdwa %[c2], %[ac], %[bcn] src0_sel:BYTE_0 src1_sel:" SEL "\n\t"                                  \
        "v_max_f64 %[E], %[HoL], %[EL]\n\t"                                                                            \
        "v_max_f64 %[Fu], %[Hou], %[Fu]\n\t"                                                                           \
        "v_add_u32_sdwa %[dh], %[pwn], %[ohi] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SEL " src1_sel:DWORD\n\t" \
        "v_addc_co_u32 %[dl], %[c2], %[K], %[olo], %[c2]"                                                              \
        : [E] "=&v"(E), [Fu] "+v"(Fu), [dh] "=&v"(dn_hi), [dl] "=&v"(dn_lo), [c2] "=&s"(c2)                            \
        : [HoL] "v"(HoL), [EL] "v"(EL), [Hou] "v"(Hou), [ohi] "v"(ohi), [olo] "v"(olo), [ac] "v"(ac), [bcn] "v"(bcn),  \
          [pwn] "v"(pwn), [K] "v"(K))
        if constexpr (NEXT_BYTE == 0) PC_CELL64_A("BYTE_0");
        else if constexpr (NEXT_BYTE == 1) PC_CELL64_A("BYTE_1");
        else if constexpr (NEXT_BYTE == 2) PC_CELL64_A("BYTE_2");
        else PC_CELL64_A("BYTE_3");
#undef PC_CELL64_A
    }
    E = pc_retag_from<T::tOE, T::tE>(E);             // E came from HoL [tOE] or EL [tE]
    Fu = pc_retag_from<T::tOF, T::tF>(Fu);           // F from Hou [tOF] or Fu [tF]
    double H;
    asm("v_max_f64 %0, %1, %2\n\tv_max_f64 %0, %0, %3" : "=&v"(H) : "v"(D), "v"(Fu), "v"(E));
    Hou = pc_pack((pc_hi(H) & ~3u) + (uint32_t)(T::tOF - 40), pc_lo(H));
    Dn = pc_pack(dn_hi, dn_lo);
}

typedef __attribute__((address_space(3))) const uint32_t pc_lds_u32;

// Every variant up to this W exists twice.  INC16: the statistics' increment comes from a second, 16-bit profile, 10
// instructions per cell, 3 x the LDS; the other compares residues, 11 instructions.  The launcher picks per launch
// class (pc_nw_class_inc16): the big profile pays while four waves per SIMD still fit beside it.
#define PC_INC16_MAX_W 24
#define PC_INC16_K 0x2000u                           // statistics word: n_ident | n_diag << 13 (both <= lb <= 64 * 24)

__host__ __device__ constexpr int pc_prof_rows(bool inc16) { return inc16 ? 25 : 24; }
__host__ __device__ constexpr int pc_prof_row_dwords(int W, bool inc16) {   // scores (4 per dword), then increments (2 per dword)
    return (W + 3) / 4 + (inc16 ? (W + 1) / 2 : 0);
}

template <int W, int C, int RULE, bool INC16>
struct PcRow {          // compile-time unrolled sweep over the lane's W columns
    static constexpr int NDM = INC16 ? (W + 1) / 2 : 1;
    static constexpr int ND = (W + 3) / 4;
    // `pw` / `pm`: THIS row's score bytes (four columns per register) and 16-bit increments (two per register),
    // single-buffered: a register is re-loaded from the next row's strip (LDS, at `nxt`: ND score dwords, then the
    // increments, 64 dwords apart) right after the cell that reads it last
    static __device__ __forceinline__ void run(double D, double HoL, double EL, double (&Hou)[W], double (&Fu)[W],
                                               const uint32_t (&bc)[(W + 3) / 4], uint32_t (&pw)[(W + 3) / 4], uint32_t (&pm)[NDM],
                                               pc_lds_u32* nxt, uint32_t ac, uint32_t K, double& E_out) {
        double E, Dn;
        constexpr int N = (C + 1 < W) ? C + 1 : -1;                      // the column whose diagonal term this cell prepares
        pc_cell64<N, RULE, INC16>(D, HoL, EL, Hou[C], Fu[C], E, Dn, ac, bc[N < 0 ? 0 : (N >> 2)], pw[N < 0 ? 0 : (N >> 2)], pm[(N < 0 || !INC16) ? 0 : (N >> 1)], K);
        if constexpr (N >= 0 && (((N & 1) && INC16) || (N & 3) == 3 || N == W - 1)) {
            __builtin_amdgcn_sched_barrier(0);           // load here, into registers that have just died: hoisted, the loads cost a register each
            if constexpr ((N & 3) == 3 || N == W - 1) pw[N >> 2] = nxt[(N >> 2) * 64];
            if constexpr (INC16) pm[N >> 1] = nxt[(ND + (N >> 1)) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (C + 1 < W) PcRow<W, C + 1, RULE, INC16>::run(Dn, Hou[C], E, Hou, Fu, bc, pw, pm, nxt, ac, K, E_out);
        else E_out = E;
    }
};

// Waves per workgroup, sharing one profile: 4, or 8 where the profile of a longer column gene would otherwise leave
// fewer than four waves per SIMD in the CU's 160 KB of LDS (chosen per launch class by pc_nw_class_waves; the kernel
// reads it from blockDim).  16-wave workgroups were tried for segments of 64 lanes: one workgroup per CU, 5-10 % slower
// than the residue-compare cell with its small profile, which those classes run instead.
#define PC_MIN_WAVES 4
__host__ __device__ constexpr int pc_max_waves(int W) { return W <= 24 ? 8 : 4; }
__host__ __device__ constexpr int pc_wave_lds_dwords(int nseg) { return 4 * 64 + 2 * PC_MAX_SEG + nseg * PC_WIN; }   // private LDS of a wave with nseg row streams

__device__ __forceinline__ void pc_wave_lds_sync() {        // LDS write -> read inside ONE wave (in-order LDS queue)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Statistics (lo) of column slot `want` (wave-uniform, runtime) picked with compile-time register indices only: a loop
// over c with `if (c == want)` lets the compiler keep a scratch-memory copy of the whole array up to date in the hot loop.
template <int W, int C>
struct PcPick {
    static __device__ __forceinline__ uint32_t get(const double (&Hou)[W], int want) {
        if constexpr (C + 1 < W) { const uint32_t rest = PcPick<W, C + 1>::get(Hou, want); return want == C ? pc_lo(Hou[C]) : rest; }
        else return pc_lo(Hou[C]);
    }
};

// Alignments of one row stream follow each other without any clearing of the lanes' state: each starts
// PC_BASE_STEP (in units of the high word: 4 x score) above the one before.  Within an alignment the biased scores
// span less than 4 x (11 x 4,096 + 65,535 + 4,096 + 2) = 458,756 above and ~200 below the base (lb <= 4,096 columns,
// la <= 65,535 rows: pc_upload's limit; BLOSUM62's largest entry is 11; bias 1 per anti-diagonal), so 2^20 keeps every
// value of the previous alignment below every value of the new one, and 1,022 alignments fit between 0x40000000
// and the first non-finite exponent 0x7ff00000.  A stream holds at most PC_TASK_ROWS / PC_MIN_WAVES alignments.
#define PC_BASE_STEP 0x100000u
static_assert(PC_TASK_ROWS / PC_MIN_WAVES + 1 <= 1000, "alignments per row stream must fit the score headroom");

typedef uint32_t pc_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t pc_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const pc_u32x2 pc_lds_u32x2;
typedef __attribute__((address_space(3))) const pc_u32x4 pc_lds_u32x4;

// The kernel's body for one workgroup task (a device function: the wide variants get a kernel of their own, the others are
// bundled by register tier into k_nw_systolic_tier below, which picks the body by the task's launch segment).
template <int W, int RULE, bool INC16>
__device__ __forceinline__ void pc_nw_body(const PcDev& d, const PcTask* __restrict__ tasks, const uint32_t task_index,
                                           const int32_t* __restrict__ bucket_row, const uint32_t* __restrict__ bucket_dest,
                                           uint2* __restrict__ res, const int ppos) {
    constexpr int ND = (W + 3) / 4;                 // score dwords per lane per residue row
    static_assert(!INC16 || W <= PC_INC16_MAX_W, "INC16 variants");
    constexpr int NDM = INC16 ? (W + 1) / 2 : 0;    // statistics increments from the profile (PcRow): their dwords per lane per residue row
    constexpr int RS = pc_prof_row_dwords(W, INC16);   // row stride: scores, then increments
    constexpr int ROWS = pc_prof_rows(INC16);       // residue rows: 24, + 1 for "any other byte" (scores as '*', identical to nothing)
    // A stream entry's high half is its profile row's offset: in bytes where the largest one fits 16 bits (then one SDWA add
    // makes the row's LDS address), in dwords otherwise (shift + add: W >= 48 only).  The profile cell's tables are 3 x the
    // size, so they are laid out for at most 32 lanes per segment -- at least 2 rows per 64-dword line, which keeps the byte
    // offsets; a 64-lane segment (percent-positives runs on long column genes only) uses two such half tables, lanes 0-31
    // the first, lanes 32-63 the second
    constexpr bool BYTE_OFF = ((INC16 ? (ROWS + 1) / 2 : ROWS) * RS * 256) < 65536;
    static_assert(!INC16 || BYTE_OFF, "profile-cell tables must keep byte offsets");
    // one dynamic LDS array (16-byte aligned): score table | 4 private wave regions | shared profile
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NWV = (int)(blockDim.x >> 6);         // waves of this workgroup
    const int lane = threadIdx.x & 63;
    int8_t (*tab)[24] = (int8_t(*)[24])smem;                         // [24][24]            576 B
    // The profile, shared by the workgroup's waves: residue row r, strip dword q, lane position k sit at dword
    // ((r / rpl) * RS + q) * 64 + (r % rpl) * Gl + k, with Gl = the class's lanes-per-segment bound (8..64; profile cell:
    // at most 32, see BYTE_OFF) and rpl = 64 / Gl rows per 64-dword line.  A strip dword's q-stride is 256 B, a compile-time immediate of the LDS
    // reads, a row's offset fits the 16 bits of a stream entry's high half, and a lane's bank is (r % rpl) * Gl + k whatever
    // it reads: the lanes of one segment never collide, lanes of different segments when their rows differ yet agree mod rpl
    // (measured: the conflict counters read the same as with a lane-major table, and LDS waits are ~1 % of the kernel)
    // profile bytes: 4 * (S + 12) (the bias note above, scaled to the score field of `hi`) + what turns the stored Ho's tag into DIAG's
    for (int i = threadIdx.x; i < 576; i += 64 * NWV) tab[i / 24][i % 24] = (int8_t)(4 * (c_b62[i / 24][i % 24] + 12) + (PcTag<RULE>::tD - PcTag<RULE>::tOF));

    const PcTask tk = tasks[task_index];
    const int lb = d.gene_len[tk.gene];
    const uint8_t* __restrict__ bp = d.codes + d.gene_off[tk.gene];
    const int G = (lb + W - 1) / W;                 // lanes per segment (<= 64 by variant choice)
    const int Gb = G <= 8 ? 8 : (G <= 16 ? 16 : (G <= 32 ? 32 : 64));
    const int Gl = (INC16 && Gb == 64) ? 32 : Gb, rpl = 64 / Gl;                  // lanes per (half) table, rows per line
    const int rpl_sh = 6 - (Gl == 8 ? 3 : (Gl == 16 ? 4 : (Gl == 32 ? 5 : 6)));       // log2(rpl): no division in row_part
    const uint32_t half_dw = (INC16 && Gb == 64) ? (uint32_t)(((ROWS + 1) / 2) * RS * 64) : 0u;   // where the second half table starts
    auto row_part = [&](uint32_t r) { return ((r >> rpl_sh) * (uint32_t)(RS * 64) + (r & (uint32_t)(rpl - 1)) * (uint32_t)Gl) * (BYTE_OFF ? 4u : 1u); };   // entry units (dword index where it indexes `prof`: see the build loop)
    const int nseg = min(64 / G, PC_MAX_SEG);
    const int NS = NWV * nseg;                      // row slots of the workgroup
    // LDS: score table | the waves' private regions (sized by nseg) | the shared profile
    uint32_t* wreg = smem + 144 + wv * pc_wave_lds_dwords(nseg);
    uint32_t* row_la = wreg;                                         // [64] this wave's rows: length
    uint32_t* row_pos = row_la + 64;                                 // [64] start of the row's record in its segment's stream
    uint32_t* row_lo = row_pos + 64;                                 // [64] code offset, low / high dword
    uint32_t* row_hi = row_lo + 64;
    uint32_t* seg_len = row_hi + 64;                                 // [16] stream length per segment
    uint32_t* seg_cur = seg_len + PC_MAX_SEG;                        // [16] local row whose record holds the window start
    uint32_t* ring = seg_cur + PC_MAX_SEG;                           // [nseg][PC_WIN] staged stream entries
    uint32_t* prof = smem + 144 + NWV * pc_wave_lds_dwords(nseg);
    const int seg = lane / G, k = lane - seg * G;
    const bool in_seg = seg < nseg;
    const bool is_head = in_seg && k == 0;
    const int k_out = (lb - 1) / W, c_out = (lb - 1) - k_out * W;
    const bool is_out = in_seg && k == k_out;
    const uint32_t kcol = (uint32_t)(k < Gl ? k : k - Gl) + (k < Gl ? 0u : half_dw);   // my column of the profile (second half table for lanes >= 32 of a 64-lane segment)
    const int R = tk.end - tk.begin;                // rows (alignments) of this workgroup task, <= PC_TASK_ROWS
    // wave-local row lr <-> task row (lr / nseg) * NS + wv * nseg + lr % nseg  (monotone in lr)
    auto task_row = [&](int lr) { return (lr / nseg) * NS + wv * nseg + (lr % nseg); };

    uint32_t bc[ND];                                 // my W column codes, 4 per register (compared with SDWA byte selects)
#pragma unroll
    for (int q = 0; q < ND; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = k * W + q * 4 + e;
            const uint32_t code = (q * 4 + e < W && in_seg && j < lb) ? (uint32_t)bp[j] : (uint32_t)PC_PADCODE;
            v |= code << (8 * e);
        }
        bc[q] = v;
    }
    __syncthreads();                                 // score table visible
    if (in_seg) {                                    // every segment of every wave writes its share of the profile rows (all segments hold
#pragma unroll 1                                     // the same column codes): 25 rows over NWV * nseg workers of G lanes
        for (int r = wv * nseg + seg; r < ROWS; r += NS) {
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                uint32_t v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = q * 4 + e;
                    if (c < W) v |= (uint32_t)(uint8_t)tab[min(r, 23)][min((int)((bc[q] >> (8 * e)) & 0xffu), 23)] << (8 * e);
                }
                prof[row_part(r) / (BYTE_OFF ? 4u : 1u) + q * 64 + kcol] = v;
            }
            if constexpr (INC16) {
                // a column whose residue is "another byte" (code >= 24) never gets here: the host sends such column genes
                // to the general kernel, because row 24 cannot tell which other byte the row residue is
#pragma unroll
                for (int q = 0; q < NDM; ++q) {
                    uint32_t v = 0;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = q * 2 + e;
                        // ppos (metrics.py:218-220): "positive" columns count too -- matrix score > 0 (read back from the LDS score
                        // table: 4 * (S + 12) + tag shift), which for two bytes outside the alphabet ('*' against '*': +1) also
                        // covers their identity, so this table is exact for any byte
                        const int bcode = (int)((bc[c >> 2] >> (8 * (c & 3))) & 0xffu);
                        if (c < W) v |= (PC_INC16_K + (uint32_t)((r < 24 && bcode == r) || (ppos && (int)tab[min(r, 23)][min(bcode, 23)] > 48 + (PcTag<RULE>::tD - PcTag<RULE>::tOF)))) << (16 * e);
                    }
                    prof[row_part(r) / (BYTE_OFF ? 4u : 1u) + (ND + q) * 64 + kcol] = v;
                }
            }
        }
    }
    const int my_r = task_row(lane);
    const int Rw = __popcll(__builtin_amdgcn_ballot_w64(my_r < R));   // this wave's rows (a prefix of lr)
    if (lane < Rw) {
        const int ga = bucket_row[tk.begin + my_r];
        const unsigned long long off = (unsigned long long)d.gene_off[ga];
        row_la[lane] = (uint32_t)d.gene_len[ga]; row_lo[lane] = (uint32_t)off; row_hi[lane] = (uint32_t)(off >> 32);
    }
    __syncthreads();                                 // profile complete; the only workgroup barrier
    if (Rw == 0) return;
    if (lane < Rw) {                                 // record start = sum of (la+1) of the earlier rows of my segment
        uint32_t acc = 0;
        for (int q = lane % nseg; q < lane; q += nseg) acc += row_la[q] + 1;
        row_pos[lane] = acc;
        if (lane + nseg >= Rw) seg_len[lane % nseg] = acc + row_la[lane] + 1;
    }
    if (lane < PC_MAX_SEG) { seg_cur[lane] = lane; if (lane >= Rw) seg_len[lane] = 0; }
    pc_wave_lds_sync();
    int T = 0;
    for (int s2 = 0; s2 < nseg; ++s2) T = max(T, (int)seg_len[s2]);
    T = __builtin_amdgcn_readfirstlane(T) + G - 1;

    using TG = PcTag<RULE>;
    double Hou[W], Fu[W];                            // previous row of my W columns: Ho [tag tOF] and F [tag tF], statistics in the low halves
#pragma unroll
    for (int c = 0; c < W; ++c) { Hou[c] = pc_pack((uint32_t)(PC_NEG4 + TG::tOF), 0u); Fu[c] = pc_pack((uint32_t)(PC_NEG4 + TG::tF), 0u); }
    double o_E = pc_pack((uint32_t)(PC_NEG4 + TG::tE), 0u);        // my last column's E of the previous step
    double p_HoL = pc_pack((uint32_t)(PC_NEG4 + TG::tOF), 0u);     // what I received last step (diagonal of column 0)
    int out_r = seg;                                 // out lane: local row of the next result ...
    int out_tr = wv * nseg + seg;                    // ... and its row of the task (= task_row(out_r), kept without the division)
    const uint32_t K = 0x10000u;
    const int half = lane / PC_WIN, hl = lane % PC_WIN;               // refill: which of the pass's segments, which entry
    // LDS byte address of my column of the profile
    const uint32_t prof_lane = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)prof + (in_seg ? kcol : 0u) * 4u;
    const uint32_t ring_lane = (uint32_t)(in_seg ? seg : 0) * PC_WIN;

    // Stage PC_WIN stream entries of every segment starting at stream position `base` (64 / PC_WIN segments per pass).
    auto refill = [&](int base) {
        pc_wave_lds_sync();
        for (int s0 = 0; s0 < nseg; s0 += 64 / PC_WIN) {
            const int sg = s0 + half;
            uint32_t entry = 0;
            if (sg < nseg) {
                const uint32_t p = (uint32_t)base + hl;
                if (p < seg_len[sg]) {
                    int r = (int)seg_cur[sg];
                    while (p >= row_pos[r] + row_la[r] + 1) r += nseg;
                    const int i = (int)(p - row_pos[r]) - 1;
                    if (i < 0) entry = PCF_RESET;
                    else {
                        const uint8_t* ap = d.codes + (((unsigned long long)row_hi[r] << 32) | row_lo[r]);
                        const uint32_t code = ap[i];
                        entry = code | (i == (int)row_la[r] - 1 ? PCF_LAST : 0) | (row_part(min(code, (uint32_t)(ROWS - 1))) << 16);
                    }
                    if (hl == PC_WIN - 1) seg_cur[sg] = (uint32_t)r;
                }
                ring[sg * PC_WIN + hl] = entry;
            }
        }
        pc_wave_lds_sync();
    };
    auto row_addr = [&](uint32_t entry) -> uint32_t {               // LDS address of my strip of the entry's profile row
        if constexpr (BYTE_OFF) {
            uint32_t addr;                                          // one instruction: the entry's high half + my column
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(entry), "v"(prof_lane));
            return addr;
        } else return (entry >> 14) + prof_lane;                    // the high half counts dwords; bits 14, 15 (flags) are zero
    };

    // Software pipeline: at step t the row code `a` and its profile strip `pw` (`pm`) are already in registers; the
    // code of step t+1 (head: ring entry, others: the left neighbour's current code) arrives in the step's prologue and
    // its strip is fetched register by register while the cells of step t execute (PcRow); the head's ring entry of
    // step t+2 is read one step ahead of that.
    refill(0);
    uint32_t a = is_head ? ring[ring_lane] : 0u;
    uint32_t e_nxt = ring[ring_lane + 1];                          // head's entry for step 1 (PC_WIN >= 2)
    uint32_t e_b = 0;                                              // entry t+3 (entries t+2, t+3 are fetched as a pair on even steps)
    uint32_t pw[ND], pm[INC16 ? NDM : 1];                          // this row's score bytes and statistics increments
    {
        pc_lds_u32* r0 = (pc_lds_u32*)(size_t)row_addr(a);
#pragma unroll
        for (int q = 0; q < ND; ++q) pw[q] = r0[q * 64];
        if constexpr (INC16) {
#pragma unroll
            for (int q = 0; q < NDM; ++q) pm[q] = r0[(ND + q) * 64];
        } else pm[0] = 0;
    }
    const unsigned long long headm = __builtin_amdgcn_ballot_w64(is_head), outm = __builtin_amdgcn_ballot_w64(is_out), headoutm = headm | outm;
    // boundary values the head lanes take (VGPR operands): E = -inf, statistics 0, the base step
    const uint32_t v_nege = (uint32_t)(PC_NEG4 + TG::tE), v_zero = 0, v_base_step = PC_BASE_STEP;
    // A head lane's boundary values sit on the base of the alignment its stream is in: Ho^(i,-1) = -22 and Ho^(-1,-1) = -12
    uint32_t v_hb = (uint32_t)(PC_S4(-22) + TG::tOF), v_h00 = (uint32_t)(PC_S4(-12) + TG::tOF);

    // One row step.  `a` is this step's stream entry, `a_nxt` receives the next step's.
    auto step = [&](int t, auto even_tag, uint32_t a, uint32_t& a_nxt) {
        constexpr bool even = decltype(even_tag)::value;
        if (even && ((t + 2) & (PC_WIN - 1)) == 0) refill(t + 2);
        // Step prologue, 9 VALU instructions.  The five neighbour exchanges are v_cndmask_b32_dpp: lane k takes lane
        // k-1's value (DPP wave_shr:1 on src0, executed with every lane active), head lanes (vcc) take src1 = their
        // boundary value instead: the next entry from the ring, Ho^(i,-1) = -22, E = -inf, stats 0.  Ho and E are 64-bit
        // words now, so the four value exchanges are their two halves each -- the same count as the r01 kernel's
        // (Ho, E, SH, SE).  The flag tests and the first cell's diagonal term use SDWA byte selects on the raw entry.
        // K.BYTE_2 == 1.
        uint32_t HoL_hi, HoL_lo, EL_hi, EL_lo, D0_hi, D0_lo;
        unsigned long long anym;
        if constexpr (INC16) {
            asm volatile(
                "s_nop 1\n\t"                                               // VALU (previous step's cells) -> DPP read: 2 wait states
                "s_mov_b64 vcc, %[hm]\n\t"
                "v_cndmask_b32_dpp %[an], %[a], %[en], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_dpp %[Hh], %[Hwh], %[hb], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_dpp %[Eh], %[oEh], %[neg], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_dpp %[Hl], %[Hwl], %[zero], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp %[El], %[oEl] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    // head: E = -inf never wins, its statistics are never read
                "v_cmp_ne_u32_sdwa %[anym], %[a], %[zero] src0_sel:BYTE_1 src1_sel:DWORD\n\t"                    // any flag (LAST, RESET) on my entry
                "v_add_u32_sdwa %[D0h], %[pw0], %[Hodh] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                "v_add_u32_sdwa %[D0l], %[pm0], %[Hodl] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
                "s_and_b64 %[anym], %[anym], %[hom]\n\t"                    // ... in a head lane or in the lane that holds column lb-1 (scalar: the compiler would do this AND on the VALU)
                : [an] "=&v"(a_nxt), [Hh] "=&v"(HoL_hi), [Eh] "=&v"(EL_hi), [Hl] "=&v"(HoL_lo), [El] "=&v"(EL_lo), [D0h] "=&v"(D0_hi),
                  [D0l] "=&v"(D0_lo), [anym] "=&s"(anym)
                : [hm] "s"(headm), [hom] "s"(headoutm), [a] "v"(a), [en] "v"(e_nxt), [Hwh] "v"(pc_hi(Hou[W - 1])), [hb] "v"(v_hb), [oEh] "v"(pc_hi(o_E)), [neg] "v"(v_nege),
                  [Hwl] "v"(pc_lo(Hou[W - 1])), [zero] "v"(v_zero), [oEl] "v"(pc_lo(o_E)), [K] "v"(K), [pw0] "v"(pw[0]), [pm0] "v"(pm[0]),
                  [Hodh] "v"(pc_hi(p_HoL)), [Hodl] "v"(pc_lo(p_HoL))
                : "vcc", "scc");
        } else {
            unsigned long long c2;
            asm volatile(
                "s_nop 1\n\t"
                "s_mov_b64 vcc, %[hm]\n\t"
                "v_cmp_eq_u32_sdwa %[c2], %[a], %[bc0] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                "v_cndmask_b32_dpp %[an], %[a], %[en], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_dpp %[Hh], %[Hwh], %[hb], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_dpp %[Eh], %[oEh], %[neg], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_dpp %[Hl], %[Hwl], %[zero], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_mov_b32_dpp %[El], %[oEl] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cmp_ne_u32_sdwa %[anym], %[a], %[zero] src0_sel:BYTE_1 src1_sel:DWORD\n\t"                    // any flag (LAST, RESET) on my entry
                "v_add_u32_sdwa %[D0h], %[pw0], %[Hodh] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                "v_addc_co_u32 %[D0l], %[c2], %[K], %[Hodl], %[c2]\n\t"
                "s_and_b64 %[anym], %[anym], %[hom]\n\t"
                : [an] "=&v"(a_nxt), [Hh] "=&v"(HoL_hi), [Eh] "=&v"(EL_hi), [Hl] "=&v"(HoL_lo), [El] "=&v"(EL_lo), [D0h] "=&v"(D0_hi),
                  [D0l] "=&v"(D0_lo), [anym] "=&s"(anym), [c2] "=&s"(c2)
                : [hm] "s"(headm), [hom] "s"(headoutm), [a] "v"(a), [en] "v"(e_nxt), [Hwh] "v"(pc_hi(Hou[W - 1])), [hb] "v"(v_hb), [oEh] "v"(pc_hi(o_E)), [neg] "v"(v_nege),
                  [Hwl] "v"(pc_lo(Hou[W - 1])), [zero] "v"(v_zero), [oEl] "v"(pc_lo(o_E)), [K] "v"(K), [bc0] "v"(bc[0]), [pw0] "v"(pw[0]),
                  [Hodh] "v"(pc_hi(p_HoL)), [Hodl] "v"(pc_lo(p_HoL))
                : "vcc", "scc");
        }
        const uint32_t nxt_addr = row_addr(a_nxt);                        // the next row's strip
        pc_lds_u32* nxt = (pc_lds_u32*)(size_t)nxt_addr;
        if (even) {                                                       // the head's entries for steps t+2 and t+3
            const uint2 e2 = *(const uint2*)&ring[ring_lane + ((t + 2) & (PC_WIN - 1))];
            e_nxt = e2.x; e_b = e2.y;
        } else e_nxt = e_b;
        // Flags are rare (two entries per row, and only a head lane or the output lane acts on them): one compare per step,
        // the two that tell them apart only when it fires
        unsigned long long rstm = 0, lastm = 0;
        asm volatile("" : "+s"(anym));                                    // scalar test (left alone, the compiler carries it as a lane mask: one VALU compare per step)
        if (anym != 0)
            asm volatile("v_cmp_lt_u32_sdwa %0, %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"       // RESET: flag byte > 1 (K.BYTE_2 == 1)
                         "v_cmp_eq_u32_sdwa %1, %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"       // LAST:  flag byte == 1
                         "s_and_b64 %0, %0, %4\n\t"                                              // head lanes whose stream starts an alignment
                         "s_and_b64 %1, %1, %5"                                                  // rows ending in the lane that holds column lb-1
                         : "=&s"(rstm), "=&s"(lastm) : "v"(K), "v"(a), "s"(headm), "s"(outm) : "scc");
        if (rstm != 0) {                                                  // a stream starts an alignment this step (virtual row -1)
            // Nothing is cleared.  The new alignment's scores sit PC_BASE_STEP above the previous one's (only the
            // path statistics leave the kernel, never a score), so whatever the lanes still hold of the previous
            // alignment -- Hou, Fu, the diagonal term -- loses every max from here on, exactly as -inf would.
            uint32_t inc;
            asm volatile("v_cndmask_b32 %0, %4, %5, %6\n\tv_add_u32 %1, %1, %0\n\tv_add_u32 %2, %2, %0\n\tv_cndmask_b32 %3, %3, %2, %6"
                         : "=&v"(inc), "+v"(v_hb), "+v"(v_h00), "+v"(HoL_hi) : "v"(v_zero), "v"(v_base_step), "s"(rstm));
        }
        const double HoL = pc_pack(HoL_hi, HoL_lo);
        p_HoL = HoL;
        PcRow<W, 0, RULE, INC16>::run(pc_pack(D0_hi, D0_lo), HoL, pc_pack(EL_hi, EL_lo), Hou, Fu, bc, pw, pm, nxt, a, K, o_E);
        asm volatile("" : "+s"(lastm));                                   // test here, not 140 instructions earlier (the compiler would carry the result as a lane mask: one VALU compare)
        if (lastm != 0) {                                                 // a row's last cell left the lane holding column lb-1
            asm volatile("" ::: "memory");                                // keep this wave-uniform (scalar) test a branch of its own
            if ((a & PCF_LAST) && is_out) {
                const uint32_t st = PcPick<W, 0>::get(Hou, c_out);
                const uint32_t n_ident = INC16 ? (st & (PC_INC16_K - 1)) : (st & 0xffffu), n_diag = INC16 ? (st >> 13) : (st >> 16);
                res[bucket_dest ? bucket_dest[tk.begin + out_tr] : (uint32_t)(tk.begin + out_tr)] = make_uint2(n_ident, row_la[out_r] + (uint32_t)lb - n_diag);
                out_r += nseg; out_tr += NS;
            }
        }
    };
    // two steps per iteration with the two stream-entry registers swapping roles: no copies.  An odd T runs one
    // extra step past the end of every stream (idle entries: no flags, no output).
    uint32_t a2 = 0;
#pragma unroll 1
    for (int t = 0; t < T; t += 2) {
        step(t, std::true_type{}, a, a2);
        step(t + 1, std::false_type{}, a2, a);
    }
}



// ---------------------------------------------------------------------------------
// Strip-mined passes for column genes LONGER than a variant's 64 x W columns (r04; the reference's aligner has no length cliff,
// metrics.py:160-175 -- r01-r03 sent such genes to a one-lane-per-alignment kernel at ~1/30 of the systolic rate).
//
// The column gene is cut into passes of 64 x W columns.  A wave aligns ONE row sequence per task against pass after pass: in
// every pass but the last, the lane holding the pass's last column writes what its right-hand neighbour would have received --
// (Ho, E) of that column, both 64-bit words, statistics included -- to a line in HBM, one 16-byte entry per row step; in the next
// pass the head lane takes its left-hand boundary from that line instead of the constants -22 / -inf / 0.  The virtual row -1 is
// an ordinary stream entry, so its boundary travels the same way, and the anti-diagonal bias and the base of the alignment ride
// inside the values.  An entry is written at step t for stream position t - 63 and read back at position t: a pass never
// overwrites what it has yet to read, so ONE line per wave serves all passes.  The line is staged 32 entries at a time by the
// idle half of the refill (agent-scope loads: the same wave wrote the line in the pass before).
// One alignment per stream: nothing follows it, so the base never steps (an alignment against 65,535 columns spans 3.4 M in
// `hi`, more than PC_BASE_STEP), and the column state is re-initialised in every pass -- what the lanes hold from the previous
// pass belongs to the SAME alignment and would not lose the maxima.
// Persistent grid (pc_launch_nw sizes it by the scratch it has): a workgroup of NWV waves takes every gridDim-th task, one row per
// wave; all waves meet at two barriers per pass (the profile of the pass's columns is shared).  The width is the LAUNCH's choice, not
// the column gene's -- any W covers any length -- so tasks of one or two rows run on narrow variants in one- / two-wave workgroups
// (PC_STRIP_W_ONE_ROW / _TWO_ROWS below), and so do such tasks of the wide variants' ordinary classes (1,537 ... 4,096 columns).
// Limits: statistics are 16-bit fields (compare cell: la, lb <= 65,535) or 13-bit ones (profile cell, used for percent-positives:
// lb <= 8,191).
// ---------------------------------------------------------------------------------
#define PC_STRIP_WAVES 4                                           // most waves per workgroup = rows per task (one row per wave)
#define PC_STRIP_WIN 32                                            // stream entries a strip wave stages per refill (the idle half of the wave stages the boundary line: fixed)
#define PC_STRIP_BND 64                                            // boundary entries a wave keeps staged (two refill windows)
__host__ __device__ constexpr int pc_strip_wave_lds_dwords() { return 16 + PC_STRIP_WIN + 4 * PC_STRIP_BND; }

// PIPE (r04, second form): the passes of ONE alignment spread over the workgroup's waves.  Wave w takes passes w, w + NWV, ... of the
// task's rows, one row after the other; each wave builds the profile of its own pass (private LDS), writes its boundary line as
// before, and the wave with the next pass reads it WHILE it is being written, 95+ row steps behind: every 32 steps a wave publishes
// how far it is (one LDS word per wave: passes done << 17 | steps done, after a release fence), and a wave stages 32 boundary
// entries only when the wave before it is 95 steps past them.  A wave never waits for the wave behind it (line reuse is safe:
// pass p + NWV follows pass p + NWV - 1 by 64 steps, and so on down to pass p + 1, the reader of pass p's line), so the chain of
// waits ends at pass 0 and the grid drains.  What it buys is LATENCY: a 6,600 x 6,600 alignment takes one wave 47 ms -- alone on
// its SIMD it issues an instruction every ~12 clocks -- and a collection's few such alignments were the critical path of small
// fills (synth_real(1000): 58 ms for 4.0e10 cells); over eight waves it takes an eighth.  Launches with many tasks keep the
// one-row-per-wave form above, which holds more waves per CU.
#define PC_PIPE_WAVES_MAX 8
template <int W, int RULE, bool INC16, bool PIPE = false>
__global__ __launch_bounds__(64 * (PIPE ? PC_PIPE_WAVES_MAX : PC_STRIP_WAVES), (W == 48 ? 2 : 1)) void k_nw_strip(PcDev d, const PcTask* __restrict__ tasks, int ntasks,
                                                                          const int32_t* __restrict__ bucket_row,
                                                                          const uint32_t* __restrict__ bucket_dest, uint2* __restrict__ res, int ppos,
                                                                          uint4* __restrict__ spill, unsigned spill_stride) {
    constexpr int ND = (W + 3) / 4;
    constexpr int NDM = INC16 ? (W + 1) / 2 : 0;
    constexpr int RS = pc_prof_row_dwords(W, INC16);
    constexpr int ROWS = pc_prof_rows(INC16);
    constexpr bool BYTE_OFF = ((INC16 ? (ROWS + 1) / 2 : ROWS) * RS * 256) < 65536;
    static_assert(!INC16 || BYTE_OFF, "profile-cell tables must keep byte offsets");
    constexpr int COLS = 64 * W;                                   // columns of a full pass
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NWV = (int)(blockDim.x >> 6);                        // waves of this workgroup: 4, or 2 / 1 for tasks of two rows / one row (narrow W)
    const int lane = threadIdx.x & 63;
    int8_t (*tab)[24] = (int8_t(*)[24])smem;
    for (int i = threadIdx.x; i < 576; i += 64 * NWV) tab[i / 24][i % 24] = (int8_t)(4 * (c_b62[i / 24][i % 24] + 12) + (PcTag<RULE>::tD - PcTag<RULE>::tOF));
    // the layout of the widest lanes-per-segment bucket, whatever a pass's own width: one (profile cell: two half) table(s) of 64 lanes
    constexpr int Gl = INC16 ? 32 : 64;
    constexpr uint32_t half_dw = INC16 ? (uint32_t)(((ROWS + 1) / 2) * RS * 64) : 0u;
    auto row_part = [&](uint32_t r) { return (INC16 ? ((r >> 1) * (uint32_t)(RS * 64) + (r & 1u) * 32u) : r * (uint32_t)(RS * 64)) * (BYTE_OFF ? 4u : 1u); };
    uint32_t* wreg = smem + 144 + wv * pc_strip_wave_lds_dwords();
    uint32_t* ring = wreg + 16;                                    // [PC_STRIP_WIN] staged stream entries
    uint32_t* bnd = ring + PC_STRIP_WIN;                                 // [PC_STRIP_BND][4] staged boundary entries (Ho.hi, Ho.lo, E.hi, E.lo)
    constexpr int PROF_DW = (INC16 ? 2 * ((ROWS + 1) / 2) : ROWS) * RS * 64;     // one profile (PIPE: one per wave)
    uint32_t* prof = smem + 144 + NWV * pc_strip_wave_lds_dwords() + (PIPE ? wv * PROF_DW : 0);
    // PIPE: word 0 of a wave's region = its progress (passes done << 17 | row steps done of the pass it is in).  Written with a
    // workgroup-scope RELEASE store and read with an ACQUIRE load (r05): the order "boundary entries, then the word that announces
    // them" is stated in the memory model, not only in the instruction stream.  What the model does not promise at that scope is that
    // the entries have reached the L2 the reader's agent-scope loads are served from -- both waves sit on one CU, its L1 is write-through
    // -- so the explicit `s_waitcnt vmcnt(0)` ahead of every publication stays (vmcnt counts stores on gfx9), and
    // tools/check_pipe_publication.py holds the compiled kernels to it: in every k_nw_strip<..., PIPE> the last vector-memory wait
    // before a progress store is vmcnt(0).  (An agent-scope release would write the L2 back, every 32 steps.)
    typedef __attribute__((address_space(3))) uint32_t pc_lds_w32;           // (LDS-typed: ds_write / ds_read, not flat accesses)
    pc_lds_w32* const prog_me = (pc_lds_w32*)(size_t)(pc_lds_w32*)wreg;
    pc_lds_w32* const prog_prev = (pc_lds_w32*)(size_t)(pc_lds_w32*)(smem + 144 + ((wv + NWV - 1) % NWV) * pc_strip_wave_lds_dwords());
    auto publish = [&](uint32_t value) {
        asm volatile("s_waitcnt vmcnt(0) ; pc_publish: boundary entries before the progress word" ::: "memory");
        if (lane == 0) __hip_atomic_store(prog_me, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const uint32_t kcol = (uint32_t)(lane < Gl ? lane : lane - Gl) + (lane < Gl ? 0u : half_dw);
    uint4* const line = spill + ((size_t)blockIdx.x * (size_t)NWV + (size_t)wv) * spill_stride;
    // the line my left-hand boundary comes from: my own (the pass before was mine), PIPE: that of the wave with the pass before mine
    const uint4* const line_in = PIPE ? spill + ((size_t)blockIdx.x * (size_t)NWV + (size_t)((wv + NWV - 1) % NWV)) * spill_stride : line;
    using TG = PcTag<RULE>;
    const uint32_t K = 0x10000u;
    const uint32_t prof_lane = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)prof + kcol * 4u;
    const uint32_t v_nege = (uint32_t)(PC_NEG4 + TG::tE), v_zero = 0;

    // a workgroup's unit of work: a task (one row per wave), PIPE: one ROW of a task (rows sub, sub + 4, ... of it: strip-mined tasks
    // hold at most PC_STRIP_WAVES rows, so normally one) -- every alignment gets a pipeline of its own
    constexpr int SUB = PIPE ? PC_STRIP_WAVES : 1;
#pragma unroll 1
    for (int unit = blockIdx.x; unit < ntasks * SUB; unit += gridDim.x) {
        const int task = unit / SUB, sub = unit % SUB;
        const PcTask tk = tasks[task];
        const int lb_all = d.gene_len[tk.gene];
        const uint8_t* __restrict__ bp_all = d.codes + d.gene_off[tk.gene];
        const int npass = (lb_all + COLS - 1) / COLS;
        const int R = tk.end - tk.begin;                           // normally <= NWV rows: one round of one row per wave
#pragma unroll 1
        for (int r0 = (PIPE ? sub : 0); r0 < R; r0 += (PIPE ? SUB : NWV)) {
        const int my_row = PIPE ? r0 : r0 + wv;                    // PIPE: every wave works on the same row
        const bool have_row = my_row < R;
        int la = 0; const uint8_t* ap = d.codes;
        if (have_row) { const int ga = bucket_row[tk.begin + my_row]; la = d.gene_len[ga]; ap = d.codes + d.gene_off[ga]; }
        const int seg_len = have_row ? la + 1 : 0;                 // the virtual row -1, then the la residues
        if constexpr (PIPE) {
            __syncthreads();                                       // score table visible; every wave is done with the row before (its lines, its progress word)
            if (lane == 0) __hip_atomic_store(prog_me, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
        }
#pragma unroll 1
        for (int pass = (PIPE ? wv : 0); pass < npass; pass += (PIPE ? NWV : 1)) {
            const int col0 = pass * COLS, lb = min(COLS, lb_all - col0);
            const uint8_t* __restrict__ bp = bp_all + col0;
            const int G = (lb + W - 1) / W;                        // lanes of this pass (64 in every pass but the last)
            const bool in_seg = lane < G, is_head = lane == 0;
            const int k_out = (lb - 1) / W, c_out = (lb - 1) - k_out * W;
            const bool last_pass = pass == npass - 1;
            const bool is_out = last_pass && lane == k_out;
            uint32_t bc[ND];
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                uint32_t v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = lane * W + q * 4 + e;
                    const uint32_t code = (q * 4 + e < W && in_seg && j < lb) ? (uint32_t)bp[j] : (uint32_t)PC_PADCODE;
                    v |= code << (8 * e);
                }
                bc[q] = v;
            }
            if constexpr (!PIPE) __syncthreads();                  // score table visible; every wave is done with the previous pass's profile
            if (in_seg) {
#pragma unroll 1
                for (int r = (PIPE ? 0 : wv); r < ROWS; r += (PIPE ? 1 : NWV)) {
#pragma unroll
                    for (int q = 0; q < ND; ++q) {
                        uint32_t v = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = q * 4 + e;
                            if (c < W) v |= (uint32_t)(uint8_t)tab[min(r, 23)][min((int)((bc[q] >> (8 * e)) & 0xffu), 23)] << (8 * e);
                        }
                        prof[row_part(r) / (BYTE_OFF ? 4u : 1u) + q * 64 + kcol] = v;
                    }
                    if constexpr (INC16) {
#pragma unroll
                        for (int q = 0; q < NDM; ++q) {
                            uint32_t v = 0;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int c = q * 2 + e;
                                const int bcode = (int)((bc[c >> 2] >> (8 * (c & 3))) & 0xffu);
                                if (c < W) v |= (PC_INC16_K + (uint32_t)((r < 24 && bcode == r) || (ppos && (int)tab[min(r, 23)][min(bcode, 23)] > 48 + (PcTag<RULE>::tD - PcTag<RULE>::tOF)))) << (16 * e);
                            }
                            prof[row_part(r) / (BYTE_OFF ? 4u : 1u) + (ND + q) * 64 + kcol] = v;
                        }
                    }
                }
            }
            if constexpr (PIPE) pc_wave_lds_sync(); else __syncthreads();   // profile of this pass complete
            const int T = have_row ? seg_len + G - 1 : 0;
            const uint32_t item = (uint32_t)(pass / NWV);          // PIPE: how many passes this wave has done before this one
            double Hou[W], Fu[W];
#pragma unroll
            for (int c = 0; c < W; ++c) { Hou[c] = pc_pack((uint32_t)(PC_NEG4 + TG::tOF), 0u); Fu[c] = pc_pack((uint32_t)(PC_NEG4 + TG::tF), 0u); }
            double o_E = pc_pack((uint32_t)(PC_NEG4 + TG::tE), 0u);
            double p_HoL = pc_pack((uint32_t)(PC_NEG4 + TG::tOF), 0u);
            uint32_t v_hb = (uint32_t)(PC_S4(-22) + TG::tOF), v_h00 = (uint32_t)(PC_S4(-12) + TG::tOF);
            const int hl = lane & (PC_STRIP_WIN - 1);
            auto refill = [&](int base) {
                if constexpr (PIPE) {
                    // `base - 2` steps of this pass are done: their boundary entries first, then the word that says so
                    // (both waves sit on one CU: the stores must have reached the L2 they share -- vmcnt counts stores on gfx9, and the
                    // wait is spelled out because the compiler trimmed the fence's own down to lgkmcnt inside this loop -- and the reader's
                    // loads bypass the L1; an agent-scope fence would also write the whole L2 back, every 32 steps)
                    if (!last_pass && base >= 2) publish((item << 17) + (uint32_t)(base - 2));
                    // entries base .. base + 31 of the line I read were written at the other wave's steps base + 63 .. base + 94
                    if (pass > 0 && base < seg_len) {
                        const uint32_t need = ((uint32_t)((pass - 1) / NWV) << 17) + (uint32_t)base + 95u;     // (a finished pass counts as 1 << 17)
                        while (__hip_atomic_load(prog_prev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(4);
                    }
                }
                pc_wave_lds_sync();
                const uint32_t p = (uint32_t)base + (uint32_t)hl;
                if (lane < PC_STRIP_WIN) {                                // the stream entries of positions base .. base + 31
                    uint32_t entry = 0;
                    if (p < (uint32_t)seg_len) {
                        const int i = (int)p - 1;
                        if (i < 0) entry = PCF_RESET;
                        else {
                            const uint32_t code = ap[i];
                            entry = code | (i == la - 1 ? PCF_LAST : 0) | (row_part(min(code, (uint32_t)(ROWS - 1))) << 16);
                        }
                    }
                    ring[hl] = entry;
                } else if (pass > 0 && p < (uint32_t)seg_len) {     // ... and the head's boundary for the same positions, from the pass before
                    const unsigned long long* src = (const unsigned long long*)(line_in + p);
                    const unsigned long long x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(uint4*)&bnd[(p & (PC_STRIP_BND - 1)) * 4] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32));
                }
                pc_wave_lds_sync();
            };
            auto row_addr = [&](uint32_t entry) -> uint32_t {
                if constexpr (BYTE_OFF) {
                    uint32_t addr;
                    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(addr) : "v"(entry), "v"(prof_lane));
                    return addr;
                } else return (entry >> 14) + prof_lane;
            };
            if (T > 0) {                                            // (wave-uniform: a wave without a row only keeps the barriers)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my own stores of the previous pass have left the wave
            refill(0);
            uint32_t a = is_head ? ring[0] : 0u;
            uint32_t e_nxt = ring[1];
            uint32_t e_b = 0;
            uint32_t pw[ND], pm[INC16 ? NDM : 1];
            {
                pc_lds_u32* r0 = (pc_lds_u32*)(size_t)row_addr(a);
#pragma unroll
                for (int q = 0; q < ND; ++q) pw[q] = r0[q * 64];
                if constexpr (INC16) {
#pragma unroll
                    for (int q = 0; q < NDM; ++q) pm[q] = r0[(ND + q) * 64];
                } else pm[0] = 0;
            }
            const unsigned long long headm = __builtin_amdgcn_ballot_w64(is_head), outm = __builtin_amdgcn_ballot_w64(is_out), headoutm = headm | outm;
            const uint32_t v_base_step = PC_BASE_STEP;
            auto step = [&](int t, const bool even, uint32_t a, uint32_t& a_nxt) {
                if (even && ((t + 2) & (PC_STRIP_WIN - 1)) == 0) refill(t + 2);
                // the head lane's left-hand boundary of this row: the constants of column -1 in the first pass, else what the
                // previous pass left for stream position t (every lane reads the one entry: a broadcast)
                uint32_t bHh = v_hb, bHl = 0u, bEh = v_nege, bEl = 0u;
                if (pass > 0) { const uint4 b = *(const uint4*)&bnd[(t & (PC_STRIP_BND - 1)) * 4]; bHh = b.x; bHl = b.y; bEh = b.z; bEl = b.w; }
                uint32_t HoL_hi, HoL_lo, EL_hi, EL_lo, D0_hi, D0_lo;
                unsigned long long anym;
                if constexpr (INC16) {
                    asm volatile(
                        "s_nop 1\n\t"
                        "s_mov_b64 vcc, %[hm]\n\t"
                        "v_cndmask_b32_dpp %[an], %[a], %[en], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[Hh], %[Hwh], %[bHh], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[Eh], %[oEh], %[bEh], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[Hl], %[Hwl], %[bHl], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[El], %[oEl], %[bEl], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cmp_ne_u32_sdwa %[anym], %[a], %[zero] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                        "v_add_u32_sdwa %[D0h], %[pw0], %[Hodh] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                        "v_add_u32_sdwa %[D0l], %[pm0], %[Hodl] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
                        "s_and_b64 %[anym], %[anym], %[hom]\n\t"
                        : [an] "=&v"(a_nxt), [Hh] "=&v"(HoL_hi), [Eh] "=&v"(EL_hi), [Hl] "=&v"(HoL_lo), [El] "=&v"(EL_lo), [D0h] "=&v"(D0_hi),
                          [D0l] "=&v"(D0_lo), [anym] "=&s"(anym)
                        : [hm] "s"(headm), [hom] "s"(headoutm), [a] "v"(a), [en] "v"(e_nxt), [Hwh] "v"(pc_hi(Hou[W - 1])), [bHh] "v"(bHh), [oEh] "v"(pc_hi(o_E)), [bEh] "v"(bEh),
                          [Hwl] "v"(pc_lo(Hou[W - 1])), [bHl] "v"(bHl), [oEl] "v"(pc_lo(o_E)), [bEl] "v"(bEl), [zero] "v"(v_zero), [pw0] "v"(pw[0]), [pm0] "v"(pm[0]),
                          [Hodh] "v"(pc_hi(p_HoL)), [Hodl] "v"(pc_lo(p_HoL))
                        : "vcc", "scc");
                } else {
                    unsigned long long c2;
                    asm volatile(
                        "s_nop 1\n\t"
                        "s_mov_b64 vcc, %[hm]\n\t"
                        "v_cmp_eq_u32_sdwa %[c2], %[a], %[bc0] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                        "v_cndmask_b32_dpp %[an], %[a], %[en], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[Hh], %[Hwh], %[bHh], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[Eh], %[oEh], %[bEh], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[Hl], %[Hwl], %[bHl], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32_dpp %[El], %[oEl], %[bEl], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cmp_ne_u32_sdwa %[anym], %[a], %[zero] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                        "v_add_u32_sdwa %[D0h], %[pw0], %[Hodh] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                        "v_addc_co_u32 %[D0l], %[c2], %[K], %[Hodl], %[c2]\n\t"
                        "s_and_b64 %[anym], %[anym], %[hom]\n\t"
                        : [an] "=&v"(a_nxt), [Hh] "=&v"(HoL_hi), [Eh] "=&v"(EL_hi), [Hl] "=&v"(HoL_lo), [El] "=&v"(EL_lo), [D0h] "=&v"(D0_hi),
                          [D0l] "=&v"(D0_lo), [anym] "=&s"(anym), [c2] "=&s"(c2)
                        : [hm] "s"(headm), [hom] "s"(headoutm), [a] "v"(a), [en] "v"(e_nxt), [Hwh] "v"(pc_hi(Hou[W - 1])), [bHh] "v"(bHh), [oEh] "v"(pc_hi(o_E)), [bEh] "v"(bEh),
                          [Hwl] "v"(pc_lo(Hou[W - 1])), [bHl] "v"(bHl), [oEl] "v"(pc_lo(o_E)), [bEl] "v"(bEl), [zero] "v"(v_zero), [K] "v"(K), [bc0] "v"(bc[0]), [pw0] "v"(pw[0]),
                          [Hodh] "v"(pc_hi(p_HoL)), [Hodl] "v"(pc_lo(p_HoL))
                        : "vcc", "scc");
                }
                pc_lds_u32* nxt = (pc_lds_u32*)(size_t)row_addr(a_nxt);
                if (even) {
                    const uint2 e2 = *(const uint2*)&ring[(t + 2) & (PC_STRIP_WIN - 1)];
                    e_nxt = e2.x; e_b = e2.y;
                } else e_nxt = e_b;
                unsigned long long rstm = 0, lastm = 0;
                asm volatile("" : "+s"(anym));
                if (anym != 0)
                    asm volatile("v_cmp_lt_u32_sdwa %0, %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"
                                 "v_cmp_eq_u32_sdwa %1, %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"
                                 "s_and_b64 %0, %0, %4\n\t"
                                 "s_and_b64 %1, %1, %5"
                                 : "=&s"(rstm), "=&s"(lastm) : "v"(K), "v"(a), "s"(headm), "s"(outm) : "scc");
                if (rstm != 0 && pass == 0) {                       // the alignment starts (first pass only: later passes read the virtual row's boundary)
                    uint32_t inc;
                    asm volatile("v_cndmask_b32 %0, %4, %5, %6\n\tv_add_u32 %1, %1, %0\n\tv_add_u32 %2, %2, %0\n\tv_cndmask_b32 %3, %3, %2, %6"
                                 : "=&v"(inc), "+v"(v_hb), "+v"(v_h00), "+v"(HoL_hi) : "v"(v_zero), "v"(v_base_step), "s"(rstm));
                }
                const double HoL = pc_pack(HoL_hi, HoL_lo);
                p_HoL = HoL;
                PcRow<W, 0, RULE, INC16>::run(pc_pack(D0_hi, D0_lo), HoL, pc_pack(EL_hi, EL_lo), Hou, Fu, bc, pw, pm, nxt, a, K, o_E);
                if (!last_pass) {                                   // what lane 64 would have received for this row: the next pass's boundary
                    const int p = t - 63;
                    if (lane == 63 && p >= 0 && p < seg_len)
                        line[p] = make_uint4(pc_hi(Hou[W - 1]), pc_lo(Hou[W - 1]), pc_hi(o_E), pc_lo(o_E));
                }
                asm volatile("" : "+s"(lastm));
                if (lastm != 0) {
                    asm volatile("" ::: "memory");
                    if ((a & PCF_LAST) && is_out) {
                        const uint32_t st = PcPick<W, 0>::get(Hou, c_out);
                        const uint32_t n_ident = INC16 ? (st & (PC_INC16_K - 1)) : (st & 0xffffu), n_diag = INC16 ? (st >> 13) : (st >> 16);
                        res[bucket_dest ? bucket_dest[tk.begin + my_row] : (uint32_t)(tk.begin + my_row)] = make_uint2(n_ident, (uint32_t)la + (uint32_t)lb_all - n_diag);
                    }
                }
            };
            uint32_t a2 = 0;
#pragma unroll 1
            for (int t = 0; t < T; t += 2) {
                step(t, true, a, a2);
                step(t + 1, false, a2, a);
            }
            }
            if constexpr (PIPE) if (!last_pass) publish((item + 1u) << 17);     // the whole line is written: whoever reads it need not look at steps any more
        }
        }
        __syncthreads();                                           // the next task rebuilds the profile
    }
}

template <int W, int RULE, bool INC16, bool PIPE = false>
int pc_strip_launch(unsigned nblocks, int nw, size_t lds, hipStream_t st, const PcDev& d, const PcTask* tasks, int ntasks,
                    const int32_t* bucket_row, const uint32_t* bucket_dest, uint2* res, int ppos, uint4* spill, unsigned spill_stride) {
    hipLaunchKernelGGL((k_nw_strip<W, RULE, INC16, PIPE>), dim3(nblocks), dim3(64 * nw), lds, st, d, tasks, ntasks, bucket_row, bucket_dest, res, ppos, spill, spill_stride);
    return (int)hipGetLastError();
}
#define PC_STRIP_SIG (unsigned, int, size_t, hipStream_t, const PcDev&, const PcTask*, int, const int32_t*, const uint32_t*, uint2*, int, uint4*, unsigned)
// the strip-mined kernel's widths: the wide variants for tasks of 3-4 rows (four waves share the profile of a pass), W = 12 / W = 8
// for tasks of two rows / one row -- the profile of a pass is 24 x W x 64 bytes of LDS PER COLUMN GENE, 49 KB at W = 32: with one
// row to align against it a CU would hold three waves; at W = 8 it holds twelve, for 9 % more instructions per cell
#define PC_STRIP_W_ONE_ROW 8
#define PC_STRIP_W_TWO_ROWS 12

// ---- kernels over the body -------------------------------------------------------------------------------------
// (second launch bound = waves per SIMD the compiler must leave room for: W = 48 needs 257 registers left to itself, one more
// than the 256 that let two waves share a SIMD)
template <int W, int RULE, bool INC16>
__global__ __launch_bounds__(64 * pc_max_waves(W), (W == 48 ? 2 : 1)) void k_nw_systolic(PcDev d, const PcTask* __restrict__ tasks,
                                                               const int32_t* __restrict__ bucket_row,
                                                               const uint32_t* __restrict__ bucket_dest,
                                                               uint2* __restrict__ res, int ppos) {
    pc_nw_body<W, RULE, INC16>(d, tasks, blockIdx.x, bucket_row, bucket_dest, res, ppos);
}

// ONE launch for the launch classes of a register tier (r04).  Launches of a stream run back to back -- a launch waits for the
// LAST workgroup of the one before it (the AQL barrier bit; hipExtLaunchKernel's hipExtAnyOrderLaunch, which would drop it, is
// ignored on gfx950: tests/hw/anyorder_probe.hip) -- and HIP maps the streams onto four hardware queues, so a fill of ~80
// launches, ~20 in a row per queue, spends a task's duration per launch with its queue feeding nothing: the 12 ms that a fill
// costs whatever its size (T(w) = 11.8 + 572 / w ms for a 1/w share of the N = 5,000 fill).  The variants of a tier need about
// the same registers -- the same waves per SIMD -- so their bodies share one kernel at no cost in occupancy; a launch then
// holds the tasks of ~20 classes (its SEGMENTS: a block range each, with the variant that runs it), far more than the chip
// holds at once, and only the last launches of a fill have a tail.  LDS and workgroup size are the launch's: the segments of a
// launch agree on waves per workgroup, and the dynamic LDS is the largest any of them needs (pc_nw.hip groups them so that
// this stays within what the tier's register budget lets a CU hold anyway).
#define PC_FUSE_MAX_SEG PC_FUSE_MAX_SEGMENTS
struct PcFuseArgs {
    int32_t nseg;
    uint32_t block_end[PC_FUSE_MAX_SEG];              // segment i holds blocks [block_end[i-1], block_end[i])
    uint32_t task_begin[PC_FUSE_MAX_SEG];             // ... which are tasks task_begin[i] + (block - block_end[i-1])
    int32_t w[PC_FUSE_MAX_SEG];                       // ... run by the body of this many columns per lane
};
// tiers by registers (waves per SIMD): 2..8 (<= 72: seven or more), 9..12 (<= 92: five), 13..19 (<= 128: four), 20..24 (<= 152: three)
#define PC_NUM_TIERS 4
__host__ __device__ constexpr int pc_tier_of(int W) { return W <= 8 ? 0 : W <= 12 ? 1 : W <= 19 ? 2 : W <= 24 ? 3 : -1; }

// (second launch bound: the waves per SIMD the tier's widest body reaches on its own -- bundled, tier 2's profile-cell kernel
// came out at 130 registers, two past the 128 of four waves per SIMD)
__host__ __device__ constexpr int pc_tier_waves_per_simd(int tier) { return tier == 0 ? 7 : tier == 1 ? 5 : tier == 2 ? 4 : 3; }
template <int TIER, int RULE, bool INC16>
__global__ __launch_bounds__(64 * 8, pc_tier_waves_per_simd(TIER)) void k_nw_systolic_tier(PcDev d, const PcTask* __restrict__ tasks, PcFuseArgs f,
                                                              const int32_t* __restrict__ bucket_row, const uint32_t* __restrict__ bucket_dest,
                                                              uint2* __restrict__ res, int ppos) {
    int seg = 0;
    uint32_t first = 0;
    while (seg + 1 < f.nseg && blockIdx.x >= f.block_end[seg]) { first = f.block_end[seg]; ++seg; }     // (scalar: blockIdx is wave-uniform)
    const uint32_t task = f.task_begin[seg] + (blockIdx.x - first);
    const int w = f.w[seg];
#define PC_BODY(WW) case WW: pc_nw_body<WW, RULE, INC16>(d, tasks, task, bucket_row, bucket_dest, res, ppos); break;
    if constexpr (TIER == 0) { switch (w) { PC_BODY(2) PC_BODY(3) PC_BODY(4) PC_BODY(5) PC_BODY(6) PC_BODY(7) PC_BODY(8) default: break; } }
    else if constexpr (TIER == 1) { switch (w) { PC_BODY(9) PC_BODY(10) PC_BODY(11) PC_BODY(12) default: break; } }
    else if constexpr (TIER == 2) { switch (w) { PC_BODY(13) PC_BODY(14) PC_BODY(15) PC_BODY(16) PC_BODY(17) PC_BODY(18) PC_BODY(19) default: break; } }
    else { switch (w) { PC_BODY(20) PC_BODY(22) PC_BODY(24) default: break; } }
#undef PC_BODY
}

// One launch of k_nw_systolic<W, RULE, INC16> (the wide variants) / k_nw_systolic_tier<TIER, RULE, INC16>; both return the
// hipError_t of the launch.  The only things a translation unit needs to instantiate (explicitly, in pc_nw_rules.hip;
// implicitly for rules 0 and 1 in pc_nw.hip).
template <int W, int RULE, bool INC16>
int pc_systolic_launch(unsigned ntasks, int nw, size_t lds, hipStream_t st, const PcDev& d, const PcTask* tasks,
                       const int32_t* bucket_row, const uint32_t* bucket_dest, uint2* res, int ppos) {
    hipLaunchKernelGGL((k_nw_systolic<W, RULE, INC16>), dim3(ntasks), dim3(64 * nw), lds, st, d, tasks, bucket_row, bucket_dest, res, ppos);
    return (int)hipGetLastError();
}
template <int TIER, int RULE, bool INC16>
int pc_tier_launch(unsigned nblocks, int nw, size_t lds, hipStream_t st, const PcDev& d, const PcTask* tasks, const PcFuseArgs& f,
                   const int32_t* bucket_row, const uint32_t* bucket_dest, uint2* res, int ppos) {
    hipLaunchKernelGGL((k_nw_systolic_tier<TIER, RULE, INC16>), dim3(nblocks), dim3(64 * nw), lds, st, d, tasks, f, bucket_row, bucket_dest, res, ppos);
    return (int)hipGetLastError();
}
#define PC_TIER_SIG (unsigned, int, size_t, hipStream_t, const PcDev&, const PcTask*, const PcFuseArgs&, const int32_t*, const uint32_t*, uint2*, int)
#define PC_FOR_TIER(M) M(0) M(1) M(2) M(3)


// X-macro over the compiled widths: PC_FOR_W2(M) for those that exist with both cells, PC_FOR_W1(M) for the wide ones
#define PC_FOR_W2(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(22) M(24)
#define PC_FOR_W1(M) M(32) M(48) M(64)
#define PC_SYSTOLIC_SIG (unsigned, int, size_t, hipStream_t, const PcDev&, const PcTask*, const int32_t*, const uint32_t*, uint2*, int)
