// pc_nw.hip -- K4: batched global alignment (Needleman-Wunsch, affine gaps 11/1,
// BLOSUM62) producing only what the reference reads from each alignment
// (metrics.py:216-217): n_ident = comp.count("|") and len(traceback.query).
//
// Replaces parasail.nw_trace_diag_16 + get_traceback (metrics.py:174-175).  No trace
// table is stored: the traceback's choice at every cell is a deterministic local rule
// (SURVEY.md 8c items 4-5), so (n_ident, n_diag) ride along with H, E and F in one
// forward pass and aln_len = la + lb - n_diag.
//   E(i,j) = max(H(i,j-1)-11, E(i,j-1)-1)   ties -> extend      (gap in query, "INS")
//   F(i,j) = max(H(i-1,j)-11, F(i-1,j)-1)   ties -> extend      (gap in ref,   "DEL")
//   H(i,j) = max(H(i-1,j-1)+S, E, F)        ties -> DIAG, then F, then E
// Those three tie-breaks are rule 0 of the TIE-RULE TABLE below (PcTag / pc_cell): parasail's source is not available
// (SURVEY.md 8c), so each is a switch -- every kernel exists for all 8 combinations and pc_set_tie_rule() picks
// one at run time (the CPU checker under tests/ has the same switch).  tests/golden/tie_sensitivity.json holds
// what each switch is worth.
// Scores are kept as Ho = H - 11 ("already opened"), which is what both E of the next
// column and F of the next row consume; the substitution profile is biased by +11.
// Stats are one u32: n_ident in the low half, n_diag in the high half, so the diagonal
// update is a single add-with-carry: SD = SHdiag + 0x10000 + (a == b).
// In the systolic kernel score, tie-break tag and stats travel as ONE 64-bit word and v_max_f64 is the
// lexicographic max over them (see "The DP cell as a LEXICOGRAPHIC MAX" below).
// VALU work: no MFMA (there is no dense contraction in this recurrence).
#include "pc_nw_systolic.h"     // the systolic kernel (shared with pc_nw_rules.hip), BLOSUM62


// One DP cell in the Ho convention.  In:  Hol/El/SHl/SEl from (i,j-1), Hou/Fu/SHu/SFu from
// (i-1,j), Hod/SHd from (i-1,j-1), sp = S(a_i,b_j)+11, eq = (a_i == b_j).  Out: Ho,E,F,SH,SE,SF.
struct PcCell { int Ho, E, F; uint32_t SH, SE, SF; };

// TIE-RULE TABLE (bits of `rule`; include/phamclust_hip.h documents the same numbering):
//   bit 0  H takes a gap state and E == F:   0: F (DEL) before E (INS)        1: E before F
//   bit 1  E: open == extend:                0: extend (open iff strictly >)  1: open
//   bit 2  F: open == extend:                0: extend                        1: open
// DIAG always wins a tie with a gap state.  This C++ cell (general kernel) reads the rule at run time; the
// systolic kernel spells the rule as tags of its 64-bit words (PcTag<RULE> below).
__device__ __forceinline__ PcCell pc_cell(int Hol, int El, uint32_t SHl, uint32_t SEl, int Hou, int Fu, uint32_t SHu,
                                          uint32_t SFu, int Hod, uint32_t SHd, int sp, bool eq, int rule) {
    PcCell c;
    const int Ee = El - PC_EXT, Fe = Fu - PC_EXT;
    c.E = max(Hol, Ee);
    c.SE = ((rule & 2) ? Hol >= Ee : Hol > Ee) ? SHl : SEl;
    c.F = max(Hou, Fe);
    c.SF = ((rule & 4) ? Hou >= Fe : Hou > Fe) ? SHu : SFu;
    const int D = Hod + sp;
    const int H = max(max(D, c.E), c.F);
    const uint32_t SD = SHd + 0x10000u + (eq ? 1u : 0u);
    const uint32_t ST = (rule & 1) ? ((H == c.E) ? c.SE : c.SF) : ((H == c.F) ? c.SF : c.SE);   // flat selects: nested ?: becomes control flow
    c.SH = (H == D) ? SD : ST;
    c.Ho = H - PC_OPEN;
    return c;
}

// ---------------------------------------------------------------------------------
// General kernel: any lengths.  One workgroup (one wave) per task, one lane per row
// sequence, all lanes share the task's column sequence b (so b_j is wave-uniform and
// the per-column state of 64 alignments is one coalesced 1 KiB line in HBM scratch).
// Used for column sequences longer than the systolic variants cover, and by tests.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_nw_general(PcDev d, const PcTask* __restrict__ tasks, int ntasks,
                                                   const int32_t* __restrict__ bucket_row,
                                                   const uint32_t* __restrict__ bucket_dest, uint2* __restrict__ res,
                                                   int4* __restrict__ scratch, int64_t scratch_stride, int ppos, int rule) {
    __shared__ int8_t tab[24][24];
    for (int i = threadIdx.x; i < 576; i += 64) tab[i / 24][i % 24] = (int8_t)(c_b62[i / 24][i % 24] + PC_OPEN);
    __syncthreads();
    int4* sc = scratch + (int64_t)blockIdx.x * scratch_stride;
    const int lane = threadIdx.x;
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const PcTask tk = tasks[task];
        const int lb = d.gene_len[tk.gene];
        const uint8_t* bp = d.codes + d.gene_off[tk.gene];
      for (int base = tk.begin; base < tk.end; base += 64) {       // a task may hold up to PC_TASK_ROWS rows
        const int row = base + lane;
        const bool active = row < tk.end;
        int la = 0; const uint8_t* ap = nullptr;
        if (active) { const int ga = bucket_row[row]; la = d.gene_len[ga]; ap = d.codes + d.gene_off[ga]; }
        int la_max = la;
        for (int o = 32; o > 0; o >>= 1) la_max = max(la_max, __shfl_xor(la_max, o));
        // row -1: Ho(-1,j) = -(11 + j) - 11, F = -inf, stats 0
        for (int j = 0; j < lb; ++j) sc[(int64_t)j * 64 + lane] = make_int4(-(PC_OPEN + j * PC_EXT) - PC_OPEN, PC_NEG, 0, 0);
        uint32_t final_stat = 0;
        for (int i = 0; i < la_max; ++i) {
            const bool live = i < la;
            const int ai = live ? ap[i] : 0;
            const int8_t* trow = tab[min(ai, 23)];
            int Hol = -(PC_OPEN + i * PC_EXT) - PC_OPEN, El = PC_NEG;
            uint32_t SHl = 0, SEl = 0;
            int Hod = (i == 0 ? 0 : -(PC_OPEN + (i - 1) * PC_EXT)) - PC_OPEN;
            uint32_t SHd = 0;
            for (int j = 0; j < lb; ++j) {
                const int bj = bp[j];
                const int4 up = sc[(int64_t)j * 64 + lane];
                const PcCell c = pc_cell(Hol, El, SHl, SEl, up.x, up.y, (uint32_t)up.z, (uint32_t)up.w, Hod, SHd,
                                         trow[min(bj, 23)], ai == bj || (ppos && trow[min(bj, 23)] > PC_OPEN), rule);   // ppos: '+' columns count too
                if (live) sc[(int64_t)j * 64 + lane] = make_int4(c.Ho, c.F, (int)c.SH, (int)c.SF);
                Hod = up.x; SHd = (uint32_t)up.z;
                Hol = c.Ho; El = c.E; SHl = c.SH; SEl = c.SE;
            }
            if (i == la - 1) final_stat = SHl;
        }
        if (active) {
            const uint32_t ident = final_stat & 0xffffu, ndiag = final_stat >> 16;
            res[bucket_dest ? bucket_dest[row] : (uint32_t)row] = make_uint2(ident, (uint32_t)(la + lb) - ndiag);
        }
      }
    }
}
// Tie rules 2..7 of the systolic kernel are instantiated in pc_nw_rules.hip (three objects): not here
#define PC_EXT1R(W, R) extern template int pc_systolic_launch<W, R, false> PC_SYSTOLIC_SIG;
#define PC_EXT1(W) PC_EXT1R(W, 2) PC_EXT1R(W, 3) PC_EXT1R(W, 4) PC_EXT1R(W, 5) PC_EXT1R(W, 6) PC_EXT1R(W, 7)
#define PC_EXT_TIER_R(T, R) extern template int pc_tier_launch<T, R, false> PC_TIER_SIG; extern template int pc_tier_launch<T, R, true> PC_TIER_SIG;
#define PC_EXT_TIER(T) PC_EXT_TIER_R(T, 2) PC_EXT_TIER_R(T, 3) PC_EXT_TIER_R(T, 4) PC_EXT_TIER_R(T, 5) PC_EXT_TIER_R(T, 6) PC_EXT_TIER_R(T, 7)
PC_FOR_TIER(PC_EXT_TIER)
PC_FOR_W1(PC_EXT1)
#define PC_EXT_STRIP_R(W, INC, R) extern template int pc_strip_launch<W, R, INC> PC_STRIP_SIG;
#define PC_EXT_STRIP(W, INC) PC_EXT_STRIP_R(W, INC, 2) PC_EXT_STRIP_R(W, INC, 3) PC_EXT_STRIP_R(W, INC, 4) PC_EXT_STRIP_R(W, INC, 5) PC_EXT_STRIP_R(W, INC, 6) PC_EXT_STRIP_R(W, INC, 7)
PC_EXT_STRIP(32, false) PC_EXT_STRIP(48, false) PC_EXT_STRIP(64, false) PC_EXT_STRIP(24, true) PC_EXT_STRIP(8, false) PC_EXT_STRIP(12, false)
#define PC_EXT_PIPE_R(W, R) extern template int pc_strip_launch<W, R, false, true> PC_STRIP_SIG;
#define PC_EXT_PIPE(W) PC_EXT_PIPE_R(W, 2) PC_EXT_PIPE_R(W, 3) PC_EXT_PIPE_R(W, 4) PC_EXT_PIPE_R(W, 5) PC_EXT_PIPE_R(W, 6) PC_EXT_PIPE_R(W, 7)
PC_EXT_PIPE(PC_STRIP_W_ONE_ROW)

// columns-per-lane of the compiled systolic variants
static const int g_variant_w[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 24, 32, 48, 64};
static const int g_num_variants = (int)(sizeof(g_variant_w) / sizeof(int));

static bool class_inc16(int W, int G);
int pc_nw_num_variants() { return g_num_variants; }
int pc_nw_variant_w(int v) { return (v >= 0 && v < g_num_variants) ? g_variant_w[v] : 0; }
int pc_nw_variant_takes_any_byte(int v) { return v < 0 || v >= g_num_variants || g_variant_w[v] > PC_INC16_MAX_W; }   // 0: some class of this variant may run the profile cell

// Variant for a column gene of lb residues.  Time per row step ~ (W + c0 + c1 nseg) cell-equivalents (x 1.014 at
// W = 22, x 1.022 at W = 24: three waves per SIMD), during which a wave retires nseg rows of lb cells; c1 = 0.535
// from a least-squares fit to measured kernel-only GCUPS of every variant over L = 60..1200
// (profiles/r01/experiments/l_variant_gcups.txt), c0 = 0.3 after the step prologue shrank to ~12 instructions (end-to-end
// sweep: the fill time is flat within 1 % over c0 = 0.3..1.0).  The nseg term stands for what short sequences pay
// per alignment and per task (virtual row, pipeline fill, profile build).  Minimise cost per retired row over the
// variants whose 64*W columns cover lb.
// Column genes beyond the widest variant's 4,096 columns run strip-mined (k_nw_strip): ceil(lb / 64 W) passes of a wide variant,
// a pass costing a row step of that variant per row whatever its width -- the fewest step-instructions win (W = 32 / 48 / 64:
// 2,048 / 3,072 / 4,096 columns per pass; penalties as below).  PC_STRIP=0 sends them to the general kernel as r01-r03 did.
static bool strip_enabled() { static const bool off = getenv("PC_STRIP") && !atoi(getenv("PC_STRIP")); return !off; }
// Does a launch of this class run on k_nw_strip (and need the scratch slab)?  Column genes beyond the variant's 64 x W columns
// always; and the one- / two-row tasks of the wide variants' ordinary classes, which narrow passes serve better (see PC_STRIP_W_*).
int pc_nw_launch_is_strip(int variant, int max_lb, int wave_mode, int ppos) {
    if (variant < 0 || variant >= g_num_variants) return 0;
    const int W = g_variant_w[variant];
    if (max_lb > 64 * W) return 1;
    return (!ppos && W >= 32 && wave_mode != PC_MODE_CLASS && strip_enabled()) ? 1 : 0;
}
int pc_nw_strip_passes(int lb, int variant) {
    if (variant < 0 || variant >= g_num_variants || lb <= 0) return 0;
    const int cols = 64 * g_variant_w[variant];
    return (lb + cols - 1) / cols;
}
static int choose_strip_variant(int lb) {
    if (!strip_enabled()) return -1;
    int best = -1; double best_cost = 0;
    for (int v = 0; v < g_num_variants; ++v) {
        const int W = g_variant_w[v];
        if (W < 32) continue;
        const double pen = W >= 64 ? 1.15 : W >= 48 ? 1.08 : 1.04;
        const double cost = (double)pc_nw_strip_passes(lb, v) * (W + 1.0) * pen;
        if (best < 0 || cost < best_cost) { best = v; best_cost = cost; }
    }
    return best;
}
int pc_nw_choose_variant(int lb) {
    if (lb <= 0) return -1;
    if (lb > 64 * PC_MAX_W) return choose_strip_variant(lb);
    int best = -1; double best_cost = 0;
    for (int v = 0; v < g_num_variants; ++v) {
        const int W = g_variant_w[v];
        const int G = (lb + W - 1) / W;
        if (G > 64) continue;
        int nseg = 64 / G; if (nseg > 16) nseg = 16;
        // W >= 32 exist for very long column genes (up to 4,096 residues).  They run at 2 or 1 waves per SIMD yet
        // measure 1.9-2.2 TCUPS at full lane use (profiles/r01/experiments/o_wide_variant_gcups.txt): one wave can keep its
        // SIMD's VALU busy, so the penalty is small
        const double pen = W >= 64 ? 1.15 : W >= 48 ? 1.08 : W >= 32 ? 1.04 : W >= 24 ? 1.022 : (W >= 22 ? 1.014 : 1.0);
        // constants of the cost model; their sweeps came out flat (profiles/r02/experiments, profiles/r04/experiments/choose_sweep_after_retag.txt) and the
        // environment knobs that drove them (PC_CHOOSE_C0 / _C1 / _CELL) are gone (r05)
        constexpr double c0 = 0.3, c1 = 0.535;
        constexpr double cell = 0.94;  // relative cost of the 10-instruction cell's classes (sweep 1.0 / 0.96 / 0.93 / 0.90: 239.6 / 237.8 / 238.1 / 237.9 ms at N=3,000)
        const double cost = (W + c0 + c1 * nseg) * pen * (class_inc16(W, G) ? cell : 1.0) / nseg;
        if (best < 0 || cost < best_cost) { best = v; best_cost = cost; }
    }
    return best;
}

// A bucket whose row count is not a multiple of its variant's nseg leaves r = n mod nseg rows for a last wave round in
// which only r of the nseg segments work.  Those r rows can go to a variant with fewer, longer segments instead:
// returns that variant, or -1 when staying is as cheap (the cost of a wave round is the step cost of the variant;
// 30 % margin for the extra workgroup).
int pc_nw_choose_remainder(int lb, int r, int main_variant) {
    static const bool off = getenv("PC_REMAINDER") && !strcmp(getenv("PC_REMAINDER"), "0");
    if (off || lb <= 0 || r <= 0 || main_variant < 0 || main_variant >= g_num_variants) return -1;
    constexpr double c0 = 0.3, c1 = 0.535;
    auto step_cost = [&](int v, int& nseg) {
        const int W = g_variant_w[v], G = (lb + W - 1) / W;
        if (G > 64) { nseg = 0; return 0.0; }
        nseg = 64 / G; if (nseg > PC_MAX_SEG) nseg = PC_MAX_SEG;
        const double pen = W >= 64 ? 1.15 : W >= 48 ? 1.08 : W >= 32 ? 1.04 : W >= 24 ? 1.022 : (W >= 22 ? 1.014 : 1.0);
        return (W + c0 + c1 * nseg) * pen;
    };
    int nseg0; const double stay = step_cost(main_variant, nseg0);
    if (nseg0 <= 1 || r >= nseg0) return -1;
    constexpr double margin = 0.7;
    int best = -1; double best_cost = margin * stay;
    for (int v = 0; v < g_num_variants; ++v) {
        int nseg; const double sc = step_cost(v, nseg);
        if (!nseg) continue;
        const double cost = sc * ((r + nseg - 1) / nseg);
        if (cost < best_cost) { best = v; best_cost = cost; }
    }
    return best;
}

// LDS of one workgroup: score table, the waves' private regions, the profile of a column gene spread over G lanes
static size_t systolic_lds_bytes(int W, int G, int nw, bool inc16, bool any_bucket = false) {
    const int Gb = pc_nw_g_bucket(G), rpl = 64 / Gb;                      // profile rows per 64-dword line (see the kernel)
    int nseg_max = (Gb == 8 || any_bucket) ? PC_MAX_SEG : 64 / (Gb / 2 + 1); if (nseg_max > PC_MAX_SEG) nseg_max = PC_MAX_SEG;   // the bucket's fewest lanes per segment (any_bucket: the launch's column genes may be shorter than its bucket)
    const size_t lines = (inc16 && Gb == 64) ? (size_t)2 * ((pc_prof_rows(inc16) + 1) / 2)       // two half tables of 2 rows per line
                                             : (size_t)((pc_prof_rows(inc16) + rpl - 1) / rpl);
    return (size_t)(144 + nw * pc_wave_lds_dwords(nseg_max)) * 4 + lines * pc_prof_row_dwords(W, inc16) * 256;
}
// Lanes-per-segment bucket of a launch class (pc_api.hip's classes use the same bounds): every column gene of a launch
// lies in one bucket, so launch, task sizes and LDS agree on the waves per workgroup without passing it around
int pc_nw_g_bucket(int G) { return G <= 8 ? 8 : (G <= 16 ? 16 : (G <= 32 ? 32 : 64)); }
// Which cell a launch class runs (measured per class with the launches serialised, profiles/r02/experiments/l_class_times.txt):
// the 16-bit increment profile wins 5-7 % where four workgroups still fit a CU beside it (segments of up to 16 lanes)
// and 3-5 % with 8-wave workgroups on segments of up to 32 lanes for W = 11..19 (W <= 9: -15..-40 %, short strips; W = 20: nothing; W = 24: -6 % in
// either bucket, its profile needs 8-wave groups even at 16 lanes); with
// one segment per wave (up to 64 lanes) the profile of a long gene leaves room for a single 16-wave workgroup per CU
// and loses 5-10 %.  Elsewhere the residue compare.
static bool class_inc16(int W, int G) {
    static const int force = getenv("PC_INC16") ? atoi(getenv("PC_INC16")) : -1;      // tuning: 0 = never, 1 = wherever compiled
    if (W > PC_INC16_MAX_W) return false;
    const int Gb = pc_nw_g_bucket(G);
    if (Gb > 32) return false;                                                        // (measured: no room for enough waves; only percent-positives launches run the profile cell there)
    if (force >= 0) return force != 0;
    return (Gb <= 16 && W <= 22) || (Gb == 32 && W >= 11 && W <= 19);
}
// Waves per workgroup: the fewest (4, 8; at most what the variant's registers allow) that put 16 waves on a CU
// given the LDS the class's largest profile takes; the most allowed if none does
// cell_mode: 0 = the class's own choice, 1 = compare cell ("any byte" classes), 2 = profile cell (percent-positives runs)
static bool mode_inc16(int W, int G, int cell_mode) {
    if (cell_mode == 2) return W <= PC_INC16_MAX_W && G <= 64;
    return cell_mode == 0 && class_inc16(W, G);
}
static int waves_for(int W, int G, int cell_mode) {
    const int Gb = pc_nw_g_bucket(G), top = pc_max_waves(W);
    const bool inc16 = mode_inc16(W, G, cell_mode);
    if (!inc16) return PC_MIN_WAVES;
    for (int nw = PC_MIN_WAVES; nw <= top; nw *= 2)
        if ((int)((size_t)160 * 1024 / systolic_lds_bytes(W, Gb, nw, inc16)) * nw >= 16) return nw;
    return top;
}
int pc_nw_class_waves(int variant, int lb, int compare_only) {
    if (variant < 0 || variant >= g_num_variants || lb <= 0) return PC_MIN_WAVES;
    const int W = g_variant_w[variant];
    return waves_for(W, (lb + W - 1) / W, compare_only != 0 ? 1 : 0);
}
// Percent-positives (aai with ppos=True) needs the profile cell -- "positive" is a property of (row residue, column residue),
// i.e. a table entry, not a residue compare: systolic where that cell can run, the general kernel elsewhere
#define PC_STRIP_PPOS_MAX_LB 8191                     // the profile cell's statistics are 13-bit fields (PC_INC16_K)
int pc_nw_ppos_systolic(int variant, int max_lb) {
    if (variant < 0 || variant >= g_num_variants || max_lb <= 0) return 0;
    const int W = g_variant_w[variant];
    if (max_lb > 64 * W) return (W == PC_INC16_MAX_W && strip_enabled() && max_lb <= PC_STRIP_PPOS_MAX_LB) ? 1 : 0;   // strip-mined passes of W = 24
    return mode_inc16(W, (max_lb + W - 1) / W, 2) ? 1 : 0;
}
// ... and for a class whose own variant cannot (the wide ones, W >= 32, have no profile cell): the widest variant that has
// one, if it can take the class's longest column gene (tasks do not depend on the variant that runs them); -1: none,
// i.e. column genes over 64 x 24 = 1,536 residues stay on the general kernel
int pc_nw_ppos_variant(int max_lb) {
    for (int v = g_num_variants - 1; v >= 0; --v) if (pc_nw_ppos_systolic(v, max_lb)) return v;
    return -1;
}

// Rows (alignments) per workgroup task for a column gene of lb residues.  A task's 4*nseg row streams each walk
// rows/(4*nseg) rows of ~lb residues, one residue per step, a step costing ~(W+1) cell slots: a fixed 208 rows makes
// the tasks of long genes (one segment per wave) run a hundred times longer than those of short ones, and the
// longest task bounds the fill from below once the work is split over ranks (or N is small).  So a stream is capped
// at ~PC_TASK_BUDGET cell slots (never below one row per stream, never above PC_TASK_ROWS rows per task).
static int task_budget() {
    static int b = 0;
    if (!b) { const char* e = getenv("PC_TASK_BUDGET"); b = e ? atoi(e) : 0; if (b <= 0) b = PC_TASK_BUDGET; }
    return b;
}
int pc_nw_task_rows(int lb, int variant, int compare_only) {
    if (variant < 0 || variant >= g_num_variants || lb <= 0) return 64;       // general kernel: one pass of 64 rows
    const int W = g_variant_w[variant];
    const int G = (lb + W - 1) / W;
    if (G > 64) return PC_STRIP_WAVES;                                        // strip-mined: one row per wave, pass after pass
    int nseg = G > 64 ? 1 : 64 / G; if (nseg > PC_MAX_SEG) nseg = PC_MAX_SEG;
    const int64_t steps = task_budget() / (W + 1);
    int64_t per_stream = steps / (lb + 1); if (per_stream < 1) per_stream = 1;
    const int nw = pc_nw_class_waves(variant, lb, compare_only);
    int64_t rows = per_stream * nw * nseg;
    const int64_t wave_cap = (int64_t)nw * (64 / nseg) * nseg;           // a wave keeps at most 64 rows (ceil(R / (nw nseg)) nseg of the task's R)
    if (rows > wave_cap) rows = wave_cap;
    return (int)(rows > PC_TASK_ROWS ? PC_TASK_ROWS : rows);
}

// Launch mode of a task of `rows` rows against a column gene of lb residues on variant W (host mirror of pc_task_mode in
// pc_plan.hip).  A workgroup's waves beyond ceil(rows / nseg) find no row and leave at once, but the LDS sized for all of them
// stays taken until the last wave is done -- with the profile cell's 20-28 KB that is five workgroups per CU with ONE live wave
// each.  Measured on uniform 207-residue genes (tools/bucket_size_bench.py, GCUPS at 1 / 2 / 4 / 8 rows per column gene):
// 4-wave profile-cell workgroups 452 / 894 / 1,726 / 2,344; one wave + compare cell 659 / 1,297 / 2,567 / 2,649; two waves
// 491 / 962 / 1,907 / 2,550 (compare cell 2,681); from 16 rows the 4-wave groups win (2,956 against 2,732-2,853).
int pc_nw_task_mode(int lb, int rows, int variant) {
    static const int off = getenv("PC_SMALL_MODES") ? !atoi(getenv("PC_SMALL_MODES")) : 0;       // PC_SMALL_MODES=0: every task in its class's own workgroup shape
    if (off || variant < 0 || variant >= g_num_variants || lb <= 0) return PC_MODE_CLASS;
    const int W = g_variant_w[variant], G = (lb + W - 1) / W;
    int nseg = G > 64 ? 1 : 64 / G; if (nseg > PC_MAX_SEG) nseg = PC_MAX_SEG;     // (G > 64: strip-mined, one row per wave)
    return rows <= nseg ? PC_MODE_ONE_WAVE : rows <= 2 * nseg ? PC_MODE_TWO_WAVES : PC_MODE_CLASS;
}
int pc_nw_small_modes_enabled() { return pc_nw_task_mode(64, 1, 0) != PC_MODE_CLASS; }

// A wave of the systolic kernel keeps at most 64 rows (pc_nw_body: a lane-indexed row table) and takes ceil(R / (nw nseg)) nseg of a
// task's R rows: nw waves hold any task of up to nw x min over nseg of floor(64 / nseg) nseg = nw x 52 rows (nseg = 13).  The planner
// cuts tasks to the CLASS's waves per workgroup and to PC_TASK_ROWS (pc_nw_task_rows); the LAUNCH's waves are worked out on their own
// (cell, percent-positives variant swap, small-task modes).  Should the two ever drift apart, rows beyond a wave's 64 would be dropped
// without a fault -- their result slots left unwritten -- so every launch checks: tasks in their class's own shape need 52 nw >=
// PC_TASK_ROWS, i.e. four waves; the one- / two-wave modes hold tasks of at most nseg / 2 nseg rows by construction (pc_nw_task_mode).
static_assert(PC_TASK_ROWS <= 4 * 52, "four-wave workgroups must hold a full-size task");
static bool pc_launch_holds_task_rows(int nw, int wave_mode) { return wave_mode != PC_MODE_CLASS || nw * 52 >= PC_TASK_ROWS; }

// Shape of a systolic launch of one class: which cell it runs, waves per workgroup, dynamic LDS
struct PcLaunchShape { int W, nw; bool inc16; size_t lds; };
static int launch_shape(int variant, int max_lb, int ppos, int compare_only, int wave_mode, PcLaunchShape& sh) {
    const int W = g_variant_w[variant];
    int Gmax = (max_lb + W - 1) / W; if (Gmax > 64) Gmax = 64; if (Gmax < 1) Gmax = 1;
    const int cell_mode = ppos ? 2 : ((compare_only != 0 || wave_mode == PC_MODE_ONE_WAVE) ? 1 : 0);
    // small tasks (pc_nw_task_mode): all their rows fit one or two waves, and a workgroup sized for them leaves its LDS to others
    sh.W = W;
    sh.nw = wave_mode == PC_MODE_ONE_WAVE ? 1 : wave_mode == PC_MODE_TWO_WAVES ? 2 : waves_for(W, Gmax, cell_mode);
    sh.inc16 = W <= PC_INC16_MAX_W && mode_inc16(W, Gmax, cell_mode);
    if (ppos && !sh.inc16) { pc_set_error("k_nw_systolic<%d>: percent-positives needs the profile cell (W <= %d)", W, PC_INC16_MAX_W); return PC_ERR_ARG; }
    sh.lds = systolic_lds_bytes(W, Gmax, sh.nw, sh.inc16, cell_mode == 2);
    return PC_OK;
}
// Launch classes that may share one k_nw_systolic_tier launch get the same key (>= 0): same register tier, same cell, same waves
// per workgroup -- and, for the one- and two-wave modes, the same lanes-per-segment bucket: their workgroups are small, so the
// LDS of the widest profile in the launch would cost the narrow ones their occupancy (a 4- or 8-wave workgroup spends at most
// ~9 KB of LDS per wave on the largest profile of its tier, within what the tier's registers let a CU hold anyway).
// -1: launched on its own (wide variants, strip-mined passes, the general kernel).
int pc_nw_fuse_key(int variant, int max_lb, int ppos, int compare_only, int wave_mode) {
    static const bool off = getenv("PC_FUSE") && !atoi(getenv("PC_FUSE"));
    if (variant < 0 || variant >= g_num_variants || max_lb > 64 * g_variant_w[variant]) return -1;
    const int tier = pc_tier_of(g_variant_w[variant]);
    if (tier < 0) return -1;
    PcLaunchShape sh;
    if (launch_shape(variant, max_lb, ppos, compare_only, wave_mode, sh) != PC_OK) return -1;
    const int W = sh.W, G = std::min(64, (max_lb + W - 1) / W);
    const int gb = wave_mode == PC_MODE_CLASS ? 0 : (pc_nw_g_bucket(G) == 8 ? 1 : pc_nw_g_bucket(G) == 16 ? 2 : pc_nw_g_bucket(G) == 32 ? 3 : 4);
    const int key = (((tier * 2 + (sh.inc16 ? 1 : 0)) * 16 + sh.nw) * 8 + gb);
    return off ? key * 64 + variant * 0 + 1000000 + (variant * 4096 + max_lb % 4096) : key;   // PC_FUSE=0: every class its own launch (A/B)
}

template <int TIER, bool INC16>
static int launch_tier(int nblocks, int nw, size_t lds, const PcFuseArgs& f, const PcDev& d, const PcTask* tasks, const int32_t* bucket_row,
                       const uint32_t* bucket_dest, uint2* res, int ppos, int rule, hipStream_t st) {
    hipError_t e = hipSuccess;
    switch (rule) {
#define PC_TIER_CASE(R) case R: e = (hipError_t)pc_tier_launch<TIER, R, INC16>((unsigned)nblocks, nw, lds, st, d, tasks, f, bucket_row, bucket_dest, res, ppos); break;
    PC_TIER_CASE(0) PC_TIER_CASE(1) PC_TIER_CASE(2) PC_TIER_CASE(3) PC_TIER_CASE(4) PC_TIER_CASE(5) PC_TIER_CASE(6) PC_TIER_CASE(7)
#undef PC_TIER_CASE
    default: pc_set_error("tie rule %d out of range 0..7", rule); return PC_ERR_ARG;
    }
    if (e != hipSuccess) { pc_set_error("k_nw_systolic_tier<%d,%d> launch: %s", TIER, rule, hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// One launch over several launch classes of one fuse key (pc_nw_fuse_key): `segs` = (first task, tasks, variant, longest column
// gene, compare-only, wave mode) each, at most PC_FUSE_MAX_SEG of them, tasks taken from task_list.
int pc_launch_nw_group(const PcNwSegment* segs, int nsegs, const PcDev& d, const PcTask* task_list, const int32_t* bucket_row,
                       const uint32_t* bucket_dest, uint2* res, int ppos, int rule, hipStream_t st) {
    if (nsegs <= 0) return PC_OK;
    if (nsegs > PC_FUSE_MAX_SEG) { pc_set_error("pc_launch_nw_group: %d segments (limit %d)", nsegs, PC_FUSE_MAX_SEG); return PC_ERR_ARG; }
    if (rule < 0 || rule >= PC_NUM_TIE_RULES) { pc_set_error("pc_launch_nw_group: tie rule %d out of range", rule); return PC_ERR_ARG; }
    PcFuseArgs f; memset(&f, 0, sizeof(f));
    int tier = -1, nw = 0; bool inc16 = false; size_t lds = 0; uint32_t blocks = 0;
    for (int i = 0; i < nsegs; ++i) {
        const PcNwSegment& sg = segs[i];
        if (sg.variant < 0 || sg.variant >= g_num_variants || sg.max_lb > 64 * g_variant_w[sg.variant] || pc_tier_of(g_variant_w[sg.variant]) < 0) {
            pc_set_error("pc_launch_nw_group: variant %d with %d columns has no tier kernel", sg.variant, sg.max_lb); return PC_ERR_ARG;
        }
        PcLaunchShape sh;
        int rc = launch_shape(sg.variant, sg.max_lb, ppos, sg.compare_only, sg.wave_mode, sh);
        if (rc != PC_OK) return rc;
        if (!pc_launch_holds_task_rows(sh.nw, sg.wave_mode)) { pc_set_error("pc_launch_nw_group: %d-wave workgroups cannot hold the rows of a full-size task (variant %d, %d columns)", sh.nw, sg.variant, sg.max_lb); return PC_ERR_STATE; }
        const int t = pc_tier_of(sh.W);
        if (i == 0) { tier = t; nw = sh.nw; inc16 = sh.inc16; }
        else if (t != tier || sh.nw != nw || sh.inc16 != inc16) { pc_set_error("pc_launch_nw_group: segments of different tier / cell / workgroup size"); return PC_ERR_ARG; }
        lds = std::max(lds, sh.lds);
        blocks += sg.ntasks;
        f.block_end[i] = blocks; f.task_begin[i] = sg.task_begin; f.w[i] = sh.W;
    }
    f.nseg = nsegs;
    if (blocks == 0) return PC_OK;
#define PC_TIER_GO(T) case T: return inc16 ? launch_tier<T, true>((int)blocks, nw, lds, f, d, task_list, bucket_row, bucket_dest, res, ppos, rule, st) \
                                           : launch_tier<T, false>((int)blocks, nw, lds, f, d, task_list, bucket_row, bucket_dest, res, ppos, rule, st);
    switch (tier) { PC_TIER_GO(0) PC_TIER_GO(1) PC_TIER_GO(2) PC_TIER_GO(3) default: break; }
#undef PC_TIER_GO
    pc_set_error("pc_launch_nw_group: no tier %d", tier); return PC_ERR_ARG;
}

// the wide variants (W = 32, 48, 64): a kernel each
template <int W>
static int launch_wide(const PcDev& d, const PcTask* tasks, int ntasks, const int32_t* bucket_row, const uint32_t* bucket_dest, uint2* res,
                       int max_lb, int wave_mode, int rule, hipStream_t st) {
    int Gmax = (max_lb + W - 1) / W; if (Gmax > 64) Gmax = 64; if (Gmax < 1) Gmax = 1;
    const int nw = wave_mode == PC_MODE_ONE_WAVE ? 1 : wave_mode == PC_MODE_TWO_WAVES ? 2 : waves_for(W, Gmax, 1);
    if (!pc_launch_holds_task_rows(nw, wave_mode)) { pc_set_error("k_nw_systolic<%d>: %d-wave workgroups cannot hold the rows of a full-size task", W, nw); return PC_ERR_STATE; }
    const size_t lds = systolic_lds_bytes(W, Gmax, nw, false, false);
    hipError_t e = hipSuccess;
    switch (rule) {
#define PC_WIDE_CASE(R) case R: e = (hipError_t)pc_systolic_launch<W, R, false>((unsigned)ntasks, nw, lds, st, d, tasks, bucket_row, bucket_dest, res, 0); break;
    PC_WIDE_CASE(0) PC_WIDE_CASE(1) PC_WIDE_CASE(2) PC_WIDE_CASE(3) PC_WIDE_CASE(4) PC_WIDE_CASE(5) PC_WIDE_CASE(6) PC_WIDE_CASE(7)
#undef PC_WIDE_CASE
    default: pc_set_error("tie rule %d out of range 0..7", rule); return PC_ERR_ARG;
    }
    if (e != hipSuccess) { pc_set_error("k_nw_systolic<%d,%d> launch: %s", W, rule, hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// Scratch of a strip-mined launch: one boundary line of (longest row + 2) 16-byte entries per wave of every resident workgroup
static size_t strip_bytes_per_block(int max_row_len, int nw = PC_STRIP_WAVES) { return (size_t)nw * ((size_t)max_row_len + 2) * sizeof(uint4); }
size_t pc_nw_strip_scratch_bytes(int max_row_len, int n_cu) {
    const size_t per = strip_bytes_per_block(max_row_len);                    // (sized for 4-wave workgroups, two per CU; one-wave launches
    size_t blocks = (size_t)3 * (n_cu > 0 ? n_cu : 256);                      //  then get 12 workgroups per CU out of the same slab)
    const size_t budget = (size_t)1 << 30;
    if (blocks * per > budget) blocks = std::max<size_t>(budget / per, 8);
    return blocks * per;
}
// Workgroup shape of a strip-mined launch: one row per wave (4 / 2 / 1 waves by the tasks' rows), or -- `pipe` -- the passes of one
// alignment dealt over 8 or 4 waves, a workgroup per row (k_nw_strip's PIPE form).  A lone wave retires a 6,600 x 6,600 alignment in
// T1 = 47 ms; eight waves in T1 / 6.4, four in T1 / 3.2, at 1 resp. 3 workgroups per CU instead of 11 one-wave ones.  So: the
// pipeline while the launch's rows fit the chip in a round or two of it (the alignment's latency is then the launch's
// duration, and on small fills the fill's), one row per wave beyond (more rows in flight, same instructions).
// PC_PIPE=0: never; PC_PIPE=n: always, n waves.
static int strip_launch_waves(int wave_mode, int ntasks, int ppos, int n_cu, bool& pipe) {
    const char* env = getenv("PC_PIPE");                                  // (read per launch: the tests switch it between fills)
    const int forced = env && *env ? atoi(env) : -1;
    const int cu = n_cu > 0 ? n_cu : 256;
    const int nw = wave_mode == PC_MODE_ONE_WAVE ? 1 : wave_mode == PC_MODE_TWO_WAVES ? 2 : PC_STRIP_WAVES;
    pipe = false;
    if (ppos || forced == 0) return nw;
    if (forced > 0) { pipe = true; return forced < PC_PIPE_WAVES_MAX ? forced : PC_PIPE_WAVES_MAX; }
    const long long rows = (long long)ntasks * nw;                       // (at most: a task of the 4- / 2- / 1-wave shape holds up to that many rows)
    if (rows <= cu) { pipe = true; return PC_PIPE_WAVES_MAX; }
    if (rows <= (wave_mode == PC_MODE_CLASS ? 3 : 6) * cu) { pipe = true; return 4; }
    return nw;
}
// ... and of ONE strip-mined launch of `ntasks` tasks that gets a region of its own (run_align_classes: such launches then run side by
// side, on their own streams, instead of one after the other on the one slab): a line per wave of as many workgroups as the
// chip holds of that shape, or as the launch has tasks
size_t pc_nw_strip_launch_bytes(int wave_mode, int ntasks, int max_row_len, int n_cu, int ppos) {
    bool pipe; const int nw = strip_launch_waves(wave_mode, ntasks, ppos, n_cu, pipe);
    const int per_cu = pipe ? (nw > 4 ? 1 : 3) : (nw == 1 ? 12 : nw == 2 ? 8 : 3);
    size_t blocks = (size_t)per_cu * (size_t)(n_cu > 0 ? n_cu : 256);
    const size_t units = (size_t)(ntasks > 0 ? ntasks : 1) * (pipe ? PC_STRIP_WAVES : 1);       // (pipelined: a workgroup per row)
    if (blocks > units) blocks = units;
    return blocks * strip_bytes_per_block(max_row_len, nw);
}
template <int W, bool INC16, bool PIPE = false>
static int launch_strip(const PcDev& d, const PcTask* tasks, int ntasks, const int32_t* bucket_row, const uint32_t* bucket_dest, uint2* res,
                        void* scratch, size_t scratch_bytes, int max_row_len, int ppos, int rule, hipStream_t st, int nw = PC_STRIP_WAVES) {
    const size_t per = strip_bytes_per_block(max_row_len, nw);
    size_t blocks = scratch ? scratch_bytes / per : 0;
    if (blocks == 0) { pc_set_error("k_nw_strip<%d>: needs %zu bytes of scratch per workgroup", W, per); return PC_ERR_ARG; }
    if (blocks > (size_t)ntasks * (PIPE ? PC_STRIP_WAVES : 1)) blocks = (size_t)ntasks * (PIPE ? PC_STRIP_WAVES : 1);
    if (blocks > 4096) blocks = 4096;
    const size_t lines = INC16 ? (size_t)2 * ((pc_prof_rows(true) + 1) / 2) : (size_t)pc_prof_rows(false);
    const size_t lds = (size_t)(144 + nw * pc_strip_wave_lds_dwords()) * 4 + (PIPE ? (size_t)nw : (size_t)1) * lines * pc_prof_row_dwords(W, INC16) * 256;   // PIPE: a profile per wave
    if (PIPE && (nw < 1 || nw > PC_PIPE_WAVES_MAX)) { pc_set_error("k_nw_strip<%d> (pipelined): %d waves", W, nw); return PC_ERR_ARG; }
    hipError_t e = hipSuccess;
    switch (rule) {
#define PC_STRIP_CASE(R) case R: e = (hipError_t)pc_strip_launch<W, R, INC16, PIPE>((unsigned)blocks, nw, lds, st, d, tasks, ntasks, bucket_row, bucket_dest, res, ppos, (uint4*)scratch, (unsigned)(max_row_len + 2)); break;
    PC_STRIP_CASE(0) PC_STRIP_CASE(1) PC_STRIP_CASE(2) PC_STRIP_CASE(3) PC_STRIP_CASE(4) PC_STRIP_CASE(5) PC_STRIP_CASE(6) PC_STRIP_CASE(7)
#undef PC_STRIP_CASE
    default: pc_set_error("tie rule %d out of range 0..7", rule); return PC_ERR_ARG;
    }
    if (e != hipSuccess) { pc_set_error("k_nw_strip<%d,%d> launch: %s", W, rule, hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

size_t pc_nw_fallback_scratch_bytes(int max_lb) {
    size_t per_block = (size_t)64 * (size_t)max_lb * sizeof(int4);
    size_t budget = (size_t)2 << 30;
    size_t blocks = budget / (per_block ? per_block : 1);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 32) blocks = 32;
    return blocks * per_block;
}

int pc_launch_nw(int variant, const PcDev& d, const PcTask* tasks, int ntasks, const int32_t* bucket_row,
                 const uint32_t* bucket_dest, uint2* res, void* scratch, size_t scratch_bytes, int max_lb, int ppos, int rule, int compare_only, hipStream_t st,
                 int wave_mode, int max_row_len) {
    if (ntasks <= 0) return PC_OK;
    if (rule < 0 || rule >= PC_NUM_TIE_RULES) { pc_set_error("pc_launch_nw: tie rule %d out of range", rule); return PC_ERR_ARG; }
    if (variant >= 0 && ppos && !pc_nw_ppos_systolic(variant, max_lb)) { pc_set_error("pc_launch_nw: percent-positives cannot run on variant %d for %d columns", variant, max_lb); return PC_ERR_ARG; }
    if (variant >= 0) {
        if (variant >= g_num_variants) { pc_set_error("pc_launch_nw: no variant %d", variant); return PC_ERR_ARG; }
        if (pc_nw_launch_is_strip(variant, max_lb, wave_mode, ppos)) {           // strip-mined passes (k_nw_strip)
            const int W = g_variant_w[variant];
            bool pipe; const int nw = strip_launch_waves(wave_mode, ntasks, ppos, d.n_cu, pipe);
            // few tasks: the passes of each alignment over the waves of a workgroup, narrow passes whatever the class's width
            if (pipe) return launch_strip<PC_STRIP_W_ONE_ROW, false, true>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 0, rule, st, nw);
            if (ppos && W == PC_INC16_MAX_W) return launch_strip<PC_INC16_MAX_W, true>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 1, rule, st, nw);
            // tasks of one row / two rows: narrow passes in one- / two-wave workgroups, whatever the class's own width
            if (!ppos && nw == 1) return launch_strip<PC_STRIP_W_ONE_ROW, false>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 0, rule, st, 1);
            if (!ppos && nw == 2) return launch_strip<PC_STRIP_W_TWO_ROWS, false>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 0, rule, st, 2);
            if (!ppos && W == 32) return launch_strip<32, false>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 0, rule, st);
            if (!ppos && W == 48) return launch_strip<48, false>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 0, rule, st);
            if (!ppos && W == 64) return launch_strip<64, false>(d, tasks, ntasks, bucket_row, bucket_dest, res, scratch, scratch_bytes, max_row_len, 0, rule, st);
            pc_set_error("pc_launch_nw: variant %d (w = %d) cannot take %d columns", variant, W, max_lb); return PC_ERR_ARG;
        }
        if (ppos && g_variant_w[variant] > PC_INC16_MAX_W) { pc_set_error("pc_launch_nw: percent-positives cannot run on variant %d", variant); return PC_ERR_ARG; }
        switch (g_variant_w[variant]) {
        case 32: return launch_wide<32>(d, tasks, ntasks, bucket_row, bucket_dest, res, max_lb, wave_mode, rule, st);
        case 48: return launch_wide<48>(d, tasks, ntasks, bucket_row, bucket_dest, res, max_lb, wave_mode, rule, st);
        case 64: return launch_wide<64>(d, tasks, ntasks, bucket_row, bucket_dest, res, max_lb, wave_mode, rule, st);
        default: {                                                               // a tier kernel with this one class as its only segment
            PcNwSegment sg; sg.task_begin = 0; sg.ntasks = (uint32_t)ntasks; sg.variant = variant; sg.max_lb = max_lb; sg.compare_only = compare_only; sg.wave_mode = wave_mode;
            return pc_launch_nw_group(&sg, 1, d, tasks, bucket_row, bucket_dest, res, ppos, rule, st);
        }
        }
    }
    // general kernel: one scratch slab of 64 * max_lb cells per resident workgroup
    const size_t per_block = (size_t)64 * (size_t)(max_lb > 0 ? max_lb : 1) * sizeof(int4);
    size_t blocks = scratch ? scratch_bytes / per_block : 0;
    if (blocks == 0) { pc_set_error("pc_launch_nw: general kernel needs %zu bytes of scratch per workgroup", per_block); return PC_ERR_ARG; }
    if (blocks > 1024) blocks = 1024;
    if (blocks > (size_t)ntasks) blocks = (size_t)ntasks;
    hipLaunchKernelGGL(k_nw_general, dim3((unsigned)blocks), dim3(64), 0, st, d, tasks, ntasks, bucket_row, bucket_dest, res,
                       (int4*)scratch, (int64_t)(per_block / sizeof(int4)), ppos, rule);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_nw_general launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}
