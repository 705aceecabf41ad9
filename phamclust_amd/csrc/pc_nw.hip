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
// Scores are kept as Ho = H - 11 ("already opened"), which is what both E of the next
// column and F of the next row consume; the substitution profile is biased by +11.
// Stats are one u32: n_ident in the low half, n_diag in the high half, so the diagonal
// update is a single add-with-carry: SD = SHdiag + 0x10000 + (a == b).
// Integer VALU work: no MFMA (there is no dense contraction in this recurrence).
#include "pc_common.h"
#include "../../include/phamclust_hip.h"

// NCBI BLOSUM62 over ARNDCQEGHILKMFPSTWYVBZX* (SURVEY.md section 8c); codes >= 23 use the '*' row.
__constant__ int8_t c_b62[24][24] = {
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

// One DP cell in the Ho convention.  In:  Hol/El/SHl/SEl from (i,j-1), Hou/Fu/SHu/SFu from
// (i-1,j), Hod/SHd from (i-1,j-1), sp = S(a_i,b_j)+11, eq = (a_i == b_j).  Out: Ho,E,F,SH,SE,SF.
struct PcCell { int Ho, E, F; uint32_t SH, SE, SF; };

__device__ __forceinline__ PcCell pc_cell(int Hol, int El, uint32_t SHl, uint32_t SEl, int Hou, int Fu, uint32_t SHu,
                                          uint32_t SFu, int Hod, uint32_t SHd, int sp, bool eq) {
    PcCell c;
    const int Ee = El - PC_EXT, Fe = Fu - PC_EXT;
    c.E = max(Hol, Ee);
    c.SE = (Hol > Ee) ? SHl : SEl;               // strictly greater opens; ties extend
    c.F = max(Hou, Fe);
    c.SF = (Hou > Fe) ? SHu : SFu;
    const int D = Hod + sp;
    const int H = max(max(D, c.E), c.F);
    const uint32_t SD = SHd + 0x10000u + (eq ? 1u : 0u);
    const uint32_t ST = (H == c.F) ? c.SF : c.SE;   // two flat selects: nested ?: becomes control flow
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
                                                   int4* __restrict__ scratch, int64_t scratch_stride) {
    __shared__ int8_t tab[24][24];
    for (int i = threadIdx.x; i < 576; i += 64) tab[i / 24][i % 24] = (int8_t)(c_b62[i / 24][i % 24] + PC_OPEN);
    __syncthreads();
    int4* sc = scratch + (int64_t)blockIdx.x * scratch_stride;
    const int lane = threadIdx.x;
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const PcTask tk = tasks[task];
        const int lb = d.gene_len[tk.gene];
        const uint8_t* bp = d.codes + d.gene_off[tk.gene];
        const int row = tk.begin + lane;
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
                                         trow[min(bj, 23)], ai == bj);
                if (live) sc[(int64_t)j * 64 + lane] = make_int4(c.Ho, c.F, (int)c.SH, (int)c.SF);
                Hod = up.x; SHd = (uint32_t)up.z;
                Hol = c.Ho; El = c.E; SHl = c.SH; SEl = c.SE;
            }
            if (i == la - 1) final_stat = SHl;
        }
        if (active) {
            const uint32_t ident = final_stat & 0xffffu, ndiag = final_stat >> 16;
            res[bucket_dest[row]] = make_uint2(ident, (uint32_t)(la + lb) - ndiag);
        }
    }
}

// ---------------------------------------------------------------------------------
// Systolic kernel (the production path): wavefront-level anti-diagonal sweep.
//
// One wave per task.  The task's column sequence b (lb residues) is cut into strips of W
// columns, one strip per lane; G = ceil(lb/W) consecutive lanes form a segment, and
// nseg = floor(64/G) segments of the same b work side by side on different row
// sequences.  Lane k of a segment keeps the previous row of its W columns (Ho, F and the
// two stats) in registers and at step t processes row t-k of the segment's row stream:
// what it needs from the left neighbour -- Ho, E, stats of that lane's last column, and
// the row's residue -- arrives by DPP wave_shr:1 from the neighbour's previous step, so
// the active cells at any step form an anti-diagonal and no LDS or barrier is involved in
// the recurrence.  The head lane of a segment feeds the stream: for every alignment a
// "virtual row -1" (flag RESET: previous row := -inf, which makes the ordinary recurrence
// produce the boundary H(-1,j) = -(11+j) with zero stats) followed by its la rows, back
// to back, so the pipeline fills once per task, not once per alignment.  The lane holding
// column lb-1 emits (n_ident, aln_len) when a row flagged LAST leaves it.
// Substitution scores come from a per-task profile in LDS: prof[r][k][c] = S(r, b_j)+11,
// one ds_read of W bytes per lane per row.
// ---------------------------------------------------------------------------------
#define PCF_RESET 0x100
#define PCF_LAST 0x200

__device__ __forceinline__ int pc_shr1(int v) {                        // lane k <- lane k-1 (lane 0 keeps 0)
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);  // DPP wave_shr:1
}

template <int W>
__global__ __launch_bounds__(64) void k_nw_systolic(PcDev d, const PcTask* __restrict__ tasks,
                                                    const int32_t* __restrict__ bucket_row,
                                                    const uint32_t* __restrict__ bucket_dest, uint2* __restrict__ res) {
    constexpr int ND = (W + 3) / 4;                 // profile dwords per lane per residue row
    extern __shared__ uint32_t prof[];              // [24][G][ND]
    __shared__ int8_t tab[24][24];
    const int lane = threadIdx.x;
    for (int i = lane; i < 576; i += 64) tab[i / 24][i % 24] = (int8_t)(c_b62[i / 24][i % 24] + PC_OPEN);

    const PcTask tk = tasks[blockIdx.x];
    const int lb = d.gene_len[tk.gene];
    const uint8_t* __restrict__ bp = d.codes + d.gene_off[tk.gene];
    const int G = (lb + W - 1) / W;                 // lanes per segment (<= 64 by variant choice)
    const int nseg = 64 / G;
    const int seg = lane / G, k = lane - seg * G;
    const bool in_seg = seg < nseg;
    const bool is_head = in_seg && k == 0;
    const int k_out = (lb - 1) / W, c_out = (lb - 1) - k_out * W;
    const bool is_out = in_seg && k == k_out;

    int bc[W];
#pragma unroll
    for (int c = 0; c < W; ++c) { const int j = k * W + c; bc[c] = (in_seg && j < lb) ? (int)bp[j] : PC_PADCODE; }
    __syncthreads();
    if (seg == 0) {                                  // segment 0 writes the shared profile
#pragma unroll 1
        for (int r = 0; r < 24; ++r) {
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                uint32_t v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = q * 4 + e;
                    if (c < W) v |= (uint32_t)(uint8_t)tab[r][min(bc[c], 23)] << (8 * e);
                }
                prof[(r * G + k) * ND + q] = v;
            }
        }
    }
    __syncthreads();

    // ---- head-lane stream state ---------------------------------------------------
    int h_row = tk.begin + seg;                      // bucket row of the current alignment
    int h_la = 0, h_i = -1;                          // rows, next row to emit (-1 = virtual row)
    const uint8_t* h_ptr = d.codes;
    int n_la = 0; const uint8_t* n_ptr = d.codes;    // prefetched next alignment
    bool h_live = false;
    int T = 0;
    if (is_head) {
        int L = 0;
        for (int r = h_row; r < tk.end; r += nseg) L += d.gene_len[bucket_row[r]] + 1;
        T = L;
        if (h_row < tk.end) {
            const int ga = bucket_row[h_row]; h_la = d.gene_len[ga]; h_ptr = d.codes + d.gene_off[ga]; h_live = true;
            if (h_row + nseg < tk.end) { const int gn = bucket_row[h_row + nseg]; n_la = d.gene_len[gn]; n_ptr = d.codes + d.gene_off[gn]; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) T = max(T, __shfl_xor(T, o));
    T += G - 1;
    int out_row = tk.begin + seg;                    // out lane: bucket row of the next result

    int Hou[W], Fu[W]; uint32_t SHu[W], SFu[W];
#pragma unroll
    for (int c = 0; c < W; ++c) { Hou[c] = PC_NEG; Fu[c] = PC_NEG; SHu[c] = 0; SFu[c] = 0; }
    int o_a = 0, o_Ho = 0, o_E = PC_NEG; uint32_t o_SH = 0, o_SE = 0;   // my last column, previous step
    int p_Hol = PC_NEG; uint32_t p_SHl = 0;                              // what I received last step (diag of slot 0)
    int a_nxt = 0;                                                        // head: residue byte loaded one step ahead
    if (is_head && h_live) a_nxt = h_ptr[0];

#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        int a = pc_shr1(o_a);
        int Hol = pc_shr1(o_Ho);
        int El = pc_shr1(o_E);
        uint32_t SHl = (uint32_t)pc_shr1((int)o_SH);
        uint32_t SEl = (uint32_t)pc_shr1((int)o_SE);
        if (is_head) {
            El = PC_NEG; SHl = 0; SEl = 0;
            if (!h_live) { a = 0; Hol = 0; }
            else if (h_i < 0) { a = PCF_RESET; Hol = -PC_OPEN; h_i = 0; }                    // H(-1,-1) = 0
            else {
                a = a_nxt | (h_i == h_la - 1 ? PCF_LAST : 0);
                Hol = -(PC_OPEN + h_i * PC_EXT) - PC_OPEN;                                    // H(i,-1)
                ++h_i;
                if (h_i == h_la) {                                                           // advance to the next alignment
                    h_row += nseg; h_i = -1; h_la = n_la; h_ptr = n_ptr; h_live = h_row < tk.end;
                    if (h_row + nseg < tk.end) { const int gn = bucket_row[h_row + nseg]; n_la = d.gene_len[gn]; n_ptr = d.codes + d.gene_off[gn]; }
                }
            }
            if (h_live) a_nxt = h_ptr[h_i < 0 ? 0 : h_i];                                   // prefetch for the next step
        }
        int Hod = p_Hol; uint32_t SHd = p_SHl;
        p_Hol = Hol; p_SHl = SHl;
        if (a & PCF_RESET) {
#pragma unroll
            for (int c = 0; c < W; ++c) { Hou[c] = PC_NEG; Fu[c] = PC_NEG; }
            Hod = PC_NEG;
        }
        const int ac = a & 0xff;
        const uint32_t* pr = prof + (min(ac, 23) * G + k) * ND;
        uint32_t pw[ND];
#pragma unroll
        for (int q = 0; q < ND; ++q) pw[q] = pr[q];
        int E_last = PC_NEG; uint32_t SE_last = 0;
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const int sp = (int)((pw[c >> 2] >> (8 * (c & 3))) & 0xffu);
            const PcCell x = pc_cell(Hol, El, SHl, SEl, Hou[c], Fu[c], SHu[c], SFu[c], Hod, SHd, sp, ac == bc[c]);
            Hod = Hou[c]; SHd = SHu[c];
            Hou[c] = x.Ho; Fu[c] = x.F; SHu[c] = x.SH; SFu[c] = x.SF;
            Hol = x.Ho; El = x.E; SHl = x.SH; SEl = x.SE;
            E_last = x.E; SE_last = x.SE;
        }
        o_a = a; o_Ho = Hol; o_E = E_last; o_SH = SHl; o_SE = SE_last;
        if ((a & PCF_LAST) && is_out) {
            uint32_t st = SHu[0];
#pragma unroll
            for (int c = 1; c < W; ++c) if (c == c_out) st = SHu[c];
            const int la = d.gene_len[bucket_row[out_row]];
            res[bucket_dest[out_row]] = make_uint2(st & 0xffffu, (uint32_t)(la + lb) - (st >> 16));
            out_row += nseg;
        }
    }
}

// columns-per-lane of the compiled systolic variants
static const int g_variant_w[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 18, 20};
static const int g_num_variants = (int)(sizeof(g_variant_w) / sizeof(int));

int pc_nw_num_variants() { return g_num_variants; }
int pc_nw_variant_w(int v) { return (v >= 0 && v < g_num_variants) ? g_variant_w[v] : 0; }

// Variant for a column gene of lb residues: minimise modelled instruction slots per
// alignment row, (19 W + 45) / nseg, over the variants whose 64*W columns cover lb.
int pc_nw_choose_variant(int lb) {
    if (lb <= 0) return -1;
    int best = -1; double best_cost = 0;
    for (int v = 0; v < g_num_variants; ++v) {
        const int W = g_variant_w[v];
        const int G = (lb + W - 1) / W;
        if (G > 64) continue;
        const int nseg = 64 / G;
        const double cost = (19.0 * W + 45.0) / nseg;
        if (best < 0 || cost < best_cost) { best = v; best_cost = cost; }
    }
    return best;
}

template <int W>
static int launch_systolic(const PcDev& d, const PcTask* tasks, int ntasks, const int32_t* bucket_row,
                           const uint32_t* bucket_dest, uint2* res, int max_lb, hipStream_t st) {
    const int ND = (W + 3) / 4;
    int Gmax = (max_lb + W - 1) / W; if (Gmax > 64) Gmax = 64; if (Gmax < 1) Gmax = 1;
    const size_t lds = (size_t)24 * Gmax * ND * 4;
    hipLaunchKernelGGL(k_nw_systolic<W>, dim3((unsigned)ntasks), dim3(64), lds, st, d, tasks, bucket_row, bucket_dest, res);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_nw_systolic<%d> launch: %s", W, hipGetErrorString(e)); return PC_ERR_HIP; }
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
                 const uint32_t* bucket_dest, uint2* res, void* scratch, size_t scratch_bytes, int max_lb, hipStream_t st) {
    if (ntasks <= 0) return PC_OK;
    if (variant >= 0) {
        if (variant >= g_num_variants || max_lb > 64 * g_variant_w[variant]) {
            pc_set_error("pc_launch_nw: variant %d cannot take %d columns", variant, max_lb); return PC_ERR_ARG;
        }
        switch (g_variant_w[variant]) {
#define PC_CASE(WW) case WW: return launch_systolic<WW>(d, tasks, ntasks, bucket_row, bucket_dest, res, max_lb, st);
        PC_CASE(2) PC_CASE(3) PC_CASE(4) PC_CASE(5) PC_CASE(6) PC_CASE(7) PC_CASE(8) PC_CASE(9) PC_CASE(10) PC_CASE(11)
        PC_CASE(12) PC_CASE(13) PC_CASE(14) PC_CASE(16) PC_CASE(18) PC_CASE(20)
#undef PC_CASE
        default: pc_set_error("pc_launch_nw: no kernel for variant %d", variant); return PC_ERR_ARG;
        }
    }
    // general kernel: one scratch slab of 64 * max_lb cells per resident workgroup
    const size_t per_block = (size_t)64 * (size_t)(max_lb > 0 ? max_lb : 1) * sizeof(int4);
    size_t blocks = scratch ? scratch_bytes / per_block : 0;
    if (blocks == 0) { pc_set_error("pc_launch_nw: general kernel needs %zu bytes of scratch per workgroup", per_block); return PC_ERR_ARG; }
    if (blocks > 1024) blocks = 1024;
    if (blocks > (size_t)ntasks) blocks = (size_t)ntasks;
    hipLaunchKernelGGL(k_nw_general, dim3((unsigned)blocks), dim3(64), 0, st, d, tasks, ntasks, bucket_row, bucket_dest, res,
                       (int4*)scratch, (int64_t)(per_block / sizeof(int4)));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_nw_general launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}
