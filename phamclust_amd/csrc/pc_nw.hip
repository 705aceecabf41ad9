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
    const int H = max(D, max(c.E, c.F));
    const uint32_t SD = SHd + 0x10000u + (eq ? 1u : 0u);
    c.SH = (H == D) ? SD : ((H == c.F) ? c.SF : c.SE);
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

size_t pc_nw_fallback_scratch_bytes(int max_lb) {
    size_t per_block = (size_t)64 * (size_t)max_lb * sizeof(int4);
    size_t budget = (size_t)2 << 30;
    size_t blocks = budget / (per_block ? per_block : 1);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 32) blocks = 32;
    return blocks * per_block;
}

int pc_nw_num_variants() { return 0; }
int pc_nw_variant_w(int) { return 0; }
int pc_nw_choose_variant(int) { return -1; }

int pc_launch_nw(int variant, const PcDev& d, const PcTask* tasks, int ntasks, const int32_t* bucket_row,
                 const uint32_t* bucket_dest, uint2* res, void* scratch, size_t scratch_bytes, int max_lb, hipStream_t st) {
    if (ntasks <= 0) return PC_OK;
    if (variant >= 0) { pc_set_error("pc_launch_nw: unknown variant %d", variant); return PC_ERR_ARG; }
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
