// pc_common.h -- internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PC_OPEN 11            // gap open   (metrics.py:160 default, parasail convention: first gap residue costs OPEN)
#define PC_EXT 1              // gap extend
#define PC_NEG (-(1 << 29))   // "-infinity" that survives a few subtractions without wrapping
#define PC_PADCODE 255        // residue code of padding: never equal to a real code
// Max alignments (row sequences) per workgroup task.  A wave of the systolic kernel keeps at most 64
// rows; rows are dealt to 4*nseg slots, so a wave receives ceil(R / (4 nseg)) * nseg of them: 208 is the
// largest R for which that is <= 64 for every nseg in 1..16.
#define PC_TASK_ROWS 208
#define PC_TASK_BUDGET 49152          // cell slots per row stream of a task (pc_nw_task_rows)
#define PC_MAX_W 64           // widest systolic variant: 64 lanes * 64 columns = 4096 columns

// Device view of the uploaded genomes (all pointers are HBM).
struct PcDev {
    int32_t N, Wb, Wstride, G;       // genomes, bitmap words per row, row stride in u64 (odd), genes
    int32_t U, ubits;                 // distinct gene sequences; bits of a sequence rank (2^ubits >= U)
    int32_t n_cu, pad_;               // compute units of the device these pointers live on (host-side grid sizing)
    int64_t E;                        // (genome, pham) entries
    const uint64_t* bitmap;           // [N][Wstride]
    const uint32_t* rankpre;          // [N][Wb]   entry index of the first set bit of word w of genome g
    const int32_t* ent_cnt;           // [E] genes of the genome in that pham
    const int32_t* ent_len;           // [E] their summed length
    const uint2* ent_pair_len;        // [E] (pham, summed length) and
    const uint2* ent_pair_cnt;        // [E] (pham, gene count): one 8-byte load per entry for k_sparse_tile
    // k_sparse_tile64's lists: entries of phams with at least two holders only, ids renumbered densely (sp_W 64-id words)
    const uint32_t* sp_end;           // [N] a genome's kept entries are [ent_off[g], sp_end[g]) of the three arrays below
    const int32_t* sp_pham;           // [E] dense id
    const uint2* sp_len;              // [E] (dense id, summed length)
    const uint2* sp_cnt;              // [E] (dense id, gene count)
    const uint32_t* sp_rank;          // [N][sp_W] first kept entry of the genome at or after dense word w
    int sp_W;
    const int32_t* ent_gene;          // [E] first gene id (genes of an entry are consecutive)
    const int32_t* ent_pham;          // [E] pham id (ascending within a genome: the set bits of its bitmap row, in order)
    const uint32_t* ent_off;          // [N+1] entries of genome g = [ent_off[g], ent_off[g+1])
    const uint32_t* para_off;         // [N+1] paralog entries of genome g (entries with more than one gene), ascending pham id:
    const int32_t* para_pham;         // [..] their pham id
    const int32_t* para_ex;           // [..] gene count - 1
    const int32_t* gene_len;          // [G]
    const int64_t* gene_off;          // [G] byte offset of the gene's codes (16-byte aligned)
    const uint8_t* codes;             // encoded residues, each gene padded to 16 B with PC_PADCODE
    const int32_t* nph;               // [N]
    const int32_t* ngen;              // [N]
    const int64_t* tlen;              // [N]
    const uint32_t* gene_q;           // [G] rank of the gene's sequence among the distinct sequences, in launch-class order
    const int32_t* q_gene;            // [U] a gene that carries sequence q (its first occurrence)
};

// Static shard: the target genomes this rank owns, ascending.
struct PcShard {
    int32_t nown;
    int32_t ident;                    // 1: owned[k] == k for every k (an unsharded context): kernels skip the table read
    const int32_t* owned;             // [nown] target genome t
    const int64_t* lbase;             // [nown+1] shard-local index of pair (0, owned[k]); pair (s,t) -> lbase[k] + s
};

// One wave task of the alignment kernels: column gene + a range of its bucket.
struct PcTask { int32_t gene, begin, end, pad; };   // pad: launch class of the task (planning only)

// Launch mode of a task, by how many of the workgroup's waves its rows can occupy (pc_nw_task_mode).  A task's launch class is
// base class * PC_WAVE_MODES + mode, so the three modes of a base class are neighbours in the sorted task list and a mode with
// too few tasks for a launch of its own is simply launched together with the one before it.
enum { PC_MODE_CLASS = 0, PC_MODE_TWO_WAVES = 1, PC_MODE_ONE_WAVE = 2, PC_WAVE_MODES = 3 };

// Per distinct column sequence q: how its bucket is cut into tasks.
struct PcTaskPlan {
    const int32_t* task_rows;         // [U] rows per task of the main variant
    const uint8_t* q_class;           // [U] launch class (variant * 4 + lanes-per-segment bucket) of the main tasks
    const uint8_t* q_nseg;            // [U] segments per wave of the main variant
    const uint8_t* rem_class;         // [U][16] launch class for a remainder of r rows (r = n mod nseg), 255: keep them in the main task
    int32_t nvar, small_modes;        // systolic variants; 1: tasks of at most nseg / 2 nseg rows get the one- / two-wave modes
    int32_t n_strip, pad_;            // strip-mined base classes (they follow the 2 x nvar x 4 systolic ones)
    int32_t variant_w[32];            // columns per lane of variant v
};

// walker modes (pc_pairs.hip)
enum { PCW_POCP = 0, PCW_AF = 1, PCW_COUNT = 2, PCW_ENUM = 3, PCW_AAI = 4, PCW_PEQ = 5 };
enum { PCW_SPARSE_GCS = 10, PCW_SPARSE_JC = 11 };   // k_sparse_tile64 only: shared-pham counts, one direction

struct PcWalkArgs {
    // COUNT
    uint32_t* na;                     // [Lp] alignments per pair
    unsigned long long* totals;       // [0] alignments [1] cells [2] residue bytes (as the reference would run them)
    unsigned long long* cost_t;       // [N] or NULL: DP cells per target genome (input of the cost-balanced deal)
    unsigned long long* aln_t;        // [N] or NULL: alignments per target genome (where a fill that exceeds its memory budget is cut)
    // ENUM: alignment slot k of a pair = off[pair] + its position in the reference's loop order
    const uint32_t* off;              // [Lp] exclusive scan of na
    unsigned long long* key;          // [A] (column sequence rank << ubits) | row sequence rank
    uint32_t* val;                    // [A] k
    // AAI / PEQ
    const uint32_t* alias;            // [A] slot -> index of its distinct (row sequence, column sequence) alignment
    const uint2* res;                 // [distinct alignments] (n_ident, aln_len)
    // output (POCP, AF, AAI, PEQ)
    double* out;
    int as_distance;
    int condensed;                    // 1: scipy condensed index, 0: shard-local index
};

const char* pc_hip_err(hipError_t e);
void pc_set_error(const char* fmt, ...);

#define PC_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess) { pc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); return PC_ERR_HIP; } \
    } while (0)

// launchers (defined next to their kernels)
int pc_launch_set_popc(const PcDev& d, const PcShard& sh, int metric, int as_distance, double* out, int condensed,
                       double* lut, bool build_lut, int sh_dim, int tot_dim, hipStream_t st);
int pc_launch_walk(int mode, const PcDev& d, const PcShard& sh, const PcWalkArgs& a, hipStream_t st);
int pc_launch_sparse(int mode, const PcDev& d, const PcShard& sh, double* out, int as_distance, int condensed, hipStream_t st);   // pocp / af
int pc_launch_pair_entries(const int32_t* pham, const int32_t* len, const int32_t* cnt, uint2* pair_len, uint2* pair_cnt, int64_t n, hipStream_t st);
int pc_launch_sp_build(int N, const uint32_t* ent_off, const int32_t* pham, const int32_t* len, const int32_t* cnt, const int32_t* dense, int W2,
                       int32_t* sp_pham, uint2* sp_len, uint2* sp_cnt, uint32_t* sp_rank, uint32_t* sp_end, hipStream_t st);
int pc_launch_sparse64(int mode, const PcDev& d, const PcShard& sh, double* out, int as_distance, int condensed, hipStream_t st); // pocp / af, large matrices
// k_sparse_col (gcs / jc / pocp, large matrices): masks over a block of target genomes kept in LDS across a run of source tiles
size_t pc_sparse_col_lds(int mode, int P64);      // 0: the masks do not fit
int pc_sparse_col_vals_cap(int P64);             // pocp / af: entries (of phams with two holders) a block of 64 targets may hold
int pc_launch_sparse_col(int mode, const PcDev& d, const PcShard& sh, double* out, int as_distance, int condensed, hipStream_t st);
int pc_scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* tmp, int64_t tmp_elems, hipStream_t st);
int64_t pc_scan_tmp_elems(int64_t n);
// residue bytes -> codes on the device (pc_plan.hip): gene k's raw bytes [seq_off[k], seq_off[k+1]) go through the LUT to
// codes + gene_off[k], padded with PC_PADCODE to a multiple of 16
struct PcLut { uint8_t v[256]; };
int pc_launch_encode(const uint8_t* raw, const int64_t* seq_off, const int64_t* gene_off, const int32_t* gene_len, const PcLut& lut,
                     uint8_t* codes, int G, hipStream_t st);
// planning of the alignment batch (pc_plan.hip)
int pc_launch_task_keys(const PcDev& d, const PcTask* tasks, unsigned long long* key, uint32_t* val, int ntasks, hipStream_t st);
int pc_launch_task_gather(const PcTask* in, const uint32_t* idx, PcTask* out, int ntasks, hipStream_t st);
size_t pc_sort_temp_bytes(int64_t n, int bits);
int pc_sort_pairs(void* temp, size_t temp_bytes, const unsigned long long* key_in, unsigned long long* key_out, const uint32_t* val_in,
                  uint32_t* val_out, int64_t n, int bits, hipStream_t st);
int pc_launch_mark_heads(const unsigned long long* skey, uint32_t* flags, int64_t n, hipStream_t st);
int pc_launch_unique(const PcDev& d, const unsigned long long* skey, const uint32_t* sval, const uint32_t* flags, const uint32_t* excl,
                     uint32_t* alias, int32_t* bucket_row, uint32_t* start_q, uint32_t* end_q, unsigned long long* totals, int64_t n, hipStream_t st);
int pc_launch_task_count(const uint32_t* start_q, const uint32_t* end_q, const PcTaskPlan& tp, uint32_t* ntask_q, int U, hipStream_t st);
int pc_launch_task_fill(const PcDev& d, const uint32_t* start_q, const uint32_t* end_q, const PcTaskPlan& tp, const uint32_t* task_off_q,
                        PcTask* tasks, int U, hipStream_t st);
int pc_launch_class_bounds(const unsigned long long* sorted_key, int ntasks, int ncls, uint32_t* cls_begin /*[ncls+1]*/, hipStream_t st);
int pc_launch_task_slice(const PcTask* sorted, int ntasks, const uint32_t* cls_begin, const uint32_t* slice_begin /*[ncls+1]*/, int rank, int world,
                         PcTask* out, hipStream_t st);   // every world-th task of each class from `rank`, compacted
int pc_launch_gather_u32(const uint32_t* src, const int32_t* idx, uint32_t* dst, int n, hipStream_t st);
int pc_launch_assemble(const double* gathered, int world, int64_t stride, int N, double* out, hipStream_t st);
int pc_launch_assemble_table(const double* gathered, int64_t stride, int N, const int32_t* t_rank, const int64_t* t_lbase, double* out, hipStream_t st);
int pc_launch_round6_probe(const double* in, double* out, int64_t n, hipStream_t st);
int pc_launch_unpack_res(const uint2* res, const int32_t* la_plus_lb, int32_t* n_ident, int32_t* n_diag, int64_t n, hipStream_t st);

// alignment kernels (pc_nw.hip)
int pc_nw_num_variants();
int pc_nw_variant_w(int v);                       // columns per lane of variant v
int pc_nw_choose_variant(int lb);                 // -1: general fallback
int pc_nw_g_bucket(int G);                        // lanes-per-segment bucket bound (8, 16, 32, 64) of a launch class
// compare_only: the launch class is the one for column genes holding a byte outside the 24-letter alphabet, which must run
// the residue-compare cell (the profile cell keeps ONE row for all such bytes: exact only while the column holds none)
int pc_nw_class_waves(int variant, int lb, int compare_only);   // waves per workgroup of the launch class a column gene of lb residues falls in
int pc_nw_variant_takes_any_byte(int v);          // 0: the variant has classes that run the profile cell, so such column genes need the compare_only classes
int pc_nw_task_rows(int lb, int variant, int compare_only);     // rows per workgroup task for that column gene
int pc_nw_ppos_systolic(int variant, int max_lb); // 1: a percent-positives launch of this class can run on the systolic kernel (its profile cell)
int pc_nw_ppos_variant(int max_lb);               // the variant to run it on when the class's own cannot; -1: general kernel
int pc_nw_choose_remainder(int lb, int r, int main_variant);   // variant for a bucket's last r < nseg rows, -1: keep them
int pc_nw_task_mode(int lb, int rows, int variant);            // PC_MODE_* of a task of `rows` rows on that variant
int pc_nw_small_modes_enabled();
int pc_launch_nw(int variant, const PcDev& d, const PcTask* tasks, int ntasks, const int32_t* bucket_row,
                 const uint32_t* bucket_dest, uint2* res, void* scratch, size_t scratch_bytes, int max_lb, int ppos, int tie_rule, int compare_only, hipStream_t st,
                 int wave_mode = 0,    // wave_mode: PC_MODE_* -- workgroup shape of the launch's tasks
                 int max_row_len = 65535);   // longest row sequence (sizes a strip-mined launch's boundary lines; see pc_nw_strip_scratch_bytes)
// several launch classes in ONE launch (k_nw_systolic_tier): classes whose pc_nw_fuse_key agrees (and is >= 0) may share it
#define PC_FUSE_MAX_SEGMENTS 32
struct PcNwSegment { uint32_t task_begin, ntasks; int variant, max_lb, compare_only, wave_mode; };
int pc_nw_fuse_key(int variant, int max_lb, int ppos, int compare_only, int wave_mode);
int pc_launch_nw_group(const PcNwSegment* segs, int nsegs, const PcDev& d, const PcTask* task_list, const int32_t* bucket_row,
                       const uint32_t* bucket_dest, uint2* res, int ppos, int tie_rule, hipStream_t st);
size_t pc_nw_strip_scratch_bytes(int max_row_len, int n_cu);    // scratch a strip-mined launch (max_lb > 64 x W of its variant) wants
size_t pc_nw_strip_launch_bytes(int wave_mode, int ntasks, int max_row_len, int n_cu, int ppos);   // ... and one such launch with a region of its own
int pc_nw_launch_is_strip(int variant, int max_lb, int wave_mode, int ppos);   // the launch runs on k_nw_strip and needs the scratch slab
int pc_nw_strip_passes(int lb, int variant);                   // passes of 64 x W columns a column gene of lb residues takes on that variant (1: not strip-mined)
size_t pc_nw_fallback_scratch_bytes(int max_lb);
