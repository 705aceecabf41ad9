/*
 * phamclust_hip.h -- C-ABI of libphamclust_hip.so: the MI355X (gfx950) implementation of
 * phamclust's pairwise genome-similarity matrix fill.
 *
 * The reference (chg60/phamclust, pure Python) has no FFI; the operator boundary this
 * library sits behind is
 *     matrix_de_novo(genomes, func, cpus, as_distance=True) -> SymMatrix
 *                                   /root/reference/src/phamclust/matrix.py:432-497
 *     METRICS = {"gcs","jc","pocp","af","aai","peq"} -> f(source, target, as_distance)
 *                                   /root/reference/src/phamclust/cli.py:30-35
 *                                   /root/reference/src/phamclust/metrics.py:26-253
 * The entry points below are what a ctypes binding for that boundary calls
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only; no exceptions or
 * aborts cross this boundary: every function returns 0 or a negative pc_status and
 * leaves a message for pc_last_error().
 *
 * Threading: one host thread drives one pc_ctx; one pc_ctx drives one GPU.  Multi-GPU, two ways:
 * one process per GPU (one ctx per rank + pc_set_shard*, the exchange is the caller's single
 * RCCL gather of the shard buffers), or one process for all of them (pc_multi_*: the library
 * owns a ctx and a host thread per device and does the exchange itself).
 */
#ifndef PHAMCLUST_HIP_H
#define PHAMCLUST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_VERSION 151   /* 0.5.0: pc_multi_peer_access */

typedef enum {
    PC_OK = 0,
    PC_ERR_ARG = -1,        /* bad argument / inconsistent packed data          */
    PC_ERR_HIP = -2,        /* a HIP runtime call failed                         */
    PC_ERR_STATE = -3,      /* call order (e.g. fill before upload)              */
    PC_ERR_LIMIT = -4,      /* problem exceeds an implementation limit           */
    PC_ERR_DATA = -5        /* data the reference cannot process either (empty translation under aai/peq) */
} pc_status;

/* metric ids follow the order of the reference's METRICS dict (cli.py:30-35) */
typedef enum { PC_GCS = 0, PC_JC = 1, PC_POCP = 2, PC_AF = 3, PC_AAI = 4, PC_PEQ = 5,
               PC_AAI_PPOS = 6   /* average_aminoacid_identity(..., ppos=True), metrics.py:218-220: not a METRICS entry, no CLI route */
} pc_metric;

typedef struct pc_ctx pc_ctx;

/*
 * Packed genomes: host memory, caller-owned, read-only during pc_upload (the library
 * copies what it needs).  Genome order is the name-sorted list order of
 * scripts/phamclust.py:221 and fixes pair orientation: for s < t, s is the reference's
 * `source`, t its `target` (matrix.py:479-486).
 */
typedef struct {
    int32_t n_genomes;          /* N                                                       */
    int32_t n_phams;            /* P                                                       */
    int32_t words_per_row;      /* W = max(1, ceil(P/64))                                  */
    int32_t reserved;           /* must be 0                                               */
    const uint64_t* bitmap;     /* [N*W] bit (p & 63) of word (p >> 6) of row g: g holds pham p  (Genome.phams keys, genome.py:28) */
    const int32_t* nph;         /* [N] len(g.phams)                  (metrics.py:46)        */
    const int32_t* ngen;        /* [N] len(g)                        (genome.py:168-169)    */
    const int64_t* tlen;        /* [N] sum of len(translation)       (metrics.py:135-147)   */
    const int64_t* gene_off;    /* [N+1] genes of genome g = [gene_off[g], gene_off[g+1])   */
    const int32_t* gene_pham;   /* [G] ascending within a genome, paralogs in list order    */
    const int64_t* seq_off;     /* [G+1] residues of gene k = [seq_off[k], seq_off[k+1])    */
    const uint8_t* residues;    /* [R] raw bytes, one per character                         */
} pc_packed;

/* Filled by the fill calls when non-NULL.  Times are HIP-event milliseconds on the
 * stream the kernels ran on. */
typedef struct {
    int64_t n_pairs;            /* genome pairs produced by this call (this rank's shard)   */
    int64_t n_alignments;       /* alignments the reference would run (aai/peq), else 0     */
    int64_t n_cells;            /* sum of la*lb over those alignments                       */
    int64_t n_tasks;            /* wave tasks launched by the alignment kernels             */
    int64_t n_residue_bytes;    /* sum of (la+lb) over alignments: algorithmic input bytes  */
    int32_t n_align_launches;   /* alignment kernel launches                                */
    int32_t n_chunks;           /* plan -> align -> reduce passes the fill took (1 unless the plan exceeded the budget) */
    float ms_total;             /* whole call, device side                                  */
    float ms_plan;              /* pair walk: counts, scans, bucketing, task build          */
    float ms_align;             /* alignment kernels only (the dominant kernels)            */
    float ms_reduce;            /* best-match select + fp64 epilogue (or the set-metric kernel) */
    int64_t n_distinct_alignments; /* distinct (row sequence, column sequence) pairs: what the kernels computed */
    int64_t n_distinct_cells;   /* sum of la*lb over the distinct alignments                */
} pc_stats;

int pc_version(void);
const char* pc_last_error(void);
/* 1 in a library compiled with -DPC_TEST_HOOKS (libphamclust_hip_hooks.so: fault injection for the tests, see the knob table
 * at the end of this header), 0 in the release library. */
int pc_test_hooks(void);

/* One context per GPU.  device_id is the HIP device ordinal. */
int pc_ctx_create(pc_ctx** out, int device_id);
void pc_ctx_destroy(pc_ctx* ctx);

/* Copy the packed genomes to HBM and build the device-side indices (rank table, gene
 * table, encoded residues).  Replaces any previous upload. */
int pc_upload(pc_ctx* ctx, const pc_packed* genomes);

/* The same in two parts.  gcs / jc / pocp / af read pham sets, gene counts and translation LENGTHS only
 * (metrics.py:26-157), and encoding, de-duplicating and ranking the residues is ~90 % of pc_upload's time:
 *   pc_upload_sets      bitmap, rank table, (genome, pham) entries, per-genome scalars -- enough for the four set metrics
 *                       and for pc_set_shard*; replaces any previous upload;
 *   pc_upload_residues  what aai / peq and pc_align_pairs need on top; `genomes` must be the packed genomes
 *                       pc_upload_sets was given (checked by size; the caller keeps them alive in between).  A no-op
 *                       when the residues are already there.
 * aai / peq fills before pc_upload_residues return PC_ERR_STATE. */
int pc_upload_sets(pc_ctx* ctx, const pc_packed* genomes);
int pc_upload_residues(pc_ctx* ctx, const pc_packed* genomes);

/*
 * Static shard of the upper-triangular pair list: rank r of `world` owns the pairs
 * (s, t), s < t, of the target genomes t it is dealt (boustrophedon deal over t, so the
 * linear cost ramp in t balances).  Default after upload: rank 0 of 1 = every pair.
 * pc_shard_pairs: pairs owned; pc_shard_stride: max over ranks (equal-count gather size).
 */
int pc_set_shard(pc_ctx* ctx, int rank, int world);

/* Same contract, cost-balanced deal: one device pass counts the alignment work (DP cells) behind every target genome,
 * then targets go, heaviest first, to the rank with the least work so far.  Deterministic, so every rank of a job
 * arrives at the same partition without communicating; use it on all ranks or on none (pc_assemble_dev follows the
 * deal that is in force).  Worth it when genomes differ in size or in how much they share. */
int pc_set_shard_balanced(pc_ctx* ctx, int rank, int world);
int64_t pc_shard_pairs(const pc_ctx* ctx);
int64_t pc_shard_stride(const pc_ctx* ctx);
/* The deal in force, for callers that assemble (or check) the gathered shards themselves: target genome t belongs to
 * rank t_rank[t] and pair (s, t), s < t, sits at index t_lbase[t] + s of that rank's shard.  Both arrays hold N entries. */
int pc_shard_table(const pc_ctx* ctx, int32_t* t_rank, int64_t* t_lbase);
/* DP cells behind each target genome (sum over s < t), as counted for the cost-balanced deal; N entries.
 * PC_ERR_STATE until pc_set_shard_balanced has run for the current upload. */
int pc_target_costs(const pc_ctx* ctx, uint64_t* cost);

/*
 * matrix_de_novo's fill (matrix.py:479-491) for the six METRICS, whole matrix, one GPU.
 * out_condensed: host f64[N(N-1)/2] in scipy condensed order (row-major strict upper
 * triangle) in packed-genome index space.  Values are already round(x, 6) exactly as
 * the reference returns them; the diagonal is not produced (matrix.py:467-468 presets it).
 */
int pc_fill(pc_ctx* ctx, int metric, int as_distance, double* out_condensed, pc_stats* stats);

/*
 * Memory-bounded batching of aai / peq fills (the reference bounds its in-flight work the same way: ~10,000 pairs per
 * CPU per batch, matrix.py:474-493).  The plan of a fill costs ~56 bytes of HBM per alignment; when that exceeds the
 * budget -- bytes, 0 = automatic: half of the HBM free at the time; environment PC_PLAN_BYTES at context creation -- or
 * the 2^31-2 alignments one plan can index, every fill entry point runs plan -> align -> reduce over successive ranges of
 * its target genomes (pc_stats.n_chunks passes) and produces the same values bit for bit.  A device allocation that fails
 * inside a fill is answered by halving the chunk, not by an error.  Only pc_plan_dev (the alignment-sliced route, which
 * keeps the whole plan resident) still refuses more than 2^31-2 alignments.
 */
int pc_set_plan_budget(pc_ctx* ctx, int64_t bytes);
/* The chunking rule itself, host arithmetic only (no GPU; exported for tests): cut count[0..n) into consecutive ranges
 * whose sums stay <= max_per_chunk (an element above it gets a range of its own).  chunk_begin receives the range starts
 * followed by n (at most cap entries are written); returns the number of ranges. */
int pc_chunk_plan(const uint64_t* count, int n, uint64_t max_per_chunk, int32_t* chunk_begin, int cap);

/* Same values, delivered in page-locked host memory owned by the context (grow-only, pinned once): *out_host points to
 * f64[N(N-1)/2] and stays valid until the next fill or upload on this context, or its destruction.  This is the call
 * matrix_de_novo uses: the result is expanded into the SymMatrix straight away, so nothing outlives the loan. */
int pc_fill_borrow(pc_ctx* ctx, int metric, int as_distance, const double** out_host, pc_stats* stats);

/* Same, result left in HBM: out_dev is a device pointer to f64[N(N-1)/2]; `stream` is a
 * hipStream_t; NULL is the legacy default stream, as everywhere in HIP (PyTorch's default stream has
 * handle 0: work the caller queued there is ordered with these launches).  Asynchronous w.r.t. the host except for
 * one small plan read-back under aai/peq. */
int pc_fill_dev(pc_ctx* ctx, int metric, int as_distance, void* out_dev, void* stream, pc_stats* stats);

/* This rank's shard only, shard-local order, into device memory f64[pc_shard_stride()]
 * (tail beyond pc_shard_pairs() is zero-filled).  Followed by the caller's RCCL gather. */
int pc_fill_shard_dev(pc_ctx* ctx, int metric, int as_distance, void* shard_dev, void* stream, pc_stats* stats);

/* Root only: permute `world` gathered shards (f64[world * pc_shard_stride()], device)
 * into scipy condensed order (device f64[N(N-1)/2]). */
int pc_assemble_dev(pc_ctx* ctx, const void* gathered_dev, int world, void* out_condensed_dev, void* stream);

/*
 * Alignment-sliced multi-GPU route for aai / peq (replaces, like pc_fill_shard_dev + the gather, the joblib fan-out of
 * matrix.py:432-497 -- but splits the ALIGNMENTS, not the genome pairs): every rank holds the same upload, unsharded, and
 *   1. pc_plan_dev        plans the whole fill (identical on every rank; stats->n_distinct_alignments = length of `res`),
 *   2. pc_align_slice_dev aligns every slice_world-th task of each launch class starting at slice_rank and writes
 *                         (n_ident, aln_len) pairs -- 8 bytes per distinct alignment -- into res_dev, zeros elsewhere,
 *   3. the caller sums the ranks' res arrays onto the root (one reduce; as 64-bit integers: exactly one rank contributes
 *      a non-zero entry), and
 *   4. pc_reduce_dev      on the root turns the summed res into the condensed matrix.
 * Every distinct (row sequence, column sequence) pair is aligned once per JOB (the pair-sharded route merges duplicates
 * per rank only) and the ranks' work is equal by construction.  The plan stays valid until the next upload, shard change
 * or fill on the context.  metric: PC_AAI, PC_PEQ or PC_AAI_PPOS.
 */
int pc_plan_dev(pc_ctx* ctx, int metric, void* stream, pc_stats* stats);
int pc_align_slice_dev(pc_ctx* ctx, int slice_rank, int slice_world, void* res_dev, void* stream, pc_stats* stats);
int pc_reduce_dev(pc_ctx* ctx, int metric, int as_distance, const void* res_dev, void* out_condensed_dev, void* stream);

/*
 * One process, several GPUs -- what SURVEY 8(b) specified as pc_ctx_create(out, device_ids, n_dev): the reference spreads the pair
 * list over `cpus` worker processes inside matrix_de_novo (matrix.py:471-493); this spreads it over the GPUs of the node from ONE
 * process, the library owning a context per device and a host thread per device for the length of each call.  Same static shard as
 * the one-process-per-GPU route above (the cost-balanced deal of the target genomes), same single exchange -- every device copies
 * its shard into the root's gather buffer, device to device (peer copies over xGMI) -- same device-side assembly; nothing to launch,
 * no interpreter, framework import or process group per GPU (those cost ~2.5 s per job: profiles/r04/final/launch_cost.txt).
 * device_ids[0] is the root (it delivers the matrix); an id may repeat (several contexts on one GPU: how the tests rehearse this on
 * a one-GPU box).  One host thread calls in.
 *   pc_multi_upload(m, g, with_residues)  every device, in parallel; with_residues = 0: what the four set metrics need
 *   pc_multi_upload_residues(m, g)        what aai / peq need on top (g: the same packed genomes)
 *   pc_multi_fill_borrow(...)             the whole matrix: *out_host -> f64[N(N-1)/2], scipy condensed order, in page-locked memory
 *                                         of the root context, valid until the next fill or upload; stats: NULL or an array of
 *                                         pc_multi_devices(m) entries (each device's own fill); exchange_ms / assemble_ms: NULL or
 *                                         the slowest device's copy to the root / the root's permutation (HIP events)
 */
typedef struct pc_multi pc_multi;
int pc_multi_create(pc_multi** out, const int* device_ids, int n_dev);
void pc_multi_destroy(pc_multi* m);
int pc_multi_devices(const pc_multi* m);
/* How each device's shard will reach the root, decided in pc_multi_create and never hidden: granted[r] (pc_multi_devices() entries,
 * may be NULL) = one of the values below.  Returns the number of devices whose copies are NOT device to device (the runtime stages
 * them through host memory: correct, slower) and leaves their reasons -- the runtime's own error text, one line per device -- for
 * pc_last_error(); 0 when every copy is a peer copy or stays on the root's device.  The reference has no counterpart: its workers
 * return results through joblib's pickling (matrix.py:488-491). */
typedef enum {
    PC_PEER_SAME_DEVICE = 2,    /* the root itself, or another context on the root's GPU (rehearsal): device-to-device copy */
    PC_PEER_ENABLED = 1,        /* hipDeviceEnablePeerAccess granted (or was already on): hipMemcpyPeerAsync over xGMI     */
    PC_PEER_UNAVAILABLE = 0,    /* hipDeviceCanAccessPeer says no                                                         */
    PC_PEER_FAILED = -1         /* the query or the enable call returned an error                                          */
} pc_peer_access;
int pc_multi_peer_access(const pc_multi* m, int32_t* granted);
int pc_multi_upload(pc_multi* m, const pc_packed* genomes, int with_residues);
int pc_multi_upload_residues(pc_multi* m, const pc_packed* genomes);
int pc_multi_set_tie_rule(pc_multi* m, int rule);
int pc_multi_fill_borrow(pc_multi* m, int metric, int as_distance, const double** out_host, pc_stats* stats, float* exchange_ms, float* assemble_ms);

/*
 * Test hook for the alignment kernels (replaces parasail.nw_trace_diag_16 +
 * get_traceback + the two counts read at metrics.py:216-217): aligns gene a_gene[k]
 * (rows, the reference's seq_a) against gene b_gene[k] (columns, seq_b) of the uploaded
 * genomes.  n_ident = comp.count("|"), n_diag = aligned (non-gap) columns, so
 * len(traceback.query) = la + lb - n_diag.  variant: 0 = as pc_fill would choose,
 * -1 = general fallback kernel, w > 0 = force the systolic kernel with w columns per lane
 * (a pair whose COLUMN gene holds a byte outside the 24-letter alphabet runs that variant's
 * residue-compare cell, as in pc_fill: the profile cell keeps one row for all such bytes).
 */
int pc_align_pairs(pc_ctx* ctx, const int32_t* a_gene, const int32_t* b_gene, int64_t n, int variant,
                   int32_t* n_ident, int32_t* n_diag);

/*
 * Tie-rule table of the aligner.  parasail resolves co-optimal alignments by three local rules (SURVEY.md 8c
 * item 4) that are recalled, not pinned (its source is absent); every alignment kernel exists for all 8
 * combinations, selected per context.  Bits: 1 = H prefers INS (E) over DEL (F) when both tie (default DEL first);
 * 2 = E opens when open == extend (default extends); 4 = F opens when open == extend (default extends).
 * Rule 0 is the default (build-time PC_TIE_RULE_DEFAULT, or environment PC_TIE_RULE at context creation).
 * Affects aai / peq only.  Replaces nothing in the reference: it is the handle by which a maintainer holding real
 * parasail vectors re-pins metrics.py:174-175.
 */
#define PC_NUM_TIE_RULES 8
int pc_set_tie_rule(pc_ctx* ctx, int rule);
int pc_get_tie_rule(const pc_ctx* ctx);

/* Test hook: columns per lane of the systolic variant the chooser picks for a column gene of lb residues; 0 = general kernel. */
int pc_variant_width(int lb);

/* Test / tuning hook: HIP-event milliseconds of the alignment kernels of the last pc_align_pairs call. */
float pc_last_align_ms(const pc_ctx* ctx);

/* Which kernel family the selector gave the last gcs / jc / pocp / af fill of this context: 0 popcount tiles, 1 sparse tiles
 * 32 x 32, 2 sparse tiles 64 x 64, 3 shared-pham walker, 4 the column kernel (gcs / jc: target-block masks kept in LDS over a run
 * of source tiles); -1 before the first such fill.  (The selector reads the collection:
 * genomes, bitmap words, phams an average pair shares -- metrics.py:26-157 have one code path, this has four.) */
int pc_last_set_kernel(const pc_ctx* ctx);

/* Test hook: the device implementation of Python's round(x, 6) (the rounding every metric
 * returns through, e.g. metrics.py:50-53) applied to n host doubles in [0, 2^20). */
int pc_round6_probe(pc_ctx* ctx, const double* in, double* out, int64_t n);

/*
 * Environment switches of the library, all of them (the product needs none: defaults are what profiles/ measures).
 *
 *   read at pc_ctx_create
 *     PC_TIE_RULE=0..7          row of the aligner's tie-rule table (pc_set_tie_rule overrides it per context)
 *     PC_PLAN_BYTES=n           HBM one chunk of an aai / peq fill's plan may take (pc_set_plan_budget overrides it)
 *     PC_ALIGN_STREAMS=1..16    streams the alignment launches of a fill are dealt to (8)
 *   read once per process
 *     PC_NO_ROCTX               do not look for libroctx64 (no marker ranges)
 *     PC_UPLOAD_TIMING          per-phase wall times of pc_upload on stderr
 *     PC_RAW_STAGE_MAX=n        largest residue set staged through page-locked memory (512 MB; beyond: copied from the caller's pages)
 *   tuning / A-B switches (read once per process; every setting gives the same matrix, tests/ hold them to that)
 *     PC_TASK_BUDGET=n          cell slots per row stream of an alignment task (49,152)
 *     PC_REMAINDER=0            no narrower variant for a bucket's left-over rows
 *     PC_INC16=0|1              the 11- / 10-instruction DP cell wherever both are compiled
 *     PC_SMALL_MODES=0          no one- / two-wave workgroups for tasks of few rows;  PC_SMALL_LAUNCH_MIN=n  fewest such tasks that get a launch of their own (192)
 *     PC_FUSE=0                 one launch per launch class instead of one per register tier
 *     PC_STRIP=0                column genes beyond 4,096 residues on the one-lane-per-alignment kernel instead of strip-mined passes
 *     PC_STRIP_STREAMS=0        strip-mined launches one after the other on the caller's stream, sharing one scratch region (default: a region and a stream each)
 *     PC_SLAB_BUDGET=n          bytes the strip-mined launches' own scratch regions may take together (3 GB); what does not fit shares the first region, in line
 *     PC_LONG_PRIORITY=0        the strip-mined launches' streams at ordinary priority (default: the highest the device offers)
 *   read per fill / per launch (the tests switch them between calls)
 *     PC_POPC_TILE=32|64        force the word-split 32 x 32 / the 64 x 64 popcount tile kernel
 *     PC_SET_KERNEL=popc|sparse|sparse64|sparsecol|walker    force a kernel family for gcs / jc / pocp / af where it exists for the metric
 *     PC_S64_CHUNKS=n           at least n mask chunks in the 64 x 64 sparse tile kernel
 *     PC_COL_SEG=n              source tiles per unit of the column kernel (gcs / jc; default: by the matrix, at most 8)
 *     PC_PIPE=0|n               strip-mined launches: never pipelined (one row per wave) / always, the passes of a row over n <= 8 waves (default: by the launch's size)
 *   only in libphamclust_hip_hooks.so (compiled with -DPC_TEST_HOOKS; pc_test_hooks() == 1)
 *     PC_FAKE_OOM_ABOVE=n       device allocations above n bytes made while a fill is planning fail (fault injection)
 * The Python package adds PHAMCLUST_DEVICE, PHAMCLUST_DIST_BACKEND, PHAMCLUST_DIST_MODE, PHAMCLUST_DIST_TIMEOUT_S, PHAMCLUST_LAUNCH_COST_S,
 * PHAMCLUST_FORCE_GPUS, PHAMCLUST_NO_TORCH and PHAMCLUST_NATIVE_VARIANT.
 */

#ifdef __cplusplus
}
#endif
#endif /* PHAMCLUST_HIP_H */
