/*
 * pc_cooptimal.c -- which alignments are pinned by MATHEMATICS, whatever the aligner's tie-breaking?
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pc_oracle.c: only tests/, tools/ reports, smoke() and bench.py's
 * checker legs may load this).
 *
 * The reference aligns with parasail.nw_trace_diag_16(seq_a, seq_b, 11, 1, blosum62) and reads two numbers off the
 * traceback: len(query) and comp.count("|")  (/root/reference/src/phamclust/metrics.py:160-175, 216-217).  parasail is
 * absent, so WHICH of several equally good alignments it traces is recalled, not pinned.  What needs no
 * recall: a global alignment under "gap of k residues costs open + (k-1)*extend" has a well-defined set of optimal
 * alignments, and every correct Needleman-Wunsch returns a member of that set.  This file computes, per sequence pair,
 *
 *     the optimal score,
 *     the NUMBER of optimal alignments (saturating u64), and
 *     the range [min, max] of n_ident and of n_diag over ALL optimal alignments,
 *
 * with a dynamic programme that shares nothing with pc_oracle.c's aligner: it is the textbook three-state (Gotoh)
 * formulation -- M: last column pairs two residues, X: last column is a gap in the query (consumes a column residue:
 * parasail's E / "INS"), Y: last column is a gap in the reference (parasail's F / "DEL") -- in which an alignment, i.e.
 * a sequence of columns, IS a state path (the state is the type of the last column), so counting optimal state paths
 * counts optimal alignments exactly, with no double counting.  (parasail's H/E/F table is a different bookkeeping of the
 * same optimisation: H = max(M, X, Y).)
 *
 * An alignment is CERTIFIED when min == max for both statistics: then (n_ident, aln_len = la + lb - n_diag) is the same
 * for every optimal alignment and any correct NW -- parasail included -- must report exactly it.  count == 1 (a unique
 * optimum) is the special case the verdict of round 2 asked for; the range form certifies strictly more.
 *
 * Pinned by: brute-force enumeration of every alignment of short pairs (tests/test_oracle.py), which must reproduce score,
 * count and both ranges.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* substitution scores and the byte -> matrix row map are DATA shared with pc_oracle.c (NCBI BLOSUM62) */
extern int pco_blosum62(int a, int b);
extern int pco_map(int byte);
extern int pco_same_byte(int x, int y);           /* the '|' rule in force (SURVEY 8c item 6 and its switch) */
extern int pco_gap_open(void);
extern int pco_gap_extend(void);

#define PCC_NEG (-(1 << 29))

typedef struct {
    int32_t sc;                 /* best score of an alignment of the two prefixes ending in this state */
    int32_t id_lo, id_hi;       /* range of n_ident over the optimal ones */
    int32_t dg_lo, dg_hi;       /* range of n_diag */
    int32_t pad;
    uint64_t cnt;               /* how many there are (saturating); 0 = state unreachable */
} pcc_node;

static inline uint64_t pcc_sat_add(uint64_t x, uint64_t y) { uint64_t s = x + y; return s < x ? UINT64_MAX : s; }

/* fold candidate predecessor `src` (+ds score, +did / +ddg statistics) into acc */
static inline void pcc_take(pcc_node* acc, const pcc_node* src, int32_t ds, int32_t did, int32_t ddg) {
    if (src->cnt == 0) return;
    const int32_t s = src->sc + ds;
    if (acc->cnt == 0 || s > acc->sc) {
        acc->sc = s; acc->cnt = src->cnt;
        acc->id_lo = src->id_lo + did; acc->id_hi = src->id_hi + did;
        acc->dg_lo = src->dg_lo + ddg; acc->dg_hi = src->dg_hi + ddg;
    } else if (s == acc->sc) {
        acc->cnt = pcc_sat_add(acc->cnt, src->cnt);
        if (src->id_lo + did < acc->id_lo) acc->id_lo = src->id_lo + did;
        if (src->id_hi + did > acc->id_hi) acc->id_hi = src->id_hi + did;
        if (src->dg_lo + ddg < acc->dg_lo) acc->dg_lo = src->dg_lo + ddg;
        if (src->dg_hi + ddg > acc->dg_hi) acc->dg_hi = src->dg_hi + ddg;
    }
}

size_t pcc_ws_bytes(int lb) { return sizeof(pcc_node) * 6u * (size_t)(lb + 1); }

/* out: score, count, range[4] = {id_lo, id_hi, dg_lo, dg_hi}.  ppos: count "identical or matrix score > 0" columns
 * (metrics.py:218-220) instead of identical ones. */
int pcc_cooptimal_ws(const uint8_t* a, int la, const uint8_t* b, int lb, int ppos,
                     int32_t* score, uint64_t* count, int32_t* range, void* ws) {
    if (la <= 0 || lb <= 0) return -1;
    const int32_t open = pco_gap_open(), ext = pco_gap_extend();
    pcc_node* M0 = (pcc_node*)ws;          /* row i-1 */
    pcc_node* X0 = M0 + (lb + 1);
    pcc_node* Y0 = X0 + (lb + 1);
    pcc_node* M1 = Y0 + (lb + 1);          /* row i */
    pcc_node* X1 = M1 + (lb + 1);
    pcc_node* Y1 = X1 + (lb + 1);
    const pcc_node none = {PCC_NEG, 0, 0, 0, 0, 0, 0};
    /* row 0: the empty prefix of a against j residues of b -- one alignment, a leading gap of j columns in the query */
    for (int j = 0; j <= lb; ++j) { M0[j] = none; X0[j] = none; Y0[j] = none; }
    M0[0].sc = 0; M0[0].cnt = 1;
    for (int j = 1; j <= lb; ++j) { X0[j].sc = -open - (j - 1) * ext; X0[j].cnt = 1; }
    int8_t srow[256]; uint8_t same[256];
    for (int i = 1; i <= la; ++i) {
        const int ra = pco_map(a[i - 1]);
        for (int v = 0; v < 256; ++v) { srow[v] = (int8_t)pco_blosum62(ra, pco_map(v)); same[v] = (uint8_t)pco_same_byte(a[i - 1], v); }
        M1[0] = none; X1[0] = none; Y1[0] = none;
        Y1[0].sc = -open - (i - 1) * ext; Y1[0].cnt = 1;           /* leading gap of i columns in the reference */
        for (int j = 1; j <= lb; ++j) {
            const uint8_t cb = b[j - 1];
            const int32_t s = srow[cb];
            const int32_t hit = same[cb] || (ppos && s > 0);
            pcc_node m = none, x = none, y = none;
            /* M(i,j): residues a_i, b_j paired after any alignment of the shorter prefixes */
            pcc_take(&m, &M0[j - 1], s, hit, 1); pcc_take(&m, &X0[j - 1], s, hit, 1); pcc_take(&m, &Y0[j - 1], s, hit, 1);
            /* X(i,j): b_j against a gap; extends a gap of the same kind or opens one */
            pcc_take(&x, &M1[j - 1], -open, 0, 0); pcc_take(&x, &X1[j - 1], -ext, 0, 0); pcc_take(&x, &Y1[j - 1], -open, 0, 0);
            /* Y(i,j): a_i against a gap */
            pcc_take(&y, &M0[j], -open, 0, 0); pcc_take(&y, &Y0[j], -ext, 0, 0); pcc_take(&y, &X0[j], -open, 0, 0);
            M1[j] = m; X1[j] = x; Y1[j] = y;
        }
        pcc_node* t;
        t = M0; M0 = M1; M1 = t; t = X0; X0 = X1; X1 = t; t = Y0; Y0 = Y1; Y1 = t;
    }
    pcc_node best = none;
    pcc_take(&best, &M0[lb], 0, 0, 0); pcc_take(&best, &X0[lb], 0, 0, 0); pcc_take(&best, &Y0[lb], 0, 0, 0);
    if (score) *score = best.sc;
    if (count) *count = best.cnt;
    range[0] = best.id_lo; range[1] = best.id_hi; range[2] = best.dg_lo; range[3] = best.dg_hi;
    return 0;
}

int pcc_cooptimal(const uint8_t* a, int la, const uint8_t* b, int lb, int ppos, int32_t* score, uint64_t* count, int32_t* range) {
    if (la <= 0 || lb <= 0) return -1;
    void* ws = malloc(pcc_ws_bytes(lb));
    if (!ws) return -1;
    const int rc = pcc_cooptimal_ws(a, la, b, lb, ppos, score, count, range, ws);
    free(ws);
    return rc;
}

/* gene-index pairs into a packed residue buffer; range: [n][4] */
int pcc_batch(const uint8_t* residues, const int64_t* seq_off, const int32_t* a_idx, const int32_t* b_idx, int64_t n, int ppos,
              int32_t* score, uint64_t* count, int32_t* range, int nthreads) {
    int err = 0;
    (void)pco_map(0);                                   /* table initialisation before the threads start */
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        void* ws = NULL; int ws_lb = -1;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int64_t k = 0; k < n; ++k) {
            const int64_t ao = seq_off[a_idx[k]], bo = seq_off[b_idx[k]];
            const int la = (int)(seq_off[a_idx[k] + 1] - ao), lb = (int)(seq_off[b_idx[k] + 1] - bo);
            if (lb > ws_lb) { free(ws); ws = malloc(pcc_ws_bytes(lb)); ws_lb = lb; }
            if (!ws || pcc_cooptimal_ws(residues + ao, la, residues + bo, lb, ppos, &score[k], &count[k], &range[4 * k], ws) != 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                err = 1;
            }
        }
        free(ws);
    }
    return err ? -1 : 0;
}

/* The alignments the reference runs for the listed genome pairs, in its own loop order (metrics.py:203-214: shared phams --
 * ascending id here --, anchor = the genome with fewer genes in the pham, tie -> source; anchor genes outer, the other
 * genome's genes inner).  a_gene = seq_a (rows), b_gene = seq_b (columns), pair_of = index into the pair list.
 * Returns the number of alignments; writes at most cap of them (call with cap = 0 to size the arrays). */
typedef struct {
    int32_t n_genomes, n_phams, words_per_row, reserved;
    const uint64_t* bitmap; const int32_t* nph; const int32_t* ngen; const int64_t* tlen;
    const int64_t* gene_off; const int32_t* gene_pham; const int64_t* seq_off; const uint8_t* residues;
} pcc_packed;

int64_t pcc_enumerate(const pcc_packed* g, const int32_t* s_idx, const int32_t* t_idx, int64_t n,
                      int32_t* a_gene, int32_t* b_gene, int64_t* pair_of, int64_t cap) {
    int64_t out = 0;
    for (int64_t q = 0; q < n; ++q) {
        int64_t i = g->gene_off[s_idx[q]], ie = g->gene_off[s_idx[q] + 1];
        int64_t j = g->gene_off[t_idx[q]], je = g->gene_off[t_idx[q] + 1];
        while (i < ie && j < je) {
            const int32_t pi = g->gene_pham[i], pj = g->gene_pham[j];
            int64_t i2 = i, j2 = j;
            while (i2 < ie && g->gene_pham[i2] == pi) ++i2;
            while (j2 < je && g->gene_pham[j2] == pj) ++j2;
            if (pi < pj) { i = i2; continue; }
            if (pj < pi) { j = j2; continue; }
            int64_t a0 = i, a1 = i2, b0 = j, b1 = j2;
            if ((i2 - i) > (j2 - j)) { a0 = j; a1 = j2; b0 = i; b1 = i2; }
            for (int64_t ka = a0; ka < a1; ++ka)
                for (int64_t kb = b0; kb < b1; ++kb) {
                    if (out < cap) { a_gene[out] = (int32_t)ka; b_gene[out] = (int32_t)kb; pair_of[out] = q; }
                    ++out;
                }
            i = i2; j = j2;
        }
    }
    return out;
}
