/*
 * pc_pack.c -- TSV -> packed genomes in one pass of plain C (host side, gcc).
 *
 * Reads the reference's input format (scripts/phamclust.py:21-47): one gene per line,
 * `genome<TAB>pham<TAB>translation`, or two columns with the translation defaulting to "M";
 * any other column count is an error.  Produces exactly what phamclust_amd.pack.pack_genomes
 * builds from the equivalent name-sorted list[Genome] (scripts/phamclust.py:221): genomes sorted
 * by name, pham ids = ranks of the sorted unique pham names, genes of a genome sorted by pham id
 * and stable in file order for paralogs.  Names are compared as bytes (== code-point order for
 * UTF-8, which is how Python sorts str).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char* p; int32_t len; } str_t;

typedef struct {
    str_t* keys; int32_t* vals; int64_t cap, n;
} map_t;

static uint64_t hash_bytes(const char* p, int32_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (int32_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ULL; }
    return h;
}
static int map_init(map_t* m, int64_t cap) {
    m->cap = cap; m->n = 0;
    m->keys = (str_t*)calloc((size_t)cap, sizeof(str_t)); m->vals = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
    return (m->keys && m->vals) ? 0 : -1;
}
static void map_free(map_t* m) { free(m->keys); free(m->vals); }
static int map_grow(map_t* m);
/* returns the id of key (inserting next_id when new); *is_new set accordingly */
static int32_t map_get(map_t* m, const char* p, int32_t len, int32_t next_id, int* is_new) {
    if (m->n * 2 >= m->cap && map_grow(m) != 0) return -1;
    uint64_t i = hash_bytes(p, len) & (uint64_t)(m->cap - 1);
    for (;;) {
        if (!m->keys[i].p) { m->keys[i].p = p; m->keys[i].len = len; m->vals[i] = next_id; ++m->n; *is_new = 1; return next_id; }
        if (m->keys[i].len == len && memcmp(m->keys[i].p, p, (size_t)len) == 0) { *is_new = 0; return m->vals[i]; }
        i = (i + 1) & (uint64_t)(m->cap - 1);
    }
}
static int map_grow(map_t* m) {
    map_t b; if (map_init(&b, m->cap * 2) != 0) return -1;
    for (int64_t i = 0; i < m->cap; ++i) if (m->keys[i].p) { int nw; map_get(&b, m->keys[i].p, m->keys[i].len, m->vals[i], &nw); }
    map_free(m); *m = b; return 0;
}

typedef struct {
    int32_t n_genomes, n_phams, words_per_row, status;   /* status: 0 ok, <0 error (see pcp_error) */
    int64_t n_genes, n_residues, names_bytes, pham_names_bytes;
    uint64_t* bitmap; int32_t* nph; int32_t* ngen; int64_t* tlen;
    int64_t* gene_off; int32_t* gene_pham; int64_t* seq_off; uint8_t* residues;
    char* names; int64_t* name_off;              /* genome names joined, [N+1] offsets */
    char* pham_names; int64_t* pham_name_off;    /* pham names joined, [P+1] offsets   */
    char* file;                                  /* the file image the strings point into */
    char error[256];
    int64_t* gene_order;                         /* [G] line rank of packed gene k in the input: what restores a genome's
                                                    own (insertion) order of phams and paralogs (genome.py:31-49) */
} pcp_data;

static int cmp_str(const str_t* a, const str_t* b) {
    int32_t n = a->len < b->len ? a->len : b->len;
    int c = memcmp(a->p, b->p, (size_t)n);
    return c ? c : (a->len > b->len) - (a->len < b->len);
}
static const str_t* g_sort_strs;
static int cmp_idx(const void* x, const void* y) { return cmp_str(&g_sort_strs[*(const int32_t*)x], &g_sort_strs[*(const int32_t*)y]); }

typedef struct { int32_t genome, pham; int64_t order; str_t seq; } gene_t;
static int cmp_gene(const void* x, const void* y) {
    const gene_t* a = (const gene_t*)x; const gene_t* b = (const gene_t*)y;
    if (a->genome != b->genome) return (a->genome > b->genome) - (a->genome < b->genome);
    if (a->pham != b->pham) return (a->pham > b->pham) - (a->pham < b->pham);
    return (a->order > b->order) - (a->order < b->order);
}

void pcp_free(pcp_data* d) {
    if (!d) return;
    free(d->bitmap); free(d->nph); free(d->ngen); free(d->tlen); free(d->gene_off); free(d->gene_pham); free(d->seq_off);
    free(d->residues); free(d->names); free(d->name_off); free(d->pham_names); free(d->pham_name_off); free(d->file); free(d->gene_order); free(d);
}

static pcp_data* fail(pcp_data* d, const char* msg, int64_t line) {
    d->status = -1;
    if (line >= 0) snprintf(d->error, sizeof(d->error), "%s (line %lld)", msg, (long long)line);
    else snprintf(d->error, sizeof(d->error), "%s", msg);
    return d;
}

pcp_data* pcp_load_tsv(const char* path) {
    pcp_data* d = (pcp_data*)calloc(1, sizeof(pcp_data));
    if (!d) return NULL;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(d, "cannot open input file", -1);
    fseek(f, 0, SEEK_END); int64_t size = ftell(f); fseek(f, 0, SEEK_SET);
    d->file = (char*)malloc((size_t)size + 2);
    if (!d->file || (int64_t)fread(d->file, 1, (size_t)size, f) != size) { fclose(f); return fail(d, "cannot read input file", -1); }
    fclose(f);
    d->file[size] = '\n'; d->file[size + 1] = 0;

    map_t gmap, pmap;
    if (map_init(&gmap, 1 << 12) || map_init(&pmap, 1 << 14)) return fail(d, "out of memory", -1);
    int64_t gcap = 1 << 16, ng = 0;
    gene_t* genes = (gene_t*)malloc((size_t)gcap * sizeof(gene_t));
    int64_t sg_cap = 1 << 12, sp_cap = 1 << 14; int32_t n_g = 0, n_p = 0;
    str_t* gnames = (str_t*)malloc((size_t)sg_cap * sizeof(str_t));
    str_t* pnames = (str_t*)malloc((size_t)sp_cap * sizeof(str_t));
    if (!genes || !gnames || !pnames) { free(genes); free(gnames); free(pnames); map_free(&gmap); map_free(&pmap); return fail(d, "out of memory", -1); }
    static const char M[] = "M";
    int64_t lineno = 0;
    for (char* p = d->file; p < d->file + size;) {
        char* e = (char*)memchr(p, '\n', (size_t)(d->file + size + 1 - p));
        ++lineno;
        /* Python's rstrip(): drop trailing whitespace (incl. \r, tabs, spaces) before splitting on tabs */
        char* q = e;
        while (q > p && (q[-1] == ' ' || q[-1] == '\t' || q[-1] == '\r' || q[-1] == '\n' || q[-1] == '\v' || q[-1] == '\f')) --q;
        str_t col[3]; int nc = 0; char* s = p;
        int too_long = 0;
        for (char* c = p; c <= q; ++c) {
            if (c == q || *c == '\t') {
                if (nc < 3) { col[nc].p = s; col[nc].len = (int32_t)(c - s); if (c - s > 0x7fffffff) too_long = 1; }
                ++nc; s = c + 1;
            }
        }
        if (too_long) { free(genes); free(gnames); free(pnames); map_free(&gmap); map_free(&pmap); return fail(d, "a field is longer than 2^31-1 bytes", lineno); }
        if (nc != 2 && nc != 3) { free(genes); free(gnames); free(pnames); map_free(&gmap); map_free(&pmap); return fail(d, "input file must either 2 or 3 columns", lineno); }
        if (nc == 2) { col[2].p = M; col[2].len = 1; }
        int is_new;
        int32_t gi = map_get(&gmap, col[0].p, col[0].len, n_g, &is_new);
        if (gi >= 0 && is_new) { if (n_g >= sg_cap) { sg_cap *= 2; void* nb = realloc(gnames, (size_t)sg_cap * sizeof(str_t)); if (!nb) gi = -1; else gnames = (str_t*)nb; } if (gi >= 0) gnames[n_g++] = col[0]; }
        int32_t pi = gi < 0 ? -1 : map_get(&pmap, col[1].p, col[1].len, n_p, &is_new);
        if (pi >= 0 && is_new) { if (n_p >= sp_cap) { sp_cap *= 2; void* nb = realloc(pnames, (size_t)sp_cap * sizeof(str_t)); if (!nb) pi = -1; else pnames = (str_t*)nb; } if (pi >= 0) pnames[n_p++] = col[1]; }
        if (pi >= 0 && ng >= gcap) { gcap *= 2; void* nb = realloc(genes, (size_t)gcap * sizeof(gene_t)); if (!nb) pi = -1; else genes = (gene_t*)nb; }
        if (gi < 0 || pi < 0) { free(genes); free(gnames); free(pnames); map_free(&gmap); map_free(&pmap); return fail(d, "out of memory", lineno); }
        genes[ng].genome = gi; genes[ng].pham = pi; genes[ng].order = ng; genes[ng].seq = col[2]; ++ng;
        p = e + 1;
    }
    map_free(&gmap); map_free(&pmap);
    if (n_g == 0) { free(genes); free(gnames); free(pnames); return fail(d, "input file holds no genes", -1); }

    /* ranks of sorted names */
    int32_t* gorder = (int32_t*)malloc((size_t)n_g * sizeof(int32_t)); int32_t* grank = (int32_t*)malloc((size_t)n_g * sizeof(int32_t));
    int32_t* porder = (int32_t*)malloc((size_t)n_p * sizeof(int32_t)); int32_t* prank = (int32_t*)malloc((size_t)n_p * sizeof(int32_t));
    for (int32_t i = 0; i < n_g; ++i) gorder[i] = i;
    for (int32_t i = 0; i < n_p; ++i) porder[i] = i;
    g_sort_strs = gnames; qsort(gorder, (size_t)n_g, sizeof(int32_t), cmp_idx);
    g_sort_strs = pnames; qsort(porder, (size_t)n_p, sizeof(int32_t), cmp_idx);
    for (int32_t i = 0; i < n_g; ++i) grank[gorder[i]] = i;
    for (int32_t i = 0; i < n_p; ++i) prank[porder[i]] = i;
    for (int64_t k = 0; k < ng; ++k) { genes[k].genome = grank[genes[k].genome]; genes[k].pham = prank[genes[k].pham]; }
    qsort(genes, (size_t)ng, sizeof(gene_t), cmp_gene);

    const int32_t N = n_g, P = n_p, W = P > 0 ? (P + 63) / 64 : 1;
    d->n_genomes = N; d->n_phams = P; d->words_per_row = W; d->n_genes = ng;
    d->bitmap = (uint64_t*)calloc((size_t)N * W, sizeof(uint64_t));
    d->nph = (int32_t*)calloc((size_t)N, sizeof(int32_t)); d->ngen = (int32_t*)calloc((size_t)N, sizeof(int32_t));
    d->tlen = (int64_t*)calloc((size_t)N, sizeof(int64_t)); d->gene_off = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
    d->gene_pham = (int32_t*)malloc((size_t)(ng ? ng : 1) * sizeof(int32_t)); d->seq_off = (int64_t*)calloc((size_t)ng + 1, sizeof(int64_t));
    d->gene_order = (int64_t*)malloc((size_t)(ng ? ng : 1) * sizeof(int64_t));
    int64_t R = 0;
    for (int64_t k = 0; k < ng; ++k) R += genes[k].seq.len;
    d->n_residues = R; d->residues = (uint8_t*)malloc((size_t)(R ? R : 1));
    int64_t r = 0;
    for (int64_t k = 0; k < ng; ++k) {
        const gene_t* ge = &genes[k];
        d->gene_pham[k] = ge->pham; d->seq_off[k] = r; d->gene_order[k] = ge->order;
        for (int32_t i = 0; i < ge->seq.len; ++i) {
            const unsigned char ch = (unsigned char)ge->seq.p[i];
            if (ch >= 0x80) { free(genes); free(gnames); free(pnames); free(gorder); free(grank); free(porder); free(prank);
                              return fail(d, "non-ASCII byte in a translation: one byte per character is required", -1); }
            d->residues[r + i] = ch;
        }
        r += ge->seq.len;
        uint64_t* row = d->bitmap + (size_t)ge->genome * W;
        const uint64_t bit = 1ULL << (ge->pham & 63);
        if (!(row[ge->pham >> 6] & bit)) { row[ge->pham >> 6] |= bit; ++d->nph[ge->genome]; }
        ++d->ngen[ge->genome]; d->tlen[ge->genome] += ge->seq.len; d->gene_off[ge->genome + 1] = k + 1;
    }
    d->seq_off[ng] = r;
    for (int32_t g = 1; g <= N; ++g) if (d->gene_off[g] < d->gene_off[g - 1]) d->gene_off[g] = d->gene_off[g - 1];
    /* name tables in sorted order */
    d->name_off = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t)); d->pham_name_off = (int64_t*)calloc((size_t)P + 1, sizeof(int64_t));
    for (int32_t i = 0; i < N; ++i) d->name_off[i + 1] = d->name_off[i] + gnames[gorder[i]].len;
    for (int32_t i = 0; i < P; ++i) d->pham_name_off[i + 1] = d->pham_name_off[i] + pnames[porder[i]].len;
    d->names_bytes = d->name_off[N]; d->pham_names_bytes = d->pham_name_off[P];
    d->names = (char*)malloc((size_t)(d->names_bytes ? d->names_bytes : 1)); d->pham_names = (char*)malloc((size_t)(d->pham_names_bytes ? d->pham_names_bytes : 1));
    for (int32_t i = 0; i < N; ++i) memcpy(d->names + d->name_off[i], gnames[gorder[i]].p, (size_t)gnames[gorder[i]].len);
    for (int32_t i = 0; i < P; ++i) memcpy(d->pham_names + d->pham_name_off[i], pnames[porder[i]].p, (size_t)pnames[porder[i]].len);
    free(genes); free(gnames); free(pnames); free(gorder); free(grank); free(porder); free(prank);
    return d;
}

/* ---------------------------------------------------------------------------------------
 * FASTA text of genome g, byte for byte what the reference's Genome.__str__ builds (genome.py:192-199): phams in
 * the order the genome first met them, paralogs in input order, one record per gene:
 *   >name=<genome>|pham=<pham>|n=<1-based index among the paralogs>\n<translation>\n
 * This text is what the pipeline hashes into its cache-directory name (scripts/phamclust.py:86-106) and stashes
 * under 01_genomes/, so the loader can serve both without materialising a Python object per gene.
 * Returns the byte count; writes only when `cap` is large enough (call with cap = 0 to size the buffer).
 * ------------------------------------------------------------------------------------- */
typedef struct { int64_t first_order; int64_t begin, end; } pham_run_t;
static int cmp_run(const void* x, const void* y) {
    const pham_run_t* a = (const pham_run_t*)x; const pham_run_t* b = (const pham_run_t*)y;
    return (a->first_order > b->first_order) - (a->first_order < b->first_order);
}
static int64_t put_dec(char* p, int64_t v) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) p[i] = tmp[n - 1 - i];
    return n;
}
int64_t pcp_genome_fasta(const pcp_data* d, int32_t g, char* out, int64_t cap) {
    if (!d || d->status != 0 || g < 0 || g >= d->n_genomes) return -1;
    const int64_t k0 = d->gene_off[g], k1 = d->gene_off[g + 1];
    const int64_t name_len = d->name_off[g + 1] - d->name_off[g];
    const char* name = d->names + d->name_off[g];
    pham_run_t* runs = (pham_run_t*)malloc((size_t)(k1 - k0 + 1) * sizeof(pham_run_t));
    if (!runs) return -1;
    int64_t nr = 0, need = 0;
    for (int64_t k = k0; k < k1;) {                                    /* genes are sorted by (pham id, input order) */
        int64_t e = k;
        while (e < k1 && d->gene_pham[e] == d->gene_pham[k]) ++e;
        runs[nr].first_order = d->gene_order[k]; runs[nr].begin = k; runs[nr].end = e; ++nr;
        const int64_t plen = d->pham_name_off[d->gene_pham[k] + 1] - d->pham_name_off[d->gene_pham[k]];
        for (int64_t j = k; j < e; ++j) need += 6 + name_len + 6 + plen + 3 + 20 + 1 + (d->seq_off[j + 1] - d->seq_off[j]) + 1;
        k = e;
    }
    if (need > cap || !out) { free(runs); return need; }                /* an upper bound (20 digits reserved per index) */
    qsort(runs, (size_t)nr, sizeof(pham_run_t), cmp_run);
    char* p = out;
    for (int64_t r = 0; r < nr; ++r) {
        const int32_t ph = d->gene_pham[runs[r].begin];
        const char* pname = d->pham_names + d->pham_name_off[ph];
        const int64_t plen = d->pham_name_off[ph + 1] - d->pham_name_off[ph];
        for (int64_t j = runs[r].begin; j < runs[r].end; ++j) {
            memcpy(p, ">name=", 6); p += 6; memcpy(p, name, (size_t)name_len); p += name_len;
            memcpy(p, "|pham=", 6); p += 6; memcpy(p, pname, (size_t)plen); p += plen;
            memcpy(p, "|n=", 3); p += 3; p += put_dec(p, j - runs[r].begin + 1);
            *p++ = '\n';
            const int64_t sl = d->seq_off[j + 1] - d->seq_off[j];
            memcpy(p, d->residues + d->seq_off[j], (size_t)sl); p += sl;
            *p++ = '\n';
        }
    }
    free(runs);
    return (int64_t)(p - out);
}

/* ---------------------------------------------------------------------------------------
 * Matrix text I/O (reference matrix.py:500-670 writes every value with "%.6f"): formatting
 * and parsing of one row at a time, so that N = 10,000 matrices (5*10^7 values, 450 MB of
 * text) do not go through a Python format call per value.  Byte-compatible with "%.6f".
 * ------------------------------------------------------------------------------------- */
#include <math.h>

/* the longest "%.6f" of a double: 1 sign + 309 digits + '.' + 6 decimals (+ NUL) */
#define PCP_FMT_MAX 320

/* "%.6f" of one double.  Fast path: |x| < 1e9 and x*1e6 is within 0.49 of an integer k that the product cannot
 * mis-round (the product's rounding error is ~1e-10 relative) -> print k as d.dddddd; anything else goes
 * through snprintf, so the bytes are always those of "%.6f". */
static char* fmt6(double x, char* p) {
    if (x == x && fabs(x) < 1e9) {
        const double y = fabs(x) * 1e6, k = nearbyint(y);
        if (fabs(y - k) < 0.49) {
            unsigned long long v = (unsigned long long)k;
            char tmp[24]; int n = 0;
            for (int i = 0; i < 6; ++i) { tmp[n++] = (char)('0' + v % 10); v /= 10; }
            tmp[n++] = '.';
            do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
            if (signbit(x)) *p++ = '-';
            while (n) *p++ = tmp[--n];
            return p;
        }
    }
    return p + snprintf(p, PCP_FMT_MAX, "%.6f", x);
}

/* values joined by tabs, terminated by '\n'; returns bytes written, or -1 when `cap` bytes cannot hold them (a value in
 * [0, 1] takes 8 bytes; the check is made before every value against the longest a double can take) */
int64_t pcp_format_row(const double* v, int64_t n, char* out, int64_t cap) {
    char* p = out; char* const end = out + cap;
    for (int64_t i = 0; i < n; ++i) {
        if (end - p < PCP_FMT_MAX + 2) return -1;
        if (i) *p++ = '\t';
        p = fmt6(v[i], p);
    }
    if (end - p < 1) return -1;
    *p++ = '\n';
    return (int64_t)(p - out);
}

/* adjacency lines "source<TAB>target<TAB>value\n" for targets j0..n-1 of one source row (reference matrix.py
 * matrix_to_adjacency); names: concatenated UTF-8, name_off[n+1].  skip_zero drops values equal to 0. */
int64_t pcp_format_adjacency(const char* src, int64_t src_len, const char* names, const int64_t* name_off, const double* row,
                             int64_t j0, int64_t n, int skip_zero, char* out, int64_t cap) {
    char* p = out; char* const end = out + cap;
    for (int64_t j = j0; j < n; ++j) {
        if (skip_zero && row[j] == 0.0) continue;
        const int64_t len = name_off[j + 1] - name_off[j];
        if (end - p < src_len + len + PCP_FMT_MAX + 3) return -1;
        memcpy(p, src, (size_t)src_len); p += src_len; *p++ = '\t';
        memcpy(p, names + name_off[j], (size_t)len); p += len; *p++ = '\t';
        p = fmt6(row[j], p); *p++ = '\n';
    }
    return (int64_t)(p - out);
}

/* tab-separated decimal fields of one line -> doubles (strtod: same values as Python's float()); returns the
 * count parsed, or -(position+1) of the first field that is not a number */
int64_t pcp_parse_row(const char* text, int64_t len, double* out, int64_t cap) {
    int64_t n = 0;
    const char* p = text; const char* end = text + len;
    while (p < end && n < cap) {
        char* q;
        out[n] = strtod(p, &q);
        if (q == p) return -(int64_t)(p - text) - 1;
        ++n; p = q;
        if (p < end && *p == '\t') ++p; else break;
    }
    return n;
}
