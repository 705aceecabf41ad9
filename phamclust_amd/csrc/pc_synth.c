/*
 * pc_synth.c -- deterministic synthetic pham/translation data (host, plain C).
 *
 * The reference's only dataset (benchmark_data.tsv) is a missing large blob
 * (/root/reference/.MISSING_LARGE_BLOBS:1), so every BASELINE.json config runs on
 * synth(N, P, seed) with the 3-column schema of scripts/phamclust.py:21-47.
 * Structure follows SURVEY.md section 8(d): clusters of 40 genomes, a 150-pham pool per
 * cluster, lognormal protein lengths (median 180, clipped to [30,1200]), cluster
 * variants at 30 % substitutions + 2 % indel events, genes at 5 % + 0.5 % on top,
 * 60..140 phams per genome (85 % from the pool), paralog copies 1/2/3 at .94/.05/.01.
 * The PRNG is xoshiro256** seeded by splitmix64, so the data depend on nothing but
 * (N, P, seed).  This is a data generator: not part of the kernels, not the oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix(uint64_t* x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed) { for (int i = 0; i < 4; ++i) r->s[i] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_u64(rng_t* r) {
    uint64_t* s = r->s; uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static inline double rng_unif(rng_t* r) { return (double)(rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t rng_below(rng_t* r, uint32_t n) { return (uint32_t)(((rng_u64(r) >> 32) * (uint64_t)n) >> 32); }
static double rng_normal(rng_t* r) {
    double u1 = rng_unif(r), u2 = rng_unif(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static int rng_geom_mean2(rng_t* r) { int k = 1; while (k < 16 && rng_unif(r) < 0.5) ++k; return k; }

static const char AA[21] = "ACDEFGHIKLMNPQRSTVWY";

typedef struct { uint8_t* p; int64_t len, cap; } buf_t;
static int buf_reserve(buf_t* b, int64_t extra) {
    if (b->len + extra <= b->cap) return 0;
    int64_t nc = b->cap ? b->cap * 2 : (1 << 20);
    while (nc < b->len + extra) nc *= 2;
    uint8_t* np = (uint8_t*)realloc(b->p, (size_t)nc);
    if (!np) return -1;
    b->p = np; b->cap = nc; return 0;
}

/* append mutate(src[0..n)) to out; returns the new length (>= 1) or -1 */
static int mutate(rng_t* r, const uint8_t* src, int n, double sub, double indel, buf_t* out) {
    if (buf_reserve(out, 2 * (int64_t)n + 64) != 0) return -1;
    int64_t start = out->len; int i = 0;
    while (i < n) {
        if (rng_unif(r) < indel) {
            int k = rng_geom_mean2(r);
            if (rng_unif(r) < 0.5) { i += k; continue; }                      /* deletion */
            if (buf_reserve(out, k + 2 * (int64_t)(n - i) + 64) != 0) return -1;
            for (int q = 0; q < k; ++q) out->p[out->len++] = (uint8_t)AA[rng_below(r, 20)];   /* insertion */
        }
        uint8_t c = src[i++];
        if (rng_unif(r) < sub) c = (uint8_t)AA[rng_below(r, 20)];
        out->p[out->len++] = c;
    }
    if (out->len == start) out->p[out->len++] = (uint8_t)AA[rng_below(r, 20)];
    return (int)(out->len - start);
}

typedef struct {
    int32_t n_genomes, n_phams;
    int64_t n_genes, n_residues;
    int64_t* gene_off;     /* [N+1] */
    int32_t* gene_pham;    /* [G] pham number in [0,P), ascending within a genome */
    int64_t* seq_off;      /* [G+1] */
    uint8_t* residues;     /* [R] */
} pcs_data;

static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

void pcs_free(pcs_data* d) {
    if (!d) return;
    free(d->gene_off); free(d->gene_pham); free(d->seq_off); free(d->residues); free(d);
}

pcs_data* pcs_generate(int32_t N, int32_t P, uint64_t seed) {
    if (N <= 0 || P <= 0) return NULL;
    rng_t rng; rng_seed(&rng, seed);
    const int POOL = P < 150 ? P : 150, CL = 40;
    int K = N / CL; if (K < 1) K = 1;

    /* ancestors */
    buf_t anc = {0, 0, 0};
    int64_t* anc_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(P + 1));
    anc_off[0] = 0;
    for (int p = 0; p < P; ++p) {
        double L = exp(log(180.0) + 0.55 * rng_normal(&rng));
        int len = (int)floor(L + 0.5); if (len < 30) len = 30; if (len > 1200) len = 1200;
        buf_reserve(&anc, len);
        for (int i = 0; i < len; ++i) anc.p[anc.len++] = (uint8_t)AA[rng_below(&rng, 20)];
        anc_off[p + 1] = anc.len;
    }
    /* cluster pools + variants */
    int32_t* pool = (int32_t*)malloc(sizeof(int32_t) * (size_t)K * POOL);
    int32_t* in_pool = (int32_t*)malloc(sizeof(int32_t) * (size_t)P);      /* slot in current cluster's pool or -1 */
    int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)P);
    buf_t var = {0, 0, 0};
    int64_t* var_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)K * POOL + 1));
    var_off[0] = 0;
    for (int p = 0; p < P; ++p) perm[p] = p;
    for (int c = 0; c < K; ++c) {
        for (int q = 0; q < POOL; ++q) {                                       /* partial Fisher-Yates */
            int j = q + (int)rng_below(&rng, (uint32_t)(P - q));
            int32_t t = perm[q]; perm[q] = perm[j]; perm[j] = t;
            pool[c * POOL + q] = perm[q];
        }
        for (int q = 0; q < POOL; ++q) {
            int p = pool[c * POOL + q];
            mutate(&rng, anc.p + anc_off[p], (int)(anc_off[p + 1] - anc_off[p]), 0.30, 0.02, &var);
            var_off[c * POOL + q + 1] = var.len;
        }
    }

    pcs_data* d = (pcs_data*)calloc(1, sizeof(pcs_data));
    d->n_genomes = N; d->n_phams = P;
    d->gene_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N + 1));
    int64_t gcap = (int64_t)N * 160 + 16, G = 0;
    d->gene_pham = (int32_t*)malloc(sizeof(int32_t) * (size_t)gcap);
    d->seq_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(gcap + 1));
    buf_t res = {0, 0, 0};
    int32_t* chosen = (int32_t*)malloc(sizeof(int32_t) * 160);
    int32_t* slots = (int32_t*)malloc(sizeof(int32_t) * (size_t)POOL);
    uint8_t* mark = (uint8_t*)calloc((size_t)P, 1);
    for (int p = 0; p < P; ++p) in_pool[p] = -1;
    d->gene_off[0] = 0; d->seq_off[0] = 0;

    for (int g = 0; g < N; ++g) {
        int c = g / CL; if (c > K - 1) c = K - 1;
        for (int q = 0; q < POOL; ++q) in_pool[pool[c * POOL + q]] = q;
        int n = 60 + (int)rng_below(&rng, 81); if (n > P) n = P;
        int n_pool = (int)floor(0.85 * n + 0.5); if (n_pool > POOL) n_pool = POOL;
        int cnt = 0;
        for (int q = 0; q < POOL; ++q) slots[q] = q;
        for (int q = 0; q < n_pool; ++q) {
            int j = q + (int)rng_below(&rng, (uint32_t)(POOL - q));
            int32_t t = slots[q]; slots[q] = slots[j]; slots[j] = t;
            int p = pool[c * POOL + slots[q]];
            if (!mark[p]) { mark[p] = 1; chosen[cnt++] = p; }
        }
        for (int q = n_pool; q < n; ++q) {
            int p = (int)rng_below(&rng, (uint32_t)P);
            if (!mark[p]) { mark[p] = 1; chosen[cnt++] = p; }
        }
        qsort(chosen, (size_t)cnt, sizeof(int32_t), cmp_i32);
        for (int k = 0; k < cnt; ++k) {
            int p = chosen[k]; mark[p] = 0;
            double u = rng_unif(&rng);
            int copies = u < 0.94 ? 1 : (u < 0.99 ? 2 : 3);
            const uint8_t* parent; int plen;
            if (in_pool[p] >= 0) { int64_t v = (int64_t)c * POOL + in_pool[p]; parent = var.p + var_off[v]; plen = (int)(var_off[v + 1] - var_off[v]); }
            else { parent = anc.p + anc_off[p]; plen = (int)(anc_off[p + 1] - anc_off[p]); }
            for (int q = 0; q < copies; ++q) {
                if (G >= gcap) { pcs_free(d); d = NULL; goto done; }
                mutate(&rng, parent, plen, 0.05, 0.005, &res);
                d->gene_pham[G] = p; d->seq_off[G + 1] = res.len; ++G;
            }
        }
        for (int q = 0; q < POOL; ++q) in_pool[pool[c * POOL + q]] = -1;
        d->gene_off[g + 1] = G;
    }
    d->n_genes = G; d->n_residues = res.len; d->residues = res.p; res.p = NULL;
done:
    free(res.p); free(chosen); free(slots); free(mark); free(anc.p); free(anc_off); free(pool); free(in_pool);
    free(perm); free(var.p); free(var_off);
    return d;
}
