// pc_pairs.hip -- genome-pair kernels: bitset-intersection popcount (gcs/jc), the
// shared-pham walker (pocp/af, alignment planning, best-match reduce for aai/peq),
// prefix sums, task building and shard assembly.
//
// Reference semantics restated here (all /root/reference/src/phamclust/):
//   metrics.py:26-53, 56-80      gcs / jc closed forms on |B[s] & B[t]|
//   metrics.py:83-115, 118-157   pocp / af: sums over shared phams of gene counts / lengths
//   metrics.py:203-227           aai: anchor rule, best match (ties -> last), weighted mean
//   metrics.py:247-253           peq = round(af,6) * round(aai,6), then round(., 6)
//   round(x, 6)                  CPython double_round: exact half-even on the binary value
//
// Data layout: the genomes x phams bitmap is staged tile by tile in LDS (32 source rows
// + 32 target rows, row stride padded to an odd number of u64 so that ds_read_b64 by 32
// lanes of distinct rows is conflict-free); each workgroup owns a 32x32 tile of pairs.
// These kernels are HBM/LDS-bound integer work: no MFMA.
#include "pc_common.h"
#include <type_traits>
#include "../../include/phamclust_hip.h"

#define TS 32          // tile edge (genomes)
#define WCH 32         // bitmap words staged per chunk (17 KB of LDS per workgroup: nine workgroups per CU hide the staging latency)

// ---------------------------------------------------------------------------------
// round(x, 6) exactly as CPython: decimal(x) correctly rounded half-even to 6 places,
// then the nearest double.  x in [0, 2^20).  x*1e6 = M * 15625 * 2^(e+6) exactly.
// ---------------------------------------------------------------------------------
__device__ __noinline__ double pc_round6_exact(double x) {
    if (!(x > 0.0)) return 0.0;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int ex = (int)((bits >> 52) & 0x7ff);
    unsigned long long M = bits & ((1ULL << 52) - 1);
    int e;
    if (ex == 0) e = -1074; else { M |= 1ULL << 52; e = ex - 1075; }
    unsigned long long lo = M * 15625ULL, hi = __umul64hi(M, 15625ULL);   // P = hi:lo < 2^67
    int sh = -(e + 6);
    unsigned long long ip;
    if (sh <= 0) {
        ip = lo << (-sh);                                  // x >= 2^47: out of the documented domain, kept monotone
    } else if (sh >= 68) {
        ip = 0;                                            // x*1e6 < 0.5
    } else if (sh < 64) {
        ip = (sh == 0 ? lo : (lo >> sh)) | (hi << (64 - sh));
        unsigned long long frac = lo & ((1ULL << sh) - 1), half = 1ULL << (sh - 1);
        if (frac > half || (frac == half && (ip & 1))) ++ip;
    } else {
        int s2 = sh - 64;                                  // 0..3
        ip = hi >> s2;
        unsigned long long frac_hi = hi & ((1ULL << s2) - 1), frac_lo = lo;
        unsigned long long half_hi = s2 ? (1ULL << (s2 - 1)) : 0, half_lo = s2 ? 0 : (1ULL << 63);
        bool gt = frac_hi > half_hi || (frac_hi == half_hi && frac_lo > half_lo);
        bool eq = frac_hi == half_hi && frac_lo == half_lo;
        if (gt || (eq && (ip & 1))) ++ip;
    }
    return (double)ip / 1000000.0;
}

// The same value, usually in a dozen instructions.  y = fl(x * 1e6) is within 2^-33 of the exact product for x < 2, so
// when y is not within 2^-30 of a half-integer the exact product rounds (half-even or not: it is no tie) to rint(y), and the
// result is that integer divided by 1e6 -- the very division the exact routine ends with.  Only values that close to a
// decimal tie (two in a million) take the 128-bit integer route.  Every metric's epilogue runs this once or twice per genome
// pair; with the exact routine alone it was most of the sparse pocp / af kernel's time.
// k / 1e6 for an integer k in [0, 2^22], correctly rounded, in three instructions instead of the ~12 of a full fp64 division:
// q0 = k * RN(1e-6) is within an ulp of the quotient, r = k - q0 * 1e6 is exact in one fma, and q0 + r * RN(1e-6) rounds to the
// correctly rounded quotient (Markstein's final step: the divisor is a constant whose reciprocal is correctly rounded).  Held to
// `k / 1000000.0` for EVERY k of that range on the host (tests/test_oracle.py::test_markstein_division_by_a_million, plain C
// arithmetic) and on the device (tests/test_gpu_parity.py::test_round6_every_millionth: round6 of every k / 1e6 is itself).
__device__ __forceinline__ double pc_div_million(double k) {
    const double R = 1.0 / 1000000.0;
    const double q0 = k * R;
    const double r = __builtin_fma(-q0, 1000000.0, k);
    return __builtin_fma(r, R, q0);
}

__device__ __forceinline__ double pc_round6(double x) {
    if (!(x > 0.0)) return 0.0;
    if (x < 2.0) {
        const double y = x * 1.0e6;
        const double k = __builtin_rint(y);
        if (__builtin_fabs(y - k) <= 0.5 - 0x1p-30) return pc_div_million(k);
    }
    return pc_round6_exact(x);
}

__device__ __forceinline__ double pc_finish(double sim, int as_distance) {
    return as_distance ? pc_round6(1.0 - sim) : pc_round6(sim);
}

__global__ void k_round6_probe(const double* in, double* out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pc_round6(in[i]);
}
int pc_launch_round6_probe(const double* in, double* out, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_round6_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}

__device__ __forceinline__ int64_t pc_out_index(const PcDev& d, const PcShard& sh, int s, int t, int k, int condensed) {
    if (condensed) return (int64_t)s * d.N - (int64_t)s * (s + 1) / 2 + (t - s - 1);
    return sh.lbase[k] + s;
}

// XCD-aware tile order for the pair kernels (1-D grids).  The dispatcher hands workgroups to the 8 XCDs round-robin by
// flat id, and every XCD has its own 4 MiB L2.  With a plain 2-D grid each XCD sees tiles from everywhere and streams
// the whole bitmap (plus rank and entry tables) through its L2 again and again: at N = 20,000 the popcount kernel
// fetched 2.4 GB for a 12.6 MB bitmap, the walker 11 GB (profiles/r02/experiments/c_counters.json).  Here tiles are grouped into
// super-tiles of up to 8 x 8 tiles and consecutive workgroups of one XCD walk one super-tile, so the ~100 workgroups
// resident on an XCD share the rows of one or two super-tiles (0.29 GB and 1.3 GB after the change).  Affinity only:
// nothing depends on where a workgroup really runs.
// Super-tile edge: 8 tiles (16 for k_sparse_tile64, whose tiles re-read 800 B of entry lists per row: 0.84 -> 0.56 GB fetched at
// N = 20,000; for the popcount tiles and the walker 16 changed nothing measurable), halved while that would leave an XCD with
// fewer than 16 super-tiles (small matrices must
// still spread over all 8 XCDs; at edge 1 the deal is tile by tile).
__host__ __device__ __forceinline__ unsigned pc_super_edge(unsigned ntx, unsigned nty, unsigned top = 8) {
    unsigned e = top;
    while (e > 1 && ((ntx + e - 1) / e) * ((nty + e - 1) / e) < 128u) e >>= 1;
    return e;
}
// XCD x takes, in super-tile row sy, the columns sx = 8c + ((x - sy) mod 8): every XCD gets every eighth super-tile of
// each row AND of each column, so the triangular (or, for a shard, trapezoid) region of live tiles is dealt evenly --
// dealing whole columns to XCDs left them 40 % apart on the triangle.
__device__ __forceinline__ bool pc_tile_of_index(unsigned n, int ntx, int nty, int& tx, int& ty, unsigned top = 8) {
    const unsigned e = pc_super_edge((unsigned)ntx, (unsigned)nty, top);
    const unsigned xcd = n & 7u, k = n >> 3;
    const unsigned stx = ((unsigned)ntx + e - 1) / e, stx8 = (stx + 7u) / 8u;
    const unsigned m = k / (e * e), within = k % (e * e);
    const unsigned sy = m / stx8, c = m % stx8;
    const unsigned sx = c * 8u + ((xcd + 8u - (sy & 7u)) & 7u);
    tx = (int)(sx * e + within % e);
    ty = (int)(sy * e + within / e);
    return tx < ntx && ty < nty;
}
__device__ __forceinline__ bool pc_tile_of_block(int ntx, int nty, int& tx, int& ty, unsigned top = 8) { return pc_tile_of_index(blockIdx.x, ntx, nty, tx, ty, top); }
static unsigned pc_tile_grid(int ntx, int nty, unsigned top = 8) {
    const unsigned e = pc_super_edge((unsigned)ntx, (unsigned)nty, top);
    const unsigned stx = ((unsigned)ntx + e - 1) / e, sty = ((unsigned)nty + e - 1) / e;
    return sty * ((stx + 7u) / 8u) * 8u * e * e;
}

// Stage one chunk of bitmap words of the tile's 32 source rows and 32 target rows.
__device__ __forceinline__ void pc_stage_tile(const PcDev& d, const PcShard& sh, int s0, int k0, int w0, int wn,
                                              uint64_t (*rs)[WCH + 1], uint64_t (*rt)[WCH + 1]) {
    for (int r = threadIdx.x >> 5; r < TS; r += 8) {
        int s = s0 + r, k = k0 + r;
        const uint64_t* ps = s < d.N ? d.bitmap + (int64_t)s * d.Wstride + w0 : nullptr;
        const uint64_t* pt = k < sh.nown ? d.bitmap + (int64_t)sh.owned[k] * d.Wstride + w0 : nullptr;
        for (int w = threadIdx.x & 31; w < wn; w += 32) {
            rs[r][w] = ps ? ps[w] : 0ULL;
            rt[r][w] = pt ? pt[w] : 0ULL;
        }
    }
}

// ---------------------------------------------------------------------------------
// K1+K3: gcs / jc.  shared = popcount(B[s] & B[t]); fp64 epilogue.  One 256-thread workgroup
// per 64x64 tile of pairs, a 4x4 register tile of pairs per thread: per bitmap word a thread
// reads 4 source-row words and 4 target-row words from LDS (broadcast / conflict-free with
// the odd row stride) and does 16 AND+popcount pairs.  Lanes 0..15 of a 16-lane group hold
// consecutive t, so each store instruction writes 128-byte runs of the condensed output.
// The epilogue value depends only on the two small integers (shared, nph_s + nph_t), so it is
// looked up in a table built once per fill by k_set_lut (exactly the same fp64 code path:
// division, 1 - x, round(., 6)); without a table (huge genomes) it is computed in place.
// ---------------------------------------------------------------------------------
#define PWCH 32        // bitmap words staged per chunk

// shared: |S n T| (gcs, jc) or the conserved gene count sum over shared phams of cnt_s + cnt_t (pocp); tot: nph_s + nph_t, resp. ngen_s + ngen_t
template <int METRIC>
__device__ __forceinline__ double pc_set_value(int shared, int tot, int as_distance) {
    double sim = 0.0;
    if (shared) {
        if (METRIC == PC_GCS) sim = (2.0 * (double)shared) / (double)tot;      // metrics.py:45-48
        else if (METRIC == PC_JC) sim = (double)shared / (double)(tot - shared);   // metrics.py:75
        else sim = (double)shared / (double)tot;                                 // metrics.py:104-110
    }
    return pc_finish(sim, as_distance);
}

// pocp on the popcount kernels (r03).  conserved(s, t) = sum over shared phams of cnt_s + cnt_t = 2 |S n T| + the EXCESS
// counts (cnt - 1) of the shared phams that are paralogs in s or in t -- and only ~6 % of a genome's entries are
// paralogs.  So the kernel counts |S n T| exactly as for jc and, per staged chunk of bitmap words, lets the few paralog
// entries (pham, cnt - 1; ascending pham, so a cursor per row walks them chunk by chunk) of the rows it holds test their
// bit in the opposite rows (LDS) and add their excess.  ONLY >= 0: probe register-tile row ONLY alone (the word-split
// kernel gives each wave one row of each side, so that the four waves do not walk the same lists four times).
// (Staging the tile's lists in LDS first -- 12 packed entries per row, spill path for longer ones -- was built and measured
// slower: N = 20,000 4.54 against 4.29 ms; what costs is the divergence of 16 different rows per wave, not the list reads.)
struct PcParaRow { uint32_t cur, end; };
template <int ROWSTEP, int LDW, int ONLY>
__device__ __forceinline__ void pc_paralog_probe(const PcDev& d, PcParaRow (&st)[4], const uint64_t (*other)[LDW], int other0, int w0, int wn,
                                                 int (&ex)[4][4], bool rows_are_first_index) {
    const int p_end = (w0 + wn) * 64;
    // the four opposite rows as arrays of 32-bit halves: a test is one ds_read_b32 + bit extract + multiply-add
    const uint32_t* half[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) half[j] = (const uint32_t*)&other[other0 + ROWSTEP * j][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (ONLY >= 0 && i != ONLY) continue;
        while (st[i].cur < st[i].end) {
            const int p = d.para_pham[st[i].cur];
            if (p >= p_end) break;
            const int e = d.para_ex[st[i].cur];
            ++st[i].cur;
            const int h = ((p >> 5) - 2 * w0), bit = p & 31;       // which 32-bit half of the staged chunk, which bit of it
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hit = (int)((half[j][h] >> bit) & 1u);
                if (rows_are_first_index) ex[i][j] += e * hit; else ex[j][i] += e * hit;
            }
        }
    }
}

template <int METRIC>
__global__ void k_set_lut(double* __restrict__ lut, int sh_dim, int tot_dim, int as_distance) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sh_dim * tot_dim) return;
    const int tot = i / sh_dim, shared = i - tot * sh_dim;
    // entries with shared > tot/2 (gcs) or shared > tot - shared (jc), conserved > total (pocp) never occur; keep them finite
    const bool possible = METRIC == PC_POCP ? shared <= tot : 2 * shared <= tot;
    lut[i] = possible ? pc_set_value<METRIC>(shared, tot, as_distance) : 0.0;
}

// target genome of shard slot k (an unsharded context owns every genome in order: no table read on the critical path)
__device__ __forceinline__ int pc_owned(const PcShard& sh, int k) { return sh.ident ? k : sh.owned[k]; }

// (the pocp instance takes 166 registers and runs three waves per SIMD where gcs / jc run four; forcing four with
// amdgpu_waves_per_eu spills 40 dwords and costs 20 %: N = 20,000 4.02 -> 4.88 ms)
template <int METRIC>
__global__ __launch_bounds__(256) void k_set_popc(PcDev d, PcShard sh, int as_distance, double* __restrict__ out, int condensed,
                                                   const double* __restrict__ lut, int sh_dim) {
    constexpr int PT = 64;
    __shared__ uint64_t rs[PT][PWCH + 1];
    __shared__ uint64_t rt[PT][PWCH + 1];
    int tile_x, tile_y;
    if (!pc_tile_of_block((d.N + PT - 1) / PT, (sh.nown + PT - 1) / PT, tile_x, tile_y)) return;
    const int s0 = tile_x * PT, k0 = tile_y * PT;
    const int klast = min(k0 + PT, sh.nown) - 1;
    if (s0 >= pc_owned(sh, klast)) return;                   // tile entirely on/below the diagonal
    // 256 threads = 16 (fx) x 16 (fy), a 4x4 register tile of pairs each: per bitmap word a thread reads 4 + 4 row words
    // from LDS for 16 AND+popcount pairs (0.5 LDS reads per pair-word: the loop is VALU-bound -- v_and at 2 clocks and
    // v_bcnt at 4 per wave64, profiles/valu_issue_rate.json -- not LDS-bound).  fx runs along the output's contiguous
    // direction: t (condensed) or s (shard-local), so each store instruction writes 128-byte runs.
    const int fx = threadIdx.x & 15, fy = threadIdx.x >> 4;
    int acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    // staging: thread (r0 = tid>>5, w = tid&31) moves word w of rows r0 + 8p (p < 8) of both tiles; the next chunk's
    // words are fetched into registers while the current chunk is being counted
    const int r0 = threadIdx.x >> 5, wl = threadIdx.x & 31;
    const uint64_t* ps[8]; const uint64_t* pt[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int s = s0 + r0 + 8 * p, k = k0 + r0 + 8 * p;
        ps[p] = s < d.N ? d.bitmap + (int64_t)s * d.Wstride : nullptr;
        pt[p] = k < sh.nown ? d.bitmap + (int64_t)pc_owned(sh, k) * d.Wstride : nullptr;
    }
    uint64_t vs[8], vt[8];
    auto fetch = [&](int w0) {
        const int w = w0 + wl;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            vs[p] = (ps[p] && w < d.Wb) ? ps[p][w] : 0ULL;
            vt[p] = (pt[p] && w < d.Wb) ? pt[p][w] : 0ULL;
        }
    };
    // the rows a thread reads: the slow index takes the tile's s rows under condensed output, its t rows otherwise
    uint64_t (*ra)[PWCH + 1] = condensed ? rs : rt;
    uint64_t (*rb)[PWCH + 1] = condensed ? rt : rs;
    // pocp: the paralog lists of my 4 + 4 rows (rows outside the matrix have empty lists)
    // The b side is probed under TRANSPOSED ownership: this thread walks the lists of b-rows fy + 16 i (the wave's lanes share
    // fy in groups of 16, so a wave sees 4 distinct lists per i, as on the a side -- with its own b-rows fx + 16 i it would see
    // 16, and the loop runs as long as the longest) against a-rows fx + 16 j, and the sums meet their owners through LDS at the end.
    int ex[4][4], ex2[4][4];
    PcParaRow st_a[4], st_b[4];
    auto genome_of = [&](bool a_side, int local) {
        if (a_side == (condensed != 0)) return s0 + local < d.N ? s0 + local : -1;                 // an s row
        return k0 + local < sh.nown ? pc_owned(sh, k0 + local) : -1;                               // a t row
    };
    if (METRIC == PC_POCP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ex[i][j] = ex2[i][j] = 0;
            const int ga = genome_of(true, fy + 16 * i), gb = genome_of(false, fy + 16 * i);
            st_a[i].cur = st_a[i].end = st_b[i].cur = st_b[i].end = 0;
            if (ga >= 0) { st_a[i].cur = d.para_off[ga]; st_a[i].end = d.para_off[ga + 1]; }
            if (gb >= 0) { st_b[i].cur = d.para_off[gb]; st_b[i].end = d.para_off[gb + 1]; }
        }
    }
    fetch(0);
    for (int w0 = 0; w0 < d.Wb; w0 += PWCH) {
        const int wn = min(PWCH, d.Wb - w0);
        if (w0) __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) { rs[r0 + 8 * p][wl] = vs[p]; rt[r0 + 8 * p][wl] = vt[p]; }
        __syncthreads();
        if (w0 + PWCH < d.Wb) fetch(w0 + PWCH);
        if (METRIC == PC_POCP) {
            pc_paralog_probe<16, PWCH + 1, -1>(d, st_a, rb, fx, w0, wn, ex, true);
            pc_paralog_probe<16, PWCH + 1, -1>(d, st_b, ra, fx, w0, wn, ex2, true);        // ex2[i][j]: b-row fy + 16 i, a-row fx + 16 j
        }
        for (int w = 0; w < wn; ++w) {
            uint64_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ra[fy + 16 * i][w];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = rb[fx + 16 * j][w];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // two v_bcnt_u32_b32, each adding into the running count (left to itself the compiler counts into a
                    // temporary and spends a third instruction on the add)
                    const uint64_t x = a[i] & b[j];
                    asm("v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0" : "+v"(acc[i][j]) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)));
                }
        }
    }
    if (METRIC == PC_POCP) {                                         // b-side sums -> the threads that own the pairs (the staging rows are free now)
        __syncthreads();
        int* xt = (int*)&rs[0][0];                                   // [64 b-rows][65]  (16,640 of rs's 16,896 bytes)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) xt[(fy + 16 * i) * 65 + fx + 16 * j] = ex2[i][j];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) ex[i][j] += xt[(fx + 16 * j) * 65 + fy + 16 * i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ls = condensed ? fy + 16 * i : fx + 16 * j;
            const int lt = condensed ? fx + 16 * j : fy + 16 * i;
            const int s = s0 + ls, k = k0 + lt;
            if (s >= d.N || k >= sh.nown) continue;
            const int t = pc_owned(sh, k);
            if (s >= t) continue;
            const int shared = METRIC == PC_POCP ? 2 * acc[i][j] + ex[i][j] : acc[i][j];
            const int tot = METRIC == PC_POCP ? d.ngen[s] + d.ngen[t] : d.nph[s] + d.nph[t];
            const double v = lut ? lut[tot * sh_dim + shared] : pc_set_value<METRIC>(shared, tot, as_distance);
            out[pc_out_index(d, sh, s, t, k, condensed)] = v;
        }
    }
}

// The same for SMALL matrices: 32x32-pair tiles, and the four waves of a workgroup split the bitmap WORDS of the tile
// between them (wave v counts words v, v+4, ... of every chunk for all 32x32 pairs, 4x4 per lane), then add their partial
// counts through LDS and each finishes a quarter of the pairs.  At N = 2,000 (BASELINE configs[1]) the 64x64 kernel is 528
// live workgroups of ~11 us per wave on 256 CUs: two waves on most SIMDs, three on some, and the kernel lasts as long as
// the three (51 us against a 16 us popcount floor; counting alone 32 us, `tools/popc_experiment.sh`).  Here a wave carries
// a quarter of that, so 8,064 of them deal out evenly, and the ~8 waves per SIMD hide each other's LDS and staging waits.
// The staging buffers double as the partial-sum array once the last chunk has been counted.
// (Tried on top: wave-PRIVATE staging -- each wave loads the 20 words it will count for all 64 rows in one burst, no workgroup
// barrier before the final reduction -- 43 KB of LDS instead of 17: N = 2,000 36 -> 43 us, N = 5,000 166 -> 210: the nine
// resident workgroups per CU hide more than the barriers cost.)
template <int METRIC>
__global__ __launch_bounds__(256) void k_set_popc_ksplit(PcDev d, PcShard sh, int as_distance, double* __restrict__ out, int condensed,
                                                          const double* __restrict__ lut, int sh_dim) {
    constexpr int PT = 32;
    __shared__ uint64_t lds[2 * PT * (PWCH + 1)];                   // rs, rt; later int part[4][16][64] (16,384 of its 16,896 bytes)
    uint64_t (*rs)[PWCH + 1] = (uint64_t (*)[PWCH + 1])lds;
    uint64_t (*rt)[PWCH + 1] = (uint64_t (*)[PWCH + 1])(lds + PT * (PWCH + 1));
    int tile_x, tile_y;
    if (!pc_tile_of_block((d.N + PT - 1) / PT, (sh.nown + PT - 1) / PT, tile_x, tile_y)) return;
    const int s0 = tile_x * PT, k0 = tile_y * PT;
    const int klast = min(k0 + PT, sh.nown) - 1;
    if (s0 >= pc_owned(sh, klast)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fx = lane & 7, fy = lane >> 3;
    // the pairs this lane FINISHES: register-tile row `wave`, columns 0..3; their nph are fetched now, off the critical path
    int fin_s[4], fin_k[4], fin_t[4], fin_tot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ls = condensed ? fy + 8 * wave : fx + 8 * j, lt = condensed ? fx + 8 * j : fy + 8 * wave;
        fin_s[j] = s0 + ls; fin_k[j] = k0 + lt;
        const bool ok = fin_s[j] < d.N && fin_k[j] < sh.nown;
        fin_t[j] = ok ? pc_owned(sh, fin_k[j]) : 0;
        fin_tot[j] = !(ok && fin_s[j] < fin_t[j]) ? -1                                          // -1: no such pair
                     : METRIC == PC_POCP ? d.ngen[fin_s[j]] + d.ngen[fin_t[j]] : d.nph[fin_s[j]] + d.nph[fin_t[j]];
    }
    int acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    // staging: thread (r0 = tid>>5, w = tid&31) moves word w of rows r0 + 8p (p < 4) of both tiles
    const int r0 = threadIdx.x >> 5, wl = threadIdx.x & 31;
    const uint64_t* ps[4]; const uint64_t* pt[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int s = s0 + r0 + 8 * p, k = k0 + r0 + 8 * p;
        ps[p] = s < d.N ? d.bitmap + (int64_t)s * d.Wstride : nullptr;
        pt[p] = k < sh.nown ? d.bitmap + (int64_t)pc_owned(sh, k) * d.Wstride : nullptr;
    }
    uint64_t vs[4], vt[4];
    auto fetch = [&](int w0) {
        const int w = w0 + wl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            vs[p] = (ps[p] && w < d.Wb) ? ps[p][w] : 0ULL;
            vt[p] = (pt[p] && w < d.Wb) ? pt[p][w] : 0ULL;
        }
    };
    uint64_t (*ra)[PWCH + 1] = condensed ? rs : rt;
    uint64_t (*rb)[PWCH + 1] = condensed ? rt : rs;
    // pocp: wave v probes register-tile row v of each side (all words of every chunk); the partial sums meet in LDS below
    int ex[4][4];
    PcParaRow st_a[4], st_b[4];
    auto genome_of = [&](bool a_side, int local) {
        if (a_side == (condensed != 0)) return s0 + local < d.N ? s0 + local : -1;
        return k0 + local < sh.nown ? pc_owned(sh, k0 + local) : -1;
    };
    if (METRIC == PC_POCP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ex[i][j] = 0;
            st_a[i].cur = st_a[i].end = st_b[i].cur = st_b[i].end = 0;
        }
        const int ga = genome_of(true, fy + 8 * wave), gb = genome_of(false, fx + 8 * wave);
        PcParaRow ra_st = {0, 0}, rb_st = {0, 0};
        if (ga >= 0) { ra_st.cur = d.para_off[ga]; ra_st.end = d.para_off[ga + 1]; }
        if (gb >= 0) { rb_st.cur = d.para_off[gb]; rb_st.end = d.para_off[gb + 1]; }
        if (wave == 0) { st_a[0] = ra_st; st_b[0] = rb_st; } else if (wave == 1) { st_a[1] = ra_st; st_b[1] = rb_st; }
        else if (wave == 2) { st_a[2] = ra_st; st_b[2] = rb_st; } else { st_a[3] = ra_st; st_b[3] = rb_st; }
    }
    fetch(0);
    for (int w0 = 0; w0 < d.Wb; w0 += PWCH) {
        const int wn = min(PWCH, d.Wb - w0);
        if (w0) __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) { rs[r0 + 8 * p][wl] = vs[p]; rt[r0 + 8 * p][wl] = vt[p]; }
        __syncthreads();
        if (w0 + PWCH < d.Wb) fetch(w0 + PWCH);
        if (METRIC == PC_POCP) {                                     // this wave's words of the chunk: wave, wave + 4, ...
            pc_paralog_probe<8, PWCH + 1, -1>(d, st_a, rb, fx, w0, wn, ex, true);      // (the other three rows' lists are empty)
            pc_paralog_probe<8, PWCH + 1, -1>(d, st_b, ra, fy, w0, wn, ex, false);
        }
#pragma unroll
        for (int q = 0; q < PWCH / 4; ++q) {
            const int w = wave + 4 * q;                              // wave-uniform
            if (w >= wn) break;
            uint64_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ra[fy + 8 * i][w];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = rb[fx + 8 * j][w];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t x = a[i] & b[j];
                    asm("v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0" : "+v"(acc[i][j]) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)));
                }
        }
    }
    __syncthreads();                                                 // every wave is done with the staged words
    int* part = (int*)lds;                                           // [wave][i * 4 + j][lane]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(wave * 16 + i * 4 + j) * 64 + lane] = METRIC == PC_POCP ? 2 * acc[i][j] + ex[i][j] : acc[i][j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int shared = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) shared += part[(v * 16 + wave * 4 + j) * 64 + lane];
        if (fin_tot[j] < 0) continue;
        const double val = lut ? lut[fin_tot[j] * sh_dim + shared] : pc_set_value<METRIC>(shared, fin_tot[j], as_distance);
        out[pc_out_index(d, sh, fin_s[j], fin_t[j], fin_k[j], condensed)] = val;
    }
}

// live 64x64 tiles below which the popcount kernel switches to the word-split 32x32 kernel (measured, jc device time, 64-tile vs
// word-split: N = 1,000 26.7 / 18.8 us, 2,000 53.6 / 35.8, 3,000 74.6 / 69.0, 5,000 157 / 167: gpurun_out r03_popc_exp2 -> profiles/)
#define PC_SMALL_GRID_TILES 1536

#define PC_SET_DISPATCH(KERNEL, GRID)                                                                                                       \
    do {                                                                                                                                   \
        if (metric == PC_GCS) hipLaunchKernelGGL(KERNEL<PC_GCS>, GRID, dim3(256), 0, st, d, sh, as_distance, out, condensed, (const double*)lut, sh_dim);       \
        else if (metric == PC_JC) hipLaunchKernelGGL(KERNEL<PC_JC>, GRID, dim3(256), 0, st, d, sh, as_distance, out, condensed, (const double*)lut, sh_dim);  \
        else hipLaunchKernelGGL(KERNEL<PC_POCP>, GRID, dim3(256), 0, st, d, sh, as_distance, out, condensed, (const double*)lut, sh_dim);                     \
    } while (0)

int pc_launch_set_popc(const PcDev& d, const PcShard& sh, int metric, int as_distance, double* out, int condensed,
                       double* lut, bool build_lut, int sh_dim, int tot_dim, hipStream_t st) {
    if (sh.nown <= 0 || d.N <= 1) return PC_OK;
    if (lut && build_lut) {
        const int n = sh_dim * tot_dim;
        if (metric == PC_GCS) hipLaunchKernelGGL(k_set_lut<PC_GCS>, dim3((n + 255) / 256), dim3(256), 0, st, lut, sh_dim, tot_dim, as_distance);
        else if (metric == PC_JC) hipLaunchKernelGGL(k_set_lut<PC_JC>, dim3((n + 255) / 256), dim3(256), 0, st, lut, sh_dim, tot_dim, as_distance);
        else hipLaunchKernelGGL(k_set_lut<PC_POCP>, dim3((n + 255) / 256), dim3(256), 0, st, lut, sh_dim, tot_dim, as_distance);
    }
    const int64_t tiles64 = (int64_t)((d.N + 63) / 64) * ((sh.nown + 63) / 64);
    const char* force_env = getenv("PC_POPC_TILE");                                         // tuning / test knob: 32 / 64 (read per launch)
    const int force = force_env ? atoi(force_env) : 0;
    const bool small = force ? force == 32 : tiles64 / 2 < PC_SMALL_GRID_TILES;             // about half of the tiles are live
    if (small) {
        dim3 grid(pc_tile_grid((d.N + 31) / 32, (sh.nown + 31) / 32));
        PC_SET_DISPATCH(k_set_popc_ksplit, grid);
    } else {
        dim3 grid(pc_tile_grid((d.N + 63) / 64, (sh.nown + 63) / 64));
        PC_SET_DISPATCH(k_set_popc, grid);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_set_popc launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// ---------------------------------------------------------------------------------
// K2 for SMALL matrices: pocp / af as a SPARSE bitset intersection (r03).  Used below ~3,500 genomes, where it beats the
// shared-pham walker (N = 2,000: 0.122 against 0.184 ms); above, the walker stays (N = 20,000: 5.6 against 7.1 ms --
// where the time goes: `profiles/r03/experiments/c_sparse_tile_experiment.txt`: the divergent per-bit add loops 3.6 ms, mask build + reads
// 2.9 ms, everything else, fp64 epilogue included, 0.7 ms).
//
// A genome holds ~100 of the P = 5,000 phams, a pair shares ~3 of them, and pocp / af need a value per SHARED pham
// (gene count, summed length: metrics.py:102-103, 135-147).  The walker scans all W words of both bitmap rows per pair
// and then chases rank table -> entry table for every hit: 2.8 x the popcount kernel, 3.5 % of HBM speed at N = 20,000.
// Here the work follows the shared phams instead of the words.  One workgroup owns a 32 x 32 tile of pairs and
//   A. transposes the tile's 32 TARGET bitmap rows into LDS: colmask[p] = which of the 32 targets hold pham p (built
//      from the targets' entry lists -- the set bits of their rows -- with LDS atomic ORs),
//   B. lets every entry (p, v) of the 32 SOURCE rows look up colmask[p] and add v into acc[source][target] for each set
//      bit (LDS atomic adds; 64-bit: value in the low 40 bits, a hit count above them, so "no shared pham" stays
//      distinguishable from "shared phams of total value 0"),
//   C. does the same with the roles swapped (colmask over the sources, the targets' entries probe), and
//   D. finishes each pair: (sum_s + sum_t) / (total_s + total_t), 1 - x, round(., 6) in fp64, one coalesced store.
// Per pair that is ~2 x 6 entry visits + ~6 atomic adds + the epilogue instead of 79 word scans + the visits.  Phams are
// processed in chunks of CH (the colmask array is CH words of dynamic LDS), entry ranges of a chunk come from the rank
// table (rankpre is the entry index at every 64-pham word boundary).  No MFMA: there is no dense contraction, and at 2 %
// density a dense one would do 50 x the work.
// ---------------------------------------------------------------------------------
#define SP_T 32                                                   // tile edge (genomes); masks are one u32
#define SP_IT 16                                                  // entries a lane loads per batch (8 lanes per row: 128 entries of a row)
template <int MODE>
__global__ __launch_bounds__(256) void k_sparse_tile(PcDev d, PcShard sh, double* __restrict__ out, int as_distance, int condensed, int CH) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
    uint32_t* colmask = sp_lds;                                                    // [CH]
    unsigned long long* acc = (unsigned long long*)(sp_lds + CH);                  // [32 sources][32 targets]  (CH is even: 8-byte aligned)
    __shared__ int g_s[SP_T], g_t[SP_T];                                           // genome of tile row r, -1: none
    int tile_x, tile_y;
    if (!pc_tile_of_block((d.N + SP_T - 1) / SP_T, (sh.nown + SP_T - 1) / SP_T, tile_x, tile_y)) return;
    const int s0 = tile_x * SP_T, k0 = tile_y * SP_T;
    const int klast = min(k0 + SP_T, sh.nown) - 1;
    if (s0 >= pc_owned(sh, klast)) return;
    if (threadIdx.x < SP_T) {
        const int s = s0 + threadIdx.x, k = k0 + threadIdx.x;
        g_s[threadIdx.x] = s < d.N ? s : -1;
        g_t[threadIdx.x] = k < sh.nown ? pc_owned(sh, k) : -1;
    }
    for (int i = threadIdx.x; i < SP_T * SP_T; i += 256) acc[i] = 0ULL;
    // a row's entries go to 8 consecutive lanes: lane (row = tid >> 3, sub = tid & 7) takes entries sub, sub + 8, ...
    // Every phase first issues ALL its global loads (SP_IT independent loads per lane and array: one memory latency per
    // phase, not one per entry -- with the loads inside the loops a tile took 52 of them back to back and the kernel was
    // slower than the walker), then works on LDS only.
    const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const uint2* __restrict__ ent = MODE == PCW_POCP ? d.ent_pair_cnt : d.ent_pair_len;
    for (int p0 = 0; p0 < d.Wb * 64; p0 += CH) {
        const int w0 = p0 >> 6, w1 = min(d.Wb, (p0 + CH) >> 6);
        __syncthreads();                                                            // g_s, g_t, acc visible / previous chunk done
        uint32_t ebs = 0, ees = 0, ebt = 0, eet = 0;                                // my rows' entries of this chunk
        if (g_s[row] >= 0) {
            ebs = d.rankpre[(int64_t)g_s[row] * d.Wb + w0];
            ees = w1 < d.Wb ? d.rankpre[(int64_t)g_s[row] * d.Wb + w1] : d.ent_off[g_s[row] + 1];
        }
        if (g_t[row] >= 0) {
            ebt = d.rankpre[(int64_t)g_t[row] * d.Wb + w0];
            eet = w1 < d.Wb ? d.rankpre[(int64_t)g_t[row] * d.Wb + w1] : d.ent_off[g_t[row] + 1];
        }
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: masks over the targets, the sources' entries probe; pass 1: the other way round
            const uint32_t bb = pass == 0 ? ebt : ebs, be = pass == 0 ? eet : ees;  // rows that BUILD the masks
            const uint32_t qb = pass == 0 ? ebs : ebt, qe = pass == 0 ? ees : eet;  // rows that PROBE them
            if (pass) __syncthreads();                                              // previous probes done
            for (int i = threadIdx.x * 4; i < CH; i += 1024) *(uint4*)&colmask[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            for (uint32_t e0 = bb + sub; e0 < be; e0 += 8 * SP_IT) {
                int ph[SP_IT];
#pragma unroll
                for (int i = 0; i < SP_IT; ++i) ph[i] = e0 + 8 * i < be ? d.ent_pham[e0 + 8 * i] - p0 : -1;
#pragma unroll
                for (int i = 0; i < SP_IT; ++i) if (ph[i] >= 0) atomicOr(&colmask[ph[i]], 1u << row);
            }
            __syncthreads();
            for (uint32_t e0 = qb + sub; e0 < qe; e0 += 8 * SP_IT) {
                int ph[SP_IT]; uint32_t vv[SP_IT], mm[SP_IT];
#pragma unroll
                for (int i = 0; i < SP_IT; ++i) {                                   // (pham, value) in one 8-byte load
                    const uint2 x = e0 + 8 * i < qe ? ent[e0 + 8 * i] : make_uint2((uint32_t)(p0 - 1), 0u);
                    ph[i] = (int)x.x - p0;
                    vv[i] = x.y;
                }
#pragma unroll
                for (int i = 0; i < SP_IT; ++i) mm[i] = ph[i] >= 0 ? colmask[ph[i]] : 0u;
#pragma unroll
                for (int i = 0; i < SP_IT; ++i) {
                    uint32_t m = mm[i];
                    const unsigned long long v = (1ULL << 40) | (unsigned long long)vv[i];
                    while (m) {
                        const int o = __ffs((int)m) - 1;
                        m &= m - 1;
                        atomicAdd(&acc[pass == 0 ? row * SP_T + o : o * SP_T + row], v);
                    }
                }
            }
        }
    }
    __syncthreads();
    // finish: 1,024 pairs, 4 per thread; consecutive lanes run along the output's contiguous direction
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = threadIdx.x + 256 * q;
        const int fast = idx & 31, slow = idx >> 5;
        const int ls = condensed ? slow : fast, lt = condensed ? fast : slow;
        const int s = g_s[ls], t = g_t[lt];
        if (s < 0 || t < 0 || s >= t) continue;
        const unsigned long long a = acc[ls * SP_T + lt];
        const bool any = (a >> 40) != 0;
        const long long cons = (long long)(a & ((1ULL << 40) - 1));
        double sim = 0.0;
        if (any) {
            if (MODE == PCW_POCP) sim = (double)cons / (double)(d.ngen[s] + d.ngen[t]);    // metrics.py:104-110
            else sim = (double)cons / (double)(d.tlen[s] + d.tlen[t]);                      // metrics.py:149-152
        }
        out[pc_out_index(d, sh, s, t, k0 + lt, condensed)] = pc_finish(sim, as_distance);
    }
}

int pc_launch_sparse(int mode, const PcDev& d, const PcShard& sh, double* out, int as_distance, int condensed, hipStream_t st) {
    if (sh.nown <= 0 || d.N <= 1) return PC_OK;
    // colmask chunk: all phams at once while that leaves three workgroups per CU (48 KB each), else 8,192 at a time
    const int P64 = d.Wb * 64;
    const int CH = P64 <= 10240 ? P64 : 8192;
    const size_t lds = (size_t)CH * 4 + (size_t)SP_T * SP_T * 8;
    dim3 grid(pc_tile_grid((d.N + SP_T - 1) / SP_T, (sh.nown + SP_T - 1) / SP_T)), block(256);
    if (mode == PCW_POCP) hipLaunchKernelGGL(k_sparse_tile<PCW_POCP>, grid, block, lds, st, d, sh, out, as_distance, condensed, CH);
    else hipLaunchKernelGGL(k_sparse_tile<PCW_AF>, grid, block, lds, st, d, sh, out, as_distance, condensed, CH);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_sparse_tile launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// ---------------------------------------------------------------------------------
// K2 for LARGE matrices (r03): the sparse formulation again, on 64 x 64 tiles with row-per-wave probes.
//
// What kept the 32 x 32 kernel behind the walker at N = 20,000 (profiles/r03/experiments/c_sparse_tile_experiment.txt): a tile pays for
// 4 x 32 entry lists (build + probe, both directions) whatever its 1,024 pairs share, and its per-bit add loops diverge --
// most probes of a source row hit no target or one, a few (phams of the target cluster's pool) hit twenty, and a wave
// runs the longest loop of its 64 lanes.  Here
//   * a tile is 64 x 64 pairs (masks are two u32 per pham): the list work per pair halves;
//   * eight waves; a wave owns eight rows of either side and loads ALL their entries (128 per row) into registers with
//     one round of coalesced loads at the start of the tile -- one memory latency per tile, not one per phase (a first
//     version that fetched row after row spent 3.0 of its 7.5 ms waiting for them);
//   * masks are built and probed from those registers; a wave probes ONE row at a time, 64 of its entries per step;
//   * a probe that hits at most two rows of the other side adds them itself (LDS atomics, two short iterations);
//   * a probe that hits more is BROADCAST (readlane): its 64-bit mask becomes the EXEC mask of one v_add into a register
//     the 64 lanes hold for the 64 rows of the other side -- no divergence, no LDS traffic; the register is flushed into
//     the LDS accumulators once per probing row.
//   * an entry is ONE 8-byte load: (pham, value) pairs, made on the device at upload (k_pair_entries); the epilogue's
//     denominators come from LDS (64 + 64 totals per tile);
//   * pocp adds count + 1 where the sources probe and count - 1 where the targets do, so in the second direction only the
//     targets' paralog entries (~6 %) probe at all.
// Accumulators: u32 in LDS, row stride 65 (both directions conflict free).  "No shared pham" is "sum == 0": the host uses
// this kernel only when every entry value is >= 1 (always true for gene counts; for summed lengths unless a translation
// is empty) and every genome's total stays below 2^31 -- otherwise the 32 x 32 kernel / the walker (af) or the popcount
// tiles (pocp) run.
// ---------------------------------------------------------------------------------
#define S6_T 64
#define S6_LD 65
#define S6_WAVES 8
#define S6_RPW (S6_T / S6_WAVES)                                  // rows (of either side) a wave owns
#define S6_GCS PCW_SPARSE_GCS                                     // MODE values beside PCW_POCP / PCW_AF: shared-pham counts only (gcs, jc)
#define S6_JC PCW_SPARSE_JC
#define S6_SUPER 16                                               // super-tile edge in tiles: 2 x 1,024 rows' entry lists = 1.6 MB of an XCD's 4-MB L2
// af: a probe step with at least this many broadcast entries takes them through LDS instead of the readlane loop (below).  Measured
// (profiles/r04/experiments/dense_broadcast_af.txt; af at N = 2,000): threshold 4: 0.120 ms, 8: 0.098, 12: 0.0905, 24: 0.0894; never: 0.127
#ifndef S6_DENSE_MIN
#define S6_DENSE_MIN 24
#endif
#define S6_STAGE_DWORDS 192                                       // per wave: 64 x (mask low, mask high, value)
// Which instances take it: af's two-batch one (small and medium collections, 128 registers).  The one-batch instances sit at 78-79 of
// the 80 registers six waves per SIMD leave and spilled 8-12 dwords with it (scratch stores reach HBM); pocp's two-batch instance
// spilled 24; the counting mode keeps its four workgroups per CU -- no LDS to spare.
__host__ __device__ constexpr bool pc_s6_dense(int mode, int batches) { return S6_DENSE_MIN > 0 && mode == PCW_AF && batches == 2; }
// S6_B: 64-entry batches of a row held in registers -- 2 when one mask chunk holds all phams (a row's ~100 entries), 1 when the
// phams take several chunks (a row then has a few dozen entries per chunk; half the loads and probe steps, and registers for a
// third workgroup per CU: the launch bound asks for six waves per SIMD there)
// Waves per SIMD the instances are compiled for.  The counting mode's one-batch instance takes 59 registers: eight waves per SIMD,
// FOUR workgroups per CU beside 4 x 38.6 KB of LDS (r04: N = 20,000 jc 1.78 -> 1.56 ms -- a tile is a chain of latencies, three
// global-load rounds and five barriers, so what a CU overlaps is what counts).  af / pocp need 78-79 and stay at six (three
// workgroups); what was tried to get them to 64 and was slower or spilled (profiles/r04/experiments/sparse64_occupancy.txt):
// 16-wave tiles with four rows per wave (64 registers with 9-12 dwords of scratch: pocp 2.38 -> 2.87 ms, af 2.58 -> 2.99), one
// register per entry (13 bits of id, 19 of value: the compiler unpacks up front or spills 17-31 dwords under the 64 bound).
__host__ __device__ constexpr int pc_s6_waves_per_simd(int mode, int batches) { return batches != 1 ? 4 : (mode >= PCW_SPARSE_GCS ? 8 : 6); }
template <int MODE, int S6_B>
__global__ __launch_bounds__(64 * S6_WAVES, pc_s6_waves_per_simd(MODE, S6_B)) void k_sparse_tile64(PcDev d, PcShard sh, double* __restrict__ out, int as_distance, int condensed, int CH, unsigned n_units) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
    uint32_t* colmask = sp_lds;                                                    // [CH][2]
    uint32_t* acc = sp_lds + 2 * CH;                                               // [64 sources][65]
    __shared__ int g_s[S6_T], g_t[S6_T];                                           // genome of tile row r, -1: none
    __shared__ long long tot_s[S6_T], tot_t[S6_T];                                 // its total (genes, resp. residues): the epilogue's denominators
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* stage = acc + S6_T * S6_LD + wave * S6_STAGE_DWORDS;                 // af / pocp: this wave's broadcast entries (pc_s6_dense)
    constexpr bool COUNT = MODE >= S6_GCS;                                          // gcs / jc: |S n T| only -- every hit adds 1, and ONE direction does it
    const uint2* __restrict__ ent = MODE == PCW_POCP ? d.sp_cnt : d.sp_len;                   // (dense pham id, value): phams with at least two holders
    // Unit n of the XCD-aware tile order goes to workgroup n mod gridDim (a multiple of 8, so a workgroup keeps to the tiles of
    // its XCD).  Tiles differ in cost by 10 x (a tile inside a cluster of related genomes shares ~85 phams per pair, one between
    // clusters ~3), so the deal must stay fine: measured at N = 20,000 with gridDim = m x the 512 resident workgroups, m = 1: 4.40 ms,
    // 8: 3.73, 64: 3.37 (a workgroup then takes ~3 units, half of them below the diagonal), one workgroup per unit: 3.51;
    // N = 8,000: best at m = 16, again ~3 units each.  Hence gridDim = units / 3.
#pragma unroll 1
    for (unsigned unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    int tile_x, tile_y;
    if (!pc_tile_of_index(unit, (d.N + S6_T - 1) / S6_T, (sh.nown + S6_T - 1) / S6_T, tile_x, tile_y, S6_SUPER)) continue;
    const int s0 = tile_x * S6_T, k0 = tile_y * S6_T;
    const int klast = min(k0 + S6_T, sh.nown) - 1;
    if (s0 >= pc_owned(sh, klast)) continue;
    // lane l looks at row l of either side: genome, then (per chunk) where its entries start and end
    const int gs_l = s0 + lane < d.N ? s0 + lane : -1, gt_l = k0 + lane < sh.nown ? pc_owned(sh, k0 + lane) : -1;
    if (wave == 0) {
        g_s[lane] = gs_l; g_t[lane] = gt_l;
        tot_s[lane] = gs_l < 0 ? 0 : (COUNT ? (long long)d.nph[gs_l] : MODE == PCW_POCP ? (long long)d.ngen[gs_l] : (long long)d.tlen[gs_l]);
        tot_t[lane] = gt_l < 0 ? 0 : (COUNT ? (long long)d.nph[gt_l] : MODE == PCW_POCP ? (long long)d.ngen[gt_l] : (long long)d.tlen[gt_l]);
    }
    for (int i = tid; i < S6_T * S6_LD; i += 64 * S6_WAVES) acc[i] = 0u;
    for (int p0 = 0; p0 < d.sp_W * 64; p0 += CH) {
        const int w0 = p0 >> 6, w1 = min(d.sp_W, (p0 + CH) >> 6);
        // my rows' entries of this chunk, sources and targets: ranges (wave-uniform), then 2 x 64 entries per row in registers
        uint32_t rl_s = 0, rh_s = 0, rl_t = 0, rh_t = 0;
        if (gs_l >= 0) { rl_s = d.sp_rank[(int64_t)gs_l * d.sp_W + w0]; rh_s = w1 < d.sp_W ? d.sp_rank[(int64_t)gs_l * d.sp_W + w1] : d.sp_end[gs_l]; }
        if (gt_l >= 0) { rl_t = d.sp_rank[(int64_t)gt_l * d.sp_W + w0]; rh_t = w1 < d.sp_W ? d.sp_rank[(int64_t)gt_l * d.sp_W + w1] : d.sp_end[gt_l]; }
        uint32_t lo_s[S6_RPW], hi_s[S6_RPW], lo_t[S6_RPW], hi_t[S6_RPW];
#pragma unroll
        for (int rr = 0; rr < S6_RPW; ++rr) {
            const int r = wave + S6_WAVES * rr;
            lo_s[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rl_s, r); hi_s[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rh_s, r);
            lo_t[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rl_t, r); hi_t[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rh_t, r);
        }
        // pocp keeps BOTH sides' gene counts of slot (rr, b) in one register (source's count below, target's above bit 16: the host
        // takes this kernel for pocp only while every genome holds fewer than 65,536 genes).  With a register each, the one-batch
        // instance needed 84 against the 80 that six waves per SIMD leave: three dwords went to scratch, and scratch stores reach
        // HBM -- the 23 % of writes beyond the matrix that the r03 counters showed for pocp alone (WRITE_SIZE 1.97 GB for 1.60 GB)
        constexpr bool PACKED = MODE == PCW_POCP;
        int ph_s[S6_RPW][S6_B], ph_t[S6_RPW][S6_B]; uint32_t v_s[S6_RPW][S6_B], v_t[PACKED ? 1 : S6_RPW][PACKED ? 1 : S6_B];
#pragma unroll
        for (int rr = 0; rr < S6_RPW; ++rr)
#pragma unroll
            for (int b = 0; b < S6_B; ++b) {
                const uint32_t es = lo_s[rr] + (uint32_t)(64 * b + lane), et = lo_t[rr] + (uint32_t)(64 * b + lane);
                const bool is = es < hi_s[rr], it = et < hi_t[rr];
                if constexpr (COUNT) {
                    ph_s[rr][b] = is ? d.sp_pham[es] - p0 : -1; v_s[rr][b] = 1u;
                    ph_t[rr][b] = it ? d.sp_pham[et] - p0 : -1; v_t[rr][b] = 1u;
                } else {
                    const uint2 xs = is ? ent[es] : make_uint2((uint32_t)(p0 - 1), 0u), xt = it ? ent[et] : make_uint2((uint32_t)(p0 - 1), 0u);
                    ph_s[rr][b] = (int)xs.x - p0; ph_t[rr][b] = (int)xt.x - p0;
                    if constexpr (PACKED) v_s[rr][b] = xs.y | (xt.y << 16);
                    else { v_s[rr][b] = xs.y; v_t[rr][b] = xt.y; }
                }
            }
        // one direction: the rows of one side build the masks, the rows of the other probe them.  TO_ROW: the probing rows are
        // the accumulator rows (sources probe), else its columns (targets probe)
        auto hit = [&](auto to_row, auto packed, int r, int ph, uint32_t v, uint32_t& hs) {
            // pocp: the entry's gene count c adds c + 1 where the sources probe and c - 1 where the targets do (see below)
            if constexpr (MODE == PCW_POCP) {
                if constexpr (decltype(packed)::value) v = decltype(to_row)::value ? (v & 0xffffu) : (v >> 16);
                v = decltype(to_row)::value ? v + 1u : v - 1u;
            }
            uint2 m = make_uint2(0u, 0u);
            if (ph >= 0) m = *(const uint2*)&colmask[2 * ph];
            const int pc = __popc(m.x) + __popc(m.y);
            if (pc > 0 && pc <= 2) {                                                // one or two hits: this lane adds them
                const unsigned long long mm = ((unsigned long long)m.y << 32) | m.x;
                const int o1 = __builtin_ctzll(mm), o2 = 63 - __builtin_clzll(mm);
                atomicAdd(&acc[decltype(to_row)::value ? r * S6_LD + o1 : o1 * S6_LD + r], v);
                if (pc == 2) atomicAdd(&acc[decltype(to_row)::value ? r * S6_LD + o2 : o2 * S6_LD + r], v);
            }
            unsigned long long heavy = __ballot(pc > 2);                            // many hits: the wave adds them, lanes = rows of the other side
            if constexpr (pc_s6_dense(MODE, S6_B)) {
                // Inside a cluster nearly every entry is such a hit (a pair shares ~85 phams), and the loop below takes ~10 instructions
                // and a VALU -> SGPR -> EXEC round trip per entry: the launch lasted as long as its slowest in-cluster tile.  Dense
                // form (r04): the 64 entries go to LDS (mask halves and value; zero where the lane's entry is not a broadcast one), and
                // lane l walks all 64 -- its own half of the mask by a broadcast read -- adding the value where its bit is set: two LDS
                // reads and two VALU instructions per entry, no scalar dependency, iterations independent.
                if (__popcll(heavy) >= S6_DENSE_MIN) {
                    const bool big = pc > 2;
                    stage[lane] = big ? m.x : 0u; stage[64 + lane] = big ? m.y : 0u; stage[128 + lane] = v;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the wave's own LDS writes -> its reads (in-order LDS queue)
                    uint32_t l2 = (uint32_t)lane;
                    asm volatile("" : "+v"(l2));                                    // (derived per step: hoisted out of the unrolled rows, `mine` and `sh` cost the one-batch instances 8-12 dwords of scratch)
                    const uint32_t* mine = stage + (l2 & 32u) * 2u;                 // lanes 0-31: low halves, 32-63: high halves
                    const uint32_t sh = l2 & 31u;
#pragma unroll 8
                    for (int k = 0; k < 64; ++k) hs += ((mine[k] >> sh) & 1u) * stage[128 + k];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // reads done before the next step rewrites the stage
                    heavy = 0;
                }
            }
            while (heavy) {
                const int k = __builtin_ctzll(heavy);
                asm("s_bitset0_b64 %0, %1" : "+s"(heavy) : "s"(k));
                const unsigned long long mk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)m.y, k) << 32) |
                                              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)m.x, k);
                const uint32_t vk = (uint32_t)__builtin_amdgcn_readlane((int)v, k);
                // (every lane of the workgroup is active here -- 512 threads, wave-uniform control flow -- so EXEC is all ones before and after)
                asm volatile("s_mov_b64 exec, %1\n\tv_add_u32 %0, %0, %2\n\ts_mov_b64 exec, -1" : "+v"(hs) : "s"(mk), "s"(vk));
            }
        };
        auto direction = [&](auto to_row, const int (&bph)[S6_RPW][S6_B], const uint32_t (&blo)[S6_RPW], const uint32_t (&bhi)[S6_RPW],
                             const int (&qph)[S6_RPW][S6_B], const uint32_t (&qv)[S6_RPW][S6_B], const uint32_t (&qlo)[S6_RPW], const uint32_t (&qhi)[S6_RPW]) {
            for (int i = tid * 4; i < 2 * CH; i += 256 * S6_WAVES) *(uint4*)&colmask[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < S6_RPW; ++rr) {
                const int r = wave + S6_WAVES * rr;
                const uint32_t bit = 1u << (r & 31); const int half = r >> 5;
#pragma unroll
                for (int b = 0; b < S6_B; ++b) if (bph[rr][b] >= 0) atomicOr(&colmask[2 * bph[rr][b] + half], bit);
                for (uint32_t e0 = blo[rr] + 64u * S6_B; e0 < bhi[rr]; e0 += 64u) {          // rows with more entries than the registers hold
                    const uint32_t e = e0 + (uint32_t)lane;
                    if (e < bhi[rr]) atomicOr(&colmask[2 * (d.sp_pham[e] - p0) + half], bit);
                }
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < S6_RPW; ++rr) {
                const int r = wave + S6_WAVES * rr;
                uint32_t hs = 0;
#pragma unroll
                for (int b = 0; b < S6_B; ++b) hit(to_row, std::integral_constant<bool, PACKED>{}, r, qph[rr][b], qv[rr][b], hs);
                for (uint32_t e0 = qlo[rr] + 64u * S6_B; e0 < qhi[rr]; e0 += 64u) {
                    const uint32_t e = e0 + (uint32_t)lane;
                    const bool in = e < qhi[rr];
                    uint2 x = make_uint2((uint32_t)(p0 - 1), 0u);
                    if (in) { if constexpr (COUNT) x = make_uint2((uint32_t)d.sp_pham[e], 1u); else x = ent[e]; }
                    hit(to_row, std::false_type{}, r, (MODE == PCW_POCP && !decltype(to_row)::value && x.y <= 1u) ? -1 : (int)x.x - p0, x.y, hs);
                }
                if (hs) atomicAdd(&acc[decltype(to_row)::value ? r * S6_LD + lane : lane * S6_LD + r], hs);
            }
            __syncthreads();                                                        // probes done before the masks are cleared again
        };
        direction(std::true_type{}, ph_t, lo_t, hi_t, ph_s, v_s, lo_s, hi_s);       // masks over the targets, the sources' entries probe
        if constexpr (COUNT) {
            // |S n T| is symmetric: the sources' probes have counted it
        } else if constexpr (MODE == PCW_POCP) {
            // conserved(s, t) = sum over shared phams of cnt_s + cnt_t = sum (cnt_s + 1) + sum (cnt_t - 1): the first direction added
            // cnt_s + 1 per hit; in the second only the targets' PARALOG entries (cnt_t > 1: ~6 %) have anything to add, the rest
            // stay out of the probes (the masks over the sources are still built from all their entries)
#pragma unroll
            for (int rr = 0; rr < S6_RPW; ++rr)
#pragma unroll
                for (int b = 0; b < S6_B; ++b) if ((v_s[rr][b] >> 16) <= 1u) ph_t[rr][b] = -1;       // (the targets' masks are not built again)
            direction(std::false_type{}, ph_s, lo_s, hi_s, ph_t, v_s, lo_t, hi_t);                   // (v_s: both sides' counts, packed)
        } else if constexpr (!PACKED) direction(std::false_type{}, ph_s, lo_s, hi_s, ph_t, v_t, lo_t, hi_t);      // the other way round
    }
    // finish: 4,096 pairs, 8 per thread; consecutive lanes run along the output's contiguous direction
#pragma unroll 4
    for (int q = 0; q < S6_T * S6_T / (64 * S6_WAVES); ++q) {
        const int idx = tid + 64 * S6_WAVES * q;
        const int fast = idx & 63, slow = idx >> 6;
        const int ls = condensed ? slow : fast, lt = condensed ? fast : slow;
        const int s = g_s[ls], t = g_t[lt];
        if (s < 0 || t < 0 || s >= t) continue;
        const uint32_t cons = acc[ls * S6_LD + lt];
        if constexpr (COUNT) {                                                       // metrics.py:45-53 (gcs), 75-80 (jc)
            out[pc_out_index(d, sh, s, t, k0 + lt, condensed)] = pc_set_value<MODE == S6_GCS ? PC_GCS : PC_JC>((int)cons, (int)(tot_s[ls] + tot_t[lt]), as_distance);
            continue;
        }
        double sim = 0.0;
        if (cons) sim = (double)cons / (double)(tot_s[ls] + tot_t[lt]);               // metrics.py:104-110 (pocp), 149-152 (af)
        out[pc_out_index(d, sh, s, t, k0 + lt, condensed)] = pc_finish(sim, as_distance);
    }
    __syncthreads();                                                                // the next tile clears acc and rewrites g_s, g_t
    }
}

__global__ void k_pair_entries(const int32_t* __restrict__ pham, const int32_t* __restrict__ len, const int32_t* __restrict__ cnt,
                               uint2* __restrict__ pair_len, uint2* __restrict__ pair_cnt, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { pair_len[e] = make_uint2((uint32_t)pham[e], (uint32_t)len[e]); pair_cnt[e] = make_uint2((uint32_t)pham[e], (uint32_t)cnt[e]); }
}
int pc_launch_pair_entries(const int32_t* pham, const int32_t* len, const int32_t* cnt, uint2* pair_len, uint2* pair_cnt, int64_t n, hipStream_t st) {
    if (n <= 0) return PC_OK;
    hipLaunchKernelGGL(k_pair_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pham, len, cnt, pair_len, pair_cnt, n);
    if (hipGetLastError() != hipSuccess) { pc_set_error("k_pair_entries launch failed"); return PC_ERR_HIP; }
    return PC_OK;
}

// A genome's entries of phams with at least two holders, dense ids, at the start of its own slot [ent_off[g], ent_off[g+1]) of the
// sp_* arrays; sp_rank[g][w] = first of them at or after dense word w; sp_end[g] = their end.  One WAVE per genome: 64 entries per
// step, kept ones compacted by ballot; then every lane finds the rank of its share of the words by bisection of the compact ids
// (one thread per genome walked ~100 entries and W2 words one after the other: 0.15 ms of a 1.9-ms upload at N = 2,000).
__global__ __launch_bounds__(256) void k_sp_build(int N, const uint32_t* __restrict__ ent_off, const int32_t* __restrict__ pham, const int32_t* __restrict__ len,
                                                  const int32_t* __restrict__ cnt, const int32_t* __restrict__ dense, int W2, int32_t* __restrict__ sp_pham,
                                                  uint2* __restrict__ sp_len, uint2* __restrict__ sp_cnt, uint32_t* __restrict__ sp_rank, uint32_t* __restrict__ sp_end) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= N) return;
    const uint32_t e0 = ent_off[g], e1 = ent_off[g + 1];
    uint32_t at = e0;
    for (uint32_t base = e0; base < e1; base += 64u) {
        const uint32_t e = base + (uint32_t)lane;
        int id = -1, l = 0, c2 = 0;
        if (e < e1) { id = dense[pham[e]]; l = len[e]; c2 = cnt[e]; }
        const unsigned long long keep = __ballot(id >= 0);
        if (id >= 0) {
            const uint32_t to = at + (uint32_t)__popcll(keep & ((1ULL << lane) - 1ULL));
            sp_pham[to] = id; sp_len[to] = make_uint2((uint32_t)id, (uint32_t)l); sp_cnt[to] = make_uint2((uint32_t)id, (uint32_t)c2);
        }
        at += (uint32_t)__popcll(keep);
    }
    if (lane == 0) sp_end[g] = at;
    __threadfence();                                                                // the wave's own stores are read back below (loads at agent scope: not from a stale L1 line)
    uint32_t* rank = sp_rank + (int64_t)g * W2;
    for (int w = lane; w < W2; w += 64) {                                           // first kept entry with id >= 64 w
        uint32_t lo = e0, hi = at;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (__hip_atomic_load(&sp_pham[mid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 64 * w) lo = mid + 1; else hi = mid; }
        rank[w] = lo;
    }
}
int pc_launch_sp_build(int N, const uint32_t* ent_off, const int32_t* pham, const int32_t* len, const int32_t* cnt, const int32_t* dense, int W2,
                       int32_t* sp_pham, uint2* sp_len, uint2* sp_cnt, uint32_t* sp_rank, uint32_t* sp_end, hipStream_t st) {
    if (N <= 0) return PC_OK;
    hipLaunchKernelGGL(k_sp_build, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, N, ent_off, pham, len, cnt, dense, W2, sp_pham, sp_len, sp_cnt, sp_rank, sp_end);
    if (hipGetLastError() != hipSuccess) { pc_set_error("k_sp_build launch failed"); return PC_ERR_HIP; }
    return PC_OK;
}

int pc_launch_sparse64(int mode, const PcDev& d, const PcShard& sh, double* out, int as_distance, int condensed, hipStream_t st) {
    if (sh.nown <= 0 || d.N <= 1) return PC_OK;
    // mask chunk: all phams at once while two workgroups still fit a CU (8 B per pham + 17 KB of accumulators: 7,680 phams), else the
    // fewest equal chunks of at most that many
    const int P64 = d.sp_W * 64;                                                    // phams with at least two holders
    // ... except that 2,048 ... 7,680 phams are split in two from ~4,000 genomes: the one-batch instances need 59 (gcs / jc), 78 (af) and --
    // held there by the launch bound, 4 dwords of scratch -- 80 (pocp) registers, and with 20 KB of masks three workgroups fit a CU instead
    // of two (N = 20,000, 5,056 phams: jc 2.06 -> 1.79 ms, af 2.89 -> 2.60, pocp 2.61 -> 2.42; below: af at N = 3,000 0.150 ms whole, 0.165 split)
    const int chunk_cap = pc_s6_dense(mode, 2) ? 6912 : 7680;                       // (6 KB of broadcast staging beside the accumulators)
    int n_chunks = (P64 + chunk_cap - 1) / chunk_cap;
    if (n_chunks == 1 && P64 >= 2048 && (int64_t)d.N * sh.nown >= (int64_t)4000 * 4000) n_chunks = 2;
    if (const char* force = getenv("PC_S64_CHUNKS")) {                              // test knob (read per launch): at least this many chunks, so that
        const int want_chunks = atoi(force);                                        // small collections reach the one-batch instances and the forced split
        if (want_chunks > n_chunks && want_chunks <= P64 / 64) n_chunks = want_chunks;
    }
    const int CH = (P64 / 64 + n_chunks - 1) / n_chunks * 64;                       // equal chunks (synth(20000,20000): 5 x 4,096: jc 2.67 ms, 3 x 6,720: 2.5)
    const size_t lds = (size_t)CH * 8 + (size_t)S6_T * S6_LD * 4 + (pc_s6_dense(mode, CH < P64 ? 1 : 2) ? (size_t)S6_WAVES * S6_STAGE_DWORDS * 4 : 0);
    const unsigned n_units = pc_tile_grid((d.N + S6_T - 1) / S6_T, (sh.nown + S6_T - 1) / S6_T, S6_SUPER);
    const unsigned resident = (unsigned)(2 * (d.n_cu > 0 ? d.n_cu : 256) + 7) / 8u * 8u;      // (the context's own device: pc_ctx_create asked it)
    // three units per workgroup (see the kernel); small matrices: one unit each, up to four times the workgroups that fit the chip
    // at once (N = 2,000: 0.158 ms with two units per workgroup, 0.129 with one)
    const unsigned want = std::max(std::min(n_units, 4u * resident), ((n_units + 2u) / 3u + 7u) / 8u * 8u);
    dim3 grid(std::min(n_units, want)), block(64 * S6_WAVES);
    // (up to 78 KB of dynamic LDS: HIP on this hardware needs no opt-in above 64 KB -- the K4 launches take up to 160 KB the same way)
#define S6_LAUNCH(M, B) hipLaunchKernelGGL((k_sparse_tile64<M, B>), grid, block, lds, st, d, sh, out, as_distance, condensed, CH, n_units)
    const bool chunked = CH < P64;
    if (mode == S6_GCS) { if (chunked) S6_LAUNCH(S6_GCS, 1); else S6_LAUNCH(S6_GCS, 2); }
    else if (mode == S6_JC) { if (chunked) S6_LAUNCH(S6_JC, 1); else S6_LAUNCH(S6_JC, 2); }
    else if (mode == PCW_POCP) { if (chunked) S6_LAUNCH(PCW_POCP, 1); else S6_LAUNCH(PCW_POCP, 2); }
    else { if (chunked) S6_LAUNCH(PCW_AF, 1); else S6_LAUNCH(PCW_AF, 2); }
#undef S6_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_sparse_tile64 launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// ---------------------------------------------------------------------------------
// K1 / K2 for large matrices (r05): all four set metrics with the masks over a block of 64 TARGET genomes kept in LDS across a run of
// source tiles, and every accumulator row owned by ONE wave -- no barrier per tile.  (metrics.py:26-157)
//
// What the r05 split of k_sparse_tile64's counting mode showed (profiles/r05/experiments/set_tile_time_split.txt, sparse_col.txt;
// N = 20,000, 1.51 ms): 0.75 ms of a launch is neither probing nor the epilogue but what a tile does before it can probe -- per mask
// chunk two dependent rounds of global loads (rank table, then entries) for both sides, a 20-KB clear, the atomicOr build, three
// barriers -- with the VALU busy 0.82 of the time.  Here
//   * a workgroup (16 waves) takes a UNIT = one block of 64 target genomes x a run of S7_SEG source tiles.  The targets' entries are
//     loaded, and the masks over them (all phams at once: 8 B per pham with at least two holders) built, ONCE per unit;
//   * per source tile a wave loads the entries of ITS four source rows (their ranges one tile ahead), reads their masks, and adds:
//     |S n T| is symmetric, so only the sources probe, and a source row's accumulator row is written by the wave that owns the row
//     alone -- its direct adds go to LDS (atomics: two entries of a row may hit the same target), its broadcast adds stay in a
//     register per row -- and the same wave finishes the row's 64 pairs (fp64 epilogue, one coalesced store).  Nothing in a tile
//     waits for another wave: no barrier, and at 8 waves per SIMD another wave is always ready;
//   * units are dealt so that XCD x takes the target blocks ty = x mod 8, run after run of source tiles (see `tx0` in the kernel): the
//     workgroups an XCD holds at a time read the same source rows.
// pocp and af run here too (value modes): a hit adds the SOURCE entry's value (gene count, resp. summed length) and the TARGET's.  The
// source's rides with the probing entry.  The target's is looked up: once per unit the block's values are laid out in LDS pham by pham
// -- val_off[p] = where pham p's start (an exclusive prefix sum over the popcounts of the masks), then one 16-bit value per target that
// holds p, in target order -- so the value of hit target o is vals[val_off[p] + popcount(mask below bit o)]: for the one or two hits an
// entry adds itself the first (and second) value of the run, for a broadcast entry the lane of target l reads the value at the rank
// v_mbcnt gives it.  Everything still lands in the probing wave's own rows: no barrier.  (Tried first and dropped, records in
// profiles/r05/experiments/sparse_col.txt: the targets' direction through a transposed bitmap and shared accumulators -- af 2.62 ms;
// pocp's paralog excess from an LDS list against the source row in the HBM bitmap -- 2.14 -- or against a per-wave bit set -- 1.69.)
// Values are 16 bits: gene counts and summed lengths of an entry below 65,536, and a block's entries within what two workgroups per
// CU leave beside masks, offsets and accumulators (~7,000 at 5,056 phams) -- the host checks both and sends the rest to k_sparse_tile64.
// Needs 8 B x phams-with-two-holders of LDS beside the accumulators: up to 7,680 such phams; beyond, or for small matrices, the
// kernels above run.  No MFMA: this is a sparse join, ~3 shared phams per pair.
// ---------------------------------------------------------------------------------
#define S7_SEG 8                                                  // source tiles per unit, at most (small matrices: fewer, see the launcher)
#define S7_B 2                                                    // 64-entry batches of a row held in registers (a row's ~100 entries)
#define S7_WAVES 16
#define S7_RPW (S6_T / S7_WAVES)                                  // rows (of either side) a wave owns
template <int MODE>
__global__ __launch_bounds__(64 * S7_WAVES, 8) void k_sparse_col(PcDev d, PcShard sh, double* __restrict__ out, int as_distance, int condensed, int P64, int nty, int seg, int nruns) {
    static_assert(MODE == S6_GCS || MODE == S6_JC || MODE == PCW_POCP || MODE == PCW_AF, "gcs, jc, pocp, af");
    constexpr bool POCP = MODE == PCW_POCP, AF = MODE == PCW_AF;
    constexpr bool VAL = POCP || AF;                                                // a hit adds a value of the source's entry, not 1
    extern __shared__ __attribute__((aligned(16))) uint32_t sp_lds[];
    uint32_t* colmask = sp_lds;                                                    // [P64][2]: which of the block's 64 targets hold the pham
    uint32_t* acc = sp_lds + 2 * P64;                                              // [64 sources][65]
    // pocp / af: the TARGETS' values, looked up per hit: vals[val_off[p] + (rank of the hit's target among the targets that hold p)]
    uint16_t* val_off = (uint16_t*)(acc + S6_T * S6_LD);                           // [P64] start of pham p's values
    uint16_t* vals = val_off + P64;                                                // [<= pc_sparse_col_vals_cap] gene counts / summed lengths, 16 bits each (the host checked)
    __shared__ uint32_t scan_w[S7_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // unit of this workgroup
    const unsigned xcd = blockIdx.x & 7u, kk = blockIdx.x >> 3;
    const unsigned nty8 = ((unsigned)nty + 7u) / 8u;
    const int ty = (int)((kk % nty8) * 8u + xcd), run = (int)(kk / nty8);
    if (ty >= nty) return;
    const int k0 = ty * S6_T;
    const int klast = min(k0 + S6_T, sh.nown) - 1;
    const int live = (pc_owned(sh, klast) + S6_T - 1) / S6_T;                      // source tiles with a pair s < t in them: s0 < the block's last target
    // Runs are ABSOLUTE ranges of source tiles: the workgroups an XCD holds at one time (consecutive blockIdx: the same run, target
    // blocks 8 apart) stream the same `seg` source tiles, which its L2 then serves -- with runs counted from each block's own diagonal,
    // as first built, every workgroup streamed tiles of its own and the source entries came from HBM once per TILE: 1.33 GB fetched
    // per jc launch at N = 20,000, 2.7 GB for pocp / af; now 0.21 / 0.44 (profiles/r05/experiments/sparse_col.txt).  Order: first, for
    // every block, the run that holds its DIAGONAL tile (pairs inside a cluster share ~85 phams: ten times the work, so they must not
    // start last -- with them in line a jc launch at N = 3,000 took 0.058 ms instead of 0.046), then the other runs, highest tiles first:
    // the last runs (tiles 0 .. seg - 1) are live for every target block, so the launch ends with the chip full.
    // (... and the run below it: a cluster of 40 genomes straddles tile boundaries, so the tile next to the diagonal one is heavy too)
    // -- taken first as well while runs are short (seg < 4: small matrices); with 8 tiles per run it mostly lies in the diagonal run, and a
    // second diagonal-relative run would cost L2 sharing: traffic 1.34 -> 1.42 x algorithmic for pocp / af at N = 20,000, same time)
    const int diag_run = (live - 1) / seg, early = seg < 4 ? 2 : 1;
    const int abs_run = run < early ? diag_run - run : nruns - 1 + early - run;
    if (abs_run < 0 || (run >= early && abs_run <= diag_run && abs_run > diag_run - early)) return;
    const int tx0 = abs_run * seg;
    if (tx0 >= live) return;
    const int tx1 = min(live, tx0 + seg);

    // ---- once per unit: the masks over the targets
    const int gt_l = k0 + lane < sh.nown ? pc_owned(sh, k0 + lane) : -1;           // lane l <-> target row l, for the whole unit
    const uint32_t tot_t_l = gt_l >= 0 ? (uint32_t)(POCP ? d.ngen[gt_l] : AF ? (int)d.tlen[gt_l] : d.nph[gt_l]) : 0u;
    const int64_t lbase_l = (!condensed && gt_l >= 0) ? sh.lbase[k0 + lane] : 0;
    {
        uint32_t rl_t = 0, rh_t = 0;
        if (gt_l >= 0) { rl_t = d.ent_off[gt_l]; rh_t = d.sp_end[gt_l]; }
        uint32_t lo_t[S7_RPW], hi_t[S7_RPW];
        int ph_t[S7_RPW][S7_B]; uint32_t ex_t[VAL ? S7_RPW : 1][VAL ? S7_B : 1];        // pocp: gene count; af: summed length
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) {
            const int r = wave + S7_WAVES * rr;
            lo_t[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rl_t, r); hi_t[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rh_t, r);
        }
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr)
#pragma unroll
            for (int b = 0; b < S7_B; ++b) {
                const uint32_t et = lo_t[rr] + (uint32_t)(64 * b + lane);
                if constexpr (VAL) { const uint2 x = et < hi_t[rr] ? (AF ? d.sp_len : d.sp_cnt)[et] : make_uint2(0xffffffffu, 0u); ph_t[rr][b] = (int)x.x; ex_t[rr][b] = x.y; }
                else ph_t[rr][b] = et < hi_t[rr] ? d.sp_pham[et] : -1;
            }
        for (int i = tid * 4; i < 2 * P64; i += 256 * S7_WAVES) *(uint4*)&colmask[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid; i < S6_T * S6_LD; i += 64 * S7_WAVES) acc[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) {
            const int r = wave + S7_WAVES * rr;
            const uint32_t bit = 1u << (r & 31); const int half = r >> 5;
#pragma unroll
            for (int b = 0; b < S7_B; ++b) if (ph_t[rr][b] >= 0) atomicOr(&colmask[2 * ph_t[rr][b] + half], bit);
            for (uint32_t e0 = lo_t[rr] + 64u * S7_B; e0 < hi_t[rr]; e0 += 64u) {            // rows with more entries than the registers hold
                const uint32_t e = e0 + (uint32_t)lane;
                if (e < hi_t[rr]) atomicOr(&colmask[2 * d.sp_pham[e] + half], bit);
            }
        }
        if constexpr (VAL) {
            // val_off = exclusive prefix sum over the phams of the number of targets that hold them (a thread takes a run of phams,
            // the waves' totals meet in LDS), then every target entry drops its value at its pham's start + its row's rank in the mask
            __syncthreads();                                                        // masks complete
            const int per = (P64 + 64 * S7_WAVES - 1) / (64 * S7_WAVES), p_lo = tid * per, p_hi = min(P64, p_lo + per);
            uint32_t mine = 0;
            for (int p = p_lo; p < p_hi; ++p) mine += (uint32_t)(__popc(colmask[2 * p]) + __popc(colmask[2 * p + 1]));
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += up; }
            if (lane == 63) scan_w[wave] = incl;
            __syncthreads();
            uint32_t base = incl - mine;
            for (int w = 0; w < wave; ++w) base += scan_w[w];
            for (int p = p_lo; p < p_hi; ++p) { val_off[p] = (uint16_t)base; base += (uint32_t)(__popc(colmask[2 * p]) + __popc(colmask[2 * p + 1])); }
            __syncthreads();
            auto drop = [&](int r, int p, uint32_t v) {
                const uint2 m = *(const uint2*)&colmask[2 * p];
                const int rank = r < 32 ? __popc(m.x & ((1u << r) - 1u)) : __popc(m.x) + __popc(m.y & ((1u << (r - 32)) - 1u));
                vals[(int)val_off[p] + rank] = (uint16_t)v;
            };
#pragma unroll
            for (int rr = 0; rr < S7_RPW; ++rr) {
                const int r = wave + S7_WAVES * rr;
#pragma unroll
                for (int b = 0; b < S7_B; ++b) if (ph_t[rr][b] >= 0) drop(r, ph_t[rr][b], ex_t[rr][b]);
                for (uint32_t e0 = lo_t[rr] + 64u * S7_B; e0 < hi_t[rr]; e0 += 64u) {
                    const uint32_t e = e0 + (uint32_t)lane;
                    if (e < hi_t[rr]) { const uint2 x = (AF ? d.sp_len : d.sp_cnt)[e]; drop(r, (int)x.x, x.y); }
                }
            }
        }
    }
    // the sources' entry ranges, one tile ahead
    uint32_t rl_s = 0, rh_s = 0;
    { const int g = tx0 * S6_T + lane; if (g < d.N) { rl_s = d.ent_off[g]; rh_s = d.sp_end[g]; } }
    __syncthreads();                                                                // masks (and values) complete: from here on no wave waits for another

    // one probe step of source row r: 64 entries (one per lane), each with the 64-bit mask m of the targets that hold its pham.  Up
    // to two hits the lane adds itself (LDS); more are BROADCAST: the mask becomes the EXEC mask of one v_add into hs, which lane l
    // holds for target l (k_sparse_tile64's step; raising the threshold with a loop per lane -- 3, 4, 6 hits -- changed nothing)
    auto step = [&](int r, uint2 m, uint32_t v, uint32_t off, uint32_t& hs) {      // v: what a hit adds (1; pocp / af: the source's gene count / length, + the target's from vals[off + rank])
        const int pc = __popc(m.x) + __popc(m.y);
        if (pc > 0 && pc <= 2) {
            const unsigned long long mm = ((unsigned long long)m.y << 32) | m.x;
            const int o1 = __builtin_ctzll(mm), o2 = 63 - __builtin_clzll(mm);
            if constexpr (VAL) {
                const uint32_t t1 = vals[off], t2 = vals[off + 1];                  // (one entry of padding behind the table)
                atomicAdd(&acc[r * S6_LD + o1], v + t1);
                if (pc == 2) atomicAdd(&acc[r * S6_LD + o2], v + t2);
            } else {
                atomicAdd(&acc[r * S6_LD + o1], v);
                if (pc == 2) atomicAdd(&acc[r * S6_LD + o2], v);
            }
        }
        unsigned long long heavy = __ballot(pc > 2);
        while (heavy) {
            const int k = __builtin_ctzll(heavy);
            asm("s_bitset0_b64 %0, %1" : "+s"(heavy) : "s"(k));
            const uint32_t mkl = (uint32_t)__builtin_amdgcn_readlane((int)m.x, k), mkh = (uint32_t)__builtin_amdgcn_readlane((int)m.y, k);
            const unsigned long long mk = ((unsigned long long)mkh << 32) | (unsigned long long)mkl;
            // (every lane of the workgroup is active here -- wave-uniform control flow -- so EXEC is all ones before and after)
            if constexpr (VAL) {
                // lane l <-> target l: where bit l of the entry's mask is set, its own value + target l's, found at the rank of bit l
                const uint32_t vk = (uint32_t)__builtin_amdgcn_readlane((int)v, k), offk = (uint32_t)__builtin_amdgcn_readlane((int)off, k);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mkh, __builtin_amdgcn_mbcnt_lo(mkl, 0u));
                if ((mk >> lane) & 1ULL) hs += vk + (uint32_t)vals[offk + rank];
            } else asm volatile("s_mov_b64 exec, %1\n\tv_add_u32 %0, %0, 1\n\ts_mov_b64 exec, -1" : "+v"(hs) : "s"(mk));
        }
    };
    const uint2* __restrict__ ent = AF ? d.sp_len : d.sp_cnt;                       // pocp / af: (dense pham id, gene count / summed length) in one 8-byte load

#pragma unroll 1
    for (int tx = tx0; tx < tx1; ++tx) {
        const int s0 = tx * S6_T;
        uint32_t lo_s[S7_RPW], hi_s[S7_RPW];
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) {
            const int r = wave + S7_WAVES * rr;
            lo_s[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rl_s, r); hi_s[rr] = (uint32_t)__builtin_amdgcn_readlane((int)rh_s, r);
        }
        int ph_s[S7_RPW][S7_B]; uint32_t v_s[VAL ? S7_RPW : 1][VAL ? S7_B : 1];
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr)
#pragma unroll
            for (int b = 0; b < S7_B; ++b) {
                const uint32_t es = lo_s[rr] + (uint32_t)(64 * b + lane);
                if constexpr (VAL) { const uint2 x = es < hi_s[rr] ? ent[es] : make_uint2(0xffffffffu, 0u); ph_s[rr][b] = (int)x.x; v_s[rr][b] = x.y; }
                else ph_s[rr][b] = es < hi_s[rr] ? d.sp_pham[es] : -1;
            }
        rl_s = 0; rh_s = 0;                                                         // the next tile's ranges: in flight while this one is probed
        { const int g = s0 + S6_T + lane; if (tx + 1 < tx1 && g < d.N) { rl_s = d.ent_off[g]; rh_s = d.sp_end[g]; } }
        // all of the tile's mask reads first, then the adds (an LDS read does not move across an LDS atomic: masks and accumulators
        // are one array to the compiler)
        uint2 ms[S7_RPW][S7_B];
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr)
#pragma unroll
            for (int b = 0; b < S7_B; ++b) ms[rr][b] = ph_s[rr][b] >= 0 ? *(const uint2*)&colmask[2 * ph_s[rr][b]] : make_uint2(0u, 0u);
        uint32_t offs[VAL ? S7_RPW : 1][VAL ? S7_B : 1];
        if constexpr (VAL) {
#pragma unroll
            for (int rr = 0; rr < S7_RPW; ++rr)
#pragma unroll
                for (int b = 0; b < S7_B; ++b) offs[rr][b] = ph_s[rr][b] >= 0 ? (uint32_t)val_off[ph_s[rr][b]] : 0u;
        }
        uint32_t hs[S7_RPW];
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) {
            const int r = wave + S7_WAVES * rr;
            hs[rr] = 0u;
#pragma unroll
            for (int b = 0; b < S7_B; ++b) step(r, ms[rr][b], VAL ? v_s[rr][b] : 1u, VAL ? offs[rr][b] : 0u, hs[rr]);
            for (uint32_t e0 = lo_s[rr] + 64u * S7_B; e0 < hi_s[rr]; e0 += 64u) {
                const uint32_t e = e0 + (uint32_t)lane;
                uint2 m = make_uint2(0u, 0u); uint32_t v = 1u, off = 0u;
                if (e < hi_s[rr]) {
                    if constexpr (VAL) { const uint2 x = ent[e]; m = *(const uint2*)&colmask[2 * x.x]; v = x.y; off = (uint32_t)val_off[x.x]; }
                    else m = *(const uint2*)&colmask[2 * d.sp_pham[e]];
                }
                step(r, m, v, off, hs[rr]);
            }
        }
        // finish the wave's own rows: lane l <-> target l.  (The wave's LDS adds above and the reads below are one in-order queue.)
        uint32_t cons_q[S7_RPW];
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) cons_q[rr] = acc[(wave + S7_WAVES * rr) * S6_LD + lane];
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) acc[(wave + S7_WAVES * rr) * S6_LD + lane] = 0u;     // (left clean for the next tile, pairs on or below the diagonal too)
#pragma unroll
        for (int rr = 0; rr < S7_RPW; ++rr) {
            const int s = s0 + wave + S7_WAVES * rr;                               // wave-uniform
            if (s >= d.N) continue;
            const int t = gt_l;
            if (t < 0 || s >= t) continue;
            const uint32_t tot = (uint32_t)(POCP ? d.ngen[s] : AF ? (int)d.tlen[s] : d.nph[s]) + tot_t_l;
            const int64_t idx = condensed ? (int64_t)s * d.N - (int64_t)s * (s + 1) / 2 + (t - s - 1) : lbase_l + s;
            if constexpr (AF) {                                                     // metrics.py:149-152: totals up to 2^32 - 2, so unsigned
                const uint32_t cons = cons_q[rr] + hs[rr];
                out[idx] = pc_finish(cons ? (double)cons / (double)tot : 0.0, as_distance);
            } else out[idx] = pc_set_value<POCP ? PC_POCP : MODE == S6_GCS ? PC_GCS : PC_JC>((int)(cons_q[rr] + hs[rr]), (int)tot, as_distance);   // metrics.py:45-53 (gcs), 75-80 (jc), 104-110 (pocp)
        }
    }
}

// LDS the column kernel takes for a collection with P64 mask entries; 0: it cannot run (masks beyond 7,680 phams)
// pocp / af: what two workgroups per CU leave for the targets' values, 16 bits each (+ one entry of padding); < 1,024: the mode is off
int pc_sparse_col_vals_cap(int P64) {
    const long long room = 80 * 1024 - 256 - ((long long)P64 * 8 + (long long)S6_T * S6_LD * 4 + (long long)P64 * 2);
    return room / 2 - 1 >= 1024 ? (int)(room / 2 - 1) : 0;
}
size_t pc_sparse_col_lds(int mode, int P64) {
    if (mode == PCW_AF || mode == PCW_POCP) return pc_sparse_col_vals_cap(P64) ? (size_t)80 * 1024 - 256 : 0;
    const size_t bytes = (size_t)P64 * 8 + (size_t)S6_T * S6_LD * 4;
    return bytes <= 78 * 1024 ? bytes : 0;                                          // two workgroups per CU
}
int pc_launch_sparse_col(int mode, const PcDev& d, const PcShard& sh, double* out, int as_distance, int condensed, hipStream_t st) {
    if (sh.nown <= 0 || d.N <= 1) return PC_OK;
    const int P64 = d.sp_W * 64;
    const size_t lds = pc_sparse_col_lds(mode, P64);
    if (!lds || (mode != S6_GCS && mode != S6_JC && mode != PCW_POCP && mode != PCW_AF)) { pc_set_error("k_sparse_col: mode %d, %d mask entries", mode, P64); return PC_ERR_LIMIT; }
    const int nty = (sh.nown + S6_T - 1) / S6_T, ntx = (d.N + S6_T - 1) / S6_T;
    // source tiles per unit: as many as leave ~2 units per workgroup slot of the chip (2 slots per CU), at most S7_SEG.  Measured, jc, ms
    // (profiles/r05/experiments/sparse_col.txt): N = 2,000 / 3,000 / 5,000 / 20,000 with 1 tile per unit 0.034 / 0.046 / 0.105 / 1.27,
    // 2: 0.047 / 0.049 / 0.092 / 1.10, 4: 0.058 / 0.060 / 0.093 / 1.02, 8: 0.081 / 0.083 / 0.100 / 0.99, 16: 0.126 / 0.127 / 0.166 / 1.005
    const int64_t live_tiles = (int64_t)nty * ntx / 2 + nty;
    int seg = (int)std::max<int64_t>(1, std::min<int64_t>(S7_SEG, live_tiles / (4 * (int64_t)(d.n_cu > 0 ? d.n_cu : 256))));
    if (const char* force = getenv("PC_COL_SEG")) { const int v = atoi(force); if (v >= 1 && v <= 64) seg = v; }
    const unsigned runs = (unsigned)((ntx + seg - 1) / seg);
    dim3 grid(((unsigned)nty + 7u) / 8u * 8u * (runs + 2u)), block(64 * S7_WAVES);     // (runs 0, 1: every block's diagonal run and the one below; then the runs, highest first)
    if (mode == S6_GCS) hipLaunchKernelGGL((k_sparse_col<S6_GCS>), grid, block, lds, st, d, sh, out, as_distance, condensed, P64, nty, seg, (int)runs);
    else if (mode == PCW_POCP) hipLaunchKernelGGL((k_sparse_col<PCW_POCP>), grid, block, lds, st, d, sh, out, as_distance, condensed, P64, nty, seg, (int)runs);
    else if (mode == PCW_AF) hipLaunchKernelGGL((k_sparse_col<PCW_AF>), grid, block, lds, st, d, sh, out, as_distance, condensed, P64, nty, seg, (int)runs);
    else hipLaunchKernelGGL((k_sparse_col<S6_JC>), grid, block, lds, st, d, sh, out, as_distance, condensed, P64, nty, seg, (int)runs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_sparse_col launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// ---------------------------------------------------------------------------------
// Shared-pham walker.  Visits the shared phams of each pair in ascending pham id;
// entry index of pham p in genome g = rankpre[g][p>>6] + popcount(B[g][p>>6] below bit p).
// ---------------------------------------------------------------------------------
struct PcPairAcc {
    uint32_t k;            // running alignment slot (ENUM / AAI / PEQ) or alignment count (COUNT)
    int64_t cons;          // conserved gene count (POCP) or conserved length (AF / PEQ)
    double num;            // sum of best_ident/len * len   (statistics.py:22 numerator)
    int64_t den;           // sum of best lengths
    int any;               // shared set non-empty
};

template <int MODE>
__device__ __forceinline__ void pc_visit(const PcDev& d, const PcWalkArgs& a, PcPairAcc& p, uint32_t es, uint32_t et,
                                         unsigned long long& cells, unsigned long long& rbytes) {
    p.any = 1;
    if (MODE == PCW_POCP) { p.cons += d.ent_cnt[es] + d.ent_cnt[et]; return; }          // metrics.py:102-103
    if (MODE == PCW_AF) { p.cons += d.ent_len[es] + d.ent_len[et]; return; }            // metrics.py:135-147
    const int cs = d.ent_cnt[es], ct = d.ent_cnt[et];
    // anchor = the genome with fewer genes in the pham; tie -> source (metrics.py:208-209)
    const bool swap = cs > ct;
    const uint32_t ea = swap ? et : es, eb = swap ? es : et;
    const int ca = swap ? ct : cs, cb = swap ? cs : ct;
    const int a0 = d.ent_gene[ea], b0 = d.ent_gene[eb];
    if (MODE == PCW_COUNT) {
        p.k += (uint32_t)(ca * cb);
        const unsigned long long la = (unsigned long long)d.ent_len[ea], lb = (unsigned long long)d.ent_len[eb];
        cells += la * lb;
        p.den += (int64_t)(la * lb);                                                     // per pair, for the cost-balanced deal
        rbytes += la * cb + lb * ca;
    } else if (MODE == PCW_ENUM) {                                                        // sort key per alignment slot (pc_plan.hip)
        for (int ia = 0; ia < ca; ++ia) {
            const unsigned long long qa = d.gene_q[a0 + ia];
            for (int ib = 0; ib < cb; ++ib) {
                const uint32_t k = p.k++;
                a.key[k] = ((unsigned long long)d.gene_q[b0 + ib] << d.ubits) | qa;
                a.val[k] = k;
            }
        }
    } else {                                                                              // AAI / PEQ
        if (MODE == PCW_PEQ) p.cons += d.ent_len[es] + d.ent_len[et];
        for (int ia = 0; ia < ca; ++ia) {
            double best = -1.0; uint32_t best_len = 0;
            for (int ib = 0; ib < cb; ++ib) {
                const uint2 r = a.res[a.alias[p.k++]];
                const double x = (double)r.x / (double)r.y;                               // metrics.py:221
                if (x >= best) { best = x; best_len = r.y; }                              // sorted(...)[-1] (metrics.py:223)
            }
            p.num = p.num + best * (double)best_len;                                      // statistics.py:22
            p.den += best_len;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_walk(PcDev d, PcShard sh, PcWalkArgs a) {
    __shared__ uint64_t rs[TS][WCH + 1];
    __shared__ uint64_t rt[TS][WCH + 1];
    __shared__ unsigned long long red[3];
    int tile_x, tile_y;
    if (!pc_tile_of_block((d.N + TS - 1) / TS, (sh.nown + TS - 1) / TS, tile_x, tile_y)) return;
    const int s0 = tile_x * TS, k0 = tile_y * TS;
    const int klast = min(k0 + TS, sh.nown) - 1;
    if (s0 >= sh.owned[klast]) return;
    const int f = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int cond = a.condensed;
    PcPairAcc acc[4];
    int ss[4], kk[4], tt[4]; bool ok[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ls = cond ? q + 8 * m : f, lt = cond ? f : q + 8 * m;
        ss[m] = s0 + ls; kk[m] = k0 + lt;
        ok[m] = ss[m] < d.N && kk[m] < sh.nown;
        tt[m] = ok[m] ? sh.owned[kk[m]] : 0;
        ok[m] = ok[m] && ss[m] < tt[m];
        acc[m].k = 0; acc[m].cons = 0; acc[m].num = 0.0; acc[m].den = 0; acc[m].any = 0;
        if (ok[m] && (MODE == PCW_ENUM || MODE == PCW_AAI || MODE == PCW_PEQ)) acc[m].k = a.off[sh.lbase[kk[m]] + ss[m]];
    }
    unsigned long long cells = 0, rbytes = 0;
    if (MODE == PCW_COUNT) { if (threadIdx.x < 3) red[threadIdx.x] = 0; }

    for (int w0 = 0; w0 < d.Wb; w0 += WCH) {
        const int wn = min(WCH, d.Wb - w0);
        if (w0) __syncthreads();
        pc_stage_tile(d, sh, s0, k0, w0, wn, rs, rt);
        __syncthreads();
        // A pair shares ~3 of its ~80 bitmap words, but some lane of the wave has a hit in almost every word: visiting
        // inside the word scan would run the (divergent, memory-touching) visit body ~70 times per pair slot.  So the
        // scan only records which words intersect (branch-free, one 32-bit mask per pair: WCH == 32), and the visits then
        // loop over the set bits of that mask -- as many iterations as the busiest lane has shared words.  Order stays
        // ascending.  A thread's four pairs share one row (t under condensed output, s otherwise): its word is read once.
        uint32_t nz[4] = {0u, 0u, 0u, 0u};
        for (int i = 0; i < wn; ++i) {
            const uint64_t common = cond ? rt[f][i] : rs[f][i];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint64_t other = cond ? rs[q + 8 * m][i] : rt[q + 8 * m][i];
                nz[m] |= ((other & common) != 0 ? 1u : 0u) << i;
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (!ok[m]) continue;
            const int ls = ss[m] - s0, lt = kk[m] - k0;
            const uint32_t* rps = d.rankpre + (int64_t)ss[m] * d.Wb + w0;
            const uint32_t* rpt = d.rankpre + (int64_t)tt[m] * d.Wb + w0;
            uint32_t todo = nz[m];
            while (todo) {
                const int w = __ffs((int)todo) - 1;
                todo &= todo - 1;
                const uint64_t sw = rs[ls][w], tw = rt[lt][w];
                uint64_t x = sw & tw;
                const uint32_t bs = rps[w], bt = rpt[w];
                while (x) {
                    const int b = __ffsll((long long)x) - 1;
                    x &= x - 1;
                    const uint64_t below = (1ULL << b) - 1;
                    pc_visit<MODE>(d, a, acc[m], bs + __popcll(sw & below), bt + __popcll(tw & below), cells, rbytes);
                }
            }
        }
    }

    if (MODE == PCW_COUNT) {
        unsigned long long nal = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (ok[m]) {
            if (a.na) a.na[sh.lbase[kk[m]] + ss[m]] = acc[m].k;
            if (a.cost_t && acc[m].den) atomicAdd(&a.cost_t[tt[m]], (unsigned long long)acc[m].den);
            if (a.aln_t && acc[m].k) atomicAdd(&a.aln_t[tt[m]], (unsigned long long)acc[m].k);
            nal += acc[m].k;
        }
        for (int o = 32; o > 0; o >>= 1) {
            nal += __shfl_down(nal, o); cells += __shfl_down(cells, o); rbytes += __shfl_down(rbytes, o);
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], nal); atomicAdd(&red[1], cells); atomicAdd(&red[2], rbytes); }
        __syncthreads();
        if (threadIdx.x < 3 && red[threadIdx.x]) atomicAdd(&a.totals[threadIdx.x], red[threadIdx.x]);
        return;
    }
    if (MODE == PCW_ENUM) return;

#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (!ok[m]) continue;
        const int s = ss[m], t = tt[m];
        double v;
        if (MODE == PCW_POCP) {
            const double sim = acc[m].any ? (double)acc[m].cons / (double)(d.ngen[s] + d.ngen[t]) : 0.0;   // metrics.py:104-110
            v = pc_finish(sim, a.as_distance);
        } else if (MODE == PCW_AF) {
            const double sim = acc[m].any ? (double)acc[m].cons / (double)(d.tlen[s] + d.tlen[t]) : 0.0;   // metrics.py:149-152
            v = pc_finish(sim, a.as_distance);
        } else {
            const double aai = acc[m].any ? acc[m].num / (double)acc[m].den : 0.0;                         // metrics.py:227
            if (MODE == PCW_AAI) v = pc_finish(aai, a.as_distance);
            else {
                const double af = acc[m].any ? (double)acc[m].cons / (double)(d.tlen[s] + d.tlen[t]) : 0.0;
                v = pc_finish(pc_round6(af) * pc_round6(aai), a.as_distance);                              // metrics.py:247-253
            }
        }
        a.out[pc_out_index(d, sh, s, t, kk[m], cond)] = v;
    }
}

int pc_launch_walk(int mode, const PcDev& d, const PcShard& sh, const PcWalkArgs& a, hipStream_t st) {
    if (sh.nown <= 0 || d.N <= 1) return PC_OK;
    dim3 grid(pc_tile_grid((d.N + TS - 1) / TS, (sh.nown + TS - 1) / TS)), block(256);
    switch (mode) {
    case PCW_POCP: hipLaunchKernelGGL(k_walk<PCW_POCP>, grid, block, 0, st, d, sh, a); break;
    case PCW_AF: hipLaunchKernelGGL(k_walk<PCW_AF>, grid, block, 0, st, d, sh, a); break;
    case PCW_COUNT: hipLaunchKernelGGL(k_walk<PCW_COUNT>, grid, block, 0, st, d, sh, a); break;
    case PCW_ENUM: hipLaunchKernelGGL(k_walk<PCW_ENUM>, grid, block, 0, st, d, sh, a); break;
    case PCW_AAI: hipLaunchKernelGGL(k_walk<PCW_AAI>, grid, block, 0, st, d, sh, a); break;
    case PCW_PEQ: hipLaunchKernelGGL(k_walk<PCW_PEQ>, grid, block, 0, st, d, sh, a); break;
    default: pc_set_error("pc_launch_walk: bad mode %d", mode); return PC_ERR_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_walk launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// ---------------------------------------------------------------------------------
// Exclusive prefix sum of u32 (n elements).  2048 elements per workgroup, block sums
// scanned recursively.  Callers that need the total pass n+1 elements with in[n] = 0.
// ---------------------------------------------------------------------------------
#define SCAN_PER_BLOCK 2048

__global__ __launch_bounds__(256) void k_scan_block(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     uint32_t* __restrict__ sums, int64_t n) {
    __shared__ uint32_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK + (int64_t)threadIdx.x * 8;
    uint32_t v[8], tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (base + i < n) ? in[base + i] : 0u; tot += v[i]; }
    // wave inclusive scan of the per-thread totals
    uint32_t incl = tot;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < wv; ++i) woff += wsum[i];
    uint32_t run = woff + incl - tot;
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255 && sums) sums[blockIdx.x] = woff + incl;
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ sums, int64_t n) {
    const uint32_t add = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK + (int64_t)threadIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (base + i < n) out[base + i] += add;
}

int64_t pc_scan_tmp_elems(int64_t n) {
    int64_t tot = 0;
    while (n > SCAN_PER_BLOCK) { n = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK; tot += n; }
    return tot + 1;
}

int pc_scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* tmp, int64_t tmp_elems, hipStream_t st) {
    if (n <= 0) return PC_OK;
    const int64_t nb = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    if (nb == 1) {
        hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(256), 0, st, in, out, (uint32_t*)nullptr, n);
    } else {
        if (tmp_elems < nb) { pc_set_error("scan: temp too small"); return PC_ERR_ARG; }
        hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(256), 0, st, in, out, tmp, n);
        int rc = pc_scan_exclusive_u32(tmp, tmp, nb, tmp + nb, tmp_elems - nb, st);
        if (rc != PC_OK) return rc;
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(256), 0, st, out, tmp, n);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("scan launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

__global__ void k_gather_u32(const uint32_t* __restrict__ src, const int32_t* __restrict__ idx, uint32_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
int pc_launch_gather_u32(const uint32_t* src, const int32_t* idx, uint32_t* dst, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_u32, dim3((n + 63) / 64), dim3(64), 0, st, src, idx, dst, n);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}

// ---------------------------------------------------------------------------------
// Shard assembly on the root: gathered[r][lbase_r(k) + s] -> condensed(s, t).
// The boustrophedon deal has closed forms: round j = t / world, rank r = pos or
// world-1-pos, and lbase_r(k) = world*k(k-1)/2 + r*ceil(k/2) + (world-1-r)*floor(k/2).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(const double* __restrict__ gathered, int world, int64_t stride, int N,
                                                   double* __restrict__ out) {
    const int s = blockIdx.y;
    const int t = s + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const int j = t / world, pos = t % world;
    const int r = (j & 1) ? world - 1 - pos : pos;
    const int64_t k = j;
    const int64_t lbase = (int64_t)world * (k * (k - 1) / 2) + (int64_t)r * ((k + 1) / 2) + (int64_t)(world - 1 - r) * (k / 2);
    out[(int64_t)s * N - (int64_t)s * (s + 1) / 2 + (t - s - 1)] = gathered[(int64_t)r * stride + lbase + s];
}
// Same for an arbitrary deal: t_rank[t] owns target t, whose pairs start at t_lbase[t] inside that rank's shard.
__global__ __launch_bounds__(256) void k_assemble_table(const double* __restrict__ gathered, int64_t stride, int N,
                                                         const int32_t* __restrict__ t_rank, const int64_t* __restrict__ t_lbase,
                                                         double* __restrict__ out) {
    const int s = blockIdx.y;
    const int t = s + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    out[(int64_t)s * N - (int64_t)s * (s + 1) / 2 + (t - s - 1)] = gathered[(int64_t)t_rank[t] * stride + t_lbase[t] + s];
}
int pc_launch_assemble_table(const double* gathered, int64_t stride, int N, const int32_t* t_rank, const int64_t* t_lbase, double* out, hipStream_t st) {
    if (N <= 1) return PC_OK;
    dim3 grid((N + 255) / 256, N - 1);
    hipLaunchKernelGGL(k_assemble_table, grid, dim3(256), 0, st, gathered, stride, N, t_rank, t_lbase, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_assemble_table launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

int pc_launch_assemble(const double* gathered, int world, int64_t stride, int N, double* out, hipStream_t st) {
    if (N <= 1) return PC_OK;
    dim3 grid((N + 255) / 256, N - 1);
    hipLaunchKernelGGL(k_assemble, grid, dim3(256), 0, st, gathered, world, stride, N, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_assemble launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// (n_ident, aln_len) -> (n_ident, n_diag) for the pc_align_pairs test hook
__global__ void k_unpack_res(const uint2* __restrict__ res, const int32_t* __restrict__ la_plus_lb,
                             int32_t* __restrict__ n_ident, int32_t* __restrict__ n_diag, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    n_ident[i] = (int32_t)res[i].x;
    n_diag[i] = la_plus_lb[i] - (int32_t)res[i].y;
}
int pc_launch_unpack_res(const uint2* res, const int32_t* la_plus_lb, int32_t* n_ident, int32_t* n_diag, int64_t n, hipStream_t st) {
    if (n <= 0) return PC_OK;
    hipLaunchKernelGGL(k_unpack_res, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, res, la_plus_lb, n_ident, n_diag, n);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}
