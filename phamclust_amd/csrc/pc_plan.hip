// pc_plan.hip -- planning of one fill's alignment batch (aai / peq).
//
// The pair walk (pc_pairs.hip, mode ENUM) writes one 64-bit key per alignment the reference would run
// (metrics.py:211-217): (rank of the column sequence << ubits) | rank of the row sequence, ranks taken over the
// DISTINCT gene sequences of the upload in launch-class order.  A radix sort of (key, slot) then does three jobs
// at once:
//   * equal keys become adjacent: each distinct (row sequence, column sequence) pair is aligned once and every
//     slot that asked for it gets an alias to the one result (phage genomes share many identical proteins;
//     the alignment depends on the two sequences only);
//   * the distinct alignments of one column sequence are contiguous: that run is the column's bucket, cut into
//     workgroup tasks that share the column's profile;
//   * buckets come out in launch-class order, so each kernel variant's tasks are one contiguous range.
// No atomics: bucket contents and order are a pure function of the input.
// The sort is rocPRIM's (33.5 M pairs, 40 key bits: 1.6 ms on MI355X); the rest are small kernels below.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/phamclust_hip.h"
#include "pc_common.h"

size_t pc_sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    if (n <= 0) return 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, (unsigned)bits, (hipStream_t)0);
    return e == hipSuccess ? bytes : 0;
}

int pc_sort_pairs(void* temp, size_t temp_bytes, const unsigned long long* key_in, unsigned long long* key_out, const uint32_t* val_in,
                  uint32_t* val_out, int64_t n, int bits, hipStream_t st) {
    if (n <= 0) return PC_OK;
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, key_in, key_out, val_in, val_out, (size_t)n, 0u, (unsigned)bits, st);
    if (e != hipSuccess) { pc_set_error("radix_sort_pairs: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// flags[i] = 1 where sorted position i starts a run of equal keys; flags[n] = 0 (so that a scan over n+1 elements
// leaves the number of distinct keys in its last slot)
__global__ __launch_bounds__(256) void k_mark_heads(const unsigned long long* __restrict__ skey, uint32_t* __restrict__ flags, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    flags[i] = (i < n && (i == 0 || skey[i] != skey[i - 1])) ? 1u : 0u;
}
int pc_launch_mark_heads(const unsigned long long* skey, uint32_t* flags, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_mark_heads, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, skey, flags, n);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}

// Sorted position i belongs to distinct alignment u = excl[i] + flags[i] - 1.  Every slot learns its alias; the
// head of a run records the row gene of alignment u and, where the column sequence changes, the bucket bounds
// [start_q, end_q) of that column (both zero-initialised: a column nobody aligns against keeps an empty bucket).
// totals[3] += distinct alignments, totals[4] += their cells.
__global__ __launch_bounds__(256) void k_unique(PcDev d, const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl,
                                                uint32_t* __restrict__ alias, int32_t* __restrict__ bucket_row,
                                                uint32_t* __restrict__ start_q, uint32_t* __restrict__ end_q,
                                                unsigned long long* __restrict__ totals, int64_t n) {
    __shared__ unsigned long long red[2];
    if (threadIdx.x < 2) red[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cells = 0, heads = 0;
    if (i < n) {
        const uint32_t f = flags[i], u = excl[i] + f - 1u;
        alias[sval[i]] = u;
        if (f) {
            const unsigned long long key = skey[i];
            const uint32_t qa = (uint32_t)(key & ((1ull << d.ubits) - 1)), qb = (uint32_t)(key >> d.ubits);
            const int ga = d.q_gene[qa];
            bucket_row[u] = ga;
            cells = (unsigned long long)d.gene_len[ga] * (unsigned long long)d.gene_len[d.q_gene[qb]];
            heads = 1;
            // position i-1 carries the key of the previous distinct alignment (runs hold equal keys): where the column
            // changes, this alignment opens its column's bucket and closes the previous column's
            if (i == 0) start_q[qb] = 0;
            else {
                const uint32_t qp = (uint32_t)(skey[i - 1] >> d.ubits);
                if (qp != qb) { start_q[qb] = u; end_q[qp] = u; }
            }
        }
        if (i == n - 1) end_q[(uint32_t)(skey[i] >> d.ubits)] = u + 1;      // the last column's bucket ends with the list
    }
    for (int o = 32; o > 0; o >>= 1) { cells += __shfl_down(cells, o); heads += __shfl_down(heads, o); }
    if ((threadIdx.x & 63) == 0 && heads) { atomicAdd(&red[0], heads); atomicAdd(&red[1], cells); }
    __syncthreads();
    if (threadIdx.x < 2 && red[threadIdx.x]) atomicAdd(&totals[3 + threadIdx.x], red[threadIdx.x]);
}
int pc_launch_unique(const PcDev& d, const unsigned long long* skey, const uint32_t* sval, const uint32_t* flags, const uint32_t* excl,
                     uint32_t* alias, int32_t* bucket_row, uint32_t* start_q, uint32_t* end_q, unsigned long long* totals, int64_t n, hipStream_t st) {
    if (n <= 0) return PC_OK;
    hipLaunchKernelGGL(k_unique, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, skey, sval, flags, excl, alias, bucket_row, start_q, end_q, totals, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_unique launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

// Workgroup tasks per column sequence q (ntask_q has U+1 slots, the last one zero, for the scan's total).  A bucket of n
// rows gives ceil(n_main / rows_per_task) tasks of its main variant; when n is not a multiple of that variant's nseg
// and a narrower variant is cheaper for the r = n mod nseg left-over rows, those form one more task of that variant.
__device__ __forceinline__ uint32_t pc_split(const PcTaskPlan& tp, int q, uint32_t n, uint32_t& rem_class) {
    const uint32_t nseg = tp.q_nseg[q], r = nseg > 1 ? n % nseg : 0u;
    rem_class = r ? tp.rem_class[(size_t)q * 16 + r] : 255u;
    return rem_class != 255u ? n - r : n;                                 // rows of the main tasks
}
__global__ void k_task_count(const uint32_t* __restrict__ start_q, const uint32_t* __restrict__ end_q, PcTaskPlan tp,
                             uint32_t* __restrict__ ntask_q, int U) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > U) return;
    uint32_t nt = 0;
    if (q < U) {
        const uint32_t n = end_q[q] - start_q[q], per = (uint32_t)tp.task_rows[q];
        uint32_t rc; const uint32_t nmain = pc_split(tp, q, n, rc);
        nt = (nmain + per - 1) / per + (rc != 255u ? 1u : 0u);
    }
    ntask_q[q] = nt;
}
int pc_launch_task_count(const uint32_t* start_q, const uint32_t* end_q, const PcTaskPlan& tp, uint32_t* ntask_q, int U, hipStream_t st) {
    hipLaunchKernelGGL(k_task_count, dim3((U + 1 + 255) / 256), dim3(256), 0, st, start_q, end_q, tp, ntask_q, U);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}

// Launch class of a task = its base class (variant, lanes-per-segment bucket, any-byte) * PC_WAVE_MODES + the mode its row count
// asks for: all rows in one wave's segments, in two waves', or the class's own workgroup (host mirror: pc_nw_task_mode).
__device__ __forceinline__ int32_t pc_launch_class(const PcTaskPlan& tp, uint32_t base_cls, int lb, uint32_t rows) {
    int mode = PC_MODE_CLASS;
    const uint32_t nv4 = (uint32_t)tp.nvar * 4u;
    if (tp.small_modes && base_cls < 2u * nv4 + (uint32_t)tp.n_strip) {     // (the last base class is the general kernel: no modes)
        const int W = base_cls < 2u * nv4 ? tp.variant_w[(base_cls % nv4) / 4u] : 64, G = (lb + W - 1) / W;   // (strip-mined classes: one row per wave)
        const uint32_t nseg = base_cls < 2u * nv4 ? (uint32_t)min(G > 64 ? 1 : 64 / G, 16) : 1u;
        mode = rows <= nseg ? PC_MODE_ONE_WAVE : rows <= 2u * nseg ? PC_MODE_TWO_WAVES : PC_MODE_CLASS;
    }
    return (int32_t)(base_cls * PC_WAVE_MODES + (uint32_t)mode);
}
__global__ void k_task_fill(PcDev d, const uint32_t* __restrict__ start_q, const uint32_t* __restrict__ end_q, PcTaskPlan tp,
                            const uint32_t* __restrict__ task_off_q, PcTask* __restrict__ tasks, int U) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= U) return;
    const uint32_t b = start_q[q], n = end_q[q] - b, per = (uint32_t)tp.task_rows[q];
    uint32_t rc; const uint32_t nmain = pc_split(tp, q, n, rc);
    uint32_t to = task_off_q[q];
    const int gene = d.q_gene[q], lb = d.gene_len[gene];
    for (uint32_t r = 0; r < nmain; r += per) {
        PcTask t; t.gene = gene; t.begin = (int32_t)(b + r); t.end = (int32_t)(b + min(nmain, r + per));
        t.pad = pc_launch_class(tp, tp.q_class[q], lb, (uint32_t)(t.end - t.begin));
        tasks[to++] = t;
    }
    if (rc != 255u) { PcTask t; t.gene = gene; t.begin = (int32_t)(b + nmain); t.end = (int32_t)(b + n); t.pad = pc_launch_class(tp, rc, lb, n - nmain); tasks[to] = t; }
}
int pc_launch_task_fill(const PcDev& d, const uint32_t* start_q, const uint32_t* end_q, const PcTaskPlan& tp, const uint32_t* task_off_q,
                        PcTask* tasks, int U, hipStream_t st) {
    if (U <= 0) return PC_OK;
    hipLaunchKernelGGL(k_task_fill, dim3((U + 255) / 256), dim3(256), 0, st, d, start_q, end_q, tp, task_off_q, tasks, U);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}

// The task list is sorted by (launch class, longest first): inside a launch the workgroups are dispatched in task
// order, the next launch of the same hardware queue cannot start before the last workgroup has finished, and a
// launch that ends on its short tasks drains quickly.  key = class << 32 | ~(rows * column length).
__global__ void k_task_keys(PcDev d, const PcTask* __restrict__ tasks, unsigned long long* __restrict__ key, uint32_t* __restrict__ val, int ntasks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntasks) return;
    const PcTask t = tasks[i];
    const int lb = d.gene_len[t.gene];
    const uint32_t work = (uint32_t)(t.end - t.begin) * (uint32_t)lb;
    key[i] = ((unsigned long long)(uint32_t)t.pad << 32) | (0xffffffffu - work);
    val[i] = (uint32_t)i;
}
__global__ void k_task_gather(const PcTask* __restrict__ in, const uint32_t* __restrict__ idx, PcTask* __restrict__ out, int ntasks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ntasks) out[i] = in[idx[i]];
}
// cls_begin[c] = first sorted task of class >= c (so class c owns [cls_begin[c], cls_begin[c+1])); one thread per class
__global__ void k_class_bounds(const unsigned long long* __restrict__ key, int ntasks, int ncls, uint32_t* __restrict__ cls_begin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncls) return;
    int lo = 0, hi = ntasks;                                              // first i with (key[i] >> 32) >= c
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)(key[mid] >> 32) >= c) hi = mid; else lo = mid + 1; }
    cls_begin[c] = (uint32_t)lo;
}
int pc_launch_task_keys(const PcDev& d, const PcTask* tasks, unsigned long long* key, uint32_t* val, int ntasks, hipStream_t st) {
    if (ntasks <= 0) return PC_OK;
    hipLaunchKernelGGL(k_task_keys, dim3((ntasks + 255) / 256), dim3(256), 0, st, d, tasks, key, val, ntasks);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}
int pc_launch_task_gather(const PcTask* in, const uint32_t* idx, PcTask* out, int ntasks, hipStream_t st) {
    if (ntasks <= 0) return PC_OK;
    hipLaunchKernelGGL(k_task_gather, dim3((ntasks + 255) / 256), dim3(256), 0, st, in, idx, out, ntasks);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}
// Every world-th task of each launch class, starting at `rank`, compacted class by class (slice_begin: host-made prefix of
// the slice's class sizes).  A task's class is its `pad`; its index inside the class is its position in the sorted list minus
// the class's begin.
__global__ void k_task_slice(const PcTask* __restrict__ sorted, int ntasks, const uint32_t* __restrict__ cls_begin,
                             const uint32_t* __restrict__ slice_begin, int rank, int world, PcTask* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntasks) return;
    const PcTask t = sorted[i];
    const uint32_t cls = (uint32_t)t.pad, j = (uint32_t)i - cls_begin[cls];
    if (j % (uint32_t)world == (uint32_t)rank) out[slice_begin[cls] + j / (uint32_t)world] = t;
}
int pc_launch_task_slice(const PcTask* sorted, int ntasks, const uint32_t* cls_begin, const uint32_t* slice_begin, int rank, int world,
                         PcTask* out, hipStream_t st) {
    if (ntasks <= 0) return PC_OK;
    hipLaunchKernelGGL(k_task_slice, dim3((ntasks + 255) / 256), dim3(256), 0, st, sorted, ntasks, cls_begin, slice_begin, rank, world, out);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}
int pc_launch_class_bounds(const unsigned long long* sorted_key, int ntasks, int ncls, uint32_t* cls_begin, hipStream_t st) {
    hipLaunchKernelGGL(k_class_bounds, dim3((ncls + 1 + 63) / 64), dim3(64), 0, st, sorted_key, ntasks, ncls, cls_begin);
    return hipGetLastError() == hipSuccess ? PC_OK : PC_ERR_HIP;
}


// ---------------------------------------------------------------------------------
// Residue bytes -> codes (pc_upload_residues): one wave per gene, 64 bytes per step; the LUT rides in the kernel arguments.
// Alphabet letters (either case) -> 0..23 in BLOSUM62 order, every other byte value its own code, padding PC_PADCODE.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_encode(const uint8_t* __restrict__ raw, const int64_t* __restrict__ seq_off,
                                                 const int64_t* __restrict__ gene_off, const int32_t* __restrict__ gene_len, PcLut lut,
                                                 uint8_t* __restrict__ codes, int G) {
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= G) return;
    const uint8_t* src = raw + seq_off[k];
    uint8_t* dst = codes + gene_off[k];
    const int len = gene_len[k], padded = (len + 15) & ~15;
    for (int i = lane; i < padded; i += 64) dst[i] = i < len ? s_lut[src[i]] : (uint8_t)PC_PADCODE;
}

int pc_launch_encode(const uint8_t* raw, const int64_t* seq_off, const int64_t* gene_off, const int32_t* gene_len, const PcLut& lut,
                     uint8_t* codes, int G, hipStream_t st) {
    if (G <= 0) return PC_OK;
    hipLaunchKernelGGL(k_encode, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, st, raw, seq_off, gene_off, gene_len, lut, codes, G);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pc_set_error("k_encode launch: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return PC_OK;
}

