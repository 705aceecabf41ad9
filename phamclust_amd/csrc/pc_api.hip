// pc_api.hip -- C-ABI of libphamclust_hip.so (include/phamclust_hip.h): context, upload,
// shard bookkeeping and the orchestration of the fill (matrix.py:432-497's hot loop).
//
// Fill plan for aai/peq (all on one stream, one small read-back in the middle):
//   1 COUNT walk     per pair: alignments; per column gene: bucket sizes; totals
//   2 scans          pair -> first result slot; gene -> bucket start; gene -> first task
//   3 read-back      alignment total + task range per kernel variant (a few words)
//   4 ENUM walk      (row gene, result slot) scattered into the column gene's bucket
//   5 K4 launches    one per kernel variant present, wave tasks of <= 64 row sequences
//   6 REDUCE walk    best match per anchor gene, fp64 weighted mean, af, round -> out
// Bucketing by column gene is what lets a wave build one substitution profile and stream
// many row sequences through it; results are written pair-major so step 6 reads them
// contiguously and in the canonical order (pham id, anchor gene, other gene).
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <memory>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include "pc_common.h"
#include "../../include/phamclust_hip.h"

#include <dlfcn.h>

// roctx ranges around the stages of a fill (the reference only logs wall clock around matrix_de_novo).  The marker
// library is looked up at run time -- the product does not link against a profiler -- and the ranges show up in a
// `rocprofv3 --marker-trace` run as upload_sets / upload_residues / fill:<metric> / count / plan / align / reduce.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        if (getenv("PC_NO_ROCTX")) return;
        void* h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
static const Roctx& roctx() { static const Roctx r; return r; }
struct PcRange {
    bool on;
    explicit PcRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~PcRange() { if (on) roctx().pop(); }
    PcRange(const PcRange&) = delete; PcRange& operator=(const PcRange&) = delete;
};
}  // namespace

static thread_local char g_err[512] = "";
void pc_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* pc_last_error(void) { return g_err; }
extern "C" int pc_version(void) { return PC_VERSION; }
extern "C" int pc_test_hooks(void) {
#ifdef PC_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

namespace {

// internal status: a device allocation failed.  A chunked fill answers it with smaller chunks; at the C-ABI it is PC_ERR_HIP.
constexpr int PC_ERR_NOMEM_INTERNAL = -100;
// Fault injection (PC_FAKE_OOM_ABOVE=bytes): device allocations above that size made while a fill is planning fail, as if HBM
// were that small.  Compiled only under -DPC_TEST_HOOKS, i.e. into libphamclust_hip_hooks.so, the twin that
// test_out_of_memory_plan_becomes_smaller_chunks loads; the release library has no such switch (pc_test_hooks() tells which is which).
#ifdef PC_TEST_HOOKS
static thread_local bool g_planning = false;
static size_t fake_oom_limit() {
    static const size_t v = [] { const char* e = getenv("PC_FAKE_OOM_ABOVE"); return e ? (size_t)atoll(e) : (size_t)0; }();
    return g_planning ? v : 0;
}
struct PlanningScope { PlanningScope() { g_planning = true; } ~PlanningScope() { g_planning = false; } };
#else
static constexpr size_t fake_oom_limit() { return 0; }
struct PlanningScope { PlanningScope() {} };
#endif
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PC_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = (fake_oom_limit() && want > fake_oom_limit()) ? hipErrorOutOfMemory : hipMalloc(&p, want);
        if (e != hipSuccess) {
            pc_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; (void)hipGetLastError();
            return e == hipErrorOutOfMemory ? PC_ERR_NOMEM_INTERNAL : PC_ERR_HIP;
        }
        cap = want; return PC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// fn(begin, end) over [0, n) on up to 16 host threads (upload-time indexing of ~10^8 residues)
template <class F> void parallel_chunks(int64_t n, F fn, int64_t grain = 4096) {
    int nt = (int)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u), (n + grain - 1) / grain);
    if (nt <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) th.emplace_back([=] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
    for (auto& x : th) x.join();
}

static inline int abi_rc(int rc) { return rc == PC_ERR_NOMEM_INTERNAL ? PC_ERR_HIP : rc; }

static int upload_raw(DevBuf& b, const void* p, size_t bytes) {
    int rc = abi_rc(b.ensure(std::max<size_t>(bytes, 16)));
    if (rc != PC_OK) return rc;
    if (bytes) PC_HIP(hipMemcpy(b.p, p, bytes, hipMemcpyHostToDevice));
    return PC_OK;
}

template <class T> int upload_vec(DevBuf& b, const std::vector<T>& v) {
    int rc = abi_rc(b.ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
    if (rc != PC_OK) return rc;
    if (!v.empty()) PC_HIP(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PC_OK;
}

}  // namespace

struct pc_ctx {
    int device = 0;
    int n_cu = 256;                         // compute units of THIS context's device (grid sizing of the persistent tile kernels)
    hipStream_t stream = nullptr;
    bool uploaded = false;                  // part 1 of the upload is on the device (set metrics can run)
    bool residues_ready = false;            // ... and part 2 (aai / peq, pc_align_pairs can run)
    int64_t n_residue_bytes_in = 0;         // residue bytes of the packed genomes part 1 saw (part 2 must be given the same)
    PcDev dev{};
    std::vector<int32_t> h_gene_len;
    std::vector<uint32_t> h_sp_n;                      // [N] a genome's entries of phams with at least two holders (k_sparse_col's pocp / af modes: values per block of targets)
    int max_ent_len = 0;                               // largest summed length of a (genome, pham) entry (k_sparse_col's af mode keeps them as 16-bit values)
    std::vector<uint8_t> h_gene_odd;                   // gene holds a byte outside the 24-letter alphabet
    int max_gene_len = 0, min_gene_len = 0, max_nph = 0, max_ngen = 0;
    int64_t max_tlen = 0;                    // largest summed translation length of a genome
    double avg_shared = 0.0;                 // phams an average genome pair shares (pocp's kernel choice)
    // kernel-variant classes over column genes
    int ncls_all = 0;                       // BASE classes: variant * 4 + lanes-per-segment bucket (twice: "any byte" columns), last = general kernel
    int nlc = 0;                            // launch classes = ncls_all * PC_WAVE_MODES (base class x workgroup shape of the task, pc_common.h)
    std::vector<int32_t> cls_max_lb;        // [nlc] longest column sequence that can land in the launch class (LDS size of its launch)
    PcTaskPlan task_plan{};
    // shard
    int rank = 0, world = 1;
    int64_t shard_pairs = 0, shard_stride = 0;
    PcShard shard{};
    bool balanced = false;                  // cost-balanced deal in force (pc_set_shard_balanced): assembly goes through the tables
    std::vector<uint64_t> target_cost;      // DP cells per target genome, computed once per upload
    std::vector<int32_t> h_t_rank;          // the deal in force, host copy: owner rank of each target genome ...
    std::vector<int64_t> h_t_lbase;         // ... and where its pairs start inside that rank's shard
    std::vector<int32_t> h_owned;           // this rank's targets, ascending, and
    std::vector<int64_t> h_lbase;           // [nown+1] the shard-local index of pair (0, owned[k]) (host copies of shard.owned / lbase)
    // persistent device arrays
    DevBuf b_raw, b_seq_tmp;                // part 2's staging: the raw residue bytes and their offsets, as uploaded (encoded into b_codes on the device)
    DevBuf b_sets;                          // part 1 of the upload, one allocation: bitmap | rank table | gene lengths | entry offsets | nph | ngen | tlen | 4 entry arrays
    uint8_t* h_stage = nullptr; size_t h_stage_cap = 0;   // its page-locked host image (grow-only)
    uint8_t* h_raw = nullptr; size_t h_raw_cap = 0;       // page-locked staging of the raw residues (part 2; grow-only, <= 512 MB)
    DevBuf b_gene_off, b_codes;
    DevBuf b_gene_q, b_q_gene, b_q_class, b_q_nseg, b_rem_class, b_cls_begin, b_task_rows, b_owned, b_lbase, b_t_rank, b_t_lbase, b_cost;
    // work buffers (grow-only)
    DevBuf b_na, b_off, b_key0, b_key1, b_val0, b_val1, b_sort_tmp, b_flags, b_excl, b_alias, b_start_q, b_end_q, b_ntask_q, b_task_off_q, b_scan_tmp;
    DevBuf b_tasks, b_tasks_sorted, b_bucket_row, b_bucket_dest, b_res, b_totals, b_plan, b_scratch, b_out, b_lut, b_slice_begin, b_aln_t;
    uint32_t* h_plan = nullptr;             // pinned: [ncls+1] task offsets, then 3 u64 totals
    // what the last alignment plan (stage_plan) left in the work buffers, for the stages that follow it
    struct PlanState {
        bool valid = false; int ppos = 0; int condensed = 1; int64_t A = 0, n_distinct = 0; uint32_t ntasks = 0;
        int k0 = 0, k1 = 0;                 // the owned targets [k0, k1) the plan covers (a chunk of the shard, or all of it)
        bool whole = false;                 // ... all of an unsharded context: what the alignment-sliced route needs
        std::vector<uint32_t> tb;           // [ncls+1] task range per launch class in b_tasks_sorted
        pc_stats st;                        // counts of the plan (alignments, cells, tasks, distinct ...)
    } plan;
    double* h_out = nullptr; size_t h_out_cap = 0;   // pinned result buffer lent out by pc_fill_borrow (grow-only)
    float last_align_ms = 0.f;              // kernel time of the last pc_align_pairs call
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    static constexpr int kAux = 15;         // + the caller's stream = up to 16 concurrent alignment launches (8 by default)
    hipStream_t aux[kAux] = {};             // alignment launches of different classes overlap on these
    hipEvent_t aux_ev[kAux + 1] = {};
    int n_streams = 8;                      // streams actually used (tuning knob: env PC_ALIGN_STREAMS at ctx creation)
    // Streams for the few launches whose TASKS run for tens of milliseconds (strip-mined passes over long genes).  Their own, so
    // that no other launch queues up behind them -- the runtime lays streams over a handful of hardware queues, and a queue runs
    // its launches one after the other: behind a 30-ms strip launch sat a dozen launches of a millisecond each -- and of HIGH
    // priority: those streams get hardware queues of their own, and their waves go first where they compete (they are the
    // fill's critical path).  PC_LONG_PRIORITY=0: ordinary priority.
    static constexpr int kLong = 4;
    hipStream_t lng[kLong] = {};
    hipEvent_t lng_ev[kLong] = {};
    int tie_rule = 0;                       // row of the aligner's tie-rule table (pc_set_tie_rule)
    int64_t lut_key = -1; const double* lut_ptr = nullptr;   // what the gcs / jc epilogue table in b_lut was built for
    hipEvent_t ev_last = nullptr;           // recorded at the end of every entry point that leaves work on a caller's stream
    bool busy = false;                      // ev_last was recorded and not waited for yet
    hipStream_t last_stream = nullptr;      // ... on this stream
    int last_set_kernel = -1;               // kernel family the last gcs / jc / pocp / af fill ran on (pc_last_set_kernel)
    int64_t plan_budget = 0;                // bytes of plan buffers one chunk of an aai / peq fill may use; 0: automatic (pc_set_plan_budget)
};

#ifndef PC_TIE_RULE_DEFAULT
#define PC_TIE_RULE_DEFAULT 0
#endif

// Every entry point runs on the context's device and leaves the calling thread's current device as it found it
// (PyTorch and other libraries in the process keep their own idea of "current device").
struct PcDeviceGuard {
    int prev = -1, dev = -1; bool ok = true;
    explicit PcDeviceGuard(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) {
            hipError_t e = hipSetDevice(dev);
            if (e != hipSuccess) { pc_set_error("hipSetDevice(%d): %s", dev, hipGetErrorString(e)); ok = false; }
        }
    }
    ~PcDeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    PcDeviceGuard(const PcDeviceGuard&) = delete;
    PcDeviceGuard& operator=(const PcDeviceGuard&) = delete;
};
#define PC_ON_DEVICE(c) PcDeviceGuard pc_guard_((c)->device); if (!pc_guard_.ok) return PC_ERR_HIP

// A fill (or plan, slice, reduce) without stats returns while its kernels still run on the CALLER's stream and still use
// the context's work buffers and shard tables.  Every such entry point ends in mark_work(): ONE event, recorded after its
// last launch.  Anything that rewrites those buffers (upload, re-shard, the test hooks, work on another stream) first
// waits for it.
static int wait_last_work(pc_ctx* c, hipStream_t next_stream, bool same_stream_is_ordered) {
    if (!c->busy) return PC_OK;
    if (same_stream_is_ordered && next_stream == c->last_stream) return PC_OK;
    PC_HIP(hipEventSynchronize(c->ev_last));
    c->busy = false;
    return PC_OK;
}
static int mark_work(pc_ctx* c, hipStream_t st) {
    PC_HIP(hipEventRecord(c->ev_last, st));
    c->busy = true; c->last_stream = st;
    return PC_OK;
}

extern "C" int pc_ctx_create(pc_ctx** out, int device_id) {
    if (!out) { pc_set_error("pc_ctx_create: out is NULL"); return PC_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    PC_HIP(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) { pc_set_error("pc_ctx_create: device %d of %d", device_id, n); return PC_ERR_ARG; }
    pc_ctx* c = new (std::nothrow) pc_ctx();
    if (!c) { pc_set_error("out of host memory"); return PC_ERR_ARG; }
    c->device = device_id;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->n_cu = cus; }
    PcDeviceGuard guard(device_id);
    hipError_t e = guard.ok ? hipSuccess : hipErrorInvalidDevice;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming);
    if (const char* env = getenv("PC_PLAN_BYTES")) { const long long v = atoll(env); if (v > 0) c->plan_budget = v; }
    for (int i = 0; i < pc_ctx::kAux && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
    for (int i = 0; i <= pc_ctx::kAux && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->aux_ev[i], hipEventDisableTiming);
    if (const char* env = getenv("PC_ALIGN_STREAMS")) { int v = atoi(env); if (v >= 1 && v <= pc_ctx::kAux + 1) c->n_streams = v; }
    {
        int least = 0, greatest = 0;
        const char* env = getenv("PC_LONG_PRIORITY");
        const bool high = !(env && !strcmp(env, "0")) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least;
        for (int i = 0; i < pc_ctx::kLong && e == hipSuccess; ++i) {
            if (high && hipStreamCreateWithPriority(&c->lng[i], hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); c->lng[i] = nullptr; }
            if (!c->lng[i]) e = hipStreamCreateWithFlags(&c->lng[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->lng_ev[i], hipEventDisableTiming);
        }
    }
    c->tie_rule = PC_TIE_RULE_DEFAULT;
    if (const char* env = getenv("PC_TIE_RULE")) { int v = atoi(env); if (v >= 0 && v < PC_NUM_TIE_RULES) c->tie_rule = v; }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_plan, 4096, hipHostMallocDefault);
    if (e != hipSuccess) { pc_set_error("pc_ctx_create: %s", hipGetErrorString(e)); pc_ctx_destroy(c); return PC_ERR_HIP; }
    *out = c;
    return PC_OK;
}

extern "C" void pc_ctx_destroy(pc_ctx* c) {
    if (!c) return;
    PcDeviceGuard guard(c->device);
    if (c->busy && c->ev_last) (void)hipEventSynchronize(c->ev_last);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_raw) (void)hipHostFree(c->h_raw);
    DevBuf* bufs[] = {&c->b_raw, &c->b_seq_tmp, &c->b_sets, &c->b_gene_off,
                      &c->b_codes, &c->b_gene_q, &c->b_q_gene, &c->b_q_class, &c->b_q_nseg, &c->b_rem_class, &c->b_cls_begin, &c->b_task_rows, &c->b_owned, &c->b_lbase, &c->b_t_rank, &c->b_t_lbase, &c->b_cost,
                      &c->b_na, &c->b_off, &c->b_key0, &c->b_key1, &c->b_val0, &c->b_val1, &c->b_sort_tmp, &c->b_flags, &c->b_excl, &c->b_alias,
                      &c->b_start_q, &c->b_end_q,
                      &c->b_ntask_q, &c->b_task_off_q, &c->b_scan_tmp, &c->b_tasks, &c->b_tasks_sorted, &c->b_bucket_row, &c->b_bucket_dest,
                      &c->b_res, &c->b_totals, &c->b_plan, &c->b_scratch, &c->b_out, &c->b_lut, &c->b_slice_begin, &c->b_aln_t};
    for (DevBuf* b : bufs) b->release();
    for (int i = 0; i < 5; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_last) (void)hipEventDestroy(c->ev_last);
    for (int i = 0; i < pc_ctx::kAux; ++i) if (c->aux[i]) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); }
    for (int i = 0; i <= pc_ctx::kAux; ++i) if (c->aux_ev[i]) (void)hipEventDestroy(c->aux_ev[i]);
    for (int i = 0; i < pc_ctx::kLong; ++i) {
        if (c->lng[i]) { (void)hipStreamSynchronize(c->lng[i]); (void)hipStreamDestroy(c->lng[i]); }
        if (c->lng_ev[i]) (void)hipEventDestroy(c->lng_ev[i]);
    }
    if (c->h_plan) (void)hipHostFree(c->h_plan);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// residue byte -> code.  Alphabet letters (either case) -> 0..23 in BLOSUM62 order; every
// other byte value keeps its own identity (codes 24..229, ASCII case folded) and scores as '*'.
static void build_code_lut(uint8_t lut[256]) {
    static const char alpha[] = "ARNDCQEGHILKMFPSTWYVBZX*";
    int assigned[256];
    for (int i = 0; i < 256; ++i) assigned[i] = -1;
    for (int k = 0; k < 24; ++k) assigned[(unsigned char)alpha[k]] = k;
    int next = 24;
    for (int v = 0; v < 256; ++v) {
        if (v >= 'a' && v <= 'z') continue;
        if (assigned[v] < 0) assigned[v] = next++;
    }
    for (int v = 'a'; v <= 'z'; ++v) assigned[v] = assigned[v - 32];
    for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)assigned[i];
}

// Base class of a column gene: variant * 4 + bucket of lanes per segment (<=8, <=16, <=32, <=64); the same again,
// nvar * 4 higher, for column genes that hold a byte outside the 24-letter alphabet ("any byte" classes: they must run
// the residue-compare cell, see pc_common.h); then one class per wide variant for column genes longer than its 64 x W columns
// (strip-mined passes, k_nw_strip); last class: the general kernel.
#define PC_STRIP_CLASSES 3                                 // W = 32, 48, 64: the last three variants
static int pc_num_classes() { return pc_nw_num_variants() * 8 + PC_STRIP_CLASSES + 1; }
static int pc_class_of(int lb, int variant, bool any_byte) {
    const int nvar = pc_nw_num_variants();
    if (variant < 0) return nvar * 8 + PC_STRIP_CLASSES;
    const int W = pc_nw_variant_w(variant);
    if (lb > 64 * W) return nvar * 8 + std::max(0, variant - (nvar - PC_STRIP_CLASSES));       // strip-mined (the chooser only picks a wide variant for these)
    const int Gs = (lb + W - 1) / W;
    const int Gb = pc_nw_g_bucket(Gs);
    return variant * 4 + (Gb == 8 ? 0 : Gb == 16 ? 1 : Gb == 32 ? 2 : 3) + (any_byte && !pc_nw_variant_takes_any_byte(variant) ? nvar * 4 : 0);
}
static int pc_class_variant(int cls) {
    const int nvar = pc_nw_num_variants();
    if (cls >= nvar * 8 + PC_STRIP_CLASSES) return -1;
    if (cls >= nvar * 8) return nvar - PC_STRIP_CLASSES + (cls - nvar * 8);
    return (cls % (nvar * 4)) / 4;
}
static int pc_class_compare_only(int cls) { const int nvar = pc_nw_num_variants(); return cls >= nvar * 4 && cls < nvar * 8; }
// a launch whose longest column gene exceeds its variant's 64 x W columns runs strip-mined and needs the scratch slab
static bool pc_launch_is_strip(int variant, int max_lb, int mode, int ppos) { return pc_nw_launch_is_strip(variant, max_lb, mode, ppos) != 0; }

static int apply_shard(pc_ctx* c, int rank, int world) {
    c->plan.valid = false;                             // a plan belongs to the shard it was made for
    const int N = c->dev.N;
    std::vector<int32_t> owned; std::vector<int64_t> lbase;
    c->h_t_rank.assign(std::max(N, 1), 0); c->h_t_lbase.assign(std::max(N, 1), 0);
    int64_t best = 0;
    for (int r = 0; r < world; ++r) {
        int64_t tot = 0;
        for (int j = 0;; ++j) {
            const int pos = (j & 1) ? world - 1 - r : r;
            const int64_t t = (int64_t)j * world + pos;
            if (t >= N) { if ((int64_t)j * world >= N) break; else continue; }
            if (r == rank) { owned.push_back((int32_t)t); lbase.push_back(tot); }
            c->h_t_rank[t] = r; c->h_t_lbase[t] = tot;
            tot += t;
        }
        if (r == rank) { lbase.push_back(tot); c->shard_pairs = tot; }
        best = std::max(best, tot);
    }
    c->shard_stride = best;
    c->rank = rank; c->world = world; c->balanced = false;
    int rc = upload_vec(c->b_owned, owned); if (rc != PC_OK) return rc;
    rc = upload_vec(c->b_lbase, lbase); if (rc != PC_OK) return rc;
    c->h_owned = owned; c->h_lbase = lbase;
    c->shard.nown = (int32_t)owned.size();
    c->shard.ident = world == 1 ? 1 : 0;
    c->shard.owned = c->b_owned.as<int32_t>();
    c->shard.lbase = c->b_lbase.as<int64_t>();
    return PC_OK;
}

// Upload, part 1: everything gcs / jc / pocp / af read -- the bitmap, the rank table, the (genome, pham) entries and the
// per-genome scalars.  The reference's set metrics never touch a translation beyond its length (metrics.py:26-157), and
// encoding, hashing and ranking 10^8 residues is 90 % of a full upload.
static int upload_sets(pc_ctx* c, const pc_packed* g) {
    PcRange range("pc:upload_sets");
    int rc = PC_OK;
    const int N = g->n_genomes, P = g->n_phams, W = g->words_per_row;
    if (N <= 0 || P < 0 || W != std::max(1, (P + 63) / 64) || g->reserved != 0 || !g->bitmap || !g->nph || !g->ngen || !g->tlen ||
        !g->gene_off || !g->seq_off) {
        pc_set_error("pc_upload: inconsistent header (N=%d P=%d W=%d)", N, P, W); return PC_ERR_ARG;
    }
    const int64_t G64 = g->gene_off[N];
    if (G64 < 0 || G64 > 0x7fffffffLL || g->gene_off[0] != 0 || (G64 > 0 && (!g->gene_pham || !g->residues))) {
        pc_set_error("pc_upload: bad gene table"); return PC_ERR_ARG;
    }
    const int G = (int)G64;
    static const bool timing = getenv("PC_UPLOAD_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "pc_upload %-22s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    if ((rc = wait_last_work(c, nullptr, false))) return rc;
    c->uploaded = false; c->residues_ready = false; c->target_cost.clear(); c->plan.valid = false;
    PC_HIP(hipStreamSynchronize(c->stream));
    // a new collection: a slab of scratch the last one's long genes needed (up to 4 GB: run_align_classes) is not kept for it
    // (grow-only inside a collection; nothing of the context is in flight here)
    if (c->b_scratch.cap > ((size_t)512 << 20)) c->b_scratch.release();

    // ---- host-side indices, written straight into ONE page-locked staging buffer the context keeps (grow-only) and sent with
    // ONE copy (eleven pageable copies were 1.2 of the 1.9 ms of this stage at N = 2,000).  Several threads: genomes are
    // independent once every genome knows where its entries start.
    const int Wstride = W | 1;
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
    const size_t o_bitmap = place((size_t)N * Wstride * 8), o_rankpre = place((size_t)N * W * 4), o_gene_len = place((size_t)std::max(G, 1) * 4),
                 o_ent_off = place(((size_t)N + 1) * 4), o_nph = place((size_t)N * 4), o_ngen = place((size_t)N * 4), o_tlen = place((size_t)N * 8);
    const size_t o_ent = off;                                           // four entry arrays of E <= G elements follow
    const size_t cap_bytes = o_ent + 6 * (((size_t)std::max(G, 1) * 4 + 255) & ~(size_t)255) + (((size_t)N + 1) * 4 + 255) + (size_t)std::max(P, 1) * 4 + 1024;   // + the paralog lists + the dense-id table
    if (cap_bytes > c->h_stage_cap) {
        if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
        const size_t want = cap_bytes + cap_bytes / 8;
        hipError_t e = hipHostMalloc((void**)&c->h_stage, want, hipHostMallocDefault);
        if (e != hipSuccess) { pc_set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); c->h_stage = nullptr; return PC_ERR_HIP; }
        c->h_stage_cap = want;
    }
    uint8_t* hs = c->h_stage;
    uint64_t* bitmap = (uint64_t*)(hs + o_bitmap); uint32_t* rankpre = (uint32_t*)(hs + o_rankpre); int32_t* gene_len_h = (int32_t*)(hs + o_gene_len);
    uint32_t* ent_off = (uint32_t*)(hs + o_ent_off); int32_t* nph_h = (int32_t*)(hs + o_nph); int32_t* ngen_h = (int32_t*)(hs + o_ngen);
    int64_t* tlen_h = (int64_t*)(hs + o_tlen);
    memcpy(nph_h, g->nph, (size_t)N * 4); memcpy(ngen_h, g->ngen, (size_t)N * 4); memcpy(tlen_h, g->tlen, (size_t)N * 8);
    std::vector<int32_t> gene_len(G);
    std::vector<int> bad(N, 0);                          // per genome: 0 ok, else the error class found by the worker
    {   // gene lengths
        std::atomic<int> over(-1);                                   // one offender to report: any will do
        parallel_chunks(G, [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; ++k) {
                const int64_t len = g->seq_off[k + 1] - g->seq_off[k];
                if (len < 0 || len > 65535) { over.store((int)k, std::memory_order_relaxed); gene_len[k] = 0; } else gene_len[k] = (int32_t)len;
                gene_len_h[k] = gene_len[k];
            }
        }, 65536);                                                   // (a thread costs ~30 us to start: few of them for small inputs)
        if (over.load() >= 0) {
            const int k = over.load();
            pc_set_error("pc_upload: gene %d has length %lld (limit 65535)", k, (long long)(g->seq_off[k + 1] - g->seq_off[k])); return PC_ERR_LIMIT;
        }
    }
    int maxlen = 0, minlen = G ? 0x7fffffff : 0;
    for (int k = 0; k < G; ++k) { maxlen = std::max(maxlen, (int)gene_len[k]); minlen = std::min(minlen, (int)gene_len[k]); }
    for (int s = 0; s < N; ++s) {
        const int64_t k0 = g->gene_off[s], k1 = g->gene_off[s + 1];
        if (k1 < k0 || k1 > G) { pc_set_error("pc_upload: gene_off not monotone at genome %d", s); return PC_ERR_ARG; }
    }
    // pass 1: entries (distinct phams) per genome = popcount of its bitmap row; the gene list is checked against it in pass 2
    ent_off[0] = 0;
    parallel_chunks(N, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            uint64_t* row = bitmap + (size_t)s * Wstride;
            memcpy(row, g->bitmap + (size_t)s * W, sizeof(uint64_t) * W);
            for (int w = W; w < Wstride; ++w) row[w] = 0;
            size_t bits = 0;
            for (int w = 0; w < W; ++w) bits += (size_t)__builtin_popcountll(row[w]);
            ent_off[(size_t)s + 1] = (uint32_t)bits;
        }
    }, 1024);
    {   // prefix sum in 64 bits, refused as soon as it passes the gene count (a malformed bitmap must not wrap the 32-bit offsets)
        uint64_t run = 0;
        for (int s = 0; s < N; ++s) {
            run += ent_off[(size_t)s + 1];
            if (run > (uint64_t)G) { pc_set_error("pc_upload: the bitmap holds more phams than there are genes"); return PC_ERR_ARG; }
            ent_off[(size_t)s + 1] = (uint32_t)run;
        }
    }
    const size_t E = ent_off[N];
    const size_t ent_stride = ((size_t)std::max<size_t>(E, 1) * 4 + 255) & ~(size_t)255;
    int32_t* ent_cnt = (int32_t*)(hs + o_ent); int32_t* ent_len = (int32_t*)(hs + o_ent + ent_stride);
    int32_t* ent_gene = (int32_t*)(hs + o_ent + 2 * ent_stride); int32_t* ent_pham = (int32_t*)(hs + o_ent + 3 * ent_stride);
    // pass 2: a genome's entries, its rank table, and the consistency checks
    parallel_chunks(N, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            const int64_t k0 = g->gene_off[s], k1 = g->gene_off[s + 1];
            const uint64_t* row = bitmap + (size_t)s * Wstride;
            const size_t ent0 = ent_off[s], cap = ent_off[(size_t)s + 1] - ent0;
            size_t ne = 0; int64_t tl = 0; int err = 0;
            for (int64_t k = k0; k < k1 && !err;) {
                const int32_t p = g->gene_pham[k];
                if (p < 0 || p >= P || !((row[p >> 6] >> (p & 63)) & 1ULL) || (k > k0 && g->gene_pham[k - 1] >= p && g->gene_pham[k - 1] != p)) { err = 1; break; }
                int64_t k2 = k; int64_t ln = 0;
                while (k2 < k1 && g->gene_pham[k2] == p) { ln += gene_len[k2]; ++k2; }
                if (ne >= cap) { err = 2; break; }
                ent_cnt[ent0 + ne] = (int32_t)(k2 - k); ent_len[ent0 + ne] = (int32_t)ln; ent_gene[ent0 + ne] = (int32_t)k; ent_pham[ent0 + ne] = p;
                ++ne; tl += ln; k = k2;
            }
            size_t bits = 0;
            for (int w = 0; w < W; ++w) { rankpre[(size_t)s * W + w] = (uint32_t)(ent0 + bits); bits += (size_t)__builtin_popcountll(row[w]); }
            if (!err && (ne != cap || (int)ne != g->nph[s] || (int)(k1 - k0) != g->ngen[s] || tl != g->tlen[s])) err = 2;
            bad[s] = err;
        }
    }, 512);
    for (int s = 0; s < N; ++s) {
        if (bad[s] == 1) { pc_set_error("pc_upload: genome %d: a gene's pham id is out of order or not in the bitmap", s); return PC_ERR_ARG; }
        if (bad[s] == 2) { pc_set_error("pc_upload: genome %d: bitmap/nph/ngen/tlen disagree with its gene list", s); return PC_ERR_ARG; }
    }
    // phams an average genome pair shares = sum over phams of holders (holders - 1) / (N (N - 1)): what decides between the set metrics' kernels
    std::vector<uint32_t> holders((size_t)std::max(P, 1), 0u);
    {
        std::mutex merge;
        parallel_chunks((int64_t)E, [&](int64_t e0, int64_t e1) {
            std::vector<uint32_t> mine((size_t)std::max(P, 1), 0u);
            for (int64_t e = e0; e < e1; ++e) ++mine[(size_t)ent_pham[e]];
            std::lock_guard<std::mutex> lock(merge);
            for (int p2 = 0; p2 < P; ++p2) holders[(size_t)p2] += mine[(size_t)p2];
        }, 1 << 17);
        double inc = 0.0;
        for (uint32_t n : holders) inc += (double)n * (double)(n > 0 ? n - 1 : 0);
        c->avg_shared = N > 1 ? inc / ((double)N * (double)(N - 1)) : 0.0;
    }
    // paralog lists (pocp): a genome's entries with more than one gene, as (pham, count - 1).  conserved proteins of a pair =
    // 2 x shared phams + the excess counts of the shared paralog phams, and only ~6 % of the entries are paralogs
    const size_t o_para_off = (o_ent + 4 * ent_stride + 255) & ~(size_t)255;
    uint32_t* para_off = (uint32_t*)(hs + o_para_off);
    const size_t o_para = (o_para_off + ((size_t)N + 1) * 4 + 255) & ~(size_t)255;
    size_t n_para = 0;
    for (size_t e = 0; e < E; ++e) n_para += ent_cnt[e] > 1;
    const size_t para_stride = ((size_t)std::max<size_t>(n_para, 1) * 4 + 255) & ~(size_t)255;
    int32_t* para_pham = (int32_t*)(hs + o_para); int32_t* para_ex = (int32_t*)(hs + o_para + para_stride);
    {
        size_t at = 0; int max_ngen = 0;
        for (int s2 = 0; s2 < N; ++s2) {
            para_off[s2] = (uint32_t)at;
            for (size_t e = ent_off[s2]; e < ent_off[(size_t)s2 + 1]; ++e)
                if (ent_cnt[e] > 1) { para_pham[at] = ent_pham[e]; para_ex[at] = ent_cnt[e] - 1; ++at; }
            max_ngen = std::max(max_ngen, (int)g->ngen[s2]);
        }
        para_off[N] = (uint32_t)at;
        c->max_ngen = max_ngen;
        c->h_sp_n.assign((size_t)N, 0u);
        int max_ent_len = 0;
        for (int s2 = 0; s2 < N; ++s2) {
            uint32_t n = 0;
            for (size_t e = ent_off[s2]; e < ent_off[(size_t)s2 + 1]; ++e) { n += holders[(size_t)ent_pham[e]] >= 2u; max_ent_len = std::max(max_ent_len, (int)ent_len[e]); }
            c->h_sp_n[(size_t)s2] = n;
        }
        c->max_ent_len = max_ent_len;
    }
    // The 64 x 64 sparse tile kernel's own lists: only phams that at least TWO genomes hold (nothing else can be shared; in real
    // collections about half of all phams have one holder), renumbered densely in pham order, each entry as (dense id, value) pairs
    // for one 8-byte load, plus the dense ids alone (gcs / jc) and a rank table over 64-id words of the dense space -- its mask
    // chunks then cover the phams that matter, not the vocabulary.  Only the renumbering table is staged; the lists are made on the
    // device (k_sp_build, one thread per genome, each genome's kept entries at the start of its own slot of the entry arrays), as
    // are (pham, summed length) and (pham, gene count) over ALL entries in original ids for the 32 x 32 kernel (k_pair_entries).
    int P2 = 0;
    const size_t o_dense = (o_para + 2 * para_stride + 255) & ~(size_t)255;
    {
        int32_t* dense = (int32_t*)(hs + o_dense);
        for (int p2 = 0; p2 < P; ++p2) dense[p2] = holders[(size_t)p2] >= 2u ? P2++ : -1;
    }
    const int W2 = std::max(1, (P2 + 63) / 64);
    const size_t total_staged = o_dense + (((size_t)std::max(P, 1) * 4 + 255) & ~(size_t)255);
    const size_t o_pair_len = (total_staged + 255) & ~(size_t)255, o_pair_cnt = o_pair_len + 2 * ent_stride;
    const size_t o_sp_end = o_pair_cnt + 2 * ent_stride, o_sp_pham = (o_sp_end + (size_t)N * 4 + 255) & ~(size_t)255, o_sp_len = o_sp_pham + ent_stride,
                 o_sp_cnt = o_sp_len + 2 * ent_stride, o_sp_rank = o_sp_cnt + 2 * ent_stride;
    const size_t total_bytes2 = o_sp_rank + (((size_t)N * W2 * 4 + 255) & ~(size_t)255);
    lap("entries, rank table");
    if ((rc = abi_rc(c->b_sets.ensure(total_bytes2)))) return rc;
    // (an idle GPU answers its first command after 10-25 ms, whatever the command -- DMA copy, blocking copy or a copy
    // kernel all showed it when uploads followed each other with nothing in between, `tools/upload_timing.py`; that is the
    // device waking up, not this copy: behind a fill the same copy takes 0.3 ms)
    PC_HIP(hipMemcpyAsync(c->b_sets.p, hs, total_staged, hipMemcpyHostToDevice, c->stream));
    {
        uint8_t* dsb = (uint8_t*)c->b_sets.p;
        if ((rc = pc_launch_pair_entries((const int32_t*)(dsb + o_ent + 3 * ent_stride), (const int32_t*)(dsb + o_ent + ent_stride), (const int32_t*)(dsb + o_ent),
                                         (uint2*)(dsb + o_pair_len), (uint2*)(dsb + o_pair_cnt), (int64_t)E, c->stream))) return rc;
        if ((rc = pc_launch_sp_build(N, (const uint32_t*)(dsb + o_ent_off), (const int32_t*)(dsb + o_ent + 3 * ent_stride), (const int32_t*)(dsb + o_ent + ent_stride),
                                     (const int32_t*)(dsb + o_ent), (const int32_t*)(dsb + o_dense), W2, (int32_t*)(dsb + o_sp_pham), (uint2*)(dsb + o_sp_len),
                                     (uint2*)(dsb + o_sp_cnt), (uint32_t*)(dsb + o_sp_rank), (uint32_t*)(dsb + o_sp_end), c->stream))) return rc;
    }
    PC_HIP(hipStreamSynchronize(c->stream));
    lap("h2d sets");
    uint8_t* ds = (uint8_t*)c->b_sets.p;
    PcDev& d = c->dev;
    memset(&d, 0, sizeof(d));
    d.N = N; d.Wb = W; d.Wstride = Wstride; d.G = G; d.E = (int64_t)E; d.n_cu = c->n_cu;
    d.bitmap = (const uint64_t*)(ds + o_bitmap); d.rankpre = (const uint32_t*)(ds + o_rankpre);
    d.ent_cnt = (const int32_t*)(ds + o_ent); d.ent_len = (const int32_t*)(ds + o_ent + ent_stride);
    d.ent_gene = (const int32_t*)(ds + o_ent + 2 * ent_stride); d.ent_pham = (const int32_t*)(ds + o_ent + 3 * ent_stride);
    d.gene_len = (const int32_t*)(ds + o_gene_len); d.ent_off = (const uint32_t*)(ds + o_ent_off);
    d.nph = (const int32_t*)(ds + o_nph); d.ngen = (const int32_t*)(ds + o_ngen); d.tlen = (const int64_t*)(ds + o_tlen);
    d.para_off = (const uint32_t*)(ds + o_para_off); d.para_pham = (const int32_t*)(ds + o_para); d.para_ex = (const int32_t*)(ds + o_para + para_stride);
    d.ent_pair_len = (const uint2*)(ds + o_pair_len); d.ent_pair_cnt = (const uint2*)(ds + o_pair_cnt);
    d.sp_end = (const uint32_t*)(ds + o_sp_end); d.sp_pham = (const int32_t*)(ds + o_sp_pham); d.sp_len = (const uint2*)(ds + o_sp_len);
    d.sp_cnt = (const uint2*)(ds + o_sp_cnt); d.sp_rank = (const uint32_t*)(ds + o_sp_rank); d.sp_W = W2;
    c->h_gene_len.swap(gene_len);
    c->max_gene_len = maxlen; c->min_gene_len = minlen;
    c->max_nph = 0;
    c->max_tlen = 0;
    for (int s2 = 0; s2 < N; ++s2) { c->max_nph = std::max(c->max_nph, (int)g->nph[s2]); c->max_tlen = std::max(c->max_tlen, (int64_t)g->tlen[s2]); }
    c->n_residue_bytes_in = g->seq_off[G] - g->seq_off[0];
    rc = apply_shard(c, 0, 1);
    if (rc != PC_OK) return rc;
    lap("device copies (sets)");
    c->uploaded = true;
    return PC_OK;
}

// Upload, part 2: what the aligner needs -- residue codes, the distinct sequences and their ranks, the launch classes.
// g must be the packed genomes part 1 was given.
static int upload_residues(pc_ctx* c, const pc_packed* g) {
    PcRange range("pc:upload_residues");
    int rc = PC_OK;
    const int N = g->n_genomes;
    const int G = c->dev.G;
    if (N != c->dev.N || g->gene_off[N] != (int64_t)G || g->seq_off[G] - g->seq_off[0] != c->n_residue_bytes_in) {
        pc_set_error("pc_upload_residues: not the genomes pc_upload_sets was given (N %d/%d, genes %lld/%d)", N, c->dev.N, (long long)g->gene_off[N], G);
        return PC_ERR_ARG;
    }
    static const bool timing = getenv("PC_UPLOAD_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "pc_upload %-22s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    if ((rc = wait_last_work(c, nullptr, false))) return rc;
    c->residues_ready = false; c->plan.valid = false;
    PC_HIP(hipStreamSynchronize(c->stream));
    const std::vector<int32_t>& gene_len = c->h_gene_len;
    const int maxlen = c->max_gene_len;
    std::vector<int64_t> gene_off(G);
    int64_t code_bytes = 0;
    for (int k = 0; k < G; ++k) { gene_off[k] = code_bytes; code_bytes += ((int64_t)gene_len[k] + 15) & ~15LL; }
    // The residues go to the device RAW and are encoded there (k_encode: code LUT, 16-byte padding per gene): the host only
    // hashes them -- 8 raw bytes per multiply -- and notes which genes hold a byte outside the alphabet.  (r02 encoded on the
    // host: a 10^8-byte buffer to fault in, fill through the LUT byte by byte, copy and unmap per upload.)  Hashing the raw
    // bytes means two translations that differ only in letter case count as two sequences: they are aligned twice, nothing else.
    std::vector<uint64_t> ghash(std::max(G, 1));
    std::vector<uint8_t> godd(std::max(G, 1), 0);      // gene holds a byte outside the 24-letter alphabet (code >= 24)
    uint8_t lut[256]; build_code_lut(lut);
    uint8_t is_odd[256];
    for (int v = 0; v < 256; ++v) is_odd[v] = (uint8_t)(lut[v] >= 24);
    const uint8_t* raw = g->residues;
    // The hashing threads also copy their genes' bytes into a page-locked staging buffer the context keeps (up to 512 MB;
    // beyond that the residues go over from the caller's pageable memory at the end), and the DMA to HBM is started as soon as
    // they are done: it runs behind the de-duplication and the launch-class tables below.
    const int64_t raw_bytes = g->seq_off[G] - g->seq_off[0];
    static const int64_t stage_max = getenv("PC_RAW_STAGE_MAX") ? atoll(getenv("PC_RAW_STAGE_MAX")) : ((int64_t)512 << 20);   // (test knob)
    const bool staged = raw_bytes > 0 && raw_bytes <= stage_max;
    if (staged && (size_t)raw_bytes > c->h_raw_cap) {
        if (c->h_raw) { (void)hipHostFree(c->h_raw); c->h_raw = nullptr; c->h_raw_cap = 0; }
        const size_t want = (size_t)raw_bytes + (size_t)raw_bytes / 8;
        hipError_t e = hipHostMalloc((void**)&c->h_raw, want, hipHostMallocDefault);
        if (e != hipSuccess) { pc_set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); c->h_raw = nullptr; return PC_ERR_HIP; }
        c->h_raw_cap = want;
    }
    if ((rc = abi_rc(c->b_raw.ensure((size_t)std::max<int64_t>(raw_bytes, 16))))) return rc;
    uint8_t* const stage = staged ? c->h_raw : nullptr;
    const int64_t raw0 = g->seq_off[0];
    parallel_chunks(G, [&](int64_t k0, int64_t k1) {
        if (stage) memcpy(stage + (g->seq_off[k0] - raw0), raw + g->seq_off[k0], (size_t)(g->seq_off[k1] - g->seq_off[k0]));
        for (int64_t k = k0; k < k1; ++k) {
            const uint8_t* src = raw + g->seq_off[k];
            const int len = gene_len[k];
            uint64_t h = 0x9e3779b97f4a7c15ULL ^ (uint64_t)len;
            uint8_t odd = 0;
            int i = 0;
            for (; i + 8 <= len; i += 8) {
                uint64_t w; memcpy(&w, src + i, 8);
                h = (h ^ w) * 0x9fb21c651e98df25ULL; h ^= h >> 32;
                odd |= (uint8_t)(is_odd[src[i]] | is_odd[src[i + 1]] | is_odd[src[i + 2]] | is_odd[src[i + 3]] | is_odd[src[i + 4]] | is_odd[src[i + 5]] |
                                 is_odd[src[i + 6]] | is_odd[src[i + 7]]);
            }
            uint64_t w = 0;
            for (int j = 0; i + j < len; ++j) { w |= (uint64_t)src[i + j] << (8 * j); odd |= is_odd[src[i + j]]; }
            h = (h ^ w) * 0x9fb21c651e98df25ULL; h ^= h >> 32;
            ghash[k] = h ^ (h >> 29);
            godd[k] = odd;
        }
    });
    if (staged) PC_HIP(hipMemcpyAsync(c->b_raw.p, stage, (size_t)raw_bytes, hipMemcpyHostToDevice, c->stream));
    lap("residue hashes");
    // distinct sequences (by raw residues).  Alignments are planned per distinct
    // (row sequence, column sequence) pair, so every sequence gets a rank q; ranks follow launch-class order: column
    // sequences grouped by the kernel variant that aligns against them and by lanes-per-segment bucket (the
    // profile's LDS footprint scales with it, and LDS sets occupancy)
    std::vector<int32_t> uid(G), u_gene;
    {   // Representative of a gene = the first gene with the same residues.  Sixteen hash partitions, one thread and one
        // open-addressing table each (every thread scans all hashes and takes its own: genes arrive in index order, so the first
        // one in is the first occurrence); equal hash and length are confirmed by comparing the residues.  Sequence ids then
        // follow first-occurrence order, exactly as a single serial table would number them.
        constexpr int NPART = 16;
        std::vector<int32_t> rep(G);
        std::vector<std::thread> th;
        const int nthreads = G >= 32768 ? NPART : 1;
        auto work = [&](int part, int nparts) {
            size_t cap = 16;
            while (cap < (size_t)G * 2 / (size_t)nparts + 16) cap <<= 1;
            std::vector<int32_t> slot(cap, -1);
            for (int k = 0; k < G; ++k) {
                if (nparts > 1 && (int)((ghash[k] >> 40) & (NPART - 1)) != part) continue;
                size_t pos = (size_t)ghash[k] & (cap - 1);
                for (;; pos = (pos + 1) & (cap - 1)) {
                    const int32_t r = slot[pos];
                    if (r < 0) { slot[pos] = k; rep[k] = k; break; }
                    if (ghash[r] == ghash[k] && gene_len[r] == gene_len[k] &&
                        !memcmp(raw + g->seq_off[r], raw + g->seq_off[k], (size_t)gene_len[k])) { rep[k] = r; break; }
                }
            }
        };
        if (nthreads == 1) work(0, 1);
        else {
            for (int t = 0; t < NPART; ++t) th.emplace_back(work, t, NPART);
            for (auto& x : th) x.join();
        }
        u_gene.reserve(G);
        for (int k = 0; k < G; ++k) {
            if (rep[k] == k) { uid[k] = (int32_t)u_gene.size(); u_gene.push_back(k); }
            else uid[k] = uid[rep[k]];
        }
    }
    const int U = (int)u_gene.size();
    lap("distinct sequences");
    const int ncls_all = pc_num_classes();             // last class: general kernel
    std::vector<int> u_cls(U), len_cls(maxlen + 1, -1), len_rows(maxlen + 1, 0), len_var(maxlen + 1, -1);   // per length: class, rows per task, variant
    std::vector<int64_t> cls_count(ncls_all, 0);
    for (int u = 0; u < U; ++u) {
        const int len = gene_len[u_gene[u]];
        if (len_cls[len] < 0) {
            const int variant = pc_nw_choose_variant(len);
            len_var[len] = variant;
            len_cls[len] = pc_class_of(len, variant, false);
            len_rows[len] = pc_nw_task_rows(len, variant, 0);
        }
        // A column sequence with a byte outside the alphabet goes to its variant's "any byte" class: the profile cell
        // takes "identical residues" from a profile row per alphabet letter plus ONE row for every other byte
        // (pc_nw.hip, PC_INC16_MAX_W), which is exact only while the column holds none of those (as a row, it is fine)
        u_cls[u] = godd[u_gene[u]] ? pc_class_of(len, len_var[len], true) : len_cls[len]; ++cls_count[u_cls[u]];
    }
    lap("  classes per sequence");
    if (ncls_all > 250 || ncls_all * PC_WAVE_MODES + 1 > 1000) { pc_set_error("too many kernel classes"); return PC_ERR_LIMIT; }   // base class ids travel in a byte, 255 = none; the plan read-back holds 1,000 words
    c->ncls_all = ncls_all;
    std::vector<int64_t> cls_pos(ncls_all, 0);
    { int64_t run = 0; for (int cls = 0; cls < ncls_all; ++cls) { cls_pos[cls] = run; run += cls_count[cls]; } }
    std::vector<int32_t> q_gene(std::max(U, 1), 0), task_rows(std::max(U, 1), PC_TASK_ROWS);
    std::vector<uint32_t> q_of_u(std::max(U, 1), 0), gene_q(std::max(G, 1), 0);
    std::vector<uint8_t> q_class(std::max(U, 1), 0), q_nseg(std::max(U, 1), 1), rem_class((size_t)std::max(U, 1) * 16, 255);
    // per length: segments per wave of the main variant and where a remainder of r rows goes (class id, 255 = stays)
    std::vector<uint8_t> len_nseg(maxlen + 1, 1), len_rem((size_t)(maxlen + 1) * 16, 255);
    c->cls_max_lb.assign(ncls_all, 0);
    for (int len = 0; len <= maxlen; ++len) if (len_cls[len] >= 0) c->cls_max_lb[len_cls[len]] = std::max(c->cls_max_lb[len_cls[len]], len);
    for (int u = 0; u < U; ++u) if (godd[u_gene[u]]) c->cls_max_lb[u_cls[u]] = std::max(c->cls_max_lb[u_cls[u]], (int)gene_len[u_gene[u]]);
    // (the remainder chooser is a cost model evaluated ~15 times per distinct length: several threads, then the class maxima)
    parallel_chunks(maxlen, [&](int64_t l0, int64_t l1) {
        for (int64_t len = l0 + 1; len <= l1; ++len) {
            const int v = len_var[len];
            if (len_cls[len] < 0 || v < 0) continue;
            const int Wv = pc_nw_variant_w(v), Gv = ((int)len + Wv - 1) / Wv;
            const int nseg = std::max(1, std::min(64 / Gv, 16));        // (Gv > 64: a strip-mined gene, one row per wave)
            len_nseg[len] = (uint8_t)nseg;
            for (int r = 1; r < nseg; ++r) {
                const int vr = pc_nw_choose_remainder((int)len, r, v);
                if (vr >= 0) len_rem[(size_t)len * 16 + r] = (uint8_t)pc_class_of((int)len, vr, false);
            }
        }
    }, 64);
    for (int len = 1; len <= maxlen; ++len)
        for (int r = 1; r < 16; ++r) {
            const uint8_t cr = len_rem[(size_t)len * 16 + r];
            if (cr != 255) c->cls_max_lb[cr] = std::max(c->cls_max_lb[cr], len);
        }
    lap("  remainder chooser");
    // ranks inside a class follow sequence length (then first occurrence): the plan's sort then hands every bucket its
    // rows in length order, so the row streams of a task, dealt round-robin, stay in step and start their alignments
    // in the same steps (the per-step cost of an alignment start is paid once per wave, not once per segment; measured
    // gain 0.3 %: rows of one pham are nearly equally long anyway)
    std::vector<int32_t> u_order(U);
    {   // stable counting sort by length (lengths <= 65,535): a comparison sort of ~5*10^5 sequences cost 60 ms of the upload
        std::vector<int32_t> at(maxlen + 2, 0);
        for (int u = 0; u < U; ++u) ++at[gene_len[u_gene[u]] + 1];
        for (int len = 0; len <= maxlen; ++len) at[len + 1] += at[len];
        for (int u = 0; u < U; ++u) u_order[at[gene_len[u_gene[u]]]++] = u;
    }
    for (int u : u_order) q_of_u[u] = (uint32_t)cls_pos[u_cls[u]]++;        // (serial: a rank is its predecessors' count)
    lap("  ranks");
    {   // "any byte" classes a remainder may be sent to: their longest column gene (serial, rare)
        for (int u = 0; u < U; ++u) {
            const int len = gene_len[u_gene[u]];
            if (!godd[u_gene[u]] || u_cls[u] == len_cls[len] || u_cls[u] == ncls_all - 1) continue;
            for (int r = 1; r < 16; ++r) {
                const uint8_t cr = len_rem[(size_t)len * 16 + r];
                if (cr == 255) continue;
                const int ca = pc_class_of(len, pc_class_variant(cr), true);
                c->cls_max_lb[ca] = std::max(c->cls_max_lb[ca], len);
            }
        }
    }
    parallel_chunks(U, [&](int64_t u0, int64_t u1) {
        for (int64_t u = u0; u < u1; ++u) {
            const int q = (int)q_of_u[u];
            const int len = gene_len[u_gene[u]];
            q_gene[q] = u_gene[u];
            q_class[q] = (uint8_t)u_cls[u];
            if (u_cls[u] == ncls_all - 1) { task_rows[q] = pc_nw_task_rows(len, -1, 0); q_nseg[q] = 1; }   // general kernel (rem_class stays 255: no remainder move)
            else if (!godd[u_gene[u]] || u_cls[u] == len_cls[len]) { task_rows[q] = len_rows[len]; q_nseg[q] = len_nseg[len]; memcpy(&rem_class[(size_t)q * 16], &len_rem[(size_t)len * 16], 16); }
            else {                                                   // "any byte" class: its own task size, remainders to the "any byte" class of their variant
                task_rows[q] = pc_nw_task_rows(len, len_var[len], 1); q_nseg[q] = len_nseg[len];
                for (int r = 1; r < 16; ++r) {
                    const uint8_t cr = len_rem[(size_t)len * 16 + r];
                    if (cr == 255) continue;
                    rem_class[(size_t)q * 16 + r] = (uint8_t)pc_class_of(len, pc_class_variant(cr), true);
                }
            }
        }
    });
    parallel_chunks(G, [&](int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; ++k) gene_q[k] = q_of_u[uid[k]]; });
    int ubits = 1;
    while ((1LL << ubits) < U) ++ubits;

    lap("launch classes");
    // ---- device copies ---------------------------------------------------------------
    // The tables go through the context's page-locked staging buffer (one memcpy each, then DMA): sent with hipMemcpy straight from
    // pageable vectors, the larger ones (rem_class 16 B and gene_off 8 B per sequence) left the driver ~20 ms of deferred work
    // that the FIRST kernel launch after the upload then waited for (tools/wake_experiment.py: fill 622 ms on the host clock
    // against 599 on the device right after an upload, 599 / 599 after 50 ms of sleep).
    std::vector<int64_t> rel(g->seq_off, g->seq_off + G + 1);
    for (auto& x : rel) x -= g->seq_off[0];
    struct Item { DevBuf* buf; const void* src; size_t bytes; };
    const Item items[] = {
        {&c->b_gene_off, gene_off.data(), gene_off.size() * 8}, {&c->b_seq_tmp, rel.data(), rel.size() * 8},
        {&c->b_gene_q, gene_q.data(), gene_q.size() * 4}, {&c->b_q_gene, q_gene.data(), q_gene.size() * 4},
        {&c->b_task_rows, task_rows.data(), task_rows.size() * 4}, {&c->b_q_class, q_class.data(), q_class.size()},
        {&c->b_q_nseg, q_nseg.data(), q_nseg.size()}, {&c->b_rem_class, rem_class.data(), rem_class.size()}};
    size_t stage_total = 0;
    for (const Item& it : items) stage_total += (it.bytes + 255) & ~(size_t)255;
    if (stage_total > c->h_stage_cap) {                       // (part 1's copy out of this buffer has completed)
        if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
        const size_t want = stage_total + stage_total / 8;
        hipError_t e = hipHostMalloc((void**)&c->h_stage, want, hipHostMallocDefault);
        if (e != hipSuccess) { pc_set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); c->h_stage = nullptr; return PC_ERR_HIP; }
        c->h_stage_cap = want;
    }
    const size_t codes_size = (size_t)std::max<int64_t>(code_bytes, 16);
    {
        size_t at = 0;
        for (const Item& it : items) {
            if ((rc = abi_rc(it.buf->ensure(std::max<size_t>(it.bytes, 16))))) return rc;
            if (!it.bytes) continue;
            memcpy(c->h_stage + at, it.src, it.bytes);
            PC_HIP(hipMemcpyAsync(it.buf->p, c->h_stage + at, it.bytes, hipMemcpyHostToDevice, c->stream));
            at += (it.bytes + 255) & ~(size_t)255;
        }
        // raw residues (already on their way when staged) -> codes, on the device
        PcLut lut_arg;
        memcpy(lut_arg.v, lut, 256);
        if ((!staged && (rc = upload_raw(c->b_raw, raw + g->seq_off[0], (size_t)raw_bytes))) || (rc = abi_rc(c->b_codes.ensure(codes_size))) ||
            (rc = abi_rc(c->b_cls_begin.ensure(((size_t)ncls_all * PC_WAVE_MODES + 1) * 4)))) return rc;
        if (code_bytes < 16) PC_HIP(hipMemsetAsync(c->b_codes.p, PC_PADCODE, 16, c->stream));
        rc = pc_launch_encode(c->b_raw.as<uint8_t>(), c->b_seq_tmp.as<int64_t>(), c->b_gene_off.as<int64_t>(), c->dev.gene_len, lut_arg,
                              c->b_codes.as<uint8_t>(), G, c->stream);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc != PC_OK) return rc;
        if (e != hipSuccess) { pc_set_error("pc_upload: residue tables / encoding: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
        // The raw bytes and their offsets were staging for k_encode only.  Kept (grow-only) they would double the residue footprint
        // for the life of the context and shrink what plan_budget_bytes() sees as free -- more chunks for exactly the collections
        // that are chunked; small ones keep them, so that repeated uploads do not pay a hipMalloc each (threshold 256 MB).
        if (c->b_raw.cap + c->b_seq_tmp.cap > ((size_t)256 << 20)) { c->b_raw.release(); c->b_seq_tmp.release(); }
    }
    c->task_plan.task_rows = c->b_task_rows.as<int32_t>(); c->task_plan.q_class = c->b_q_class.as<uint8_t>();
    c->task_plan.q_nseg = c->b_q_nseg.as<uint8_t>(); c->task_plan.rem_class = c->b_rem_class.as<uint8_t>();
    c->task_plan.nvar = pc_nw_num_variants(); c->task_plan.small_modes = pc_nw_small_modes_enabled(); c->task_plan.n_strip = PC_STRIP_CLASSES; c->task_plan.pad_ = 0;
    for (int v = 0; v < 32; ++v) c->task_plan.variant_w[v] = v < pc_nw_num_variants() ? pc_nw_variant_w(v) : 0;
    {   // launch classes: every base class in its three workgroup shapes, each with the base class's longest column gene
        std::vector<int32_t> per_base; per_base.swap(c->cls_max_lb);
        c->nlc = ncls_all * PC_WAVE_MODES;
        c->cls_max_lb.resize((size_t)c->nlc);
        for (int lc = 0; lc < c->nlc; ++lc) c->cls_max_lb[(size_t)lc] = per_base[(size_t)(lc / PC_WAVE_MODES)];
    }
    PcDev& d = c->dev;
    d.U = U; d.ubits = ubits; d.gene_q = c->b_gene_q.as<uint32_t>(); d.q_gene = c->b_q_gene.as<int32_t>();
    d.gene_off = c->b_gene_off.as<int64_t>(); d.codes = c->b_codes.as<uint8_t>();
    c->h_gene_odd.swap(godd);
    lap("device copies (residues)");
    c->residues_ready = true;
    return PC_OK;
}

extern "C" int pc_upload_sets(pc_ctx* c, const pc_packed* g) {
    if (!c || !g) { pc_set_error("pc_upload_sets: NULL argument"); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    return upload_sets(c, g);
}
extern "C" int pc_upload_residues(pc_ctx* c, const pc_packed* g) {
    if (!c || !g) { pc_set_error("pc_upload_residues: NULL argument"); return PC_ERR_ARG; }
    if (!c->uploaded) { pc_set_error("pc_upload_residues: pc_upload_sets first"); return PC_ERR_STATE; }
    PC_ON_DEVICE(c);
    if (c->residues_ready) return PC_OK;
    return upload_residues(c, g);
}
extern "C" int pc_upload(pc_ctx* c, const pc_packed* g) {
    if (!c || !g) { pc_set_error("pc_upload: NULL argument"); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    int rc = upload_sets(c, g);
    if (rc == PC_OK) rc = upload_residues(c, g);
    if (rc != PC_OK) c->uploaded = false;
    return rc;
}

extern "C" int pc_set_shard(pc_ctx* c, int rank, int world) {
    if (!c || !c->uploaded) { pc_set_error("pc_set_shard: upload first"); return PC_ERR_STATE; }
    if (world < 1 || rank < 0 || rank >= world) { pc_set_error("pc_set_shard: rank %d of %d", rank, world); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    int rc = wait_last_work(c, nullptr, false); if (rc != PC_OK) return rc;
    PC_HIP(hipStreamSynchronize(c->stream));
    return apply_shard(c, rank, world);
}
// Cost-balanced deal.  The boustrophedon deal balances pair counts; alignment work per target genome also follows its
// gene count and how much it shares with the genomes before it.  One COUNT walk over all pairs gives the DP cells per
// target (integer sums: identical on every rank), then targets go, heaviest first, to the rank with the least work
// so far (ties: lowest rank) -- the same static, host-decided partition on every rank, no communication.
extern "C" int pc_set_shard_balanced(pc_ctx* c, int rank, int world) {
    if (!c || !c->uploaded) { pc_set_error("pc_set_shard_balanced: upload first"); return PC_ERR_STATE; }
    if (world < 1 || rank < 0 || rank >= world) { pc_set_error("pc_set_shard_balanced: rank %d of %d", rank, world); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    int rc = wait_last_work(c, nullptr, false); if (rc != PC_OK) return rc;
    PC_HIP(hipStreamSynchronize(c->stream));
    const int N = c->dev.N;
    if (c->target_cost.empty()) {
        if ((rc = apply_shard(c, 0, 1))) return rc;                       // walk every pair
        if ((rc = c->b_cost.ensure((size_t)N * 8)) || (rc = c->b_totals.ensure(64))) return abi_rc(rc);
        PC_HIP(hipMemsetAsync(c->b_cost.p, 0, (size_t)N * 8, c->stream));
        PC_HIP(hipMemsetAsync(c->b_totals.p, 0, 64, c->stream));
        PcWalkArgs a; memset(&a, 0, sizeof(a));
        a.totals = c->b_totals.as<unsigned long long>(); a.cost_t = c->b_cost.as<unsigned long long>(); a.condensed = 1;
        if (c->dev.G > 0 && (rc = pc_launch_walk(PCW_COUNT, c->dev, c->shard, a, c->stream))) return rc;
        c->target_cost.resize(N);
        PC_HIP(hipMemcpyAsync(c->target_cost.data(), c->b_cost.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
        PC_HIP(hipStreamSynchronize(c->stream));
    }
    // every pair also costs a walk visit and an output value: a floor of 2,000 cell-equivalents per pair keeps the
    // set metrics and sparse data balanced too
    std::vector<int> order(N);
    std::iota(order.begin(), order.end(), 0);
    auto cost = [&](int t) { return c->target_cost[t] + (uint64_t)t * 2000u; };
    std::sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) != cost(y) ? cost(x) > cost(y) : x < y; });
    std::vector<uint64_t> load(world, 0);
    std::vector<int32_t> t_rank(std::max(N, 1), 0);
    for (int t : order) {
        int best = 0;
        for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
        t_rank[t] = best; load[best] += cost(t);
    }
    std::vector<int64_t> t_lbase(std::max(N, 1), 0), fill(world, 0);
    std::vector<int32_t> owned; std::vector<int64_t> lbase;
    for (int t = 0; t < N; ++t) {
        const int r = t_rank[t];
        t_lbase[t] = fill[r];
        if (r == rank) { owned.push_back(t); lbase.push_back(fill[r]); }
        fill[r] += t;
    }
    lbase.push_back(fill[rank]);
    c->shard_pairs = fill[rank];
    c->shard_stride = *std::max_element(fill.begin(), fill.end());
    c->rank = rank; c->world = world; c->balanced = true;
    if ((rc = upload_vec(c->b_owned, owned)) || (rc = upload_vec(c->b_lbase, lbase)) || (rc = upload_vec(c->b_t_rank, t_rank)) ||
        (rc = upload_vec(c->b_t_lbase, t_lbase))) return rc;
    c->h_t_rank = t_rank; c->h_t_lbase = t_lbase;
    c->h_owned = owned; c->h_lbase = lbase; c->plan.valid = false;
    c->shard.nown = (int32_t)owned.size();
    c->shard.ident = world == 1 ? 1 : 0;
    c->shard.owned = c->b_owned.as<int32_t>();
    c->shard.lbase = c->b_lbase.as<int64_t>();
    return PC_OK;
}
extern "C" int pc_shard_table(const pc_ctx* c, int32_t* t_rank, int64_t* t_lbase) {
    if (!c || !c->uploaded) { pc_set_error("pc_shard_table: upload first"); return PC_ERR_STATE; }
    if (!t_rank || !t_lbase) { pc_set_error("pc_shard_table: NULL argument"); return PC_ERR_ARG; }
    memcpy(t_rank, c->h_t_rank.data(), sizeof(int32_t) * (size_t)c->dev.N);
    memcpy(t_lbase, c->h_t_lbase.data(), sizeof(int64_t) * (size_t)c->dev.N);
    return PC_OK;
}
extern "C" int pc_target_costs(const pc_ctx* c, uint64_t* cost) {
    if (!c || !c->uploaded) { pc_set_error("pc_target_costs: upload first"); return PC_ERR_STATE; }
    if (!cost) { pc_set_error("pc_target_costs: NULL argument"); return PC_ERR_ARG; }
    if (c->target_cost.size() != (size_t)c->dev.N) { pc_set_error("pc_target_costs: no cost-balanced deal was computed for this upload (pc_set_shard_balanced)"); return PC_ERR_STATE; }
    memcpy(cost, c->target_cost.data(), sizeof(uint64_t) * (size_t)c->dev.N);
    return PC_OK;
}
extern "C" int64_t pc_shard_pairs(const pc_ctx* c) { return c && c->uploaded ? c->shard_pairs : -1; }
extern "C" int64_t pc_shard_stride(const pc_ctx* c) { return c && c->uploaded ? c->shard_stride : -1; }

// Step 5 of the plan: launch the alignment kernels for every launch class that has tasks.  Classes are
// independent (disjoint result slots), so their launches are spread over the caller's stream and
// seven auxiliary streams: the drain of one class overlaps the next one's start.
static int small_launch_min() {              // fewest tasks that earn a one- / two-wave mode a launch of its own
    static const int v = [] { const char* e = getenv("PC_SMALL_LAUNCH_MIN"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 192; }();
    return v;
}
static int run_align_classes(pc_ctx* c, const PcTask* task_list, const uint32_t* task_begin /*[nlc+1]*/, const int32_t* cls_max_lb,
                             uint2* res, hipStream_t st, pc_stats* stats, int ppos) {
    const int nbase = c->ncls_all;
    // One launch = a run of neighbouring launch classes of ONE base class, run in the workgroup shape of the first of them.  The
    // modes of a base class follow each other in the sorted task list (own shape, two waves, one wave), so a small-task mode with
    // too few tasks to pay for a launch of its own -- every launch holds its hardware queue until its last workgroup is done --
    // rides at the end of the launch before it: correct in any shape, merely less snug.
    struct Launch { uint32_t begin, end; int base, mode, max_lb; };
    std::vector<Launch> launches;
    for (int b = 0; b < nbase; ++b) {
        const uint32_t* tb = task_begin + (size_t)b * PC_WAVE_MODES;
        if (tb[PC_WAVE_MODES] == tb[0]) continue;
        const uint32_t n0 = tb[1] - tb[0], n1 = tb[2] - tb[1], n2 = tb[3] - tb[2];
        // (the wide variants' tasks are hundreds of times a small gene's, and their one- / two-row tasks run on another kernel
        // altogether -- narrow strip-mined passes: always worth a launch)
        const int bv = pc_class_variant(b);
        const uint32_t least = (bv >= 0 && pc_nw_variant_w(bv) >= 32) ? 1u : (uint32_t)small_launch_min();
        const bool own2 = n2 >= least, own1 = n1 + (own2 ? 0u : n2) >= least;      // one-wave tasks alone? two-wave (+ folded one-wave) alone?
        uint32_t at = tb[0];
        const int max_lb = cls_max_lb[b * PC_WAVE_MODES];
        auto put = [&](uint32_t n, int mode) { if (n) launches.push_back({at, at + n, b, mode, max_lb}); at += n; };
        if (own1) { put(n0, PC_MODE_CLASS); put(n1 + (own2 ? 0u : n2), n1 ? PC_MODE_TWO_WAVES : PC_MODE_ONE_WAVE); }
        else put(n0 + n1 + (own2 ? 0u : n2), n0 ? PC_MODE_CLASS : n1 ? PC_MODE_TWO_WAVES : PC_MODE_ONE_WAVE);
        if (own2) put(n2, PC_MODE_ONE_WAVE);
    }
    if (launches.empty()) return PC_OK;
    // longest tasks first (a task's duration grows with its column gene's length): the tail of the fill is then made
    // of short tasks
    std::stable_sort(launches.begin(), launches.end(), [&](const Launch& x, const Launch& y) {
        if (x.max_lb != y.max_lb) return x.max_lb > y.max_lb;
        return x.base != y.base ? x.base < y.base : x.mode < y.mode;
    });
    // Scratch slab (sized before anything is launched, never re-allocated between launches).  The general kernel's launches share
    // its first region and stay in order on the caller's stream.  Every strip-mined launch gets a region of its OWN behind it and one
    // of the context's long-task streams (pc_ctx::lng): a collection's long-gene launches are few tasks of tens of milliseconds each -- a 6,600 x 6,600
    // alignment on one wave takes 47 ms -- and lined up on one stream they were the critical path of the fill (synth_real(5000):
    // three strip launches, 112 + 41 + 48 ms end to end, the last two on a nearly empty chip, under a fill of 243 ms).  Only
    // what does not fit PC_SLAB_BUDGET (3 GB) shares the first region, in order, as before.
    // percent-positives: systolic where the profile cell can run (it reads "positive" from a table), general kernel elsewhere
    auto launch_variant = [&](const Launch& l) { const int v = pc_class_variant(l.base); return (ppos && !pc_nw_ppos_systolic(v, l.max_lb)) ? pc_nw_ppos_variant(l.max_lb) : v; };
    auto uses_slab = [&](const Launch& l) { const int v = launch_variant(l); return v < 0 || pc_launch_is_strip(v, l.max_lb, l.mode, ppos); };
    struct Region { size_t off, bytes; bool own; };
    std::vector<Region> region(launches.size(), Region{0, 0, false});
    static const size_t slab_budget = [] { const char* e = getenv("PC_SLAB_BUDGET"); const long long v = e ? atoll(e) : 0; return v > 0 ? (size_t)v : (size_t)3 << 30; }();
    static const bool strips_in_line = getenv("PC_STRIP_STREAMS") && !strcmp(getenv("PC_STRIP_STREAMS"), "0");     // A/B: the r04 order
    size_t sbytes = 0;
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    auto lay_out = [&](bool in_line) {                                    // regions of the slab; in_line: every strip-mined launch shares the first
        size_t shared = 0, own_total = 0;
        std::fill(region.begin(), region.end(), Region{0, 0, false});
        for (const Launch& l : launches) if (launch_variant(l) < 0) shared = std::max(shared, pc_nw_fallback_scratch_bytes(l.max_lb));
        for (size_t i = 0; i < launches.size(); ++i) {
            const Launch& l = launches[i];
            const int v = launch_variant(l);
            if (v < 0 || !pc_launch_is_strip(v, l.max_lb, l.mode, ppos)) continue;
            const size_t need = up256(pc_nw_strip_launch_bytes(l.mode, (int)(l.end - l.begin), c->max_gene_len, c->n_cu, ppos));
            if (!in_line && c->n_streams > 1 && own_total + need <= slab_budget) { region[i] = Region{own_total, need, true}; own_total += need; }
            else shared = std::max(shared, pc_nw_strip_scratch_bytes(c->max_gene_len, c->n_cu));
        }
        shared = up256(shared);
        for (size_t i = 0; i < launches.size(); ++i) {
            if (region[i].own) region[i].off += shared;
            else if (uses_slab(launches[i])) region[i] = Region{0, shared, false};
        }
        sbytes = shared + own_total;
        return own_total;
    };
    const size_t own_total = lay_out(strips_in_line);
    if (sbytes) {
        int rc = c->b_scratch.ensure(sbytes);
        // The slab's own regions can reach 3 GB + 1 GB shared, and their size follows the chip and the longest gene, not the chunk: a
        // chunked fill on a nearly full device would halve its chunk again and again without the slab getting any smaller.  So: once
        // more with every strip-mined launch in line on the shared region (fewer bytes, same values) before the chunk is given up.
        if (rc == PC_ERR_NOMEM_INTERNAL && own_total > 0) { lay_out(true); rc = sbytes ? c->b_scratch.ensure(sbytes) : PC_OK; }
        if (rc != PC_OK) return rc;
    }
    // Launch classes of one register tier, cell and workgroup size share ONE launch (k_nw_systolic_tier, pc_nw_fuse_key): the
    // hardware queues run launches back to back, each waiting for the last workgroup of the one before it, and a fill's ~80
    // launches cost it a task's duration each -- bundled they are ~15, each holding more tasks than the chip does at once.
    // Groups keep the order of their first member (longest column genes first); inside a group the classes follow that order too.
    struct Group { int key; std::vector<int> members; };
    std::vector<Group> groups;
    for (int i = 0; i < (int)launches.size(); ++i) {
        const Launch& l = launches[i];
        const int key = pc_nw_fuse_key(launch_variant(l), l.max_lb, ppos, pc_class_compare_only(l.base), l.mode);
        size_t g = groups.size();
        if (key >= 0) for (size_t k = 0; k < groups.size(); ++k) if (groups[k].key == key && groups[k].members.size() < PC_FUSE_MAX_SEGMENTS) { g = k; break; }
        if (g == groups.size()) groups.push_back({key, {}});
        groups[g].members.push_back(i);
    }
    // Launches of little work (fewer wave-tasks than four rounds of the chip's wave slots: a fill has two dozen, a millisecond or less
    // each) are ISSUED first: at the head of the streams they are done within the fill's first milliseconds.  In the order above the
    // last of them sat behind the launch that runs for most of the fill, in its hardware queue, and ran one after the other on an
    // empty chip when it ended -- the fill's fixed cost (profiles/r05/experiments/k4_fill_timeline.txt: 3.2 ms at N = 5,000, 2.6 of a
    // 75-ms rank of the 8-rank shard).  Worth -0.5 % at N = 2,000, nothing at 5,000, -1.0 % for that rank: T(w) = 3.3 + 548 / w.
    {
        const uint64_t small_below = (uint64_t)4 * 32 * (uint64_t)(c->n_cu > 0 ? c->n_cu : 256);
        std::stable_partition(groups.begin(), groups.end(), [&](const Group& grp) {
            uint64_t wave_tasks = 0;
            for (int i : grp.members) {
                const Launch& l = launches[i];
                if (region[i].bytes) return false;                             // (launches on the scratch slab keep their place)
                wave_tasks += (uint64_t)(l.end - l.begin) * (l.mode == PC_MODE_ONE_WAVE ? 1u : l.mode == PC_MODE_TWO_WAVES ? 2u : 4u);
            }
            return wave_tasks < small_below;
        });
    }
    constexpr int kAux = pc_ctx::kAux;
    const int n_aux = std::min((int)groups.size(), c->n_streams) - 1;        // auxiliary streams this fill uses
    int n_long = 0;                                                          // launches with a scratch region of their own: on the long-task streams
    for (const Region& rg : region) if (rg.own) ++n_long;
    n_long = std::min(n_long, (int)pc_ctx::kLong);
    PC_HIP(hipEventRecord(c->aux_ev[kAux], st));
    for (int k = 0; k < n_aux; ++k) PC_HIP(hipStreamWaitEvent(c->aux[k], c->aux_ev[kAux], 0));
    for (int k = 0; k < n_long; ++k) PC_HIP(hipStreamWaitEvent(c->lng[k], c->aux_ev[kAux], 0));
    int slot = 0, long_slot = 0, first_error = PC_OK;
    for (const Group& grp : groups) {
        int rc = PC_OK;
        if (grp.key >= 0) {
            PcNwSegment segs[PC_FUSE_MAX_SEGMENTS];
            int ns = 0;
            for (int i : grp.members) {
                const Launch& l = launches[i];
                segs[ns++] = {l.begin, l.end - l.begin, launch_variant(l), l.max_lb, pc_class_compare_only(l.base), l.mode};
            }
            hipStream_t ls = slot == 0 ? st : c->aux[slot - 1];
            rc = pc_launch_nw_group(segs, ns, c->dev, task_list, c->b_bucket_row.as<int32_t>(), nullptr /* result slot = position in the sorted list */,
                                    res, ppos, c->tie_rule, ls);
        } else {
            const Launch& l = launches[grp.members[0]];
            const int nt = (int)(l.end - l.begin);
            const int variant = launch_variant(l);
            // launches that share the slab's first region stay in order on the caller's stream
            const Region& rg = region[grp.members[0]];
            const bool slab = rg.bytes != 0;
            hipStream_t ls = rg.own ? c->lng[long_slot++ % n_long] : ((slab || slot == 0) ? st : c->aux[slot - 1]);
            rc = pc_launch_nw(variant, c->dev, task_list + l.begin, nt, c->b_bucket_row.as<int32_t>(),
                              nullptr, res, slab ? (void*)((char*)c->b_scratch.p + rg.off) : nullptr,
                              slab ? rg.bytes : 0, l.max_lb, ppos, c->tie_rule, pc_class_compare_only(l.base), ls, l.mode, c->max_gene_len);
        }
        if (rc != PC_OK) { first_error = rc; break; }
        if (stats) ++stats->n_align_launches;
        if (!(grp.key < 0 && region[grp.members[0]].own)) slot = (slot + 1) % (n_aux + 1);
    }
    // join the auxiliary streams back into the caller's stream -- also after a failed launch, so that what was
    // already queued on them is ordered before anything the caller does next
    for (int k = 0; k < n_aux; ++k) {
        PC_HIP(hipEventRecord(c->aux_ev[k], c->aux[k]));
        PC_HIP(hipStreamWaitEvent(st, c->aux_ev[k], 0));
    }
    for (int k = 0; k < n_long; ++k) {
        PC_HIP(hipEventRecord(c->lng_ev[k], c->lng[k]));
        PC_HIP(hipStreamWaitEvent(st, c->lng_ev[k], 0));
    }
    return first_error;
}

// ---- the three stages of an aai / peq fill.  pc_fill* run them back to back; the alignment-sliced multi-GPU route
// (pc_plan_dev, pc_align_slice_dev, pc_reduce_dev) runs them with a collective between the last two.

// Memory-bounded batching (the reference never holds more than ~10,000 pairs per CPU in flight, matrix.py:474-493, and so
// runs any N).  The plan buffers of an aai / peq fill take PC_PLAN_BYTES_PER_ALIGNMENT bytes per alignment; when that
// exceeds the budget -- or the 2^31-1 alignments a plan can index -- the fill runs plan -> align -> reduce over successive
// ranges of the shard's target genomes.  Cutting by target keeps every pair's alignments in one chunk.
#define PC_PLAN_BYTES_PER_ALIGNMENT 56
#define PC_PLAN_MAX_ALIGNMENTS 0x7ffffffeLL

// Pure host arithmetic, exported for tests: cut [0, n) into consecutive ranges whose sums stay <= max_per_chunk (a single
// element above it gets a range of its own).  chunk_begin receives the range starts followed by n (at most cap entries are
// written); returns the number of ranges.
extern "C" int pc_chunk_plan(const uint64_t* count, int n, uint64_t max_per_chunk, int32_t* chunk_begin, int cap) {
    if (!count || n < 0 || max_per_chunk == 0) { pc_set_error("pc_chunk_plan: bad argument"); return PC_ERR_ARG; }
    int nch = 0, start = 0; uint64_t run = 0;
    auto put = [&](int v) { if (chunk_begin && nch < cap) chunk_begin[nch] = v; ++nch; };
    if (n > 0) put(0);
    for (int k = 0; k < n; ++k) {                      // (the same rule as fill_aligned's loop: extend while the sum stays within the limit)
        if (k > start && (run + count[k] > max_per_chunk || run + count[k] < run)) { put(k); run = 0; start = k; }
        run += count[k];
    }
    if (chunk_begin && nch < cap) chunk_begin[nch] = n;
    return nch;
}

extern "C" int pc_set_plan_budget(pc_ctx* c, int64_t bytes) {
    if (!c || bytes < 0) { pc_set_error("pc_set_plan_budget: bad argument"); return PC_ERR_ARG; }
    c->plan_budget = bytes;
    return PC_OK;
}

// bytes one chunk's plan buffers may take: the caller's figure (pc_set_plan_budget / PC_PLAN_BYTES), else half of what
// is free now plus what the grow-only plan buffers already hold
static int64_t plan_budget_bytes(pc_ctx* c) {
    if (c->plan_budget > 0) return c->plan_budget;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return (int64_t)16 << 30; }
    const size_t held = c->b_key0.cap + c->b_key1.cap + c->b_val0.cap + c->b_val1.cap + c->b_flags.cap + c->b_excl.cap + c->b_alias.cap +
                        c->b_bucket_row.cap + c->b_res.cap + c->b_sort_tmp.cap;
    return (int64_t)((free_b + held) / 2);
}
static void release_plan_buffers(pc_ctx* c) {
    DevBuf* bufs[] = {&c->b_key0, &c->b_key1, &c->b_val0, &c->b_val1, &c->b_flags, &c->b_excl, &c->b_alias, &c->b_bucket_row, &c->b_res, &c->b_sort_tmp,
                      &c->b_tasks, &c->b_tasks_sorted};
    for (DevBuf* b : bufs) b->release();
}

// COUNT over the whole shard: alignments per pair (the reference's loop nest, metrics.py:204-224) into b_na, and the
// totals (alignments, cells, residue bytes); one read-back.
static int stage_count(pc_ctx* c, int condensed, hipStream_t st, uint64_t tot[3]) {
    PcRange range("pc:count");
    int rc = PC_OK;
    const PcDev& d = c->dev;
    const int64_t Lp = c->shard_pairs;
    if (d.G > 0 && c->min_gene_len == 0) {
        pc_set_error("fill: an empty translation cannot be aligned (aai/peq); the reference fails on it too"); return PC_ERR_DATA;
    }
    if ((rc = c->b_na.ensure((Lp + 1) * 4)) || (rc = c->b_off.ensure((Lp + 1) * 4)) || (rc = c->b_totals.ensure(64))) return rc;
    PC_HIP(hipMemsetAsync(c->b_na.p, 0, (Lp + 1) * 4, st));
    PC_HIP(hipMemsetAsync(c->b_totals.p, 0, 64, st));
    PcWalkArgs a; memset(&a, 0, sizeof(a));
    a.na = c->b_na.as<uint32_t>(); a.totals = c->b_totals.as<unsigned long long>();
    a.as_distance = 0; a.condensed = condensed;
    if ((rc = pc_launch_walk(PCW_COUNT, d, c->shard, a, st))) return rc;
    uint64_t* h_tot = (uint64_t*)(c->h_plan + 1000);
    PC_HIP(hipMemcpyAsync(h_tot, c->b_totals.p, 24, hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));                                     // first read-back: the batch size
    tot[0] = h_tot[0]; tot[1] = h_tot[1]; tot[2] = h_tot[2];
    return PC_OK;
}

// alignments behind each owned target genome (a second COUNT walk, only when a fill has to be cut into chunks)
static int count_per_target(pc_ctx* c, int condensed, hipStream_t st, std::vector<uint64_t>& per_owned) {
    int rc = PC_OK;
    const int N = c->dev.N;
    if ((rc = c->b_aln_t.ensure((size_t)N * 8))) return rc;
    PC_HIP(hipMemsetAsync(c->b_aln_t.p, 0, (size_t)N * 8, st));
    PcWalkArgs a; memset(&a, 0, sizeof(a));
    a.totals = c->b_totals.as<unsigned long long>() + 5;                  // (slots 5..7: scratch, nobody reads them)
    a.aln_t = c->b_aln_t.as<unsigned long long>(); a.condensed = condensed;
    if ((rc = pc_launch_walk(PCW_COUNT, c->dev, c->shard, a, st))) return rc;
    std::vector<uint64_t> all(N);
    PC_HIP(hipMemcpyAsync(all.data(), c->b_aln_t.p, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    PC_HIP(hipStreamSynchronize(st));
    per_owned.resize(c->h_owned.size());
    for (size_t k = 0; k < c->h_owned.size(); ++k) per_owned[k] = all[c->h_owned[k]];
    return PC_OK;
}

// PLAN of the owned targets [k0, k1) holding A alignments (b_na is filled): scan, ENUM, sort, distinct alignments, tasks
// sorted by launch class.  Leaves its results in the context's work buffers and c->plan; two small read-backs.
static int stage_plan(pc_ctx* c, int ppos, int condensed, hipStream_t st, int k0, int k1, uint64_t A) {
    int rc = PC_OK;
    PcRange range("pc:plan");
    PlanningScope planning;
    const PcDev& d = c->dev;
    pc_ctx::PlanState& P = c->plan;
    P.valid = false; P.ppos = ppos; P.condensed = condensed; P.A = (int64_t)A; P.n_distinct = 0; P.ntasks = 0; P.tb.assign(c->nlc + 1, 0);
    P.k0 = k0; P.k1 = k1; P.whole = c->world == 1 && condensed == 1 && k0 == 0 && k1 == c->shard.nown;
    memset(&P.st, 0, sizeof(P.st));
    pc_stats& local = P.st;
    const int64_t base = c->h_lbase[k0], Lc = c->h_lbase[k1] - base;
    local.n_pairs = Lc;
    local.n_alignments = (int64_t)A;
    if (A > (uint64_t)PC_PLAN_MAX_ALIGNMENTS) {
        pc_set_error("plan: %llu alignments behind ONE target genome exceed the 2^31-2 a plan can index", (unsigned long long)A); return PC_ERR_LIMIT;
    }
    const int U = d.U;
    const int ncls = c->nlc;
    const int64_t tmp_fixed = std::max<int64_t>(Lc + 1, U + 1);
    if ((rc = c->b_start_q.ensure((U + 1) * 4)) || (rc = c->b_end_q.ensure((U + 1) * 4)) || (rc = c->b_ntask_q.ensure((U + 1) * 4)) ||
        (rc = c->b_task_off_q.ensure((U + 1) * 4)) || (rc = c->b_scan_tmp.ensure(pc_scan_tmp_elems(tmp_fixed) * 4)) || (rc = c->b_plan.ensure(4096)))
        return rc;
    // alignment slot of a pair = exclusive scan of the chunk's counts (slots start at 0 in every chunk)
    if (Lc > 0 && (rc = pc_scan_exclusive_u32(c->b_na.as<uint32_t>() + base, c->b_off.as<uint32_t>() + base, Lc, c->b_scan_tmp.as<uint32_t>(),
                                              (int64_t)(c->b_scan_tmp.cap / 4), st))) return rc;
    PcShard sub = c->shard;
    sub.nown = k1 - k0; sub.owned = c->shard.owned + k0; sub.lbase = c->shard.lbase + k0; sub.ident = c->shard.ident && k0 == 0;
    PcWalkArgs a; memset(&a, 0, sizeof(a));
    a.as_distance = 0; a.condensed = condensed;
    a.off = c->b_off.as<uint32_t>();
    uint64_t* h_tot = (uint64_t*)(c->h_plan + 1000);
    if (A > 0) {
        const int64_t An = (int64_t)A;
        const int key_bits = 2 * d.ubits;
        const size_t sort_bytes = pc_sort_temp_bytes(An, key_bits);
        if ((rc = c->b_key0.ensure(A * 8)) || (rc = c->b_key1.ensure(A * 8)) || (rc = c->b_val0.ensure(A * 4)) || (rc = c->b_val1.ensure(A * 4)) ||
            (rc = c->b_sort_tmp.ensure(std::max<size_t>(sort_bytes, 16))) || (rc = c->b_flags.ensure((A + 1) * 4)) || (rc = c->b_excl.ensure((A + 1) * 4)) ||
            (rc = c->b_alias.ensure(A * 4)) || (rc = c->b_bucket_row.ensure(A * 4)) || (rc = c->b_res.ensure(A * 8)) ||
            (rc = c->b_scan_tmp.ensure(pc_scan_tmp_elems(std::max<int64_t>(tmp_fixed, An + 1)) * 4)))
            return rc;
        const int64_t tmp_elems = (int64_t)(c->b_scan_tmp.cap / 4);
        // 2 ENUM: one sort key per alignment slot; 3 sort; 4 distinct alignments, aliases, buckets (pc_plan.hip)
        a.key = c->b_key0.as<unsigned long long>(); a.val = c->b_val0.as<uint32_t>();
        if ((rc = pc_launch_walk(PCW_ENUM, d, sub, a, st))) return rc;
        if ((rc = pc_sort_pairs(c->b_sort_tmp.p, c->b_sort_tmp.cap, c->b_key0.as<unsigned long long>(), c->b_key1.as<unsigned long long>(),
                                c->b_val0.as<uint32_t>(), c->b_val1.as<uint32_t>(), An, key_bits, st))) return rc;
        if ((rc = pc_launch_mark_heads(c->b_key1.as<unsigned long long>(), c->b_flags.as<uint32_t>(), An, st))) return rc;
        if ((rc = pc_scan_exclusive_u32(c->b_flags.as<uint32_t>(), c->b_excl.as<uint32_t>(), An + 1, c->b_scan_tmp.as<uint32_t>(), tmp_elems, st))) return rc;
        PC_HIP(hipMemsetAsync(c->b_start_q.p, 0, (U + 1) * 4, st));
        PC_HIP(hipMemsetAsync(c->b_end_q.p, 0, (U + 1) * 4, st));
        PC_HIP(hipMemsetAsync(c->b_totals.as<unsigned long long>() + 3, 0, 16, st));        // distinct alignments / cells of THIS chunk
        if ((rc = pc_launch_unique(d, c->b_key1.as<unsigned long long>(), c->b_val1.as<uint32_t>(), c->b_flags.as<uint32_t>(), c->b_excl.as<uint32_t>(),
                                   c->b_alias.as<uint32_t>(), c->b_bucket_row.as<int32_t>(), c->b_start_q.as<uint32_t>(), c->b_end_q.as<uint32_t>(),
                                   c->b_totals.as<unsigned long long>(), An, st))) return rc;
        // 5 workgroup tasks per column sequence (a bucket's left-over rows may go to a narrower variant); second
        //   read-back: number of tasks, distinct totals
        if ((rc = pc_launch_task_count(c->b_start_q.as<uint32_t>(), c->b_end_q.as<uint32_t>(), c->task_plan, c->b_ntask_q.as<uint32_t>(), U, st))) return rc;
        if ((rc = pc_scan_exclusive_u32(c->b_ntask_q.as<uint32_t>(), c->b_task_off_q.as<uint32_t>(), U + 1, c->b_scan_tmp.as<uint32_t>(), tmp_elems, st))) return rc;
        PC_HIP(hipMemcpyAsync(c->h_plan, c->b_task_off_q.as<uint32_t>() + U, 4, hipMemcpyDeviceToHost, st));
        PC_HIP(hipMemcpyAsync(h_tot, c->b_totals.p, 40, hipMemcpyDeviceToHost, st));
        PC_HIP(hipStreamSynchronize(st));
        const uint32_t ntasks = c->h_plan[0];
        local.n_tasks = ntasks; local.n_distinct_alignments = (int64_t)h_tot[3]; local.n_distinct_cells = (int64_t)h_tot[4];
        P.ntasks = ntasks; P.n_distinct = (int64_t)h_tot[3];
        int cbits = 1; while ((1 << cbits) < ncls) ++cbits;
        const size_t tb_bytes = pc_sort_temp_bytes((int64_t)std::max<uint32_t>(ntasks, 1), 32 + cbits);
        if ((rc = c->b_tasks.ensure(std::max<uint32_t>(ntasks, 1) * sizeof(PcTask))) || (rc = c->b_tasks_sorted.ensure(std::max<uint32_t>(ntasks, 1) * sizeof(PcTask))) ||
            (rc = c->b_key0.ensure((size_t)ntasks * 8)) || (rc = c->b_key1.ensure((size_t)ntasks * 8)) || (rc = c->b_val0.ensure((size_t)ntasks * 4)) ||
            (rc = c->b_val1.ensure((size_t)ntasks * 4)) || (rc = c->b_sort_tmp.ensure(std::max<size_t>(tb_bytes, 16))))
            return rc;
        if ((rc = pc_launch_task_fill(d, c->b_start_q.as<uint32_t>(), c->b_end_q.as<uint32_t>(), c->task_plan,
                                      c->b_task_off_q.as<uint32_t>(), c->b_tasks.as<PcTask>(), U, st))) return rc;
        // 6 the task list sorted by (launch class, longest first) with the same radix sort (the key/value buffers of the
        //   alignment sort are free again); third read-back: task range and longest column per launch class
        if ((rc = pc_launch_task_keys(d, c->b_tasks.as<PcTask>(), c->b_key0.as<unsigned long long>(),
                                      c->b_val0.as<uint32_t>(), (int)ntasks, st))) return rc;
        if ((rc = pc_sort_pairs(c->b_sort_tmp.p, c->b_sort_tmp.cap, c->b_key0.as<unsigned long long>(), c->b_key1.as<unsigned long long>(),
                                c->b_val0.as<uint32_t>(), c->b_val1.as<uint32_t>(), (int64_t)ntasks, 32 + cbits, st))) return rc;
        if ((rc = pc_launch_task_gather(c->b_tasks.as<PcTask>(), c->b_val1.as<uint32_t>(), c->b_tasks_sorted.as<PcTask>(), (int)ntasks, st))) return rc;
        if ((rc = pc_launch_class_bounds(c->b_key1.as<unsigned long long>(), (int)ntasks, ncls, c->b_cls_begin.as<uint32_t>(), st))) return rc;
        PC_HIP(hipMemcpyAsync(c->h_plan, c->b_cls_begin.p, (size_t)(ncls + 1) * 4, hipMemcpyDeviceToHost, st));
        PC_HIP(hipStreamSynchronize(st));
        P.tb.assign(c->h_plan, c->h_plan + ncls + 1);
    }
    P.valid = true;
    return PC_OK;
}

// ALIGN: the K4 launches over the planned tasks -- all of them, or every world-th task of each launch class starting at
// slice_rank (tasks of a class are sorted longest first, so the slices of a class carry equal work); results go to
// res[position of the distinct alignment], entries of tasks outside the slice are left zero.
static int stage_align(pc_ctx* c, int slice_rank, int slice_world, uint2* res, hipStream_t st, pc_stats* stats) {
    PcRange range("pc:align");
    pc_ctx::PlanState& P = c->plan;
    if (!P.valid) { pc_set_error("align: no plan (pc_plan_dev first)"); return PC_ERR_STATE; }
    if (P.A <= 0 || P.ntasks == 0) return PC_OK;
    int rc = PC_OK;
    const int ncls = c->nlc;
    const PcTask* task_list = c->b_tasks_sorted.as<PcTask>();
    std::vector<uint32_t> tb = P.tb;
    if (slice_world > 1) {
        std::vector<uint32_t> sb(ncls + 1, 0);
        for (int i = 0; i < ncls; ++i) {
            const uint32_t n = P.tb[i + 1] - P.tb[i];
            sb[i + 1] = sb[i] + (n > (uint32_t)slice_rank ? (n - (uint32_t)slice_rank + (uint32_t)slice_world - 1) / (uint32_t)slice_world : 0u);
        }
        if ((rc = upload_vec(c->b_slice_begin, sb))) return rc;
        if ((rc = pc_launch_task_slice(task_list, (int)P.ntasks, c->b_cls_begin.as<uint32_t>(), c->b_slice_begin.as<uint32_t>(), slice_rank, slice_world,
                                       c->b_tasks.as<PcTask>(), st))) return rc;       // (b_tasks: the unsorted list, free again)
        PC_HIP(hipMemsetAsync(res, 0, (size_t)std::max<int64_t>(P.n_distinct, 1) * 8, st));
        task_list = c->b_tasks.as<PcTask>(); tb = sb;
    }
    return run_align_classes(c, task_list, tb.data(), c->cls_max_lb.data(), res, st, stats, P.ppos);
}

// REDUCE: best match per anchor gene through the aliases, fp64 epilogue (metrics.py:204-232, 247-253), over the plan's targets
static int stage_reduce(pc_ctx* c, int metric, int as_distance, const uint2* res, double* out, hipStream_t st) {
    PcRange range("pc:reduce");
    pc_ctx::PlanState& P = c->plan;
    if (!P.valid) { pc_set_error("reduce: no plan (pc_plan_dev first)"); return PC_ERR_STATE; }
    PcShard sub = c->shard;
    sub.nown = P.k1 - P.k0; sub.owned = c->shard.owned + P.k0; sub.lbase = c->shard.lbase + P.k0; sub.ident = c->shard.ident && P.k0 == 0;
    PcWalkArgs a; memset(&a, 0, sizeof(a));
    a.off = c->b_off.as<uint32_t>(); a.alias = c->b_alias.as<uint32_t>(); a.res = res; a.out = out;
    a.as_distance = as_distance ? 1 : 0; a.condensed = P.condensed;
    return pc_launch_walk(metric == PC_AAI ? PCW_AAI : PCW_PEQ, c->dev, sub, a, st);
}

static void add_plan_stats(pc_stats& acc, const pc_stats& ps) {
    acc.n_tasks += ps.n_tasks; acc.n_distinct_alignments += ps.n_distinct_alignments; acc.n_distinct_cells += ps.n_distinct_cells;
}

// aai / peq: COUNT once, then plan -> align -> reduce -- in one piece when the plan fits the budget, else chunk by chunk
static int fill_aligned(pc_ctx* c, int metric, int ppos, int as_distance, double* out, int condensed, hipStream_t st, pc_stats& local, bool timed) {
    int rc = PC_OK;
    if (!c->residues_ready) { pc_set_error("fill: aai / peq need the residues on the device (pc_upload, or pc_upload_residues after pc_upload_sets)"); return PC_ERR_STATE; }
    uint64_t tot[3] = {0, 0, 0};
    if ((rc = stage_count(c, condensed, st, tot))) return rc;
    local.n_alignments = (int64_t)tot[0]; local.n_cells = (int64_t)tot[1]; local.n_residue_bytes = (int64_t)tot[2];
    const int nown = c->shard.nown;
    const uint64_t A = tot[0];
    // does the whole plan fit?  (hipMemGetInfo only when the question is open: plans under 1 GiB always do)
    uint64_t max_aln = (uint64_t)PC_PLAN_MAX_ALIGNMENTS;
    if (c->plan_budget > 0 || A * PC_PLAN_BYTES_PER_ALIGNMENT > ((uint64_t)1 << 30))
        max_aln = std::min<uint64_t>(max_aln, (uint64_t)std::max<int64_t>(plan_budget_bytes(c) / PC_PLAN_BYTES_PER_ALIGNMENT, 1));
    local.n_chunks = 0;
    if (A <= max_aln) {
        rc = stage_plan(c, ppos, condensed, st, 0, nown, A);
        if (rc == PC_OK) {
            PC_HIP(hipEventRecord(c->ev[1], st));
            rc = stage_align(c, 0, 1, c->b_res.as<uint2>(), st, &local);
        }
        if (rc == PC_OK) {
            PC_HIP(hipEventRecord(c->ev[2], st));
            rc = stage_reduce(c, metric, as_distance, c->b_res.as<uint2>(), out, st);
        }
        if (rc == PC_OK) {
            add_plan_stats(local, c->plan.st);
            local.n_chunks = 1;
            PC_HIP(hipEventRecord(c->ev[3], st));
            return PC_OK;
        }
        if (rc != PC_ERR_NOMEM_INTERNAL) return rc;
        PC_HIP(hipStreamSynchronize(st));                                 // out of HBM: free the plan (and the strip-mined launches' slab), go on in chunks of half the size
        release_plan_buffers(c);
        c->b_scratch.release();
        max_aln = std::max<uint64_t>(A / 2, 1);
        local.n_tasks = 0; local.n_distinct_alignments = local.n_distinct_cells = 0; local.n_align_launches = 0;
    }
    // ---- chunked: successive ranges of the owned targets, each planned, aligned and reduced before the next
    std::vector<uint64_t> per_owned;
    if ((rc = count_per_target(c, condensed, st, per_owned))) return rc;
    float ms_plan = 0.f, ms_align = 0.f, ms_reduce = 0.f;
    int k = 0, nchunks = 0;
    while (k < nown) {
        // the next chunk: as many targets from k on as stay within max_aln (pc_chunk_plan's rule)
        uint64_t run = per_owned[k]; int k1 = k + 1;
        while (k1 < nown && run + per_owned[k1] <= max_aln) { run += per_owned[k1]; ++k1; }
        if (timed) PC_HIP(hipEventRecord(c->ev[4], st));
        rc = stage_plan(c, ppos, condensed, st, k, k1, run);
        if (rc == PC_OK) { if (timed) PC_HIP(hipEventRecord(c->ev[1], st)); rc = stage_align(c, 0, 1, c->b_res.as<uint2>(), st, &local); }
        if (rc == PC_OK) { if (timed) PC_HIP(hipEventRecord(c->ev[2], st)); rc = stage_reduce(c, metric, as_distance, c->b_res.as<uint2>(), out, st); }
        if (rc == PC_ERR_NOMEM_INTERNAL && max_aln > 1 && k1 - k > 1) {                 // a retry with a smaller chunk, not an error
            PC_HIP(hipStreamSynchronize(st));
            release_plan_buffers(c);
            c->b_scratch.release();
            max_aln = std::max<uint64_t>(std::min(max_aln, run) / 2, 1);
            continue;
        }
        if (rc != PC_OK) return rc;
        PC_HIP(hipEventRecord(c->ev[3], st));
        add_plan_stats(local, c->plan.st);
        if (timed) {
            float x = 0.f;
            PC_HIP(hipEventSynchronize(c->ev[3]));
            PC_HIP(hipEventElapsedTime(&x, c->ev[4], c->ev[1])); ms_plan += x;
            PC_HIP(hipEventElapsedTime(&x, c->ev[1], c->ev[2])); ms_align += x;
            PC_HIP(hipEventElapsedTime(&x, c->ev[2], c->ev[3])); ms_reduce += x;
        }
        ++nchunks; k = k1;
    }
    c->plan.valid = false;                                                // the last chunk's plan is not "the plan of the fill"
    local.n_chunks = nchunks;
    local.ms_plan = ms_plan; local.ms_align = ms_align; local.ms_reduce = ms_reduce;
    return PC_OK;
}

#ifndef PC_COL_MIN_N
#define PC_COL_MIN_N 2200        // genomes from which k_sparse_col takes over from the popcount tiles (r05 sweep: profiles/r05/experiments/sparse_col.txt)
#endif
static int fill_impl(pc_ctx* c, int metric, int as_distance, double* out, int condensed, hipStream_t st, pc_stats* stats) {
    if (!c || !c->uploaded) { pc_set_error("fill: upload first"); return PC_ERR_STATE; }
    if (metric < PC_GCS || metric > PC_AAI_PPOS) { pc_set_error("fill: metric %d", metric); return PC_ERR_ARG; }
    const int ppos = metric == PC_AAI_PPOS;
    if (ppos) metric = PC_AAI;
    if (!out) { pc_set_error("fill: out is NULL"); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    static const char* const fill_names[] = {"pc:fill:gcs", "pc:fill:jc", "pc:fill:pocp", "pc:fill:af", "pc:fill:aai", "pc:fill:peq"};
    PcRange range(fill_names[metric]);
    int rc = PC_OK;
    // st == NULL is HIP's legacy default stream, used as such: a caller whose producers / consumers run on it (PyTorch's
    // default stream has handle 0) is ordered with these launches; the library's own streams are non-blocking
    if ((rc = wait_last_work(c, st, true))) return rc;
    const PcDev& d = c->dev;
    const int64_t Lp = c->shard_pairs;
    pc_stats local; memset(&local, 0, sizeof(local));
    local.n_pairs = Lp;
    as_distance = as_distance ? 1 : 0;
    PC_HIP(hipEventRecord(c->ev[0], st));

    // Which kernel fills a set metric (measured crossovers, `profiles/r03/experiments/p_sparse64_record.txt`, `r03_z_pocp_kernel_by_density.txt`;
    // PC_SET_KERNEL = popc | sparse | sparse64 | walker forces one where it exists, for A/B runs and for the tests that keep every
    // one of them honest):
    //   gcs, jc          popcount tiles; a collection of many phams (long bitmap rows, few of them shared): the 64 x 64 sparse tile
    //                    kernel in its counting mode
    //   pocp             popcount tiles + paralog excess; from ~2,500 genomes the 64 x 64 sparse tile kernel where pairs share few
    //                    enough of the phams
    //   af               the 64 x 64 sparse tile kernel (the 32 x 32 one where that kernel's preconditions fail), the column kernel from ~1,400 genomes
    // The popcount tiles cost ~ pairs x bitmap words W, the sparse tiles ~ pairs x (a constant + the phams a pair shares).  Measured on
    // synth(5000, P), P = 300 ... 40,000, in ms: pocp 0.15 + 0.002 W against 0.207 + 0.0085 shared (sparse wins where W > 28 + 4.3 shared:
    // the synthetic collection's 79 words and 2.85 shared phams yes, 300 phams -- 5 words, 34 shared -- three times no); gcs / jc
    // 0.05 + 0.0014 W against 0.155 + 0.0004 W + ~0.005 shared (W > 113 + 5.4 shared: from ~7,500 phams; at 40,000: 0.43 against 0.90).
    // `shared` of an average pair = sum over phams of n_p (n_p - 1) / (N (N - 1)), counted at upload.
    // The 64 x 64 kernel takes "sum == 0" for "no shared pham" and sums in 32 bits: it needs every entry value >= 1 (a
    // genome with an empty translation fails that for af) and genome totals below 2^31; else af falls back to the
    // 32 x 32 kernel / the shared-pham walker (crossover ~3,500 genomes), pocp to the popcount tiles.
    enum { K_POPC, K_SPARSE32, K_SPARSE64, K_WALKER, K_SPARSE_COL };
    int kernel = K_POPC;
    {
        const char* set_force = getenv("PC_SET_KERNEL");                   // (read per fill: the tests switch it between launches)
        const int64_t area = (int64_t)d.N * c->shard.nown;
        const double shared = std::max(c->avg_shared, 0.0);
        const bool counts = metric == PC_GCS || metric == PC_JC;
        const bool s64_ok = counts ? c->max_nph < (1 << 30) : metric == PC_POCP ? c->max_ngen < (1 << 16) /* two gene counts per register */ : (c->min_gene_len >= 1 && c->max_tlen < (int64_t)1 << 31);
        if (counts) kernel = (((double)d.Wb > 113.0 + 5.4 * shared && area >= (int64_t)3000 * 3000) ||
                              ((double)d.Wb > 60.0 + 5.4 * shared && area >= (int64_t)6000 * 6000)) ? K_SPARSE64 : K_POPC;   // (the sparse tiles gain on the popcount tiles as N grows: 5,056 phams, r04 with four workgroups per CU: N = 5,000 0.162 against 0.157 ms, 6,000 0.218 / 0.219, 7,000 0.258 / 0.282, 20,000 1.56 / 2.03)
        else if (metric == PC_POCP) kernel = (s64_ok && (double)d.Wb > 28.0 + 4.3 * shared && area >= (int64_t)2500 * 2500) ? K_SPARSE64 : K_POPC;
        else if (s64_ok) kernel = K_SPARSE64;                                  // (af; r05, ms, 32 x 32 / 64 x 64 tiles: N = 200 0.060 / 0.058, 800 0.095 / 0.061, 1,300 0.082 / 0.069 -- since r04's dense broadcast path the larger tile wins at every size)
        else kernel = area > (int64_t)3500 * 3500 ? K_WALKER : K_SPARSE32;
        // r05: gcs / jc / pocp: the column form of the counting mode (k_sparse_col: the masks over a block of targets stay in LDS for a run
        // of source tiles, no barrier per tile) -- while its masks (pocp: and its paralog list and bit sets) fit 78 KB of LDS.  Against the popcount tiles
        // (profiles/r05/experiments/sparse_col.txt; ms, popcount / column): 5,056 phams (79 words, 2.85 shared) N = 2,000 0.035 / 0.034,
        // 3,000 0.070 / 0.046, 8,000 0.35 / 0.20, 20,000 2.03 / 0.99; 2,500 phams (40 words) N = 5,000 0.098 / 0.110; 1,200: 0.067 / 0.146
        const int sp_mode = counts ? (metric == PC_GCS ? PCW_SPARSE_GCS : PCW_SPARSE_JC) : metric == PC_POCP ? PCW_POCP : PCW_AF;
        bool col_ok = s64_ok && pc_sparse_col_lds(sp_mode, d.sp_W * 64) > 0;
        if (col_ok && !counts) {                                           // ... pocp / af: every block's entries fit its LDS value table, as 16-bit values
            const int cap = pc_sparse_col_vals_cap(d.sp_W * 64);
            col_ok = metric == PC_POCP || c->max_ent_len < 65536;         // (pocp: s64_ok already holds the gene counts below 65,536)
            for (size_t k0 = 0; k0 < c->h_owned.size() && col_ok; k0 += 64) {
                uint32_t n = 0;
                for (size_t k = k0; k < std::min(k0 + 64, c->h_owned.size()); ++k) n += c->h_sp_n[(size_t)c->h_owned[k]];
                col_ok = n <= (uint32_t)cap;
            }
        }
        // (ms, popcount tiles / 64 x 64 sparse tiles / column -- pocp: N = 2,000 0.066 / 0.082 / 0.078, 3,000 0.137 / 0.118 / 0.083, 5,000 0.304 / 0.217 / 0.156,
        // 20,000 3.89 / 2.23 / 1.45; af: 2,000 - / 0.089 / 0.078, 3,000 - / 0.121 / 0.081, 5,000 - / 0.258 / 0.150, 20,000 - / 2.41 / 1.42)
        const int64_t col_min_n = metric == PC_AF ? 1400 : PC_COL_MIN_N;        // (af, 64 x 64 tiles / column: N = 1,000 0.062 / 0.072, 1,300 0.069 / 0.073, 1,500 0.087 / 0.074, 1,800 0.088 / 0.077)
        if (col_ok && (double)d.Wb > 40.0 + 8.0 * shared && area >= col_min_n * col_min_n) kernel = K_SPARSE_COL;
        if (set_force) {
            if (!strcmp(set_force, "sparsecol") && col_ok) kernel = K_SPARSE_COL;
            if (!strcmp(set_force, "popc") && metric != PC_AF) kernel = K_POPC;
            else if (!strcmp(set_force, "sparse") && !counts) kernel = K_SPARSE32;
            else if (!strcmp(set_force, "sparse64") && s64_ok) kernel = K_SPARSE64;
            else if (!strcmp(set_force, "walker") && !counts) kernel = K_WALKER;
        }
    }
    if (metric < PC_AAI) c->last_set_kernel = kernel;
    if (metric < PC_AAI && kernel == K_SPARSE_COL) {
        rc = pc_launch_sparse_col(metric == PC_GCS ? PCW_SPARSE_GCS : metric == PC_JC ? PCW_SPARSE_JC : metric == PC_POCP ? PCW_POCP : PCW_AF, d, c->shard, out, as_distance, condensed, st);
        if (rc != PC_OK) return rc;
        PC_HIP(hipEventRecord(c->ev[3], st));
        local.n_chunks = 1;
    } else if ((metric == PC_GCS || metric == PC_JC) && kernel == K_SPARSE64) {
        rc = pc_launch_sparse64(metric == PC_GCS ? PCW_SPARSE_GCS : PCW_SPARSE_JC, d, c->shard, out, as_distance, condensed, st);
        if (rc != PC_OK) return rc;
        PC_HIP(hipEventRecord(c->ev[3], st));
        local.n_chunks = 1;
    } else if (metric == PC_GCS || metric == PC_JC || (metric == PC_POCP && kernel == K_POPC)) {
        // epilogue table: gcs / jc over (shared, nph_s + nph_t), at most (max_nph+1) x (2 max_nph+1) doubles; pocp over
        // (conserved, ngen_s + ngen_t), (2 max_ngen+1)^2; skipped when huge.  It depends on (metric, as_distance, that maximum)
        // only, so it is rebuilt only when one of them changes.
        const int top = metric == PC_POCP ? c->max_ngen : c->max_nph;
        const int sh_dim = metric == PC_POCP ? 2 * top + 1 : top + 1, tot_dim = 2 * top + 1;
        double* lut = nullptr; bool build_lut = false; int64_t lut_key_now = -1;
        if ((int64_t)sh_dim * tot_dim <= (4 << 20)) {
            if ((rc = c->b_lut.ensure((size_t)sh_dim * tot_dim * 8))) return rc == PC_ERR_NOMEM_INTERNAL ? PC_ERR_HIP : rc;
            lut = c->b_lut.as<double>();
            lut_key_now = ((int64_t)metric << 40) | ((int64_t)as_distance << 32) | (int64_t)top;
            build_lut = lut_key_now != c->lut_key || lut != c->lut_ptr;
        }
        rc = pc_launch_set_popc(d, c->shard, metric, as_distance, out, condensed, lut, build_lut, sh_dim, tot_dim, st);
        if (rc != PC_OK) { c->lut_key = -1; c->lut_ptr = nullptr; return rc; }           // (whatever the table holds now, it is not trusted)
        if (lut) { c->lut_key = lut_key_now; c->lut_ptr = lut; }                          // remembered only once its build was launched
        PC_HIP(hipEventRecord(c->ev[3], st));
        local.n_chunks = 1;
    } else if (metric == PC_POCP || metric == PC_AF) {
        const int mode = metric == PC_POCP ? PCW_POCP : PCW_AF;
        if (kernel == K_WALKER) {
            PcWalkArgs a; memset(&a, 0, sizeof(a));
            a.out = out; a.as_distance = as_distance; a.condensed = condensed;
            rc = pc_launch_walk(mode, d, c->shard, a, st);
        } else if (kernel == K_SPARSE64) rc = pc_launch_sparse64(mode, d, c->shard, out, as_distance, condensed, st);
        else rc = pc_launch_sparse(mode, d, c->shard, out, as_distance, condensed, st);
        if (rc != PC_OK) return rc;
        PC_HIP(hipEventRecord(c->ev[3], st));
        local.n_chunks = 1;
    } else {
        rc = fill_aligned(c, metric, ppos, as_distance, out, condensed, st, local, stats != nullptr);
        if (rc != PC_OK) { (void)mark_work(c, st); return rc == PC_ERR_NOMEM_INTERNAL ? PC_ERR_HIP : rc; }
    }
    if ((rc = mark_work(c, st))) return rc;
    if (stats) {
        PC_HIP(hipEventSynchronize(c->ev[3]));
        c->busy = false;
        PC_HIP(hipEventElapsedTime(&local.ms_total, c->ev[0], c->ev[3]));
        if (metric >= PC_AAI) {
            if (local.n_chunks == 1) {
                PC_HIP(hipEventElapsedTime(&local.ms_plan, c->ev[0], c->ev[1]));
                PC_HIP(hipEventElapsedTime(&local.ms_align, c->ev[1], c->ev[2]));
                PC_HIP(hipEventElapsedTime(&local.ms_reduce, c->ev[2], c->ev[3]));
            }
        } else {
            local.ms_reduce = local.ms_total;
        }
        *stats = local;
    }
    return PC_OK;
}

// ---- alignment-sliced multi-GPU route (aai / peq): every rank plans the whole (unsharded) fill -- milliseconds --, aligns
// every world-th task of each launch class, the per-alignment results are summed to the root (entries of foreign tasks are
// zero), and the root reduces.  Each distinct (row sequence, column sequence) pair is then aligned once in the whole JOB,
// not once per rank, and the ranks carry equal work by construction (no cost-balanced deal, no COUNT pass for it).
extern "C" int pc_plan_dev(pc_ctx* c, int metric, void* stream, pc_stats* stats) {
    if (!c || !c->uploaded) { pc_set_error("pc_plan_dev: upload first"); return PC_ERR_STATE; }
    if (metric != PC_AAI && metric != PC_PEQ && metric != PC_AAI_PPOS) { pc_set_error("pc_plan_dev: metric %d has no alignment plan", metric); return PC_ERR_ARG; }
    if (c->world != 1) { pc_set_error("pc_plan_dev: context is sharded (%d/%d); the alignment-sliced route plans the whole matrix", c->rank, c->world); return PC_ERR_STATE; }
    if (!c->residues_ready) { pc_set_error("pc_plan_dev: the residues are not on the device (pc_upload_residues)"); return PC_ERR_STATE; }
    PC_ON_DEVICE(c);
    hipStream_t st = (hipStream_t)stream;
    int rc = wait_last_work(c, st, true); if (rc != PC_OK) return rc;
    PC_HIP(hipEventRecord(c->ev[0], st));
    uint64_t tot[3] = {0, 0, 0};
    if ((rc = stage_count(c, 1, st, tot))) return abi_rc(rc);
    if (tot[0] > (uint64_t)PC_PLAN_MAX_ALIGNMENTS) {
        pc_set_error("pc_plan_dev: %llu alignments exceed the 2^31-2 one plan can index; the alignment-sliced route keeps the whole plan resident "
                     "-- use the pair-sharded route (pc_fill_shard_dev), which fills in chunks", (unsigned long long)tot[0]);
        return PC_ERR_LIMIT;
    }
    rc = stage_plan(c, metric == PC_AAI_PPOS, 1, st, 0, c->shard.nown, tot[0]);
    (void)mark_work(c, st);
    if (rc != PC_OK) return abi_rc(rc);
    c->plan.st.n_cells = (int64_t)tot[1]; c->plan.st.n_residue_bytes = (int64_t)tot[2];
    PC_HIP(hipEventRecord(c->ev[1], st));
    if ((rc = mark_work(c, st))) return rc;
    if (stats) {
        PC_HIP(hipEventSynchronize(c->ev[1]));
        c->busy = false;
        *stats = c->plan.st;
        stats->n_chunks = 1;
        PC_HIP(hipEventElapsedTime(&stats->ms_plan, c->ev[0], c->ev[1]));
        stats->ms_total = stats->ms_plan;
    }
    return PC_OK;
}

extern "C" int pc_align_slice_dev(pc_ctx* c, int slice_rank, int slice_world, void* res_dev, void* stream, pc_stats* stats) {
    if (!c || !c->uploaded || !c->plan.valid) { pc_set_error("pc_align_slice_dev: pc_plan_dev first"); return PC_ERR_STATE; }
    if (!c->plan.whole) { pc_set_error("pc_align_slice_dev: the plan in the context is not a whole-matrix plan of an unsharded context (pc_plan_dev)"); return PC_ERR_STATE; }
    if (slice_world < 1 || slice_rank < 0 || slice_rank >= slice_world) { pc_set_error("pc_align_slice_dev: slice %d of %d", slice_rank, slice_world); return PC_ERR_ARG; }
    if (!res_dev && c->plan.n_distinct > 0) { pc_set_error("pc_align_slice_dev: res_dev is NULL"); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    hipStream_t st = (hipStream_t)stream;
    // a slice still running on ANOTHER stream reads the task tables this call rewrites
    int rc = wait_last_work(c, st, true); if (rc != PC_OK) return rc;
    pc_stats local = c->plan.st;
    PC_HIP(hipEventRecord(c->ev[1], st));
    rc = stage_align(c, slice_rank, slice_world, (uint2*)res_dev, st, &local);
    PC_HIP(hipEventRecord(c->ev[2], st));
    int rc2 = mark_work(c, st);
    if (rc != PC_OK) return abi_rc(rc);
    if (rc2 != PC_OK) return rc2;
    if (stats) {
        PC_HIP(hipEventSynchronize(c->ev[2]));
        c->busy = false;
        PC_HIP(hipEventElapsedTime(&local.ms_align, c->ev[1], c->ev[2]));
        local.ms_total = local.ms_align;
        local.n_chunks = 1;
        *stats = local;
    }
    return PC_OK;
}

extern "C" int pc_reduce_dev(pc_ctx* c, int metric, int as_distance, const void* res_dev, void* out_condensed_dev, void* stream) {
    if (!c || !c->uploaded || !c->plan.valid) { pc_set_error("pc_reduce_dev: pc_plan_dev first"); return PC_ERR_STATE; }
    if (!c->plan.whole) { pc_set_error("pc_reduce_dev: the plan in the context is not a whole-matrix plan of an unsharded context (pc_plan_dev)"); return PC_ERR_STATE; }
    if (metric == PC_AAI_PPOS) metric = PC_AAI;
    if (metric != PC_AAI && metric != PC_PEQ) { pc_set_error("pc_reduce_dev: metric %d", metric); return PC_ERR_ARG; }
    if (!out_condensed_dev || (!res_dev && c->plan.n_distinct > 0)) { pc_set_error("pc_reduce_dev: NULL argument"); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    hipStream_t st = (hipStream_t)stream;
    int rc = wait_last_work(c, st, true); if (rc != PC_OK) return rc;
    rc = stage_reduce(c, metric, as_distance, (const uint2*)res_dev, (double*)out_condensed_dev, st);
    int rc2 = mark_work(c, st);
    return rc != PC_OK ? abi_rc(rc) : rc2;
}

extern "C" int pc_fill_dev(pc_ctx* c, int metric, int as_distance, void* out_dev, void* stream, pc_stats* stats) {
    if (c && c->uploaded && c->world != 1) { pc_set_error("pc_fill_dev: context is sharded (%d/%d); use pc_fill_shard_dev", c->rank, c->world); return PC_ERR_STATE; }
    return fill_impl(c, metric, as_distance, (double*)out_dev, 1, (hipStream_t)stream, stats);
}

extern "C" int pc_fill(pc_ctx* c, int metric, int as_distance, double* out_condensed, pc_stats* stats) {
    if (!c || !c->uploaded) { pc_set_error("pc_fill: upload first"); return PC_ERR_STATE; }
    if (!out_condensed) { pc_set_error("pc_fill: out is NULL"); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    int rc = PC_OK;
    const int64_t np = (int64_t)c->dev.N * (c->dev.N - 1) / 2;
    if ((rc = c->b_out.ensure(std::max<int64_t>(np, 1) * 8))) return abi_rc(rc);
    if ((rc = pc_fill_dev(c, metric, as_distance, c->b_out.p, c->stream, stats))) return rc;
    if (np) PC_HIP(hipMemcpyAsync(out_condensed, c->b_out.p, np * 8, hipMemcpyDeviceToHost, c->stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    c->busy = false;
    return PC_OK;
}

// Whole matrix into page-locked host memory that the CONTEXT owns: the D2H copy of an N = 20,000 matrix (1.6 GB) runs at
// PCIe speed (~30 ms) instead of through pageable staging (~165 ms), and the 1.6 GB are pinned once, not per call.
// *out_host stays valid until the next fill / upload on this context or its destruction.
extern "C" int pc_fill_borrow(pc_ctx* c, int metric, int as_distance, const double** out_host, pc_stats* stats) {
    if (!c || !c->uploaded) { pc_set_error("pc_fill_borrow: upload first"); return PC_ERR_STATE; }
    if (!out_host) { pc_set_error("pc_fill_borrow: out_host is NULL"); return PC_ERR_ARG; }
    *out_host = nullptr;
    PC_ON_DEVICE(c);
    int rc = PC_OK;
    const int64_t np = (int64_t)c->dev.N * (c->dev.N - 1) / 2;
    const size_t bytes = (size_t)std::max<int64_t>(np, 1) * 8;
    if ((rc = c->b_out.ensure(bytes))) return abi_rc(rc);
    if (bytes > c->h_out_cap) {
        if (c->h_out) { (void)hipHostFree(c->h_out); c->h_out = nullptr; c->h_out_cap = 0; }
        const size_t want = bytes + bytes / 8;
        hipError_t e = hipHostMalloc((void**)&c->h_out, want, hipHostMallocDefault);
        if (e != hipSuccess) { pc_set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); c->h_out = nullptr; return PC_ERR_HIP; }
        c->h_out_cap = want;
    }
    if ((rc = pc_fill_dev(c, metric, as_distance, c->b_out.p, c->stream, stats))) return rc;
    if (np) PC_HIP(hipMemcpyAsync(c->h_out, c->b_out.p, (size_t)np * 8, hipMemcpyDeviceToHost, c->stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    c->busy = false;
    *out_host = c->h_out;
    return PC_OK;
}

extern "C" int pc_fill_shard_dev(pc_ctx* c, int metric, int as_distance, void* shard_dev, void* stream, pc_stats* stats) {
    if (!c || !c->uploaded) { pc_set_error("pc_fill_shard_dev: upload first"); return PC_ERR_STATE; }
    PC_ON_DEVICE(c);
    hipStream_t st = (hipStream_t)stream;
    if (c->shard_stride > c->shard_pairs)
        PC_HIP(hipMemsetAsync((double*)shard_dev + c->shard_pairs, 0, (c->shard_stride - c->shard_pairs) * 8, st));
    return fill_impl(c, metric, as_distance, (double*)shard_dev, 0, st, stats);
}

extern "C" int pc_assemble_dev(pc_ctx* c, const void* gathered_dev, int world, void* out_condensed_dev, void* stream) {
    if (!c || !c->uploaded) { pc_set_error("pc_assemble_dev: upload first"); return PC_ERR_STATE; }
    if (world != c->world) { pc_set_error("pc_assemble_dev: world %d != shard world %d", world, c->world); return PC_ERR_ARG; }
    PC_ON_DEVICE(c);
    if (c->balanced)
        return pc_launch_assemble_table((const double*)gathered_dev, c->shard_stride, c->dev.N, c->b_t_rank.as<int32_t>(), c->b_t_lbase.as<int64_t>(),
                                        (double*)out_condensed_dev, (hipStream_t)stream);
    return pc_launch_assemble((const double*)gathered_dev, world, c->shard_stride, c->dev.N, (double*)out_condensed_dev,
                              (hipStream_t)stream);
}

extern "C" int pc_align_pairs(pc_ctx* c, const int32_t* a_gene, const int32_t* b_gene, int64_t n, int variant,
                              int32_t* n_ident, int32_t* n_diag) {
    if (!c || !c->uploaded) { pc_set_error("pc_align_pairs: upload first"); return PC_ERR_STATE; }
    if (n < 0 || (n > 0 && (!a_gene || !b_gene || !n_ident || !n_diag))) { pc_set_error("pc_align_pairs: NULL argument"); return PC_ERR_ARG; }
    if (n == 0) return PC_OK;
    if (n >= 0x7fffffffLL) { pc_set_error("pc_align_pairs: too many pairs"); return PC_ERR_LIMIT; }
    if (!c->residues_ready) { pc_set_error("pc_align_pairs: the residues are not on the device (pc_upload_residues)"); return PC_ERR_STATE; }
    PC_ON_DEVICE(c);
    int rc = wait_last_work(c, nullptr, false); if (rc != PC_OK) return rc;
    c->plan.valid = false;                             // this call reuses the plan's task, bucket and result buffers
    const int G = c->dev.G;
    const int nvar = pc_nw_num_variants();
    int forced = -2;                                   // -2: automatic
    if (variant < 0) forced = -1;
    else if (variant > 0) {
        for (int v = 0; v < nvar; ++v) if (pc_nw_variant_w(v) == variant) forced = v;
        if (forced == -2) { pc_set_error("pc_align_pairs: no systolic variant with %d columns per lane", variant); return PC_ERR_ARG; }
    }
    std::vector<int> cls(n);
    std::vector<int32_t> sums(n);
    for (int64_t k = 0; k < n; ++k) {
        if (a_gene[k] < 0 || a_gene[k] >= G || b_gene[k] < 0 || b_gene[k] >= G) { pc_set_error("pc_align_pairs: gene index out of range at %lld", (long long)k); return PC_ERR_ARG; }
        const int la = c->h_gene_len[a_gene[k]], lb = c->h_gene_len[b_gene[k]];
        if (la == 0 || lb == 0) { pc_set_error("pc_align_pairs: empty translation at %lld", (long long)k); return PC_ERR_DATA; }
        int v = forced == -2 ? pc_nw_choose_variant(lb) : forced;
        if (v >= 0 && lb > 64 * pc_nw_variant_w(v) && pc_nw_variant_w(v) < 32) { pc_set_error("pc_align_pairs: column gene of %d residues does not fit variant w=%d (strip-mined passes exist for w = 32, 48, 64)", lb, pc_nw_variant_w(v)); return PC_ERR_ARG; }
        cls[k] = pc_class_of(lb, v, c->h_gene_odd[b_gene[k]] != 0);                           // as pc_upload classes such column genes
        sums[k] = la + lb;
    }
    std::vector<int64_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
        if (cls[x] != cls[y]) return cls[x] < cls[y];
        if (b_gene[x] != b_gene[y]) return b_gene[x] < b_gene[y];
        return x < y;
    });
    std::vector<int32_t> rows(n); std::vector<uint32_t> dest(n); std::vector<PcTask> tasks;
    const int nlc = pc_num_classes() * PC_WAVE_MODES;
    std::vector<uint32_t> cls_task_begin(nlc + 1, 0);
    std::vector<int> cls_maxlb(nlc, 0);
    {
        // Buckets (runs of one column gene) are cut the way the fill's planner cuts them (pc_plan.hip): tasks of the class's row
        // count; with the automatic variant also the left-over rows of a wave round to the remainder chooser's variant, and every
        // task in the launch mode its row count asks for.  A forced variant keeps its class's own workgroup shape.
        for (int64_t i = 0; i < n;) {
            const int64_t k = order[i];
            int64_t j = i;
            while (j < n && cls[order[j]] == cls[k] && b_gene[order[j]] == b_gene[k]) ++j;
            const int lb = c->h_gene_len[b_gene[k]], v = pc_class_variant(cls[k]);
            const bool odd = c->h_gene_odd[b_gene[k]] != 0;
            const int per = pc_nw_task_rows(lb, v, pc_class_compare_only(cls[k]));
            int64_t jmain = j; int rem_cls = -1;
            if (forced == -2 && v >= 0) {
                const int W = pc_nw_variant_w(v), G = (lb + W - 1) / W, nseg = std::min(G > 64 ? 1 : 64 / G, 16);
                const int r = nseg > 1 ? (int)((j - i) % nseg) : 0;
                const int vr = r ? pc_nw_choose_remainder(lb, r, v) : -1;
                if (vr >= 0) { jmain = j - r; rem_cls = pc_class_of(lb, vr, odd); }
            }
            auto put = [&](int64_t r0, int64_t r1, int base) {
                PcTask t; t.gene = b_gene[k]; t.begin = (int32_t)r0; t.end = (int32_t)r1;
                const int mode = forced == -2 ? pc_nw_task_mode(lb, (int)(r1 - r0), pc_class_variant(base)) : PC_MODE_CLASS;
                t.pad = base * PC_WAVE_MODES + mode;
                cls_maxlb[t.pad] = std::max(cls_maxlb[t.pad], lb);
                tasks.push_back(t);
            };
            for (int64_t r = i; r < jmain; r += per) put(r, std::min<int64_t>(jmain, r + per), cls[k]);
            if (rem_cls >= 0) put(jmain, j, rem_cls);
            for (int64_t r = i; r < j; ++r) { rows[r] = a_gene[order[r]]; dest[r] = (uint32_t)order[r]; }
            i = j;
        }
        // By launch class, longest tasks first inside a class, as pc_fill's plan orders them.  Workgroups go to the 8 XCDs
        // round-robin by block index, so a list that alternates full tasks and left-overs (every bucket cut the same way) puts
        // all the full ones on half of the XCDs: measured 2x the time on uniform test data.
        std::stable_sort(tasks.begin(), tasks.end(), [&](const PcTask& x, const PcTask& y) {
            if (x.pad != y.pad) return x.pad < y.pad;
            const int64_t wx = (int64_t)(x.end - x.begin) * c->h_gene_len[x.gene], wy = (int64_t)(y.end - y.begin) * c->h_gene_len[y.gene];
            return wx > wy;
        });
        size_t at = 0;
        for (int lc = 0; lc <= nlc; ++lc) { while (at < tasks.size() && tasks[at].pad < lc) ++at; cls_task_begin[lc] = (uint32_t)at; }
    }
    DevBuf d_sums, d_ident, d_diag;
    hipStream_t st = c->stream;
    auto cleanup = [&]() { d_sums.release(); d_ident.release(); d_diag.release(); };
    if ((rc = upload_vec(c->b_bucket_row, rows)) || (rc = upload_vec(c->b_bucket_dest, dest)) || (rc = upload_vec(c->b_tasks, tasks)) ||
        (rc = c->b_res.ensure(n * 8)) || (rc = upload_vec(d_sums, sums)) || (rc = d_ident.ensure(n * 4)) || (rc = d_diag.ensure(n * 4))) { cleanup(); return abi_rc(rc); }
    (void)hipEventRecord(c->ev[1], st);
    for (int lc = 0; lc < nlc; ++lc) {
        const int nt = (int)(cls_task_begin[lc + 1] - cls_task_begin[lc]);
        if (nt <= 0) continue;
        void* scratch = nullptr; size_t sbytes = 0;
        const int base = lc / PC_WAVE_MODES, v = pc_class_variant(base);
        if (v < 0 || pc_launch_is_strip(v, cls_maxlb[lc], lc % PC_WAVE_MODES, 0)) {
            sbytes = v < 0 ? pc_nw_fallback_scratch_bytes(cls_maxlb[lc]) : pc_nw_strip_scratch_bytes(c->max_gene_len, c->n_cu);
            if ((rc = c->b_scratch.ensure(sbytes))) { cleanup(); return abi_rc(rc); }
            scratch = c->b_scratch.p; sbytes = c->b_scratch.cap;
        }
        rc = pc_launch_nw(v, c->dev, c->b_tasks.as<PcTask>() + cls_task_begin[lc], nt, c->b_bucket_row.as<int32_t>(),
                          c->b_bucket_dest.as<uint32_t>(), c->b_res.as<uint2>(), scratch, sbytes, cls_maxlb[lc], 0, c->tie_rule, pc_class_compare_only(base), st,
                          lc % PC_WAVE_MODES, c->max_gene_len);
        if (rc != PC_OK) { cleanup(); return rc; }
    }
    (void)hipEventRecord(c->ev[2], st);
    rc = pc_launch_unpack_res(c->b_res.as<uint2>(), d_sums.as<int32_t>(), d_ident.as<int32_t>(), d_diag.as<int32_t>(), n, st);
    if (rc == PC_OK) {
        hipError_t e = hipMemcpyAsync(n_ident, d_ident.p, n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(n_diag, d_diag.p, n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipEventElapsedTime(&c->last_align_ms, c->ev[1], c->ev[2]);
        if (e != hipSuccess) { pc_set_error("pc_align_pairs: %s", hipGetErrorString(e)); rc = PC_ERR_HIP; }
    }
    cleanup();
    return rc;
}

extern "C" int pc_variant_width(int lb) {
    const int v = pc_nw_choose_variant(lb);
    return v < 0 ? 0 : pc_nw_variant_w(v);
}

extern "C" float pc_last_align_ms(const pc_ctx* c) { return c ? c->last_align_ms : -1.f; }
extern "C" int pc_last_set_kernel(const pc_ctx* c) { return c ? c->last_set_kernel : -1; }

extern "C" int pc_set_tie_rule(pc_ctx* c, int rule) {
    if (!c) { pc_set_error("pc_set_tie_rule: NULL context"); return PC_ERR_ARG; }
    if (rule < 0 || rule >= PC_NUM_TIE_RULES) { pc_set_error("pc_set_tie_rule: rule %d not in 0..%d", rule, PC_NUM_TIE_RULES - 1); return PC_ERR_ARG; }
    c->tie_rule = rule;
    return PC_OK;
}
extern "C" int pc_get_tie_rule(const pc_ctx* c) { return c ? c->tie_rule : -1; }

// test hook for the device round(x, 6)
extern "C" int pc_round6_probe(pc_ctx* c, const double* in, double* out, int64_t n) {
    if (!c || n < 0) { pc_set_error("pc_round6_probe: bad argument"); return PC_ERR_ARG; }
    if (n == 0) return PC_OK;
    PC_ON_DEVICE(c);
    int rc = PC_OK;
    DevBuf a, b;
    if ((rc = a.ensure(n * 8)) || (rc = b.ensure(n * 8))) { a.release(); b.release(); return abi_rc(rc); }
    hipError_t e = hipMemcpyAsync(a.p, in, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { rc = pc_launch_round6_probe(a.as<double>(), b.as<double>(), n, c->stream); }
    if (e == hipSuccess && rc == PC_OK) e = hipMemcpyAsync(out, b.p, n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    a.release(); b.release();
    if (e != hipSuccess) { pc_set_error("pc_round6_probe: %s", hipGetErrorString(e)); return PC_ERR_HIP; }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// One process, several GPUs ("pc_ctx_create(out, device_ids, n_dev) ... the library owns one host thread per
// GPU ... invisible to Python").  The torch.distributed route (one PROCESS per GPU, pc_set_shard* + the caller's RCCL gather) costs
// ~2.5 s before the first pair -- a launcher, an interpreter, a torch import and a process group per rank
// (profiles/r04/final/launch_cost.txt) -- against a 0.6-s fill at N = 5,000.  pc_multi_* does the same static shard with none of
// that: one pc_ctx per device, a host thread per device for the length of a call, every device uploaded in parallel, each filling
// the pairs of the target genomes it is dealt (the cost-balanced deal: deterministic, so every context arrives at the same
// partition by itself), and ONE exchange -- every device copies its shard to the root's gather buffer, device to device (peer
// copies: xGMI between the GPUs of a node) -- before the root permutes the shards into condensed order and delivers them to the
// host.  The reference spreads the same pair list over `cpus` worker processes (matrix.py:471-493).
// ---------------------------------------------------------------------------------------------------------------------
struct pc_multi {
    std::vector<pc_ctx*> ctx;               // ctx[0] is the root: it holds the gather buffer, the assembled matrix and the pinned result
    DevBuf b_gather;                        // root device: world x stride doubles
    std::vector<DevBuf> b_shard;            // device r: its shard, stride doubles (allocated on that device)
    std::vector<int32_t> peer;              // device r -> root: PC_PEER_* (how its shard will travel), see pc_multi_peer_access
    std::vector<std::string> peer_note;     // the runtime's own words where access was refused
    bool uploaded = false, residues = false;
};

namespace {
// fn(rank) on one host thread per device; the first failure (status, message) is re-raised on the calling thread
template <class F> int multi_each(pc_multi* m, F fn) {
    const int n = (int)m->ctx.size();
    std::vector<int> rc(n, PC_OK);
    std::vector<std::string> msg(n);
    auto work = [&](int r) { rc[r] = fn(r); if (rc[r] != PC_OK) msg[r] = pc_last_error(); };
    if (n == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n; ++r) if (rc[r] != PC_OK) { pc_set_error("device %d (rank %d of %d): %s", m->ctx[r]->device, r, n, msg[r].c_str()); return rc[r]; }
    return PC_OK;
}
}  // namespace

extern "C" int pc_multi_create(pc_multi** out, const int* device_ids, int n_dev) {
    if (!out) { pc_set_error("pc_multi_create: out is NULL"); return PC_ERR_ARG; }
    *out = nullptr;
    if (!device_ids || n_dev < 1 || n_dev > 64) { pc_set_error("pc_multi_create: %d devices", n_dev); return PC_ERR_ARG; }
    pc_multi* m = new (std::nothrow) pc_multi();
    if (!m) { pc_set_error("out of host memory"); return PC_ERR_ARG; }
    m->b_shard.resize((size_t)n_dev);
    for (int r = 0; r < n_dev; ++r) {
        pc_ctx* c = nullptr;
        const int rc = pc_ctx_create(&c, device_ids[r]);
        if (rc != PC_OK) { pc_multi_destroy(m); return rc; }
        m->ctx.push_back(c);
    }
    // peer access root <- every other device where the hardware offers it.  Without it the copies still work (the runtime stages
    // them through host memory), only slower: so a refusal is no error, but it is RECORDED per device (pc_multi_peer_access) --
    // a node whose exchange crawls must be able to say why.
    m->peer.assign((size_t)n_dev, PC_PEER_SAME_DEVICE);
    m->peer_note.assign((size_t)n_dev, std::string());
    for (int r = 1; r < n_dev; ++r) {
        if (m->ctx[r]->device == m->ctx[0]->device) continue;
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, m->ctx[r]->device, m->ctx[0]->device);
        if (e != hipSuccess) {
            m->peer[r] = PC_PEER_FAILED;
            m->peer_note[r] = std::string("hipDeviceCanAccessPeer: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            continue;
        }
        if (!can) { m->peer[r] = PC_PEER_UNAVAILABLE; m->peer_note[r] = "hipDeviceCanAccessPeer says no: copies staged by the runtime"; continue; }
        PcDeviceGuard guard(m->ctx[r]->device);
        e = hipDeviceEnablePeerAccess(m->ctx[0]->device, 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) m->peer[r] = PC_PEER_ENABLED;
        else { m->peer[r] = PC_PEER_FAILED; m->peer_note[r] = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); }
        (void)hipGetLastError();
    }
    *out = m;
    return PC_OK;
}

extern "C" void pc_multi_destroy(pc_multi* m) {
    if (!m) return;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        if (!m->ctx[r]) continue;
        { PcDeviceGuard guard(m->ctx[r]->device); if (r < m->b_shard.size()) m->b_shard[r].release(); if (r == 0) m->b_gather.release(); }
        pc_ctx_destroy(m->ctx[r]);
    }
    delete m;
}

extern "C" int pc_multi_devices(const pc_multi* m) { return m ? (int)m->ctx.size() : -1; }

// How each device's shard reaches the root (decided once, in pc_multi_create): granted[r] = PC_PEER_*.  Returns the number of
// devices whose copies will NOT go device to device (PC_PEER_UNAVAILABLE / PC_PEER_FAILED); their reasons, one per line, are left
// for pc_last_error() -- as a note, not a failure: the call's status is that count (>= 0) or PC_ERR_ARG.
extern "C" int pc_multi_peer_access(const pc_multi* m, int32_t* granted) {
    if (!m) { pc_set_error("pc_multi_peer_access: NULL"); return PC_ERR_ARG; }
    int staged = 0;
    std::string note;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        if (granted) granted[r] = m->peer[r];
        if (m->peer[r] == PC_PEER_UNAVAILABLE || m->peer[r] == PC_PEER_FAILED) {
            ++staged;
            char line[256];
            snprintf(line, sizeof line, "device %d -> root %d: %s\n", m->ctx[r]->device, m->ctx[0]->device, m->peer_note[r].c_str());
            note += line;
        }
    }
    if (staged) pc_set_error("%s", note.c_str());
    return staged;
}

// Every device gets the same packed genomes (they are replicated: 0.2 GB at N = 5,000), in parallel.  with_residues = 0: part 1 only
// (all the set metrics need); an aai / peq fill uploads the residues on demand.
extern "C" int pc_multi_upload(pc_multi* m, const pc_packed* g, int with_residues) {
    if (!m || !g) { pc_set_error("pc_multi_upload: NULL argument"); return PC_ERR_ARG; }
    m->uploaded = false; m->residues = false;
    int rc = multi_each(m, [&](int r) { return with_residues ? pc_upload(m->ctx[r], g) : pc_upload_sets(m->ctx[r], g); });
    if (rc != PC_OK) return rc;
    m->uploaded = true; m->residues = with_residues != 0;
    return PC_OK;
}
extern "C" int pc_multi_upload_residues(pc_multi* m, const pc_packed* g) {
    if (!m || !g) { pc_set_error("pc_multi_upload_residues: NULL argument"); return PC_ERR_ARG; }
    if (!m->uploaded) { pc_set_error("pc_multi_upload_residues: pc_multi_upload first"); return PC_ERR_STATE; }
    int rc = multi_each(m, [&](int r) { return pc_upload_residues(m->ctx[r], g); });
    if (rc == PC_OK) m->residues = true;
    return rc;
}
extern "C" int pc_multi_set_tie_rule(pc_multi* m, int rule) {
    if (!m) { pc_set_error("pc_multi_set_tie_rule: NULL"); return PC_ERR_ARG; }
    for (pc_ctx* c : m->ctx) { const int rc = pc_set_tie_rule(c, rule); if (rc != PC_OK) return rc; }
    return PC_OK;
}

// The whole matrix over the devices of `m`: *out_host points to f64[N(N-1)/2] in page-locked memory the ROOT context owns (valid until
// the next fill or upload).  stats (optional, an array of pc_multi_devices() entries): every device's own fill.  exchange_ms /
// assemble_ms (optional): HIP-event times of the slowest device's copy to the root and of the root's permutation.
extern "C" int pc_multi_fill_borrow(pc_multi* m, int metric, int as_distance, const double** out_host, pc_stats* stats, float* exchange_ms, float* assemble_ms) {
    if (!m || !out_host) { pc_set_error("pc_multi_fill_borrow: NULL argument"); return PC_ERR_ARG; }
    *out_host = nullptr;
    if (!m->uploaded) { pc_set_error("pc_multi_fill_borrow: pc_multi_upload first"); return PC_ERR_STATE; }
    const int world = (int)m->ctx.size();
    pc_ctx* root = m->ctx[0];
    if (world == 1) {
        if (exchange_ms) *exchange_ms = 0.f;
        if (assemble_ms) *assemble_ms = 0.f;
        return pc_fill_borrow(root, metric, as_distance, out_host, stats);
    }
    const int N = root->dev.N;
    const int64_t np = (int64_t)N * (N - 1) / 2;
    // 1 the deal (every context computes the same one), shard buffers
    int rc = multi_each(m, [&](int r) -> int {
        pc_ctx* c = m->ctx[r];
        int e = pc_set_shard_balanced(c, r, world);
        if (e != PC_OK) return e;
        PcDeviceGuard guard(c->device);
        return abi_rc(m->b_shard[r].ensure((size_t)std::max<int64_t>(c->shard_stride, 1) * 8));
    });
    if (rc != PC_OK) return rc;
    const int64_t stride = root->shard_stride;
    for (pc_ctx* c : m->ctx) if (c->shard_stride != stride) { pc_set_error("pc_multi_fill_borrow: the devices disagree on the deal (stride %lld vs %lld)", (long long)c->shard_stride, (long long)stride); return PC_ERR_STATE; }
    {
        PcDeviceGuard guard(root->device);
        if ((rc = abi_rc(m->b_gather.ensure((size_t)std::max<int64_t>(stride, 1) * 8 * (size_t)world))) ||
            (rc = abi_rc(root->b_out.ensure((size_t)std::max<int64_t>(np, 1) * 8)))) return rc;
        const size_t bytes = (size_t)std::max<int64_t>(np, 1) * 8;
        if (bytes > root->h_out_cap) {
            if (root->h_out) { (void)hipHostFree(root->h_out); root->h_out = nullptr; root->h_out_cap = 0; }
            const size_t want = bytes + bytes / 8;
            hipError_t e = hipHostMalloc((void**)&root->h_out, want, hipHostMallocDefault);
            if (e != hipSuccess) { pc_set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); root->h_out = nullptr; return PC_ERR_HIP; }
            root->h_out_cap = want;
        }
    }
    // 2 every device fills its shard, then copies it to its slice of the root's gather buffer: the one exchange
    std::vector<float> xms((size_t)world, 0.f);
    std::vector<pc_stats> local((size_t)world);
    rc = multi_each(m, [&](int r) -> int {
        pc_ctx* c = m->ctx[r];
        int e = pc_fill_shard_dev(c, metric, as_distance, m->b_shard[r].p, c->stream, &local[r]);       // (with stats: returns when the fill is done)
        if (e != PC_OK) return e;
        PcDeviceGuard guard(c->device);
        PC_HIP(hipEventRecord(c->ev[0], c->stream));
        double* dst = m->b_gather.as<double>() + (size_t)r * (size_t)stride;
        if (c->device == root->device) PC_HIP(hipMemcpyAsync(dst, m->b_shard[r].p, (size_t)stride * 8, hipMemcpyDeviceToDevice, c->stream));
        else PC_HIP(hipMemcpyPeerAsync(dst, root->device, m->b_shard[r].p, c->device, (size_t)stride * 8, c->stream));
        PC_HIP(hipEventRecord(c->ev[1], c->stream));
        PC_HIP(hipStreamSynchronize(c->stream));
        PC_HIP(hipEventElapsedTime(&xms[r], c->ev[0], c->ev[1]));
        return (int)PC_OK;
    });
    if (rc != PC_OK) return rc;
    // 3 the root permutes the shards into condensed order and delivers
    {
        PcDeviceGuard guard(root->device);
        PC_HIP(hipEventRecord(root->ev[0], root->stream));
        if ((rc = pc_assemble_dev(root, m->b_gather.p, world, root->b_out.p, root->stream))) return rc;
        PC_HIP(hipEventRecord(root->ev[1], root->stream));
        if (np) PC_HIP(hipMemcpyAsync(root->h_out, root->b_out.p, (size_t)np * 8, hipMemcpyDeviceToHost, root->stream));
        PC_HIP(hipStreamSynchronize(root->stream));
        if (assemble_ms) PC_HIP(hipEventElapsedTime(assemble_ms, root->ev[0], root->ev[1]));
    }
    if (exchange_ms) *exchange_ms = *std::max_element(xms.begin(), xms.end());
    if (stats) for (int r = 0; r < world; ++r) stats[r] = local[r];
    *out_host = root->h_out;
    return PC_OK;
}
