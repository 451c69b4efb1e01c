// pg_shim.cpp — C-ABI of include/pangenie_hmm.h on top of the HIP kernels (pg_kernels.hip).
//
// Host side of the boundary: argument checking, one device arena per job, H2D/D2H,
// launch order, hipEvent timing per kernel class.  No compute happens here except the
// ProbabilityTable, which the reference also builds on the host in long double before any
// HMM runs (reference src/commands.cpp:846, src/probabilitytable.cpp:28-45).
// There is NO CPU fallback: without a HIP device every entry point returns PG_ERR_DEVICE.
//
// A job = a set of CHAINS (one per (contig, path subset) — or per (sample, contig) in a cohort
// job) over a set of INDEX contigs.  Index arrays (positions, alleles, k-mer masks, path ->
// allele) are uploaded once per index contig; chains of a cohort job share them and own only
// their read counts, coverage, intermediates and results (reference src/commands.cpp:118-138:
// the index is sample-independent, update_readcount / set_coverage are the per-sample part).
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pangenie_hmm.h"
#include "pg_device.h"

extern "C" {
void pgk_launch_prep(const DevContig*, uint32_t, uint32_t, uint32_t, uint32_t, DevTable, hipStream_t);
void pgk_launch_compact(const DevContig*, uint32_t, hipStream_t);
void pgk_launch_index(const DevContig*, uint32_t, uint32_t, uint32_t, int, hipStream_t);
void pgk_launch_prep_split(const DevContig*, uint32_t, uint32_t, uint32_t, uint32_t, DevTable, hipStream_t);
void pgk_launch_records(const DevContig*, uint32_t, uint32_t, hipStream_t);
void pgk_launch_bins(const DevContig*, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);
void pgk_launch_sweep(const DevContig*, uint32_t, uint32_t, int, hipStream_t);
void pgk_launch_sweep_chunk(const DevContig*, uint32_t, uint32_t, uint32_t, hipStream_t);
void pgk_launch_post(const DevContig*, uint32_t, uint32_t, uint32_t, hipStream_t);
uint32_t pgk_post_blocks(uint32_t, uint32_t, uint32_t*);
void pgk_launch_phase2_persistent(const DevContig*, uint32_t, uint32_t, hipStream_t, hipStream_t);
void pgk_launch_stream_handshake(uint32_t*, hipStream_t, hipStream_t);
void pgk_launch_sweep_small(const DevContig*, const uint32_t*, uint32_t, int, uint32_t, double*, hipStream_t);
void pgk_launch_sweep_smallx(const DevContig*, const uint32_t*, uint32_t, int, uint32_t, double*, hipStream_t);
void pgk_launch_emission_single(const DevContig*, DevTable, uint32_t, double*, int*, hipStream_t);
void pgk_launch_transition_single(double, uint32_t, int, double*, hipStream_t);
uint32_t pgk_threads_for_hp(uint32_t);
void pgk_launch_viterbi(const DevContig*, uint32_t, uint32_t, uint32_t, hipStream_t);  // pg_viterbi.hip
}

namespace {

void set_err(char* err, size_t errlen, const char* fmt, ...) {
    if (!err || errlen == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            set_err(err, errlen, "%s failed: %s", #call, hipGetErrorString(e_));          \
            return PG_ERR_DEVICE;                                                         \
        }                                                                                 \
    } while (0)

constexpr int PG_MAX_DEVICES_HOST = 64;
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// ---------------------------------------------------------------------------------------
//  ProbabilityTable (host, long double) — reference src/probabilitytable.cpp
// ---------------------------------------------------------------------------------------
struct pg_table {
    uint16_t cov_min = 0, cov_max = 0, count_max = 0;
    long double reg = 0.0L;
    std::vector<long double> p;  // [count][cov - cov_min][3], as the reference indexes it
    std::mutex mu;
};

namespace {

// reference src/probabilitytable.cpp:7-19
double error_param(double cov) {
    if (cov < 10.0) return 0.99;
    if (cov < 20) return 0.95;
    if (cov < 40) return 0.9;
    return 0.8;
}
// reference src/probabilitytable.cpp:75-81 (log(i): double overload; log(mean), exp: long double)
long double poisson_ld(long double mean, unsigned value) {
    long double sum = 0.0L;
    int v = (int)value;
    for (size_t i = 1; i <= value; ++i) sum += ::log((double)i);
    long double log_val = -mean + v * logl(mean) - sum;
    return expl(log_val);
}
// reference src/probabilitytable.cpp:83-85
long double geometric_ld(long double prob, unsigned value) { return powl(1.0L - prob, (long double)value) * prob; }

// reference src/probabilitytable.cpp:55-65 + src/copynumber.cpp:14-41
void compute_probability(const pg_table* t, uint16_t cov, uint16_t count, long double out[3]) {
    long double c0 = geometric_ld(error_param(cov), count);
    long double c1 = poisson_ld(cov / 2.0, count);
    long double c2 = poisson_ld(cov, count);
    if (t->reg > 0) {
        long double sum = c0 + c1 + c2 + 3.0L * t->reg;
        out[0] = (c0 + t->reg) / sum;
        out[1] = (c1 + t->reg) / sum;
        out[2] = 1.0L - out[0] - out[1];
    } else {
        out[0] = c0; out[1] = c1; out[2] = c2;
    }
}

bool in_table(const pg_table* t, uint16_t cov, uint16_t count) {
    return cov >= t->cov_min && cov < t->cov_max && count < t->count_max;
}

// (mantissa, exponent) view of the dense table, layout [cov][count][3].  Every job gets its OWN device
// copy inside its arena: a later pg_table_modify / pg_table_destroy cannot pull it away from under a
// resident job.
void table_snapshot(pg_table* t, std::vector<double>& m, std::vector<int32_t>& e, std::vector<unsigned char>& packed, DevTable* meta) {
    std::lock_guard<std::mutex> lock(t->mu);
    const uint32_t ncov = t->cov_max > t->cov_min ? t->cov_max - t->cov_min : 0;
    meta->cov_min = t->cov_min; meta->cov_max = t->cov_max; meta->count_max = t->count_max; meta->pad = 0;
    meta->reg = (double)t->reg;
    const size_t n = (size_t)ncov * t->count_max * 3;
    m.assign(n ? n : 1, 0.0);
    e.assign(n ? n : 1, 0);
    for (uint32_t c = 0; c < ncov; ++c)
        for (uint32_t k = 0; k < t->count_max; ++k)
            for (int i = 0; i < 3; ++i) {
                long double v = t->p[((size_t)k * ncov + c) * 3 + i];
                int ex = 0;
                long double mant = (v == 0.0L || v != v || isinf((double)v)) ? v : frexpl(v, &ex);
                m[((size_t)c * t->count_max + k) * 3 + i] = (double)mant;
                e[((size_t)c * t->count_max + k) * 3 + i] = ex;
            }
    // the same entries as 32-byte pieces {m0, m1, m2, int16 e0, e1, e2, 0} (DevTable::packed)
    packed.assign(n ? n / 3 * 32 : 32, 0);
    for (size_t q = 0; q < n / 3; ++q) {
        memcpy(&packed[q * 32], &m[q * 3], 24);
        for (int i = 0; i < 3; ++i) {
            const int32_t ex = e[q * 3 + i];
            const int16_t e16 = (int16_t)(ex < -32768 ? -32768 : (ex > 32767 ? 32767 : ex));   // (|exponent| <= 16445: long double's range)
            memcpy(&packed[q * 32 + 24 + 2 * i], &e16, 2);
        }
    }
}

}  // namespace

extern "C" pg_table* pg_table_create(uint16_t cov_min, uint16_t cov_max, uint16_t count_max, long double regularization) {
    pg_table* t = new pg_table();
    t->cov_min = cov_min; t->cov_max = cov_max; t->count_max = count_max; t->reg = regularization;
    const size_t ncov = cov_max > cov_min ? (size_t)(cov_max - cov_min) : 0;
    t->p.resize((size_t)count_max * ncov * 3);
    for (uint32_t i = 0; i < count_max; ++i)
        for (uint32_t j = 0; j < ncov; ++j)
            compute_probability(t, (uint16_t)(j + cov_min), (uint16_t)i, &t->p[((size_t)i * ncov + j) * 3]);
    return t;
}
extern "C" pg_table* pg_table_create_default(void) { return pg_table_create(0, 0, 0, 0.0L); }

extern "C" int pg_table_modify(pg_table* t, uint16_t cov, uint16_t count, long double p0, long double p1, long double p2) {
    if (!t || !in_table(t, cov, count)) return PG_ERR_INVALID;  // reference throws runtime_error here
    std::lock_guard<std::mutex> lock(t->mu);
    const size_t ncov = (size_t)(t->cov_max - t->cov_min);
    long double* e = &t->p[((size_t)count * ncov + (cov - t->cov_min)) * 3];
    e[0] = p0; e[1] = p1; e[2] = p2;
    return PG_OK;
}
extern "C" int pg_table_get(const pg_table* t, uint16_t cov, uint16_t count, long double out3[3]) {
    if (!t) return PG_ERR_INVALID;
    if (in_table(t, cov, count)) {
        const size_t ncov = (size_t)(t->cov_max - t->cov_min);
        const long double* e = &t->p[((size_t)count * ncov + (cov - t->cov_min)) * 3];
        out3[0] = e[0]; out3[1] = e[1]; out3[2] = e[2];
    } else {
        compute_probability(t, cov, count, out3);
    }
    return PG_OK;
}
extern "C" void pg_table_destroy(pg_table* t) { delete t; }

// ---------------------------------------------------------------------------------------
//  misc
// ---------------------------------------------------------------------------------------
extern "C" int pg_hmm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" const char* pg_hmm_version(void) { return "pangenie-hmm-mi355x 0.2 (gfx950, fp64)"; }

extern "C" int pg_hmm_geno_offsets(const pg_contig_batch* b, uint64_t* geno_off) {
    if (!b || !geno_off) return PG_ERR_INVALID;
    geno_off[0] = 0;
    for (uint32_t v = 0; v < b->n_variants; ++v) {
        const uint64_t A = b->allele_off[v + 1] - b->allele_off[v];
        geno_off[v + 1] = geno_off[v] + A * (A + 1) / 2;
    }
    return PG_OK;
}

// ---------------------------------------------------------------------------------------
//  jobs
// ---------------------------------------------------------------------------------------
static const char* const kKernelNames[PG_N_KERNEL_CLASSES] = {"k_prep", "k_compact", "k_records",
                                                              "k_sweep_phase1", "k_sweep_phase2", "k_bins"};

namespace {

struct IndexHost {   // one index contig
    uint32_t V = 0, H = 0, HP = 0, T = 0, RB = 0, part_slots = 2, pair_n = 1;
    bool lean = false;  // every object biallelic and H = HP = 64: the store-only phases run on k_sweep_lean
    bool cls4 = false;  // HP = 16 / 32, H = HP, every object biallelic, fused job: class sums instead of per-thread partials (DevContig::cls4)
    bool leanx = false; // HP = 128 / 64 and every object has at most PG_AMAX alleles (not `lean`): the store-only phases run on k_sweep_leanx
    bool small = false; // every object biallelic and H = HP = 16: the store-only phases run on k_sweep_small16
    bool smallx = false; // H = HP = 16 with multiallelic objects (the 15 + 1 sampled paths): k_sweep_small16x (pg_small16x.h)
    std::vector<uint32_t> list_b;            // ... and k_prep_bi's own
    size_t o_list_b = 0;
    std::vector<uint32_t> list_m4, list_w;   // prep_fast == 2: the objects of k_prep_m4 (3 .. PG_AMAX alleles, <= 64 k-mers) / of k_prep (neither that nor k_prep_bi's)
    size_t o_list_m4 = 0, o_list_w = 0;
    std::vector<uint32_t> auxidx;    // [V] aux slot offset / 16 of every variant with more than two alleles (smallx), PG_WIDE_NONE otherwise
    uint64_t aux_bytes = 0;
    uint32_t n_wide_cand = 0;        // variants with more than PG_AMAX alleles (each may be a wide column)
    bool widef_cand = false, widef = false;   // 64-path chains on the general kernel with such variants: in a fused job a wide column costs that column (DevContig::widef)
    size_t o_auxidx = 0;
    uint32_t prep_fast = 0;  // 1: every object has exactly two alleles and <= 32 k-mers, H <= 64: k_prep_bi; 2: at least half of them (k_prep the rest)
    // The split path (pg_device.h, pg_split.h): 1 / 2 = the chains over this index contig are small / smallx chains whose
    // per-variant work is split into what the index alone decides (formed once per upload) and what the sample's counts decide
    uint32_t split = 0;
    bool all_sb = false;     // every object is k_prep_s_bi's (two alleles, <= 32 k-mers): no lists
    std::vector<uint32_t> list_big;   // variants with more than 32 alleles: the wave-per-object index kernels' list (the others: one thread each)
    size_t o_list_big = 0;
    // index-level device arrays (every chain over this contig points at them)
    size_t o_kept = 0, o_apres = 0, o_colv = 0, o_colof = 0, o_ixpd = 0, o_ixrec = 0, o_ixbin = 0, o_ixwl = 0, o_ixnw = 0;
    uint32_t sumK = 0, sumA = 0;
    uint64_t n_lik = 0, wide_bytes = 0;
    std::vector<uint16_t> n_kmers;   // [V] K of every variant
    std::vector<uint32_t> widx;      // [V] wide entry offset / 16 (only if wide_bytes)
    std::vector<uint64_t> goff;      // [V+1]
    std::vector<uint64_t> pos;       // [V] host copy of the positions (run_phasing: the Viterbi's transition probabilities)
    std::vector<uint32_t> koff, aoff;  // [V+1] the per-variant layout the arena, goff / widx and the kernel choice were planned for
    // device
    size_t o_pos = 0, o_koff = 0, o_aoff = 0, o_aid = 0, o_aflag = 0, o_akoff = 0, o_akmask = 0, o_pa = 0, o_goff = 0, o_widx = 0;
};

struct ChainHost {
    uint32_t index = 0;
    std::vector<uint16_t> coverage;  // [V] host copy (GenotypingResult::set_coverage)
    size_t o_cov = 0, o_kcnt = 0;
    uint64_t lik_first = 0;          // offset of this chain's bins inside the packed lik / lik_exp regions
    DevContig d;
    uint32_t n_cols_host = 0;
};

// entries of posterior partials per column and slot pair (DevContig::T), the ONE place the arena plan and the chain
// descriptors take it from: threads per chain workgroup, half of them at HP = 32 (lanes l and l + 32 are folded: fold32)
uint32_t part_entries(const IndexHost& x) { return x.HP == 32u ? x.T / 2u : x.T; }

uint32_t pad_paths(uint32_t H) {
    for (uint32_t hp = 16; hp <= PG_MAX_PATHS; hp <<= 1)
        if (H <= hp) return hp;
    return 0;
}

int check_batch(const pg_contig_batch* b, bool need_counts, char* err, size_t errlen) {
    if (!b) { set_err(err, errlen, "null batch"); return PG_ERR_INVALID; }
    const uint32_t V = b->n_variants;
    if (V > 0) {
        if (b->n_paths == 0) {  // reference src/columnindexer.cpp:18-22
            set_err(err, errlen, "HMM::index_columns: column 0 is not covered by any paths.");
            return PG_ERR_NO_PATHS;
        }
        if (!b->variant_pos || !b->kmer_off || !b->allele_off || !b->allele_id ||
            !b->allele_flags || !b->allele_kmer_off || !b->allele_kmer_mask || !b->path_allele || (need_counts && !b->coverage)) {
            set_err(err, errlen, "batch has null arrays");
            return PG_ERR_INVALID;
        }
        if (b->kmer_off[0] != 0 || b->allele_off[0] != 0) { set_err(err, errlen, "offset arrays must start at 0"); return PG_ERR_INVALID; }
        uint32_t maxA = 0;
        for (uint32_t v = 0; v < V; ++v) {
            if (b->kmer_off[v + 1] < b->kmer_off[v]) { set_err(err, errlen, "kmer_off is not monotonic at variant %u", v); return PG_ERR_INVALID; }
            if (b->allele_off[v + 1] <= b->allele_off[v]) {
                set_err(err, errlen, "variant %u has no alleles (allele_off must be strictly increasing)", v);
                return PG_ERR_INVALID;
            }
            const uint32_t A = b->allele_off[v + 1] - b->allele_off[v];
            if (A > maxA) maxA = A;
        }
        if (maxA > PG_MAX_ALLELES_PER_VARIANT) {
            set_err(err, errlen, "a variant has %u alleles; the device path supports at most %d per UniqueKmers object", maxA, PG_MAX_ALLELES_PER_VARIANT);
            return PG_ERR_UNSUPPORTED;
        }
        if (need_counts && b->kmer_off[V] > 0 && !b->kmer_count) { set_err(err, errlen, "kmer_count is null"); return PG_ERR_INVALID; }
    }
    if (b->n_paths > PG_MAX_PATHS) {
        set_err(err, errlen, "device path supports at most %d selected paths per chain (got %u); use path subsets (-a)", PG_MAX_PATHS, b->n_paths);
        return PG_ERR_UNSUPPORTED;
    }
    return PG_OK;
}

// Device arenas of finished one-shot jobs, kept for the next calls (the one-shot call creates and destroys a job per call,
// and a device allocation of tens of GB costs more than the job: the pages are mapped on first touch).  A POOL: the
// reference runs N constructors at a time on thread-pool workers (src/commands.cpp:949-978), so several arenas of
// different sizes are in use at once.  take(): the smallest cached arena of the device that is large enough.
// put(): keeps the arena unless the pool would then hold more than PG_ARENA_POOL_GB (default 200) — the smallest
// entries go first.  pg_hmm_release_cache() empties it; an allocation failure empties it and retries.
struct ArenaPool {
    struct Entry { int device; unsigned char* ptr; size_t bytes; };
    std::mutex mu;
    std::vector<Entry> free_list;
    // per device: PG_ARENA_POOL_GB, else 70 % of that device's memory (hipMemGetInfo; the caller has made `device` current)
    size_t limit(int device) {
        static const double env_gb = [] { const char* e = getenv("PG_ARENA_POOL_GB"); return e ? strtod(e, nullptr) : -1.0; }();
        if (env_gb >= 0.0) return (size_t)(env_gb * 1073741824.0);
        static size_t of_device[64] = {0};
        if (device < 0 || device >= 64) return (size_t)200 << 30;
        if (of_device[device] == 0) {
            size_t free_b = 0, total_b = 0;
            of_device[device] = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) ? (size_t)(0.7 * (double)total_b) : ((size_t)200 << 30);
        }
        return of_device[device];
    }
    bool take(int device, size_t need, unsigned char** ptr, size_t* bytes) {
        std::lock_guard<std::mutex> lock(mu);
        int best = -1;
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].device == device && free_list[i].bytes >= need && (best < 0 || free_list[i].bytes < free_list[(size_t)best].bytes)) best = (int)i;
        if (best < 0) return false;
        *ptr = free_list[(size_t)best].ptr; *bytes = free_list[(size_t)best].bytes;
        free_list.erase(free_list.begin() + best);
        return true;
    }
    void put(int device, unsigned char* ptr, size_t bytes) {  // (the caller has made `device` current)
        std::vector<Entry> drop;
        {
            std::lock_guard<std::mutex> lock(mu);
            free_list.push_back({device, ptr, bytes});
            size_t total = 0;
            for (const Entry& e : free_list) if (e.device == device) total += e.bytes;
            const size_t lim = limit(device);
            while (total > lim) {   // (this device's smallest entries go first)
                long k = -1;
                for (size_t i = 0; i < free_list.size(); ++i)
                    if (free_list[i].device == device && (k < 0 || free_list[i].bytes < free_list[(size_t)k].bytes)) k = (long)i;
                if (k < 0) break;
                total -= free_list[(size_t)k].bytes;
                drop.push_back(free_list[(size_t)k]);
                free_list.erase(free_list.begin() + k);
            }
        }
        for (const Entry& e : drop)
            if (hipSetDevice(e.device) == hipSuccess) hipFree(e.ptr);
        if (!drop.empty()) hipSetDevice(device);
    }
    void clear() {
        std::vector<Entry> all;
        { std::lock_guard<std::mutex> lock(mu); all.swap(free_list); }
        for (const Entry& e : all)
            if (hipSetDevice(e.device) == hipSuccess) hipFree(e.ptr);
    }
} g_pool;

// Pinned staging buffers of the pipelined first run (upload_inputs: group B's copies).  Copies that are to run BESIDE a kernel
// must be on a stream that does not share its hardware queue with the kernel's (measured, profiles/r06_persist.txt: pageable
// copies on streams made for the purpose, and the same out of pinned memory, returned when group A's phase 1 ended, 122 - 135 ms
// later; on the job's second stream — the one the stream handshake vouches for — 17 ms), and a host thread can only feed such
// a stream asynchronously out of pinned memory.  So group B's host threads stage their sources through these themselves:
// 16 MB pieces, two per thread, kept for the life of the process.
struct PinnedPool {
    static constexpr size_t kBytes = (size_t)16 << 20;
    std::mutex mu;
    std::vector<unsigned char*> free_list;
    unsigned char* get() {
        { std::lock_guard<std::mutex> l(mu); if (!free_list.empty()) { unsigned char* p = free_list.back(); free_list.pop_back(); return p; } }
        void* p = nullptr;
        if (hipHostMalloc(&p, kBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return (unsigned char*)p;
    }
    void put(unsigned char* p) { if (p) { std::lock_guard<std::mutex> l(mu); free_list.push_back(p); } }
    void clear() { std::lock_guard<std::mutex> l(mu); for (unsigned char* p : free_list) hipHostFree(p); free_list.clear(); }
} g_pinned;

}  // namespace

struct pg_job {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<IndexHost> index;
    std::vector<ChainHost> chains;
    uint32_t n_contigs = 0, n_samples = 1;  // chains = n_samples * n_contigs (sample-major)
    bool cohort = false;
    DevContig* d_contigs = nullptr;
    unsigned char* arena = nullptr;
    size_t arena_bytes = 0;
    bool cache_arena = false;  // hand the arena to the process cache on destroy (one-shot call)
    // regions zeroed at the start of every run
    unsigned char* zero_base = nullptr;
    size_t zero_bytes = 0;
    uint32_t* d_small = nullptr;  // [n_small] chain ids of the H = 16 chains
    uint32_t n_small = 0;
    uint32_t* d_smallx = nullptr; // [n_smallx] ... of those with multiallelic objects (k_sweep_small16x), behind d_small's
    uint32_t n_smallx = 0;
    bool smallx_phase2 = false;
    bool small_phase2 = false;    // some of them (fused jobs, class sums: DevContig::small == 2) also run their phase 2 on k_sweep_small16
    double* d_dump = nullptr;
    uint32_t* d_ncols = nullptr;  // [n_index] kept columns of every index contig (k_compact, when the index is uploaded)
    uint32_t* d_ixerr = nullptr;  // [n_index] error bits of the index-level kernels
    DevContig* d_reps = nullptr;  // [n_index] one chain descriptor per index contig: what the index-level kernels walk
    unsigned char* ix_base = nullptr;   // the index-derived arrays of all contigs, one run of the arena (zeroed before every index pass)
    size_t ix_bytes = 0;
    unsigned char* srec_base = nullptr; // the sample records of all split chains, one run (zeroed with every index pass: k_prep_s_bi leaves the zero pieces out)
    size_t srec_bytes = 0;
    bool any_split = false;
    uint32_t max_sb = 0, max_sm4 = 0, max_sw = 0;   // grid extents of the split path's emission kernels
    bool any_legacy_prep = false;
    hipEvent_t ev_ix[2];
    double index_ms = 0.0;
    std::string plan_text;
    std::vector<unsigned char> tab_p;
    size_t o_tab_p = 0;
    uint32_t* d_err = nullptr;    // [n_chains]
    double* d_lik = nullptr;      // packed, chain after chain
    int32_t* d_likexp = nullptr;
    uint64_t n_lik_total = 0;
    size_t o_tab_m = 0, o_tab_e = 0;
    std::vector<double> tab_m;
    std::vector<int32_t> tab_e;
    DevTable tab;
    uint32_t hp_mask = 0, max_v = 0;
    uint32_t max_wide = 0;   // grid extent of k_bins_wide: the longest list of wide-column candidates
    uint32_t max_prep_w = 0, max_prep_m4 = 0;   // grid extents of k_prep (lists or all variants) and k_prep_m4
    uint32_t bins_which = 0;   // bit 0: chains whose bins k_bins forms, bit 1: chains on k_sweep_lean2 (k_bins_lean2), bit 2: k_bins_thin
    uint32_t vit_bits = 0;     // run_phasing: 1 / 2 / 4 = chains with 16 / 32 / 64 padded paths
    hipEvent_t ev_vit[2];
    double vit_ms = 0.0;
    bool vit_tq_ready = false;  // the transition probabilities of the kept columns are on the device (they depend on the index only)
    hipEvent_t ev[PG_N_KERNEL_CLASSES + 1];
    bool events = false;
    double ms[PG_N_KERNEL_CLASSES] = {0, 0, 0, 0, 0, 0};
    double host_s[4] = {0, 0, 0, 0};
    uint64_t up_bytes[2] = {0, 0};
    pg_hmm_params params;
    bool ran = false;
    // Sweep mode.  fused: phase 2 forms the posterior partials inline (k_bins reduces them) — least
    // HBM traffic, the choice when hundreds of chains fill the chip.  chunked: with few chains the
    // chip is idle and a chain's time is its per-column latency, so phase 2 runs as store-only
    // chunks (as cheap per column as phase 1) and k_post forms the posteriors of each finished
    // chunk on the idle CUs from a second stream.
    bool chunked = false;
    uint32_t chunk_cols = 0, n_chunks = 0;
    // chunked jobs whose chains are ALL lean chains, few enough that every workgroup of k_sweep_lean<4> and k_post_loop has a CU
    // of its own: phase 2 is ONE launch of each, chunks handed over through DevContig::sync (PG_KERNELS=nopersist: a launch per chunk)
    bool persist = false;
    uint32_t post_blocks = 0;
    // The pipelined first run of a one-shot job (job_build with cache_arena: upload and run inside ONE call, the host arrays stay
    // valid): the inputs of the LONG chains (group A: the job's wall time) are uploaded first and their preparation + phase 1
    // start while the other chains' inputs (group B, most of the bytes) are still crossing PCIe; B's index pass, preparation and
    // phase 1 then run on the second stream beside A's.  24 chromosomes x 64 paths: 27 ms of H2D in front of a 252 ms run.
    bool pipeline = false, pipeline_capable = false, late_pending = false;
    uint32_t nA = 0;
    std::vector<uint32_t> grp_order;          // chains (= index contigs: 1:1 in such jobs), group A first
    DevContig* d_contigs_g = nullptr;         // the chain descriptors in that order
    DevContig* d_reps_g = nullptr;            // the index descriptors in that order
    std::vector<std::thread> late_threads;    // group B's copies
    std::vector<int> late_rc;
    hipEvent_t ev_late[2];
    bool ev_late_made = false;
    hipStream_t persist_checked = nullptr;   // the stream whose kernels are known to run beside stream2's (streams_concurrent)
    bool persist_checked_any = false;
    uint32_t* d_handshake = nullptr;
    uint32_t persist_runs = 0, persist_fallbacks = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_sweep[PG_SCRATCH_BUFS], ev_post[PG_SCRATCH_BUFS];
    bool events2 = false;
    // Per-sample inputs (read k-mer counts, local coverage) of all chains lie in ONE contiguous run of the arena,
    // [sample_lo, sample_lo + sample_bytes): a cohort's next batch of samples is packed by host threads into a pinned
    // staging buffer of the same layout and moved with a handful of large copies (sample_upload) — and, through
    // pg_job_upload_begin / _end, into a SECOND set of arrays while the current one is being genotyped.
    size_t sample_lo = 0, sample_bytes = 0;
    unsigned char* staging = nullptr;         // pinned, sample_bytes (allocated on first use)
    bool staging_refused = false;             // hipHostMalloc failed once: sample uploads copy chain by chain from pageable memory
    unsigned char* alt_samples = nullptr;     // device: the other set of per-sample arrays (allocated on first pg_job_upload_begin)
    DevContig* d_contigs_alt = nullptr;       // chain descriptors pointing into the other set (the spare of the two descriptor arrays)
    DevContig* d_contigs_owned = nullptr;     // the descriptor array that is not part of the arena (to free)
    std::vector<DevContig> h_contigs;         // host copy of the descriptors of the set in use
    unsigned char* cur_samples = nullptr;     // base of the set in use (arena + sample_lo, or alt_samples)
    hipStream_t copy_stream = nullptr;
    std::thread uploader;                     // pg_job_upload_begin's worker
    bool upload_pending = false;
    int upload_rc = PG_OK;
    std::string upload_err;
    std::vector<std::vector<uint16_t>> next_coverage;   // host copies of the pending batch's coverage arrays
};

extern "C" void pg_job_destroy(pg_job* job) {
    if (!job) return;
    hipSetDevice(job->device);
    if (job->uploader.joinable()) job->uploader.join();
    for (auto& t : job->late_threads) if (t.joinable()) t.join();
    if (job->ev_late_made) { hipEventDestroy(job->ev_late[0]); hipEventDestroy(job->ev_late[1]); }
    if (job->copy_stream) { hipStreamSynchronize(job->copy_stream); hipStreamDestroy(job->copy_stream); }
    if (job->staging) hipHostFree(job->staging);
    if (job->alt_samples) hipFree(job->alt_samples);
    if (job->d_contigs_owned) hipFree(job->d_contigs_owned);
    if (job->stream) hipStreamSynchronize(job->stream);
    if (job->stream2) hipStreamSynchronize(job->stream2);
    if (job->events) {
        for (auto& e : job->ev) hipEventDestroy(e);
        for (auto& e : job->ev_vit) hipEventDestroy(e);
        for (auto& e : job->ev_ix) hipEventDestroy(e);
    }
    if (job->events2)
        for (int q = 0; q < (int)PG_SCRATCH_BUFS; ++q) { hipEventDestroy(job->ev_sweep[q]); hipEventDestroy(job->ev_post[q]); }
    if (job->arena) {
        if (job->cache_arena) g_pool.put(job->device, job->arena, job->arena_bytes);
        else hipFree(job->arena);
    }
    if (job->stream2) hipStreamDestroy(job->stream2);
    if (job->stream) hipStreamDestroy(job->stream);
    delete job;
}

extern "C" void pg_hmm_release_cache(void) {
    int cur = -1;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    g_pool.clear();
    g_pinned.clear();
    if (have) hipSetDevice(cur);
}

namespace {

struct ChainSpec { uint32_t index; const uint16_t* kmer_count; const uint16_t* coverage; };

// The per-sample arrays of a COHORT job (host memory by contract, one pair of small arrays per chain: 8192 pageable copies
// for 4096 chains cost more than the kernels they feed) go through ONE pinned staging buffer laid out like the arena's
// sample run: `pieces` host threads pack a contiguous range of chains each and queue one large H2D copy for it.  The
// reference re-reads the counts per sample (src/commands.cpp:118-138); this is that step for a batch of samples.
// `dev_base` = device address of the sample run to fill; `cov_out[c]` takes the host copy of chain c's coverage.
int sample_upload(pg_job* job, const std::vector<ChainSpec>& specs, unsigned char* dev_base, hipStream_t stream,
                  std::vector<std::vector<uint16_t>>* cov_out, uint64_t* bytes, char* err, size_t errlen) {
    const size_t n = job->chains.size();
    if (!job->staging && !job->staging_refused) {
        hipError_t he = hipHostMalloc((void**)&job->staging, job->sample_bytes ? job->sample_bytes : 8, hipHostMallocDefault);
        if (he != hipSuccess) { (void)hipGetLastError(); job->staging = nullptr; job->staging_refused = true; }
    }
    if (!job->staging) {
        // The host would not pin sample_bytes (tens of GB for a large cohort): the chains' arrays go straight from the
        // caller's pageable buffers, one pair of copies per chain — slower (the runtime stages every copy), never a failure.
        const size_t lo0 = job->sample_lo;
        for (size_t c = 0; c < n; ++c) {
            ChainHost& ch = job->chains[c];
            const IndexHost& x = job->index[ch.index];
            if (x.V == 0) continue;
            HIP_TRY(hipMemcpyAsync(dev_base + (ch.o_cov - lo0), specs[c].coverage, (size_t)x.V * 2, hipMemcpyHostToDevice, stream));
            if (x.sumK) HIP_TRY(hipMemcpyAsync(dev_base + (ch.o_kcnt - lo0), specs[c].kmer_count, (size_t)x.sumK * 2, hipMemcpyHostToDevice, stream));
            if (bytes) *bytes += (uint64_t)x.V * 2 + (uint64_t)x.sumK * 2;
            if (cov_out) (*cov_out)[c].assign(specs[c].coverage, specs[c].coverage + x.V);
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return PG_OK;
    }
    size_t pieces = n / 64;   // (at most one copy per 64 chains)
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t cap = hw ? (hw > 32 ? 16 : (hw + 1) / 2) : 4;
    if (pieces > cap) pieces = cap;
    if (pieces < 1) pieces = 1;
    std::vector<int> rcs(pieces, (int)hipSuccess);
    std::vector<uint64_t> moved(pieces, 0);
    const size_t lo0 = job->sample_lo;
    auto work = [&](size_t k) {
        if (hipSetDevice(job->device) != hipSuccess) { rcs[k] = (int)hipErrorInvalidDevice; return; }
        const size_t c0 = n * k / pieces, c1 = n * (k + 1) / pieces;
        if (c0 >= c1) return;
        for (size_t c = c0; c < c1; ++c) {
            ChainHost& ch = job->chains[c];
            const IndexHost& x = job->index[ch.index];
            if (x.V == 0) continue;
            memcpy(job->staging + (ch.o_cov - lo0), specs[c].coverage, (size_t)x.V * 2);
            if (x.sumK) memcpy(job->staging + (ch.o_kcnt - lo0), specs[c].kmer_count, (size_t)x.sumK * 2);
            moved[k] += (uint64_t)x.V * 2 + (uint64_t)x.sumK * 2;
            if (cov_out) (*cov_out)[c].assign(specs[c].coverage, specs[c].coverage + x.V);
        }
        const size_t b0 = job->chains[c0].o_cov - lo0;
        const size_t b1 = c1 < n ? job->chains[c1].o_cov - lo0 : job->sample_bytes;
        if (b1 > b0) rcs[k] = (int)hipMemcpyAsync(dev_base + b0, job->staging + b0, b1 - b0, hipMemcpyHostToDevice, stream);
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < pieces; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    for (size_t k = 0; k < pieces; ++k) {
        if (rcs[k] != (int)hipSuccess) { set_err(err, errlen, "sample upload: %s", hipGetErrorString((hipError_t)rcs[k])); return PG_ERR_DEVICE; }
        if (bytes) *bytes += moved[k];
    }
    HIP_TRY(hipStreamSynchronize(stream));   // (the staging buffer is free again, the set is complete)
    return PG_OK;
}

bool streams_concurrent(pg_job* job, hipStream_t s);   // (below, in front of pg_job_run)
// H2D of the inputs into a planned job.  Copies are queued on the job's stream; pageable sources are
// staged by the runtime, so every call returns when its source has been read.
int upload_inputs(pg_job* job, const pg_contig_batch* batches, const std::vector<ChainSpec>& specs, bool with_index,
                  char* err, size_t errlen) {
    const double t0 = now_s();
    unsigned char* A = job->arena;
    hipStream_t s = job->stream;
    uint64_t bi = 0, bs = 0;
    // the copies are collected first: a large upload from pageable buffers (a whole genome through the one-shot call: 264
    // arrays, 1.06 GB) is then issued from several host threads on streams of their own — the runtime stages a pageable
    // source through its bounce buffers on the calling thread, one copy after the other: 24 ms on one thread
    struct Copy { void* dst; const void* src; size_t bytes; };
    std::vector<Copy> copies, late_copies;   // late: group B of a pipelined first run (pg_job::pipeline) — issued here, waited for in pg_job_run
    // (the pipelined first run needs the job's two streams to run side by side: asked once, here, where nothing is in flight)
    if (with_index && job->pipeline && !streams_concurrent(job, s)) job->pipeline = false;
    const bool pipe = with_index && job->pipeline;
    job->pipeline = false;   // (one upload: asked for again by whoever uploads and runs inside one call)
    std::vector<char> late(job->chains.size(), 0);
    if (pipe) for (size_t k = job->nA; k < job->grp_order.size(); ++k) late[job->grp_order[k]] = 1;
    bool cur_late = false;
#define UP(off, src, bytes, acc)                                                                              \
    do {                                                                                                      \
        if ((bytes) > 0) {                                                                                    \
            (cur_late ? late_copies : copies).push_back({(void*)(A + (off)), (const void*)(src), (size_t)(bytes)});  /* (host or device source) */ \
            acc += (uint64_t)(bytes);                                                                         \
        }                                                                                                     \
    } while (0)
    if (with_index) {
        for (size_t i = 0; i < job->index.size(); ++i) {
            const pg_contig_batch& b = batches[i];
            IndexHost& x = job->index[i];
            if (x.V == 0) continue;
            cur_late = pipe && late[i];   // (pipelined jobs: chain i over index contig i)
            UP(x.o_pos, b.variant_pos, (size_t)x.V * 8, bi);
            if (job->params.run_phasing) { x.pos.assign(b.variant_pos, b.variant_pos + x.V); job->vit_tq_ready = false; }
            UP(x.o_koff, b.kmer_off, ((size_t)x.V + 1) * 4, bi);
            UP(x.o_aoff, b.allele_off, ((size_t)x.V + 1) * 4, bi);
            UP(x.o_aid, b.allele_id, (size_t)x.sumA * 2, bi);
            UP(x.o_aflag, b.allele_flags, (size_t)x.sumA, bi);
            UP(x.o_akoff, b.allele_kmer_off, (size_t)x.sumA * 2, bi);
            UP(x.o_akmask, b.allele_kmer_mask, (size_t)x.sumA * 4, bi);
            UP(x.o_pa, b.path_allele, (size_t)x.V * x.H * 2, bi);
            UP(x.o_goff, x.goff.data(), ((size_t)x.V + 1) * 8, bi);
            if (x.wide_bytes) UP(x.o_widx, x.widx.data(), (size_t)x.V * 4, bi);
            if (x.aux_bytes) UP(x.o_auxidx, x.auxidx.data(), (size_t)x.V * 4, bi);
            UP(x.o_list_m4, x.list_m4.data(), x.list_m4.size() * 4, bi);
            UP(x.o_list_w, x.list_w.data(), x.list_w.size() * 4, bi);
            UP(x.o_list_b, x.list_b.data(), x.list_b.size() * 4, bi);
            UP(x.o_list_big, x.list_big.data(), x.list_big.size() * 4, bi);
        }
        cur_late = false;
        UP(job->o_tab_m, job->tab_m.data(), job->tab_m.size() * sizeof(double), bi);
        UP(job->o_tab_e, job->tab_e.data(), job->tab_e.size() * sizeof(int32_t), bi);
        UP(job->o_tab_p, job->tab_p.data(), job->tab_p.size(), bi);
    }
    if (job->cohort) {   // (host arrays by contract: packed into pinned staging, a few large copies)
        std::vector<std::vector<uint16_t>> cov(job->chains.size());
        const int rc = sample_upload(job, specs, job->cur_samples, s, &cov, &bs, err, errlen);
        if (rc != PG_OK) return rc;
        for (size_t c = 0; c < job->chains.size(); ++c) job->chains[c].coverage.swap(cov[c]);
    } else {
        const size_t rebase = (size_t)(job->cur_samples - (A + job->sample_lo));   // (0: jobs with device-resident counts never switch sets)
        for (size_t c = 0; c < job->chains.size(); ++c) {
            ChainHost& ch = job->chains[c];
            const IndexHost& x = job->index[ch.index];
            if (x.V == 0) continue;
            cur_late = pipe && late[c];
            UP(ch.o_cov + rebase, specs[c].coverage, (size_t)x.V * 2, bs);
            UP(ch.o_kcnt + rebase, specs[c].kmer_count, (size_t)x.sumK * 2, bs);
            ch.coverage.assign(specs[c].coverage, specs[c].coverage + x.V);
        }
    }
#undef UP
    // `list` split over up to four host threads (largest first onto the least loaded), each on a stream of its own; a thread
    // returns when its copies have been read AND have arrived
    auto shares_of = [](const std::vector<Copy>& list, unsigned nt) {
        std::vector<std::vector<Copy>> share(nt);
        std::vector<size_t> load(nt, 0), order(list.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return list[x].bytes > list[y].bytes; });
        for (size_t i : order) {
            unsigned best = 0;
            for (unsigned t = 1; t < nt; ++t) if (load[t] < load[best]) best = t;
            share[best].push_back(list[i]); load[best] += list[i].bytes;
        }
        return share;
    };
    const int device = job->device;
    auto copy_worker = [device](std::vector<Copy> mine, int* rc) {   // (by value: group B's threads outlive this call)
        hipStream_t st = nullptr;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (const Copy& c : mine) { if (e != hipSuccess) break; e = hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDefault, st); }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (st) hipStreamDestroy(st);
        *rc = (int)e;
    };
    {
        size_t total = 0;
        for (const Copy& c : copies) total += c.bytes;
        for (const Copy& c : late_copies) total += c.bytes;
        const unsigned nt = (total >= ((size_t)64 << 20) && copies.size() + late_copies.size() >= 8) ? 4u : 1u;
        if (nt > 1) {
            std::vector<std::vector<Copy>> share = shares_of(copies, nt);
            std::vector<int> rcs(nt, (int)hipSuccess);
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(copy_worker, share[t], &rcs[t]);
            copy_worker(share[0], &rcs[0]);
            for (auto& t : th) t.join();
            for (unsigned t = 0; t < nt; ++t)
                if (rcs[t] != (int)hipSuccess) { set_err(err, errlen, "input upload: %s", hipGetErrorString((hipError_t)rcs[t])); return PG_ERR_DEVICE; }
        } else {
            for (const Copy& c : copies) HIP_TRY(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDefault, s));
        }
        if (pipe) {
            // group B: on its way from here on; pg_job_run joins the threads when group A's kernels are running.  Staged through
            // pinned pieces by the thread itself (g_pinned: a pageable copy would wait for group A's kernels to end); a source
            // that is device memory is copied as it is.
            // All of them on the job's SECOND stream — the one stream known to run beside `s` (streams_concurrent): a stream made here
            // may share its hardware queue with `s`, and its copies then wait for group A's phase 1 to end (measured: 130 ms).
            hipStream_t st2 = job->stream2;
            auto staged_worker = [device, st2](std::vector<Copy> mine, int* rc) {
                hipStream_t st = st2;
                hipEvent_t ev[2] = {nullptr, nullptr};
                unsigned char* buf[2] = {g_pinned.get(), g_pinned.get()};
                hipError_t e = hipSetDevice(device);
                for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
                if (e == hipSuccess && (!buf[0] || !buf[1])) e = hipErrorOutOfMemory;
                size_t piece = 0;
                for (const Copy& c : mine) {
                    if (e != hipSuccess) break;
                    hipPointerAttribute_t at;
                    const bool on_device = hipPointerGetAttributes(&at, c.src) == hipSuccess && at.type == hipMemoryTypeDevice;
                    if (!on_device) (void)hipGetLastError();
                    if (on_device) { e = hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToDevice, st); continue; }
                    for (size_t off = 0; off < c.bytes && e == hipSuccess; off += PinnedPool::kBytes, ++piece) {
                        const size_t nby = std::min(PinnedPool::kBytes, c.bytes - off);
                        const int k = (int)(piece & 1u);
                        if (piece >= 2) e = hipEventSynchronize(ev[k]);   // (the piece two back has left this buffer)
                        if (e != hipSuccess) break;
                        memcpy(buf[k], (const unsigned char*)c.src + off, nby);
                        e = hipMemcpyAsync((unsigned char*)c.dst + off, buf[k], nby, hipMemcpyHostToDevice, st);
                        if (e == hipSuccess) e = hipEventRecord(ev[k], st);
                    }
                }
                // (the last pieces have left the pinned buffers before they go back to the pool; the stream itself is not waited for:
                //  group B's kernels are queued behind these copies, on the same stream)
                for (int k = 0; k < 2; ++k) if (ev[k] && piece > (size_t)k) { const hipError_t e2 = hipEventSynchronize(ev[k]); if (e == hipSuccess) e = e2; }
                for (int k = 0; k < 2; ++k) { if (ev[k]) hipEventDestroy(ev[k]); g_pinned.put(buf[k]); }
                *rc = (int)e;
            };
            const unsigned ntb = late_copies.size() >= 8 ? 4u : 1u;
            std::vector<std::vector<Copy>> share = shares_of(late_copies, ntb);
            job->late_rc.assign(ntb, (int)hipSuccess);
            job->late_threads.clear();
            for (unsigned t = 0; t < ntb; ++t) job->late_threads.emplace_back(staged_worker, share[t], &job->late_rc[t]);
            job->late_pending = true;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (with_index) {
        // The index pass (pg_split.h): what the index alone decides — kept columns, present alleles, the column list; for split
        // chains the per-column index records — is formed NOW, once, for every chain over the index (reference: ColumnIndexer and
        // TransitionProbabilityComputer read positions and path alleles only, src/columnindexer.cpp:8-33,
        // src/transitionprobabilitycomputer.cpp:8-19).  A run of the job forms only what hangs on the sample's counts.
        HIP_TRY(hipEventRecord(job->ev_ix[0], s));
        HIP_TRY(hipMemsetAsync(job->ix_base, 0, job->ix_bytes, s));
        if (job->srec_bytes) HIP_TRY(hipMemsetAsync(job->srec_base, 0, job->srec_bytes, s));
        uint32_t max_big = 0;
        for (const IndexHost& x : job->index) max_big = std::max<uint32_t>(max_big, (uint32_t)x.list_big.size());
        if (pipe) pgk_launch_index(job->d_reps_g, job->nA, job->max_v, max_big, 0, s);   // (group B's: pg_job_run, when its inputs have arrived)
        else pgk_launch_index(job->d_reps, (uint32_t)job->index.size(), job->max_v, max_big, job->any_split ? 1 : 0, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(job->ev_ix[1], s));
        HIP_TRY(hipStreamSynchronize(s));
        float ims = 0.f;
        HIP_TRY(hipEventElapsedTime(&ims, job->ev_ix[0], job->ev_ix[1]));
        job->index_ms = ims;
    }
    job->up_bytes[0] = bi; job->up_bytes[1] = bs;
    job->host_s[1] = now_s() - t0;
    return PG_OK;
}

// PG_KERNELS: the ONE override of the kernel choice — a comma-separated list of tokens that put a second implementation
// of a step beside the default one (cross-checks in tests/, measurements in tools/).  Nothing here is needed
// in production: the library takes every decision from the job (chain count, panel width, allele structure).
//   general | generic   no lean / lean-x / small kernels | the generic kernel for every HP >= 64
//   leanx | noleanx     k_sweep_leanx also in phase 1 of fused jobs | never
//   small | nosmall     k_sweep_small16 whatever the chain count | never
//   nolean2 notri nocls4 prepwave   phase 2 of triangle chains on the general kernel's ring | full columns | per-thread
//                       partials instead of class sums | k_prep for every object
//   fullcols            fused jobs at HP = 32 store and fetch whole 32 x 32 columns (DevContig::live = HP)
//   nosmall2            phase 2 of the 16-path chains of fused jobs on the general kernel (k_sweep_small16 for phase 1 only)
//   noleanx2            phase 2 of the 64-path triangle chains with multiallelic objects on the general kernel's triangle ring (per-thread
//                       partials) instead of k_sweep_leanx2 — cross-check
//   nowidef             a job with wide columns on 64-path chains of the general kernel runs chunked (as until round 6) instead of fused with
//                       those columns through their aux slots + k_bins_wide (DevContig::widef) — cross-check
//   persist             chunked jobs whose chains are all lean chains run phase 2 as the persistent pair k_sweep_lean<4> + k_post_loop
//                       (one launch each, chunks handed over on the device) instead of one launch per chunk (k_sweep_lean<3> + k_post).
//                       Opt-in: measured at par or behind on the whole-genome job (profiles/r06_persist.txt).  nopersist: the default, spelled out
//   nosplit             the 16-path chains of fused jobs prepare every variant per sample (k_prep*, k_records, k_bins_lean2 / _x) instead of
//                       taking the split path (pg_split.h)
struct KernelChoice {
    bool general = false, generic = false, nolean2 = false, notri = false, nocls4 = false, prepwave = false, fullcols = false, nosmall2 = false, nosplit = false;
    bool persist = false, noleanx2 = false, nowidef = false;
    int leanx = -1, small = -1;   // -1: by the job, 0 / 1: forced
    std::string unknown;          // a token this list does not know
};
KernelChoice kernel_choice() {
    KernelChoice k;
    const char* e = getenv("PG_KERNELS");
    if (!e) return k;
    std::string tok;
    auto take = [&]() {
        if (tok == "general") k.general = true; else if (tok == "generic") k.generic = true;
        else if (tok == "leanx") k.leanx = 1; else if (tok == "noleanx") k.leanx = 0;
        else if (tok == "small") k.small = 1; else if (tok == "nosmall") k.small = 0;
        else if (tok == "nolean2") k.nolean2 = true; else if (tok == "notri") k.notri = true;
        else if (tok == "nocls4") k.nocls4 = true; else if (tok == "prepwave") k.prepwave = true;
        else if (tok == "fullcols") k.fullcols = true; else if (tok == "nosmall2") k.nosmall2 = true;
        else if (tok == "nosplit") k.nosplit = true;
        else if (tok == "noleanx2") k.noleanx2 = true;
        else if (tok == "nowidef") k.nowidef = true;
        else if (tok == "persist") k.persist = true; else if (tok == "nopersist") k.persist = false;
        else if (!tok.empty()) k.unknown = tok;   // (a typo would quietly test the default path against itself: job creation fails)
        tok.clear();
    };
    for (const char* c = e; *c; ++c) { if (*c == ',' || *c == ' ') take(); else tok.push_back(*c); }
    take();
    return k;
}

// f(i) for every index contig: on worker threads when the contigs together hold enough variants to pay for them
template <class F>
void parallel_contigs(uint32_t n, const pg_contig_batch* batches, F&& f) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += batches[i].n_variants;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? (nt > 16 ? 16 : nt) : 4;
    if (nt > n) nt = n;
    if (n < 2 || total < 200000 || nt < 2) { for (uint32_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    auto work = [&]() { for (uint32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i); };
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

// (chunk_cap: 0, or the most chunk columns this attempt may plan for — the retry of a chunked job whose arena did not fit)
int job_build(int device, uint32_t n_index, const pg_contig_batch* batches, const std::vector<ChainSpec>& specs,
              uint32_t n_samples, bool cohort, const pg_table* table, const pg_hmm_params* params, bool cache_arena,
              pg_job** out, char* err, size_t errlen, size_t chunk_cap = 0) {
    *out = nullptr;
    if (!batches || !table || !params || n_index == 0 || specs.empty()) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
#ifdef PG_HOST_TIMING   // (measurement builds: where the host time of a job's construction goes)
    const double tb0 = now_s();
    double tb_last = tb0;
    auto lap = [&](const char* what) { const double t = now_s(); fprintf(stderr, "[job_build] %-28s %8.3f ms\n", what, (t - tb_last) * 1e3); tb_last = t; };
#else
    auto lap = [](const char*) {};
#endif
    if (params->run_phasing) {
        // Viterbi (pg_viterbi.hip): a row of states is one wave's lanes, so at most 64 selected paths (the reference's
        // own callers pass at most 30, src/commands.cpp:939)
        for (const ChainSpec& sp : specs)
            if (batches[sp.index].n_paths > 64u) {
                set_err(err, errlen, "run_phasing: %u selected paths, the device Viterbi takes at most 64", batches[sp.index].n_paths);
                return PG_ERR_UNSUPPORTED;
            }
    }
    {   // (per-variant validation: contigs on worker threads when there is enough of it — a merged one-shot job of a whole
        //  genome spent 3 + 22 ms here and in the planning loop below, single-threaded, per round)
        std::vector<int> rcs(n_index, PG_OK);
        std::vector<std::string> msgs(n_index);
        parallel_contigs(n_index, batches, [&](uint32_t i) {
            char e[256] = {0};
            rcs[i] = check_batch(&batches[i], !cohort, e, sizeof(e));
            if (rcs[i] != PG_OK) msgs[i] = e;
        });
        for (uint32_t i = 0; i < n_index; ++i)
            if (rcs[i] != PG_OK) { set_err(err, errlen, "%s", msgs[i].c_str()); return rcs[i]; }
    }
    for (const ChainSpec& sp : specs) {
        const pg_contig_batch& b = batches[sp.index];
        if (b.n_variants > 0 && (!sp.coverage || (b.kmer_off[b.n_variants] > 0 && !sp.kmer_count))) {
            set_err(err, errlen, "a chain has null kmer_count / coverage arrays");
            return PG_ERR_INVALID;
        }
    }
    lap("batch checks");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(err, errlen, "no HIP device available (no CPU fallback)"); return PG_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "bad device %d", device); return PG_ERR_INVALID; }
    if (hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "hipSetDevice failed"); return PG_ERR_DEVICE; }

    const uint32_t n_chains = (uint32_t)specs.size();
    pg_job* job = new pg_job();
    job->device = device;
    job->params = *params;
    job->cache_arena = cache_arena;
    job->cohort = cohort;
    job->n_samples = n_samples;
    job->n_contigs = n_index;
    auto fail = [&](int rc, const char* what, hipError_t e) -> int {
        set_err(err, errlen, "%s: %s", what, hipGetErrorString(e));
        pg_job_destroy(job);
        return rc;
    };
    hipError_t he;
    if ((he = hipStreamCreate(&job->stream)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipStreamCreate", he);
    for (auto& e : job->ev)
        if ((he = hipEventCreate(&e)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipEventCreate", he);
    for (auto& e : job->ev_vit)
        if ((he = hipEventCreate(&e)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipEventCreate", he);
    for (auto& e : job->ev_ix)
        if ((he = hipEventCreate(&e)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipEventCreate", he);
    job->events = true;
    table_snapshot(const_cast<pg_table*>(table), job->tab_m, job->tab_e, job->tab_p, &job->tab);

    lap("streams, events, table");
    // ---- index contigs -------------------------------------------------------------------
    job->index.resize(n_index);
    bool wide_candidates = false, generic_needed = false;
    uint32_t max_v = 0;
    bool lean_ok = true, force_generic = false;
    const KernelChoice kc = kernel_choice();
    if (!kc.unknown.empty()) {
        set_err(err, errlen, "PG_KERNELS: unknown token '%s'", kc.unknown.c_str());
        pg_job_destroy(job);
        return PG_ERR_INVALID;
    }
    force_generic = kc.generic; lean_ok = !kc.general && !kc.generic;
    std::vector<char> wide_overflow(n_index, 0);
    parallel_contigs(n_index, batches, [&](uint32_t i) {
        const pg_contig_batch& b = batches[i];
        IndexHost& x = job->index[i];
        x.V = b.n_variants; x.H = b.n_paths; x.HP = pad_paths(x.H ? x.H : 1);
        x.T = pgk_threads_for_hp(x.HP); x.RB = pg_rec_bytes(x.HP);
        x.sumK = x.V ? b.kmer_off[x.V] : 0; x.sumA = x.V ? b.allele_off[x.V] : 0;
        x.goff.assign((size_t)x.V + 1, 0);
        x.n_kmers.resize(x.V);
        if (x.V) { x.koff.assign(b.kmer_off, b.kmer_off + x.V + 1); x.aoff.assign(b.allele_off, b.allele_off + x.V + 1); }
        uint64_t maxA = 1, woff = 0;
        bool two_alleles = true;
        uint32_t maxK = 0, n_bi = 0;
        for (uint32_t v = 0; v < x.V; ++v) {
            const uint64_t A = b.allele_off[v + 1] - b.allele_off[v];
            x.goff[v + 1] = x.goff[v] + A * (A + 1) / 2;
            if (A > maxA) maxA = A;
            if (A > 32u) x.list_big.push_back(v);
            if (A != 2) two_alleles = false;
            x.n_kmers[v] = (uint16_t)(b.kmer_off[v + 1] - b.kmer_off[v]);
            if (b.kmer_off[v + 1] - b.kmer_off[v] > maxK) maxK = b.kmer_off[v + 1] - b.kmer_off[v];
            if (A == 2 && b.kmer_off[v + 1] - b.kmer_off[v] <= 32u) n_bi += 1;
        }
        {
            // k_prep_bi (four variants per wave) takes the two-allele objects with <= 32 k-mers of chains with <= 64 paths:
            // 1 = the whole chain is such objects, 2 = at least half of it (k_prep takes the rest: HPRC-style panels and the
            // 15 + 1 sampled paths have a fifth of their objects multiallelic); PG_KERNELS=prepwave: k_prep for everything (cross-check)
            const bool ok = x.H <= 64u && x.V > 0 && !kc.prepwave;
            x.prep_fast = !ok ? 0u : ((two_alleles && maxK <= 32u) ? 1u : (2u * n_bi >= x.V ? 2u : 0u));
            x.all_sb = two_alleles && maxK <= 32u;
            // (the same lists serve the split path's emission kernels — 16-path chains, whatever the share of biallelic objects)
            if (x.prep_fast == 2u || (lean_ok && x.HP == 16u && x.H == 16u && !x.all_sb)) {   // who prepares what: k_prep_bi scans the chain for its own, the others walk lists
                for (uint32_t v = 0; v < x.V; ++v) {
                    const uint64_t A = b.allele_off[v + 1] - b.allele_off[v];
                    const uint32_t Kv = b.kmer_off[v + 1] - b.kmer_off[v];
                    if (A == 2 && Kv <= 32u) { x.list_b.push_back(v); continue; }
                    if (A >= 3 && A <= PG_AMAX && Kv <= 64u) x.list_m4.push_back(v);
                    else x.list_w.push_back(v);
                }
            }
        }
        x.n_lik = x.goff[x.V];
        x.pair_n = (uint32_t)(maxA < PG_AMAX ? maxA : PG_AMAX);
        x.part_slots = (x.pair_n + 1u) & ~1u;  // partials are stored as allele pairs (16-byte stores)
        if (x.part_slots < 2) x.part_slots = 2;
        if (maxA > PG_AMAX) {
            // variants with more than PG_AMAX alleles may turn into WIDE columns (as many distinct alleles
            // on the selected paths): room for min(A, H) local alleles each
            x.widx.assign(x.V, PG_WIDE_NONE);
            for (uint32_t v = 0; v < x.V; ++v) {
                const uint64_t A = b.allele_off[v + 1] - b.allele_off[v];
                if (A > PG_AMAX) {
                    uint64_t nl = A < x.H ? A : x.H;
                    if (nl > PG_WIDE_MAX) nl = PG_WIDE_MAX;
                    x.widx[v] = (uint32_t)(woff / 16);
                    woff += pg_wide_entry_bytes((uint32_t)nl);
                    if (woff / 16 >= 0xFFFFFFF0ull) { wide_overflow[i] = 1; return; }
                }
            }
            x.wide_bytes = woff;
        }
        x.lean = lean_ok && x.HP == 64 && x.H == 64 && maxA == 2 && x.V > 0;
        x.small = lean_ok && x.HP == 16 && x.H == 16 && maxA == 2 && x.V > 0;   // (and enough of them: below)
        // the same shape with multiallelic objects — 15 sampled paths + the reference path, bubbles keeping every allele those
        // paths carry (src/commands.cpp:799-803, src/multiallelicuniquekmers.cpp:195-232): k_sweep_small16x, any allele count
        // per object (narrow columns from the record's table, wide ones from the side table: per COLUMN)
        x.smallx = lean_ok && x.HP == 16 && x.H == 16 && maxA > 2 && x.V > 0;
        if (x.smallx) {
            // aux slots: a variant with 3 .. PG_AMAX alleles leaves phase 2 as its (up to) fifteen bins (128 bytes), one with
            // more as a whole column (a wide column) — or as bins, if its paths carry at most PG_AMAX of them
            x.auxidx.assign(x.V, PG_WIDE_NONE);
            uint64_t ao = 0;
            for (uint32_t v = 0; v < x.V; ++v) {
                const uint64_t A = b.allele_off[v + 1] - b.allele_off[v];
                if (A <= 2) continue;
                const uint64_t small_slot = 128u /* fifteen bins */, col_slot = (uint64_t)x.HP * x.HP * 8u;
                x.auxidx[v] = (uint32_t)(ao / 16);
                if (A > PG_AMAX) x.n_wide_cand += 1;
                ao += A > PG_AMAX ? (col_slot > small_slot ? col_slot : small_slot) : small_slot;
            }
            if (ao / 16 >= 0xFFFFFFF0ull) x.smallx = false;   // (the general kernel then)
            else x.aux_bytes = ao;
        }
        {
            // PG_KERNELS=nocls4: per-thread partials + k_bins (cross-check)
            // (the class sums are formed by a half-chain's ONE compute wave: 16 paths, and 32 when the kernel is built with 16 rows per lane)
            x.cls4 = (x.HP == 16 || (x.HP == 32 && pgk_threads_for_hp(32) == 64u)) && x.H == x.HP && maxA == 2 && x.V > 0 && !kc.nocls4;
        }
        {
            // PG_KERNELS=noleanx: the general kernel (cross-check)
            x.leanx = lean_ok && (x.HP == 128 || x.HP == 64) && !x.lean && maxA >= 1 && maxA <= PG_AMAX && x.V > 0 && kc.leanx != 0;
        }
        if (x.HP == 64 && maxA > PG_AMAX && x.V > 0 && !kc.nowidef && !force_generic && params->run_genotyping && !params->run_phasing) {
            // DevContig::widef: an aux slot of one column (8 HP^2 bytes) for every variant that may turn into a wide column
            x.auxidx.assign(x.V, PG_WIDE_NONE);
            uint64_t ao = 0;
            for (uint32_t v = 0; v < x.V; ++v) {
                const uint64_t A = b.allele_off[v + 1] - b.allele_off[v];
                if (A <= PG_AMAX) continue;
                x.auxidx[v] = (uint32_t)(ao / 16);
                x.n_wide_cand += 1;
                ao += (uint64_t)x.HP * x.HP * 8u;
            }
            if (ao / 16 < 0xFFFFFFF0ull) { x.aux_bytes = ao; x.widef_cand = true; }
            else { x.auxidx.clear(); x.n_wide_cand = 0; }
        }
    });
    for (uint32_t i = 0; i < n_index; ++i) {
        const IndexHost& x = job->index[i];
        if (wide_overflow[i]) { pg_job_destroy(job); set_err(err, errlen, "wide-column tables exceed 64 GB"); return PG_ERR_UNSUPPORTED; }
        if (x.wide_bytes) wide_candidates = true;
        if (x.HP >= 256) generic_needed = true;
        if (x.V > max_v) max_v = x.V;
    }
    job->max_v = max_v;
    lap("index planning");
    {
        // k_sweep_small16 packs four H = 16 half-chains into a wave: a throughput kernel.  A single chain is faster on the
        // general kernel (four states per lane instead of sixteen: 375 vs 470 ns per column); PG_KERNELS=small / nosmall forces.
        // Where the whole STEP crosses over (round 6, tools/exp_small_crossover.py, profiles/r06_small_crossover.txt: 8 contigs x
        // 8000 variants per sample, fused; with the small kernels comes the split path, 0.6 against 2.5 ms of per-variant work at
        // 512 chains) — step ms, small kernels / general kernel:
        //   chains        64           128          256          512           1024
        //   biallelic     5.99 / 5.39  6.13 / 5.82  6.33 / 7.35  6.98 / 9.28   9.16 / 14.01
        //   a fifth 3-5   8.15 / 5.32  8.39 / 6.15  8.71 / 8.33  9.30 / 12.01  11.81 / 19.48
        // -> from 256 chains on when every such chain is biallelic, from 320 when some have multiallelic objects (512 for both
        //    until round 6: a 256-chain biallelic cohort ran 14 % below what it could).
        size_t n_small_chains = 0, n_x_chains = 0;
        for (const ChainSpec& sp : specs) {
            n_small_chains += (job->index[sp.index].small || job->index[sp.index].smallx) ? 1 : 0;
            n_x_chains += job->index[sp.index].smallx ? 1 : 0;
        }
        const bool use = kc.small >= 0 ? kc.small == 1 : n_small_chains >= (n_x_chains ? 320u : 256u);
        if (!use) for (auto& x : job->index) { x.small = false; x.smallx = false; if (!x.widef_cand) x.aux_bytes = 0; }
    }
    // A WIDE column (more than PG_AMAX alleles on the selected paths) costs that column, not the job, on the kernels that
    // take it inside a fused phase 2 — k_sweep_small16x (its column goes to the aux slot, k_bins_wide forms the bins); a
    // chain of any other kernel with such an object still makes the job chunked (k_post is their only wide path)
    wide_candidates = false;
    for (const IndexHost& x : job->index) {
        // (a k_sweep_small16x chain left with ONE column that is wide has its bins formed by k_bins_wide_s — the split path; jobs
        //  that cannot take the split path — run_phasing, 2^32 bins — go chunked as before round 5 instead of failing such a chain
        //  with PG_DEVERR_WIDE_FUSED at run time.  PG_KERNELS=nosplit keeps the fused job: a cross-check switch.)
        const bool split_possible = kc.nosplit || (!params->run_phasing && params->run_genotyping && x.n_lik < 0xFFFFFFF0ull);
        if (x.wide_bytes && !(x.smallx && !kc.nosmall2 && split_possible) && !x.widef_cand) wide_candidates = true;
    }

    // ---- sweep mode ------------------------------------------------------------------------
    {
        size_t per_col = 0;  // scratch bytes per chunk column over all chains (PG_SCRATCH_BUFS buffers x 2 roles)
        for (const ChainSpec& sp : specs) { const IndexHost& x = job->index[sp.index]; per_col += (size_t)2 * PG_SCRATCH_BUFS * x.HP * x.HP * sizeof(double); }
        // wide columns of chains that are not k_sweep_small16x's (wide_candidates, above) and HP >= 256 have their posteriors formed by k_post only: such jobs always run chunked
        bool want = n_chains * 2u < 128u;  // fewer workgroups than half the CUs
        // merged one-shot calls (cache_arena): always the mode — and with it the kernels — every one of them runs alone, so
        // that a caller's result does not depend on who else happened to be in flight (bit for bit, by construction)
        if (cache_arena) want = true;
        if (const char* m = getenv("PG_SWEEP_MODE")) {
            if (!strcmp(m, "fused")) want = false;
            else if (!strcmp(m, "chunked")) want = true;
        }
        if (wide_candidates || generic_needed || force_generic) want = true;
        job->chunked = want && max_v > 0 && params->run_genotyping;
        // k_sweep_leanx is a latency kernel (lone chains: 1470 vs 2645 ns per column at 128 paths); phase 1 of a fused job
        // with hundreds of chains is bound by HBM writes, where the general kernel measured faster (7.8 vs 8.9 ms on 128 chains
        // of 128 paths).  PG_KERNELS=leanx forces it there too.
        if (job->chunked) for (auto& x : job->index) { x.cls4 = false; x.aux_bytes = 0; }   // (chunked jobs form their posteriors in k_post)
        if (kc.nosmall2) for (auto& x : job->index) if (!x.widef_cand) x.aux_bytes = 0;
        for (auto& x : job->index) x.widef = x.widef_cand && !job->chunked && x.aux_bytes != 0;
        if (!job->chunked) {
            if (kc.leanx != 1) for (auto& x : job->index) x.leanx = false;
        }
        if (job->chunked) {
            // 8192 columns per chunk (round 6; 4096 before): 25 chunk rounds instead of 50 on the whole genome.  With k_post no longer
            // the bound (post_lean64, the idle CUs shared out over the chains still running) what a chunk round costs beyond its columns
            // is its launches and resume prologues, and how much THAT is varies from box to box: 4096 / 6144 / 8192 / 12288 columns:
            // 128.7 / 128.5 / 126.9 / - ms of phase 2 on one box, 133.8 / 127.8 / 127.9 / 127.5 on another (profiles/r06_persist.txt 8).
            size_t k = 8192;
            const size_t budget = (size_t)39 << 30;   // (three buffers: 8192 columns for the 24 chains of a whole genome at 64 paths)
            if (per_col * k > budget) k = budget / per_col;
            if (const char* e = getenv("PG_CHUNK_COLS")) { const long v = strtol(e, nullptr, 0); if (v > 0) k = (size_t)v; }
            if (chunk_cap && k > chunk_cap) k = chunk_cap;
            const size_t half = (size_t)max_v / 2 + 1;
            if (k > half) k = half;
            if (k < 1) k = 1;
            job->chunk_cols = (uint32_t)k;
            job->n_chunks = (uint32_t)((half + k - 1) / k);
            if ((he = hipStreamCreateWithFlags(&job->stream2, hipStreamNonBlocking)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipStreamCreate", he);
            for (int q = 0; q < (int)PG_SCRATCH_BUFS; ++q) {
                if ((he = hipEventCreateWithFlags(&job->ev_sweep[q], hipEventDisableTiming)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipEventCreate", he);
                if ((he = hipEventCreateWithFlags(&job->ev_post[q], hipEventDisableTiming)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipEventCreate", he);
            }
            job->events2 = true;
            // the persistent phase 2: every chain a lean chain (or empty), an even chunk (the lean step runs two columns per loop
            // iteration) unless a chunk is the whole half, and a CU for every workgroup of the two kernels that wait for each other
            bool all_lean = kc.persist, any_lean = false;
            for (const ChainSpec& sp : specs) { const IndexHost& x = job->index[sp.index]; if (x.V && !x.lean) all_lean = false; if (x.lean) any_lean = true; }
            if (all_lean && any_lean && ((job->chunk_cols & 1u) == 0u || job->n_chunks == 1u)) {
                uint32_t cus = 0;
                uint32_t pb = pgk_post_blocks((uint32_t)n_chains, job->chunk_cols, &cus);
                // (the sweep's workgroups MUST all be running — a k_post_loop block that is not merely takes no work —: eight CUs are
                //  left free beside the two grids, whatever else may hold a CU)
                const uint32_t margin = 8u;
                if (n_chains <= 64u && cus > 2u * n_chains + margin) {   // (k_post_loop looks at all chains with one lane each)
                    const uint32_t room = (cus - 2u * n_chains - margin) / n_chains;
                    if (pb > room) pb = room;
                    if (const char* e = getenv("PG_POST_BLOCKS")) { const long v = strtol(e, nullptr, 0); if (v >= 1 && (uint32_t)v < pb) pb = (uint32_t)v; }   // (experiments: fewer)
                    if (pb >= 1u) { job->persist = true; job->post_blocks = pb; }
                }
            }
        }
    }

    // ---- the split path: which index contigs' chains take it (pg_device.h) ---------------------------
    for (auto& x : job->index) {
        const bool s1 = x.small && !job->chunked && x.cls4 && !kc.nosmall2;   // (DevContig::small == 2 below)
        const bool s2 = x.smallx && !job->chunked && !kc.nosmall2;            // (DevContig::smallx == 2)
        const bool ok = !kc.nosplit && !params->run_phasing && params->run_genotyping && x.n_lik < 0xFFFFFFF0ull && x.V > 0;
        x.split = !ok ? 0u : (s1 ? 1u : (s2 ? 2u : 0u));
        if (x.split) job->any_split = true;
    }

    // ---- plan the arena -----------------------------------------------------------------------
    job->chains.resize(n_chains);
    size_t off = 0;
    auto take = [&](size_t bytes, size_t al = 256) { off = align_up(off, al); size_t o = off; off += (bytes ? bytes : 8); return o; };
    struct Plan { size_t sync, wcols, aux, frec, scratch, wide, vpair, xbuf, prof, fback, fscale, bscale, bsum, vrec, colrec, fwd, part, cprec,
                  vtq, vback, vbest, hap1, hap2; };
    std::vector<Plan> plan(n_chains);
    const size_t o_contigs = take(sizeof(DevContig) * n_chains);
    const size_t o_reps = take(sizeof(DevContig) * n_index);
    const size_t o_contigs_g = take(sizeof(DevContig) * n_chains), o_reps_g = take(sizeof(DevContig) * n_index);   // (the pipelined first run: group order)
    const size_t o_small = take(sizeof(uint32_t) * n_chains);   // chain ids of the H = 16 chains (k_sweep_small16)
    const size_t o_dump = take(64 * 8 * 16 + 8 * 16 * 16 * 16);  // scrap column for the stores of its rows that are done
    const size_t o_handshake = take(64);   // k_stream_handshake's words
    // zeroed-every-run block: n_cols, err, per chain kept / fallback flags / profile counters / allele_present,
    // then the packed lik and lik_exp regions (chain after chain, no gaps: one range each for a gather)
    const size_t zero_lo = align_up(off);
    const size_t o_err = take(sizeof(uint32_t) * n_chains);
    uint64_t lik_total = 0;
    for (uint32_t c = 0; c < n_chains; ++c) {
        ChainHost& ch = job->chains[c];
        ch.index = specs[c].index;
        const IndexHost& x = job->index[ch.index];
        plan[c].fback = take(x.V);
        plan[c].prof = take(64 * sizeof(unsigned long long));
        plan[c].wcols = take(4);   // (the count of the chain's wide-column list: zeroed with the rest of this block; the list itself below)
        plan[c].sync = take(job->chunked ? PG_SYNC_WORDS * sizeof(uint32_t) : 0);   // (chunk hand-over words of the persistent phase 2)
        const bool vit = params->run_phasing != 0;
        plan[c].vbest = take(vit ? 4 : 0);
        plan[c].hap1 = take(vit ? (size_t)x.V * 2 : 0);
        plan[c].hap2 = take(vit ? (size_t)x.V * 2 : 0);
        ch.lik_first = lik_total;
        lik_total += x.n_lik;
    }
    job->n_lik_total = lik_total;
    const size_t o_lik = take(lik_total * sizeof(double));
    const size_t o_likexp = take(lik_total * sizeof(int32_t));
    const size_t zero_hi = off;
    job->o_tab_m = take(job->tab_m.size() * sizeof(double));
    job->o_tab_e = take(job->tab_e.size() * sizeof(int32_t));
    job->o_tab_p = take(job->tab_p.size());
    // what the index alone decides, formed by the index-level kernels whenever an index is uploaded (pg_split.h): one run of the
    // arena, zeroed before every such pass — column flags, present alleles, the column list and its inverse for every index
    // contig; for split contigs the per-column records and the list of wide columns
    const size_t ix_lo = align_up(off);
    const size_t o_ncols = take(sizeof(uint32_t) * n_index);
    const size_t o_ixerr = take(sizeof(uint32_t) * n_index);
    for (uint32_t i = 0; i < n_index; ++i) {
        IndexHost& x = job->index[i];
        x.o_kept = take(x.V); x.o_apres = take(x.sumA);
        x.o_colv = take((size_t)x.V * 4); x.o_colof = take((size_t)x.V * 4);
        x.o_ixpd = take(x.split ? (size_t)x.V * PG_IXPD_BYTES : 0);
        x.o_ixrec = take(x.split ? (size_t)x.V * PG_IXREC_BYTES : 0);
        x.o_ixbin = take(x.split ? (size_t)x.V * PG_IXBIN_BYTES : 0);
        x.o_ixwl = take(x.split == 2u && x.wide_bytes ? (size_t)x.n_wide_cand * 4 : 0);
        x.o_ixnw = take(4);
    }
    const size_t ix_hi = off > ix_lo ? off : ix_lo;
    for (uint32_t i = 0; i < n_index; ++i) {
        IndexHost& x = job->index[i];
        x.o_pos = take((size_t)x.V * 8);
        x.o_koff = take(((size_t)x.V + 1) * 4);
        x.o_aoff = take(((size_t)x.V + 1) * 4); x.o_aid = take((size_t)x.sumA * 2);
        x.o_aflag = take(x.sumA); x.o_akoff = take((size_t)x.sumA * 2); x.o_akmask = take((size_t)x.sumA * 4);
        x.o_pa = take((size_t)x.V * x.H * 2);
        x.o_goff = take(((size_t)x.V + 1) * 8);
        x.o_widx = take(x.wide_bytes ? (size_t)x.V * 4 : 0);
        x.o_auxidx = take(x.aux_bytes ? (size_t)x.V * 4 : 0);
        x.o_list_m4 = take(x.list_m4.size() * 4);
        x.o_list_w = take(x.list_w.size() * 4);
        x.o_list_b = take(x.list_b.size() * 4);
        x.o_list_big = take(x.list_big.size() * 4);
    }
    job->sample_lo = align_up(off);   // the per-sample arrays of all chains, one contiguous run (pg_job::sample_lo)
    for (uint32_t c = 0; c < n_chains; ++c) {
        ChainHost& ch = job->chains[c];
        const IndexHost& x = job->index[ch.index];
        ch.o_cov = take((size_t)x.V * 2); ch.o_kcnt = take((size_t)x.sumK * 2);
    }
    job->sample_bytes = align_up(off) - job->sample_lo;
    std::vector<char> tri_of_chain(n_chains, 0);
    std::vector<size_t> plan_wlist(n_chains, 0);
    // The variant records of all chains in ONE run of the arena, zeroed once when the job is built: k_prep_bi writes only the
    // pieces of a biallelic object's record that are not zero by construction (header, row bits, four table entries, the path
    // alleles: 96 of 384 bytes at 16 paths — it is a kernel that waits for its memory 70 % of the time); what it leaves out — the
    // rest of the 6 x 6 table, the padding — has to BE zero for every reader of the full record.
    const size_t vrec_lo = align_up(off);
    for (uint32_t c = 0; c < n_chains; ++c) { const IndexHost& x = job->index[job->chains[c].index]; plan[c].vrec = take(x.split ? 0 : (size_t)x.V * x.RB); }
    const size_t vrec_hi = off;
    // ... and the sample records of the split chains in another: zeroed with every index pass (k_prep_s_bi stores only the pieces
    // of a biallelic column's 128-byte record that are not zero by construction; which columns those are hangs on the index alone)
    const size_t srec_lo = align_up(off);
    for (uint32_t c = 0; c < n_chains; ++c) {
        const IndexHost& x = job->index[job->chains[c].index];
        if (x.split) plan[c].frec = take((size_t)x.V * (x.split == 1u ? PG_SREC1_BYTES : PG_SREC2_BYTES));
    }
    const size_t srec_hi = off > srec_lo ? off : srec_lo;   // (no split chain: an empty run)
    for (uint32_t c = 0; c < n_chains; ++c) {
        ChainHost& ch = job->chains[c];
        const IndexHost& x = job->index[ch.index];
        Plan& p = plan[c];
        p.colrec = take(x.split ? 0 : (size_t)x.V * x.RB);
        p.cprec = take(x.split ? (size_t)x.V * PG_CPREC_BYTES : 0);
        // fused jobs: lean chains store / read their columns as compact upper triangles (18 KB instead of 32 KB per
        // column: half the sweep's HBM bytes and of the arena); PG_KERNELS=notri keeps full columns (cross-check)
        // (round 6: and the 64-path chains with multiallelic objects — every column narrow — whose phase 1 runs on the general
        //  kernel: it stores the same triangles, phase 2 reads them through its triangle ring: DevContig::tri == 1)
        // (DevContig::widef chains too: their wide columns leave phase 2 through aux slots, k_bins_wide reads the stored triangle)
        const bool tri_multi = !x.lean && !x.leanx && x.HP == 64 && x.H == 64 && (!x.wide_bytes || x.widef) && x.pair_n > 2 && !force_generic && !kc.general;
        const bool tri = (x.lean || tri_multi) && !job->chunked && !kc.notri;
        tri_of_chain[c] = tri;
        const bool geno = params->run_genotyping != 0;  // (a phasing-only job has no sweep: no columns, no partials)
        p.fwd = take(geno ? (size_t)x.V * (tri ? 2304u : (size_t)x.HP * x.HP) * sizeof(double) : 0);
        // fused mode: posterior partials; chunked mode: the chunk scratch instead (k_post writes lik directly)
        // (x chains of fused jobs: four class sums per column — plus the per-thread partials of ONE column, should the chain
        //  be left with a single column and run its phase 2 on the general kernel)
        const bool x2 = x.smallx && !job->chunked && !kc.nosmall2;
        p.part = take(job->chunked || !geno ? 0 : (x2 ? (size_t)x.V * 32u + (size_t)x.part_slots * part_entries(x) * sizeof(double)
                                                      : (size_t)x.V * x.part_slots * part_entries(x) * sizeof(double)));
        p.aux = take((x2 || x.widef) && geno ? x.aux_bytes : 0);
        const size_t o_wlist = take((x2 || x.widef) && geno && x.wide_bytes && !x.split ? (size_t)x.n_wide_cand * 4 : 0);
        plan_wlist[c] = o_wlist;
        // Viterbi: transition probabilities and one 2-byte backpointer per state and column
        p.vtq = take(params->run_phasing ? (size_t)x.V * 8 * sizeof(double) : 0);
        p.vback = take(params->run_phasing ? (size_t)x.V * x.H * x.HP * sizeof(uint16_t) : 0);
        if (params->run_phasing) job->vit_bits |= x.HP == 16 ? 1u : (x.HP == 32 ? 2u : 4u);
        p.scratch = take(job->chunked ? (size_t)2 * PG_SCRATCH_BUFS * job->chunk_cols * x.HP * x.HP * sizeof(double) : 0);
        p.wide = take(x.wide_bytes);
        p.vpair = take(x.split ? 0 : (size_t)x.V * pg_pair_bytes(x.pair_n));
        p.xbuf = take((x.HP >= 256 || (force_generic && x.HP >= 64)) ? (size_t)2 * x.HP * x.HP * sizeof(double) : 0);
        if (!x.split) p.frec = take((x.lean || x.small) ? (size_t)x.V * 64 : (x.smallx ? (size_t)x.V * 192 : 0));
        if (x.lean) job->hp_mask |= 64u;
        if (x.leanx) job->hp_mask |= x.HP == 128 ? 512u : 1024u;
        p.fscale = take((size_t)x.V * sizeof(double));
        p.bscale = take((size_t)x.V * sizeof(double));
        p.bsum = take((size_t)x.V * sizeof(double));
        if (x.HP >= 256) job->hp_mask |= 16u;
        else if (force_generic && x.HP >= 64) job->hp_mask |= 32u;
        // (split chains never run on the general kernel; nor do, in a chunked job — no phase 2 —, the chains whose store-only phases
        //  have a kernel of their own: the general kernel's launch returned at once for them, fifty empty launches on the
        //  whole-genome job's phase 2)
        else if (!x.split && !(job->chunked && (x.lean || x.small || x.smallx || x.leanx)))
            job->hp_mask |= x.HP == 16 ? 1u : x.HP == 32 ? 2u : x.HP == 64 ? 4u : 8u;
    }
    if (job->hp_mask & 32u) job->hp_mask |= 16u;  // one generic launch covers both
    job->arena_bytes = align_up(off);
    const double t_alloc = now_s();
    if (cache_arena) g_pool.take(device, job->arena_bytes, &job->arena, &job->arena_bytes);
    if (!job->arena) {
        he = hipMalloc((void**)&job->arena, job->arena_bytes);
        if (he != hipSuccess) {
            (void)hipGetLastError();
            pg_hmm_release_cache();  // cached arenas may be what stands in the way
            hipSetDevice(device);
            he = hipMalloc((void**)&job->arena, job->arena_bytes);
        }
        if (he != hipSuccess) {
            job->arena = nullptr;
            (void)hipGetLastError();
            // A chunked job's scratch (PG_SCRATCH_BUFS buffers of chunk_cols columns per half-chain: up to 18 GB) is a choice, not
            // a need: before giving up, plan again with half the chunk columns (the fixed budget could fail a job
            // that fitted with smaller chunks — on a smaller device, or next to other resident jobs)
            if (job->chunked && job->chunk_cols > 64u) {
                const size_t next_cap = job->chunk_cols / 2u;
                pg_job_destroy(job);
                return job_build(device, n_index, batches, specs, n_samples, cohort, table, params, cache_arena, out, err, errlen, next_cap);
            }
            set_err(err, errlen, "hipMalloc(%zu bytes) failed: %s", job->arena_bytes, hipGetErrorString(he));
            pg_job_destroy(job);
            return PG_ERR_NOMEM;
        }
    }
    if ((he = hipMemsetAsync(job->arena + vrec_lo, 0, vrec_hi - vrec_lo, job->stream)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipMemset (variant records)", he);
    job->host_s[0] = now_s() - t_alloc;
    lap("arena plan + allocation");
    unsigned char* A = job->arena;
    job->d_contigs = (DevContig*)(A + o_contigs);
    job->d_reps = (DevContig*)(A + o_reps);
    job->d_contigs_g = (DevContig*)(A + o_contigs_g); job->d_reps_g = (DevContig*)(A + o_reps_g);
    job->d_ixerr = (uint32_t*)(A + o_ixerr);
    job->ix_base = A + ix_lo; job->ix_bytes = ix_hi - ix_lo;
    job->srec_base = A + srec_lo; job->srec_bytes = srec_hi - srec_lo;
    job->d_ncols = (uint32_t*)(A + o_ncols);
    job->d_err = (uint32_t*)(A + o_err);
    job->d_lik = (double*)(A + o_lik);
    job->d_likexp = (int32_t*)(A + o_likexp);
    job->zero_base = A + zero_lo;
    job->zero_bytes = zero_hi - zero_lo;
    job->tab.mant = (const double*)(A + job->o_tab_m);
    job->tab.expo = (const int32_t*)(A + job->o_tab_e);
    job->tab.packed = A + job->o_tab_p;

    // ---- chain descriptors ----------------------------------------------------------------------
    std::vector<DevContig> hd(n_chains);
    const long double dist_scale = 0.000004L * ((long double)params->recombrate) * params->effective_N;
    for (uint32_t c = 0; c < n_chains; ++c) {
        ChainHost& ch = job->chains[c];
        const IndexHost& x = job->index[ch.index];
        const Plan& p = plan[c];
        DevContig& d = hd[c];
        memset(&d, 0, sizeof(d));
        d.V = x.V; d.H = x.H; d.HP = x.HP; d.RB = x.RB; d.T = part_entries(x); d.part_slots = x.part_slots; d.pair_n = x.pair_n;
        d.dist_scale = (double)dist_scale; d.uniform = params->uniform ? 1 : 0;
        d.debug = 8u;   // (bit 3: the in-kernel cycle counters of -DPG_CHAIN_PROF builds; the product build has none)
        d.pos = (const uint64_t*)(A + x.o_pos); d.cov = (const uint16_t*)(A + ch.o_cov);
        d.kmer_off = (const uint32_t*)(A + x.o_koff); d.kmer_count = (const uint16_t*)(A + ch.o_kcnt);
        d.allele_off = (const uint32_t*)(A + x.o_aoff); d.allele_id = (const uint16_t*)(A + x.o_aid);
        d.allele_flags = (const uint8_t*)(A + x.o_aflag); d.allele_koff = (const uint16_t*)(A + x.o_akoff);
        d.allele_kmask = (const uint32_t*)(A + x.o_akmask); d.path_allele = (const uint16_t*)(A + x.o_pa);
        d.geno_off = (const uint64_t*)(A + x.o_goff);
        d.vrec = A + p.vrec; d.kept = A + x.o_kept; d.allele_present = A + x.o_apres;
        d.n_cols = job->d_ncols + ch.index; d.col_variant = (uint32_t*)(A + x.o_colv); d.colrec = A + p.colrec;
        d.col_of = (const uint32_t*)(A + x.o_colof); d.ix_err = job->d_ixerr + ch.index;
        d.ix_big = (const uint32_t*)(A + x.o_list_big); d.n_ix_big = (uint32_t)x.list_big.size();
        d.split = x.split;
        if (x.split) {
            d.ix_pd = A + x.o_ixpd; d.ix_rec = A + x.o_ixrec; d.ix_bin = A + x.o_ixbin; d.cprec = A + p.cprec;
        }
        d.fwd = (double*)(A + p.fwd); d.part = (double*)(A + p.part); d.fwd_fallback = A + p.fback; d.prof = (unsigned long long*)(A + p.prof);
        d.fscale = (double*)(A + p.fscale); d.bscale = (double*)(A + p.bscale); d.bsum = (double*)(A + p.bsum); d.err = job->d_err + c;
        d.lik = job->d_lik + ch.lik_first; d.lik_exp = job->d_likexp + ch.lik_first;
        d.scratch = (double*)(A + p.scratch); d.chunk_cols = job->chunk_cols; d.sync = (uint32_t*)(A + p.sync);
        d.wide = A + p.wide; d.wide_idx = x.wide_bytes ? (const uint32_t*)(A + x.o_widx) : nullptr;
        d.vpair = A + p.vpair; d.xbuf = (double*)(A + p.xbuf);
        d.frec = (double*)(A + p.frec); d.lean = x.lean ? 1u : 0u; d.small = x.small ? ((!job->chunked && x.cls4 && !kc.nosmall2) ? 2u : 1u) : 0u; d.leanx = x.leanx ? 1u : 0u; d.cls4 = x.cls4 ? 1u : 0u;
        d.smallx = x.smallx ? ((!job->chunked && !kc.nosmall2) ? 2u : 1u) : 0u;
        d.widef = x.widef ? 1u : 0u;
        d.aux = A + p.aux; d.aux_idx = ((d.smallx == 2u || x.widef) && x.aux_bytes) ? (const uint32_t*)(A + x.o_auxidx) : nullptr;
        if ((d.smallx == 2u || x.widef) && x.wide_bytes && x.n_wide_cand) {
            // (split chains: the wide columns hang on the index alone — one list per index contig, made by k_index_cols)
            if (x.split) { d.wcols = (uint32_t*)(A + x.o_ixwl); d.n_wcols = (uint32_t*)(A + x.o_ixnw); }
            else { d.wcols = (uint32_t*)(A + plan_wlist[c]); d.n_wcols = (uint32_t*)(A + p.wcols); }
            job->max_wide = std::max(job->max_wide, x.n_wide_cand);
        }
        d.live = (!job->chunked && x.HP == 32u && !kc.fullcols) ? std::min<uint32_t>(x.HP, (x.H + 3u) & ~3u) : x.HP;
        d.prep_fast = x.prep_fast;
        if (x.split) {
            // the split path's emission kernels: lists unless every object is k_prep_s_bi's
            if (!x.all_sb) {
                d.prep_m4 = (const uint32_t*)(A + x.o_list_m4); d.n_prep_m4 = (uint32_t)x.list_m4.size();
                d.prep_w = (const uint32_t*)(A + x.o_list_w); d.n_prep_w = (uint32_t)x.list_w.size();
                d.prep_b = (const uint32_t*)(A + x.o_list_b); d.n_prep_b = (uint32_t)x.list_b.size();
            }
            job->max_sb = std::max(job->max_sb, x.all_sb ? x.V : d.n_prep_b);
            job->max_sm4 = std::max(job->max_sm4, d.n_prep_m4);
            job->max_sw = std::max(job->max_sw, d.n_prep_w);
        } else {
        job->any_legacy_prep = true;
        if (x.prep_fast == 2u) {   // (a non-null list pointer = "walk the list", also when it is empty)
            d.prep_m4 = (const uint32_t*)(A + x.o_list_m4); d.n_prep_m4 = (uint32_t)x.list_m4.size();
            d.prep_w = (const uint32_t*)(A + x.o_list_w); d.n_prep_w = (uint32_t)x.list_w.size();
            d.prep_b = (const uint32_t*)(A + x.o_list_b); d.n_prep_b = (uint32_t)x.list_b.size();
            job->max_prep_m4 = std::max(job->max_prep_m4, d.n_prep_m4);
            job->max_prep_w = std::max(job->max_prep_w, d.n_prep_w);
        } else if (x.prep_fast == 0u) job->max_prep_w = std::max(job->max_prep_w, x.V);
        }
        if (params->run_phasing) {
            d.vit_tq = (double*)(A + p.vtq); d.vit_back = (uint16_t*)(A + p.vback); d.vit_best = (uint32_t*)(A + p.vbest);
            d.hap1 = (uint16_t*)(A + p.hap1); d.hap2 = (uint16_t*)(A + p.hap2);
        }
        d.tri = tri_of_chain[c] ? ((kc.nolean2 || !x.lean) ? 1u : 2u) : 0u;   // (PG_KERNELS=nolean2, and chains with multiallelic objects: phase 2 of triangle chains on the general kernel's triangle ring)
        d.col_stride = d.tri ? 2304u : x.HP * x.HP;
        if (d.tri == 1u && !x.lean && x.HP == 64u && !kc.noleanx2 && !params->run_phasing && !x.widef) {   // phase 2 on k_sweep_leanx2 (no wide columns there)
            d.leanx2 = 1u; d.T = 64u; job->hp_mask |= 8192u;
        }
        if (d.tri) job->hp_mask |= 128u;
        if (d.tri && !x.lean) {
            // phase 1 of such chains: the lean-x step with triangle stores (DevContig::leanx == 2; it needs the column-order records
            // of the general kernel, which phase 2 reads too) — PG_KERNELS=noleanx: the general kernel with triangle stores
            if (kc.leanx != 0) { d.leanx = 2u; job->hp_mask |= x.widef ? 16384u : 4096u; }   // (DevContig::widef chains: k_sweep_leanx_triw — leanx_forward / _backward's wide_fix)
            else job->hp_mask |= 2048u;
        }
        if (d.tri == 2u) job->hp_mask |= 256u;
        // (k_bins_thin: what bins_thin() in pg_kernels.hip says — at most 64 partial entries per column, fused job)
        if (x.split) job->bins_which |= 32u | ((x.split == 2u && x.wide_bytes) ? 64u : 0u);   // k_bins_s, k_bins_wide_s
        else {
        job->bins_which |= d.leanx2 ? 128u : (d.tri == 2u || d.cls4) ? 2u : ((d.T <= 64u && d.HP <= 32u && !job->chunked) ? 4u : 1u);   // (bit 7: k_bins_q)
        if (d.smallx == 2u) job->bins_which |= 8u | (x.wide_bytes ? 16u : 0u);   // k_bins_x, k_bins_wide (a chain left with one column: k_bins_thin, above)
        if (x.widef && x.n_wide_cand) job->bins_which |= 16u;   // k_bins_wide for the wide columns of DevContig::widef chains
        }
        ch.d = d;
    }
    {
        // chain ids of the four-half-chains-per-wave kernels: the all-biallelic chains first, then the x chains; each list in the
        // order of the index contigs (stable), so that the four rows of a wave are — in a cohort — samples of ONE contig: the same
        // columns are multiallelic / wide in all four (the wide branch of k_sweep_small16x is wave-uniform)
        std::vector<uint32_t> small_ids, x_ids;
        for (uint32_t c = 0; c < n_chains; ++c) {
            if (hd[c].small) { small_ids.push_back(c); if (hd[c].small == 2u) job->small_phase2 = true; }
            if (hd[c].smallx) { x_ids.push_back(c); if (hd[c].smallx == 2u) job->smallx_phase2 = true; }
        }
        auto by_index = [&](uint32_t a, uint32_t b) { return job->chains[a].index < job->chains[b].index; };
        std::stable_sort(small_ids.begin(), small_ids.end(), by_index);
        std::stable_sort(x_ids.begin(), x_ids.end(), by_index);
        job->n_small = (uint32_t)small_ids.size();
        job->n_smallx = (uint32_t)x_ids.size();
        job->d_small = (uint32_t*)(A + o_small);
        job->d_smallx = job->d_small + job->n_small;
        job->d_dump = (double*)(A + o_dump);
        job->d_handshake = (uint32_t*)(A + o_handshake);
        small_ids.insert(small_ids.end(), x_ids.begin(), x_ids.end());
        if (!small_ids.empty() && (he = hipMemcpyAsync(job->d_small, small_ids.data(), sizeof(uint32_t) * small_ids.size(), hipMemcpyHostToDevice, job->stream)) != hipSuccess)
            return fail(PG_ERR_DEVICE, "hipMemcpy small ids", he);
        if (!small_ids.empty() && (he = hipStreamSynchronize(job->stream)) != hipSuccess) return fail(PG_ERR_DEVICE, "hipMemcpy small ids", he);
    }
    {   // the plan as text (pg_job_plan): one line per group of chains with the same kernels
        struct Group { std::string text; uint32_t count = 0; };
        std::vector<Group> groups;
        for (uint32_t c = 0; c < n_chains; ++c) {
            const DevContig& d = hd[c];
            const IndexHost& x = job->index[job->chains[c].index];
            if (x.V == 0) continue;
            const bool multi = x.pair_n > 2, wide = x.wide_bytes != 0;
            std::string prep, p1, p2, bins;
            const char* gen = d.HP >= 256 || force_generic ? "k_sweep_generic" : (d.HP == 16 ? "k_sweep<16,4>" : d.HP == 32 ? "k_sweep<32,8>" : d.HP == 64 ? "k_sweep<64,16>" : "k_sweep<128,32>");
            if (d.split) {
                prep = std::string("index pass once (k_index_scan, k_compact, k_index_cols); per run k_prep_s_bi") + (x.all_sb ? "" : " + k_prep_s_m4 + k_prep_s_w");
                p1 = d.split == 1u ? "k_sweep_small16<1>" : "k_sweep_small16x<1>";
                p2 = d.split == 1u ? "k_sweep_small16<2>" : "k_sweep_small16x<2>";
                bins = std::string("k_bins_s") + (wide ? " + k_bins_wide_s" : "");
            } else {
                prep = "index pass once (k_index_scan, k_compact); per run ";
                prep += x.prep_fast == 1u ? "k_prep_bi" : (x.prep_fast == 2u ? "k_prep_bi + k_prep_m4 + k_prep" : "k_prep");
                prep += " + k_records";
                p1 = d.lean ? (d.tri ? "k_sweep_lean_tri<1>" : "k_sweep_lean<1>") : d.leanx == 2u ? (d.widef ? "k_sweep_leanx_triw" : "k_sweep_leanx_tri") : d.leanx ? "k_sweep_leanx<1>"
                     : d.small ? "k_sweep_small16<1>" : d.smallx ? "k_sweep_small16x<1>" : (d.tri ? "k_sweep_tri1" : std::string(gen) + "<1>");
                if (job->chunked && job->persist) {
                    p2 = "k_sweep_lean<4> (one launch, all chunks) + k_post_loop";
                    bins = "(k_post_loop)";
                } else if (job->chunked) {
                    p2 = d.lean ? "k_sweep_lean<3>" : d.leanx ? "k_sweep_leanx<3>" : d.small ? "k_sweep_small16<3>" : d.smallx ? "k_sweep_small16x<3>" : std::string(gen) + "<3>";
                    p2 += " chunks + k_post";
                    bins = "(k_post)";
                } else {
                    p2 = d.leanx2 ? "k_sweep_leanx2" : d.tri == 2u ? "k_sweep_lean2" : d.small == 2u ? "k_sweep_small16<2>" : d.smallx == 2u ? "k_sweep_small16x<2>"
                         : std::string(gen) + (d.tri ? "<2> (triangle ring)" : "<2>");
                    bins = d.leanx2 ? "k_bins_q" : (d.tri == 2u || d.cls4) ? "k_bins_lean2" : d.smallx == 2u ? (wide ? "k_bins_x + k_bins_wide" : "k_bins_x")
                           : (d.T <= 64u && d.HP <= 32u) ? "k_bins_thin" : "k_bins";
                    if (d.widef) { p2 += " (wide columns to their aux slots)"; bins += " + k_bins_wide"; }
                }
            }
            char head[160];
            snprintf(head, sizeof(head), "%u path(s) (padded %u), %s%s%s columns: ", d.H, d.HP, multi ? "multiallelic" : "biallelic", wide ? " + wide" : "",
                     d.tri ? ", triangle" : "");
            std::string text = std::string(head) + "prep = " + prep + "; phase 1 = " + p1 + "; phase 2 = " + p2 + "; bins = " + bins;
            bool found = false;
            for (auto& g : groups) if (g.text == text) { g.count += 1; found = true; break; }
            if (!found) groups.push_back({text, 1});
        }
        job->plan_text = std::string(job->chunked ? "chunked" : "fused") + " job, " + std::to_string(n_chains) + " chain(s)" + (params->run_phasing ? ", run_phasing (k_viterbi)" : "") + "\n";
        for (const auto& g : groups) job->plan_text += "  " + std::to_string(g.count) + " x " + g.text + "\n";
    }
    job->h_contigs = hd;
    job->cur_samples = A + job->sample_lo;
    {   // one descriptor per index contig for the index-level kernels (any chain over it: they touch index-level fields only)
        std::vector<DevContig> reps(n_index);
        for (auto& r : reps) memset(&r, 0, sizeof(r));
        std::vector<char> have(n_index, 0);
        for (uint32_t c = 0; c < n_chains; ++c) { const uint32_t i = job->chains[c].index; if (!have[i]) { have[i] = 1; reps[i] = hd[c]; } }
        if ((he = hipMemcpyAsync(job->d_reps, reps.data(), sizeof(DevContig) * n_index, hipMemcpyHostToDevice, job->stream)) != hipSuccess ||
            (he = hipStreamSynchronize(job->stream)) != hipSuccess)
            return fail(PG_ERR_DEVICE, "hipMemcpy index descriptors", he);
    }
    if ((he = hipMemcpyAsync(job->d_contigs, hd.data(), sizeof(DevContig) * n_chains, hipMemcpyHostToDevice, job->stream)) != hipSuccess ||
        (he = hipStreamSynchronize(job->stream)) != hipSuccess)
        return fail(PG_ERR_DEVICE, "hipMemcpy contigs", he);
    lap("descriptors");
    {
        // The pipelined first run (pg_job::pipeline): a one-shot job — upload and run inside one call — of few chains (chunked: bound
        // by its longest chain, not by throughput), every chain over an index contig of its own, on kernels that take a sub-range
        // of the descriptor array (no chain-id lists, no split path), with enough bytes to upload that hiding them matters.
        // Group A = the chains with at least 0.78 of the longest one's variants (they decide the wall time; the others, starting
        // later by their upload — ~10 % of a run at 40 GB/s —, still end phase 1 first), unless that is most of the bytes anyway.
        bool ok = job->chunked && !cohort && n_index == n_chains && n_chains >= 2 && !job->any_split &&
                  params->run_genotyping && !params->run_phasing && getenv("PG_NO_PIPELINE") == nullptr;
        for (uint32_t c = 0; ok && c < n_chains; ++c) ok = job->chains[c].index == c && !job->index[c].small && !job->index[c].smallx;
        if (ok) {
            uint64_t vmax = 0, wA = 0, wAll = 0;
            for (const IndexHost& x : job->index) vmax = std::max<uint64_t>(vmax, x.V);
            std::vector<uint32_t> a, b;
            for (uint32_t c = 0; c < n_chains; ++c) {
                const IndexHost& x = job->index[c];
                const uint64_t w = (uint64_t)x.V * x.H * 2u + (uint64_t)x.sumK * 2u + (uint64_t)x.V * 24u;
                wAll += w;
                if ((double)x.V >= 0.78 * (double)vmax) { a.push_back(c); wA += w; } else b.push_back(c);
            }
            uint64_t min_bytes = (uint64_t)128 << 20;   // (PG_PIPELINE_MIN_MB: tests put small jobs through the pipelined run)
            if (const char* e = getenv("PG_PIPELINE_MIN_MB")) min_bytes = (uint64_t)strtoul(e, nullptr, 0) << 20;
            if (!b.empty() && wAll >= min_bytes && (double)wA <= 0.6 * (double)wAll) {
                job->grp_order = a;
                job->grp_order.insert(job->grp_order.end(), b.begin(), b.end());
                job->nA = (uint32_t)a.size();
                std::vector<DevContig> g(n_chains);
                for (uint32_t k = 0; k < n_chains; ++k) g[k] = hd[job->grp_order[k]];
                // (chain c over index c: the chain's descriptor serves as its index contig's)
                if ((he = hipMemcpyAsync(job->d_contigs_g, g.data(), sizeof(DevContig) * n_chains, hipMemcpyHostToDevice, job->stream)) != hipSuccess ||
                    (he = hipMemcpyAsync(job->d_reps_g, g.data(), sizeof(DevContig) * n_chains, hipMemcpyHostToDevice, job->stream)) != hipSuccess ||
                    (he = hipStreamSynchronize(job->stream)) != hipSuccess)
                    return fail(PG_ERR_DEVICE, "hipMemcpy group descriptors", he);
                if (hipEventCreateWithFlags(&job->ev_late[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&job->ev_late[1], hipEventDisableTiming) == hipSuccess) {
                    job->ev_late_made = true;
                    job->pipeline_capable = true;
                    job->pipeline = cache_arena;   // (a resident job's first upload returns to a caller who may free the arrays: pg_job_upload_run asks for it)
                }
            }
        }
    }
    const int rc = upload_inputs(job, batches, specs, true, err, errlen);
    if (rc != PG_OK) { pg_job_destroy(job); return rc; }
    lap("upload of the inputs");
    *out = job;
    return PG_OK;
}

}  // namespace

extern "C" int pg_job_new(int device, uint32_t n_contigs, const pg_contig_batch* batches, const pg_table* table,
                          const pg_hmm_params* params, pg_job** out, char* err, size_t errlen) {
    if (!out) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    *out = nullptr;
    if (!batches || n_contigs == 0) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    std::vector<ChainSpec> specs(n_contigs);
    for (uint32_t i = 0; i < n_contigs; ++i) specs[i] = {i, batches[i].kmer_count, batches[i].coverage};
    return job_build(device, n_contigs, batches, specs, 1, false, table, params, false, out, err, errlen);
}

extern "C" pg_job* pg_job_create(int device, uint32_t n_contigs, const pg_contig_batch* batches,
                                 const pg_table* table, const pg_hmm_params* params, char* err, size_t errlen) {
    pg_job* job = nullptr;
    pg_job_new(device, n_contigs, batches, table, params, &job, err, errlen);
    return job;
}

extern "C" int pg_cohort_new(int device, uint32_t n_contigs, const pg_contig_batch* index, uint32_t n_samples,
                             const pg_sample_counts* samples, const pg_table* table, const pg_hmm_params* params,
                             pg_job** out, char* err, size_t errlen) {
    if (!out) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    *out = nullptr;
    if (!index || !samples || n_contigs == 0 || n_samples == 0) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    std::vector<ChainSpec> specs((size_t)n_samples * n_contigs);
    for (uint32_t s = 0; s < n_samples; ++s) {
        if (!samples[s].kmer_count || !samples[s].coverage) { set_err(err, errlen, "sample %u has null arrays", s); return PG_ERR_INVALID; }
        for (uint32_t c = 0; c < n_contigs; ++c)
            specs[(size_t)s * n_contigs + c] = {c, samples[s].kmer_count[c], samples[s].coverage[c]};
    }
    return job_build(device, n_contigs, index, specs, n_samples, true, table, params, false, out, err, errlen);
}

extern "C" int pg_job_upload_run(pg_job* job, const pg_contig_batch* batches, const pg_sample_counts* samples, char* err, size_t errlen) {
    if (!job) { set_err(err, errlen, "null job"); return PG_ERR_INVALID; }
    // upload + run inside ONE call: the host arrays stay valid throughout, so the upload of the shorter chains may hide behind
    // the long chains' phase 1 (pg_job::pipeline) where the job is capable of it; otherwise exactly pg_job_upload, pg_job_run
    job->pipeline = job->pipeline_capable && batches != nullptr && getenv("PG_NO_PIPELINE") == nullptr;
    int rc = pg_job_upload(job, batches, samples, err, errlen);
    job->pipeline = false;
    if (rc == PG_OK) rc = pg_job_run(job, nullptr, err, errlen);
    return rc;
}

extern "C" int pg_job_upload(pg_job* job, const pg_contig_batch* batches, const pg_sample_counts* samples, char* err, size_t errlen) {
    if (!job) { set_err(err, errlen, "null job"); return PG_ERR_INVALID; }
    if (job->upload_pending) { set_err(err, errlen, "an asynchronous upload is in flight: call pg_job_upload_end first"); return PG_ERR_INVALID; }
    HIP_TRY(hipSetDevice(job->device));
    const uint32_t n = (uint32_t)job->chains.size();
    std::vector<ChainSpec> specs(n);
    if (batches)  // the layout must be the one the arena, the genotype / wide offsets and the kernel choice were planned for:
        for (uint32_t i = 0; i < job->n_contigs; ++i) {  // per-variant K and A, not just their totals
            const IndexHost& x = job->index[i];
            const int rc = check_batch(&batches[i], !job->cohort, err, errlen);
            if (rc != PG_OK) return rc;
            if (batches[i].n_variants != x.V || batches[i].n_paths != x.H ||
                (x.V && (memcmp(batches[i].kmer_off, x.koff.data(), ((size_t)x.V + 1) * 4) != 0 ||
                         memcmp(batches[i].allele_off, x.aoff.data(), ((size_t)x.V + 1) * 4) != 0))) {
                set_err(err, errlen, "contig %u: shape (variants, paths, k-mers or alleles per variant) differs from the resident job", i);
                return PG_ERR_INVALID;
            }
        }
    if (job->cohort) {
        if (!samples) { set_err(err, errlen, "cohort job: samples must be given"); return PG_ERR_INVALID; }
        for (uint32_t s = 0; s < job->n_samples; ++s) {
            if (!samples[s].kmer_count || !samples[s].coverage) { set_err(err, errlen, "sample %u has null arrays", s); return PG_ERR_INVALID; }
            for (uint32_t c = 0; c < job->n_contigs; ++c)
                specs[(size_t)s * job->n_contigs + c] = {c, samples[s].kmer_count[c], samples[s].coverage[c]};
        }
    } else {
        if (!batches) { set_err(err, errlen, "batches must be given"); return PG_ERR_INVALID; }
        for (uint32_t i = 0; i < n; ++i) specs[i] = {i, batches[i].kmer_count, batches[i].coverage};
    }
    for (uint32_t c = 0; c < n; ++c) {
        const IndexHost& x = job->index[job->chains[c].index];
        if (x.V > 0 && (!specs[c].coverage || (x.sumK > 0 && !specs[c].kmer_count))) {
            set_err(err, errlen, "chain %u has null kmer_count / coverage arrays", c);
            return PG_ERR_INVALID;
        }
    }
    return upload_inputs(job, batches, specs, batches != nullptr, err, errlen);
}

namespace {
int cohort_specs(pg_job* job, const pg_sample_counts* samples, std::vector<ChainSpec>& specs, char* err, size_t errlen) {
    if (!job->cohort) { set_err(err, errlen, "not a cohort job"); return PG_ERR_INVALID; }
    if (!samples) { set_err(err, errlen, "cohort job: samples must be given"); return PG_ERR_INVALID; }
    specs.resize(job->chains.size());
    for (uint32_t s = 0; s < job->n_samples; ++s) {
        if (!samples[s].kmer_count || !samples[s].coverage) { set_err(err, errlen, "sample %u has null arrays", s); return PG_ERR_INVALID; }
        for (uint32_t c = 0; c < job->n_contigs; ++c)
            specs[(size_t)s * job->n_contigs + c] = {c, samples[s].kmer_count[c], samples[s].coverage[c]};
    }
    for (size_t c = 0; c < specs.size(); ++c) {
        const IndexHost& x = job->index[job->chains[c].index];
        if (x.V > 0 && (!specs[c].coverage || (x.sumK > 0 && !specs[c].kmer_count))) {
            set_err(err, errlen, "chain %zu has null kmer_count / coverage arrays", c);
            return PG_ERR_INVALID;
        }
    }
    return PG_OK;
}
}  // namespace

// The next batch of samples of a cohort job, uploaded WHILE the current one is being genotyped: a worker thread packs
// the counts into the pinned staging buffer and copies them into the job's second set of per-sample arrays on a copy
// stream of its own; pg_job_upload_end waits for it and makes that set the one pg_job_run reads.  The caller's arrays
// must stay valid (and unchanged) until pg_job_upload_end returns.
extern "C" int pg_job_upload_begin(pg_job* job, const pg_sample_counts* samples, char* err, size_t errlen) {
    if (!job) { set_err(err, errlen, "null job"); return PG_ERR_INVALID; }
    if (job->upload_pending) { set_err(err, errlen, "an upload is already in flight: call pg_job_upload_end first"); return PG_ERR_INVALID; }
    HIP_TRY(hipSetDevice(job->device));
    auto specs = std::make_shared<std::vector<ChainSpec>>();
    const int rc = cohort_specs(job, samples, *specs, err, errlen);
    if (rc != PG_OK) return rc;
    const size_t n = job->chains.size();
    if (!job->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&job->copy_stream, hipStreamNonBlocking));
    if (!job->alt_samples) {   // the second set + descriptors that point into it
        hipError_t he = hipMalloc((void**)&job->alt_samples, job->sample_bytes ? job->sample_bytes : 8);
        if (he != hipSuccess) {   // cached arenas of finished one-shot calls may be what stands in the way (as in job_build)
            (void)hipGetLastError();
            job->alt_samples = nullptr;
            pg_hmm_release_cache();
            hipSetDevice(job->device);
            he = hipMalloc((void**)&job->alt_samples, job->sample_bytes ? job->sample_bytes : 8);
        }
        if (he == hipSuccess) he = hipMalloc((void**)&job->d_contigs_owned, sizeof(DevContig) * n);
        job->d_contigs_alt = job->d_contigs_owned;
        if (he != hipSuccess) {
            if (job->alt_samples) { hipFree(job->alt_samples); job->alt_samples = nullptr; }
            job->d_contigs_alt = job->d_contigs_owned = nullptr;
            set_err(err, errlen, "hipMalloc (second sample set, %zu bytes) failed: %s", job->sample_bytes, hipGetErrorString(he));
            return PG_ERR_NOMEM;
        }
    }
    unsigned char* other = job->cur_samples == job->alt_samples ? job->arena + job->sample_lo : job->alt_samples;
    DevContig* d_other = job->d_contigs_alt;   // (the spare descriptor array: pg_job_upload_end swaps the two)
    job->next_coverage.assign(n, {});
    job->upload_pending = true;
    job->upload_rc = PG_OK;
    job->upload_err.clear();
    if (job->uploader.joinable()) job->uploader.join();
    job->uploader = std::thread([job, specs, other, d_other, n]() {
        char e[256] = {0};
        uint64_t bytes = 0;
        int r = hipSetDevice(job->device) == hipSuccess ? PG_OK : PG_ERR_DEVICE;
        if (r == PG_OK) r = sample_upload(job, *specs, other, job->copy_stream, &job->next_coverage, &bytes, e, sizeof(e));
        if (r == PG_OK) {   // the descriptors of that set (cov / kmer_count rebased), next to it on the device
            std::vector<DevContig> hd = job->h_contigs;
            for (size_t c = 0; c < n; ++c) {
                hd[c].cov = (const uint16_t*)(other + (job->chains[c].o_cov - job->sample_lo));
                hd[c].kmer_count = (const uint16_t*)(other + (job->chains[c].o_kcnt - job->sample_lo));
            }
            if (hipMemcpyAsync(d_other, hd.data(), sizeof(DevContig) * n, hipMemcpyHostToDevice, job->copy_stream) != hipSuccess ||
                hipStreamSynchronize(job->copy_stream) != hipSuccess) { r = PG_ERR_DEVICE; snprintf(e, sizeof(e), "descriptor upload failed"); }
            else job->h_contigs.swap(hd);
        }
        job->up_bytes[0] = 0; job->up_bytes[1] = bytes;
        job->upload_rc = r;
        job->upload_err = e;
    });
    return PG_OK;
}

extern "C" int pg_job_upload_end(pg_job* job, char* err, size_t errlen) {
    if (!job) { set_err(err, errlen, "null job"); return PG_ERR_INVALID; }
    if (!job->upload_pending) { set_err(err, errlen, "no upload in flight"); return PG_ERR_INVALID; }
    const double t0 = now_s();
    if (job->uploader.joinable()) job->uploader.join();
    job->upload_pending = false;
    job->host_s[1] = now_s() - t0;   // (what the caller waited: 0 when the upload hid behind the run)
    if (job->upload_rc != PG_OK) { set_err(err, errlen, "%s", job->upload_err.c_str()); return job->upload_rc; }
    // the freshly filled set becomes the current one: its descriptors sit in d_contigs_alt — swap the two descriptor arrays
    std::swap(job->d_contigs, job->d_contigs_alt);
    job->cur_samples = job->cur_samples == job->alt_samples ? job->arena + job->sample_lo : job->alt_samples;
    for (size_t c = 0; c < job->chains.size(); ++c) {
        job->chains[c].coverage.swap(job->next_coverage[c]);
        job->chains[c].d = job->h_contigs[c];
    }
    job->ran = false;
    return PG_OK;
}

namespace {
// Transition probabilities of the Viterbi, per kept column, formed on the host in long double exactly as the
// reference forms them (TransitionProbabilityComputer, src/transitionprobabilitycomputer.cpp:8-19, :33-39) and split
// into exact (hi, lo) double pairs: {t0, t1, t2} = {p^2, pq, q^2}.  The reference's decisions hang on p/q - 1, which
// drops below fp64's resolution once distance / H > 37 (pg_viterbi.hip).  Column 0 has no predecessor.
int viterbi_transitions(pg_job* job, hipStream_t s, char* err, size_t errlen) {
    const uint32_t n = (uint32_t)job->chains.size();
    std::vector<std::vector<double>> tq_of_index(job->index.size());
    std::vector<char> done(job->index.size(), 0);
    std::vector<uint32_t> colv;
    for (uint32_t c = 0; c < n; ++c) {
        const ChainHost& ch = job->chains[c];
        const IndexHost& x = job->index[ch.index];
        if (x.V == 0) continue;
        std::vector<double>& tq = tq_of_index[ch.index];
        if (!done[ch.index]) {  // (the kept columns depend on the index alone: once per index contig)
            done[ch.index] = 1;
            uint32_t C = 0;
            HIP_TRY(hipMemcpy(&C, ch.d.n_cols, sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (C > x.V) { set_err(err, errlen, "chain %u: bad column count", c); return PG_ERR_DEVICE; }
            colv.resize(C ? C : 1);
            if (C) HIP_TRY(hipMemcpy(colv.data(), ch.d.col_variant, (size_t)C * sizeof(uint32_t), hipMemcpyDeviceToHost));
            tq.assign((size_t)C * 8, 0.0);
            const long double H = (long double)x.H;
            for (uint32_t k = 0; k < C; ++k) {
                long double t[3] = {1.0L, 1.0L, 1.0L};
                if (k > 0 && !job->params.uniform) {
                    const long double distance = (x.pos[colv[k]] - x.pos[colv[k - 1]]) * 0.000004L * ((long double)job->params.recombrate) * job->params.effective_N;
                    const long double recomb_prob = (1.0L - expl(-distance / H)) * (1.0L / H);
                    const long double no_recomb_prob = expl(-distance / H) + recomb_prob;
                    t[0] = no_recomb_prob * no_recomb_prob; t[1] = no_recomb_prob * recomb_prob; t[2] = recomb_prob * recomb_prob;
                }
                for (int q = 0; q < 3; ++q) {
                    const double hi = (double)t[q];
                    tq[(size_t)k * 8 + 2 * q] = hi;
                    tq[(size_t)k * 8 + 2 * q + 1] = (double)(t[q] - (long double)hi);  // exact: 64 - 53 bits are left
                }
            }
        }
        if (!tq.empty()) HIP_TRY(hipMemcpyAsync(ch.d.vit_tq, tq.data(), tq.size() * sizeof(double), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipStreamSynchronize(s));  // (tq_of_index goes out of scope)
    return PG_OK;
}
}  // namespace

namespace {
// The persistent phase 2 needs the kernels of `s` and of the job's second stream to run SIDE BY SIDE (they wait for each other).
// Streams are mapped onto a few hardware queues, and two streams on one queue run one after the other: ask the device once per
// stream (k_stream_handshake); a second stream that shares the queue is replaced by a new one, a few times.  False: this run
// takes a launch per chunk.
bool streams_concurrent(pg_job* job, hipStream_t s) {
    if (job->persist_checked_any && job->persist_checked == s) return true;
    for (int attempt = 0; attempt < 6; ++attempt) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (hipMemcpyAsync(job->d_handshake, w, sizeof(w), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;
        pgk_launch_stream_handshake(job->d_handshake, s, job->stream2);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipStreamSynchronize(job->stream2) != hipSuccess) break;
        if (hipMemcpy(w, job->d_handshake, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) break;
        if (w[2] == 1u) { job->persist_checked = s; job->persist_checked_any = true; return true; }
        hipStream_t fresh = nullptr;   // (created BEFORE the old one is destroyed: the runtime hands a new stream the least used queue)
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
        (void)hipStreamDestroy(job->stream2);
        job->stream2 = fresh;
    }
    (void)hipGetLastError();
    return false;
}
}  // namespace

extern "C" int pg_job_run(pg_job* job, void* stream_, char* err, size_t errlen) {
    if (!job) { set_err(err, errlen, "null job"); return PG_ERR_INVALID; }
    const double t_run = now_s();
    HIP_TRY(hipSetDevice(job->device));
    hipStream_t s = stream_ ? (hipStream_t)stream_ : job->stream;
    const uint32_t n = (uint32_t)job->chains.size();
    job->host_s[3] = 0.0;
    const bool persist_now = job->persist && job->max_v > 0 && job->params.run_genotyping && streams_concurrent(job, s);   // (may replace stream2)
    if (job->persist && job->max_v > 0 && job->params.run_genotyping) { if (persist_now) job->persist_runs += 1; else job->persist_fallbacks += 1; }
    HIP_TRY(hipMemsetAsync(job->zero_base, 0, job->zero_bytes, s));
    if (job->max_v > 0 && job->params.run_genotyping) {
        HIP_TRY(hipEventRecord(job->ev[0], s));
        if (job->late_pending && s == job->stream) {
            // The pipelined first run (pg_job::pipeline): group A's preparation and phase 1 start NOW, on `s`; the host then waits for
            // group B's inputs (they have been crossing PCIe since job_build) and puts B's index pass, preparation and phase 1 on the
            // second stream, beside A's.  Phase 2 (below) starts when both are done.  (The timing classes: B's preparation counts
            // into the phase-1 class.)
            const uint32_t nA = job->nA, nB = n - nA;
            HIP_TRY(hipEventRecord(job->ev_late[0], s));   // (behind the zeroing of the per-run block)
            pgk_launch_prep(job->d_contigs_g, nA, job->max_v, job->max_prep_w, job->max_prep_m4, job->tab, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(job->ev[1], s));
            HIP_TRY(hipEventRecord(job->ev[2], s));
            pgk_launch_records(job->d_contigs_g, nA, job->max_v, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(job->ev[3], s));
            pgk_launch_sweep(job->d_contigs_g, nA, job->hp_mask, 1, s);
            HIP_TRY(hipGetLastError());
            for (auto& t : job->late_threads) if (t.joinable()) t.join();   // (their copies are ISSUED — 17 ms on the whole genome —, on stream2)
            job->late_threads.clear();
            job->late_pending = false;
            for (int rcb : job->late_rc)
                if (rcb != (int)hipSuccess) { set_err(err, errlen, "input upload: %s", hipGetErrorString((hipError_t)rcb)); return PG_ERR_DEVICE; }
            hipStream_t sb = job->stream2;
            HIP_TRY(hipStreamWaitEvent(sb, job->ev_late[0], 0));
            uint32_t max_big = 0;
            for (const IndexHost& x : job->index) max_big = std::max<uint32_t>(max_big, (uint32_t)x.list_big.size());
            pgk_launch_index(job->d_reps_g + nA, nB, job->max_v, max_big, 0, sb);
            pgk_launch_prep(job->d_contigs_g + nA, nB, job->max_v, job->max_prep_w, job->max_prep_m4, job->tab, sb);
            pgk_launch_records(job->d_contigs_g + nA, nB, job->max_v, sb);
            pgk_launch_sweep(job->d_contigs_g + nA, nB, job->hp_mask, 1, sb);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(job->ev_late[1], sb));
            HIP_TRY(hipStreamWaitEvent(s, job->ev_late[1], 0));
            HIP_TRY(hipEventRecord(job->ev[4], s));
        } else {
        if (job->late_pending) {
            // (a caller's own stream: no pipelining — group B's inputs are waited for, its index pass made up, then the run as ever)
            for (auto& t : job->late_threads) if (t.joinable()) t.join();
            job->late_threads.clear();
            job->late_pending = false;
            for (int rcb : job->late_rc)
                if (rcb != (int)hipSuccess) { set_err(err, errlen, "input upload: %s", hipGetErrorString((hipError_t)rcb)); return PG_ERR_DEVICE; }
            uint32_t max_big = 0;
            for (const IndexHost& x : job->index) max_big = std::max<uint32_t>(max_big, (uint32_t)x.list_big.size());
            pgk_launch_index(job->d_reps_g + job->nA, n - job->nA, job->max_v, max_big, 0, s);
            HIP_TRY(hipGetLastError());
        }
        if (job->any_legacy_prep) pgk_launch_prep(job->d_contigs, n, job->max_v, job->max_prep_w, job->max_prep_m4, job->tab, s);
        if (job->any_split) pgk_launch_prep_split(job->d_contigs, n, job->max_sb, job->max_sm4, job->max_sw, job->tab, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(job->ev[1], s));
        // (no k_compact: the column list is the index's, made when it was uploaded — the class stays in the timing table, at zero)
        HIP_TRY(hipEventRecord(job->ev[2], s));
        if (job->any_legacy_prep) pgk_launch_records(job->d_contigs, n, job->max_v, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(job->ev[3], s));
        pgk_launch_sweep(job->d_contigs, n, job->hp_mask, 1, s);
        pgk_launch_sweep_small(job->d_contigs, job->d_small, job->n_small, 1, 0, job->d_dump, s);
        pgk_launch_sweep_smallx(job->d_contigs, job->d_smallx, job->n_smallx, 1, 0, job->d_dump, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(job->ev[4], s));
        }
        if (!job->chunked) {
            pgk_launch_sweep(job->d_contigs, n, job->hp_mask, 2, s);
            if (job->small_phase2) pgk_launch_sweep_small(job->d_contigs, job->d_small, job->n_small, 2, 0, job->d_dump, s);
            if (job->smallx_phase2) pgk_launch_sweep_smallx(job->d_contigs, job->d_smallx, job->n_smallx, 2, 0, job->d_dump, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(job->ev[5], s));
            pgk_launch_bins(job->d_contigs, n, job->max_v, job->bins_which, job->max_wide, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(job->ev[6], s));
        } else {
            // chunk i: store-only sweep on s -> ev_sweep -> k_post on stream2 -> ev_post; the sweep of chunk
            // i + PG_SCRATCH_BUFS reuses scratch buffer i % PG_SCRATCH_BUFS and therefore waits for the posteriors of chunk i
            hipStream_t s2 = job->stream2;
            if (persist_now) {
                // one launch each: the chunks are handed over on the device (DevContig::sync, zeroed above); k_post_loop must not
                // start (and wait) before phase 1 has ended
                HIP_TRY(hipStreamWaitEvent(s2, job->ev[4], 0));
                pgk_launch_phase2_persistent(job->d_contigs, n, job->post_blocks, s, s2);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(job->ev_post[0], s2));
                HIP_TRY(hipStreamWaitEvent(s, job->ev_post[0], 0));
            } else
            for (uint32_t i = 0; i < job->n_chunks; ++i) {
                const int b = (int)PG_SCR_BUF(i);
                if (i >= PG_SCRATCH_BUFS) HIP_TRY(hipStreamWaitEvent(s, job->ev_post[b], 0));
                pgk_launch_sweep_chunk(job->d_contigs, n, job->hp_mask, i, s);
                pgk_launch_sweep_small(job->d_contigs, job->d_small, job->n_small, 3, i, job->d_dump, s);
                pgk_launch_sweep_smallx(job->d_contigs, job->d_smallx, job->n_smallx, 3, i, job->d_dump, s);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(job->ev_sweep[b], s));
                HIP_TRY(hipStreamWaitEvent(s2, job->ev_sweep[b], 0));
                pgk_launch_post(job->d_contigs, n, job->chunk_cols, i, s2);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(job->ev_post[b], s2));
            }
            if (!persist_now) for (uint32_t q = 0; q < PG_SCRATCH_BUFS && q < job->n_chunks; ++q) HIP_TRY(hipStreamWaitEvent(s, job->ev_post[q], 0));
            HIP_TRY(hipEventRecord(job->ev[5], s));  // "k_sweep_phase2" = all chunks incl. their posteriors
            HIP_TRY(hipEventRecord(job->ev[6], s));  // (no k_bins in this mode)
        }
    } else if (job->max_v > 0) {
        // run_genotyping == false: the variant records (what the Viterbi reads; the column list is the index's)
        pgk_launch_prep(job->d_contigs, n, job->max_v, job->max_prep_w, job->max_prep_m4, job->tab, s);
        HIP_TRY(hipGetLastError());
    }
    if (job->max_v > 0 && job->params.run_phasing) {
        // Viterbi phasing (reference src/hmm.cpp:47-49): reads k_prep's records, independent of the sweep
        if (!job->vit_tq_ready) {  // first run with this index: the transition probabilities of the kept columns
            HIP_TRY(hipStreamSynchronize(s));
            if (job->stream2) HIP_TRY(hipStreamSynchronize(job->stream2));
            const int rc = viterbi_transitions(job, s, err, errlen);
            if (rc != PG_OK) return rc;
            job->vit_tq_ready = true;
        }
        HIP_TRY(hipEventRecord(job->ev_vit[0], s));
        pgk_launch_viterbi(job->d_contigs, n, job->max_v, job->vit_bits, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(job->ev_vit[1], s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (job->stream2) HIP_TRY(hipStreamSynchronize(job->stream2));
    if (job->max_v > 0 && job->params.run_genotyping) {
        for (int i = 0; i < PG_N_KERNEL_CLASSES; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, job->ev[i], job->ev[i + 1]));
            job->ms[i] = ms;
        }
    }
    if (job->max_v > 0 && job->params.run_phasing) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, job->ev_vit[0], job->ev_vit[1]));
        job->vit_ms = ms;
    }
    std::vector<uint32_t> ncols(n), errs(n);
    {   // (column counts and the index-level kernels' error bits are per index contig)
        const size_t ni = job->index.size();
        std::vector<uint32_t> nc_ix(ni), err_ix(ni);
        HIP_TRY(hipMemcpy(nc_ix.data(), job->d_ncols, sizeof(uint32_t) * ni, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(err_ix.data(), job->d_ixerr, sizeof(uint32_t) * ni, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(errs.data(), job->d_err, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) { ncols[i] = nc_ix[job->chains[i].index]; errs[i] |= err_ix[job->chains[i].index]; }
    }
    job->ran = true;
    job->host_s[2] = now_s() - t_run;
    for (uint32_t i = 0; i < n; ++i) {
        job->chains[i].n_cols_host = ncols[i];
        if (errs[i] & PG_DEVERR_ALLELE_NOT_FOUND) {
            set_err(err, errlen, "chain %u: a path_allele value is not in the variant's allele list", i);
            return PG_ERR_INVALID;
        }
        if (errs[i] & PG_DEVERR_TOO_MANY_ALLELES) {
            set_err(err, errlen, "chain %u: a variant has more than %d alleles (device limit)", i, PG_MAX_ALLELES_PER_VARIANT);
            return PG_ERR_UNSUPPORTED;
        }
        if (errs[i] & PG_DEVERR_SYNC_TIMEOUT) {
            set_err(err, errlen, "chain %u: the persistent phase-2 kernels lost each other (a chunk flag did not arrive within 10 s); without PG_KERNELS=persist phase 2 takes a launch per chunk", i);
            return PG_ERR_DEVICE;
        }
        if (errs[i] & PG_DEVERR_WIDE_FUSED) {
            set_err(err, errlen, "chain %u: its only column carries more than %d alleles on the selected paths, which a fused job cannot genotype (PG_SWEEP_MODE=chunked)", i, PG_AMAX);
            return PG_ERR_UNSUPPORTED;
        }
        if (errs[i] & PG_DEVERR_TOO_MANY_LOCAL) {
            set_err(err, errlen, "chain %u: a column has more than %d distinct alleles on the selected paths (device limit)", i, PG_WIDE_MAX);
            return PG_ERR_UNSUPPORTED;
        }
    }
    return PG_OK;
}

extern "C" int pg_job_fetch(pg_job* job, uint32_t ci, pg_contig_result* out, char* err, size_t errlen) {
    if (!job || !out || ci >= job->chains.size()) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    if (!job->ran) { set_err(err, errlen, "pg_job_run has not been called"); return PG_ERR_INVALID; }
    const double t0 = now_s();
    HIP_TRY(hipSetDevice(job->device));
    const ChainHost& c = job->chains[ci];
    const IndexHost& x = job->index[c.index];
    out->n_columns = c.n_cols_host;
    if (x.V == 0) return PG_OK;
    if (out->lik && x.n_lik) HIP_TRY(hipMemcpy(out->lik, c.d.lik, x.n_lik * sizeof(double), hipMemcpyDeviceToHost));
    if (out->lik_exp && x.n_lik) HIP_TRY(hipMemcpy(out->lik_exp, c.d.lik_exp, x.n_lik * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out->kept) HIP_TRY(hipMemcpy(out->kept, c.d.kept, x.V, hipMemcpyDeviceToHost));
    if (out->allele_present && x.sumA) HIP_TRY(hipMemcpy(out->allele_present, c.d.allele_present, x.sumA, hipMemcpyDeviceToHost));
    // reference src/hmm.cpp:94,106-109: set only when there is at least one column
    const bool fill = job->params.run_genotyping && c.n_cols_host > 0;
    // the Viterbi backtrace sets both at the COLUMN index (sic, reference src/hmm.cpp:164-165): entries [0, C)
    const size_t nvit = job->params.run_phasing ? c.n_cols_host : 0;
    if (out->n_kmers) {
        if (fill) memcpy(out->n_kmers, x.n_kmers.data(), (size_t)x.V * 2);
        else { memset(out->n_kmers, 0, (size_t)x.V * 2); memcpy(out->n_kmers, x.n_kmers.data(), nvit * 2); }
    }
    if (out->coverage) {
        if (fill) memcpy(out->coverage, c.coverage.data(), (size_t)x.V * 2);
        else { memset(out->coverage, 0, (size_t)x.V * 2); memcpy(out->coverage, c.coverage.data(), nvit * 2); }
    }
    if (job->params.run_phasing) {
        if (out->haplotype_1) HIP_TRY(hipMemcpy(out->haplotype_1, c.d.hap1, (size_t)x.V * 2, hipMemcpyDeviceToHost));
        if (out->haplotype_2) HIP_TRY(hipMemcpy(out->haplotype_2, c.d.hap2, (size_t)x.V * 2, hipMemcpyDeviceToHost));
    }
    job->host_s[3] += now_s() - t0;
    return PG_OK;
}

// All chains at once: the copies are queued on the job's stream and waited for once (24 chains x 6 blocking copies
// were 13 ms of the whole-genome job's 360 ms end-to-end time).
extern "C" int pg_job_fetch_all(pg_job* job, pg_contig_result* outs, char* err, size_t errlen) {
    if (!job || !outs) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    if (!job->ran) { set_err(err, errlen, "pg_job_run has not been called"); return PG_ERR_INVALID; }
    const double t0 = now_s();
    HIP_TRY(hipSetDevice(job->device));
    hipStream_t s = job->stream;
    for (size_t ci = 0; ci < job->chains.size(); ++ci) {
        const ChainHost& c = job->chains[ci];
        const IndexHost& x = job->index[c.index];
        pg_contig_result* out = &outs[ci];
        out->n_columns = c.n_cols_host;
        if (x.V == 0) continue;
        if (out->lik && x.n_lik) HIP_TRY(hipMemcpyAsync(out->lik, c.d.lik, x.n_lik * sizeof(double), hipMemcpyDeviceToHost, s));
        if (out->lik_exp && x.n_lik) HIP_TRY(hipMemcpyAsync(out->lik_exp, c.d.lik_exp, x.n_lik * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (out->kept) HIP_TRY(hipMemcpyAsync(out->kept, c.d.kept, x.V, hipMemcpyDeviceToHost, s));
        if (out->allele_present && x.sumA) HIP_TRY(hipMemcpyAsync(out->allele_present, c.d.allele_present, x.sumA, hipMemcpyDeviceToHost, s));
        if (job->params.run_phasing) {
            if (out->haplotype_1) HIP_TRY(hipMemcpyAsync(out->haplotype_1, c.d.hap1, (size_t)x.V * 2, hipMemcpyDeviceToHost, s));
            if (out->haplotype_2) HIP_TRY(hipMemcpyAsync(out->haplotype_2, c.d.hap2, (size_t)x.V * 2, hipMemcpyDeviceToHost, s));
        }
        // host-side meta data while the copies run (rules: pg_job_fetch)
        const bool fill = job->params.run_genotyping && c.n_cols_host > 0;
        const size_t nvit = job->params.run_phasing ? c.n_cols_host : 0;
        if (out->n_kmers) {
            if (fill) memcpy(out->n_kmers, x.n_kmers.data(), (size_t)x.V * 2);
            else { memset(out->n_kmers, 0, (size_t)x.V * 2); memcpy(out->n_kmers, x.n_kmers.data(), nvit * 2); }
        }
        if (out->coverage) {
            if (fill) memcpy(out->coverage, c.coverage.data(), (size_t)x.V * 2);
            else { memset(out->coverage, 0, (size_t)x.V * 2); memcpy(out->coverage, c.coverage.data(), nvit * 2); }
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    job->host_s[3] += now_s() - t0;
    return PG_OK;
}

// The inputs a job holds on the device, back on the host (the panel of a job made by pg_sampler_then_job never was there).
extern "C" int pg_job_panel_sizes(const pg_job* job, uint32_t ci, uint32_t* n_variants, uint32_t* n_paths, uint64_t* sum_kmers, uint64_t* sum_alleles) {
    if (!job || ci >= job->chains.size()) return PG_ERR_INVALID;
    const IndexHost& x = job->index[job->chains[ci].index];
    if (n_variants) *n_variants = x.V;
    if (n_paths) *n_paths = x.H;
    if (sum_kmers) *sum_kmers = x.sumK;
    if (sum_alleles) *sum_alleles = x.sumA;
    return PG_OK;
}
extern "C" int pg_job_fetch_panel(pg_job* job, uint32_t ci, uint32_t* kmer_off, uint16_t* kmer_count, uint32_t* allele_off, uint16_t* allele_id,
                                  uint8_t* allele_flags, uint16_t* allele_kmer_off, uint32_t* allele_kmer_mask, uint16_t* path_allele,
                                  char* err, size_t errlen) {
    if (!job || ci >= job->chains.size()) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    HIP_TRY(hipSetDevice(job->device));
    const ChainHost& c = job->chains[ci];
    const IndexHost& x = job->index[c.index];
    if (x.V == 0) return PG_OK;
    const unsigned char* A = job->arena;
    if (kmer_off) HIP_TRY(hipMemcpy(kmer_off, A + x.o_koff, ((size_t)x.V + 1) * 4, hipMemcpyDeviceToHost));
    if (kmer_count && x.sumK) HIP_TRY(hipMemcpy(kmer_count, job->cur_samples + (c.o_kcnt - job->sample_lo), (size_t)x.sumK * 2, hipMemcpyDeviceToHost));
    if (allele_off) HIP_TRY(hipMemcpy(allele_off, A + x.o_aoff, ((size_t)x.V + 1) * 4, hipMemcpyDeviceToHost));
    if (allele_id) HIP_TRY(hipMemcpy(allele_id, A + x.o_aid, (size_t)x.sumA * 2, hipMemcpyDeviceToHost));
    if (allele_flags) HIP_TRY(hipMemcpy(allele_flags, A + x.o_aflag, (size_t)x.sumA, hipMemcpyDeviceToHost));
    if (allele_kmer_off) HIP_TRY(hipMemcpy(allele_kmer_off, A + x.o_akoff, (size_t)x.sumA * 2, hipMemcpyDeviceToHost));
    if (allele_kmer_mask) HIP_TRY(hipMemcpy(allele_kmer_mask, A + x.o_akmask, (size_t)x.sumA * 4, hipMemcpyDeviceToHost));
    if (path_allele) HIP_TRY(hipMemcpy(path_allele, A + x.o_pa, (size_t)x.V * x.H * 2, hipMemcpyDeviceToHost));
    return PG_OK;
}

extern "C" int pg_job_device_results(pg_job* job, uint32_t ci, void** d_lik, uint64_t* n_lik, void** d_lik_exp, uint64_t* n_variants) {
    if (!job || ci >= job->chains.size()) return PG_ERR_INVALID;
    const ChainHost& c = job->chains[ci];
    const IndexHost& x = job->index[c.index];
    if (d_lik) *d_lik = c.d.lik;
    if (n_lik) *n_lik = x.n_lik;
    if (d_lik_exp) *d_lik_exp = c.d.lik_exp;
    if (n_variants) *n_variants = x.V;
    return PG_OK;
}

extern "C" int pg_job_packed_results(pg_job* job, void** d_lik, void** d_lik_exp, uint64_t* n_lik_total) {
    if (!job) return PG_ERR_INVALID;
    if (d_lik) *d_lik = job->d_lik;
    if (d_lik_exp) *d_lik_exp = job->d_likexp;
    if (n_lik_total) *n_lik_total = job->n_lik_total;
    return PG_OK;
}

extern "C" uint32_t pg_job_n_chains(const pg_job* job) { return job ? (uint32_t)job->chains.size() : 0; }

extern "C" int pg_job_host_seconds(const pg_job* job, double out4[4]) {
    if (!job || !out4) return PG_ERR_INVALID;
    for (int i = 0; i < 4; ++i) out4[i] = job->host_s[i];
    return PG_OK;
}
extern "C" int pg_job_upload_bytes(const pg_job* job, uint64_t out2[2]) {
    if (!job || !out2) return PG_ERR_INVALID;
    out2[0] = job->up_bytes[0]; out2[1] = job->up_bytes[1];
    return PG_OK;
}

extern "C" int pg_job_profile_counters(pg_job* job, uint32_t ci, uint64_t out64[64]) {
    if (!job || !out64 || ci >= job->chains.size()) return PG_ERR_INVALID;
    if (hipSetDevice(job->device) != hipSuccess) return PG_ERR_DEVICE;
    if (hipMemcpy(out64, job->chains[ci].d.prof, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return PG_ERR_DEVICE;
    return PG_OK;
}

extern "C" int pg_job_kernel_ms(const pg_job* job, double ms[PG_N_KERNEL_CLASSES]) {
    if (!job || !ms) return PG_ERR_INVALID;
    for (int i = 0; i < PG_N_KERNEL_CLASSES; ++i) ms[i] = job->ms[i];
    return PG_OK;
}
extern "C" const char* pg_job_kernel_name(int cls) { return (cls >= 0 && cls < PG_N_KERNEL_CLASSES) ? kKernelNames[cls] : ""; }
extern "C" uint64_t pg_job_device_bytes(const pg_job* job) { return job ? job->arena_bytes : 0; }
extern "C" int pg_job_sweep_mode(const pg_job* job, uint32_t* chunk_cols) {
    if (!job) return PG_ERR_INVALID;
    if (chunk_cols) *chunk_cols = job->chunk_cols;
    return job->chunked ? 1 : 0;
}

extern "C" double pg_job_viterbi_ms(const pg_job* job) { return job ? job->vit_ms : 0.0; }

extern "C" double pg_job_index_ms(const pg_job* job) { return job ? job->index_ms : 0.0; }
extern "C" size_t pg_job_plan(const pg_job* job, char* out, size_t len) {
    if (!job) return 0;
    std::string t = job->plan_text;
    if (job->persist) {   // (what the runs so far really did: a run whose streams share a hardware queue takes a launch per chunk)
        char line[160];
        snprintf(line, sizeof(line), "  persistent phase 2: %u run(s), %u more fell back to a launch per chunk\n", job->persist_runs, job->persist_fallbacks);
        t += line;
    }
    if (out && len) { const size_t n = t.size() < len - 1 ? t.size() : len - 1; memcpy(out, t.data(), n); out[n] = 0; }
    return t.size() + 1;
}
extern "C" uint32_t pg_job_triangle_chains(const pg_job* job) {
    if (!job) return 0;
    uint32_t n = 0;
    for (const auto& ch : job->chains) n += ch.d.tri ? 1u : 0u;
    return n;
}

namespace {

// the body of the one-shot call for ONE chain: a job of its own (arena from / to the pool)
int genotype_single(const pg_contig_batch* batch, const pg_table* table, const pg_hmm_params* params, int device,
                    pg_contig_result* out, char* err, size_t errlen) {
    std::vector<ChainSpec> specs(1);
    specs[0] = {0, batch->kmer_count, batch->coverage};
    pg_job* job = nullptr;
    int rc = job_build(device, 1, batch, specs, 1, false, table, params, true, &job, err, errlen);
    if (rc != PG_OK) return rc;
    rc = pg_job_run(job, nullptr, err, errlen);
    if (rc == PG_OK) rc = pg_job_fetch(job, 0, out, err, errlen);
    pg_job_destroy(job);
    return rc;
}

// ---------------------------------------------------------------------------------------
//  Coalescing of concurrent one-shot calls.
//
//  The reference constructs one HMM per (contig x subset) on N thread-pool workers at a time
//  (src/commands.cpp:949-978, run_genotyping :155-185).  N one-shot jobs side by side would each bring their own
//  two streams (more streams than hardware queues: kernels of different jobs then queue behind each other's
//  half-chain kernels, which run for the whole phase), ~100 launches per job, and a k_post grid sized as if the job
//  had the chip to itself.  So calls that are in flight at the same time on the same device with the same table and
//  parameters are merged into ONE job — exactly the resident multi-chain job of pg_job_new — by whichever caller
//  arrives first (the leader); the others sleep until their results are in their buffers.  Chains of a job are
//  independent, so every caller gets bit for bit what it would have got alone.
//    * a leader launches when a job slot of the device is free (at most 2 merged jobs at a
//      time per device), every ANNOUNCED call has arrived (pg_hmm_announce: the C++ adapter announces at the top of
//      the HMM constructor, before it flattens its UniqueKmers — the leader then knows who is still coming; bounded by
//      PG_COALESCE_WAIT_MS = 250), and nobody new has joined for 300 us (only when other
//      callers have been seen at all: a single-threaded host never waits);
//    * if the merged job fails (one malformed batch, a device limit, no memory for the sum) the leader runs the
//      requests one by one, so every caller gets its own error code and message;
//    * PG_COALESCE=0 turns it off.
// ---------------------------------------------------------------------------------------
struct CoRequest {
    const pg_contig_batch* batch; pg_contig_result* out; char* err; size_t errlen;
    int rc = PG_OK; bool done = false;
};
struct CoBatch {
    int device; const pg_table* table; pg_hmm_params params;
    std::vector<CoRequest*> reqs;
    std::chrono::steady_clock::time_point last_arrival;
};
struct Coalescer {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::shared_ptr<CoBatch>> open;
    int callers_inside = 0;
    int announced[PG_MAX_DEVICES_HOST] = {0};
    int inflight[PG_MAX_DEVICES_HOST] = {0};
    std::chrono::steady_clock::time_point last_concurrency = std::chrono::steady_clock::time_point::min();
    uint64_t stat_batches = 0, stat_requests = 0, stat_largest = 0;
} g_co;

long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? strtol(e, nullptr, 0) : dflt; }

bool same_params(const pg_hmm_params& a, const pg_hmm_params& b) {
    return a.effective_N == b.effective_N && a.recombrate == b.recombrate && (a.uniform != 0) == (b.uniform != 0) &&
           (a.run_genotyping != 0) == (b.run_genotyping != 0) && (a.run_phasing != 0) == (b.run_phasing != 0);
}

void run_merged(CoBatch& b) {
    const size_t n = b.reqs.size();
    pg_hmm_params prm = b.params;
    prm.reserved = 0;
    if (n > 1) {
        std::vector<pg_contig_batch> bs(n);
        std::vector<ChainSpec> specs(n);
        for (size_t i = 0; i < n; ++i) { bs[i] = *b.reqs[i]->batch; specs[i] = {(uint32_t)i, bs[i].kmer_count, bs[i].coverage}; }
        char err[512] = {0};
        pg_job* job = nullptr;
        int rc = job_build(b.device, (uint32_t)n, bs.data(), specs, 1, false, b.table, &prm, true, &job, err, sizeof(err));
        if (rc == PG_OK) rc = pg_job_run(job, nullptr, err, sizeof(err));
        if (rc == PG_OK) {
            std::vector<pg_contig_result> outs(n);
            for (size_t i = 0; i < n; ++i) outs[i] = *b.reqs[i]->out;
            rc = pg_job_fetch_all(job, outs.data(), err, sizeof(err));
            for (size_t i = 0; i < n; ++i) b.reqs[i]->out->n_columns = outs[i].n_columns;
        }
        if (job) pg_job_destroy(job);
        if (rc == PG_OK) { for (CoRequest* r : b.reqs) r->rc = PG_OK; return; }
        // fall through: one by one, every caller its own verdict
    }
    for (CoRequest* r : b.reqs) r->rc = genotype_single(r->batch, b.table, &prm, b.device, r->out, r->err, r->errlen);
}

}  // namespace

extern "C" void pg_hmm_announce(int device) {
    if (device < 0 || device >= PG_MAX_DEVICES_HOST) return;
    std::lock_guard<std::mutex> lock(g_co.mu);
    g_co.announced[device] += 1;
}
extern "C" void pg_hmm_retract(int device) {
    if (device < 0 || device >= PG_MAX_DEVICES_HOST) return;
    std::lock_guard<std::mutex> lock(g_co.mu);
    if (g_co.announced[device] > 0) g_co.announced[device] -= 1;
    g_co.cv.notify_all();
}
extern "C" int pg_hmm_coalesce_stats(uint64_t out3[3]) {
    if (!out3) return PG_ERR_INVALID;
    std::lock_guard<std::mutex> lock(g_co.mu);
    out3[0] = g_co.stat_batches; out3[1] = g_co.stat_requests; out3[2] = g_co.stat_largest;
    return PG_OK;
}

extern "C" int pg_hmm_genotype_contig(const pg_contig_batch* batch, const pg_table* table, const pg_hmm_params* params,
                                      int device, pg_contig_result* out, char* err, size_t errlen) {
    if (!batch || !table || !params || !out) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    using clock = std::chrono::steady_clock;
    static const bool enabled = env_long("PG_COALESCE", 1) != 0;
    static const long wait_ms = env_long("PG_COALESCE_WAIT_MS", 250);
    const long window_us = 300, max_inflight = 2, max_batch = 256;   // (measured settings)
    const bool announced = (params->reserved & PG_CALL_ANNOUNCED) != 0;
    if (!enabled || device < 0 || device >= PG_MAX_DEVICES_HOST) {
        if (announced) pg_hmm_retract(device);
        return genotype_single(batch, table, params, device, out, err, errlen);
    }
    CoRequest req{batch, out, err, errlen};
    std::shared_ptr<CoBatch> mine;
    {
        std::unique_lock<std::mutex> lk(g_co.mu);
        if (announced && g_co.announced[device] > 0) g_co.announced[device] -= 1;
        g_co.callers_inside += 1;
        if (g_co.callers_inside > 1) g_co.last_concurrency = clock::now();
        for (auto& ob : g_co.open)
            if (ob->device == device && ob->table == table && same_params(ob->params, *params) && (long)ob->reqs.size() < max_batch) {
                ob->reqs.push_back(&req);
                ob->last_arrival = clock::now();
                g_co.cv.notify_all();
                g_co.cv.wait(lk, [&] { return req.done; });
                g_co.callers_inside -= 1;
                return req.rc;
            }
        // leader of a new batch
        mine = std::make_shared<CoBatch>();
        mine->device = device; mine->table = table; mine->params = *params;
        mine->reqs.push_back(&req);
        mine->last_arrival = clock::now();
        g_co.open.push_back(mine);
        g_co.cv.notify_all();
        const clock::time_point hard = clock::now() + std::chrono::milliseconds(wait_ms);
        for (;;) {
            const clock::time_point now = clock::now();
            const bool concurrent = g_co.callers_inside > 1 || (g_co.last_concurrency != clock::time_point::min() &&
                                                                now - g_co.last_concurrency < std::chrono::seconds(2));
            const clock::time_point quiet_until = mine->last_arrival + std::chrono::microseconds(concurrent ? window_us : 0);
            const bool slot = g_co.inflight[device] < max_inflight;
            const bool coming = g_co.announced[device] > 0 && now < hard;
            const bool full = (long)mine->reqs.size() >= max_batch;
            if (slot && (full || (!coming && now >= quiet_until))) break;
            clock::time_point until = now + std::chrono::milliseconds(50);
            if (slot && !coming && quiet_until < until) until = quiet_until;
            if (coming && hard < until) until = hard;
            g_co.cv.wait_until(lk, until);
        }
        for (size_t i = 0; i < g_co.open.size(); ++i)
            if (g_co.open[i] == mine) { g_co.open.erase(g_co.open.begin() + (long)i); break; }
        g_co.inflight[device] += 1;
        g_co.stat_batches += 1; g_co.stat_requests += mine->reqs.size();
        if (mine->reqs.size() > g_co.stat_largest) g_co.stat_largest = mine->reqs.size();
    }
    try {
        run_merged(*mine);
    } catch (...) {   // (std::bad_alloc of the host-side vectors, ...): every caller of the batch gets an error — and is woken up below
        for (CoRequest* r : mine->reqs) {
            r->rc = PG_ERR_DEVICE;
            set_err(r->err, r->errlen, "merged one-shot job failed with a host exception");
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_co.mu);
        g_co.inflight[device] -= 1;
        g_co.callers_inside -= 1;
        for (CoRequest* r : mine->reqs) r->done = true;
        g_co.cv.notify_all();
    }
    return req.rc;
}

// ---------------------------------------------------------------------------------------
//  unit-level entry points
// ---------------------------------------------------------------------------------------
extern "C" int pg_emission_table(const pg_contig_batch* batch, const pg_table* table, uint32_t v, int device,
                                 long double* out, int32_t* all_zeros_out, char* err, size_t errlen) {
    if (!batch || !table || !out || v >= batch->n_variants) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    pg_hmm_params p;
    memset(&p, 0, sizeof(p));
    p.effective_N = 25000.0L; p.recombrate = 1.26; p.run_genotyping = 1;
    pg_job* job = nullptr;
    int rc = pg_job_new(device, 1, batch, table, &p, &job, err, errlen);
    if (rc != PG_OK) return rc;
    const uint32_t A = batch->allele_off[v + 1] - batch->allele_off[v];
    double* dm = nullptr; int* de = nullptr;
    std::vector<double> m((size_t)A * A);
    std::vector<int> e((size_t)A * A);
    if (hipMalloc((void**)&dm, m.size() * 8 + 8) != hipSuccess || hipMalloc((void**)&de, e.size() * 4 + 8) != hipSuccess) {
        set_err(err, errlen, "hipMalloc failed"); rc = PG_ERR_NOMEM;
    } else {
        pgk_launch_emission_single(job->d_contigs, job->tab, v, dm, de, job->stream);
        if (hipStreamSynchronize(job->stream) != hipSuccess ||
            hipMemcpy(m.data(), dm, m.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(e.data(), de, e.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            set_err(err, errlen, "emission kernel failed: %s", hipGetErrorString(hipGetLastError())); rc = PG_ERR_DEVICE;
        }
    }
    if (dm) hipFree(dm);
    if (de) hipFree(de);
    pg_job_destroy(job);
    if (rc != PG_OK) return rc;
    bool all_zeros = true;  // reference src/emissionprobabilitycomputer.cpp:24,31-34
    for (size_t i = 0; i < m.size(); ++i) {
        out[i] = ldexpl((long double)m[i], e[i]);
        if (out[i] > 0) all_zeros = false;
    }
    if (all_zeros)
        for (size_t i = 0; i < m.size(); ++i) out[i] = 1.0L;
    if (all_zeros_out) *all_zeros_out = all_zeros ? 1 : 0;
    return PG_OK;
}

extern "C" int pg_transition_probs(uint64_t from_pos, uint64_t to_pos, double recombrate, uint32_t nr_paths, int uniform,
                                   long double effective_N, int device, double out3[3], char* err, size_t errlen) {
    if (!out3 || nr_paths == 0) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(err, errlen, "no HIP device available (no CPU fallback)"); return PG_ERR_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 3 * sizeof(double)));
    const long double dist = (to_pos - from_pos) * 0.000004L * ((long double)recombrate) * effective_N;
    pgk_launch_transition_single((double)dist, nr_paths, uniform, d, nullptr);
    hipError_t he = hipMemcpy(out3, d, 3 * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d);
    if (he != hipSuccess) { set_err(err, errlen, "transition kernel failed: %s", hipGetErrorString(he)); return PG_ERR_DEVICE; }
    return PG_OK;
}
