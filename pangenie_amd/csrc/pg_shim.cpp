// pg_shim.cpp — C-ABI of include/pangenie_hmm.h on top of the HIP kernels (pg_kernels.hip).
//
// Host side of the boundary: argument checking, one device arena per job, H2D/D2H,
// launch order, hipEvent timing per kernel class.  No compute happens here except the
// ProbabilityTable, which the reference also builds on the host in long double before any
// HMM runs (reference src/commands.cpp:846, src/probabilitytable.cpp:28-45).
// There is NO CPU fallback: without a HIP device every entry point returns PG_ERR_DEVICE.
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/pangenie_hmm.h"
#include "pg_device.h"

extern "C" {
void pgk_launch_prep(const DevContig*, uint32_t, uint32_t, DevTable, hipStream_t);
void pgk_launch_compact(const DevContig*, uint32_t, hipStream_t);
void pgk_launch_records(const DevContig*, uint32_t, uint32_t, hipStream_t);
void pgk_launch_bins(const DevContig*, uint32_t, uint32_t, hipStream_t);
void pgk_launch_sweep(const DevContig*, uint32_t, uint32_t, int, hipStream_t);
void pgk_launch_sweep_chunk(const DevContig*, uint32_t, uint32_t, uint32_t, hipStream_t);
void pgk_launch_post(const DevContig*, uint32_t, uint32_t, uint32_t, hipStream_t);
void pgk_launch_emission_single(const DevContig*, DevTable, uint32_t, double*, int*, hipStream_t);
void pgk_launch_transition_single(double, uint32_t, int, double*, hipStream_t);
uint32_t pgk_threads_for_hp(uint32_t);
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

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace

// ---------------------------------------------------------------------------------------
//  ProbabilityTable (host, long double) — reference src/probabilitytable.cpp
// ---------------------------------------------------------------------------------------
struct pg_table {
    uint16_t cov_min = 0, cov_max = 0, count_max = 0;
    long double reg = 0.0L;
    std::vector<long double> p;  // [count][cov - cov_min][3], as the reference indexes it
    uint64_t version = 1;
    // device copies, one per device, rebuilt when `version` changes
    struct DevCopy { int device; uint64_t version; double* mant; int32_t* expo; };
    std::vector<DevCopy> dev;
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

// (mantissa, exponent) view of the dense table on `device`, layout [cov][count][3]
int table_on_device(pg_table* t, int device, DevTable* out, char* err, size_t errlen) {
    std::lock_guard<std::mutex> lock(t->mu);
    const uint32_t ncov = t->cov_max > t->cov_min ? t->cov_max - t->cov_min : 0;
    out->cov_min = t->cov_min; out->cov_max = t->cov_max; out->count_max = t->count_max; out->pad = 0;
    out->reg = (double)t->reg;
    for (auto& d : t->dev)
        if (d.device == device && d.version == t->version) { out->mant = d.mant; out->expo = d.expo; return PG_OK; }
    const size_t n = (size_t)ncov * t->count_max * 3;
    std::vector<double> m(n ? n : 1);
    std::vector<int32_t> e(n ? n : 1);
    for (uint32_t c = 0; c < ncov; ++c)
        for (uint32_t k = 0; k < t->count_max; ++k)
            for (int i = 0; i < 3; ++i) {
                long double v = t->p[((size_t)k * ncov + c) * 3 + i];
                int ex = 0;
                long double mant = (v == 0.0L || v != v || isinf((double)v)) ? v : frexpl(v, &ex);
                m[((size_t)c * t->count_max + k) * 3 + i] = (double)mant;
                e[((size_t)c * t->count_max + k) * 3 + i] = ex;
            }
    double* dm = nullptr; int32_t* de = nullptr;
    HIP_TRY(hipMalloc((void**)&dm, m.size() * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&de, e.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(dm, m.data(), m.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(de, e.data(), e.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    bool replaced = false;
    for (auto& d : t->dev)
        if (d.device == device) { hipFree(d.mant); hipFree(d.expo); d = {device, t->version, dm, de}; replaced = true; }
    if (!replaced) t->dev.push_back({device, t->version, dm, de});
    out->mant = dm; out->expo = de;
    return PG_OK;
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
    t->version++;
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
extern "C" void pg_table_destroy(pg_table* t) {
    if (!t) return;
    for (auto& d : t->dev) {
        if (hipSetDevice(d.device) == hipSuccess) { hipFree(d.mant); hipFree(d.expo); }
    }
    delete t;
}

// ---------------------------------------------------------------------------------------
//  misc
// ---------------------------------------------------------------------------------------
extern "C" int pg_hmm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" const char* pg_hmm_version(void) { return "pangenie-hmm-mi355x 0.1 (gfx950, fp64)"; }

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

struct ContigHost {
    uint32_t V = 0, H = 0, HP = 0, T = 0, RB = 0, part_slots = 1;
    uint32_t sumK = 0, sumA = 0, n_wide = 0;
    uint64_t n_lik = 0;
    std::vector<uint16_t> n_kmers, coverage;
    DevContig d;
    uint32_t n_cols_host = 0;
};

struct pg_job {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<ContigHost> contigs;
    DevContig* d_contigs = nullptr;
    unsigned char* arena = nullptr;
    size_t arena_bytes = 0;
    // regions zeroed at the start of every run
    unsigned char* zero_base = nullptr;
    size_t zero_bytes = 0;
    uint32_t* d_ncols = nullptr;  // [n]
    uint32_t* d_err = nullptr;    // [n]
    DevTable tab;
    uint32_t hp_mask = 0, max_v = 0;
    hipEvent_t ev[PG_N_KERNEL_CLASSES + 1];
    bool events = false;
    double ms[PG_N_KERNEL_CLASSES] = {0, 0, 0, 0, 0, 0};
    pg_hmm_params params;
    bool ran = false;
    // Sweep mode.  fused: phase 2 forms the posterior partials inline (k_bins reduces them) — least
    // HBM traffic, the choice when hundreds of chains fill the chip.  chunked: with few chains the
    // chip is idle and a chain's time is its per-column latency, so phase 2 runs as store-only
    // chunks (as cheap per column as phase 1) and k_post forms the posteriors of each finished
    // chunk on the idle CUs from a second stream.
    bool chunked = false;
    uint32_t chunk_cols = 0, n_chunks = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_sweep[2], ev_post[2];
    bool events2 = false;
};

namespace {

uint32_t pad_paths(uint32_t H) {
    if (H <= 16) return 16;
    if (H <= 32) return 32;
    if (H <= 64) return 64;
    if (H <= 128) return 128;
    return 0;
}

int check_batch(const pg_contig_batch* b, char* err, size_t errlen) {
    if (!b) { set_err(err, errlen, "null batch"); return PG_ERR_INVALID; }
    if (b->n_variants > 0) {
        if (b->n_paths == 0) {  // reference src/columnindexer.cpp:18-22
            set_err(err, errlen, "HMM::index_columns: column 0 is not covered by any paths.");
            return PG_ERR_NO_PATHS;
        }
        if (!b->variant_pos || !b->coverage || !b->kmer_off || !b->allele_off || !b->allele_id ||
            !b->allele_flags || !b->allele_kmer_off || !b->allele_kmer_mask || !b->path_allele) {
            set_err(err, errlen, "batch has null arrays");
            return PG_ERR_INVALID;
        }
        if (b->kmer_off[0] != 0 || b->allele_off[0] != 0) { set_err(err, errlen, "offset arrays must start at 0"); return PG_ERR_INVALID; }
        if (b->kmer_off[b->n_variants] > 0 && !b->kmer_count) { set_err(err, errlen, "kmer_count is null"); return PG_ERR_INVALID; }
    }
    if (b->n_paths > 128) {
        set_err(err, errlen, "device path supports at most 128 selected paths per chain (got %u); use path subsets (-a)", b->n_paths);
        return PG_ERR_UNSUPPORTED;
    }
    return PG_OK;
}

}  // namespace

extern "C" void pg_job_destroy(pg_job* job) {
    if (!job) return;
    hipSetDevice(job->device);
    if (job->events)
        for (auto& e : job->ev) hipEventDestroy(e);
    if (job->events2)
        for (int q = 0; q < 2; ++q) { hipEventDestroy(job->ev_sweep[q]); hipEventDestroy(job->ev_post[q]); }
    if (job->arena) hipFree(job->arena);
    if (job->stream2) hipStreamDestroy(job->stream2);
    if (job->stream) hipStreamDestroy(job->stream);
    delete job;
}

extern "C" pg_job* pg_job_create(int device, uint32_t n_contigs, const pg_contig_batch* batches,
                                 const pg_table* table, const pg_hmm_params* params, char* err, size_t errlen) {
    if (!batches || !table || !params || n_contigs == 0) { set_err(err, errlen, "null argument"); return nullptr; }
    if (params->run_phasing) {
        set_err(err, errlen, "run_phasing (Viterbi, reference src/hmm.cpp:112-173) is not on the device path");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(err, errlen, "no HIP device available (no CPU fallback)"); return nullptr; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "bad device %d", device); return nullptr; }
    for (uint32_t i = 0; i < n_contigs; ++i)
        if (check_batch(&batches[i], err, errlen) != PG_OK) return nullptr;
    if (hipSetDevice(device) != hipSuccess) { set_err(err, errlen, "hipSetDevice failed"); return nullptr; }

    pg_job* job = new pg_job();
    job->device = device;
    job->params = *params;
    auto fail = [&](const char* what, hipError_t e) -> pg_job* {
        set_err(err, errlen, "%s: %s", what, hipGetErrorString(e));
        pg_job_destroy(job);
        return nullptr;
    };
    hipError_t he;
    if ((he = hipStreamCreate(&job->stream)) != hipSuccess) return fail("hipStreamCreate", he);
    for (auto& e : job->ev)
        if ((he = hipEventCreate(&e)) != hipSuccess) return fail("hipEventCreate", he);
    job->events = true;
    if (table_on_device(const_cast<pg_table*>(table), device, &job->tab, err, errlen) != PG_OK) { pg_job_destroy(job); return nullptr; }

    // ---- sweep mode ------------------------------------------------------------------
    {
        uint32_t max_hp = 0, max_v = 0;
        size_t per_col = 0;  // scratch bytes per chunk column over all chains (2 buffers x 2 roles)
        for (uint32_t i = 0; i < n_contigs; ++i) {
            const uint32_t hp = pad_paths(batches[i].n_paths ? batches[i].n_paths : 1);
            if (hp > max_hp) max_hp = hp;
            if (batches[i].n_variants > max_v) max_v = batches[i].n_variants;
            per_col += (size_t)4 * hp * hp * sizeof(double);
        }
        // variants with more than PG_AMAX alleles may turn into WIDE columns, whose posteriors only
        // k_post can form: such jobs always run chunked
        bool wide_candidates = false;
        for (uint32_t i = 0; i < n_contigs && !wide_candidates; ++i)
            for (uint32_t v = 0; v < batches[i].n_variants; ++v)
                if (batches[i].allele_off[v + 1] - batches[i].allele_off[v] > PG_AMAX) { wide_candidates = true; break; }
        bool want = n_contigs * 2u < 128u;  // fewer workgroups than half the CUs
        if (const char* m = getenv("PG_SWEEP_MODE")) {
            if (!strcmp(m, "fused")) want = false;
            else if (!strcmp(m, "chunked")) want = true;
        }
        if (wide_candidates) want = true;
        job->chunked = want && max_v > 0 && params->run_genotyping;
        if (job->chunked) {
            size_t k = 4096;
            const size_t budget = (size_t)12 << 30;
            if (per_col * k > budget) k = budget / per_col;
            if (const char* e = getenv("PG_CHUNK_COLS")) { const long v = strtol(e, nullptr, 0); if (v > 0) k = (size_t)v; }
            const size_t half = (size_t)max_v / 2 + 1;
            if (k > half) k = half;
            if (k < 1) k = 1;
            job->chunk_cols = (uint32_t)k;
            job->n_chunks = (uint32_t)((half + k - 1) / k);
            if ((he = hipStreamCreateWithFlags(&job->stream2, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", he);
            for (int q = 0; q < 2; ++q) {
                if ((he = hipEventCreateWithFlags(&job->ev_sweep[q], hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", he);
                if ((he = hipEventCreateWithFlags(&job->ev_post[q], hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", he);
            }
            job->events2 = true;
        }
    }

    // ---- plan the arena -------------------------------------------------------------
    job->contigs.resize(n_contigs);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + (bytes ? bytes : 8)); return o; };
    struct Plan { size_t scratch, wide, widx, prof, fback, fscale, bscale, bsum, pos, cov, koff, kcnt, aoff, aid, aflag, akoff, akmask, pa, goff, vrec, cvar, colrec, fwd, part, kept, apres, lik, likexp; };
    std::vector<Plan> plan(n_contigs);
    const size_t o_contigs = take(sizeof(DevContig) * n_contigs);
    // zeroed-every-run block: n_cols, err, then per contig kept / allele_present / lik / lik_exp
    const size_t zero_lo = off;
    const size_t o_ncols = take(sizeof(uint32_t) * n_contigs);
    const size_t o_err = take(sizeof(uint32_t) * n_contigs);
    for (uint32_t i = 0; i < n_contigs; ++i) {
        const pg_contig_batch& b = batches[i];
        ContigHost& c = job->contigs[i];
        c.V = b.n_variants; c.H = b.n_paths; c.HP = pad_paths(c.H ? c.H : 1);
        c.T = pgk_threads_for_hp(c.HP); c.RB = pg_rec_bytes(c.HP);
        c.sumK = c.V ? b.kmer_off[c.V] : 0; c.sumA = c.V ? b.allele_off[c.V] : 0;
        uint64_t nl = 0, maxA = 1;
        for (uint32_t v = 0; v < c.V; ++v) { uint64_t A = b.allele_off[v + 1] - b.allele_off[v]; nl += A * (A + 1) / 2; if (A > maxA) maxA = A; }
        c.part_slots = (uint32_t)(maxA < PG_AMAX ? maxA : PG_AMAX);
        c.part_slots = (c.part_slots + 1u) & ~1u;  // partials are stored as allele pairs (16-byte stores)
        if (c.part_slots < 2) c.part_slots = 2;
        c.n_lik = nl;
        plan[i].kept = take(c.V);
        plan[i].fback = take(c.V);
        plan[i].prof = take(64 * sizeof(unsigned long long));
        plan[i].apres = take(c.sumA);
        plan[i].lik = take(nl * sizeof(double));
        plan[i].likexp = take((size_t)c.V * sizeof(int32_t));
    }
    const size_t zero_hi = off;
    for (uint32_t i = 0; i < n_contigs; ++i) {
        ContigHost& c = job->contigs[i];
        Plan& p = plan[i];
        p.pos = take((size_t)c.V * 8); p.cov = take((size_t)c.V * 2);
        p.koff = take(((size_t)c.V + 1) * 4); p.kcnt = take((size_t)c.sumK * 2);
        p.aoff = take(((size_t)c.V + 1) * 4); p.aid = take((size_t)c.sumA * 2);
        p.aflag = take(c.sumA); p.akoff = take((size_t)c.sumA * 2); p.akmask = take((size_t)c.sumA * 4);
        p.pa = take((size_t)c.V * c.H * 2);
        p.goff = take(((size_t)c.V + 1) * 8);
        p.vrec = take((size_t)c.V * c.RB);
        p.cvar = take((size_t)c.V * 4);
        p.colrec = take((size_t)c.V * c.RB);
        p.fwd = take((size_t)c.V * c.HP * c.HP * sizeof(double));
        // fused mode: posterior partials; chunked mode: the chunk scratch instead (k_post writes lik directly)
        p.part = take(job->chunked ? 0 : (size_t)c.V * c.part_slots * c.T * sizeof(double));
        p.scratch = take(job->chunked ? (size_t)4 * job->chunk_cols * c.HP * c.HP * sizeof(double) : 0);
        c.n_wide = 0;
        for (uint32_t v = 0; v < c.V; ++v) c.n_wide += (batches[i].allele_off[v + 1] - batches[i].allele_off[v] > PG_AMAX) ? 1u : 0u;
        p.wide = take((size_t)c.n_wide * PG_WIDE_ENTRY_BYTES);
        p.widx = take(c.n_wide ? (size_t)c.V * sizeof(uint32_t) : 0);
        p.fscale = take((size_t)c.V * sizeof(double));
        p.bscale = take((size_t)c.V * sizeof(double));
        p.bsum = take((size_t)c.V * sizeof(double));
        job->hp_mask |= c.HP == 16 ? 1u : c.HP == 32 ? 2u : c.HP == 64 ? 4u : 8u;
        if (c.V > job->max_v) job->max_v = c.V;
    }
    job->arena_bytes = off;
    if ((he = hipMalloc((void**)&job->arena, job->arena_bytes)) != hipSuccess) {
        set_err(err, errlen, "hipMalloc(%zu bytes) failed: %s", job->arena_bytes, hipGetErrorString(he));
        pg_job_destroy(job);
        return nullptr;
    }
    unsigned char* A = job->arena;
    job->d_contigs = (DevContig*)(A + o_contigs);
    job->d_ncols = (uint32_t*)(A + o_ncols);
    job->d_err = (uint32_t*)(A + o_err);
    job->zero_base = A + zero_lo;
    job->zero_bytes = zero_hi - zero_lo;

    // ---- upload -----------------------------------------------------------------------
    std::vector<DevContig> hd(n_contigs);
    const long double dist_scale = 0.000004L * ((long double)params->recombrate) * params->effective_N;
    for (uint32_t i = 0; i < n_contigs; ++i) {
        const pg_contig_batch& b = batches[i];
        ContigHost& c = job->contigs[i];
        Plan& p = plan[i];
        DevContig& d = hd[i];
        memset(&d, 0, sizeof(d));
        d.V = c.V; d.H = c.H; d.HP = c.HP; d.RB = c.RB; d.T = c.T; d.part_slots = c.part_slots;
        d.dist_scale = (double)dist_scale; d.uniform = params->uniform ? 1 : 0;
        { const char* dbg = getenv("PG_DEBUG"); d.debug = dbg ? (uint32_t)strtoul(dbg, nullptr, 0) : 0u; }
        d.pos = (const uint64_t*)(A + p.pos); d.cov = (const uint16_t*)(A + p.cov);
        d.kmer_off = (const uint32_t*)(A + p.koff); d.kmer_count = (const uint16_t*)(A + p.kcnt);
        d.allele_off = (const uint32_t*)(A + p.aoff); d.allele_id = (const uint16_t*)(A + p.aid);
        d.allele_flags = (const uint8_t*)(A + p.aflag); d.allele_koff = (const uint16_t*)(A + p.akoff);
        d.allele_kmask = (const uint32_t*)(A + p.akmask); d.path_allele = (const uint16_t*)(A + p.pa);
        d.geno_off = (const uint64_t*)(A + p.goff);
        d.vrec = A + p.vrec; d.kept = A + p.kept; d.allele_present = A + p.apres;
        d.n_cols = job->d_ncols + i; d.col_variant = (uint32_t*)(A + p.cvar); d.colrec = A + p.colrec;
        d.fwd = (double*)(A + p.fwd); d.part = (double*)(A + p.part); d.fwd_fallback = A + p.fback; d.prof = (unsigned long long*)(A + p.prof);
        d.fscale = (double*)(A + p.fscale); d.bscale = (double*)(A + p.bscale); d.bsum = (double*)(A + p.bsum); d.err = job->d_err + i;
        d.lik = (double*)(A + p.lik); d.lik_exp = (int32_t*)(A + p.likexp);
        d.scratch = (double*)(A + p.scratch); d.chunk_cols = job->chunk_cols;
        d.wide = A + p.wide; d.wide_idx = c.n_wide ? (const uint32_t*)(A + p.widx) : nullptr;
        c.d = d;
        if (c.V == 0) continue;
        std::vector<uint64_t> goff((size_t)c.V + 1);
        pg_hmm_geno_offsets(&b, goff.data());
        c.n_kmers.resize(c.V); c.coverage.resize(c.V);
        for (uint32_t v = 0; v < c.V; ++v) {
            c.n_kmers[v] = (uint16_t)(b.kmer_off[v + 1] - b.kmer_off[v]);
            c.coverage[v] = b.coverage[v];
        }
#define UP(dst, src, bytes)                                                                         \
    if ((bytes) > 0 && (he = hipMemcpy((void*)(dst), (src), (bytes), hipMemcpyHostToDevice)) != hipSuccess) \
        return fail("hipMemcpy H2D", he);
        UP(d.pos, b.variant_pos, (size_t)c.V * 8);
        UP(d.cov, b.coverage, (size_t)c.V * 2);
        UP(d.kmer_off, b.kmer_off, ((size_t)c.V + 1) * 4);
        UP(d.kmer_count, b.kmer_count, (size_t)c.sumK * 2);
        UP(d.allele_off, b.allele_off, ((size_t)c.V + 1) * 4);
        UP(d.allele_id, b.allele_id, (size_t)c.sumA * 2);
        UP(d.allele_flags, b.allele_flags, (size_t)c.sumA);
        UP(d.allele_koff, b.allele_kmer_off, (size_t)c.sumA * 2);
        UP(d.allele_kmask, b.allele_kmer_mask, (size_t)c.sumA * 4);
        UP(d.path_allele, b.path_allele, (size_t)c.V * c.H * 2);
        UP(d.geno_off, goff.data(), ((size_t)c.V + 1) * 8);
        if (c.n_wide) {
            std::vector<uint32_t> widx(c.V, PG_WIDE_NONE);
            uint32_t k = 0;
            for (uint32_t v = 0; v < c.V; ++v)
                if (b.allele_off[v + 1] - b.allele_off[v] > PG_AMAX) widx[v] = k++;
            UP(d.wide_idx, widx.data(), (size_t)c.V * 4);
        }
#undef UP
    }
    if ((he = hipMemcpy(job->d_contigs, hd.data(), sizeof(DevContig) * n_contigs, hipMemcpyHostToDevice)) != hipSuccess)
        return fail("hipMemcpy contigs", he);
    return job;
}

extern "C" int pg_job_run(pg_job* job, void* stream_, char* err, size_t errlen) {
    if (!job) { set_err(err, errlen, "null job"); return PG_ERR_INVALID; }
    HIP_TRY(hipSetDevice(job->device));
    hipStream_t s = stream_ ? (hipStream_t)stream_ : job->stream;
    const uint32_t n = (uint32_t)job->contigs.size();
    HIP_TRY(hipMemsetAsync(job->zero_base, 0, job->zero_bytes, s));
    if (job->max_v > 0 && job->params.run_genotyping) {
        HIP_TRY(hipEventRecord(job->ev[0], s));
        pgk_launch_prep(job->d_contigs, n, job->max_v, job->tab, s);
        HIP_TRY(hipEventRecord(job->ev[1], s));
        pgk_launch_compact(job->d_contigs, n, s);
        HIP_TRY(hipEventRecord(job->ev[2], s));
        pgk_launch_records(job->d_contigs, n, job->max_v, s);
        HIP_TRY(hipEventRecord(job->ev[3], s));
        pgk_launch_sweep(job->d_contigs, n, job->hp_mask, 1, s);
        HIP_TRY(hipEventRecord(job->ev[4], s));
        if (!job->chunked) {
            pgk_launch_sweep(job->d_contigs, n, job->hp_mask, 2, s);
            HIP_TRY(hipEventRecord(job->ev[5], s));
            pgk_launch_bins(job->d_contigs, n, job->max_v, s);
            HIP_TRY(hipEventRecord(job->ev[6], s));
        } else {
            // chunk i: store-only sweep on s -> ev_sweep -> k_post on stream2 -> ev_post; the sweep of
            // chunk i+2 reuses scratch buffer i&1 and therefore waits for the posteriors of chunk i
            hipStream_t s2 = job->stream2;
            for (uint32_t i = 0; i < job->n_chunks; ++i) {
                const int b = (int)(i & 1u);
                if (i >= 2) HIP_TRY(hipStreamWaitEvent(s, job->ev_post[b], 0));
                pgk_launch_sweep_chunk(job->d_contigs, n, job->hp_mask, i, s);
                HIP_TRY(hipEventRecord(job->ev_sweep[b], s));
                HIP_TRY(hipStreamWaitEvent(s2, job->ev_sweep[b], 0));
                pgk_launch_post(job->d_contigs, n, job->chunk_cols, i, s2);
                HIP_TRY(hipEventRecord(job->ev_post[b], s2));
            }
            HIP_TRY(hipStreamWaitEvent(s, job->ev_post[0], 0));
            if (job->n_chunks > 1) HIP_TRY(hipStreamWaitEvent(s, job->ev_post[1], 0));
            HIP_TRY(hipEventRecord(job->ev[5], s));  // "k_sweep_phase2" = all chunks incl. their posteriors
            HIP_TRY(hipEventRecord(job->ev[6], s));  // (no k_bins in this mode)
        }
        HIP_TRY(hipGetLastError());
    } else if (job->max_v > 0) {
        // run_genotyping == false: only the ColumnIndexer part is meaningful (no likelihoods)
        pgk_launch_prep(job->d_contigs, n, job->max_v, job->tab, s);
        pgk_launch_compact(job->d_contigs, n, s);
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (job->max_v > 0 && job->params.run_genotyping) {
        for (int i = 0; i < PG_N_KERNEL_CLASSES; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, job->ev[i], job->ev[i + 1]));
            job->ms[i] = ms;
        }
    }
    std::vector<uint32_t> ncols(n), errs(n);
    HIP_TRY(hipMemcpy(ncols.data(), job->d_ncols, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(errs.data(), job->d_err, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    job->ran = true;
    for (uint32_t i = 0; i < n; ++i) {
        job->contigs[i].n_cols_host = ncols[i];
        if (errs[i] & PG_DEVERR_ALLELE_NOT_FOUND) {
            set_err(err, errlen, "contig %u: a path_allele value is not in the variant's allele list", i);
            return PG_ERR_INVALID;
        }
        if (errs[i] & PG_DEVERR_TOO_MANY_ALLELES) {
            set_err(err, errlen, "contig %u: a variant has more than %d alleles (device limit this release)", i, PG_MAX_ALLELES_PER_VARIANT);
            return PG_ERR_UNSUPPORTED;
        }
        if (errs[i] & PG_DEVERR_TOO_MANY_LOCAL) {
            set_err(err, errlen, "contig %u: a column has more than %d distinct alleles on the selected paths (device limit this release)", i, PG_WIDE_MAX);
            return PG_ERR_UNSUPPORTED;
        }
    }
    return PG_OK;
}

extern "C" int pg_job_fetch(pg_job* job, uint32_t ci, pg_contig_result* out, char* err, size_t errlen) {
    if (!job || !out || ci >= job->contigs.size()) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    if (!job->ran) { set_err(err, errlen, "pg_job_run has not been called"); return PG_ERR_INVALID; }
    HIP_TRY(hipSetDevice(job->device));
    const ContigHost& c = job->contigs[ci];
    out->n_columns = c.n_cols_host;
    if (c.V == 0) return PG_OK;
    if (out->lik && c.n_lik) HIP_TRY(hipMemcpy(out->lik, c.d.lik, c.n_lik * sizeof(double), hipMemcpyDeviceToHost));
    if (out->lik_exp) HIP_TRY(hipMemcpy(out->lik_exp, c.d.lik_exp, (size_t)c.V * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out->kept) HIP_TRY(hipMemcpy(out->kept, c.d.kept, c.V, hipMemcpyDeviceToHost));
    if (out->allele_present && c.sumA) HIP_TRY(hipMemcpy(out->allele_present, c.d.allele_present, c.sumA, hipMemcpyDeviceToHost));
    // reference src/hmm.cpp:94,106-109: set only when there is at least one column
    const bool fill = job->params.run_genotyping && c.n_cols_host > 0;
    if (out->n_kmers) {
        if (fill) memcpy(out->n_kmers, c.n_kmers.data(), (size_t)c.V * 2);
        else memset(out->n_kmers, 0, (size_t)c.V * 2);
    }
    if (out->coverage) {
        if (fill) memcpy(out->coverage, c.coverage.data(), (size_t)c.V * 2);
        else memset(out->coverage, 0, (size_t)c.V * 2);
    }
    return PG_OK;
}

extern "C" int pg_job_device_results(pg_job* job, uint32_t ci, void** d_lik, uint64_t* n_lik, void** d_lik_exp, uint64_t* n_variants) {
    if (!job || ci >= job->contigs.size()) return PG_ERR_INVALID;
    const ContigHost& c = job->contigs[ci];
    if (d_lik) *d_lik = c.d.lik;
    if (n_lik) *n_lik = c.n_lik;
    if (d_lik_exp) *d_lik_exp = c.d.lik_exp;
    if (n_variants) *n_variants = c.V;
    return PG_OK;
}

extern "C" int pg_job_profile_counters(pg_job* job, uint32_t ci, uint64_t out64[64]) {
    if (!job || !out64 || ci >= job->contigs.size()) return PG_ERR_INVALID;
    if (hipSetDevice(job->device) != hipSuccess) return PG_ERR_DEVICE;
    if (hipMemcpy(out64, job->contigs[ci].d.prof, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return PG_ERR_DEVICE;
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

extern "C" int pg_hmm_genotype_contig(const pg_contig_batch* batch, const pg_table* table, const pg_hmm_params* params,
                                      int device, pg_contig_result* out, char* err, size_t errlen) {
    if (!batch || !table || !params || !out) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    int rc = check_batch(batch, err, errlen);
    if (rc != PG_OK) return rc;
    if (params->run_phasing) {
        set_err(err, errlen, "run_phasing (Viterbi, reference src/hmm.cpp:112-173) is not on the device path");
        return PG_ERR_UNSUPPORTED;
    }
    pg_job* job = pg_job_create(device, 1, batch, table, params, err, errlen);
    if (!job) return PG_ERR_DEVICE;
    rc = pg_job_run(job, nullptr, err, errlen);
    if (rc == PG_OK) rc = pg_job_fetch(job, 0, out, err, errlen);
    pg_job_destroy(job);
    return rc;
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
    pg_job* job = pg_job_create(device, 1, batch, table, &p, err, errlen);
    if (!job) return PG_ERR_DEVICE;
    const uint32_t A = batch->allele_off[v + 1] - batch->allele_off[v];
    double* dm = nullptr; int* de = nullptr;
    int rc = PG_OK;
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
