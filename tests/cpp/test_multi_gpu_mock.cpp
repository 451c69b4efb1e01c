// test_multi_gpu_mock.cpp — pangenie::run_contigs_multi_gpu without a GPU: the device calls are replaced AT THE C-ABI
// SEAM (pg_job_new / run / fetch / destroy, pg_comm_init_all / destroy, pg_hmm_gather_to_host are defined in this
// executable and, exported with -rdynamic, take precedence over libpangenie_hmm.so's for libpangenie_host.so's calls),
// so that what runs here is exactly the host side of the multi-GPU job loop: flattening, the longest-processing-time
// plan, one job per device on its own thread, the ONE exchange with its per-rank block sizes and offsets, and the
// reassembly of every task's GenotypingResults.  The mock "posterior" of a genotype bin is a function of (the
// contig's first position, variant count, bin index) alone, so the expected likelihood of every (task, variant,
// genotype) is known whatever device the task was planned on.  What stays untested without hardware is RCCL itself.
// Test infrastructure (tests/ only).  Reference: the job loop of src/commands.cpp:955-978.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <vector>

#include "../../include/pangenie_hmm.h"
#include "../../pangenie_amd/host/pangenie_host.hpp"

// ------------------------------------------------------------------ the mocked seam
struct pg_job {
    int device = 0;
    std::vector<pg_contig_batch> batches;
    std::vector<std::vector<uint64_t>> goff;
    std::vector<double> lik;        // packed, chain after chain
    std::vector<int32_t> lexp;
    bool ran = false;
};
struct pg_comm { int rank = 0, device = 0; };

static std::mutex g_mu;
static std::vector<std::pair<int, uint32_t>> g_jobs_made;   // (device, chains)
static int g_gathers = 0, g_fetches = 0;
static std::vector<uint64_t> g_last_plan;

static void mock_bin(uint64_t key, uint64_t g, double& m, int32_t& e) {   // deterministic, device independent
    uint64_t x = key * 0x9E3779B97F4A7C15ull + g * 0xD1B54A32D192ED03ull + 12345;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    m = 0.5 + (double)(x % 1000003) / 2000006.0;
    e = -(int32_t)(x % 37);
}
static uint64_t key_of(const pg_contig_batch& b) { return b.n_variants ? (uint64_t)b.variant_pos[0] * 1000003ull + b.n_variants : 7; }

extern "C" {
int pg_job_new(int device, uint32_t n_contigs, const pg_contig_batch* batches, const pg_table*, const pg_hmm_params* p, pg_job** out, char*, size_t) {
    if (!p || !p->run_genotyping) return PG_ERR_INVALID;
    pg_job* j = new pg_job;
    j->device = device;
    j->batches.assign(batches, batches + n_contigs);
    for (uint32_t c = 0; c < n_contigs; ++c) {
        std::vector<uint64_t> g((size_t)batches[c].n_variants + 1, 0);
        pg_hmm_geno_offsets(&batches[c], g.data());
        j->goff.push_back(g);
    }
    { std::lock_guard<std::mutex> lk(g_mu); g_jobs_made.push_back({device, n_contigs}); }
    *out = j;
    return PG_OK;
}
int pg_job_run(pg_job* j, void*, char*, size_t) {
    j->lik.clear(); j->lexp.clear();
    for (size_t c = 0; c < j->batches.size(); ++c) {
        const uint64_t n = j->goff[c].back(), key = key_of(j->batches[c]);
        for (uint64_t g = 0; g < n; ++g) { double m; int32_t e; mock_bin(key, g, m, e); j->lik.push_back(m); j->lexp.push_back(e); }
    }
    j->ran = true;
    return PG_OK;
}
int pg_job_fetch(pg_job* j, uint32_t ci, pg_contig_result* out, char*, size_t) {
    if (!j->ran || ci >= j->batches.size()) return PG_ERR_INVALID;
    uint64_t off = 0;
    for (uint32_t c = 0; c < ci; ++c) off += j->goff[c].back();
    const uint64_t n = j->goff[ci].back();
    if (out->lik) memcpy(out->lik, j->lik.data() + off, n * sizeof(double));
    if (out->lik_exp) memcpy(out->lik_exp, j->lexp.data() + off, n * sizeof(int32_t));
    { std::lock_guard<std::mutex> lk(g_mu); g_fetches += 1; }
    return PG_OK;
}
void pg_job_destroy(pg_job* j) { delete j; }
int pg_comm_init_all(int n, const int* devices, pg_comm** out, char*, size_t) {
    for (int i = 0; i < n; ++i) { out[i] = new pg_comm; out[i]->rank = i; out[i]->device = devices[i]; }
    return PG_OK;
}
void pg_comm_destroy(pg_comm* c) { delete c; }
int pg_hmm_gather_to_host(int n_local, pg_comm* const* comms, pg_job* const* jobs, int root, const uint64_t* n_lik_per_rank,
                          double* h_lik_all, int32_t* h_exp_all, char* err, size_t errlen) {
    if (root != 0) return PG_ERR_INVALID;
    uint64_t off = 0;
    { std::lock_guard<std::mutex> lk(g_mu); g_gathers += 1; g_last_plan.assign(n_lik_per_rank, n_lik_per_rank + n_local); }
    for (int r = 0; r < n_local; ++r) {
        if (!comms[r] || comms[r]->rank != r) { snprintf(err, errlen, "bad communicator %d", r); return PG_ERR_INVALID; }
        const uint64_t n = n_lik_per_rank[r];
        if (jobs[r]) {
            if (jobs[r]->lik.size() != n) { snprintf(err, errlen, "rank %d holds %zu bins, the plan says %llu", r, jobs[r]->lik.size(), (unsigned long long)n); return PG_ERR_INVALID; }
            memcpy(h_lik_all + off, jobs[r]->lik.data(), n * sizeof(double));
            memcpy(h_exp_all + off, jobs[r]->lexp.data(), n * sizeof(int32_t));
        } else if (n != 0) { snprintf(err, errlen, "rank %d has no job but %llu bins planned", r, (unsigned long long)n); return PG_ERR_INVALID; }
        off += n;
    }
    return PG_OK;
}
}  // extern "C"

// ------------------------------------------------------------------ the test
using namespace pangenie;
static int g_checks = 0, g_failed = 0;
#define CHECK(c) do { ++g_checks; if (!(c)) { ++g_failed; printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

static std::vector<std::shared_ptr<UniqueKmers>> make_contig(size_t V, unsigned H, size_t first_pos, unsigned seed) {
    std::vector<std::shared_ptr<UniqueKmers>> out;
    uint64_t x = seed * 2654435761u + 17;
    auto rnd = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(x >> 33); };
    for (size_t v = 0; v < V; ++v) {
        const bool multi = rnd() % 4 == 0;
        std::vector<unsigned short> alleles(H);
        for (unsigned p = 0; p < H; ++p) alleles[p] = (unsigned short)(multi ? rnd() % 3 : rnd() % 2);
        alleles[0] = 0; alleles[1] = 1;   // (a column: an ALT allele on a path)
        std::shared_ptr<UniqueKmers> u;
        if (multi) u = std::make_shared<MultiallelicUniqueKmers>(first_pos + 700 * v, alleles);
        else u = std::make_shared<BiallelicUniqueKmers>(first_pos + 700 * v, alleles);
        for (unsigned k = 0; k < 6; ++k) { std::vector<unsigned short> on = {(unsigned short)(k % (multi ? 3 : 2))}; u->insert_kmer((unsigned short)(rnd() % 40), on); }
        u->set_coverage((unsigned short)(25 + rnd() % 6));
        out.push_back(u);
    }
    return out;
}

int main() {
    const size_t sizes[] = {40, 5, 23, 0, 17, 31, 9, 12, 28};
    const unsigned paths[] = {6, 4, 8, 4, 6, 5, 4, 7, 6};
    const size_t T = sizeof(sizes) / sizeof(sizes[0]);
    std::vector<std::vector<std::shared_ptr<UniqueKmers>>> contigs;
    for (size_t t = 0; t < T; ++t) contigs.push_back(make_contig(sizes[t], paths[t], 1000 + 1000000 * t, (unsigned)t + 1));
    ProbabilityTable table(6, 108, 54, 0.01L);
    auto run = [&](const std::vector<int>& devices) {
        std::vector<ContigTask> tasks(T);
        for (size_t t = 0; t < T; ++t) tasks[t].unique_kmers = &contigs[t];
        return run_contigs_multi_gpu(tasks, &table, 1.26, false, 25000.0L, devices);
    };
    // expected likelihood of (task t, variant v, alleles a <= b), straight from the mock's rule
    auto expect = [&](size_t t, size_t v, unsigned short a, unsigned short b, const std::vector<uint64_t>& goff, uint32_t A, uint32_t ia, uint32_t ib) {
        (void)a; (void)b;
        const uint64_t idx = goff[v] + (uint64_t)ia * A - (uint64_t)ia * (ia - 1) / 2 + (ib - ia);
        double m; int32_t e;
        mock_bin((uint64_t)(1000 + 1000000 * t) * 1000003ull + sizes[t], idx, m, e);
        return ldexpl((long double)m, e);
    };
    for (const std::vector<int>& devices : {std::vector<int>{0}, std::vector<int>{0, 1}, std::vector<int>{0, 1, 2}, std::vector<int>{3, 1, 0, 2, 5, 4, 7, 6}, std::vector<int>(12, 0)}) {
        for (size_t i = 0; i < devices.size() && devices.size() == 12; ++i) const_cast<std::vector<int>&>(devices)[i] = (int)i;   // (more devices than tasks)
        { std::lock_guard<std::mutex> lk(g_mu); g_jobs_made.clear(); g_gathers = 0; g_fetches = 0; g_last_plan.clear(); }
        const std::vector<std::vector<GenotypingResult>> out = run(devices);
        const size_t D = devices.size();
        printf("%zu device(s): %zu job(s), %d gather(s), %d fetch(es)\n", D, g_jobs_made.size(), g_gathers, g_fetches);
        CHECK(out.size() == T);
        // one job per device that got tasks, every task in exactly one job, ONE exchange (none on a single device)
        uint32_t chains = 0;
        std::set<int> used;
        for (const auto& j : g_jobs_made) { chains += j.second; CHECK(used.insert(j.first).second); }
        CHECK(chains == T && g_jobs_made.size() <= D && g_jobs_made.size() == std::min(D, T));
        CHECK(D == 1 ? (g_gathers == 0 && g_fetches == (int)T) : (g_gathers == 1 && g_fetches == 0));
        if (D > 1) {   // the plan's block sizes add up to all bins; the heaviest task (t = 0: 40 variants x 6^2) sits alone on its device when there are enough
            uint64_t total = 0, want = 0;
            for (uint64_t n : g_last_plan) total += n;
            for (size_t t = 0; t < T; ++t) {
                FlatContig f; std::vector<uint64_t> g(sizes[t] + 1, 0);
                flatten(&contigs[t], nullptr, f); pg_hmm_geno_offsets(&f.batch, g.data()); want += g.back();
            }
            CHECK(total == want && g_last_plan.size() == D);
        }
        for (size_t t = 0; t < T; ++t) {
            CHECK(out[t].size() == sizes[t]);
            FlatContig f;
            flatten(&contigs[t], nullptr, f);
            std::vector<uint64_t> goff(sizes[t] + 1, 0);
            pg_hmm_geno_offsets(&f.batch, goff.data());
            for (size_t v = 0; v < out[t].size(); ++v) {
                const uint32_t a0 = f.allele_off[v], A = f.allele_off[v + 1] - a0;
                CHECK(out[t][v].coverage() == contigs[t][v]->get_coverage() && out[t][v].nr_unique_kmers() == 6);
                for (uint32_t ia = 0; ia < A; ++ia)
                    for (uint32_t ib = ia; ib < A; ++ib) {
                        const unsigned short a = f.allele_id[a0 + ia], b = f.allele_id[a0 + ib];
                        const long double got = out[t][v].get_genotype_likelihood(a, b), want = expect(t, v, a, b, goff, A, ia, ib);
                        // (alleles no path carries have no bin: 0)
                        bool pa = false, pb = false;
                        for (unsigned p = 0; p < paths[t]; ++p) { pa = pa || f.path_allele[v * paths[t] + p] == a; pb = pb || f.path_allele[v * paths[t] + p] == b; }
                        if (pa && pb) CHECK(got == want); else CHECK(got == 0.0L);
                    }
            }
        }
    }
    printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
