/*
 * pg_sampler_oracle.c — TEST INFRASTRUCTURE ONLY (see pg_oracle.h).
 *
 * Plain-C restatement of the reference's HaplotypeSampler (src/haplotypesampler.cpp,
 * src/samplingemissions.cpp, src/samplingtransitions.cpp, SampledPaths in src/haplotypesampler.hpp)
 * on the flat pg_contig_batch.  Every function cites the reference lines whose behaviour it follows.
 * Pinned on the reference's own tests (tests/HaplotypeSamplerTest.cpp, SamplingEmissionsTest.cpp,
 * SamplingTransitionsTest.cpp) transcribed into tests/golden/sampler_known_answers.json.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pg_oracle.h"

#define UMAX 0xFFFFFFFFu

/* kmer on allele slot: reference src/kmerpath.cpp:33-48 */
static unsigned on_slot(const pg_contig_batch* b, uint32_t slot, uint32_t k) {
    uint32_t off = b->allele_kmer_off[slot];
    if (k < off || k >= off + 32u) return 0;
    return (b->allele_kmer_mask[slot] >> (k - off)) & 1u;
}

/* SamplingEmissions::SamplingEmissions, reference src/samplingemissions.cpp:9-37;
 * fraction_present_kmers_on_allele src/multiallelicuniquekmers.cpp:155-168 (count >= 3 is "present") */
void pgo_sampler_emission_costs(const pg_contig_batch* b, uint16_t* cost) {
    for (uint32_t v = 0; v < b->n_variants; ++v) {
        uint32_t k0 = b->kmer_off[v], K = b->kmer_off[v + 1] - k0;
        for (uint32_t s = b->allele_off[v]; s < b->allele_off[v + 1]; ++s) {
            if (b->allele_flags[s] & 1) { cost[s] = 50; continue; }
            /* total = KmerPath::nr_kmers (popcount of the window, src/kmerpath.cpp:50-55); present = k-mers of the
             * variant with a read count >= 3 that lie on the allele (src/multiallelicuniquekmers.cpp:155-162) */
            unsigned short total = (unsigned short)__builtin_popcount(b->allele_kmer_mask[s]), present = 0;
            for (uint32_t k = 0; k < K; ++k)
                if (b->kmer_count[k0 + k] >= 3 && on_slot(b, s, k)) present += 1;
            float fraction = total > 0 ? present / (float)total : 1.0f;
            /* `log10(fraction)` on a float picks the float overload in the reference (<cmath>, using namespace std) */
            if (fraction > 0.0) cost[s] = (unsigned short)(-10.0 * log10f(fraction));
            else cost[s] = 25;
        }
    }
}

/* SamplingTransitions::SamplingTransitions, reference src/samplingtransitions.cpp:5-14.
 * That file includes <cmath> without `using namespace std`, so its unqualified exp() and log10() are
 * the C DOUBLE functions (the long double arguments are narrowed) while the products around them are
 * long double.  Checked against the reference's own translation unit (oracle/_ref/libref_transitions.so,
 * tests/test_sampler.py).  Coincident positions give log10(0): the conversion of +inf to unsigned is
 * undefined in the reference; here it is pinned to 0xFFFFFFFF (the DP saturates). */
uint32_t pgo_sampler_transition_cost(uint64_t from_pos, uint64_t to_pos, double recombrate, uint32_t nr_paths,
                                     long double effective_N) {
    long double distance = (to_pos - from_pos) * 0.000004L * ((long double)recombrate) * effective_N;
    long double recomb_prob = (1.0L - exp((double)(-distance / (long double)nr_paths))) * (1.0L / (long double)nr_paths);
    double cost = -10.0 * log10((double)recomb_prob);
    if (!(cost < 4294967295.0)) return UMAX;
    if (cost < 0.0) return 0;
    return (unsigned int)cost;
}

/* HaplotypeSampler::get_column_minima, reference src/haplotypesampler.cpp:79-107 */
void pgo_sampler_column_minima(const uint32_t* column, const uint8_t* mask, uint32_t n, uint32_t out4[4]) {
    uint32_t first_val = UMAX, second_val = UMAX, first_id = UMAX, second_id = UMAX;
    for (uint32_t i = 0; i < n; ++i) {
        if (!mask[i]) continue;
        if (column[i] < first_val) {
            second_val = first_val; second_id = first_id;
            first_val = column[i]; first_id = i;
        } else if (column[i] < second_val && i != first_id) {
            second_val = column[i]; second_id = i;
        }
    }
    out4[0] = first_id; out4[1] = second_id; out4[2] = first_val; out4[3] = second_val;
}

static uint32_t slot_of(const pg_contig_batch* b, uint32_t v, uint16_t allele) {
    for (uint32_t s = b->allele_off[v]; s < b->allele_off[v + 1]; ++s)
        if (b->allele_id[s] == allele) return s;
    return UMAX; /* not listed at this variant (malformed input): cost 0, nothing to penalise */
}

/* HaplotypeSampler ctor + compute_viterbi_path + compute_viterbi_column,
 * reference src/haplotypesampler.cpp:20-77, :110-171, :173-294.  All columns are kept (the
 * sqrt(V) checkpointing of the reference only saves memory). */
int pgo_sampler_run(const pg_contig_batch* b, uint32_t size, double recombrate, long double effective_N,
                    uint16_t allele_penalty, uint32_t* sampled_paths, uint32_t* best_scores) {
    const uint32_t V = b->n_variants, P = b->n_paths;
    if (size < 1 || V == 0) return PG_OK;
    if (P < 2) return PG_ERR_INVALID;
    uint32_t sumA = b->allele_off[V];
    uint16_t* cost = (uint16_t*)malloc(sizeof(uint16_t) * (sumA ? sumA : 1));
    pgo_sampler_emission_costs(b, cost);
    uint32_t* col = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)V * P);
    uint32_t* bt = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)V * P);
    uint8_t* cur_mask = (uint8_t*)malloc(P);
    uint8_t* prev_mask = (uint8_t*)malloc(P);
    for (uint32_t it = 0; it < size; ++it) {
        for (uint32_t c = 0; c < V; ++c) {
            /* SampledPaths::mask_indexes, src/haplotypesampler.hpp:22-35 */
            memset(cur_mask, 1, P);
            for (uint32_t q = 0; q < it; ++q) cur_mask[sampled_paths[(size_t)q * V + c]] = 0;
            uint32_t* cc = col + (size_t)c * P;
            uint32_t* bc = bt + (size_t)c * P;
            uint32_t tcost = 0, m4[4] = {UMAX, UMAX, UMAX, UMAX};
            if (c > 0) {
                tcost = pgo_sampler_transition_cost(b->variant_pos[c - 1], b->variant_pos[c], recombrate, P, effective_N);
                pgo_sampler_column_minima(cc - P, prev_mask, P, m4);
            }
            for (uint32_t i = 0; i < P; ++i) {
                bc[i] = UMAX;
                if (!cur_mask[i]) { cc[i] = UMAX; continue; }
                uint32_t previous_cell = 0;
                if (c > 0) {
                    uint32_t hv = (i == m4[0]) ? m4[3] : m4[2], hid = (i == m4[0]) ? m4[1] : m4[0];
                    previous_cell = hv + tcost;
                    if (previous_cell < hv) previous_cell = UMAX;
                    bc[i] = hid;
                    if (prev_mask[i]) {
                        uint32_t same = (cc - P)[i] + 0u;
                        if (same < previous_cell) { previous_cell = same; bc[i] = i; }
                    }
                }
                uint32_t sl = slot_of(b, c, b->path_allele[(size_t)c * P + i]);
                uint32_t e = sl == UMAX ? 0u : cost[sl];
                cc[i] = previous_cell + e;
                if (cc[i] < previous_cell) cc[i] = UMAX;
            }
            memcpy(prev_mask, cur_mask, P);
        }
        /* best value in the last column: the FIRST minimum (src/haplotypesampler.cpp:129-139) */
        const uint32_t* last = col + (size_t)(V - 1) * P;
        uint32_t best_index = 0, best_value = last[0];
        for (uint32_t i = 1; i < P; ++i)
            if (last[i] < best_value) { best_value = last[i]; best_index = i; }
        if (best_scores) best_scores[it] = best_value;
        /* backtracking + penalties (src/haplotypesampler.cpp:147-170; SamplingEmissions::penalize
         * src/samplingemissions.cpp:43-49) */
        for (uint32_t c = V; c-- > 0;) {
            sampled_paths[(size_t)it * V + c] = best_index;
            uint32_t s = slot_of(b, c, b->path_allele[(size_t)c * P + best_index]);
            if (s != UMAX) {
                cost[s] = (uint16_t)(cost[s] + allele_penalty); /* unsigned short arithmetic, as in the reference */
                if (cost[s] > 25) cost[s] = 25;
            }
            if (c > 0) best_index = (bt + (size_t)c * P)[best_index];
        }
    }
    free(cost); free(col); free(bt); free(cur_mask); free(prev_mask);
    return PG_OK;
}
