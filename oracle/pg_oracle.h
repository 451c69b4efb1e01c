/*
 * pg_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, x87 80-bit `long double`, same operation order) of
 * the reference's genotyping hot path, written from the reference's behaviour:
 *   src/hmm.cpp, src/emissionprobabilitycomputer.cpp,
 *   src/transitionprobabilitycomputer.cpp, src/columnindexer.cpp,
 *   src/probabilitytable.cpp, src/copynumber.cpp   (eblerjana/pangenie v4.2.1).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link, load or call anything in this directory — as the checker, never as the
 * product.  The product (pangenie_amd/csrc) has no CPU fallback.
 *
 * Parity pin: the reference's HMM cannot be compiled in this image (its
 * translation units include <cereal/...>, which is not installed, and no
 * stand-in headers may be written).  What does compile from its own sources is
 * built by `make ref` into oracle/_ref/ and used as the pin where it reaches:
 * src/probabilitytable.cpp + src/copynumber.cpp (the table: bit for bit) and
 * src/samplingtransitions.cpp (the sampler's recombination cost).  The rest of
 * the oracle is pinned on the reference's own known-answer unit tests
 * (tests/HMMTest.cpp, EmissionProbabilityComputerTest.cpp,
 * TransitionProbabilityComputerTest.cpp, ProbabilityTableTest.cpp,
 * CopyNumberTest.cpp, ColumnIndexerTest.cpp), transcribed as data in
 * tests/golden/ and checked at the reference's own tolerance (1e-7 absolute,
 * tests/utils.cpp:9-11) by tests/test_oracle_golden.py.
 */
#ifndef PG_ORACLE_H
#define PG_ORACLE_H

#include "../include/pangenie_hmm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ProbabilityTable restatement (reference src/probabilitytable.cpp:21-85). */
typedef struct pgo_table pgo_table;
pgo_table* pgo_table_create(uint16_t cov_min, uint16_t cov_max, uint16_t count_max,
                            long double regularization);
pgo_table* pgo_table_create_default(void);
/* modify_probability(cov, count, CopyNumber(p0,p1,p2)) — src/probabilitytable.cpp:67-73 */
int  pgo_table_modify(pgo_table* t, uint16_t cov, uint16_t count,
                      long double p0, long double p1, long double p2);
/* get_probability(cov,count).get_probability_of(0..2) — src/probabilitytable.cpp:47-53 */
void pgo_table_get(const pgo_table* t, uint16_t cov, uint16_t count, long double out3[3]);
void pgo_table_destroy(pgo_table* t);

/* CopyNumber(cn0,cn1,cn2,regularization) -> get_probability_of(0..2)
 * (reference src/copynumber.cpp:22-41). */
void pgo_copynumber_regularized(long double cn0, long double cn1, long double cn2,
                                long double reg, long double out3[3]);

/* TransitionProbabilityComputer ctor (reference src/transitionprobabilitycomputer.cpp:8-19,33-39). */
void pgo_transition_probs(uint64_t from_pos, uint64_t to_pos, double recombrate,
                          uint32_t nr_paths, int uniform, long double effective_N,
                          long double out3[3]);

/* EmissionProbabilityComputer (reference src/emissionprobabilitycomputer.cpp:9-53):
 * A x A table over all allele slots of variant v after the all_zeros rule. */
int pgo_emission_table(const pg_contig_batch* b, const pgo_table* t, uint32_t v,
                       long double* out_AxA, int32_t* all_zeros);

/* HMM::HMM with run_genotyping (reference src/hmm.cpp:25-110,175-405).
 * Outputs are the reference's exact unnormalised long double likelihoods
 * (normalize=false); layout as pg_contig_result but lik is long double. */
typedef struct pgo_result {
    long double* lik;         /* [geno_off[V]] */
    uint8_t*  kept;           /* [V]    */
    uint8_t*  allele_present; /* [sumA] */
    uint16_t* n_kmers;        /* [V]    */
    uint16_t* coverage;       /* [V]    */
    uint32_t  n_columns;
    uint32_t  reserved;
} pgo_result;

int pgo_genotype_contig(const pg_contig_batch* b, const pgo_table* t,
                        const pg_hmm_params* p, pgo_result* out);

/* HMM with run_phasing: Viterbi path over ordered path pairs (reference src/hmm.cpp:112-173, :408-511).
 * form 0 = the reference's O(H^4) loop, form 1 = the same maxima in O(H^2) (see pg_oracle.c).
 * hap1/hap2 [V]: haplotype alleles at kept variants; n_kmers / coverage [V]: set at the COLUMN index (sic). */
int pgo_viterbi_contig(const pg_contig_batch* b, const pgo_table* t, const pg_hmm_params* p, int form,
                       uint16_t* hap1, uint16_t* hap2, uint8_t* kept, uint32_t* n_columns,
                       uint16_t* n_kmers, uint16_t* coverage);

/* HaplotypeSampler restatement (pg_sampler_oracle.c; reference src/haplotypesampler.cpp,
 * src/samplingemissions.cpp, src/samplingtransitions.cpp): same flat batch, over all panel paths. */
void pgo_sampler_emission_costs(const pg_contig_batch* b, uint16_t* cost_sumA);
uint32_t pgo_sampler_transition_cost(uint64_t from_pos, uint64_t to_pos, double recombrate, uint32_t nr_paths,
                                     long double effective_N);
void pgo_sampler_column_minima(const uint32_t* column, const uint8_t* mask, uint32_t n, uint32_t out4[4]);
int pgo_sampler_run(const pg_contig_batch* b, uint32_t size, double recombrate, long double effective_N,
                    uint16_t allele_penalty, uint32_t* sampled_paths, uint32_t* best_scores);

/* geno_off helper (same rule as pg_hmm_geno_offsets). */
void pgo_geno_offsets(const pg_contig_batch* b, uint64_t* geno_off);

#ifdef __cplusplus
}
#endif
#endif
