/*
 * pangenie_sampler.h — C ABI of the MI355X-native HaplotypeSampler: the step the
 * reference runs right before the genotyping HMM on large panels (default above 100 paths,
 * reference src/commands.cpp:799-803): `size` passes of an integer (phred-cost) Viterbi over the H
 * panel paths pick a mosaic panel of `size` paths per variant (reference
 * src/haplotypesampler.cpp:20-77, :110-294; emission costs src/samplingemissions.cpp:9-45,
 * transition cost src/samplingtransitions.cpp:5-23).
 *
 * Input = the same flat batch as the genotyping path (include/pangenie_hmm.h: pg_contig_batch) over
 * ALL paths of the panel (n_paths = UniqueKmers::get_nr_paths()).  Everything is integer work: results
 * (sampled path ids, best scores) are bit-exact with the reference.  Emission and transition costs
 * are formed on the host exactly as the reference forms them (float / long double, then truncation);
 * the passes themselves — column minima, DP update, backtrace, penalties — run on the GPU.
 */
#ifndef PANGENIE_SAMPLER_H
#define PANGENIE_SAMPLER_H

#include "pangenie_hmm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SamplingEmissions ctor (src/samplingemissions.cpp:9-37) for every allele slot of every variant:
 * 50 for an undefined allele, (unsigned short)(-10 log10(fraction of the allele's k-mers with a read
 * count >= 3)), 25 if none is present; an allele without k-mers has fraction 1 -> 0.  Host. */
int pg_sampler_emission_costs(const pg_contig_batch* panel, uint16_t* cost_sumA);
/* SamplingTransitions ctor (src/samplingtransitions.cpp:5-14): (unsigned int)(-10 log10 q),
 * q = (1 - e^(-d/H)) / H in long double.  Host. */
uint32_t pg_sampler_transition_cost(uint64_t from_pos, uint64_t to_pos, double recombrate,
                                    uint32_t nr_paths, long double effective_N);
/* HaplotypeSampler::get_column_minima (src/haplotypesampler.cpp:79-107) on the device: smallest and
 * second smallest unmasked entry, ties to the smaller index.  out4 = {first_id, second_id, first_val,
 * second_val} (ids 0xFFFFFFFF when there is none).  Unit-level entry for the reference's own tests. */
int pg_sampler_column_minima(const uint32_t* column, const uint8_t* mask, uint32_t n, int device,
                             uint32_t out4[4], char* err, size_t errlen);
/* HaplotypeSampler ctor body for one contig: `size` Viterbi passes.  sampled_paths[s * V + v] = path id
 * picked by pass s at variant v (SampledPaths::sampled_paths, src/haplotypesampler.hpp:17-20);
 * best_scores[s] = DP score of pass s (may be NULL).  The caller applies UniqueKmers::update_paths
 * (and appends the reference path 0 when add_reference is set) as the reference's ctor does
 * (src/haplotypesampler.cpp:44, :296-309). */
int pg_sampler_run(const pg_contig_batch* panel, uint32_t size, double recombrate, long double effective_N,
                   uint16_t allele_penalty, int device, uint32_t* sampled_paths, uint32_t* best_scores,
                   char* err, size_t errlen);
/* The same for several contigs at once — one workgroup per contig and pass, which is how the sampler fills
 * more than one CU (the passes and the columns of a contig are sequential by construction).  panels[g],
 * sampled_paths[g] ([size * V_g]) and best_scores[g] ([size], the array or single entries may be NULL)
 * belong to contig g; contigs without variants are skipped. */
int pg_sampler_run_batch(const pg_contig_batch* panels, uint32_t n_contigs, uint32_t size, double recombrate,
                         long double effective_N, uint16_t allele_penalty, int device,
                         uint32_t* const* sampled_paths, uint32_t* const* best_scores, char* err, size_t errlen);
/* Sampler -> UniqueKmers::update_paths -> genotyping job with the panel staying on the device (the reference's
 * constructor tail, src/haplotypesampler.cpp:44, :296-309; src/biallelicuniquekmers.cpp:223-260,
 * src/multiallelicuniquekmers.cpp:195-232, followed by run_genotyping on the sampled panel, src/commands.cpp:138-152):
 * `size` passes over every contig of `panels`, then the panel reduced — on the GPU — to the sampled paths (+ the
 * reference path 0 when add_reference), the alleles they carry and the k-mers on those alleles, and a resident job
 * (include/pangenie_hmm.h: pg_job_run / pg_job_fetch) over the reduced panel: chain g = contig g, size (+ 1) paths.
 * Only two counts per variant travel to the host (the job's memory is planned from them).  sampled_paths /
 * best_scores as in pg_sampler_run_batch, or NULL.  The reduced panel (its allele ids give the genotype bins their
 * meaning) is read back with pg_job_fetch_panel.  At most 1024 kept paths, 1024 alleles and 2048 k-mers per variant. */
int pg_sampler_then_job(const pg_contig_batch* panels, uint32_t n_contigs, uint32_t size, int add_reference,
                        double sampling_recombrate, long double sampling_effective_N, uint16_t allele_penalty,
                        const pg_table* table, const pg_hmm_params* params, int device,
                        uint32_t* const* sampled_paths, uint32_t* const* best_scores,
                        pg_job** out_job, char* err, size_t errlen);
/* Kernel milliseconds of the last pg_sampler_run[_batch] of the calling thread, summed over the passes:
 * [0] cost expansion, [1] forward passes, [2] backtraces; *kernel (may be NULL) = waves per workgroup of
 * the relative-value kernel, 0 when the general (saturating) kernel ran. */
int pg_sampler_last_ms(double out3[3], int* kernel);

#ifdef __cplusplus
}
#endif
#endif /* PANGENIE_SAMPLER_H */
