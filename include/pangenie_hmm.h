/*
 * pangenie_hmm.h — C ABI of the MI355X-native PanGenie genotyping hot path.
 *
 * This is the drop-in boundary: everything the reference does inside
 *     HMM::HMM(...)                      (reference src/hmm.hpp:38, src/hmm.cpp:25-63)
 * for run_genotyping=true (and run_phasing=true: the Viterbi path, src/hmm.cpp:112-173, 408-511,
 * pangenie_amd/csrc/pg_viterbi.hip) — ColumnIndexer (src/columnindexer.cpp:8-33),
 * EmissionProbabilityComputer (src/emissionprobabilitycomputer.cpp:9-53),
 * TransitionProbabilityComputer (src/transitionprobabilitycomputer.cpp:8-19),
 * forward/backward columns + posterior accumulation (src/hmm.cpp:76-110, 175-405) —
 * happens behind pg_hmm_genotype_contig() / pg_job_run() on the GPU.
 *
 * Plain C: pointers + sizes, no STL, no exceptions, no torch types.  All host
 * pointers are caller-owned and only read during the call (same ownership rule
 * as the reference's borrowed unique_kmers / probabilities / only_paths
 * pointers, src/hmm.cpp:25-31).  Error convention: 0 = ok, negative = error
 * with a message in `err` (the C++ adapter rethrows std::runtime_error, which
 * is what the reference throws, e.g. src/columnindexer.cpp:18-22).
 *
 * Numeric contract: the device computes in fp64 and returns every unnormalised
 * genotype likelihood as  lik[g] * 2^lik_exp[g]  — one fp64 mantissa in [0.5,1)
 * (or 0) and one int32 exponent PER GENOTYPE BIN; the host rebuilds the
 * reference's 80-bit `long double` value with ldexpl() and does normalisation /
 * GT / GQ in long double (reference src/genotypingresult.cpp:118-210).
 * Guaranteed range: every bin holds 1e-6 relative (observed 1e-13) down to the
 * reference's own long double underflow (2^-16445), however many decades it lies
 * below its variant's largest bin, for every chain whose transitions mix
 * (recombination probability q > 0 between neighbouring columns: any gap of
 * >= 1 bp at recombrate * effective_N >= 1e-12).  The emission of a bin is
 * applied once, to the finished bin, as (mantissa, exponent); the stored columns
 * carry a bounded dynamic range (>= q^2 of their sum).  Chains WITHOUT mixing
 * (recombrate == 0 or effective_N == 0: every state evolves on its own) keep
 * 2^-1400 relative to a column's sum in fp64 where the reference keeps 2^-16445.
 */
#ifndef PANGENIE_HMM_H
#define PANGENIE_HMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_ERR_INVALID (-1)     /* bad argument / malformed batch            */
#define PG_ERR_NO_PATHS (-2)    /* "column is not covered by any paths"      */
#define PG_ERR_UNSUPPORTED (-3) /* device limits: > 1024 selected paths (> 64 with run_phasing), > 256 alleles per variant */
#define PG_ERR_DEVICE (-4)      /* HIP runtime error / no GPU                */
#define PG_ERR_NOMEM (-5)

/* ------------------------------------------------------------------ *
 *  Input: one (contig, path-subset) chain, flattened to SoA.
 *  Replaces std::vector<std::shared_ptr<UniqueKmers>>* + only_paths
 *  (reference src/uniquekmers.hpp:22-69, src/hmm.hpp:38).
 * ------------------------------------------------------------------ */
typedef struct pg_contig_batch {
    uint32_t n_variants; /* V : unique_kmers->size()                                        */
    uint32_t n_paths;    /* H : number of SELECTED paths (ColumnIndexer::nr_paths,          */
                         /*     columnindexer.cpp:23,51-53); states per column = H*H        */
    const uint64_t* variant_pos;  /* [V]   UniqueKmers::get_variant_position()              */
    const uint16_t* coverage;     /* [V]   UniqueKmers::get_coverage() (float -> u16)       */
    const uint32_t* kmer_off;     /* [V+1] prefix offsets into kmer_count                   */
    const uint16_t* kmer_count;   /* [sumK] UniqueKmers::get_readcount_of(k)                */
    const uint32_t* allele_off;   /* [V+1] prefix offsets into the allele_* arrays          */
    const uint16_t* allele_id;    /* [sumA] get_allele_ids(): ALL alleles of the object,    */
                                  /*        ascending (std::map order)                      */
    const uint8_t*  allele_flags; /* [sumA] bit0 = is_undefined_allele(a)                   */
    const uint16_t* allele_kmer_off;  /* [sumA] KmerPath::offset  (kmerpath.hpp:33)         */
    const uint32_t* allele_kmer_mask; /* [sumA] KmerPath::kmers   (16-bit masks widened):   */
                                  /* kmer k on allele a <=> 0<=k-off<32 && mask>>(k-off)&1  */
                                  /* (kmerpath.cpp:33-48)                                   */
    const uint16_t* path_allele;  /* [V*H] allele id carried by selected path p at variant  */
                                  /* v: UniqueKmers::get_allele(paths[p])                   */
} pg_contig_batch;

/* HMM constructor arguments (reference src/hmm.hpp:38, call site src/commands.cpp:160). */
typedef struct pg_hmm_params {
    long double effective_N; /* default 25000.0L; production 1e-5 (pangenie-genotype.cpp:33-45) */
    double recombrate;       /* default 1.26                                                     */
    int32_t uniform;         /* uniform transition probabilities                                 */
    int32_t run_genotyping;  /* forward-backward                                                 */
    int32_t run_phasing;     /* Viterbi path over ordered path pairs (src/hmm.cpp:112-173, 408-511): */
                             /* fills haplotype_1 / haplotype_2; at most 64 selected paths          */
    int32_t reserved;        /* call flags of the one-shot entry point: PG_CALL_ANNOUNCED; else 0                */
} pg_hmm_params;
#define PG_CALL_ANNOUNCED 1  /* this pg_hmm_genotype_contig call was announced with pg_hmm_announce(device) */

/* ------------------------------------------------------------------ *
 *  Output, caller-allocated.  Genotype bins of variant v live at
 *  lik[geno_off[v] .. geno_off[v+1]) with geno_off from
 *  pg_hmm_geno_offsets(): bin index of allele SLOTS (a<=b) of an A-allele
 *  variant is  a*A - a*(a-1)/2 + (b-a)  (lexicographic (a,b), i.e. the
 *  iteration order of the reference's std::map<pair<u16,u16>,long double>,
 *  src/genotypingresult.hpp:84).  A bin is a key of the reference's map iff
 *  kept[v] && allele_present[a] && allele_present[b] (src/hmm.cpp:368).
 * ------------------------------------------------------------------ */
typedef struct pg_contig_result {
    double*   lik;            /* [geno_off[V]] unnormalised likelihood, mantissa in [0.5,1) or 0 */
    int32_t*  lik_exp;        /* [geno_off[V]] power-of-two exponent per bin: L = lik * 2^lik_exp */
    uint8_t*  kept;           /* [V] 1 = variant is an HMM column (columnindexer.cpp:24-31)   */
    uint8_t*  allele_present; /* [sumA] 1 = allele slot occurs on a selected path             */
    uint16_t* n_kmers;        /* [V] GenotypingResult::set_unique_kmers (hmm.cpp:106-109)     */
    uint16_t* coverage;       /* [V] GenotypingResult::set_coverage                           */
    uint32_t  n_columns;      /* C = ColumnIndexer::size()                                    */
    uint32_t  reserved;
    /* run_phasing only (may be NULL): alleles of the Viterbi path's two haplotypes at kept variants  */
    /* (GenotypingResult::add_first/second_haplotype_allele, src/hmm.cpp:146-162), 0 elsewhere.  With */
    /* run_phasing the reference also sets unique_kmers / coverage of results[c] for every COLUMN     */
    /* index c < C from variant c (sic, src/hmm.cpp:164-165): n_kmers / coverage above follow that.   */
    uint16_t* haplotype_1;    /* [V] */
    uint16_t* haplotype_2;    /* [V] */
} pg_contig_result;

/* geno_off[V+1] from allele_off: geno_off[v+1]-geno_off[v] = A_v*(A_v+1)/2. */
int pg_hmm_geno_offsets(const pg_contig_batch* batch, uint64_t* geno_off);

/* ------------------------------------------------------------------ *
 *  ProbabilityTable (reference src/probabilitytable.hpp:13-29).  Built on the
 *  host in long double exactly as the reference does before the HMM runs
 *  (src/commands.cpp:846); the device receives it as (mantissa, exponent)
 *  pairs.  Out-of-range (coverage,count) pairs are evaluated on the fly
 *  (src/probabilitytable.cpp:47-53) — on the device in fp64.
 * ------------------------------------------------------------------ */
typedef struct pg_table pg_table;
pg_table* pg_table_create(uint16_t cov_min, uint16_t cov_max, uint16_t count_max,
                          long double regularization);            /* probabilitytable.cpp:28-45 */
pg_table* pg_table_create_default(void);                           /* probabilitytable.cpp:21-26 */
int  pg_table_modify(pg_table* t, uint16_t coverage, uint16_t count,
                     long double p0, long double p1, long double p2); /* :67-73, test hook */
int  pg_table_get(const pg_table* t, uint16_t coverage, uint16_t count,
                  long double out3[3]);                             /* :47-53 */
void pg_table_destroy(pg_table* t);

/* ------------------------------------------------------------------ *
 *  One-shot blocking call = the body of HMM::HMM for one (contig, subset).
 *  Thread-safe and re-entrant, the way the reference calls its constructor: N thread-pool workers at
 *  a time, one call per (contig x subset), sharing the table read-only (src/commands.cpp:949-978,
 *  :155-185).  Calls that are in flight together on one device with the same table and parameters are
 *  MERGED into one multi-chain device job by the first of them (the others sleep until their results
 *  are in their buffers); chains of a job are independent, so each caller gets exactly what it would
 *  have got alone, and an error of one caller's batch is that caller's alone (the merged job is then
 *  re-run call by call).  Environment: PG_COALESCE=0 (off), PG_COALESCE_WAIT_MS (250: the bound on waiting
 *  for an announced call); the join window (300 us), the merged jobs in flight per device (2) and the batch
 *  size (256 calls) are constants.
 *  Device arenas of finished calls are pooled for the next ones (pg_hmm_release_cache frees them).
 * ------------------------------------------------------------------ */
int pg_hmm_device_count(void);
const char* pg_hmm_version(void);
int pg_hmm_genotype_contig(const pg_contig_batch* batch, const pg_table* table,
                           const pg_hmm_params* params, int device,
                           pg_contig_result* out, char* err, size_t errlen);
/* Optional: a worker that WILL call pg_hmm_genotype_contig on `device` shortly (it is still
 * flattening its UniqueKmers) says so; a leader about to launch waits for announced calls (bounded by
 * PG_COALESCE_WAIT_MS).  The call itself then carries PG_CALL_ANNOUNCED in params->reserved; a worker
 * that gives up before calling retracts. */
void pg_hmm_announce(int device);
void pg_hmm_retract(int device);
/* {merged jobs launched, calls served, largest merge} since the process started */
int  pg_hmm_coalesce_stats(uint64_t out3[3]);

/* ------------------------------------------------------------------ *
 *  Resident job API: upload once, run many times (benchmarks, pipelines,
 *  multi-contig batches that share one launch).  All contigs of a job run
 *  concurrently (one persistent workgroup per chain direction).
 * ------------------------------------------------------------------ */
typedef struct pg_job pg_job;
/* (The six big arrays of a batch — kmer_count, allele_id, allele_flags, allele_kmer_off, allele_kmer_mask,
 * path_allele — may also be DEVICE pointers of `device`; the offset arrays, positions and coverage are read on the host.) */
pg_job* pg_job_create(int device, uint32_t n_contigs, const pg_contig_batch* batches,
                      const pg_table* table, const pg_hmm_params* params,
                      char* err, size_t errlen);
/* Runs the whole device path (prep -> forward/backward sweep -> bins) and
 * synchronises.  `stream` = hipStream_t to launch on, NULL = the job's own. */
int  pg_job_run(pg_job* job, void* stream, char* err, size_t errlen);
/* Copies contig c's results to host buffers. */
int  pg_job_fetch(pg_job* job, uint32_t contig, pg_contig_result* out, char* err, size_t errlen);
/* The same for all chains with one synchronisation: outs[n_chains], chain order. */
int  pg_job_fetch_all(pg_job* job, pg_contig_result* outs, char* err, size_t errlen);
/* Device-resident result buffers of contig c: lik f64 [n_lik], lik_exp i32 [n_lik]. */
int  pg_job_device_results(pg_job* job, uint32_t contig, void** d_lik, uint64_t* n_lik,
                           void** d_lik_exp, uint64_t* n_variants);
/* Per-kernel-class elapsed milliseconds of the LAST pg_job_run, measured with
 * hipEvents on the launch stream.  Order given by pg_job_kernel_name(). */
#define PG_N_KERNEL_CLASSES 6
int  pg_job_kernel_ms(const pg_job* job, double ms[PG_N_KERNEL_CLASSES]);
const char* pg_job_kernel_name(int cls);
/* Profiling hook: 64 in-kernel cycle counters of contig c's chain kernels, filled when the
 * library was built with -DPG_CHAIN_PROF (a measurement build; layout documented in DESIGN.md); zeros otherwise. */
int  pg_job_profile_counters(pg_job* job, uint32_t contig, uint64_t out64[64]);
/* Bytes of device memory held by the job. */
uint64_t pg_job_device_bytes(const pg_job* job);
/* How pg_job_run schedules the second half of every half-chain: 0 = fused (posterior partials formed
 * inside the sweep, one launch; chosen when many chains fill the chip), 1 = chunked (store-only sweep
 * chunks of *chunk_cols columns, posteriors of each finished chunk on the idle CUs; chosen for few
 * chains).  Override with the environment variables PG_SWEEP_MODE=fused|chunked, PG_CHUNK_COLS=n.
 * A column with more than five distinct alleles on the selected paths ("wide") is genotyped inside a fused job when its
 * chain has 16 paths (the sampled panels of production: the column costs that column); a chain of any other width with an
 * object of more than five alleles makes its job chunked, whatever is asked for.  (Round 5 refused a 16-path chain whose ONLY
 * column is wide in a fused job; since round 6 such a chain needs no sweep and is genotyped — only with PG_KERNELS=nosplit or
 * run_phasing, where the per-sample preparation of round 5 runs, pg_job_run still answers PG_ERR_UNSUPPORTED for it.) */
int  pg_job_sweep_mode(const pg_job* job, uint32_t* chunk_cols);
/* Number of chains of the job whose columns are kept as upper triangles (fused mode, H = 64, every object with at most five
 * alleles: the columns are symmetric, so phase 1 writes and phase 2 reads only the stored half — half of the
 * 16 H^2 bytes per variant of the full formulation; PG_KERNELS=notri turns it off).  For traffic accounting. */
uint32_t pg_job_triangle_chains(const pg_job* job);
/* The index pass (round 6): what the INDEX alone decides — the ColumnIndexer flags and the column list of every contig
 * (reference src/columnindexer.cpp:8-33); for the 16-path chains of fused jobs also every column's path -> local allele map,
 * transition constants (src/transitionprobabilitycomputer.cpp:8-19) and bin addresses — is formed ONCE per uploaded index,
 * inside pg_job_new / pg_cohort_new / pg_job_upload(batches != NULL), and shared by every chain (sample) over that index contig
 * (src/commands.cpp:118-138: one index, per-sample counts).  pg_job_run forms only what hangs on the sample's counts.
 * Elapsed milliseconds of the LAST index pass (hipEvents); it is NOT part of pg_job_kernel_ms. */
double pg_job_index_ms(const pg_job* job);
/* Which kernels run for which chains of this job, as text (one line per group of chains with the same plan: count, paths,
 * sweep mode, the kernels of the preparation / phase 1 / phase 2 / bins).  Writes at most `len` bytes incl. the terminator,
 * returns the length the full text needs.  For logs and DESIGN.md's table; the library takes every decision itself. */
size_t pg_job_plan(const pg_job* job, char* out, size_t len);
/* Elapsed milliseconds of the Viterbi kernels (run_phasing) of the LAST pg_job_run, hipEvents on the launch stream. */
double pg_job_viterbi_ms(const pg_job* job);
void pg_job_destroy(pg_job* job);

/* The inputs a job holds on the device: sizes of chain `contig`'s panel, and the arrays themselves back on the host
 * (caller-allocated by those sizes; any pointer may be NULL).  For jobs whose panel was formed on the device
 * (include/pangenie_sampler.h: pg_sampler_then_job) this is the only way to see it. */
int  pg_job_panel_sizes(const pg_job* job, uint32_t contig, uint32_t* n_variants, uint32_t* n_paths,
                        uint64_t* sum_kmers, uint64_t* sum_alleles);
int  pg_job_fetch_panel(pg_job* job, uint32_t contig, uint32_t* kmer_off, uint16_t* kmer_count,
                        uint32_t* allele_off, uint16_t* allele_id, uint8_t* allele_flags,
                        uint16_t* allele_kmer_off, uint32_t* allele_kmer_mask, uint16_t* path_allele,
                        char* err, size_t errlen);

/* pg_job_create with an error code instead of a NULL: PG_ERR_INVALID (malformed batch),
 * PG_ERR_UNSUPPORTED (limits above), PG_ERR_NOMEM (device allocation), PG_ERR_DEVICE. */
int  pg_job_new(int device, uint32_t n_contigs, const pg_contig_batch* batches,
                const pg_table* table, const pg_hmm_params* params,
                pg_job** out, char* err, size_t errlen);
/* Number of chains of a job (= n_contigs, or n_samples * n_contigs for a cohort job): the range of
 * the `contig` argument of pg_job_fetch / pg_job_device_results / pg_job_profile_counters. */
uint32_t pg_job_n_chains(const pg_job* job);

/* ------------------------------------------------------------------ *
 *  Cohort jobs: many samples against ONE index.
 *  The index (`*_UniqueKmersMap.cereal`: positions, alleles, k-mer masks, path -> allele) is
 *  shared by all samples; only the read k-mer counts and the local coverage differ per
 *  sample (reference src/commands.cpp:118-138, README.md:128).  The index arrays are
 *  uploaded once; every (sample, contig) pair is an independent chain of the job:
 *      chain id = sample * n_contigs + contig.
 *  `index[c].kmer_count` / `.coverage` are ignored (may be NULL).
 * ------------------------------------------------------------------ */
typedef struct pg_sample_counts {
    const uint16_t* const* kmer_count; /* [n_contigs] -> [sumK of contig c]  UniqueKmers::update_readcount */
    const uint16_t* const* coverage;   /* [n_contigs] -> [V of contig c]     UniqueKmers::set_coverage     */
} pg_sample_counts;
int  pg_cohort_new(int device, uint32_t n_contigs, const pg_contig_batch* index,
                   uint32_t n_samples, const pg_sample_counts* samples,
                   const pg_table* table, const pg_hmm_params* params,
                   pg_job** out, char* err, size_t errlen);

/* Re-uploads the inputs of a resident job (same shapes as at creation): what a pipeline does
 * with the next set of read counts.  `samples` = NULL for a job made by pg_job_create/new (the
 * batches carry their own counts), else the cohort's per-sample arrays (`batches` may then be
 * NULL to keep the resident index and upload the counts only). */
int  pg_job_upload(pg_job* job, const pg_contig_batch* batches, const pg_sample_counts* samples,
                   char* err, size_t errlen);
/* pg_job_upload followed by pg_job_run (on the job's own stream) in ONE call — the arrays stay valid until it returns, so a job of a
 * few long chains (a whole genome's chromosomes) uploads the inputs of its longest chains first, starts their preparation and phase 1
 * and lets the other chains' inputs cross PCIe meanwhile (round 6; the one-shot pg_hmm_genotype_contig does the same inside).  Same
 * results as the two calls, bit for bit.  pg_job_host_seconds: [1] then counts only what the caller waited for in front of the run. */
int  pg_job_upload_run(pg_job* job, const pg_contig_batch* batches, const pg_sample_counts* samples, char* err, size_t errlen);
/* The same for a cohort job WITHOUT stalling the device: pg_job_upload_begin starts copying the next batch of samples
 * (reference: one PanGenie run per sample re-reads its counts, src/commands.cpp:118-138) into the job's second set of
 * per-sample arrays — host threads pack the counts into a pinned staging buffer, a few large H2D copies on a copy stream of
 * the job's own — and returns at once; pg_job_run keeps genotyping the current batch meanwhile.  pg_job_upload_end waits
 * for the copy and makes the new batch the one the next pg_job_run reads.  Call order of a pipeline:
 *     upload_begin(n+1); run(n); fetch(n); upload_end(n+1); upload_begin(n+2); run(n+1); ...
 * The arrays behind `samples` must stay valid and unchanged until pg_job_upload_end returns; results of the previous run
 * must be fetched BEFORE pg_job_upload_end (it invalidates them: fetch then returns PG_ERR_INVALID until the next run). */
int  pg_job_upload_begin(pg_job* job, const pg_sample_counts* samples, char* err, size_t errlen);
int  pg_job_upload_end(pg_job* job, char* err, size_t errlen);
/* Host wall seconds: [0] device allocation at creation, [1] last input upload (H2D; after pg_job_upload_end: what it waited),
 * [2] last pg_job_run, [3] last pg_job_fetch_all / sum of pg_job_fetch since the last run. */
int  pg_job_host_seconds(const pg_job* job, double out4[4]);
/* Input bytes moved H2D by the last upload: [0] index arrays, [1] per-sample arrays. */
int  pg_job_upload_bytes(const pg_job* job, uint64_t out2[2]);
/* All chains' posteriors as two packed device ranges, chain after chain (what a multi-GPU
 * gather sends): lik f64 [n_lik_total], lik_exp i32 [n_lik_total]. */
int  pg_job_packed_results(pg_job* job, void** d_lik, void** d_lik_exp, uint64_t* n_lik_total);
/* The one-shot call keeps the device arenas of its finished jobs in a pool for the next calls
 * (per device at most PG_ARENA_POOL_GB, default 70 % of the device's memory); this releases them. */
void pg_hmm_release_cache(void);

/* ------------------------------------------------------------------ *
 *  Multi-GPU: contigs (x subsets x samples) shard across the GPUs of a node with no data-path
 *  collective — the reference already runs them as independent jobs and only merges results
 *  (src/commands.cpp:955-978, 163-177).  The ONE exchange is the collection of every rank's
 *  packed posteriors on the root: grouped RCCL point-to-point sends over xGMI, each block exactly
 *  as long as that rank's data (lik f64 + lik_exp i32 per genotype bin).  RCCL is loaded at run
 *  time, only when a communicator is made.
 * ------------------------------------------------------------------ */
typedef struct pg_comm pg_comm;
/* process-per-GPU: rank 0 makes the id, the host hands it to the other ranks (file, MPI, ...) */
int  pg_comm_unique_id(uint8_t id128[128], char* err, size_t errlen);
int  pg_comm_init(const uint8_t id128[128], int world, int rank, int device,
                  pg_comm** out, char* err, size_t errlen);
/* one process, several GPUs: out_comms[i] = rank i on devices[i] */
int  pg_comm_init_all(int n_devices, const int* devices, pg_comm** out_comms, char* err, size_t errlen);
int  pg_comm_rank(const pg_comm* comm);
int  pg_comm_world(const pg_comm* comm);
void pg_comm_destroy(pg_comm* comm);
/* Every rank calls this once per run with its job (NULL on a rank without chains) and the plan
 * n_lik_per_rank[world] (genotype bins each rank holds; known to every host up front).  On `root`,
 * d_lik_all (f64) / d_exp_all (i32) are device buffers of sum(n_lik_per_rank) elements: rank r's
 * block lands at offset sum_{q<r} n_lik_per_rank[q], in that rank's chain order.  Blocking. */
int  pg_hmm_gather(pg_comm* comm, pg_job* job, int root, const uint64_t* n_lik_per_rank,
                   void* d_lik_all, void* d_exp_all, char* err, size_t errlen);
/* the same for the n_local communicators one process holds (pg_comm_init_all) */
int  pg_hmm_gather_all(int n_local, pg_comm* const* comms, pg_job* const* jobs, int root,
                       const uint64_t* n_lik_per_rank, void* d_lik_all, void* d_exp_all,
                       char* err, size_t errlen);
/* the same into HOST buffers of the root's process (temporary device buffers inside) */
int  pg_hmm_gather_to_host(int n_local, pg_comm* const* comms, pg_job* const* jobs, int root,
                           const uint64_t* n_lik_per_rank, double* h_lik_all, int32_t* h_exp_all,
                           char* err, size_t errlen);

/* ------------------------------------------------------------------ *
 *  Unit-level entry points (device), mirroring the reference classes the
 *  reference's own unit tests exercise.
 * ------------------------------------------------------------------ */
/* EmissionProbabilityComputer (emissionprobabilitycomputer.cpp:9-34): A x A table over ALL
 * allele slots of variant `v` (row-major, slot order), after the all_zeros rule. */
int pg_emission_table(const pg_contig_batch* batch, const pg_table* table, uint32_t v,
                      int device, long double* out_AxA, int32_t* all_zeros,
                      char* err, size_t errlen);
/* TransitionProbabilityComputer (transitionprobabilitycomputer.cpp:8-19):
 * out3 = {no switch, one switch, two switches}. */
int pg_transition_probs(uint64_t from_pos, uint64_t to_pos, double recombrate,
                        uint32_t nr_paths, int uniform, long double effective_N,
                        int device, double out3[3], char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif /* PANGENIE_HMM_H */
