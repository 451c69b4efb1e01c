// pangenie_host.hpp — C++ host side of the boundary: the reference's interface for the
// genotyping hot path (same class names, method names, argument meaning and error behaviour),
// implemented over the C ABI of include/pangenie_hmm.h.
//
//   UniqueKmers / BiallelicUniqueKmers / MultiallelicUniqueKmers   reference src/uniquekmers.hpp:22-69,
//                                                                  src/biallelicuniquekmers.hpp, src/multiallelicuniquekmers.hpp
//   KmerPath (window 16 or 32)                                     reference src/kmerpath.hpp, src/kmerpath16.hpp
//   CopyNumber, ProbabilityTable                                   reference src/copynumber.hpp, src/probabilitytable.hpp
//   ColumnIndexer                                                  reference src/columnindexer.hpp
//   EmissionProbabilityComputer, TransitionProbabilityComputer     reference src/emissionprobabilitycomputer.hpp,
//                                                                  src/transitionprobabilitycomputer.hpp   (device)
//   GenotypingResult                                               reference src/genotypingresult.hpp
//   HMM                                                            reference src/hmm.hpp:26-47                (device)
//
// Everything that computes probabilities along the chain runs on the GPU through the C ABI;
// there is no CPU implementation of the HMM in this library.  Containers, bookkeeping and the
// long double post-processing (normalise, GT, GQ) stay on the host exactly where the reference
// has them.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pangenie_hmm.h"
#include "../../include/pangenie_sampler.h"

namespace pangenie {

/** Sequence of k-mer presence bits: (offset, mask) over a window of W k-mer indices. */
class KmerPath {
public:
    explicit KmerPath(unsigned window = 32) : offset_(0), kmers_(0), window_(window) {}
    KmerPath(unsigned window, unsigned short offset, uint32_t mask) : offset_(offset), kmers_(mask), window_(window) {}  // from an archive
    void set_position(unsigned short index);
    unsigned int get_position(unsigned short index) const;
    size_t nr_kmers() const;
    std::string convert_to_string() const;
    unsigned short offset() const { return offset_; }
    uint32_t mask() const { return kmers_; }

private:
    unsigned short offset_;
    uint32_t kmers_;
    unsigned window_;
};
std::ostream& operator<<(std::ostream& os, const KmerPath& p);

/** Probabilities of a k-mer having copy number 0, 1, 2. */
class CopyNumber {
public:
    CopyNumber();
    CopyNumber(long double cn_0, long double cn_1, long double cn_2);
    CopyNumber(long double cn_0, long double cn_1, long double cn_2, long double regularization_const);
    long double get_probability_of(int cn) const;
    bool operator==(const CopyNumber& other) const;
    bool operator!=(const CopyNumber& other) const;

private:
    long double p_[3];
};

/** Pre-computed copy-number probabilities per (k-mer coverage, read k-mer count).
 *  Owns a pg_table: the same object feeds the device. */
class ProbabilityTable {
public:
    ProbabilityTable();
    ProbabilityTable(unsigned short cov_min, unsigned short cov_max, unsigned short count_max,
                     long double regularization_const);
    ProbabilityTable(const ProbabilityTable&) = delete;
    ProbabilityTable& operator=(const ProbabilityTable&) = delete;
    ProbabilityTable(ProbabilityTable&& o) noexcept : t_(o.t_) { o.t_ = nullptr; }
    ProbabilityTable& operator=(ProbabilityTable&& o) noexcept;
    ~ProbabilityTable();
    CopyNumber get_probability(unsigned short kmer_coverage, unsigned short read_kmer_count) const;
    /** test hook, throws std::runtime_error outside the precomputed box */
    void modify_probability(unsigned short kmer_coverage, unsigned short read_kmer_count, CopyNumber prob);
    const pg_table* handle() const { return t_; }

private:
    pg_table* t_;
};

/** The set of unique k-mers of one variant position (abstract, as in the reference). */
class UniqueKmers {
public:
    virtual ~UniqueKmers() = default;
    virtual size_t get_variant_position() const = 0;
    virtual void insert_kmer(unsigned short readcount, std::vector<unsigned short>& allele_ids) = 0;
    virtual bool kmer_on_path(size_t kmer_index, size_t path_id) const = 0;
    virtual bool kmer_on_allele(size_t kmer_index, size_t allele_id) const = 0;
    virtual unsigned short get_readcount_of(size_t kmer_index) = 0;
    virtual void update_readcount(size_t kmer_index, unsigned short new_count) = 0;
    virtual size_t size() const = 0;
    virtual unsigned short get_nr_paths() const = 0;
    virtual void get_path_ids(std::vector<unsigned short>& paths, std::vector<unsigned short>& alleles,
                              std::vector<unsigned short>* only_include = nullptr) = 0;
    virtual void get_allele_ids(std::vector<unsigned short>& a) = 0;
    virtual void get_defined_allele_ids(std::vector<unsigned short>& a) = 0;
    virtual void set_coverage(unsigned short local_coverage) = 0;
    virtual unsigned short get_coverage() const = 0;
    virtual std::map<unsigned short, int> kmers_on_alleles() const = 0;
    virtual unsigned short kmers_on_allele(unsigned short allele_id) const = 0;
    virtual unsigned short present_kmers_on_allele(unsigned short allele_id) const = 0;
    virtual float fraction_present_kmers_on_allele(unsigned short allele_id) const = 0;
    virtual bool is_undefined_allele(unsigned short allele_id) const = 0;
    virtual void set_undefined_allele(unsigned short allele_id) = 0;
    virtual unsigned short get_allele(unsigned short path_id) const = 0;
    virtual void update_paths(std::vector<unsigned short>& path_ids) = 0;
    /** (offset, mask) of the allele's k-mer bits — what the flat batch carries (include/pangenie_hmm.h). */
    virtual std::pair<unsigned short, uint32_t> kmer_bits(unsigned short allele_id) const = 0;
};

/** Shared implementation; BIALLELIC restricts alleles to {0,1} and uses the 16-k-mer window. */
template <bool BIALLELIC>
class UniqueKmersT : public UniqueKmers {
public:
    UniqueKmersT() : variant_pos_(0), local_coverage_(0) {}
    UniqueKmersT(size_t variant_position, std::vector<unsigned short>& alleles);
    size_t get_variant_position() const override { return variant_pos_; }
    void insert_kmer(unsigned short readcount, std::vector<unsigned short>& allele_ids) override;
    bool kmer_on_path(size_t kmer_index, size_t path_id) const override;
    bool kmer_on_allele(size_t kmer_index, size_t allele_id) const override;
    unsigned short get_readcount_of(size_t kmer_index) override;
    void update_readcount(size_t kmer_index, unsigned short new_count) override;
    size_t size() const override { return counts_.size(); }
    unsigned short get_nr_paths() const override { return (unsigned short)path_to_allele_.size(); }
    void get_path_ids(std::vector<unsigned short>& paths, std::vector<unsigned short>& alleles,
                      std::vector<unsigned short>* only_include = nullptr) override;
    void get_allele_ids(std::vector<unsigned short>& a) override;
    void get_defined_allele_ids(std::vector<unsigned short>& a) override;
    void set_coverage(unsigned short local_coverage) override { local_coverage_ = local_coverage; }
    unsigned short get_coverage() const override { return (unsigned short)local_coverage_; }
    std::map<unsigned short, int> kmers_on_alleles() const override;
    unsigned short kmers_on_allele(unsigned short allele_id) const override;
    unsigned short present_kmers_on_allele(unsigned short allele_id) const override;
    float fraction_present_kmers_on_allele(unsigned short allele_id) const override;
    bool is_undefined_allele(unsigned short allele_id) const override;
    void set_undefined_allele(unsigned short allele_id) override;
    unsigned short get_allele(unsigned short path_id) const override;
    void update_paths(std::vector<unsigned short>& path_ids) override;
    std::pair<unsigned short, uint32_t> kmer_bits(unsigned short allele_id) const override;

    /** The stored fields as plain data (archive I/O, cereal_io.hpp). */
    struct RawAllele { unsigned short offset = 0; uint32_t mask = 0; bool is_undefined = false; };
    struct Raw {
        size_t variant_pos = 0;
        float local_coverage = 0.f;
        std::vector<unsigned short> counts;
        std::map<unsigned short, RawAllele> alleles;
        std::vector<unsigned short> path_to_allele;
    };
    explicit UniqueKmersT(const Raw& raw);
    Raw raw() const;

private:
    struct AlleleInfo {
        KmerPath kmer_path{BIALLELIC ? 16u : 32u};
        bool is_undefined = false;
    };
    void check_allele(unsigned short a, const char* where) const;
    size_t variant_pos_;
    float local_coverage_;
    std::vector<unsigned short> counts_;
    std::map<unsigned short, AlleleInfo> alleles_;
    std::vector<unsigned short> path_to_allele_;
};
using BiallelicUniqueKmers = UniqueKmersT<true>;
using MultiallelicUniqueKmers = UniqueKmersT<false>;

/** Which variants become HMM columns, which paths are selected (host restatement; the device
 *  applies the same rule in k_prep — this class serves callers and tests). */
class ColumnIndexer {
public:
    ColumnIndexer(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, std::vector<unsigned short>* only_paths);
    size_t get_variant_id(size_t column_index) const;
    size_t size() const { return columns_.size(); }
    unsigned short nr_paths() const { return (unsigned short)paths_.size(); }
    unsigned short get_path(unsigned short path_index) const;
    unsigned short get_allele(unsigned short path_index, size_t column_index) const;
    std::pair<unsigned short, unsigned short> get_path_ids_at(size_t position) const;
    const std::vector<unsigned short>& paths() const { return paths_; }

private:
    std::vector<size_t> columns_;
    std::vector<unsigned short> paths_;
    std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers_;
};

/** Genotyping / phasing result of one position. */
class GenotypingResult {
public:
    GenotypingResult();
    void add_to_likelihood(unsigned short allele1, unsigned short allele2, long double value);
    void add_first_haplotype_allele(unsigned short allele) { haplotype_1_ = allele; }
    void add_second_haplotype_allele(unsigned short allele) { haplotype_2_ = allele; }
    long double get_genotype_likelihood(unsigned short allele1, unsigned short allele2) const;
    std::vector<long double> get_all_likelihoods(size_t nr_alleles) const;
    GenotypingResult get_specific_likelihoods(std::vector<unsigned short>& alleles) const;
    size_t get_genotype_quality(unsigned short allele1, unsigned short allele2) const;
    std::pair<unsigned short, unsigned short> get_haplotype() const { return {haplotype_1_, haplotype_2_}; }
    void divide_likelihoods_by(long double value);
    std::pair<int, int> get_likeliest_genotype() const;
    void combine(GenotypingResult& likelihoods);
    void normalize();
    void set_unique_kmers(unsigned short nr_unique_kmers) { unique_kmers_ = nr_unique_kmers; }
    void set_coverage(unsigned short coverage) { local_coverage_ = coverage; }
    unsigned short nr_unique_kmers() const { return unique_kmers_; }
    unsigned short coverage() const { return local_coverage_; }
    bool contains_no_likelihoods() const { return genotype_to_likelihood_.empty(); }
    const std::map<std::pair<unsigned short, unsigned short>, long double>& get_stored_likelihoods() const {
        return genotype_to_likelihood_;
    }
    friend std::ostream& operator<<(std::ostream& os, const GenotypingResult& res);

private:
    std::map<std::pair<unsigned short, unsigned short>, long double> genotype_to_likelihood_;
    unsigned short haplotype_1_, haplotype_2_, local_coverage_, unique_kmers_;
};

/** The sample column of a genotyped VCF record, `GT:GQ:GL:KC`, as Graph::write_genotypes prints it (reference
 *  src/graph.cpp:217-273): an empty result becomes L(0/0) = 1 (:225-227), genotypes with undefined alleles are
 *  dropped and the rest renormalised (:229-233), GT = likeliest genotype or `.` when there is no unique maximum
 *  (:246-258), GQ (genotypingresult.cpp:118-137), GL = log10 with 4 significant digits in VCF order (:268-273),
 *  KC = local k-mer coverage.  `result` must be normalised; `nr_alleles` = all alleles of the record. */
std::string genotype_field(const GenotypingResult& result, std::vector<unsigned short>& defined_alleles, size_t nr_alleles,
                           bool ignore_imputed = false);

/** Li-Stephens transition probabilities between two variants — computed on the device. */
class TransitionProbabilityComputer {
public:
    TransitionProbabilityComputer(size_t from_variant, size_t to_variant, double recomb_rate, unsigned short nr_paths,
                                  bool uniform = false, long double effective_N = 25000.0L);
    long double compute_transition_prob(unsigned short path_id1, unsigned short path_id2, unsigned short path_id3,
                                        unsigned short path_id4);
    long double compute_transition_prob(unsigned short nr_switches);

private:
    long double probabilities_[3];
    bool uniform_;
};

/** Emission probabilities of one variant position — computed on the device. */
class EmissionProbabilityComputer {
public:
    EmissionProbabilityComputer(std::shared_ptr<UniqueKmers> uniquekmers, ProbabilityTable* probabilities);
    long double get_emission_probability(unsigned short allele_id1, unsigned short allele_id2) const;

private:
    std::vector<unsigned short> allele_ids_;
    std::vector<long double> table_;  // A x A over allele slots, after the all_zeros rule
};

/** Flat view of a UniqueKmers vector (+ only_paths): owns the arrays a pg_contig_batch points to. */
struct FlatContig {
    std::vector<uint64_t> variant_pos;
    std::vector<uint16_t> coverage, kmer_count, allele_id, allele_kmer_off, path_allele;
    std::vector<uint32_t> kmer_off, allele_off, allele_kmer_mask;
    std::vector<uint8_t> allele_flags;
    std::vector<unsigned short> paths;  // selected path ids (ColumnIndexer::paths)
    pg_contig_batch batch{};
    void bind();
};
/** Throws std::runtime_error("HMM::index_columns: column N is not covered by any paths.") like ColumnIndexer. */
void flatten(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, std::vector<unsigned short>* only_paths, FlatContig& out);

/** The genotyping HMM.  All work happens in the constructor, on the GPU: forward-backward genotyping and, with
 *  run_phasing, the Viterbi path (get_haplotype(); pangenie_amd/csrc/pg_viterbi.hip, up to 64 selected paths — more are
 *  refused with std::runtime_error; the reference's callers pass at most 30, src/commands.cpp:939). */
class HMM {
public:
    HMM() = default;
    HMM(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, ProbabilityTable* probabilities, bool run_genotyping,
        bool run_phasing, double recombrate = 1.26, bool uniform = false, long double effective_N = 25000.0L,
        std::vector<unsigned short>* only_paths = nullptr, bool normalize = true);
    void combine_likelihoods(HMM& other);
    void normalize();
    std::vector<GenotypingResult> get_genotyping_result() const { return genotyping_result_; }
    std::vector<GenotypingResult> move_genotyping_result() { return std::move(genotyping_result_); }
    /** reference src/hmm.hpp:49-52 — `archive(genotyping_result)`: what a cereal binary archive holds of an HMM, the vector of its
     *  GenotypingResults (u64 count, then per element the layout host/cereal_io.hpp gives for the `-w` Results archive, whose
     *  per-chromosome vectors are these same bytes).  deserialize() is the loading direction of the same template; it needs no
     *  device.  Defined in host/cereal_io.cpp. */
    std::vector<unsigned char> serialize() const;
    static HMM deserialize(const std::vector<unsigned char>& bytes);
    /** device the calling thread's HMMs run on (default 0) */
    static void set_device(int device);
    static int device_count();

private:
    std::vector<GenotypingResult> genotyping_result_;
};

/** One (contig, path subset) to genotype: what run_genotyping receives (reference src/commands.cpp:155-160). */
struct ContigTask {
    std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers = nullptr;
    std::vector<unsigned short>* only_paths = nullptr;
};

/** The reference's job loop (src/commands.cpp:955-978: one HMM per contig x subset on a thread pool) across the
 *  GPUs of one node: tasks are assigned to `devices` by longest-processing-time-first (weight = variants x H^2),
 *  every device runs ONE resident job over its tasks (all its chains concurrently, driven by its own host
 *  thread), and the posteriors of all devices are collected with ONE RCCL exchange over xGMI
 *  (pg_hmm_gather_to_host); with a single device there is no exchange at all.  Returns, per task, what
 *  HMM(...).get_genotyping_result() returns for it (unnormalised: normalize = false, as run_genotyping calls it). */
std::vector<std::vector<GenotypingResult>> run_contigs_multi_gpu(std::vector<ContigTask>& tasks, ProbabilityTable* probabilities,
                                                                 double recombrate, bool uniform, long double effective_N,
                                                                 const std::vector<int>& devices);

/** One sample's numbers against a shared index: per chromosome the read count of every unique k-mer (variant after
 *  variant, k-mer after k-mer) and the local coverage of every variant — all that differs between the samples of a cohort
 *  (reference src/commands.cpp:118-138: update_readcount / set_coverage). */
struct SampleCounts {
    std::map<std::string, std::vector<uint16_t>> kmer_count, coverage;
    /** the numbers the objects of `chromosomes` hold right now (after fill_read_kmercounts for this sample) */
    static SampleCounts of(const std::map<std::string, std::vector<std::shared_ptr<UniqueKmers>>>& chromosomes);
};

/** Many samples against ONE index in one device job (C ABI pg_cohort_new): the index arrays of
 *  `chromosomes` go to the device once, every (sample, chromosome) pair is an independent chain over ALL paths of the index,
 *  and what comes back is, per sample and chromosome, what HMM(...).get_genotyping_result() returns for that sample alone
 *  (unnormalised, as run_genotyping asks for it, src/commands.cpp:160).  `probabilities` serves every sample: a table
 *  entry depends on (coverage, count) only, so one table whose box spans the samples' peaks (min peak / 4 .. max peak * 4,
 *  counts up to 2 * max peak) gives each sample the values of its own table.  Throws std::runtime_error on samples whose
 *  arrays do not fit the index. */
std::vector<std::map<std::string, std::vector<GenotypingResult>>> genotype_cohort(
    std::map<std::string, std::vector<std::shared_ptr<UniqueKmers>>>& chromosomes, const std::vector<SampleCounts>& samples,
    ProbabilityTable* probabilities, double recombrate = 1.26, bool uniform = false, long double effective_N = 25000.0L, int device = 0);

// ------------------------------------------------------------------ haplotype sampling (include/pangenie_sampler.h)
/** reference src/haplotypesampler.hpp:16-59 */
struct SampledPaths {
    std::vector<std::vector<size_t>> sampled_paths;
    std::vector<bool> mask_indexes(size_t column_index, size_t max_index);
    bool recombination(size_t column_index, size_t path_id);
};

/** reference src/samplingemissions.hpp / .cpp:9-45 (costs formed by pg_sampler_emission_costs) */
class SamplingEmissions {
public:
    SamplingEmissions(std::shared_ptr<UniqueKmers> uniquekmers);
    unsigned int get_emission_cost(unsigned short allele_id) const;
    void penalize(unsigned short allele_id, unsigned short penalty);

private:
    std::vector<unsigned short> allele_penalties;
    unsigned int default_penalty;
};

/** reference src/samplingtransitions.hpp / .cpp:5-23 */
class SamplingTransitions {
public:
    SamplingTransitions(size_t from_variant, size_t to_variant, double recomb_rate, unsigned short nr_paths, long double effective_N = 25000.0L);
    unsigned int compute_transition_cost(bool recombination);

private:
    unsigned int cost;
};

/** reference src/haplotypesampler.hpp:62-96.  The `size` Viterbi passes run on the GPU (pg_sampler_run); the
 *  constructor then reduces the UniqueKmers objects to the sampled paths exactly as the reference does
 *  (update_unique_kmers).  path_output / chromosome: the per-position table of sampled path ids the reference
 *  writes when asked to. */
class HaplotypeSampler {
public:
    HaplotypeSampler(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, size_t size, double recombrate = 1.26,
                     long double effective_N = 25000.0L, std::vector<unsigned int>* best_scores = nullptr, bool add_reference = false,
                     std::string path_output = "", std::string chromosome = "None", unsigned short allele_penalty = 10,
                     double* time = nullptr);
    void get_column_minima(std::vector<unsigned int>& column, std::vector<bool>& mask, size_t& first_id, size_t& second_id,
                           unsigned int& first_val, unsigned int& second_val) const;
    SampledPaths get_sampled_paths() const { return sampled_paths; }
    static void set_device(int device);

private:
    void update_unique_kmers();
    std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers;
    SampledPaths sampled_paths;
};

}  // namespace pangenie
