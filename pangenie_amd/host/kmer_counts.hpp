// kmer_counts.hpp — the step that turns an index into a sample's input: read k-mer counts and local coverage of every
// variant (reference fill_read_kmercounts, src/commands.cpp:74-139; src/kmerparser.cpp; src/kmercounter.hpp) —
// SURVEY.md §8(f)-3.  Host code; not on the GPU path (hash lookups and text parsing).
//
// The reference's count source is Jellyfish (third party, pinned jellyfish=2.2.10 in environment.yml; not in this
// image): JellyfishCounter runs `mer_counter` over the read file with canonical = true (src/jellyfishcounter.cpp:26-49),
// i.e. every window of k letters over {A,C,G,T} of every read is counted under the lexicographically smaller of the
// k-mer and its reverse complement; windows containing any other letter are skipped.  ExactKmerCounter restates that
// with an exact hash map — for small inputs (tests, regions); k <= 32.
// Pinned on the reference's own fixtures: tests/data/index_UniqueKmersMap.cereal + index_chr1_kmers.tsv.gz +
// region-reads.fa must give tests/data/region_UniqueKmersList.cereal (what tests/CommandsTest.cpp:59-93 feeds its HMM).
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "cereal_io.hpp"

namespace pangenie {

/** reference src/kmercounter.hpp:8-24 */
class KmerCounter {
public:
    virtual ~KmerCounter() {}
    /** abundance of the given k-mer (canonical representation is used) */
    virtual size_t getKmerAbundance(std::string kmer) = 0;
};

/** exact canonical k-mer counts of a FASTA / FASTQ file (see above) */
class ExactKmerCounter : public KmerCounter {
public:
    ExactKmerCounter(const std::string& readfile, size_t kmer_size);
    size_t getKmerAbundance(std::string kmer) override;
    size_t distinct_kmers() const { return counts_.size(); }
    /** reference JellyfishCounter::computeHistogram (src/jellyfishcounter.cpp:119-153) over the counted k-mers:
     *  histogram of the abundances, smoothed, its largest / second largest peak */
    size_t computeHistogram(size_t max_count, bool largest_peak) const;

private:
    void add_sequence(const std::string& seq);
    bool encode_canonical(const char* s, uint64_t& code) const;
    size_t k_;
    std::unordered_map<uint64_t, uint64_t> counts_;
};

/** reference src/histogram.hpp:7-20, src/histogram.cpp: k-mer abundance histogram (count -> number of k-mers) */
class Histogram {
public:
    explicit Histogram(size_t max_value);
    /** "count <tab> value" lines (plain or gzipped); counts above max_value are dropped (src/histogram.cpp:12-24) */
    Histogram(const std::string& filename, size_t max_value);
    void add_value(size_t value);
    void smooth_histogram();
    void find_peaks(std::vector<size_t>& peak_ids, std::vector<size_t>& peak_values) const;
    const std::vector<size_t>& values() const { return histogram_; }

private:
    std::vector<size_t> histogram_;
};
/** reference src/sequenceutils.cpp:42-83: the largest (or second largest) peak of the smoothed histogram = the k-mer
 *  abundance peak the ProbabilityTable is built from (src/commands.cpp:840-846) */
size_t compute_kmer_coverage(std::vector<size_t>& peak_ids, std::vector<size_t>& peak_values, bool largest_peak);

/** reference src/kmerparser.cpp:16-30: one line of `<prefix>_<chromosome>_kmers.tsv(.gz)` */
void parse_kmer_line(std::string line, std::string& chrom, size_t& start, std::vector<std::string>& kmers,
                     std::vector<std::string>& flanking_kmers, bool& is_header);

/** reference src/kmerparser.cpp:32-53: mean count of the flanking k-mers within [coverage / 4, coverage * 4] */
unsigned short compute_local_coverage(std::vector<std::string>& kmers, KmerCounter& read_counts, size_t kmer_coverage);

/** The count-filling part of the reference's fill_read_kmercounts (src/commands.cpp:74-139): for every variant of
 *  `chromosome`, update_readcount(i, count of the i-th unique k-mer) and set_coverage(local coverage), reading the
 *  k-mers from the gzipped table PanGenie-index wrote.  (The reference's function goes on to run the HaplotypeSampler:
 *  that is pangenie::HaplotypeSampler, on the GPU.)  Throws std::runtime_error where the reference asserts. */
void fill_read_kmercounts(const std::string& chromosome, UniqueKmersMap* unique_kmers_map, KmerCounter& read_kmer_counts,
                          const std::string& kmers_tsv_gz, size_t kmer_coverage);

}  // namespace pangenie
