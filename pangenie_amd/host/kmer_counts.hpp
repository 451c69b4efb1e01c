// kmer_counts.hpp — the step that turns an index into a sample's input: read k-mer counts and local coverage of every
// variant (reference fill_read_kmercounts, src/commands.cpp:74-139; src/kmerparser.cpp; src/kmercounter.hpp) —
// SURVEY.md §8(f)-3.  Host code; not on the GPU path (hash lookups and text parsing).
//
// The reference's count source is Jellyfish (third party, pinned jellyfish=2.2.10 in environment.yml; not in this
// image): JellyfishCounter runs `mer_counter` over the read file with canonical = true (src/jellyfishcounter.cpp:26-49),
// i.e. every window of k letters over {A,C,G,T} of every read is counted under the lexicographically smaller of the
// k-mer and its reverse complement; windows containing any other letter are skipped.  ExactKmerCounter restates that
// with an exact hash map of EVERY k-mer of the reads — for small inputs (tests, regions); TargetedKmerCounter counts only
// the k-mers the index asks about (memory proportional to the index, not to the reads; plain or gzipped reads, worker
// threads) — the same numbers for those k-mers, for read sets of any size; k <= 32.
// The k-mer abundance peak (`kmer_coverage`) is an ARGUMENT here: finding it (the reference's Histogram /
// compute_kmer_coverage over Jellyfish's histogram) is upstream of this path and out of scope.
// Pinned on the reference's own fixtures: tests/data/index_UniqueKmersMap.cereal + index_chr1_kmers.tsv.gz +
// region-reads.fa must give tests/data/region_UniqueKmersList.cereal (what tests/CommandsTest.cpp:59-93 feeds its HMM).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
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
    size_t distinct_kmers() const { return filled_; }

private:
    void add_sequence(const std::string& seq);
    bool encode_canonical(const char* s, uint64_t& code) const;
    void bump(uint64_t code);
    void grow();
    // canonical code -> count: open addressing with linear probing, 12 bytes a slot, doubled at 60 % load (a graph's or a
    // region's worth of k-mers: tens of millions of entries, where a node-based map spends its time in the allocator)
    static constexpr uint64_t kFree = ~0ull;   // (never a canonical code: the reverse complement of all-T is all-A = 0)
    size_t k_;
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> seen_;     // saturates at 2^32 - 1
    size_t filled_ = 0;
};

/** Counts of a GIVEN set of k-mers in read files of any size.  fill_read_kmercounts only ever asks for the unique and
 *  the flanking k-mers listed in the `_kmers.tsv.gz` tables of the index, so those are registered first
 *  (add_targets_from_table for every chromosome), then the reads are streamed once (count(): FASTA / FASTQ, plain or
 *  gzipped; one reader, `threads` counting workers) and every window that is a registered k-mer is counted under its
 *  canonical code in an open-addressing table.  getKmerAbundance of a k-mer that was never registered throws (a silent 0
 *  would be a wrong count).  Same counts as ExactKmerCounter / the reference's Jellyfish pass for registered k-mers. */
class TargetedKmerCounter : public KmerCounter {
public:
    /** `unregistered_counts_zero`: what getKmerAbundance does for a k-mer that was never registered — false (default): throw;
     *  true: return 0, which is what the reference's graph-only JellyfishCounter answers (tests/KmerCounterTest.cpp:21-32) */
    explicit TargetedKmerCounter(size_t kmer_size, bool unregistered_counts_zero = false);
    /** register a k-mer (any orientation); k-mers with letters outside ACGT can never be counted and are ignored */
    void add_target(std::string_view kmer);
    /** register every unique and flanking k-mer of a `<prefix>_<chromosome>_kmers.tsv(.gz)` table; returns how many rows */
    size_t add_targets_from_table(const std::string& kmers_tsv_gz);
    /** register every window of a FASTA / FASTQ file — `<prefix>_path_segments.fasta` makes this counter the reference's
     *  default count source (JellyfishCounter(readfile, {segment_file}, ...), src/jellyfishcounter.cpp:51-84: only k-mers of
     *  the graph are counted); returns how many windows were registered (with repeats) */
    size_t add_targets_from_sequences(const std::string& fasta);
    /** register every window of a sequence held in memory */
    void add_targets_of(std::string_view sequence);
    /** stream a read file and count; may be called for several files (counts add up).  No targets may be added afterwards. */
    void count(const std::string& readfile, unsigned threads = 1);
    size_t getKmerAbundance(std::string kmer) override;
    size_t targets() const { return n_targets_; }
    /** how many registered k-mers were seen c times, c = 1..max_count (index 0 stays 0, larger counts are left out) — the
     *  numbers the reference's JellyfishCounter::computeHistogram collects (src/jellyfishcounter.cpp:119-126) before it
     *  looks for the abundance peak; the peak rule itself is not part of this build (see above) */
    std::vector<size_t> abundance_histogram(size_t max_count);
    size_t kmers_seen() const { return windows_; }   // windows over ACGT of all reads streamed so far

private:
    bool encode_canonical(const char* s, uint64_t& code) const;
    void freeze(unsigned threads = 1);
    size_t find(uint64_t code) const;   // slot, or npos
    void count_sequence(const char* s, size_t n, uint64_t& windows);
    static constexpr uint64_t kEmpty = ~0ull;   // (never a canonical code: the reverse complement of all-T is all-A = 0)
    size_t k_;
    std::vector<uint64_t> pending_;     // codes registered before the table is built
    struct Slot { uint64_t key, count; };
    // (resize() of this vector leaves new slots untouched — freeze() fills them from its workers, so that the pages of a
    // table of gigabytes are first touched in parallel)
    template <class T> struct RawAllocator : std::allocator<T> {
        template <class U> struct rebind { using other = RawAllocator<U>; };
        template <class U, class... A> void construct(U* p, A&&... a) {
            if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
        }
    };
    std::vector<Slot, RawAllocator<Slot>> slots_;   // open addressing, power-of-two size, linear probing
    size_t n_targets_ = 0;
    uint64_t windows_ = 0;
    bool frozen_ = false;
    bool lenient_ = false;
};

/** one row of `<prefix>_<chromosome>_kmers.tsv(.gz)` — the reference's interface (src/kmerparser.hpp); implemented
 *  over an in-place column scanner (kmer_counts.cpp: KmerRow) */
void parse_kmer_line(std::string line, std::string& chrom, size_t& start, std::vector<std::string>& kmers,
                     std::vector<std::string>& flanking_kmers, bool& is_header);

/** local coverage of a variant = integer mean of the flanking k-mers' read counts inside [coverage / 4, coverage * 4],
 *  the given coverage when none qualifies (behaviour: src/kmerparser.cpp:32-53) */
unsigned short compute_local_coverage(std::vector<std::string>& kmers, KmerCounter& read_counts, size_t kmer_coverage);

/** The count-filling part of the reference's fill_read_kmercounts (src/commands.cpp:74-139): for every variant of
 *  `chromosome`, update_readcount(i, count of the i-th unique k-mer) and set_coverage(local coverage), reading the
 *  k-mers from the gzipped table PanGenie-index wrote.  (The reference's function goes on to run the HaplotypeSampler:
 *  that is pangenie::HaplotypeSampler, on the GPU.)  Throws std::runtime_error where the reference asserts. */
void fill_read_kmercounts(const std::string& chromosome, UniqueKmersMap* unique_kmers_map, KmerCounter& read_kmer_counts,
                          const std::string& kmers_tsv_gz, size_t kmer_coverage);

/** fill_read_kmercounts for every chromosome of the index, `threads` chromosomes at a time (the reference's thread pool
 *  around it, src/commands.cpp:857-868), tables at `<prefix>_<chromosome>_kmers.tsv.gz`.  The counter is only read; the
 *  first exception of a worker is rethrown after all have stopped. */
void fill_read_kmercounts_all(UniqueKmersMap* unique_kmers_map, KmerCounter& read_kmer_counts, const std::string& prefix,
                              size_t kmer_coverage, unsigned threads = 1);

}  // namespace pangenie
