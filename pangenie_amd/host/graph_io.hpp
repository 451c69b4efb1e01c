// graph_io.hpp — the consumer side of the path: the `<prefix>_<chromosome>_Graph.cereal` archive PanGenie-index writes
// (reference Graph::save, src/graph.hpp:79-87) read WITHOUT cereal, the un-merging of combined variant bubbles
// (behaviour: Variant::separate_variants, src/variant.cpp:308-383) and the text of a genotyped VCF (behaviour:
// Graph::write_genotypes, src/graph.cpp:118-278) — SURVEY.md §8(f)-4.  Host code; the likelihoods it prints come from
// the device (pangenie::HMM).
//
// Archive layout (cereal binary: little endian, no framing; member order of the reference's serialize functions):
//   Graph    = FastaReader · chromosome string · kmer_size u64 · add_reference u8 · variants_deleted u8 ·
//              vector<shared_ptr<Variant>> · variant_ids vector<vector<string>>                  (src/graph.hpp:79-87)
//   FastaReader = map<string, shared_ptr<DnaSequence>>                                           (src/fastareader.hpp:40-43)
//   DnaSequence = vector<u8> (two bases per byte, first base in the high nibble: A C G T = 0 1 2 3, anything else 4) ·
//              even_length u8 · is_undefined u8                                                  (src/dnasequence.hpp:48-51)
//   Variant  = left_flank · right_flank · inner_flanks vector<DnaSequence> · chromosome · start_position u64 ·
//              allele_sequences vector<vector<DnaSequence>> (one list per merged VCF record) · allele_combinations
//              vector<vector<u16>> (bubble allele -> allele of each record) · uncovered_alleles vector<vector<u16>> ·
//              paths vector<u16> (bubble allele of every panel path) · flanks_added u8          (src/variant.hpp:93-96)
//   shared_ptr<T> (T not polymorphic) = id u32, MSB set the first time the object occurs (its data follows), 0 = null;
//              ids count up from 1 over ALL shared pointers of the archive
//   string = u64 length + bytes · vector<T> = u64 n + elements (arithmetic T: raw)
// Checked byte for byte on the reference's fixture tests/data/index_chr1_Graph.cereal (kept as data under tests/golden/).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "pangenie_host.hpp"

namespace pangenie {

/** A DNA string in the archive's packed form. */
class DnaSequence {
public:
    DnaSequence() = default;
    explicit DnaSequence(const std::string& bases);
    size_t size() const { return packed_.size() * 2 - (even_length_ ? 0 : 1); }
    char operator[](size_t position) const;
    std::string to_string() const;
    void append(const DnaSequence& other);
    /** the archive's flag (set when a base outside ACGT was appended; not recomputed from the content) */
    bool contains_undefined() const { return undefined_; }
    bool operator==(const DnaSequence& o) const { return packed_ == o.packed_ && even_length_ == o.even_length_; }
    // archive form
    const std::vector<unsigned char>& packed() const { return packed_; }
    bool even_length() const { return even_length_; }
    static DnaSequence from_archive(std::vector<unsigned char> packed, bool even_length, bool undefined);

private:
    std::vector<unsigned char> packed_;
    bool even_length_ = true, undefined_ = false;
};

/** One record of the output VCF: a (single, un-merged) variant with its alleles, what the panel paths carry and the
 *  genotype likelihoods that belong to it. */
struct VcfSite {
    std::string chromosome;
    size_t start = 0;                         // 0-based
    std::vector<std::string> alleles;         // [0] = REF
    std::vector<bool> undefined;              // allele contains a base outside ACGT
    std::vector<unsigned short> paths;        // allele of every panel path
    GenotypingResult likelihoods;             // over this record's alleles
    std::vector<unsigned short> sampled;      // allele of every sampled haplotype (records() with a sampled panel)
};

/** what the HaplotypeSampler left of a bubble: the bubble allele of every sampled haplotype and the number of unique k-mers
 *  still counted (reference src/sampledpanel.hpp; filled from UniqueKmers::get_path_ids + size(), src/commands.cpp:994-1005) */
struct SampledPanel {
    std::vector<unsigned short> path_to_allele;
    size_t unique_kmers = 0;
};

/** A variant bubble of the graph: one or several VCF records closer than the k-mer size, merged
 *  (reference src/variant.hpp:28-121). */
class Variant {
public:
    Variant() = default;
    /** a bubble from its stored parts (what the archive holds; see the layout above): `records` = the allele strings of
     *  every merged VCF record, `between` = the reference sequence between neighbouring records, `combinations` = for
     *  every bubble allele the allele of each record, `paths` = bubble allele of every panel path */
    static Variant from_parts(const std::string& chromosome, size_t start_position, const std::string& left_flank, const std::string& right_flank,
                              const std::vector<std::vector<std::string>>& records, const std::vector<std::string>& between,
                              const std::vector<std::vector<unsigned short>>& combinations, const std::vector<unsigned short>& paths,
                              bool flanks_added);
    size_t nr_of_alleles() const { return allele_combinations_.size(); }
    size_t nr_of_paths() const { return paths_.size(); }
    size_t nr_of_records() const { return allele_sequences_.size(); }
    bool is_combined() const { return allele_sequences_.size() > 1; }
    size_t get_start_position() const { return start_position_; }
    size_t get_end_position() const;
    const std::string& get_chromosome() const { return chromosome_; }
    unsigned short get_allele_on_path(size_t path) const { return paths_.at(path); }
    /** sequence of bubble allele `index`: the records' alleles joined by the reference between them, with the flanks
     *  when the bubble carries them */
    std::string get_allele_string(size_t index) const;
    bool is_undefined_allele(size_t index) const;
    /** the bubble as its single records (positions, alleles, path alleles), and `result` — likelihoods over bubble
     *  alleles — folded onto each record's own alleles (genotypes that agree on a record's alleles add up);
     *  coverage / unique k-mer count are the bubble's.  `result` may be null. */
    std::vector<VcfSite> records(const GenotypingResult* result, const SampledPanel* sampled = nullptr) const;

private:
    friend class Graph;
    DnaSequence left_flank_, right_flank_;
    std::vector<DnaSequence> inner_flanks_;
    std::string chromosome_;
    size_t start_position_ = 0;
    std::vector<std::vector<DnaSequence>> allele_sequences_;
    std::vector<std::vector<unsigned short>> allele_combinations_, uncovered_alleles_;
    std::vector<unsigned short> paths_;
    bool flanks_added_ = false;
};

class Graph {
public:
    Graph() = default;
    /** throw std::runtime_error on malformed input */
    static Graph parse(const std::vector<unsigned char>& bytes);
    static Graph load(const std::string& path);
    std::vector<unsigned char> serialize() const;
    /** a graph from bubbles that exist already (tests; hosts that build their bubbles elsewhere): `variant_ids` = one row
     *  per single VCF record in graph order, the ids of its ALT alleles in the lexicographic order of the allele sequences
     *  (what the archive stores), or empty */
    static Graph from_parts(const std::string& chromosome, size_t kmer_size, bool reference_added, const std::vector<Variant>& variants,
                            const std::vector<std::vector<std::string>>& variant_ids);
    /** the same with the chromosome's reference sequence (what PanGenie-index keeps in the graph: index_builder.hpp) */
    static Graph from_parts(const std::string& chromosome, size_t kmer_size, bool reference_added, const std::vector<Variant>& variants,
                            const std::vector<std::vector<std::string>>& variant_ids, const std::string& reference_bases);

    size_t get_kmer_size() const { return kmer_size_; }
    const std::string& get_chromosome() const { return chromosome_; }
    bool reference_added() const { return add_reference_; }
    size_t size() const { return variants_.size(); }
    const Variant& get_variant(size_t index) const;
    const std::vector<std::vector<std::string>>& variant_ids() const { return variant_ids_; }
    /** reference sequence the graph was built on */
    std::string reference(const std::string& name) const;

    /** the header lines of a genotyped VCF (reference src/graph.cpp:137-149); `date` = yyyymmdd, today when empty */
    static std::vector<std::string> genotypes_header(const std::string& sample, const std::string& date = "");
    /** the record lines: one per single variant, in graph order; `genotyping_result` = one (normalised) result per
     *  bubble, as HMM::get_genotyping_result() returns them */
    std::vector<std::string> genotypes_records(const std::vector<GenotypingResult>& genotyping_result, bool ignore_imputed = false) const;
    /** header (when asked for) + records into `filename` (appending without header), like the reference's
     *  Graph::write_genotypes */
    void write_genotypes(const std::string& filename, const std::vector<GenotypingResult>& genotyping_result, bool write_header,
                         const std::string& sample, bool ignore_imputed = false) const;

    /** the phasing VCF of a `-p` run (reference Graph::write_phasing, src/graph.cpp:280-412): the same fixed columns, `GT:KC`
     *  with the Viterbi haplotypes `a|b` (HMM with run_phasing; `.` for an allele of undefined sequence) */
    static std::vector<std::string> phasing_header(const std::string& sample, const std::string& date = "");
    std::vector<std::string> phasing_records(const std::vector<GenotypingResult>& genotyping_result, bool ignore_imputed = false) const;
    void write_phasing(const std::string& filename, const std::vector<GenotypingResult>& genotyping_result, bool write_header,
                       const std::string& sample, bool ignore_imputed = false) const;

    /** the sampled panel as a VCF (`-d`; reference Graph::write_sampled_panel, src/graph.cpp:414-547): one column per sampled
     *  haplotype (`sampledHT<i>`), its allele among the record's defined ones, `.` where it carries undefined sequence */
    static std::vector<std::string> sampled_panel_header(size_t nr_paths, const std::string& date = "");
    std::vector<std::string> sampled_panel_records(const std::vector<SampledPanel>& sampled_paths) const;
    void write_sampled_panel(const std::string& filename, const std::vector<SampledPanel>& sampled_paths, bool write_header) const;

private:
    std::vector<std::string> sample_records(const std::vector<GenotypingResult>& genotyping_result, bool ignore_imputed, bool phasing,
                                            const std::vector<SampledPanel>* sampled_paths = nullptr) const;
    std::vector<std::pair<std::string, std::shared_ptr<DnaSequence>>> fasta_;   // name -> sequence, archive (= sorted) order
    std::string chromosome_;
    size_t kmer_size_ = 0;
    bool add_reference_ = false, variants_deleted_ = false;
    std::vector<std::shared_ptr<Variant>> variants_;
    std::vector<std::vector<std::string>> variant_ids_;
};

}  // namespace pangenie
