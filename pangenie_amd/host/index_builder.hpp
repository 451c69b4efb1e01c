// index_builder.hpp — the step in FRONT of the path: a phased multi-sample VCF and its reference become the files
// PanGenie-genotype starts from (reference run_index_command, src/commands.cpp:592-735):
//   <prefix>_path_segments.fasta      every sequence of the graph: reference stretches between bubbles, every allele of every
//                                     bubble with k-1 flanking bases (GraphBuilder::write_path_segments, src/graphbuilder.cpp:290-352)
//   <prefix>_<chromosome>_Graph.cereal   the bubbles: VCF records closer than k-1 bases merged, the alleles the panel paths carry
//                                     (GraphBuilder::construct_graph, src/graphbuilder.cpp:69-276; Variant::combine_variants,
//                                     src/variant.cpp:249-306; Graph::add_variant_cluster, src/graph.cpp:67-104)
//   <prefix>_<chromosome>_kmers.tsv.gz   per bubble the k-mers that identify one allele and occur nowhere else in the graph, and up
//                                     to 12 + 12 single-copy k-mers of the reference on either side (StepwiseUniqueKmerComputer,
//                                     src/stepwiseuniquekmercomputer.cpp:40-264)
//   <prefix>_UniqueKmersMap.cereal    the same k-mers as UniqueKmers objects without counts — the index fill_read_kmercounts fills
// Host code ("config #1 is reproduced only after the next rows: own k-mer counter + graph builder"); the k-mer
// counts over the graph's sequences come from ExactKmerCounter (the reference runs Jellyfish over the segment file).
// Pinned on the reference's own fixture: tests/data/region.fa + region.vcf at k = 31 give its tests/data/index_* files — the
// segment file and the k-mer table text for text, the two archives byte for byte (tests/cpp/test_host.cpp).
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "cereal_io.hpp"
#include "graph_io.hpp"
#include "kmer_counts.hpp"

namespace pangenie {

/** The reference sequences of a FASTA file as the graph sees them: names up to the first blank, letters as DnaSequence stores
 *  them (ACGT in either case, everything else N). */
class ReferenceSequences {
public:
    explicit ReferenceSequences(const std::string& fasta);
    bool contains(const std::string& name) const { return bases_.count(name) != 0; }
    const std::string& of(const std::string& name) const;     // throws std::runtime_error when absent
    std::vector<std::string> names() const;                   // sorted
private:
    std::map<std::string, std::string> bases_;
};

/** VCF + reference -> one Graph per chromosome (records merged into bubbles).  `skipped` counts records left out the way the
 *  reference leaves them out (ALT with letters outside ACGT, closer than 2 k to a chromosome end). */
struct BuiltGraphs {
    std::map<std::string, Graph> graphs;
    std::vector<std::string> chromosomes;   // by decreasing number of bubbles (ties: by name, descending) — the reference's order
    size_t nr_paths = 0, skipped = 0;
};
BuiltGraphs build_graphs(const std::string& vcf, const ReferenceSequences& reference, size_t kmer_size, bool add_reference);

/** the text of <prefix>_path_segments.fasta */
std::string path_segments_fasta(const BuiltGraphs& built, const ReferenceSequences& reference);

/** One chromosome's unique k-mers: the rows of the k-mer table (without the header) and the UniqueKmers objects (no counts,
 *  coverage 0).  `graph_kmers` must hold the canonical counts of the segment file. */
struct ChromosomeKmers {
    std::vector<std::string> rows;
    std::vector<std::shared_ptr<UniqueKmers>> objects;
};
/** `threads` workers share the bubbles; the counter is only read (getKmerAbundance must be safe to call concurrently:
 *  ExactKmerCounter is) */
ChromosomeKmers unique_kmers_of(const Graph& graph, KmerCounter& graph_kmers, unsigned threads = 1);

/** Every k-mer unique_kmers_of(graph, ...) can ask its counter about — the windows of the bubbles' alleles and of the
 *  reference stretches either side of them — registered with `counter`: a TargetedKmerCounter that then streams the
 *  segment file holds the graph's counts for ONE chromosome's questions instead of every k-mer of the graph. */
void register_candidate_kmers(const Graph& graph, TargetedKmerCounter& counter);

/** Everything at once, written under `prefix`; returns the chromosomes in the reference's order.  The graph's k-mer counts
 *  are taken chromosome by chromosome (register_candidate_kmers + one pass over the segment file each, `threads` counting
 *  workers): memory follows one chromosome's bubbles, not the whole graph; `whole_graph_counts` = true keeps every k-mer
 *  of the graph in one table instead (one pass; the reference's way, with Jellyfish). */
std::vector<std::string> build_index(const std::string& reference_fasta, const std::string& vcf, const std::string& prefix,
                                     size_t kmer_size = 31, bool add_reference = true, unsigned threads = 1,   // threads 0 = all cores
                                     bool whole_graph_counts = false);

}  // namespace pangenie
