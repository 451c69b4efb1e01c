// index_builder.cpp — see index_builder.hpp.
#include "index_builder.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string_view>

namespace pangenie {

namespace {

std::vector<std::string_view> split_view(std::string_view text, char sep) {   // (like repeated std::getline: a trailing separator opens no field)
    std::vector<std::string_view> out;
    size_t at = 0;
    while (at < text.size()) {
        const size_t next = text.find(sep, at);
        out.push_back(text.substr(at, next == std::string_view::npos ? std::string_view::npos : next - at));
        if (next == std::string_view::npos) break;
        at = next + 1;
    }
    return out;
}

inline bool is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/** the letters a DnaSequence gives back for `text`: ACGT (either case) stay, everything else reads N */
std::string normalised(std::string_view text) {
    std::string s(text);
    for (char& c : s) {
        switch (c) {
            case 'A': case 'a': c = 'A'; break;
            case 'C': case 'c': c = 'C'; break;
            case 'G': case 'g': c = 'G'; break;
            case 'T': case 't': c = 'T'; break;
            default: c = 'N';
        }
    }
    return s;
}

/** one VCF record that made it into the graph, before bubbles are formed */
struct Record {
    size_t start = 0, end = 0;                    // 0-based, end exclusive
    std::vector<std::string> alleles;             // [0] = REF; one "N" per missing haplotype at the end
    std::vector<unsigned short> paths;            // allele of every panel path
    std::vector<std::string> ids;                 // INFO ID=, one per ALT allele (may be empty)
};

/** Forward (not canonical) 2-bit code of every window of k letters over ACGT in `seq`, with how often it occurs there.
 *  A sequence SHORTER than k gives one entry all the same: the reference's enumeration (stepwise_unique_kmers,
 *  src/stepwiseuniquekmercomputer.cpp:11-35) shifts the letters into a k-mer register that starts as k A's and counts the
 *  register once more after its loop, so what it sees of n < k letters is (k - n) A's followed by them.  That happens to the
 *  reference stretch between two bubbles exactly k - 1 apart (closer ones are merged), and the k-mer is a real one of the
 *  graph whenever the bubble before ends in A's.  (With a letter outside ACGT among the n the register's content is
 *  Jellyfish's business; no entry then.) */
std::map<uint64_t, size_t> window_counts(const std::string& seq, size_t k) {
    std::map<uint64_t, size_t> counts;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t code = 0;
    size_t valid = 0;
    for (const char c : seq) {
        if (!is_acgt(c)) { valid = 0; code = 0; continue; }
        code = ((code << 2) | (uint64_t)(c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3)) & mask;
        if (++valid >= k) counts[code] += 1;
    }
    if (seq.size() < k && valid == seq.size()) counts[code] += 1;   // (A = 0: the padded register is the code so far)
    return counts;
}

std::string code_to_kmer(uint64_t code, size_t k) {
    std::string s(k, 'A');
    for (size_t i = 0; i < k; ++i) s[k - 1 - i] = "ACGT"[(code >> (2 * i)) & 3];
    return s;
}

}  // namespace

// ------------------------------------------------------------------ reference sequences
ReferenceSequences::ReferenceSequences(const std::string& fasta) {
    std::ifstream in(fasta);
    if (!in.good()) throw std::runtime_error("ReferenceSequences: reference file cannot be opened.");
    std::string line, *current = nullptr;
    while (std::getline(in, line)) {
        const size_t first = line.find_first_not_of(" \t\r\n");
        if (first == std::string::npos) continue;
        const size_t last = line.find_last_not_of(" \t\r\n");
        const std::string_view text(line.data() + first, last - first + 1);
        if (text[0] == '>') {
            const size_t from = text.find_first_not_of(" \t", 1);
            std::string name;
            if (from != std::string_view::npos) {
                const size_t to = text.find_first_of(" \t", from);
                name = std::string(text.substr(from, to == std::string_view::npos ? std::string_view::npos : to - from));
            }
            current = &bases_[name];
            current->clear();   // a name seen twice: the later record replaces the earlier one
        } else {
            if (!current) throw std::runtime_error("ReferenceSequences: file is malformatted.");
            current->append(normalised(text));
        }
    }
}

const std::string& ReferenceSequences::of(const std::string& name) const {
    const auto it = bases_.find(name);
    if (it == bases_.end()) throw std::runtime_error("ReferenceSequences: chromosome " + name + " is not present in FASTA-file.");
    return it->second;
}

std::vector<std::string> ReferenceSequences::names() const {
    std::vector<std::string> out;
    for (const auto& e : bases_) out.push_back(e.first);
    return out;
}

// ------------------------------------------------------------------ VCF -> bubbles
namespace {

/** A run of records less than k-1 bases apart becomes ONE bubble: its alleles are the distinct combinations of record alleles
 *  the panel paths carry (plus all-REF), in lexicographic order of the combinations — what merging the records pairwise from
 *  the left with an ordered map arrives at. */
Variant bubble_of(const std::string& chromosome, const std::vector<Record>& run, const std::string& reference, size_t k) {
    const size_t n_paths = run.front().paths.size();
    if (run.size() == 1) {
        // a record on its own keeps EVERY allele of the VCF line, carried by a path or not, bubble allele = VCF allele
        // (Variant's constructor, src/variant.cpp:52-77; Graph::add_variant_cluster prunes nothing): the sequences of
        // uncovered alleles stay in the graph (segment file, k-mer counts, exclusion of shared k-mers), a record whose paths
        // carry alleles 0 and 2 stays a three-allele object.  Only MERGING reduces to the combinations the paths carry.
        const Record& rec = run.front();
        if (rec.alleles.size() > 65535) throw std::runtime_error("build_graphs: more than 65535 alleles in one bubble");
        std::vector<std::vector<unsigned short>> combinations;
        for (size_t a = 0; a < rec.alleles.size(); ++a) combinations.push_back({(unsigned short)a});
        return Variant::from_parts(chromosome, rec.start, reference.substr(rec.start - (k - 1), k - 1), reference.substr(rec.end, k - 1),
                                   {rec.alleles}, {}, combinations, rec.paths, /*flanks_added=*/true);
    }
    std::set<std::vector<unsigned short>> distinct;
    distinct.insert(std::vector<unsigned short>(run.size(), 0));
    std::vector<std::vector<unsigned short>> of_path(n_paths, std::vector<unsigned short>(run.size()));
    for (size_t p = 0; p < n_paths; ++p) {
        for (size_t r = 0; r < run.size(); ++r) of_path[p][r] = run[r].paths[p];
        distinct.insert(of_path[p]);
    }
    if (distinct.size() > 65535) throw std::runtime_error("build_graphs: more than 65535 alleles in one bubble");
    const std::vector<std::vector<unsigned short>> combinations(distinct.begin(), distinct.end());
    std::vector<unsigned short> paths(n_paths);
    for (size_t p = 0; p < n_paths; ++p)
        paths[p] = (unsigned short)(std::lower_bound(combinations.begin(), combinations.end(), of_path[p]) - combinations.begin());
    std::vector<std::vector<std::string>> records;
    std::vector<std::string> between;
    for (size_t r = 0; r < run.size(); ++r) {
        records.push_back(run[r].alleles);
        if (r + 1 < run.size()) between.push_back(reference.substr(run[r].end, run[r + 1].start - run[r].end));
    }
    const size_t start = run.front().start, end = run.back().end;
    return Variant::from_parts(chromosome, start, reference.substr(start - (k - 1), k - 1), reference.substr(end, k - 1), records, between,
                               combinations, paths, /*flanks_added=*/true);
}

/** ids of a record's ALT alleles in the lexicographic order of the (defined) ALT sequences — the order the archive keeps */
std::vector<std::string> ordered_ids(const Record& rec) {
    if (rec.ids.empty()) return {};
    std::vector<std::string> alts;
    for (size_t a = 1; a < rec.alleles.size(); ++a)
        if (rec.alleles[a].find('N') == std::string::npos) alts.push_back(rec.alleles[a]);
    if (alts.size() != rec.ids.size()) throw std::runtime_error("build_graphs: number of variant IDs does not match the number of alternative alleles");
    std::vector<size_t> order(alts.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return alts[a] < alts[b]; });   // (not stable, like the reference's; equal ALT sequences do not occur in a VCF record)
    std::vector<std::string> out;
    for (const size_t i : order) out.push_back(rec.ids[i]);
    return out;
}

}  // namespace

BuiltGraphs build_graphs(const std::string& vcf, const ReferenceSequences& reference, size_t k, bool add_reference) {
    if (k < 2 || k > 32) throw std::runtime_error("build_graphs: k-mer size must be 2..32");
    if (vcf.size() >= 3 && vcf.compare(vcf.size() - 3, 3, ".gz") == 0) throw std::runtime_error("build_graphs: Uncompressed VCF-file is required.");
    std::ifstream in(vcf);
    if (!in.good()) throw std::runtime_error("build_graphs: input VCF file cannot be opened.");
    static const char* const kColumns[9] = {"#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"};

    BuiltGraphs out;
    struct Chromosome { std::vector<Variant> bubbles; std::vector<std::vector<std::string>> ids; };
    std::map<std::string, Chromosome> done;
    std::string chrom;                    // chromosome of the run in progress
    std::vector<Record> run;
    size_t previous_end = 0;
    auto close_run = [&]() {
        if (run.empty()) return;
        Chromosome& c = done[chrom];
        c.bubbles.push_back(bubble_of(chrom, run, reference.of(chrom), k));
        for (const Record& r : run) c.ids.push_back(ordered_ids(r));
        run.clear();
    };

    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        const std::vector<std::string_view> col = split_view(line, '\t');
        if (col.empty()) continue;
        if (col[0].substr(0, 2) == "##") continue;
        if (col[0][0] == '#') {
            if (col.size() < 9) throw std::runtime_error("build_graphs: not a proper VCF-file.");
            if (col.size() < 10) throw std::runtime_error("build_graphs: no haplotype paths given.");
            for (int i = 0; i < 9; ++i)
                if (col[i] != kColumns[i]) throw std::runtime_error("build_graphs: VCF header line is malformed.");
            out.nr_paths = (col.size() - 9) * 2 + (add_reference ? 1 : 0);
            if (out.nr_paths > 65535) throw std::runtime_error("build_graphs: number of paths is limited to 65534 in current implementation.");
            continue;
        }
        if (col.size() < 10) throw std::runtime_error("build_graphs: malformed VCF-file, or no haplotype paths given in VCF.");
        const std::string name(col[0]);
        Record rec;
        {   // VCF positions are 1-based decimal numbers
            unsigned long long pos = 0;
            bool digits = !col[1].empty() && col[1].size() <= 18;
            for (const char c : col[1]) { digits = digits && c >= '0' && c <= '9'; pos = pos * 10 + (unsigned long long)(c - '0'); }
            if (!digits || pos == 0) throw std::runtime_error("build_graphs: malformed VCF-file: position '" + std::string(col[1]) + "'");
            rec.start = (size_t)pos - 1;
        }
        if (name == chrom && rec.start < previous_end)
            throw std::runtime_error("build_graphs: variant at " + name + ":" + std::to_string(rec.start) + " overlaps previous one. VCF does not represent a pangenome graph.");
        // (the reference hands a chromosome's sequence over to its Graph when the first bubble is seen; records of that
        // chromosome after another one's no longer find it: src/graphbuilder.cpp:142-146, src/fastareader.cpp:86-92)
        if (name != chrom && done.count(name))
            throw std::runtime_error("build_graphs: chromosome " + name + " is not present in FASTA-file. (its records are not in one block of the VCF)");
        const std::string& ref_bases = reference.of(name);
        const std::string ref_allele = normalised(col[3]);
        rec.end = rec.start + ref_allele.size();
        if (rec.end > ref_bases.size() || ref_bases.compare(rec.start, ref_allele.size(), ref_allele) != 0)
            throw std::runtime_error("build_graphs: reference allele given in VCF does not match allele in reference fasta file at that position.");
        // ALT alleles have to be spelled out over ACGT
        bool explicit_alt = !col[4].empty();
        for (const char c : col[4]) explicit_alt = explicit_alt && (c == ',' || normalised(std::string_view(&c, 1))[0] != 'N');
        if (!explicit_alt) { out.skipped += 1; continue; }
        rec.alleles.push_back(ref_allele);
        for (const std::string_view alt : split_view(col[4], ',')) rec.alleles.push_back(normalised(alt));
        if (rec.alleles.size() > 65535) throw std::runtime_error("build_graphs: number of alternative alleles is limited to 65534 in current implementation.");
        // too close to an end of the chromosome for flanks and overhangs
        if (rec.start < 2 * k || rec.end + 2 * k > ref_bases.size()) { out.skipped += 1; continue; }
        if (name != chrom || rec.start - previous_end >= k - 1) {   // the record starts a new run
            close_run();
            chrom = name;
        }
        for (const std::string_view field : split_view(col[7], ';'))
            if (field.substr(0, 3) == "ID=")
                for (const std::string_view id : split_view(field.substr(3), ',')) rec.ids.emplace_back(id);
        if (add_reference) rec.paths.push_back(0);
        for (size_t s = 9; s < col.size(); ++s) {
            if (col[s].find('/') != std::string_view::npos) throw std::runtime_error("build_graphs: Found unphased genotype.");
            const std::vector<std::string_view> gt = split_view(col[s], '|');
            if (gt.size() != 2) throw std::runtime_error("build_graphs: Found invalid genotype. Genotypes must be diploid (.|. if missing).");
            for (const std::string_view g : gt) {
                if (g == ".") {   // a missing haplotype gets an allele of its own: "N"
                    rec.paths.push_back((unsigned short)rec.alleles.size());
                    rec.alleles.push_back("N");
                } else {
                    const unsigned long a = std::strtoul(std::string(g).c_str(), nullptr, 10);
                    if (a >= rec.alleles.size()) throw std::runtime_error("build_graphs: invalid genotype in VCF.");
                    rec.paths.push_back((unsigned short)a);
                }
            }
        }
        if (rec.paths.size() != out.nr_paths) throw std::runtime_error("build_graphs: a record does not list every sample of the header");
        previous_end = rec.end;
        run.push_back(std::move(rec));
    }
    close_run();

    std::vector<std::pair<size_t, std::string>> by_size;
    for (auto& e : done) {
        by_size.emplace_back(e.second.bubbles.size(), e.first);
        out.graphs.emplace(e.first, Graph::from_parts(e.first, k, add_reference, e.second.bubbles, e.second.ids, reference.of(e.first)));
    }
    std::sort(by_size.rbegin(), by_size.rend());
    for (const auto& e : by_size) out.chromosomes.push_back(e.second);
    return out;
}

// ------------------------------------------------------------------ the segment file
std::string path_segments_fasta(const BuiltGraphs& built, const ReferenceSequences& reference) {
    std::ostringstream out;
    std::vector<std::string> order = built.chromosomes;   // chromosomes of the VCF first, then the rest of the reference by name
    for (const std::string& name : reference.names())
        if (!built.graphs.count(name)) order.push_back(name);
    for (const std::string& name : order) {
        const std::string& bases = reference.of(name);
        size_t covered = 0;
        const auto g = built.graphs.find(name);
        if (g != built.graphs.end()) {
            for (size_t v = 0; v < g->second.size(); ++v) {
                const Variant& bubble = g->second.get_variant(v);
                out << '>' << name << "_reference_" << bubble.get_start_position() << '\n' << bases.substr(covered, bubble.get_start_position() - covered) << '\n';
                for (size_t a = 0; a < bubble.nr_of_alleles(); ++a)
                    out << '>' << name << '_' << bubble.get_start_position() << '_' << a << '\n' << bubble.get_allele_string(a) << '\n';
                covered = bubble.get_end_position();
            }
        }
        out << '>' << name << "_reference_end\n" << bases.substr(covered) << '\n';
    }
    return out.str();
}

// ------------------------------------------------------------------ unique k-mers of a chromosome
namespace {
// the reference on either side of bubble v: up to 2 k bases, not into the neighbouring bubbles
struct Flanks { size_t left_from, start, end, right_to; };
Flanks flanks_of(const Graph& graph, size_t v, size_t k, size_t reference_size) {
    const Variant& bubble = graph.get_variant(v);
    const size_t start = bubble.get_start_position(), end = bubble.get_end_position();
    const size_t left_limit = v > 0 ? graph.get_variant(v - 1).get_end_position() : 0;
    const size_t right_limit = v + 1 < graph.size() ? graph.get_variant(v + 1).get_start_position() : reference_size;
    return {std::max(left_limit, start >= 2 * k ? start - 2 * k : 0), start, end, std::min(right_limit, end + 2 * k)};
}
}  // namespace

void register_candidate_kmers(const Graph& graph, TargetedKmerCounter& counter) {
    const size_t k = graph.get_kmer_size();
    const std::string reference = graph.reference(graph.get_chromosome());
    for (size_t v = 0; v < graph.size(); ++v) {
        const Variant& bubble = graph.get_variant(v);
        for (size_t a = 0; a < bubble.nr_of_alleles(); ++a)
            if (!bubble.is_undefined_allele(a)) counter.add_targets_of(bubble.get_allele_string(a));
        const Flanks f = flanks_of(graph, v, k, reference.size());
        for (const std::string_view stretch : {std::string_view(reference).substr(f.left_from, f.start - f.left_from),
                                               std::string_view(reference).substr(f.end, f.right_to - f.end)}) {
            counter.add_targets_of(stretch);
            if (stretch.size() < k) counter.add_target(std::string(k - stretch.size(), 'A') + std::string(stretch));   // (window_counts: the padded register)
        }
    }
}

ChromosomeKmers unique_kmers_of(const Graph& graph, KmerCounter& graph_kmers, unsigned threads) {
    const size_t k = graph.get_kmer_size();
    const std::string reference = graph.reference(graph.get_chromosome());
    ChromosomeKmers out;
    out.rows.resize(graph.size());
    out.objects.resize(graph.size());
    // up to 12 k-mers of a reference stretch that occur once in it and once in the whole graph, smallest first
    auto single_copy = [&](const std::string& stretch, std::vector<std::string>& into) {
        size_t taken = 0;
        for (const auto& e : window_counts(stretch, k)) {
            if (taken >= 12) break;
            if (e.second != 1) continue;
            const std::string kmer = code_to_kmer(e.first, k);
            if (graph_kmers.getKmerAbundance(kmer) == 1) { into.push_back(kmer); taken += 1; }
        }
    };
    // one bubble: independent of every other (the counter is only read), so bubbles are dealt to `threads` workers
    auto one_bubble = [&](size_t v) {
        const Variant& bubble = graph.get_variant(v);
        std::vector<unsigned short> path_alleles(bubble.nr_of_paths());
        bool two_alleles_only = true;
        for (size_t p = 0; p < path_alleles.size(); ++p) {
            path_alleles[p] = bubble.get_allele_on_path(p);
            two_alleles_only = two_alleles_only && path_alleles[p] <= 1;
        }
        std::shared_ptr<UniqueKmers> object;
        if (two_alleles_only) object = std::make_shared<BiallelicUniqueKmers>(bubble.get_start_position(), path_alleles);
        else object = std::make_shared<MultiallelicUniqueKmers>(bubble.get_start_position(), path_alleles);
        object->set_coverage(0);
        // k-mers that occur exactly once inside an allele -> the alleles they do that in
        std::map<uint64_t, std::vector<unsigned short>> once_in;
        for (size_t a = 0; a < bubble.nr_of_alleles(); ++a) {
            if (bubble.is_undefined_allele(a)) { object->set_undefined_allele((unsigned short)a); continue; }
            for (const auto& e : window_counts(bubble.get_allele_string(a), k))
                if (e.second == 1) once_in[e.first].push_back((unsigned short)a);
        }
        // candidates per allele, smallest k-mer first: on one allele only, nowhere else in the graph, allele carried by a path
        std::set<unsigned short> carried(path_alleles.begin(), path_alleles.end());
        std::map<unsigned short, std::vector<uint64_t>> candidates;
        for (const auto& e : once_in) {
            if (e.second.size() != 1 || !carried.count(e.second[0])) continue;
            if (graph_kmers.getKmerAbundance(code_to_kmer(e.first, k)) != 1) continue;
            candidates[e.second[0]].push_back(e.first);
        }
        // dealt out allele by allele, one k-mer a round: at most 16 (two-allele objects) / 32 per allele, max(paths, 301) in all
        const size_t per_allele = two_alleles_only ? 16 : 32, in_all = std::max<size_t>(path_alleles.size(), 301);
        std::map<unsigned short, std::vector<uint64_t>> chosen;
        size_t total = 0;
        for (size_t round = 0; total < in_all; ++round) {
            bool any = false;
            for (const auto& c : candidates) {
                if (round < c.second.size() && round < per_allele) {
                    chosen[c.first].push_back(c.second[round]);
                    any = true;
                    if (++total >= in_all) break;
                }
            }
            if (!any) break;
        }
        std::string unique_column;
        for (const auto& c : chosen)
            for (const uint64_t code : c.second) {
                std::vector<unsigned short> on = {c.first};
                object->insert_kmer(0, on);
                if (!unique_column.empty()) unique_column += ',';
                unique_column += code_to_kmer(code, k);
            }
        const Flanks f = flanks_of(graph, v, k, reference.size());
        const size_t start = f.start, end = f.end;
        std::vector<std::string> flanking;
        single_copy(reference.substr(f.left_from, f.start - f.left_from), flanking);
        single_copy(reference.substr(f.end, f.right_to - f.end), flanking);
        std::string flank_column;
        for (const std::string& f : flanking) { if (!flank_column.empty()) flank_column += ','; flank_column += f; }
        out.rows[v] = bubble.get_chromosome() + '\t' + std::to_string(start) + '\t' + std::to_string(end) + '\t' +
                      (unique_column.empty() ? "nan" : unique_column) + '\t' + (flank_column.empty() ? "nan" : flank_column);
        out.objects[v] = object;
    };
    if (threads <= 1 || graph.size() < 64) {
        for (size_t v = 0; v < graph.size(); ++v) one_bubble(v);
        return out;
    }
    std::atomic<size_t> next{0};
    std::exception_ptr failure;
    std::mutex failure_lock;
    std::vector<std::thread> workers;
    for (unsigned t = 0; t < threads; ++t)
        workers.emplace_back([&] {
            try {
                for (size_t from = next.fetch_add(64); from < graph.size(); from = next.fetch_add(64))
                    for (size_t v = from; v < std::min(from + 64, graph.size()); ++v) one_bubble(v);
            } catch (...) {
                std::lock_guard<std::mutex> hold(failure_lock);
                if (!failure) failure = std::current_exception();
                next.store(graph.size());
            }
        });
    for (std::thread& w : workers) w.join();
    if (failure) std::rethrow_exception(failure);
    return out;
}

// the k-mer table as gzip: the rows are cut into pieces of a few MB, every piece becomes a gzip member of its own (deflated by
// one of `threads` workers), the members follow each other in the file — which gzread / zcat read as one stream
static void write_gz_members(const std::string& path, const std::string& header, const std::vector<std::string>& rows, unsigned threads) {
    std::vector<std::string> pieces(1, header);
    for (const std::string& row : rows) {
        if (pieces.back().size() > (4u << 20)) pieces.emplace_back();
        pieces.back() += row;
        pieces.back() += '\n';
    }
    std::vector<std::vector<unsigned char>> members(pieces.size());
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto work = [&] {
        for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) {
            z_stream z{};
            if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed = true; return; }
            members[i].resize(deflateBound(&z, (uLong)pieces[i].size()) + 64);
            z.next_in = (Bytef*)pieces[i].data(); z.avail_in = (uInt)pieces[i].size();
            z.next_out = members[i].data(); z.avail_out = (uInt)members[i].size();
            const int rc = deflate(&z, Z_FINISH);
            members[i].resize(z.total_out);
            deflateEnd(&z);
            if (rc != Z_STREAM_END) { failed = true; return; }
            std::string().swap(pieces[i]);
        }
    };
    if (threads <= 1 || pieces.size() == 1) work();
    else {
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < std::min<size_t>(threads, pieces.size()); ++t) workers.emplace_back(work);
        for (std::thread& w : workers) w.join();
    }
    std::ofstream f(path, std::ios::binary);
    if (failed || !f.good()) throw std::runtime_error("build_index: File " + path + " cannot be created. Note that the filename must not contain non-existing directories.");
    for (const auto& m : members) f.write((const char*)m.data(), (std::streamsize)m.size());
    if (!f.good()) throw std::runtime_error("build_index: File " + path + " cannot be written.");
}

// ------------------------------------------------------------------ everything
std::vector<std::string> build_index(const std::string& reference_fasta, const std::string& vcf, const std::string& prefix, size_t k, bool add_reference,
                                     unsigned threads, bool whole_graph_counts) {
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    const bool verbose = std::getenv("PG_INDEX_VERBOSE") != nullptr;   // stage times on stderr
    auto clock = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (verbose) std::fprintf(stderr, "build_index: %-28s %8.3f s\n", what, std::chrono::duration<double>(now - clock).count());
        clock = now;
    };
    const ReferenceSequences reference(reference_fasta);
    stage("reference read");
    const BuiltGraphs built = build_graphs(vcf, reference, k, add_reference);
    stage("graphs built");
    const std::string segments = prefix + "_path_segments.fasta";
    {
        std::ofstream f(segments);
        if (!f.good()) throw std::runtime_error("build_index: File " + segments + " cannot be created. Note that the filename must not contain non-existing directories.");
        f << path_segments_fasta(built, reference);
    }
    stage("segment file written");
    std::unique_ptr<ExactKmerCounter> whole_graph;
    if (whole_graph_counts) {
        whole_graph.reset(new ExactKmerCounter(segments, k));
        stage("graph k-mers counted");
    }
    UniqueKmersMap map;
    map.kmersize = k;
    map.add_reference = add_reference;
    for (const std::string& name : built.chromosomes) {
        const Graph& graph = built.graphs.at(name);
        {
            const std::vector<unsigned char> bytes = graph.serialize();
            std::ofstream f(prefix + "_" + name + "_Graph.cereal", std::ios::binary);
            if (!f.good()) throw std::runtime_error("build_index: cannot write the graph of " + name);
            f.write((const char*)bytes.data(), (std::streamsize)bytes.size());
        }
        stage("graph archive written");
        std::unique_ptr<TargetedKmerCounter> asked;
        if (!whole_graph_counts) {
            asked.reset(new TargetedKmerCounter(k));
            register_candidate_kmers(graph, *asked);
            asked->count(segments, threads);
            stage("candidate k-mers counted");
        }
        KmerCounter& graph_kmers = whole_graph_counts ? static_cast<KmerCounter&>(*whole_graph) : static_cast<KmerCounter&>(*asked);
        ChromosomeKmers kmers = unique_kmers_of(graph, graph_kmers, threads);
        stage("unique k-mers selected");
        write_gz_members(prefix + "_" + name + "_kmers.tsv.gz", "#chromosome\tstart\tend\tunique_kmers\tunique_kmers_overhang\n", kmers.rows, threads);
        map.unique_kmers[name] = std::move(kmers.objects);
        map.runtimes[name] = 0.0;
        stage("k-mer table written");
    }
    save_unique_kmers_map(map, prefix + "_UniqueKmersMap.cereal");
    stage("index archive written");
    return built.chromosomes;
}

}  // namespace pangenie
