// graph_io.cpp — see graph_io.hpp.
#include "graph_io.hpp"

#include <algorithm>
#include <cmath>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <map>
#include <numeric>
#include <sstream>
#include <stdexcept>

#include "archive_bytes.hpp"

namespace pangenie {

using archive_bytes::MSB;
using archive_bytes::Reader;
using archive_bytes::Writer;

// ------------------------------------------------------------------ DnaSequence
namespace {
inline unsigned char base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
inline char base_letter(unsigned char code) { return code < 4 ? "ACGT"[code] : 'N'; }
}  // namespace

DnaSequence::DnaSequence(const std::string& bases) {
    packed_.reserve((bases.size() + 1) / 2);
    for (size_t i = 0; i < bases.size(); ++i) {
        const unsigned char c = base_code(bases[i]);
        undefined_ = undefined_ || c == 4;
        if (i % 2 == 0) packed_.push_back((unsigned char)(c << 4));
        else packed_.back() |= c;
    }
    even_length_ = bases.size() % 2 == 0;
}

DnaSequence DnaSequence::from_archive(std::vector<unsigned char> packed, bool even_length, bool undefined) {
    if (packed.empty() && !even_length) throw std::runtime_error("DnaSequence: an empty sequence of odd length");
    DnaSequence d;
    d.packed_ = std::move(packed); d.even_length_ = even_length; d.undefined_ = undefined;
    return d;
}

char DnaSequence::operator[](size_t position) const {
    if (position >= size()) throw std::runtime_error("DnaSequence::operator[]: index out of bounds.");
    const unsigned char byte = packed_[position / 2];
    return base_letter(position % 2 == 0 ? (unsigned char)(byte >> 4) : (unsigned char)(byte & 15));
}

std::string DnaSequence::to_string() const {
    std::string s(size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) s[i] = (*this)[i];
    return s;
}

void DnaSequence::append(const DnaSequence& other) {
    const size_t n = other.size();
    for (size_t i = 0; i < n; ++i) {  // nibble by nibble: the packed form of the concatenation
        const unsigned char byte = other.packed_[i / 2];
        const unsigned char code = i % 2 == 0 ? (unsigned char)(byte >> 4) : (unsigned char)(byte & 15);
        if (even_length_) packed_.push_back((unsigned char)(code << 4));
        else packed_.back() |= code;
        even_length_ = !even_length_;
    }
    undefined_ = undefined_ || other.undefined_;
}

// ------------------------------------------------------------------ Variant
Variant Variant::from_parts(const std::string& chromosome, size_t start_position, const std::string& left_flank, const std::string& right_flank,
                            const std::vector<std::vector<std::string>>& records, const std::vector<std::string>& between,
                            const std::vector<std::vector<unsigned short>>& combinations, const std::vector<unsigned short>& paths,
                            bool flanks_added) {
    if (records.empty() || between.size() + 1 != records.size()) throw std::runtime_error("Variant::from_parts: records / sequences between them do not match");
    Variant v;
    v.chromosome_ = chromosome; v.start_position_ = start_position;
    v.left_flank_ = DnaSequence(left_flank); v.right_flank_ = DnaSequence(right_flank);
    for (const std::string& b : between) v.inner_flanks_.push_back(DnaSequence(b));
    for (const auto& rec : records) {
        if (rec.empty()) throw std::runtime_error("Variant::from_parts: a record without alleles");
        v.allele_sequences_.emplace_back();
        for (const std::string& a : rec) v.allele_sequences_.back().push_back(DnaSequence(a));
    }
    for (const auto& combo : combinations) {
        if (combo.size() != records.size()) throw std::runtime_error("Variant::from_parts: allele combination of the wrong length");
        for (size_t r = 0; r < combo.size(); ++r)
            if (combo[r] >= records[r].size()) throw std::runtime_error("Variant::from_parts: allele combination names an allele the record does not have");
    }
    v.allele_combinations_ = combinations;
    for (unsigned short p : paths)
        if (p >= combinations.size()) throw std::runtime_error("Variant::from_parts: a path carries an allele the variant does not have");
    v.paths_ = paths;
    // alleles of each record that no path carries
    for (size_t r = 0; r < records.size(); ++r) {
        std::vector<unsigned short> uncovered;
        for (unsigned short a = 0; a < records[r].size(); ++a) {
            bool covered = false;
            for (unsigned short p : paths) covered = covered || combinations[p][r] == a;
            if (!covered) uncovered.push_back(a);
        }
        v.uncovered_alleles_.push_back(uncovered);
    }
    v.flanks_added_ = flanks_added;
    return v;
}

size_t Variant::get_end_position() const {
    size_t end = start_position_;
    for (size_t r = 0; r < allele_sequences_.size(); ++r) {
        end += allele_sequences_[r].at(0).size();
        if (r + 1 < allele_sequences_.size()) end += inner_flanks_.at(r).size();
    }
    return end;
}

std::string Variant::get_allele_string(size_t index) const {
    if (index >= allele_combinations_.size()) throw std::runtime_error("Variant::get_allele_string: Index out of bounds.");
    const std::vector<unsigned short>& combo = allele_combinations_[index];
    DnaSequence s;
    if (flanks_added_) s = left_flank_;
    for (size_t r = 0; r < combo.size(); ++r) {
        s.append(allele_sequences_.at(r).at(combo[r]));
        if (r + 1 < combo.size()) s.append(inner_flanks_.at(r));
    }
    if (flanks_added_) s.append(right_flank_);
    return s.to_string();
}

bool Variant::is_undefined_allele(size_t index) const {
    const std::vector<unsigned short>& combo = allele_combinations_.at(index);
    for (size_t r = 0; r < combo.size(); ++r)
        if (allele_sequences_.at(r).at(combo[r]).contains_undefined()) return true;
    return false;
}

std::vector<VcfSite> Variant::records(const GenotypingResult* result, const SampledPanel* sampled) const {
    const size_t n_records = allele_sequences_.size();
    std::vector<VcfSite> out(n_records);
    size_t position = start_position_;
    for (size_t r = 0; r < n_records; ++r) {
        VcfSite& site = out[r];
        site.chromosome = chromosome_;
        site.start = position;
        for (const DnaSequence& a : allele_sequences_[r]) {
            site.alleles.push_back(a.to_string());
            site.undefined.push_back(a.contains_undefined());
        }
        // what a bubble allele means for this record
        std::vector<unsigned short> own(allele_combinations_.size());
        for (size_t a = 0; a < own.size(); ++a) {
            if (allele_combinations_[a].size() != n_records) throw std::runtime_error("Variant: allele combination of the wrong length");
            own[a] = allele_combinations_[a][r];
        }
        site.paths.reserve(paths_.size());
        for (unsigned short bubble_allele : paths_) site.paths.push_back(own.at(bubble_allele));
        if (sampled)
            for (unsigned short bubble_allele : sampled->path_to_allele) site.sampled.push_back(own.at(bubble_allele));
        if (result) {
            for (const auto& entry : result->get_stored_likelihoods())
                site.likelihoods.add_to_likelihood(own.at(entry.first.first), own.at(entry.first.second), entry.second);
            const std::pair<unsigned short, unsigned short> hap = result->get_haplotype();
            if (hap.first < own.size() && hap.second < own.size()) {
                site.likelihoods.add_first_haplotype_allele(own[hap.first]);
                site.likelihoods.add_second_haplotype_allele(own[hap.second]);
            }
            site.likelihoods.set_coverage(result->coverage());
            site.likelihoods.set_unique_kmers(result->nr_unique_kmers());
        }
        position += allele_sequences_[r].at(0).size();
        if (r + 1 < n_records) position += inner_flanks_.at(r).size();
    }
    return out;
}

// ------------------------------------------------------------------ archive
namespace {

DnaSequence read_dna(Reader& r) {
    const uint64_t n = r.count(1);
    std::vector<unsigned char> packed(r.p + r.o, r.p + r.o + n);
    r.o += (size_t)n;
    const bool even = r.take<uint8_t>() != 0, undefined = r.take<uint8_t>() != 0;
    return DnaSequence::from_archive(std::move(packed), even, undefined);
}
void write_dna(Writer& w, const DnaSequence& d) {
    w.put<uint64_t>(d.packed().size());
    w.out.insert(w.out.end(), d.packed().begin(), d.packed().end());
    w.put<uint8_t>(d.even_length() ? 1 : 0);
    w.put<uint8_t>(d.contains_undefined() ? 1 : 0);
}
std::vector<unsigned short> read_u16s(Reader& r) {
    const uint64_t n = r.count(2);
    std::vector<unsigned short> v((size_t)n);
    for (auto& x : v) x = r.take<uint16_t>();
    return v;
}
void write_u16s(Writer& w, const std::vector<unsigned short>& v) {
    w.put<uint64_t>(v.size());
    for (unsigned short x : v) w.put<uint16_t>(x);
}

}  // namespace

Graph Graph::parse(const std::vector<unsigned char>& bytes) {
    Reader r{bytes.data(), bytes.size()};
    Graph g;
    std::map<uint32_t, std::shared_ptr<DnaSequence>> sequences;   // shared-pointer id -> object
    std::map<uint32_t, std::shared_ptr<Variant>> variants;
    const uint64_t n_fasta = r.count(12);
    for (uint64_t i = 0; i < n_fasta; ++i) {
        const std::string name = r.str();
        const uint32_t id = r.take<uint32_t>();
        std::shared_ptr<DnaSequence> seq;
        if (id & MSB) { seq = std::make_shared<DnaSequence>(read_dna(r)); sequences[id & ~MSB] = seq; }
        else if (id != 0) {
            if (!sequences.count(id)) throw std::runtime_error("Graph archive: dangling pointer id");
            seq = sequences[id];
        }
        g.fasta_.emplace_back(name, seq);
    }
    g.chromosome_ = r.str();
    g.kmer_size_ = (size_t)r.take<uint64_t>();
    g.add_reference_ = r.take<uint8_t>() != 0;
    g.variants_deleted_ = r.take<uint8_t>() != 0;
    const uint64_t n_variants = r.count(4);
    for (uint64_t i = 0; i < n_variants; ++i) {
        const uint32_t id = r.take<uint32_t>();
        if (id == 0) { g.variants_.push_back(nullptr); continue; }
        if (!(id & MSB)) {
            if (!variants.count(id)) throw std::runtime_error("Graph archive: dangling pointer id");
            g.variants_.push_back(variants[id]);
            continue;
        }
        auto v = std::make_shared<Variant>();
        v->left_flank_ = read_dna(r);
        v->right_flank_ = read_dna(r);
        const uint64_t n_inner = r.count(10);
        for (uint64_t k = 0; k < n_inner; ++k) v->inner_flanks_.push_back(read_dna(r));
        v->chromosome_ = r.str();
        v->start_position_ = (size_t)r.take<uint64_t>();
        const uint64_t n_records = r.count(8);
        v->allele_sequences_.resize((size_t)n_records);
        for (auto& list : v->allele_sequences_) {
            const uint64_t n_alleles = r.count(10);
            for (uint64_t a = 0; a < n_alleles; ++a) list.push_back(read_dna(r));
        }
        const uint64_t n_combos = r.count(8);
        for (uint64_t a = 0; a < n_combos; ++a) v->allele_combinations_.push_back(read_u16s(r));
        const uint64_t n_uncovered = r.count(8);
        for (uint64_t a = 0; a < n_uncovered; ++a) v->uncovered_alleles_.push_back(read_u16s(r));
        v->paths_ = read_u16s(r);
        v->flanks_added_ = r.take<uint8_t>() != 0;
        // what the writer below relies on
        if (v->allele_sequences_.empty() || v->inner_flanks_.size() + 1 != v->allele_sequences_.size())
            throw std::runtime_error("Graph archive: a variant without records / with the wrong number of inner flanks");
        for (const auto& combo : v->allele_combinations_) {
            if (combo.size() != v->allele_sequences_.size()) throw std::runtime_error("Graph archive: allele combination of the wrong length");
            for (size_t k = 0; k < combo.size(); ++k)
                if (combo[k] >= v->allele_sequences_[k].size()) throw std::runtime_error("Graph archive: allele combination names an allele the record does not have");
        }
        for (unsigned short p : v->paths_)
            if (p >= v->allele_combinations_.size()) throw std::runtime_error("Graph archive: a path carries an allele the variant does not have");
        variants[id & ~MSB] = v;
        g.variants_.push_back(v);
    }
    const uint64_t n_ids = r.count(8);
    for (uint64_t i = 0; i < n_ids; ++i) {
        const uint64_t m = r.count(8);
        std::vector<std::string> ids;
        for (uint64_t k = 0; k < m; ++k) ids.push_back(r.str());
        g.variant_ids_.push_back(ids);
    }
    if (r.o != r.n) throw std::runtime_error("Graph archive: trailing bytes");
    return g;
}

Graph Graph::load(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse(bytes);
}

std::vector<unsigned char> Graph::serialize() const {
    Writer w;
    uint32_t next_id = 1;                             // one counter over all shared pointers of the archive
    std::map<const void*, uint32_t> seen;
    auto pointer = [&](const void* p) -> bool {       // writes the id; true = the object's data must follow
        if (!p) { w.put<uint32_t>(0); return false; }
        auto it = seen.find(p);
        if (it != seen.end()) { w.put<uint32_t>(it->second); return false; }
        seen[p] = next_id;
        w.put<uint32_t>(next_id++ | MSB);
        return true;
    };
    w.put<uint64_t>(fasta_.size());
    for (const auto& entry : fasta_) {
        w.str(entry.first);
        if (pointer(entry.second.get())) write_dna(w, *entry.second);
    }
    w.str(chromosome_);
    w.put<uint64_t>(kmer_size_);
    w.put<uint8_t>(add_reference_ ? 1 : 0);
    w.put<uint8_t>(variants_deleted_ ? 1 : 0);
    w.put<uint64_t>(variants_.size());
    for (const auto& vp : variants_) {
        if (!pointer(vp.get())) continue;
        const Variant& v = *vp;
        write_dna(w, v.left_flank_);
        write_dna(w, v.right_flank_);
        w.put<uint64_t>(v.inner_flanks_.size());
        for (const auto& d : v.inner_flanks_) write_dna(w, d);
        w.str(v.chromosome_);
        w.put<uint64_t>(v.start_position_);
        w.put<uint64_t>(v.allele_sequences_.size());
        for (const auto& list : v.allele_sequences_) {
            w.put<uint64_t>(list.size());
            for (const auto& d : list) write_dna(w, d);
        }
        w.put<uint64_t>(v.allele_combinations_.size());
        for (const auto& c : v.allele_combinations_) write_u16s(w, c);
        w.put<uint64_t>(v.uncovered_alleles_.size());
        for (const auto& c : v.uncovered_alleles_) write_u16s(w, c);
        write_u16s(w, v.paths_);
        w.put<uint8_t>(v.flanks_added_ ? 1 : 0);
    }
    w.put<uint64_t>(variant_ids_.size());
    for (const auto& ids : variant_ids_) {
        w.put<uint64_t>(ids.size());
        for (const auto& s : ids) w.str(s);
    }
    return w.out;
}

Graph Graph::from_parts(const std::string& chromosome, size_t kmer_size, bool reference_added, const std::vector<Variant>& variants,
                        const std::vector<std::vector<std::string>>& variant_ids) {
    Graph g;
    g.chromosome_ = chromosome; g.kmer_size_ = kmer_size; g.add_reference_ = reference_added;
    size_t n_records = 0;
    for (const Variant& v : variants) { g.variants_.push_back(std::make_shared<Variant>(v)); n_records += v.nr_of_records(); }
    if (variant_ids.size() != n_records) throw std::runtime_error("Graph::from_parts: one row of variant ids per VCF record");
    g.variant_ids_ = variant_ids;
    return g;
}

Graph Graph::from_parts(const std::string& chromosome, size_t kmer_size, bool reference_added, const std::vector<Variant>& variants,
                        const std::vector<std::vector<std::string>>& variant_ids, const std::string& reference_bases) {
    Graph g = from_parts(chromosome, kmer_size, reference_added, variants, variant_ids);
    g.fasta_.emplace_back(chromosome, std::make_shared<DnaSequence>(reference_bases));
    return g;
}

const Variant& Graph::get_variant(size_t index) const {
    if (index >= variants_.size()) throw std::runtime_error("Graph::get_variant: index out of bounds.");
    if (!variants_[index]) throw std::runtime_error("Graph::get_variant: variant was previously destroyed by delete_variant function.");
    return *variants_[index];
}

std::string Graph::reference(const std::string& name) const {
    for (const auto& entry : fasta_)
        if (entry.first == name && entry.second) return entry.second->to_string();
    throw std::runtime_error("Graph::reference: no sequence named " + name);
}

// ------------------------------------------------------------------ VCF text
std::vector<std::string> Graph::genotypes_header(const std::string& sample, const std::string& date) {
    std::string d = date;
    if (d.empty()) {
        const std::time_t t = std::time(nullptr);
        char buf[16];
        std::strftime(buf, sizeof(buf), "%Y%m%d", std::localtime(&t));
        d = buf;
    }
    return {
        "##fileformat=VCFv4.2",
        "##fileDate=" + d,
        "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">",
        "##INFO=<ID=UK,Number=1,Type=Integer,Description=\"Total number of unique kmers.\">",
        "##INFO=<ID=AK,Number=R,Type=Integer,Description=\"Number of unique kmers per allele. Will be -1 for alleles not covered by any input haplotype path\">",
        "##INFO=<ID=MA,Number=1,Type=Integer,Description=\"Number of alleles missing in panel haplotypes.\">",
        "##INFO=<ID=ID,Number=A,Type=String,Description=\"Variant IDs.\">",
        "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">",
        "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality: phred scaled probability that the genotype is wrong.\">",
        "##FORMAT=<ID=GL,Number=G,Type=Float,Description=\"Comma-separated log10-scaled genotype likelihoods for absent, heterozygous, homozygous.\">",
        "##FORMAT=<ID=KC,Number=1,Type=Float,Description=\"Local kmer coverage.\">",
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample,
    };
}

std::vector<std::string> Graph::genotypes_records(const std::vector<GenotypingResult>& genotyping_result, bool ignore_imputed) const {
    return sample_records(genotyping_result, ignore_imputed, false);
}

std::vector<std::string> Graph::phasing_records(const std::vector<GenotypingResult>& genotyping_result, bool ignore_imputed) const {
    return sample_records(genotyping_result, ignore_imputed, true);
}

// the haplotype pair of a record as `a|b` (reference src/graph.cpp:388-407): an allele of undefined sequence is written as
// `.`, the others as their index among the defined alleles
static std::string phased_field(const VcfSite& site, const std::vector<unsigned short>& defined_in, bool ignore_imputed) {
    std::ostringstream out;
    if (ignore_imputed && site.likelihoods.nr_unique_kmers() == 0) out << "./.";
    else {
        std::vector<unsigned short> defined = defined_in;
        const std::pair<unsigned short, unsigned short> in_record = site.likelihoods.get_haplotype();
        const std::pair<unsigned short, unsigned short> among_defined =
            defined.size() < site.alleles.size() ? site.likelihoods.get_specific_likelihoods(defined).get_haplotype() : in_record;
        if (site.undefined.at(in_record.first)) out << ".|"; else out << (unsigned int)among_defined.first << "|";
        if (site.undefined.at(in_record.second)) out << "."; else out << (unsigned int)among_defined.second;
    }
    out << ":" << site.likelihoods.coverage();
    return out.str();
}

std::vector<std::string> Graph::sample_records(const std::vector<GenotypingResult>& genotyping_result, bool ignore_imputed, bool phasing,
                                               const std::vector<SampledPanel>* sampled_paths) const {
    const char* who = sampled_paths ? "Graph::write_sampled_panel" : phasing ? "Graph::write_phasing_of" : "Graph::write_genotypes_of";
    if (variants_deleted_) throw std::runtime_error(std::string(who) + ": variants have been deleted by delete_variant funtion. Re-build object.");
    if ((sampled_paths ? sampled_paths->size() : genotyping_result.size()) != size())
        throw std::runtime_error(std::string(who) + ": number of variants and number of computed " + (phasing || sampled_paths ? "phasings" : "genotypes") + " differ.");
    std::vector<std::string> lines;
    size_t record_index = 0;   // over single records: the row of variant_ids
    for (size_t i = 0; i < size(); ++i) {
        for (const VcfSite& site : sampled_paths ? get_variant(i).records(nullptr, &(*sampled_paths)[i]) : get_variant(i).records(&genotyping_result[i])) {
            const size_t n_all = site.alleles.size();
            if (n_all < 2) throw std::runtime_error(std::string(who) + ": less than 2 alleles given for variant at position " + std::to_string(site.start));
            // ALT = the defined alternative alleles; genotypes over undefined alleles are dropped below
            std::vector<unsigned short> defined = {0};
            std::vector<std::string> alts;
            for (size_t a = 1; a < n_all; ++a)
                if (!site.undefined[a]) { defined.push_back((unsigned short)a); alts.push_back(site.alleles[a]); }
            // allele frequencies over the panel paths (without the reference path when it was added as one)
            std::vector<float> freq(n_all, 0.0f);
            for (unsigned short a : site.paths) freq.at(a) += 1.0f;
            unsigned int n_paths = (unsigned int)site.paths.size();
            if (add_reference_) { n_paths -= 1; freq[0] -= 1.0f; }
            std::ostringstream line;
            line << site.chromosome << '\t' << (site.start + 1) << "\t.\t" << site.alleles[0] << '\t';
            for (size_t k = 0; k < alts.size(); ++k) line << (k ? "," : "") << alts[k];
            line << "\t.\tPASS\tAF=";
            for (size_t k = 1; k < defined.size(); ++k) line << (k > 1 ? "," : "") << std::setprecision(6) << freq[defined[k]] / n_paths;
            line << ";UK=" << (sampled_paths ? (*sampled_paths)[i].unique_kmers : (size_t)site.likelihoods.nr_unique_kmers()) << ";MA=" << (n_all - defined.size());
            const std::vector<std::string>& ids = variant_ids_.at(record_index);
            if (!ids.empty()) {
                // the ids are kept in the lexicographic order of their ALT alleles: back into ALT order
                if (ids.size() != alts.size()) throw std::runtime_error(std::string(who) + ": number of variant ids and of ALT alleles differ");
                std::vector<size_t> by_sequence(alts.size());
                std::iota(by_sequence.begin(), by_sequence.end(), (size_t)0);
                std::sort(by_sequence.begin(), by_sequence.end(), [&](size_t x, size_t y) { return alts[x] < alts[y]; });
                std::vector<const std::string*> in_alt_order(alts.size(), nullptr);
                for (size_t rank = 0; rank < by_sequence.size(); ++rank) in_alt_order[by_sequence[rank]] = &ids[rank];
                line << ";ID=";
                for (size_t k = 0; k < in_alt_order.size(); ++k) line << (k ? "," : "") << *in_alt_order[k];
            }
            if (sampled_paths) {
                // every sampled haplotype's allele as its index among the defined ones
                std::vector<int> among_defined(n_all, -1);
                for (size_t k = 0; k < defined.size(); ++k) among_defined[defined[k]] = (int)k;
                line << "\tGT";
                for (const unsigned short a : site.sampled) {
                    if (among_defined.at(a) < 0) line << "\t."; else line << '\t' << among_defined[a];
                }
            }
            else if (phasing) line << "\tGT:KC\t" << phased_field(site, defined, ignore_imputed);
            else line << "\tGT:GQ:GL:KC\t" << genotype_field(site.likelihoods, defined, n_all, ignore_imputed);
            lines.push_back(line.str());
            record_index += 1;
        }
    }
    return lines;
}

void Graph::write_genotypes(const std::string& filename, const std::vector<GenotypingResult>& genotyping_result, bool write_header,
                            const std::string& sample, bool ignore_imputed) const {
    const std::vector<std::string> records = genotypes_records(genotyping_result, ignore_imputed);   // (throws before the file is touched)
    std::ofstream out(filename, write_header ? std::ios::out : std::ios::app);
    if (!out.is_open()) throw std::runtime_error("Graph::write_genotypes_of: genotyping output file cannot be opened. Note that the filename must not contain non-existing directories.");
    if (write_header)
        for (const std::string& h : genotypes_header(sample)) out << h << '\n';
    for (const std::string& l : records) out << l << '\n';
}

std::vector<std::string> Graph::sampled_panel_header(size_t nr_paths, const std::string& date) {
    // the genotyping header without AK and the per-sample FORMAT lines other than GT; one column per sampled haplotype
    std::string columns;
    for (size_t i = 0; i < nr_paths; ++i) columns += (i ? "\tsampledHT" : "sampledHT") + std::to_string(i);
    std::vector<std::string> lines;
    for (const std::string& l : genotypes_header(columns, date)) {
        if (l.rfind("##INFO=<ID=AK", 0) == 0) continue;
        if (l.rfind("##FORMAT=", 0) == 0 && l.rfind("##FORMAT=<ID=GT,", 0) != 0) continue;
        lines.push_back(l);
    }
    return lines;
}

std::vector<std::string> Graph::sampled_panel_records(const std::vector<SampledPanel>& sampled_paths) const {
    return sample_records({}, false, false, &sampled_paths);
}

void Graph::write_sampled_panel(const std::string& filename, const std::vector<SampledPanel>& sampled_paths, bool write_header) const {
    const std::vector<std::string> records = sampled_panel_records(sampled_paths);
    std::ofstream out(filename, write_header ? std::ios::out : std::ios::app);
    if (!out.is_open()) throw std::runtime_error("Graph::write_sampled_panel: panel output file cannot be opened. Note that the filename must not contain non-existing directories.");
    if (write_header)
        for (const std::string& h : sampled_panel_header(sampled_paths.empty() ? 0 : sampled_paths[0].path_to_allele.size())) out << h << '\n';
    for (const std::string& l : records) out << l << '\n';
}

std::vector<std::string> Graph::phasing_header(const std::string& sample, const std::string& date) {
    // the genotyping header without the GQ / GL lines (reference src/graph.cpp:296-308; its AK description ends with a full stop here)
    std::vector<std::string> lines;
    for (const std::string& l : genotypes_header(sample, date)) {
        if (l.rfind("##FORMAT=<ID=GQ", 0) == 0 || l.rfind("##FORMAT=<ID=GL", 0) == 0) continue;
        if (l.rfind("##INFO=<ID=AK", 0) == 0) { lines.push_back(l.substr(0, l.size() - 2) + ".\">"); continue; }
        lines.push_back(l);
    }
    return lines;
}

void Graph::write_phasing(const std::string& filename, const std::vector<GenotypingResult>& genotyping_result, bool write_header,
                          const std::string& sample, bool ignore_imputed) const {
    const std::vector<std::string> records = phasing_records(genotyping_result, ignore_imputed);
    std::ofstream out(filename, write_header ? std::ios::out : std::ios::app);
    if (!out.is_open()) throw std::runtime_error("Graph::write_phasing_of: phasing output file cannot be opened. Note that the filename must not contain non-existing directories.");
    if (write_header)
        for (const std::string& h : phasing_header(sample)) out << h << '\n';
    for (const std::string& l : records) out << l << '\n';
}

}  // namespace pangenie
