// kmer_counts.cpp — see kmer_counts.hpp.
#include "kmer_counts.hpp"

#include <zlib.h>

#include <fstream>
#include <charconv>
#include <string_view>
#include <stdexcept>

namespace pangenie {

namespace {
inline int base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}
}  // namespace

// ------------------------------------------------------------------ ExactKmerCounter
ExactKmerCounter::ExactKmerCounter(const std::string& readfile, size_t kmer_size) : k_(kmer_size) {
    if (k_ == 0 || k_ > 32) throw std::runtime_error("ExactKmerCounter: k-mer size must be 1..32");
    std::ifstream in(readfile);
    if (!in.good()) throw std::runtime_error("ExactKmerCounter: cannot open " + readfile);
    // FASTA (">" header, sequence on one or more lines) or FASTQ ("@" header, sequence, "+", qualities)
    std::string line, seq;
    enum { NONE, FASTA, FQ_SEQ, FQ_PLUS, FQ_QUAL } state = NONE;
    size_t qual_left = 0;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (state == FQ_QUAL) {
            qual_left = line.size() >= qual_left ? 0 : qual_left - line.size();
            if (qual_left == 0) state = NONE;
            continue;
        }
        if (state == FQ_SEQ && !line.empty() && line[0] == '+') {
            add_sequence(seq);
            qual_left = seq.size();
            seq.clear();
            state = qual_left ? FQ_QUAL : NONE;
            continue;
        }
        if (!line.empty() && line[0] == '>' && state != FQ_SEQ) {
            if (state == FASTA) add_sequence(seq);
            seq.clear();
            state = FASTA;
            continue;
        }
        if (!line.empty() && line[0] == '@' && (state == NONE || state == FASTA)) {
            if (state == FASTA) add_sequence(seq);
            seq.clear();
            state = FQ_SEQ;
            continue;
        }
        if (state == FASTA || state == FQ_SEQ) seq += line;
    }
    if (state == FASTA || state == FQ_SEQ) add_sequence(seq);
}

void ExactKmerCounter::add_sequence(const std::string& seq) {
    // rolling 2-bit codes of the window and of its reverse complement; a letter outside {A,C,G,T} restarts the window
    const uint64_t mask = k_ == 32 ? ~0ull : ((1ull << (2 * k_)) - 1ull);
    uint64_t fwd = 0, rev = 0;
    size_t filled = 0;
    for (char c : seq) {
        const int b = base_code(c);
        if (b < 0) { filled = 0; fwd = rev = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)b) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
        if (++filled >= k_) counts_[fwd < rev ? fwd : rev] += 1;  // (2-bit codes order like the letters: A < C < G < T)
    }
}

bool ExactKmerCounter::encode_canonical(const char* s, uint64_t& code) const {
    uint64_t fwd = 0, rev = 0;
    for (size_t i = 0; i < k_; ++i) {
        const int b = base_code(s[i]);
        if (b < 0) return false;
        fwd = (fwd << 2) | (uint64_t)b;
        rev = (rev >> 2) | ((uint64_t)(3 - b) << (2 * (k_ - 1)));
    }
    code = fwd < rev ? fwd : rev;
    return true;
}

size_t ExactKmerCounter::getKmerAbundance(std::string kmer) {
    if (kmer.size() != k_) throw std::runtime_error("ExactKmerCounter::getKmerAbundance: k-mer of length " + std::to_string(kmer.size()) + ", counter holds " + std::to_string(k_) + "-mers");
    uint64_t code;
    if (!encode_canonical(kmer.data(), code)) return 0;
    const auto it = counts_.find(code);
    return it == counts_.end() ? 0 : (size_t)it->second;
}

// ------------------------------------------------------------------ the k-mer table (behaviour: src/kmerparser.cpp)
// A row of `<prefix>_<chromosome>_kmers.tsv` is five tab-separated columns: chromosome, start, (a column this step
// does not use), the variant's unique k-mers, the flanking k-mers — the two lists comma-separated, "nan" when empty;
// rows whose first column starts with '#' are headers.  KmerRow scans a row in place: columns and list items are
// views into the line, nothing is copied until a caller asks for strings.
namespace {
struct KmerRow {
    std::string_view column[5];
    bool header = false;
    explicit KmerRow(std::string_view line) {
        size_t n = 0, at = 0;
        while (true) {
            const size_t tab = line.find('\t', at);
            if (n == 5) { n = 6; break; }  // a sixth column
            column[n++] = line.substr(at, tab == std::string_view::npos ? std::string_view::npos : tab - at);
            if (tab == std::string_view::npos) break;
            at = tab + 1;
            if (at == line.size()) break;  // (a trailing tab opens no further column: std::getline semantics of the reference's tokenizer)
        }
        if (n != 5) throw std::runtime_error("parse_kmer_line: expected 5 tab-separated fields");
        header = !column[0].empty() && column[0].front() == '#';
    }
    size_t start() const {  // leading decimal digits of column 1 (0 when there are none)
        size_t value = 0;
        std::from_chars(column[1].data(), column[1].data() + column[1].size(), value);
        return value;
    }
    template <class F>
    static void each_item(std::string_view list, F&& f) {  // comma-separated items of a list column; "nan" = no items
        if (list == "nan") return;
        while (!list.empty()) {
            const size_t comma = list.find(',');
            f(list.substr(0, comma));
            if (comma == std::string_view::npos) break;
            list.remove_prefix(comma + 1);
        }
    }
};
/** mean (integer division) of the counts inside [expected / 4, expected * 4]; `expected` itself when no count lies in
 *  the window or the window holds only zeros */
unsigned short windowed_mean(const std::vector<size_t>& counts, size_t expected) {
    const size_t lowest = expected / 4, highest = expected * 4;
    size_t sum = 0, used = 0;
    for (const size_t c : counts)
        if (c >= lowest && c <= highest) { sum += c; ++used; }
    return (unsigned short)((used && sum) ? sum / used : expected);
}
}  // namespace

void parse_kmer_line(std::string line, std::string& chrom, size_t& start, std::vector<std::string>& kmers,
                     std::vector<std::string>& flanking_kmers, bool& is_header) {
    const KmerRow row(line);
    if (row.header) { is_header = true; return; }
    chrom.assign(row.column[0]);
    start = row.start();
    KmerRow::each_item(row.column[3], [&](std::string_view k) { kmers.emplace_back(k); });
    KmerRow::each_item(row.column[4], [&](std::string_view k) { flanking_kmers.emplace_back(k); });
}

unsigned short compute_local_coverage(std::vector<std::string>& kmers, KmerCounter& read_counts, size_t kmer_coverage) {
    std::vector<size_t> counts;
    counts.reserve(kmers.size());
    for (const std::string& k : kmers) counts.push_back(read_counts.getKmerAbundance(k));
    return windowed_mean(counts, kmer_coverage);
}

// ------------------------------------------------------------------ fill_read_kmercounts
void fill_read_kmercounts(const std::string& chromosome, UniqueKmersMap* unique_kmers_map, KmerCounter& read_kmer_counts,
                          const std::string& kmers_tsv_gz, size_t kmer_coverage) {
    gzFile file = gzopen(kmers_tsv_gz.c_str(), "rb");
    if (!file) throw std::runtime_error("fill_read_kmercounts: kmer file cannot be opened.");
    auto& objects = unique_kmers_map->unique_kmers[chromosome];
    const int buffer_size = 1024;
    char buffer[buffer_size];
    std::string line;
    size_t var_index = 0;
    try {
        while (gzgets(file, buffer, buffer_size) != nullptr) {
            line += buffer;
            if (line.empty() || line.back() != '\n') continue;
            line.pop_back();
            const KmerRow row(line);
            if (!row.header) {
                if (row.column[0] != chromosome) throw std::runtime_error("fill_read_kmercounts: line of chromosome " + std::string(row.column[0]) + " in the table of " + chromosome);
                if (var_index >= objects.size()) throw std::runtime_error("fill_read_kmercounts: more lines than variants");
                UniqueKmers& u = *objects[var_index];
                if (row.start() != u.get_variant_position()) throw std::runtime_error("fill_read_kmercounts: position " + std::to_string(row.start()) + " does not match the index");
                size_t i = 0;
                KmerRow::each_item(row.column[3], [&](std::string_view k) {
                    u.update_readcount(i++, (unsigned short)read_kmer_counts.getKmerAbundance(std::string(k)));  // (size_t -> unsigned short as in the reference)
                });
                std::vector<size_t> flank_counts;
                KmerRow::each_item(row.column[4], [&](std::string_view k) { flank_counts.push_back(read_kmer_counts.getKmerAbundance(std::string(k))); });
                u.set_coverage(windowed_mean(flank_counts, kmer_coverage));
                var_index += 1;
            }
            line.clear();
        }
    } catch (...) {
        gzclose(file);
        throw;
    }
    gzclose(file);
}

}  // namespace pangenie
