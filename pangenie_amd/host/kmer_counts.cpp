// kmer_counts.cpp — see kmer_counts.hpp.
#include "kmer_counts.hpp"

#include <zlib.h>

#include <fstream>
#include <sstream>
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
void split(std::vector<std::string>& result, const std::string& line, char sep) {  // reference src/kmerparser.cpp:8-14
    std::string token;
    std::istringstream iss(line);
    while (std::getline(iss, token, sep)) result.push_back(token);
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

size_t ExactKmerCounter::computeHistogram(size_t max_count, bool largest_peak) const {
    Histogram histogram(max_count);
    for (const auto& kv : counts_)
        if (kv.second > 0) histogram.add_value((size_t)kv.second);
    histogram.smooth_histogram();
    std::vector<size_t> peak_ids, peak_values;
    histogram.find_peaks(peak_ids, peak_values);
    return compute_kmer_coverage(peak_ids, peak_values, largest_peak);
}

// ------------------------------------------------------------------ Histogram (reference src/histogram.cpp)
Histogram::Histogram(size_t max_value) : histogram_(max_value + 1, 0) {}

Histogram::Histogram(const std::string& filename, size_t max_value) : histogram_(max_value + 1, 0) {
    gzFile file = gzopen(filename.c_str(), "rb");  // (reads plain files as well)
    if (!file) throw std::runtime_error("Histogram: cannot open " + filename);
    char buffer[256];
    while (gzgets(file, buffer, sizeof(buffer)) != nullptr) {
        std::istringstream iss(buffer);
        size_t count, value;
        if (!(iss >> count >> value)) break;  // (the reference's `while (histfile >> count >> value)` stops at the first line that is not two numbers)
        if (count <= max_value) histogram_[count] = value;
    }
    gzclose(file);
}

void Histogram::add_value(size_t value) {
    if (value < histogram_.size()) histogram_[value] += 1;
}

void Histogram::smooth_histogram() {  // in place: entry i - 1 is already smoothed when entry i is formed (src/histogram.cpp:43-47)
    for (size_t i = 1; i + 1 < histogram_.size(); ++i) histogram_[i] = (histogram_[i - 1] + histogram_[i] + histogram_[i + 1]) / 3;
}

void Histogram::find_peaks(std::vector<size_t>& peak_ids, std::vector<size_t>& peak_values) const {
    bool direction = 0;
    size_t prev_val = 0;
    for (size_t i = 0; i < histogram_.size(); ++i) {
        const size_t value = histogram_[i];
        if (prev_val < value) direction = 0;
        else if (prev_val > value) {
            if (direction != 1) { peak_ids.push_back(i - 1); peak_values.push_back(prev_val); }
            direction = 1;
        }
        prev_val = value;
    }
}

size_t compute_kmer_coverage(std::vector<size_t>& peak_ids, std::vector<size_t>& peak_values, bool largest_peak) {
    if (peak_ids.size() == 0) throw std::runtime_error("sequenceutils::computeHistogram: no peak found in kmer-count histogram.");
    if (peak_ids.size() < 2) return peak_ids[0];
    size_t largest, second, largest_id, second_id;
    if (peak_values[0] < peak_values[1]) { largest = peak_values[1]; largest_id = peak_ids[1]; second = peak_values[0]; second_id = peak_ids[0]; }
    else { largest = peak_values[0]; largest_id = peak_ids[0]; second = peak_values[1]; second_id = peak_ids[1]; }
    for (size_t i = 0; i < peak_values.size(); ++i) {
        if (peak_values[i] > largest) { second = largest; second_id = largest_id; largest = peak_values[i]; largest_id = peak_ids[i]; }
        else if ((peak_values[i] > second) && (peak_values[i] != largest)) { second = peak_values[i]; second_id = peak_ids[i]; }
    }
    return largest_peak ? largest_id : second_id;
}

// ------------------------------------------------------------------ kmerparser
void parse_kmer_line(std::string line, std::string& chrom, size_t& start, std::vector<std::string>& kmers,
                     std::vector<std::string>& flanking_kmers, bool& is_header) {
    std::vector<std::string> tokens;
    split(tokens, line, '\t');
    if (tokens.size() != 5) throw std::runtime_error("parse_kmer_line: expected 5 tab-separated fields");  // (the reference asserts)
    if (tokens[0][0] == '#') { is_header = true; return; }
    chrom = tokens[0];
    start = (size_t)atoi(tokens[1].c_str());
    if (tokens[3] != "nan") split(kmers, tokens[3], ',');
    if (tokens[4] != "nan") split(flanking_kmers, tokens[4], ',');
}

unsigned short compute_local_coverage(std::vector<std::string>& kmers, KmerCounter& read_counts, size_t kmer_coverage) {
    size_t total_coverage = 0, total_kmers = 0;
    const size_t min_cov = kmer_coverage / 4, max_cov = kmer_coverage * 4;
    for (auto& kmer : kmers) {
        const size_t read_count = read_counts.getKmerAbundance(kmer);
        if ((read_count < min_cov) || (read_count > max_cov)) continue;  // ignore too extreme counts
        total_coverage += read_count;
        total_kmers += 1;
    }
    if ((total_kmers > 0) && (total_coverage > 0)) return (unsigned short)(total_coverage / total_kmers);
    return (unsigned short)kmer_coverage;
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
            std::vector<std::string> kmers, flanking_kmers;
            bool is_header = false;
            std::string chrom;
            size_t start = 0;
            parse_kmer_line(line, chrom, start, kmers, flanking_kmers, is_header);
            line.clear();
            if (is_header) continue;
            if (chrom != chromosome) throw std::runtime_error("fill_read_kmercounts: line of chromosome " + chrom + " in the table of " + chromosome);
            if (var_index >= objects.size()) throw std::runtime_error("fill_read_kmercounts: more lines than variants");
            UniqueKmers& u = *objects[var_index];
            if (start != u.get_variant_position()) throw std::runtime_error("fill_read_kmercounts: position " + std::to_string(start) + " does not match the index");
            for (size_t i = 0; i < kmers.size(); ++i)
                u.update_readcount(i, (unsigned short)read_kmer_counts.getKmerAbundance(kmers[i]));  // (size_t -> unsigned short as in the reference)
            u.set_coverage(compute_local_coverage(flanking_kmers, read_kmer_counts, kmer_coverage));
            var_index += 1;
        }
    } catch (...) {
        gzclose(file);
        throw;
    }
    gzclose(file);
}

}  // namespace pangenie
