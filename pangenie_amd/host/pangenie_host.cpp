// pangenie_host.cpp — see pangenie_host.hpp.  Host containers + the HMM adapter over the C ABI.
#include "pangenie_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <limits>
#include <iomanip>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_set>

namespace pangenie {

namespace {
thread_local int g_device = 0;
[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error(msg); }
void check_rc(int rc, const char* err) {
    if (rc != PG_OK) fail(err && err[0] ? std::string(err) : ("pangenie_hmm error " + std::to_string(rc)));
}
}  // namespace

// ------------------------------------------------------------------ KmerPath
// behaviour: reference src/kmerpath.cpp:13-48 (window 32), src/kmerpath16.cpp (window 16)
void KmerPath::set_position(unsigned short index) {
    if (kmers_ == 0) offset_ = index;  // first k-mer fixes the window
    const unsigned lo = offset_, hi = offset_ + window_;
    if (index < lo || index >= hi) fail("KmerPath::KmerPath: index is invalid");
    kmers_ |= (uint32_t(1) << (index - offset_));
}
unsigned int KmerPath::get_position(unsigned short index) const {
    const unsigned lo = offset_, hi = offset_ + window_;
    if (index < lo || index >= hi) return 0;
    return (kmers_ >> (index - offset_)) & 1u;
}
size_t KmerPath::nr_kmers() const { return (size_t)__builtin_popcount(kmers_); }
std::string KmerPath::convert_to_string() const {
    std::string s;
    for (unsigned i = 0; i < offset_ + window_; ++i) s += get_position((unsigned short)i) ? '1' : '0';
    return s;
}
std::ostream& operator<<(std::ostream& os, const KmerPath& p) { return os << p.convert_to_string(); }

// ------------------------------------------------------------------ CopyNumber
// behaviour: reference src/copynumber.cpp:10-58
CopyNumber::CopyNumber() : p_{1.0L, 0.0L, 0.0L} {}
CopyNumber::CopyNumber(long double a, long double b, long double c) : p_{a, b, c} {}
CopyNumber::CopyNumber(long double cn_0, long double cn_1, long double cn_2, long double reg) {
    const long double sum = cn_0 + cn_1 + cn_2 + 3.0L * reg;
    p_[0] = (cn_0 + reg) / sum;
    p_[1] = (cn_1 + reg) / sum;
    p_[2] = 1.0L - p_[0] - p_[1];
}
long double CopyNumber::get_probability_of(int cn) const {
    if (cn < 0 || cn > 2) fail("CopyNumber::get_probability_of: Invalid copy number: " + std::to_string(cn));
    return p_[cn];
}
bool CopyNumber::operator==(const CopyNumber& o) const { return p_[0] == o.p_[0] && p_[1] == o.p_[1] && p_[2] == o.p_[2]; }
bool CopyNumber::operator!=(const CopyNumber& o) const { return !(*this == o); }

// ------------------------------------------------------------------ ProbabilityTable
// the long double table lives in the C ABI object so that host and device see the same entries
ProbabilityTable::ProbabilityTable() : t_(pg_table_create_default()) {}
ProbabilityTable::ProbabilityTable(unsigned short cov_min, unsigned short cov_max, unsigned short count_max, long double reg)
    : t_(pg_table_create(cov_min, cov_max, count_max, reg)) {}
ProbabilityTable& ProbabilityTable::operator=(ProbabilityTable&& o) noexcept {
    if (this != &o) { pg_table_destroy(t_); t_ = o.t_; o.t_ = nullptr; }
    return *this;
}
ProbabilityTable::~ProbabilityTable() { pg_table_destroy(t_); }
CopyNumber ProbabilityTable::get_probability(unsigned short cov, unsigned short count) const {
    long double p[3];
    pg_table_get(t_, cov, count, p);
    return CopyNumber(p[0], p[1], p[2]);
}
void ProbabilityTable::modify_probability(unsigned short cov, unsigned short count, CopyNumber prob) {
    if (pg_table_modify(t_, cov, count, prob.get_probability_of(0), prob.get_probability_of(1), prob.get_probability_of(2)) != PG_OK)
        fail("ProbabilityTable::modify_probability: no precomputed values for these parameters.");
}

// ------------------------------------------------------------------ UniqueKmers
template <bool BI>
void UniqueKmersT<BI>::check_allele(unsigned short a, const char* where) const {
    if (BI && a != 0 && a != 1) fail(std::string(where) + ": provided alleles need to be either 0 or 1 (biallelic).");
}
template <bool BI>
UniqueKmersT<BI>::UniqueKmersT(size_t variant_position, std::vector<unsigned short>& alleles)
    : variant_pos_(variant_position), local_coverage_(0), path_to_allele_(alleles.size()) {
    for (size_t i = 0; i < alleles.size(); ++i) {
        check_allele(alleles[i], BI ? "BiallelicUniqueKmers::BiallelicUniqueKmers" : "MultiallelicUniqueKmers");
        path_to_allele_[i] = alleles[i];
        alleles_[alleles[i]];  // every allele carried by a path gets an entry
    }
}
template <bool BI>
void UniqueKmersT<BI>::insert_kmer(unsigned short readcount, std::vector<unsigned short>& allele_ids) {
    const size_t index = counts_.size();
    for (unsigned short a : allele_ids) check_allele(a, "BiallelicUniqueKmers::insert_kmer");
    counts_.push_back(readcount);
    for (unsigned short a : allele_ids) alleles_[a].kmer_path.set_position((unsigned short)index);  // creates the allele if new
}
template <bool BI>
bool UniqueKmersT<BI>::kmer_on_path(size_t kmer_index, size_t path_index) const {
    if (path_index >= path_to_allele_.size()) fail("UniqueKmers::kmer_on_path: path_index " + std::to_string(path_index) + " does not exist.");
    if (kmer_index >= counts_.size()) fail("UniqueKmers::kmer_on_path: requested kmer index: " + std::to_string(kmer_index) + " does not exist.");
    return alleles_.at(path_to_allele_[path_index]).kmer_path.get_position((unsigned short)kmer_index) > 0;
}
template <bool BI>
bool UniqueKmersT<BI>::kmer_on_allele(size_t kmer_index, size_t allele_id) const {
    return alleles_.at((unsigned short)(BI ? (allele_id != 0) : allele_id)).kmer_path.get_position((unsigned short)kmer_index);
}
template <bool BI>
unsigned short UniqueKmersT<BI>::get_readcount_of(size_t kmer_index) {
    if (kmer_index >= counts_.size()) fail("UniqueKmers::get_readcount_of: requested kmer index: " + std::to_string(kmer_index) + " does not exist.");
    return counts_[kmer_index];
}
template <bool BI>
void UniqueKmersT<BI>::update_readcount(size_t kmer_index, unsigned short new_count) {
    if (kmer_index >= counts_.size()) fail("UniqueKmers::update_readcount: requested kmer index: " + std::to_string(kmer_index) + " does not exist.");
    counts_[kmer_index] = new_count;
}
template <bool BI>
void UniqueKmersT<BI>::get_path_ids(std::vector<unsigned short>& p, std::vector<unsigned short>& a, std::vector<unsigned short>* only_include) {
    if (only_include) {
        for (unsigned short id : *only_include)
            if (id < path_to_allele_.size()) { p.push_back(id); a.push_back(path_to_allele_[id]); }
    } else {
        for (size_t i = 0; i < path_to_allele_.size(); ++i) { p.push_back((unsigned short)i); a.push_back(path_to_allele_[i]); }
    }
}
template <bool BI>
void UniqueKmersT<BI>::get_allele_ids(std::vector<unsigned short>& a) {
    for (auto& kv : alleles_) a.push_back(kv.first);
}
template <bool BI>
void UniqueKmersT<BI>::get_defined_allele_ids(std::vector<unsigned short>& a) {
    for (auto& kv : alleles_)
        if (!kv.second.is_undefined) a.push_back(kv.first);
}
template <bool BI>
std::map<unsigned short, int> UniqueKmersT<BI>::kmers_on_alleles() const {
    std::map<unsigned short, int> r;
    for (auto& kv : alleles_) r[kv.first] = (int)kv.second.kmer_path.nr_kmers();
    return r;
}
template <bool BI>
unsigned short UniqueKmersT<BI>::kmers_on_allele(unsigned short allele_id) const {
    check_allele(allele_id, "BiallelicUniqueKmers::kmers_on_allele");
    return (unsigned short)alleles_.at(allele_id).kmer_path.nr_kmers();
}
template <bool BI>
unsigned short UniqueKmersT<BI>::present_kmers_on_allele(unsigned short allele_id) const {
    check_allele(allele_id, "BiallelicUniqueKmers::present_kmers_on_allele");
    const KmerPath& kp = alleles_.at(allele_id).kmer_path;
    unsigned short n = 0;
    for (size_t i = 0; i < counts_.size(); ++i)
        if (counts_[i] >= 3 && kp.get_position((unsigned short)i)) ++n;  // read-supported = count >= 3
    return n;
}
template <bool BI>
float UniqueKmersT<BI>::fraction_present_kmers_on_allele(unsigned short allele_id) const {
    const unsigned short total = kmers_on_allele(allele_id);
    return total > 0 ? present_kmers_on_allele(allele_id) / (float)total : 1.0f;
}
template <bool BI>
bool UniqueKmersT<BI>::is_undefined_allele(unsigned short allele_id) const {
    check_allele(allele_id, "BiallelicUniqueKmers::is_undefined_allele");
    auto it = alleles_.find(allele_id);
    return it != alleles_.end() && it->second.is_undefined;
}
template <bool BI>
void UniqueKmersT<BI>::set_undefined_allele(unsigned short allele_id) {
    auto it = alleles_.find(allele_id);
    if ((BI && allele_id > 1) || it == alleles_.end())
        fail("UniqueKmers::set_undefined_allele: allele_id " + std::to_string(allele_id) + " does not exist.");
    it->second.is_undefined = true;
}
template <bool BI>
unsigned short UniqueKmersT<BI>::get_allele(unsigned short path_id) const {
    if (path_id >= path_to_allele_.size()) fail("UniqueKmers:get_allele: index out of bounds.");
    return path_to_allele_[path_id];
}
template <bool BI>
std::pair<unsigned short, uint32_t> UniqueKmersT<BI>::kmer_bits(unsigned short allele_id) const {
    const KmerPath& kp = alleles_.at(allele_id).kmer_path;
    return {kp.offset(), kp.mask()};
}
// keep only the given paths; k-mers that sit on none of the surviving alleles are dropped
// (behaviour: reference src/biallelicuniquekmers.cpp:223-260, src/multiallelicuniquekmers.cpp:196-232)
template <bool BI>
void UniqueKmersT<BI>::update_paths(std::vector<unsigned short>& path_ids) {
    std::vector<unsigned short> new_p2a(path_ids.size());
    std::map<unsigned short, AlleleInfo> kept;
    for (size_t i = 0; i < path_ids.size(); ++i) {
        const unsigned short a = get_allele(path_ids[i]);
        new_p2a[i] = a;
        kept[a] = alleles_[a];
    }
    std::map<size_t, std::vector<unsigned short>> kmer_to_alleles;
    std::vector<unsigned short> undefined;
    for (auto& kv : kept) {
        for (size_t k = 0; k < counts_.size(); ++k)
            if (kv.second.kmer_path.get_position((unsigned short)k)) kmer_to_alleles[k].push_back(kv.first);
        if (kv.second.is_undefined) undefined.push_back(kv.first);
    }
    const std::vector<unsigned short> old_counts = counts_;
    path_to_allele_ = new_p2a;
    alleles_.clear();
    for (unsigned short a : new_p2a) alleles_[a];
    counts_.clear();
    for (unsigned short a : undefined) set_undefined_allele(a);
    for (auto& kv : kmer_to_alleles) insert_kmer(old_counts[kv.first], kv.second);
}
template class UniqueKmersT<true>;
template class UniqueKmersT<false>;

// ------------------------------------------------------------------ ColumnIndexer
// behaviour: reference src/columnindexer.cpp:8-78
ColumnIndexer::ColumnIndexer(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, std::vector<unsigned short>* only_paths)
    : unique_kmers_(unique_kmers) {
    for (size_t v = 0; v < unique_kmers->size(); ++v) {
        std::vector<unsigned short> p, a;
        UniqueKmers& uk = *unique_kmers->at(v);
        uk.get_path_ids(p, a, only_paths);
        if (p.empty()) fail("HMM::index_columns: column " + std::to_string(v) + " is not covered by any paths.");
        if (v == 0) paths_ = p;
        bool any_alt = false;
        for (unsigned short al : a)
            if (al != 0 && !uk.is_undefined_allele(al)) any_alt = true;
        if (any_alt) columns_.push_back(v);
    }
}
size_t ColumnIndexer::get_variant_id(size_t c) const {
    if (c >= columns_.size()) fail("ColumnIndexer::get_variant_id: column index does not exist.");
    return columns_[c];
}
unsigned short ColumnIndexer::get_path(unsigned short i) const {
    if (i >= paths_.size()) fail("ColumnIndexer::get_path: path_index does not exist.");
    return paths_[i];
}
unsigned short ColumnIndexer::get_allele(unsigned short path_index, size_t column_index) const {
    const unsigned short path = get_path(path_index);
    if (column_index >= columns_.size()) fail("ColumnIndex::get_allele: column_index does not exist.");
    return unique_kmers_->at(columns_[column_index])->get_allele(path);
}
std::pair<unsigned short, unsigned short> ColumnIndexer::get_path_ids_at(size_t position) const {
    const size_t n = paths_.size();
    if (position >= n * n) fail("ColumnIndexer::get_path_ids_at: index out of bounds.");
    return {(unsigned short)(position / n), (unsigned short)(position % n)};
}

// ------------------------------------------------------------------ GenotypingResult
// behaviour: reference src/genotypingresult.cpp
GenotypingResult::GenotypingResult() : haplotype_1_(0), haplotype_2_(0), local_coverage_(0), unique_kmers_(0) {}
static std::pair<unsigned short, unsigned short> ordered(unsigned short a, unsigned short b) {
    return a < b ? std::make_pair(a, b) : std::make_pair(b, a);
}
void GenotypingResult::add_to_likelihood(unsigned short a1, unsigned short a2, long double value) {
    genotype_to_likelihood_[ordered(a1, a2)] += value;
}
long double GenotypingResult::get_genotype_likelihood(unsigned short a1, unsigned short a2) const {
    auto it = genotype_to_likelihood_.find(ordered(a1, a2));
    return it == genotype_to_likelihood_.end() ? 0.0L : it->second;
}
std::vector<long double> GenotypingResult::get_all_likelihoods(size_t nr_alleles) const {
    std::vector<long double> out(nr_alleles * (nr_alleles + 1) / 2, 0.0L);
    for (auto& kv : genotype_to_likelihood_) {
        const size_t a1 = kv.first.first, a2 = kv.first.second;
        const size_t idx = a2 * (a2 + 1) / 2 + a1;  // VCF genotype ordering
        if (idx >= out.size()) fail("GenotypeResult::get_all_likelihoods: genotype does not match number of alleles.");
        out[idx] = kv.second;
    }
    return out;
}
GenotypingResult GenotypingResult::get_specific_likelihoods(std::vector<unsigned short>& alleles) const {
    GenotypingResult res;
    std::map<unsigned short, unsigned short> index;
    for (unsigned short i = 0; i < alleles.size(); ++i) index[alleles[i]] = i;
    long double sum = 0.0L;
    for (auto& kv : genotype_to_likelihood_) {
        auto i1 = index.find(kv.first.first), i2 = index.find(kv.first.second);
        if (i1 == index.end() || i2 == index.end()) continue;
        if (haplotype_1_ == kv.first.first) res.haplotype_1_ = i1->second;
        if (haplotype_2_ == kv.first.second) res.haplotype_2_ = i2->second;
        res.add_to_likelihood(i1->second, i2->second, kv.second);
        sum += kv.second;
    }
    if (sum > 0) res.divide_likelihoods_by(sum);
    return res;
}
size_t GenotypingResult::get_genotype_quality(unsigned short a1, unsigned short a2) const {
    long double sum = 0.0L;
    for (auto& kv : genotype_to_likelihood_) sum += kv.second;
    if (fabsl(sum - 1) > 0.0000000001L)
        fail("GenotypingResult::get_genotype_quality: genotype quality can only be computed from normalized likelihoods.");
    const long double prob_wrong = 1.0L - get_genotype_likelihood(a1, a2);
    if (prob_wrong > 0.0L) return (size_t)(-10 * log10l(prob_wrong));
    return 10000;
}
void GenotypingResult::divide_likelihoods_by(long double value) {
    for (auto& kv : genotype_to_likelihood_) kv.second = kv.second / value;
}
std::pair<int, int> GenotypingResult::get_likeliest_genotype() const {
    if (genotype_to_likelihood_.empty()) return {-1, -1};
    long double best_value = 0.0L;
    std::pair<unsigned short, unsigned short> best(0, 0);
    for (auto& kv : genotype_to_likelihood_)
        if (kv.second >= best_value) { best_value = kv.second; best = kv.first; }
    for (auto& kv : genotype_to_likelihood_)
        if (kv.first != best && fabsl(kv.second - best_value) < 0.0000000001L) return {-1, -1};  // no unique maximum
    if (best_value > 0.0L) return {best.first, best.second};
    return {-1, -1};
}
void GenotypingResult::combine(GenotypingResult& other) {
    for (auto& kv : other.genotype_to_likelihood_) genotype_to_likelihood_[kv.first] += kv.second;
}
void GenotypingResult::normalize() {
    long double sum = 0.0L;
    for (auto& kv : genotype_to_likelihood_) sum += kv.second;
    if (sum > 0) divide_likelihoods_by(sum);
}
std::ostream& operator<<(std::ostream& os, const GenotypingResult& r) {
    os << "haplotype allele 1: " << r.haplotype_1_ << "\nhaplotype allele 2: " << r.haplotype_2_
       << "\nlocal coverage: " << r.local_coverage_ << "\nnr of unique kmers: " << r.unique_kmers_ << "\n";
    for (auto& kv : r.genotype_to_likelihood_) os << kv.first.first << "/" << kv.first.second << ": " << (double)kv.second << "\n";
    return os;
}

template <bool B>
UniqueKmersT<B>::UniqueKmersT(const Raw& raw)
    : variant_pos_(raw.variant_pos), local_coverage_(raw.local_coverage), counts_(raw.counts), path_to_allele_(raw.path_to_allele) {
    for (auto& kv : raw.alleles) {
        AlleleInfo info;
        info.kmer_path = KmerPath(B ? 16u : 32u, kv.second.offset, kv.second.mask);
        info.is_undefined = kv.second.is_undefined;
        alleles_[kv.first] = info;
    }
}
template <bool B>
typename UniqueKmersT<B>::Raw UniqueKmersT<B>::raw() const {
    Raw r;
    r.variant_pos = variant_pos_; r.local_coverage = local_coverage_; r.counts = counts_; r.path_to_allele = path_to_allele_;
    for (auto& kv : alleles_) {
        RawAllele a;
        a.offset = kv.second.kmer_path.offset(); a.mask = kv.second.kmer_path.mask(); a.is_undefined = kv.second.is_undefined;
        r.alleles[kv.first] = a;
    }
    return r;
}
template UniqueKmersT<true>::UniqueKmersT(const Raw&);
template UniqueKmersT<false>::UniqueKmersT(const Raw&);
template UniqueKmersT<true>::Raw UniqueKmersT<true>::raw() const;
template UniqueKmersT<false>::Raw UniqueKmersT<false>::raw() const;

// ------------------------------------------------------------------ VCF sample column
std::string genotype_field(const GenotypingResult& result, std::vector<unsigned short>& defined_alleles, size_t nr_alleles, bool ignore_imputed) {
    GenotypingResult tmp = result;
    if (tmp.contains_no_likelihoods()) tmp.add_to_likelihood(0, 0, 1.0);   // reference src/graph.cpp:225-227
    const size_t nr_missing = nr_alleles - defined_alleles.size();
    GenotypingResult gl = nr_missing > 0 ? tmp.get_specific_likelihoods(defined_alleles) : tmp;   // :229-233
    nr_alleles = defined_alleles.size();
    std::ostringstream out;
    std::pair<int, int> genotype = gl.get_likeliest_genotype();
    if (ignore_imputed && result.nr_unique_kmers() == 0) genotype = {-1, -1};
    if (genotype.first != -1 && genotype.second != -1)
        out << genotype.first << "/" << genotype.second << ":" << gl.get_genotype_quality((unsigned short)genotype.first, (unsigned short)genotype.second) << ":";
    else
        out << ".:.:";
    std::vector<long double> likelihoods = gl.get_all_likelihoods(nr_alleles);
    if (likelihoods.size() < 3) fail("Graph::write_genotypes_of: too few likelihoods (" + std::to_string(likelihoods.size()) + ") computed");
    out << std::setprecision(4) << std::log10(likelihoods[0]);
    for (size_t j = 1; j < likelihoods.size(); ++j) out << "," << std::setprecision(4) << std::log10(likelihoods[j]);
    out << ":" << result.coverage();
    return out.str();
}

// ------------------------------------------------------------------ flatten
void FlatContig::bind() {
    static const uint16_t z16 = 0; static const uint32_t z32 = 0; static const uint8_t z8 = 0; static const uint64_t z64 = 0;
    batch.n_variants = (uint32_t)variant_pos.size();
    batch.n_paths = (uint32_t)paths.size();
    batch.variant_pos = variant_pos.empty() ? &z64 : variant_pos.data();
    batch.coverage = coverage.empty() ? &z16 : coverage.data();
    batch.kmer_off = kmer_off.data();
    batch.kmer_count = kmer_count.empty() ? &z16 : kmer_count.data();
    batch.allele_off = allele_off.data();
    batch.allele_id = allele_id.empty() ? &z16 : allele_id.data();
    batch.allele_flags = allele_flags.empty() ? &z8 : allele_flags.data();
    batch.allele_kmer_off = allele_kmer_off.empty() ? &z16 : allele_kmer_off.data();
    batch.allele_kmer_mask = allele_kmer_mask.empty() ? &z32 : allele_kmer_mask.data();
    batch.path_allele = path_allele.empty() ? &z16 : path_allele.data();
}

void flatten(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, std::vector<unsigned short>* only_paths, FlatContig& f) {
    const size_t V = unique_kmers->size();
    f = FlatContig();
    f.kmer_off.assign(1, 0);
    f.allele_off.assign(1, 0);
    f.variant_pos.reserve(V); f.coverage.reserve(V); f.kmer_off.reserve(V + 1); f.allele_off.reserve(V + 1);
    std::vector<unsigned short> p, a, ids;
    for (size_t v = 0; v < V; ++v) {
        UniqueKmers& uk = *unique_kmers->at(v);
        p.clear(); a.clear(); ids.clear();
        uk.get_path_ids(p, a, only_paths);
        if (p.empty()) fail("HMM::index_columns: column " + std::to_string(v) + " is not covered by any paths.");
        if (v == 0) {  // the selected paths are those of the first variant (ColumnIndexer)
            f.paths = p;
            f.path_allele.reserve(V * p.size()); f.kmer_count.reserve(V * (uk.size() + 4)); f.allele_id.reserve(V * 2);
        }
        f.variant_pos.push_back(uk.get_variant_position());
        f.coverage.push_back(uk.get_coverage());
        for (size_t k = 0; k < uk.size(); ++k) f.kmer_count.push_back(uk.get_readcount_of(k));
        f.kmer_off.push_back((uint32_t)f.kmer_count.size());
        uk.get_allele_ids(ids);
        for (unsigned short id : ids) {
            f.allele_id.push_back(id);
            f.allele_flags.push_back(uk.is_undefined_allele(id) ? 1 : 0);
            auto bits = uk.kmer_bits(id);
            f.allele_kmer_off.push_back(bits.first);
            f.allele_kmer_mask.push_back(bits.second);
        }
        f.allele_off.push_back((uint32_t)f.allele_id.size());
        if (p == f.paths) f.path_allele.insert(f.path_allele.end(), a.begin(), a.end());  // (get_path_ids already gave the alleles)
        else for (unsigned short path : f.paths) f.path_allele.push_back(uk.get_allele(path));
    }
    f.bind();
}

// ------------------------------------------------------------------ device-backed computers
TransitionProbabilityComputer::TransitionProbabilityComputer(size_t from_variant, size_t to_variant, double recomb_rate,
                                                             unsigned short nr_paths, bool uniform, long double effective_N)
    : uniform_(uniform) {
    double out[3];
    char err[256] = {0};
    check_rc(pg_transition_probs(from_variant, to_variant, recomb_rate, nr_paths, uniform ? 1 : 0, effective_N, g_device, out, err, sizeof(err)), err);
    for (int i = 0; i < 3; ++i) probabilities_[i] = out[i];
}
long double TransitionProbabilityComputer::compute_transition_prob(unsigned short p1, unsigned short p2, unsigned short p3, unsigned short p4) {
    if (uniform_) return 1.0L;
    return probabilities_[(p1 != p3) + (p2 != p4)];
}
long double TransitionProbabilityComputer::compute_transition_prob(unsigned short nr_switches) {
    if (uniform_) return 1.0L;
    return probabilities_[nr_switches];
}

EmissionProbabilityComputer::EmissionProbabilityComputer(std::shared_ptr<UniqueKmers> uniquekmers, ProbabilityTable* probabilities) {
    std::vector<std::shared_ptr<UniqueKmers>> one{uniquekmers};
    FlatContig f;
    flatten(&one, nullptr, f);
    allele_ids_.assign(f.allele_id.begin(), f.allele_id.end());
    const size_t A = allele_ids_.size();
    table_.assign(A * A, 0.0L);
    char err[256] = {0};
    int32_t all_zeros = 0;
    check_rc(pg_emission_table(&f.batch, probabilities->handle(), 0, g_device, table_.data(), &all_zeros, err, sizeof(err)), err);
}
long double EmissionProbabilityComputer::get_emission_probability(unsigned short a1, unsigned short a2) const {
    const size_t A = allele_ids_.size();
    size_t s1 = A, s2 = A;
    for (size_t i = 0; i < A; ++i) { if (allele_ids_[i] == a1) s1 = i; if (allele_ids_[i] == a2) s2 = i; }
    if (s1 == A || s2 == A) fail("EmissionProbabilityComputer: unknown allele");
    return table_[s1 * A + s2];
}

// ------------------------------------------------------------------ HMM
void HMM::set_device(int device) { g_device = device; }
int HMM::device_count() { return pg_hmm_device_count(); }

// behaviour of the constructor: reference src/hmm.cpp:25-63 — everything between ColumnIndexer
// and the optional normalisation runs on the GPU behind pg_hmm_genotype_contig().
HMM::HMM(std::vector<std::shared_ptr<UniqueKmers>>* unique_kmers, ProbabilityTable* probabilities, bool run_genotyping,
         bool run_phasing, double recombrate, bool uniform, long double effective_N, std::vector<unsigned short>* only_paths,
         bool normalize_results)
    : genotyping_result_(unique_kmers->size()) {
    // N of these constructors run at a time on the reference's thread-pool workers (src/commands.cpp:949-978): tell the
    // library a call is coming BEFORE the (host-side, size-dependent) flattening, so that whichever worker reaches the
    // device first knows whom to wait for and all of them end up as chains of ONE device job (pg_hmm_announce).
    struct Announcement {
        int device; bool open;
        explicit Announcement(int d) : device(d), open(true) { pg_hmm_announce(d); }
        ~Announcement() { if (open) pg_hmm_retract(device); }
    } announcement(g_device);
    FlatContig f;
    flatten(unique_kmers, only_paths, f);
    const size_t V = f.variant_pos.size();
    std::vector<uint64_t> geno_off(V + 1, 0);
    pg_hmm_geno_offsets(&f.batch, geno_off.data());
    std::vector<double> lik(geno_off[V] ? geno_off[V] : 1);
    std::vector<int32_t> lik_exp(geno_off[V] ? geno_off[V] : 1);  // one exponent per genotype bin
    std::vector<uint8_t> kept(V ? V : 1), present(f.allele_id.size() ? f.allele_id.size() : 1);
    std::vector<uint16_t> n_kmers(V ? V : 1), cov(V ? V : 1), hap1(V ? V : 1), hap2(V ? V : 1);
    // Viterbi on the device (pg_viterbi.hip): at most 64 selected paths (the reference's callers pass at most 30,
    // src/commands.cpp:939); above that the C ABI refuses (PG_ERR_UNSUPPORTED -> std::runtime_error).  There is no host
    // compute path behind this constructor.
    const bool phase_on_device = run_phasing;
    pg_contig_result r{};
    r.lik = lik.data(); r.lik_exp = lik_exp.data(); r.kept = kept.data(); r.allele_present = present.data();
    r.n_kmers = n_kmers.data(); r.coverage = cov.data();
    r.haplotype_1 = hap1.data(); r.haplotype_2 = hap2.data();
    pg_hmm_params prm{};
    prm.effective_N = effective_N; prm.recombrate = recombrate; prm.uniform = uniform ? 1 : 0;
    prm.run_genotyping = run_genotyping ? 1 : 0; prm.run_phasing = phase_on_device ? 1 : 0;
    char err[512] = {0};
    if (run_genotyping || phase_on_device) {
        prm.reserved = PG_CALL_ANNOUNCED;
        announcement.open = false;  // (the call retires the announcement itself)
        check_rc(pg_hmm_genotype_contig(&f.batch, probabilities->handle(), &prm, announcement.device, &r, err, sizeof(err)), err);
    }
    if (run_genotyping) {
        for (size_t v = 0; v < V; ++v) {
            GenotypingResult& g = genotyping_result_[v];
            if (kept[v]) {
                const uint32_t a0 = f.allele_off[v], A = f.allele_off[v + 1] - a0;
                for (uint32_t a = 0; a < A; ++a) {
                    if (!present[a0 + a]) continue;
                    for (uint32_t b = a; b < A; ++b) {
                        if (!present[a0 + b]) continue;
                        const uint64_t idx = geno_off[v] + (uint64_t)a * A - (uint64_t)a * (a - 1) / 2 + (b - a);
                        // the device returns lik[g] * 2^lik_exp[g]; rebuild the reference's long double
                        g.add_to_likelihood(f.allele_id[a0 + a], f.allele_id[a0 + b], ldexpl((long double)lik[idx], lik_exp[idx]));
                    }
                }
            }
            g.set_unique_kmers(n_kmers[v]);
            g.set_coverage(cov[v]);
        }
        if (normalize_results) normalize();  // reference src/hmm.cpp:36-45
    }
    if (phase_on_device) {  // reference src/hmm.cpp:47-49, :144-172
        for (size_t v = 0; v < V; ++v)
            if (kept[v]) {
                genotyping_result_[v].add_first_haplotype_allele(hap1[v]);
                genotyping_result_[v].add_second_haplotype_allele(hap2[v]);
            }
        // (sic: the reference sets these two by COLUMN index, not by variant — src/hmm.cpp:164-165)
        for (size_t c = 0; c < r.n_columns; ++c) {
            genotyping_result_[c].set_unique_kmers((unsigned short)(f.kmer_off[c + 1] - f.kmer_off[c]));
            genotyping_result_[c].set_coverage(f.coverage[c]);
        }
    }
}
// ------------------------------------------------------------------ results of one chain of a job
// vector<GenotypingResult> of one chain from its packed posteriors (reference src/hmm.cpp:364-368, 106-109); kept columns
// and allele presence by the ColumnIndexer rule on the host (src/columnindexer.cpp:24-31).  `coverage` = the chain's own
// local coverages (a cohort's chains share the index, not these).
static std::vector<GenotypingResult> results_of_chain(const FlatContig& f, const uint16_t* coverage, const std::vector<uint64_t>& goff,
                                                      const double* lik, const int32_t* lexp) {
    const size_t V = f.variant_pos.size(), H = f.paths.size();
    std::vector<GenotypingResult> out(V);
    size_t columns = 0;
    std::vector<uint8_t> kept(V, 0), present(f.allele_id.size(), 0);
    for (size_t v = 0; v < V; ++v) {
        const uint32_t a0 = f.allele_off[v], A = f.allele_off[v + 1] - a0;
        for (size_t p = 0; p < H; ++p) {
            const uint16_t a = f.path_allele[v * H + p];
            for (uint32_t q = 0; q < A; ++q)
                if (f.allele_id[a0 + q] == a) { present[a0 + q] = 1; if (a != 0 && !(f.allele_flags[a0 + q] & 1)) kept[v] = 1; }
        }
        columns += kept[v];
    }
    for (size_t v = 0; v < V; ++v) {
        GenotypingResult& g = out[v];
        if (kept[v]) {
            const uint32_t a0 = f.allele_off[v], A = f.allele_off[v + 1] - a0;
            for (uint32_t a = 0; a < A; ++a) {
                if (!present[a0 + a]) continue;
                for (uint32_t b = a; b < A; ++b) {
                    if (!present[a0 + b]) continue;
                    const uint64_t idx = goff[v] + (uint64_t)a * A - (uint64_t)a * (a - 1) / 2 + (b - a);
                    g.add_to_likelihood(f.allele_id[a0 + a], f.allele_id[a0 + b], ldexpl((long double)lik[idx], lexp[idx]));
                }
            }
        }
        if (columns > 0) {  // reference src/hmm.cpp:94,106-109
            g.set_unique_kmers((unsigned short)(f.kmer_off[v + 1] - f.kmer_off[v]));
            g.set_coverage(coverage[v]);
        }
    }
    return out;
}

// ------------------------------------------------------------------ multi-GPU job loop
std::vector<std::vector<GenotypingResult>> run_contigs_multi_gpu(std::vector<ContigTask>& tasks, ProbabilityTable* probabilities,
                                                                 double recombrate, bool uniform, long double effective_N,
                                                                 const std::vector<int>& devices) {
    if (devices.empty()) fail("run_contigs_multi_gpu: no devices");
    const size_t T = tasks.size(), D = devices.size();
    std::vector<FlatContig> flat(T);
    std::vector<std::vector<uint64_t>> goff(T);
    std::vector<double> weight(T);
    for (size_t t = 0; t < T; ++t) {
        flatten(tasks[t].unique_kmers, tasks[t].only_paths, flat[t]);
        const size_t V = flat[t].variant_pos.size();
        goff[t].assign(V + 1, 0);
        pg_hmm_geno_offsets(&flat[t].batch, goff[t].data());
        weight[t] = (double)V * (double)flat[t].paths.size() * (double)flat[t].paths.size();
    }
    // longest processing time first; ties by index: every host computes the same plan
    std::vector<size_t> order(T);
    for (size_t t = 0; t < T; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
    std::vector<std::vector<size_t>> plan(D);
    std::vector<double> load(D, 0.0);
    for (size_t t : order) {
        size_t best = 0;
        for (size_t d = 1; d < D; ++d) if (load[d] < load[best]) best = d;
        plan[best].push_back(t);
        load[best] += weight[t];
    }
    for (auto& p : plan) std::sort(p.begin(), p.end());

    pg_hmm_params prm{};
    prm.effective_N = effective_N; prm.recombrate = recombrate; prm.uniform = uniform ? 1 : 0; prm.run_genotyping = 1;
    std::vector<pg_job*> jobs(D, nullptr);
    std::vector<std::string> errors(D);
    std::vector<std::thread> threads;
    for (size_t d = 0; d < D; ++d)
        threads.emplace_back([&, d] {
            if (plan[d].empty()) return;
            std::vector<pg_contig_batch> b;
            for (size_t t : plan[d]) b.push_back(flat[t].batch);
            char err[512] = {0};
            int rc = pg_job_new(devices[d], (uint32_t)b.size(), b.data(), probabilities->handle(), &prm, &jobs[d], err, sizeof(err));
            if (rc == PG_OK) rc = pg_job_run(jobs[d], nullptr, err, sizeof(err));
            if (rc != PG_OK) errors[d] = err[0] ? err : "pangenie_hmm error";
        });
    for (auto& th : threads) th.join();
    auto cleanup = [&] { for (pg_job* j : jobs) if (j) pg_job_destroy(j); };
    for (size_t d = 0; d < D; ++d)
        if (!errors[d].empty()) { cleanup(); fail(errors[d]); }

    // ONE exchange: every device's packed posteriors to devices[0], then to the host
    std::vector<uint64_t> n_lik(D, 0);
    uint64_t total = 0;
    for (size_t d = 0; d < D; ++d) { for (size_t t : plan[d]) n_lik[d] += goff[t].back(); total += n_lik[d]; }
    std::vector<double> lik(total ? total : 1);
    std::vector<int32_t> lexp(total ? total : 1);
    char err[512] = {0};
    if (D == 1) {
        uint64_t off = 0;
        for (size_t k = 0; k < plan[0].size(); ++k) {
            pg_contig_result r{};
            r.lik = lik.data() + off; r.lik_exp = lexp.data() + off;
            const int rc = pg_job_fetch(jobs[0], (uint32_t)k, &r, err, sizeof(err));
            if (rc != PG_OK) { cleanup(); check_rc(rc, err); }
            off += goff[plan[0][k]].back();
        }
    } else {
        std::vector<pg_comm*> comms(D, nullptr);
        int rc = pg_comm_init_all((int)D, devices.data(), comms.data(), err, sizeof(err));
        if (rc == PG_OK) rc = pg_hmm_gather_to_host((int)D, comms.data(), jobs.data(), 0, n_lik.data(), lik.data(), lexp.data(), err, sizeof(err));
        for (pg_comm* c : comms) if (c) pg_comm_destroy(c);
        if (rc != PG_OK) { cleanup(); check_rc(rc, err); }
    }
    cleanup();

    // rebuild the GenotypingResults
    std::vector<std::vector<GenotypingResult>> out(T);
    uint64_t base = 0;
    for (size_t d = 0; d < D; ++d)
        for (size_t t : plan[d]) {
            out[t] = results_of_chain(flat[t], flat[t].coverage.data(), goff[t], lik.data() + base, lexp.data() + base);
            base += goff[t].back();
        }
    return out;
}

// ------------------------------------------------------------------ cohort job
SampleCounts SampleCounts::of(const std::map<std::string, std::vector<std::shared_ptr<UniqueKmers>>>& chromosomes) {
    SampleCounts s;
    for (const auto& kv : chromosomes) {
        std::vector<uint16_t>& counts = s.kmer_count[kv.first];
        std::vector<uint16_t>& cov = s.coverage[kv.first];
        for (const std::shared_ptr<UniqueKmers>& u : kv.second) {
            for (size_t i = 0; i < u->size(); ++i) counts.push_back(u->get_readcount_of(i));
            cov.push_back(u->get_coverage());
        }
    }
    return s;
}

std::vector<std::map<std::string, std::vector<GenotypingResult>>> genotype_cohort(
    std::map<std::string, std::vector<std::shared_ptr<UniqueKmers>>>& chromosomes, const std::vector<SampleCounts>& samples,
    ProbabilityTable* probabilities, double recombrate, bool uniform, long double effective_N, int device) {
    const size_t C = chromosomes.size(), S = samples.size();
    std::vector<std::map<std::string, std::vector<GenotypingResult>>> out(S);
    if (C == 0 || S == 0) return out;
    std::vector<std::string> names;
    std::vector<FlatContig> flat(C);
    std::vector<std::vector<uint64_t>> goff(C);
    std::vector<pg_contig_batch> index(C);
    size_t c = 0;
    for (auto& kv : chromosomes) {
        names.push_back(kv.first);
        flatten(&kv.second, nullptr, flat[c]);
        goff[c].assign(flat[c].variant_pos.size() + 1, 0);
        pg_hmm_geno_offsets(&flat[c].batch, goff[c].data());
        index[c] = flat[c].batch;
        c += 1;
    }
    // per sample the two pointer rows of pg_sample_counts; sizes checked against the index
    static const uint16_t none = 0;
    std::vector<std::vector<const uint16_t*>> count_rows(S, std::vector<const uint16_t*>(C)), cov_rows(S, std::vector<const uint16_t*>(C));
    std::vector<pg_sample_counts> rows(S);
    for (size_t s = 0; s < S; ++s) {
        for (c = 0; c < C; ++c) {
            const auto k = samples[s].kmer_count.find(names[c]), v = samples[s].coverage.find(names[c]);
            if (k == samples[s].kmer_count.end() || v == samples[s].coverage.end() || k->second.size() != flat[c].kmer_count.size() ||
                v->second.size() != flat[c].variant_pos.size())
                fail("genotype_cohort: sample " + std::to_string(s) + " does not fit the index on " + names[c]);
            count_rows[s][c] = k->second.empty() ? &none : k->second.data();
            cov_rows[s][c] = v->second.empty() ? &none : v->second.data();
        }
        rows[s].kmer_count = count_rows[s].data();
        rows[s].coverage = cov_rows[s].data();
    }
    pg_hmm_params prm{};
    prm.effective_N = effective_N; prm.recombrate = recombrate; prm.uniform = uniform ? 1 : 0; prm.run_genotyping = 1;
    char err[512] = {0};
    pg_job* job = nullptr;
    int rc = pg_cohort_new(device, (uint32_t)C, index.data(), (uint32_t)S, rows.data(), probabilities->handle(), &prm, &job, err, sizeof(err));
    if (rc == PG_OK) rc = pg_job_run(job, nullptr, err, sizeof(err));
    if (rc != PG_OK) { if (job) pg_job_destroy(job); check_rc(rc, err); }
    for (size_t s = 0; s < S && rc == PG_OK; ++s)
        for (c = 0; c < C && rc == PG_OK; ++c) {
            const uint64_t n = goff[c].back();
            std::vector<double> lik(n ? n : 1);
            std::vector<int32_t> lexp(n ? n : 1);
            pg_contig_result r{};
            r.lik = lik.data(); r.lik_exp = lexp.data();
            rc = pg_job_fetch(job, (uint32_t)(s * C + c), &r, err, sizeof(err));   // chain id = sample * n_contigs + contig
            if (rc == PG_OK) out[s][names[c]] = results_of_chain(flat[c], cov_rows[s][c], goff[c], lik.data(), lexp.data());
        }
    pg_job_destroy(job);
    check_rc(rc, err);
    return out;
}

void HMM::combine_likelihoods(HMM& other) {
    if (genotyping_result_.size() != other.genotyping_result_.size())
        fail("HMM::combine_likelihoods: HMMs to be combined must be of the same size.");
    for (size_t i = 0; i < genotyping_result_.size(); ++i) genotyping_result_[i].combine(other.genotyping_result_[i]);
}
void HMM::normalize() {
    for (auto& g : genotyping_result_) g.normalize();
}

// ------------------------------------------------------------------ haplotype sampling
std::vector<bool> SampledPaths::mask_indexes(size_t column_index, size_t max_index) {
    std::vector<bool> masked(max_index + 1, true);
    for (size_t i = 0; i < sampled_paths.size(); ++i) {
        if (column_index >= sampled_paths[i].size())
            throw std::runtime_error("HaplotypeSampler::SampledPaths::mask_indexes: column_index exceeds number of columns.");
        const size_t index = sampled_paths[i][column_index];
        if (index > max_index) throw std::runtime_error("HaplotypeSampler::SampledPaths::mask_indexes: observed index exceeds max_index.");
        masked[index] = false;
    }
    return masked;
}

bool SampledPaths::recombination(size_t column_index, size_t path_id) {
    if (path_id >= sampled_paths.size()) throw std::runtime_error("HaplotypeSampler::SampledPaths::recombination: path_id does not exist.");
    if (column_index >= sampled_paths[path_id].size())
        throw std::runtime_error("HaplotypeSampler::SampledPaths::recombination: column_id does not exist.");
    if (column_index > 0) return sampled_paths[path_id][column_index - 1] != sampled_paths[path_id][column_index];
    return false;
}

SamplingEmissions::SamplingEmissions(std::shared_ptr<UniqueKmers> uk) : default_penalty(25) {
    std::vector<std::shared_ptr<UniqueKmers>> one{uk};
    FlatContig f;
    std::vector<unsigned short> all;  // every path: the costs do not depend on the selection, but flatten needs one
    flatten(&one, nullptr, f);
    std::vector<uint16_t> cost(f.allele_id.size());
    if (pg_sampler_emission_costs(&f.batch, cost.data()) != PG_OK) throw std::runtime_error("SamplingEmissions: pg_sampler_emission_costs failed");
    unsigned short max_allele = 0;
    for (uint16_t a : f.allele_id) max_allele = std::max<unsigned short>(max_allele, a);
    allele_penalties.assign((size_t)max_allele + 1, 0);
    for (size_t s = 0; s < f.allele_id.size(); ++s) allele_penalties[f.allele_id[s]] = cost[s];
}

unsigned int SamplingEmissions::get_emission_cost(unsigned short allele_id) const { return allele_penalties.at(allele_id); }

void SamplingEmissions::penalize(unsigned short allele_id, unsigned short penalty) {
    allele_penalties.at(allele_id) += penalty;
    if (allele_penalties[allele_id] > default_penalty) allele_penalties[allele_id] = (unsigned short)default_penalty;
}

SamplingTransitions::SamplingTransitions(size_t from_variant, size_t to_variant, double recomb_rate, unsigned short nr_paths, long double effective_N)
    : cost(pg_sampler_transition_cost(from_variant, to_variant, recomb_rate, nr_paths, effective_N)) {}

unsigned int SamplingTransitions::compute_transition_cost(bool recombination) { return recombination ? cost : 0u; }

namespace {
thread_local int g_sampler_device = 0;
}
void HaplotypeSampler::set_device(int device) { g_sampler_device = device; }

HaplotypeSampler::HaplotypeSampler(std::vector<std::shared_ptr<UniqueKmers>>* uks, size_t size, double recombrate, long double effective_N,
                                   std::vector<unsigned int>* best_scores, bool add_reference, std::string path_output,
                                   std::string chromosome, unsigned short allele_penalty, double* time)
    : unique_kmers(uks) {
    const auto t0 = std::chrono::steady_clock::now();
    if (size < 1) return;
    const size_t V = uks->size();
    if (V > 0) {
        FlatContig f;
        flatten(uks, nullptr, f);  // ALL paths of the panel
        std::vector<uint32_t> sampled(size * V), best(size);
        char err[512] = {0};
        const int rc = pg_sampler_run(&f.batch, (uint32_t)size, recombrate, effective_N, allele_penalty, g_sampler_device, sampled.data(),
                                      best.data(), err, sizeof(err));
        if (rc != PG_OK) throw std::runtime_error(std::string("HaplotypeSampler: ") + err);
        for (size_t i = 0; i < size; ++i) {
            sampled_paths.sampled_paths.emplace_back(sampled.begin() + i * V, sampled.begin() + (i + 1) * V);
            if (best_scores != nullptr) best_scores->push_back(best[i]);
        }
    } else {
        for (size_t i = 0; i < size; ++i) sampled_paths.sampled_paths.emplace_back();
    }
    if (add_reference) sampled_paths.sampled_paths.push_back(std::vector<size_t>(V, 0));
    if (path_output != "") {  // reference src/haplotypesampler.cpp:46-66
        std::ofstream out(path_output);
        out << "#chromosome\tposition";
        for (size_t p = 0; p < sampled_paths.sampled_paths.size(); ++p) out << "\tHaplotypeID_path" << p << "\tRecombination_path" << p;
        out << std::endl;
        for (size_t c = 0; c < V; ++c) {
            out << chromosome << "\t" << uks->at(c)->get_variant_position();
            for (size_t p = 0; p < sampled_paths.sampled_paths.size(); ++p)
                out << "\t" << sampled_paths.sampled_paths[p][c] << "\t" << sampled_paths.recombination(c, p);
            out << std::endl;
        }
    }
    update_unique_kmers();
    if (time != nullptr) *time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HaplotypeSampler::get_column_minima(std::vector<unsigned int>& column, std::vector<bool>& mask, size_t& first_id, size_t& second_id,
                                         unsigned int& first_val, unsigned int& second_val) const {
    std::vector<uint8_t> m(mask.begin(), mask.end());
    uint32_t out4[4];
    char err[256] = {0};
    if (pg_sampler_column_minima(column.data(), m.data(), (uint32_t)column.size(), g_sampler_device, out4, err, sizeof(err)) != PG_OK)
        throw std::runtime_error(std::string("HaplotypeSampler::get_column_minima: ") + err);
    // absent entries: numeric_limits<unsigned int>::max() widened to size_t, as in the reference
    first_id = out4[0]; second_id = out4[1]; first_val = out4[2]; second_val = out4[3];
}

void HaplotypeSampler::update_unique_kmers() {
    const size_t nr_paths = sampled_paths.sampled_paths.size();
    for (size_t i = 0; i < unique_kmers->size(); ++i) {
        std::vector<unsigned short> p(nr_paths);
        for (size_t j = 0; j < nr_paths; ++j) p[j] = (unsigned short)sampled_paths.sampled_paths[j][i];
        (*unique_kmers)[i]->update_paths(p);
    }
}

}  // namespace pangenie
