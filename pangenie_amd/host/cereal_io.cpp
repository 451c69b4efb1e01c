// cereal_io.cpp — see cereal_io.hpp.
#include "cereal_io.hpp"
#include "archive_bytes.hpp"

#include <cstring>
#include <fstream>
#include <stdexcept>

namespace pangenie {

namespace {
constexpr uint32_t MSB = archive_bytes::MSB;
const char* const kBi = "BiallelicUniqueKmers";
const char* const kMulti = "MultiallelicUniqueKmers";

using archive_bytes::Reader;
using archive_bytes::Writer;

template <bool BI>
std::shared_ptr<UniqueKmers> read_object(Reader& r) {
    typename UniqueKmersT<BI>::Raw raw;
    raw.variant_pos = (size_t)r.take<uint64_t>();
    raw.local_coverage = r.take<float>();
    (void)r.take<uint64_t>();  // current_index: number of k-mers inserted so far
    const uint64_t nk = r.count(2);
    for (uint64_t i = 0; i < nk; ++i) raw.counts.push_back(r.take<uint16_t>());
    const uint64_t na = r.count(BI ? 6 : 9);
    for (uint64_t i = 0; i < na; ++i) {
        const unsigned short key = BI ? (unsigned short)r.take<uint8_t>() : r.take<uint16_t>();
        typename UniqueKmersT<BI>::RawAllele a;
        a.offset = r.take<uint16_t>();
        a.mask = BI ? (uint32_t)r.take<uint16_t>() : r.take<uint32_t>();
        a.is_undefined = r.take<uint8_t>() != 0;
        raw.alleles[key] = a;
    }
    const uint64_t np = r.count(BI ? 1 : 2);
    for (uint64_t i = 0; i < np; ++i) raw.path_to_allele.push_back(BI ? (unsigned short)r.take<uint8_t>() : r.take<uint16_t>());
    return std::shared_ptr<UniqueKmers>(new UniqueKmersT<BI>(raw));
}

template <bool BI>
void write_object(Writer& w, const UniqueKmersT<BI>& u) {
    const typename UniqueKmersT<BI>::Raw raw = u.raw();
    w.put<uint64_t>(raw.variant_pos);
    w.put<float>(raw.local_coverage);
    w.put<uint64_t>(raw.counts.size());
    w.put<uint64_t>(raw.counts.size());
    for (unsigned short c : raw.counts) w.put<uint16_t>(c);
    w.put<uint64_t>(raw.alleles.size());
    for (auto& kv : raw.alleles) {
        if (BI) w.put<uint8_t>((uint8_t)kv.first); else w.put<uint16_t>(kv.first);
        w.put<uint16_t>(kv.second.offset);
        if (BI) w.put<uint16_t>((uint16_t)kv.second.mask); else w.put<uint32_t>(kv.second.mask);
        w.put<uint8_t>(kv.second.is_undefined ? 1 : 0);
    }
    w.put<uint64_t>(raw.path_to_allele.size());
    for (unsigned short a : raw.path_to_allele) { if (BI) w.put<uint8_t>((uint8_t)a); else w.put<uint16_t>(a); }
}

std::map<std::string, double> read_str_double(Reader& r) {
    std::map<std::string, double> m;
    const uint64_t n = r.count(16);
    for (uint64_t i = 0; i < n; ++i) { std::string k = r.str(); m[k] = r.take<double>(); }
    return m;
}
}  // namespace

UniqueKmersMap parse_unique_kmers_map(const std::vector<unsigned char>& bytes) {
    Reader r{bytes.data(), bytes.size()};
    UniqueKmersMap m;
    m.kmersize = (size_t)r.take<uint64_t>();
    std::map<uint32_t, bool> type_is_bi;                         // polymorphic id -> biallelic?
    std::map<uint32_t, std::shared_ptr<UniqueKmers>> objects;    // shared-pointer id -> object
    const uint64_t nmap = r.count(16);
    for (uint64_t e = 0; e < nmap; ++e) {
        const std::string name = r.str();
        std::vector<std::shared_ptr<UniqueKmers>>& list = m.unique_kmers[name];
        const uint64_t nv = r.count(4);
        for (uint64_t i = 0; i < nv; ++i) {
            uint32_t tid = r.take<uint32_t>();
            if (tid & MSB) {
                const std::string tname = r.str();
                if (tname != kBi && tname != kMulti) throw std::runtime_error("UniqueKmersMap archive: unknown type " + tname);
                type_is_bi[tid & ~MSB] = tname == kBi;
            } else if (tid == 0) {
                list.push_back(nullptr);
                continue;
            }
            tid &= ~MSB;
            if (!type_is_bi.count(tid)) throw std::runtime_error("UniqueKmersMap archive: type id used before its name");
            const uint32_t pid = r.take<uint32_t>();
            if (pid & MSB) {
                std::shared_ptr<UniqueKmers> obj = type_is_bi[tid] ? read_object<true>(r) : read_object<false>(r);
                objects[pid & ~MSB] = obj;
                list.push_back(obj);
            } else {
                if (!objects.count(pid)) throw std::runtime_error("UniqueKmersMap archive: dangling pointer id");
                list.push_back(objects[pid]);
            }
        }
    }
    m.runtimes = read_str_double(r);
    m.sampling_runtimes = read_str_double(r);
    m.add_reference = r.take<uint8_t>() != 0;
    if (r.o != r.n) throw std::runtime_error("UniqueKmersMap archive: trailing bytes");
    return m;
}

UniqueKmersMap load_unique_kmers_map(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse_unique_kmers_map(bytes);
}

std::vector<unsigned char> serialize_unique_kmers_map(const UniqueKmersMap& m) {
    Writer w;
    w.put<uint64_t>(m.kmersize);
    w.put<uint64_t>(m.unique_kmers.size());
    std::map<bool, uint32_t> type_id;
    std::map<const UniqueKmers*, uint32_t> seen;
    uint32_t next_ptr = 1;
    for (auto& kv : m.unique_kmers) {
        w.str(kv.first);
        w.put<uint64_t>(kv.second.size());
        for (auto& sp : kv.second) {
            if (!sp) { w.put<uint32_t>(0); continue; }
            const auto* bi = dynamic_cast<const BiallelicUniqueKmers*>(sp.get());
            const auto* mu = dynamic_cast<const MultiallelicUniqueKmers*>(sp.get());
            if (!bi && !mu) throw std::runtime_error("serialize_unique_kmers_map: unknown UniqueKmers type");
            const bool is_bi = bi != nullptr;
            if (type_id.count(is_bi)) w.put<uint32_t>(type_id[is_bi]);
            else {
                const uint32_t id = (uint32_t)type_id.size() + 1;
                type_id[is_bi] = id;
                w.put<uint32_t>(id | MSB);
                w.str(is_bi ? kBi : kMulti);
            }
            if (seen.count(sp.get())) { w.put<uint32_t>(seen[sp.get()]); continue; }
            seen[sp.get()] = next_ptr;
            w.put<uint32_t>(next_ptr++ | MSB);
            if (bi) write_object<true>(w, *bi); else write_object<false>(w, *mu);
        }
    }
    for (const auto* mp : {&m.runtimes, &m.sampling_runtimes}) {
        w.put<uint64_t>(mp->size());
        for (auto& kv : *mp) { w.str(kv.first); w.put<double>(kv.second); }
    }
    w.put<uint8_t>(m.add_reference ? 1 : 0);
    return w.out;
}

void save_unique_kmers_map(const UniqueKmersMap& m, const std::string& path) {
    const std::vector<unsigned char> b = serialize_unique_kmers_map(m);
    std::ofstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("cannot write " + path);
    f.write((const char*)b.data(), (std::streamsize)b.size());
}

// ------------------------------------------------------------------ Results (`-w`: <out>_genotyping.cereal)
// vector<GenotypingResult> as cereal writes it: what HMM::serialize archives (reference src/hmm.hpp:49-52) and what every
// chromosome of the Results archive holds
static void read_genotyping_results(Reader& r, std::vector<GenotypingResult>& vec) {
    const uint64_t nv = r.count(16);
    vec.resize((size_t)nv);
    for (uint64_t v = 0; v < nv; ++v) {
        GenotypingResult& g = vec[(size_t)v];
        const uint64_t nl = r.count(20);
        for (uint64_t l = 0; l < nl; ++l) {
            const unsigned short a1 = r.take<uint16_t>(), a2 = r.take<uint16_t>();
            if (16 > r.n - r.o) throw std::runtime_error("Results archive: truncated");
            long double lik = 0.0L;
            std::memcpy(&lik, r.p + r.o, 10);  // the 80-bit value; 6 bytes of padding follow
            r.o += 16;
            g.add_to_likelihood(a1, a2, lik);
        }
        g.add_first_haplotype_allele(r.take<uint16_t>());
        g.add_second_haplotype_allele(r.take<uint16_t>());
        g.set_coverage(r.take<uint16_t>());
        g.set_unique_kmers(r.take<uint16_t>());
    }
}
static void write_genotyping_results(Writer& w, const std::vector<GenotypingResult>& vec) {
    static_assert(sizeof(long double) == 16, "x86-64 long double");
    w.put<uint64_t>(vec.size());
    for (const GenotypingResult& g : vec) {
        const auto& m = g.get_stored_likelihoods();
        w.put<uint64_t>(m.size());
        for (const auto& e : m) {
            w.put<uint16_t>(e.first.first);
            w.put<uint16_t>(e.first.second);
            unsigned char b[16] = {0};
            std::memcpy(b, &e.second, 10);
            w.out.insert(w.out.end(), b, b + 16);
        }
        w.put<uint16_t>(g.get_haplotype().first);
        w.put<uint16_t>(g.get_haplotype().second);
        w.put<uint16_t>(g.coverage());
        w.put<uint16_t>(g.nr_unique_kmers());
    }
}

std::vector<unsigned char> HMM::serialize() const {
    Writer w;
    write_genotyping_results(w, genotyping_result_);
    return w.out;
}
HMM HMM::deserialize(const std::vector<unsigned char>& bytes) {
    Reader r{bytes.data(), bytes.size()};
    HMM h;
    read_genotyping_results(r, h.genotyping_result_);
    if (r.o != r.n) throw std::runtime_error("HMM archive: trailing bytes");
    return h;
}

Results parse_results(const std::vector<unsigned char>& bytes) {
    Reader r{bytes.data(), bytes.size()};
    Results out;
    const uint64_t nc = r.count(16);
    for (uint64_t c = 0; c < nc; ++c) {
        const std::string name = r.str();
        read_genotyping_results(r, out.result[name]);
    }
    out.runtimes = read_str_double(r);
    if (r.o != r.n) throw std::runtime_error("Results archive: trailing bytes");
    return out;
}

std::vector<unsigned char> serialize_results(const Results& res) {
    Writer w;
    w.put<uint64_t>(res.result.size());
    for (const auto& kv : res.result) {
        w.str(kv.first);
        write_genotyping_results(w, kv.second);
    }
    w.put<uint64_t>(res.runtimes.size());
    for (const auto& kv : res.runtimes) { w.str(kv.first); w.put<double>(kv.second); }
    return w.out;
}

Results load_results(const std::string& path) {
    std::ifstream is(path, std::ios::binary);
    if (!is) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    return parse_results(bytes);
}

void save_results(const Results& r, const std::string& path) {
    std::ofstream os(path, std::ios::binary);
    if (!os) throw std::runtime_error("cannot open " + path);
    const std::vector<unsigned char> bytes = serialize_results(r);
    os.write((const char*)bytes.data(), (std::streamsize)bytes.size());
}

}  // namespace pangenie
