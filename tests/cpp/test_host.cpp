// test_host.cpp — C++ tests of the host interface (pangenie_amd/host/pangenie_host.hpp).
//
//   test_host cpu   host-only classes: KmerPath, UniqueKmers, CopyNumber, ProbabilityTable,
//                   GenotypingResult, ColumnIndexer, flatten  (no device work)
//   test_host gpu   HMM / EmissionProbabilityComputer / TransitionProbabilityComputer through
//                   the C ABI on the GPU, against the reference's known answers
//
// The scenarios and expected numbers are those of the reference's unit tests
// (tests/HMMTest.cpp, EmissionProbabilityComputerTest.cpp, TransitionProbabilityComputerTest.cpp,
// UniqueKmersTest.cpp, GenotypingResultTest.cpp, CopyNumberTest.cpp, ColumnIndexerTest.cpp,
// KmerPathTest.cpp); tolerance 1e-7 absolute as in reference tests/utils.cpp:9-11.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iomanip>
#include <fstream>
#include <sstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <zlib.h>
#include <unordered_set>
#include <vector>

#include "../../pangenie_amd/host/cereal_io.hpp"
#include "../../pangenie_amd/host/graph_io.hpp"
#include "../../pangenie_amd/host/index_builder.hpp"
#include "../../pangenie_amd/host/kmer_counts.hpp"
#include "../../pangenie_amd/host/pangenie_host.hpp"

using namespace pangenie;
using std::shared_ptr;
using std::vector;
typedef vector<unsigned short> us;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                                       \
    do {                                                                                  \
        ++g_checks;                                                                       \
        if (!(cond)) { ++g_failed; std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)
#define CHECK_THROWS(expr)                                                                \
    do {                                                                                  \
        ++g_checks;                                                                       \
        bool threw_ = false;                                                              \
        try { (void)(expr); } catch (const std::exception&) { threw_ = true; }            \
        if (!threw_) { ++g_failed; std::printf("  FAILED %s:%d: no throw: %s\n", __FILE__, __LINE__, #expr); } \
    } while (0)
static bool close(long double a, long double b) { return std::fabs((double)(a - b)) < 1e-7; }
static bool close_all(const vector<double>& a, const vector<double>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (!close(a[i], b[i])) { std::printf("    [%zu] got %.12g expected %.12g\n", i, a[i], b[i]); return false; }
    return true;
}
static void run(const char* name, const std::function<void()>& f) {
    const int before = g_failed;
    try { f(); } catch (const std::exception& e) { ++g_failed; std::printf("  EXCEPTION in %s: %s\n", name, e.what()); }
    std::printf("%s %s\n", g_failed == before ? "ok  " : "FAIL", name);
}
static shared_ptr<UniqueKmers> bi(size_t pos, us paths) { return shared_ptr<UniqueKmers>(new BiallelicUniqueKmers(pos, paths)); }
static shared_ptr<UniqueKmers> multi(size_t pos, us paths) { return shared_ptr<UniqueKmers>(new MultiallelicUniqueKmers(pos, paths)); }
static void kmer(shared_ptr<UniqueKmers>& u, unsigned short count, us alleles) { u->insert_kmer(count, alleles); }
static vector<double> triples(const vector<GenotypingResult>& rs) {
    vector<double> out;
    for (auto& r : rs) {
        out.push_back((double)r.get_genotype_likelihood(0, 0));
        out.push_back((double)r.get_genotype_likelihood(0, 1));
        out.push_back((double)r.get_genotype_likelihood(1, 1));
    }
    return out;
}

// ----------------------------------------------------------------------------------- CPU
static void cpu_tests() {
    run("KmerPath windows", [] {
        KmerPath p32(32), p16(16);
        p32.set_position(5); p32.set_position(36);
        CHECK(p32.get_position(5) == 1 && p32.get_position(36) == 1 && p32.get_position(6) == 0 && p32.get_position(40) == 0);
        CHECK(p32.nr_kmers() == 2 && p32.offset() == 5);
        CHECK_THROWS(p32.set_position(37));
        CHECK_THROWS(p32.set_position(4));
        p16.set_position(3);
        CHECK_THROWS(p16.set_position(19));
        p16.set_position(18);
        CHECK(p16.nr_kmers() == 2 && p16.mask() == ((1u << 0) | (1u << 15)));
        CHECK(p16.convert_to_string().substr(0, 5) == "00010");
    });
    run("UniqueKmers biallelic basics", [] {
        us paths = {0, 1, 1};
        BiallelicUniqueKmers u(1000, paths);
        CHECK(u.get_variant_position() == 1000 && u.get_nr_paths() == 3 && u.size() == 0);
        us a0 = {0}, a1 = {1}, both = {0, 1}, bad = {2};
        u.insert_kmer(4, a0); u.insert_kmer(6, a1); u.insert_kmer(8, both);
        CHECK(u.size() == 3 && u.get_readcount_of(1) == 6);
        CHECK(u.kmer_on_allele(0, 0) && !u.kmer_on_allele(0, 1) && u.kmer_on_allele(2, 0) && u.kmer_on_allele(2, 1));
        CHECK(u.kmer_on_path(1, 1) && !u.kmer_on_path(1, 0));
        CHECK_THROWS(u.insert_kmer(1, bad));
        CHECK_THROWS(u.kmer_on_path(0, 3));
        CHECK_THROWS(u.kmer_on_path(7, 0));
        CHECK_THROWS(u.get_readcount_of(3));
        u.update_readcount(0, 9);
        CHECK(u.get_readcount_of(0) == 9);
        CHECK_THROWS(u.update_readcount(5, 1));
        u.set_coverage(27);
        CHECK(u.get_coverage() == 27);
        us p, a;
        u.get_path_ids(p, a);
        CHECK(p == us({0, 1, 2}) && a == us({0, 1, 1}));
        us only = {2, 7}; p.clear(); a.clear();
        u.get_path_ids(p, a, &only);
        CHECK(p == us({2}) && a == us({1}));
        CHECK(!u.is_undefined_allele(1));
        u.set_undefined_allele(1);
        CHECK(u.is_undefined_allele(1));
        CHECK_THROWS(u.is_undefined_allele(2));
        us ids, defined;
        u.get_allele_ids(ids); u.get_defined_allele_ids(defined);
        CHECK(ids == us({0, 1}) && defined == us({0}));
        CHECK(u.kmers_on_allele(0) == 2 && u.present_kmers_on_allele(0) == 2 && u.kmers_on_alleles()[1] == 2);
        CHECK(u.get_allele(2) == 1);
        CHECK_THROWS(u.get_allele(3));
        us wrong = {0, 3};
        CHECK_THROWS(BiallelicUniqueKmers(5, wrong));
    });
    run("UniqueKmers multiallelic + update_paths", [] {
        us paths = {0, 2, 1, 1};
        MultiallelicUniqueKmers u(2000, paths);
        us a0 = {0}, a1 = {1}, a2 = {2}, a3 = {3};
        u.insert_kmer(10, a0); u.insert_kmer(11, a1); u.insert_kmer(12, a2); u.insert_kmer(2, a3);
        us ids;
        u.get_allele_ids(ids);
        CHECK(ids == us({0, 1, 2, 3}));  // inserting a k-mer on an unseen allele creates it
        CHECK(!u.is_undefined_allele(9));
        CHECK_THROWS(u.set_undefined_allele(9));
        u.set_undefined_allele(2);
        CHECK(u.present_kmers_on_allele(3) == 0 && u.fraction_present_kmers_on_allele(3) == 0.0f);
        us keep = {0, 1};
        u.update_paths(keep);
        CHECK(u.get_nr_paths() == 2 && u.get_allele(1) == 2 && u.size() == 2);
        CHECK(u.get_readcount_of(0) == 10 && u.get_readcount_of(1) == 12 && u.is_undefined_allele(2));
        ids.clear(); u.get_allele_ids(ids);
        CHECK(ids == us({0, 2}));
    });
    run("CopyNumber", [] {
        CHECK(close(CopyNumber(0.9, 0.1, 0.0).get_probability_of(0), 0.9));
        CHECK(close(CopyNumber(0.0, 0.0, 1.0).get_probability_of(2), 1.0));
        CopyNumber s(0.001, 0.6, 0.0004, 0.0);
        CHECK(close(s.get_probability_of(0), 0.001 / 0.6014) && close(s.get_probability_of(1), 0.6 / 0.6014) && close(s.get_probability_of(2), 0.0004 / 0.6014));
        CopyNumber r(0.2, 0.9, 1.1, 100.0);
        CHECK(close(r.get_probability_of(0), 0.33156849768) && close(r.get_probability_of(1), 0.33388484447) && close(r.get_probability_of(2), 0.33454665784));
        CHECK(CopyNumber(0.1, 0.2, 0.7) == CopyNumber(0.1, 0.2, 0.7) && CopyNumber(0.1, 0.2, 0.7) != CopyNumber(0.0, 1.0, 0.0));
        CHECK_THROWS(r.get_probability_of(3));
        CHECK(close(CopyNumber().get_probability_of(0), 1.0) && close(CopyNumber().get_probability_of(2), 0.0));
    });
    run("ProbabilityTable", [] {
        ProbabilityTable p(4, 7, 2, 0.0L);
        CHECK(close(p.get_probability(5, 0).get_probability_of(0), 0.99) && close(p.get_probability(5, 0).get_probability_of(1), 0.08208499862));
        CHECK(close(p.get_probability(5, 1).get_probability_of(2), 0.03368973499) && close(p.get_probability(6, 1).get_probability_of(1), 0.149361205103));
        CHECK(close(p.get_probability(6, 5).get_probability_of(0), 0.99 * std::pow(0.01, 5)));  // outside the box: on the fly
        p.modify_probability(5, 1, CopyNumber(0.1, 0.2, 0.7));
        CHECK(close(p.get_probability(5, 1).get_probability_of(2), 0.7));
        CHECK_THROWS(p.modify_probability(9, 0, CopyNumber()));
    });
    run("GenotypingResult", [] {
        GenotypingResult r;
        CHECK(r.contains_no_likelihoods() && r.get_likeliest_genotype() == std::make_pair(-1, -1));
        r.add_to_likelihood(0, 1, 0.1); r.add_to_likelihood(1, 0, 0.1); r.add_to_likelihood(0, 0, 0.05); r.add_to_likelihood(1, 1, 0.25);
        CHECK(close(r.get_genotype_likelihood(1, 0), 0.2) && close(r.get_genotype_likelihood(2, 2), 0.0));
        CHECK_THROWS(r.get_genotype_quality(1, 1));  // not normalised yet
        r.normalize();
        CHECK(close(r.get_genotype_likelihood(1, 1), 0.5) && r.get_likeliest_genotype() == std::make_pair(1, 1));
        CHECK(r.get_genotype_quality(1, 1) == 3);
        vector<long double> all = r.get_all_likelihoods(2);
        CHECK(all.size() == 3 && close(all[0], 0.1) && close(all[1], 0.4) && close(all[2], 0.5));
        CHECK_THROWS(r.get_all_likelihoods(1));
        GenotypingResult tie;
        tie.add_to_likelihood(0, 0, 0.5); tie.add_to_likelihood(0, 1, 0.5);
        CHECK(tie.get_likeliest_genotype() == std::make_pair(-1, -1));
        GenotypingResult sure;
        sure.add_to_likelihood(0, 1, 1.0);
        CHECK(sure.get_genotype_quality(0, 1) == 10000);
        us only = {1};
        GenotypingResult spec = r.get_specific_likelihoods(only);
        CHECK(close(spec.get_genotype_likelihood(0, 0), 1.0));
        GenotypingResult other;
        other.add_to_likelihood(0, 2, 0.3);
        r.combine(other);
        CHECK(close(r.get_genotype_likelihood(0, 2), 0.3) && r.get_stored_likelihoods().size() == 4);
        r.set_coverage(7); r.set_unique_kmers(9);
        CHECK(r.coverage() == 7 && r.nr_unique_kmers() == 9);
    });
    run("ColumnIndexer + flatten", [] {
        auto u1 = bi(2000, {0, 1, 0, 0, 0}); kmer(u1, 10, {0}); kmer(u1, 10, {1}); u1->set_coverage(5);
        auto u2 = bi(2500, {0, 0, 1, 1, 1}); kmer(u2, 10, {0}); kmer(u2, 20, {1});
        auto u3 = bi(3000, {0, 0, 1, 1, 1}); kmer(u3, 20, {0}); kmer(u3, 5, {1}); u3->set_coverage(5);
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2, u3};
        us only = {2, 3};
        ColumnIndexer ci(&uks, &only);
        CHECK(ci.size() == 2 && ci.get_variant_id(0) == 1 && ci.get_variant_id(1) == 2 && ci.nr_paths() == 2);
        CHECK(ci.get_path(0) == 2 && ci.get_path(1) == 3 && ci.get_allele(0, 0) == 1 && ci.get_allele(1, 1) == 1);
        ColumnIndexer all(&uks, nullptr);
        CHECK(all.size() == 3 && all.nr_paths() == 5 && all.get_allele(1, 0) == 1 && all.get_allele(4, 2) == 1 && all.get_allele(1, 2) == 0);
        CHECK_THROWS(all.get_variant_id(3));
        CHECK_THROWS(all.get_path(5));
        CHECK_THROWS(all.get_allele(3, 3));
        CHECK(all.get_path_ids_at(7) == std::make_pair((unsigned short)1, (unsigned short)2));
        FlatContig f;
        flatten(&uks, &only, f);
        CHECK(f.batch.n_variants == 3 && f.batch.n_paths == 2 && f.path_allele == vector<uint16_t>({0, 0, 1, 1, 1, 1}));
        CHECK(f.kmer_off == vector<uint32_t>({0, 2, 4, 6}) && f.allele_kmer_off[1] == 1 && f.allele_kmer_mask[1] == 1 && f.coverage[0] == 5);
        us none = {8, 9};
        CHECK_THROWS(flatten(&uks, &none, f));
    });
}

// ----------------------------------------------------------------------------------- GPU
static const double R01 = 446.287102628;  // recombination rate that gives recombination probability 0.1

static std::string g_golden_dir = "tests/golden";
// the Results object of the archive tests: two chromosomes, likelihood maps, haplotypes, meta data
static Results sample_results() {
    Results r;
    GenotypingResult a;
    a.add_to_likelihood(0, 0, 0.5L); a.add_to_likelihood(0, 1, 0.25L); a.add_to_likelihood(1, 1, 1.0L / 3.0L);
    a.add_first_haplotype_allele(1); a.add_second_haplotype_allele(0); a.set_coverage(27); a.set_unique_kmers(20);
    GenotypingResult b;  // no likelihoods (a skipped variant)
    b.set_coverage(3);
    GenotypingResult c;
    c.add_to_likelihood(2, 5, 1e-4000L);  // below the double range: the archive keeps the long double
    c.add_first_haplotype_allele(5); c.add_second_haplotype_allele(2); c.set_unique_kmers(301);
    r.result["chr1"] = {a, b};
    r.result["chr10"] = {c};
    r.runtimes["chr1"] = 1.5; r.runtimes["chr10"] = 0.125;
    return r;
}

static std::vector<unsigned char> read_file(const std::string& path) {
    std::vector<unsigned char> bytes;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return bytes;
    unsigned char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return bytes;
}

static std::string gunzip_text(const std::string& path) {
    gzFile z = gzopen(path.c_str(), "rb");
    std::string text;
    char buf[1 << 14];
    int got;
    while (z && (got = gzread(z, buf, sizeof buf)) > 0) text.append(buf, (size_t)got);
    if (z) gzclose(z);
    return text;
}

static void index_builder_cpu_tests() {
    run("build_index: the reference's region.fa + region.vcf give its index_* fixtures (PanGenie-index)", [] {
        // tests/data/region.{fa,vcf} at k = 31 with the reference path added -> tests/data/index_path_segments.fasta,
        // index_chr1_Graph.cereal, index_chr1_kmers.tsv.gz, index_UniqueKmersMap.cereal: the files tests/CommandsTest.cpp runs
        // PanGenie-genotype on.  Segment file and k-mer table text for text; the archives byte for byte (the measured run time
        // the index archive carries is copied over).
        const std::string prefix = "/tmp/pg_test_index";
        const std::vector<std::string> chromosomes = build_index(g_golden_dir + "/region.fa", g_golden_dir + "/region.vcf", prefix, 31, true);
        CHECK(chromosomes == std::vector<std::string>({"chr1"}));
        const std::vector<unsigned char> seg = read_file(prefix + "_path_segments.fasta"), seg_want = read_file(g_golden_dir + "/index_path_segments.fasta");
        CHECK(seg == seg_want);
        CHECK(read_file(prefix + "_chr1_Graph.cereal") == read_file(g_golden_dir + "/index_chr1_Graph.cereal"));
        CHECK(gunzip_text(prefix + "_chr1_kmers.tsv.gz") == gunzip_text(g_golden_dir + "/index_chr1_kmers.tsv.gz"));
        UniqueKmersMap got = load_unique_kmers_map(prefix + "_UniqueKmersMap.cereal");
        const UniqueKmersMap want = load_unique_kmers_map(g_golden_dir + "/index_UniqueKmersMap.cereal");
        CHECK(got.kmersize == 31 && got.add_reference && got.unique_kmers["chr1"].size() == 2);
        got.runtimes = want.runtimes;
        got.sampling_runtimes = want.sampling_runtimes;
        CHECK(serialize_unique_kmers_map(got) == read_file(g_golden_dir + "/index_UniqueKmersMap.cereal"));
    });
    run("build_graphs on the reference's small VCFs (tests/GraphBuilderTest.cpp:18-74, :103-199, :259-318, :397-405, :432-438)", [] {
        const std::string dir = g_golden_dir + "/graphbuilder/";
        const ReferenceSequences reference(dir + "small1.fa");
        {   // get_allele_string: small1.vcf at k = 10 with the reference path
            const BuiltGraphs b = build_graphs(dir + "small1.vcf", reference, 10, true);
            CHECK(b.chromosomes == std::vector<std::string>({"chrA", "chrB"}) && b.nr_paths == 5 && b.graphs.size() == 2);
            const Graph& a = b.graphs.at("chrA");
            const Graph& bb = b.graphs.at("chrB");
            CHECK(a.get_kmer_size() == 10 && bb.get_kmer_size() == 10 && a.get_chromosome() == "chrA" && bb.get_chromosome() == "chrB");
            CHECK(a.size() == 7 && bb.size() == 2 && a.get_variant(2).nr_of_alleles() == 3 && a.get_variant(2).nr_of_paths() == 5);
            const std::vector<std::vector<std::string>> want = {
                {"GGAATTCCGACATAAGTTA", "GGAATTCCGTCATAAGTTA"}, {"CCTTAGCTACGAAGCCAGT", "CCTTAGCTAGGGGGAAGCCAGT"},
                {"GAAGCCAGTGCCCCGAGACGGCCAAA", "GAAGCCAGTTCCCCGAGACGGCCAAA", "GAAGCCAGTTCCCCTACGGCCAAA"},
                {"ACGTCCGTTCAGCCTTAGC", "ACGTCCGTTTAGCCTTAGC"}, {"CCGATTTTCTTGTGCTATA", "CCGATTTTCCTGTGCTATA"},
                {"GGAGGGTATGAAGCCATCAC", "GGAGGGTATTCAGCCATCAC"}, {"TGTGGACTTATTTGGCTAA", "TGTGGACTTGTTTGGCTAA"}};
            for (size_t v = 0; v < want.size() && a.size() == 7; ++v)
                for (size_t al = 0; al < want[v].size(); ++al) CHECK(a.get_variant(v).get_allele_string(al) == want[v][al]);
            CHECK(bb.get_variant(0).get_allele_string(0) == "CCACTTCATCAAGACACAA" && bb.get_variant(1).get_allele_string(0) == "GAGTATTTTGATCATAAAT");
        }
        {   // a record on its own keeps every VCF allele, carried or not, bubble allele = VCF allele (the reference's Variant
            // constructor; tests/VariantTest.cpp:384-390 "uncovered_single": {A, G, T} with paths {0,0,1,0} stays three alleles);
            // merged records keep the combinations their paths carry (+ all-REF)
            const std::string dir2 = "/tmp/pg_test_uncovered";
            { std::ofstream f(dir2 + ".fa"); f << ">chrU\n" << std::string(30, 'A') + "CAGTCAGTCAGGTTTACCATGACCATGGCAT" + std::string(30, 'C') << "\n"; }
            {
                std::ofstream f(dir2 + ".vcf");
                f << "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n";
                f << "chrU\t35\t.\tC\tT,G\t.\tPASS\t.\tGT\t1|0\t0|1\n";      // ALT G carried by nobody
                f << "chrU\t55\t.\tA\tG,C\t.\tPASS\t.\tGT\t0|2\t2|0\n";      // ALT G carried by nobody, paths carry 0 and 2
            }
            const ReferenceSequences ref2(dir2 + ".fa");
            const BuiltGraphs b = build_graphs(dir2 + ".vcf", ref2, 10, false);
            const Graph& g = b.graphs.at("chrU");
            CHECK(g.size() == 2);
            if (g.size() == 2) {
                const Variant& v0 = g.get_variant(0);
                const Variant& v1 = g.get_variant(1);
                CHECK(v0.nr_of_alleles() == 3 && v1.nr_of_alleles() == 3);
                CHECK(v0.get_allele_on_path(0) == 1 && v0.get_allele_on_path(1) == 0 && v0.get_allele_on_path(2) == 0 && v0.get_allele_on_path(3) == 1);
                CHECK(v1.get_allele_on_path(0) == 0 && v1.get_allele_on_path(1) == 2 && v1.get_allele_on_path(2) == 2 && v1.get_allele_on_path(3) == 0);
                const std::string a2 = v0.get_allele_string(2);
                CHECK(a2.size() == 19 && a2[9] == 'G');   // the uncovered allele's sequence is in the graph
            }
        }
        auto lines_of = [](const std::string& text) { std::vector<std::string> l; std::istringstream is(text); std::string t; while (std::getline(is, t)) l.push_back(t); return l; };
        auto file_text = [](const std::string& path) { const std::vector<unsigned char> raw = read_file(path); return std::string(raw.begin(), raw.end()); };
        auto trimmed = [](std::string t) { const size_t b = t.find_first_not_of(" \t\r\n"), e = t.find_last_not_of(" \t\r\n"); return b == std::string::npos ? std::string() : t.substr(b, e - b + 1); };
        const BuiltGraphs plain = build_graphs(dir + "small1.vcf", reference, 10, false);
        CHECK(plain.nr_paths == 4);
        {   // write_path_segments: the stretches of reference between the bubbles (small1-expected-ref-segments.fa)
            std::vector<std::string> got, want;
            for (const std::string& l : lines_of(file_text(dir + "small1-expected-ref-segments.fa"))) want.push_back(trimmed(l));
            bool take = false;
            for (const std::string& l : lines_of(path_segments_fasta(plain, reference))) {
                if (l.empty()) continue;
                if (l[0] == '>') { take = l.find("reference") != std::string::npos; continue; }
                if (take) got.push_back(trimmed(l));
            }
            CHECK(got == want);
        }
        {   // no variants at all: the segment file is the reference (empty.vcf)
            const BuiltGraphs none = build_graphs(dir + "empty.vcf", reference, 10, false);
            std::vector<std::string> got, want;
            for (const std::string& l : lines_of(path_segments_fasta(none, reference))) if (!l.empty() && l[0] != '>') got.push_back(trimmed(l));
            for (const std::string& n : reference.names()) want.push_back(reference.of(n));
            std::sort(got.begin(), got.end()); std::sort(want.begin(), want.end());
            CHECK(got == want && got.size() >= 2);
        }
        {   // write_genotypes_of / write_phasing_of with the likelihoods of the reference's test: small1-genotypes.vcf and
            // small1-phasing.vcf are what an early release wrote for them (AK / KC still in INFO, the first GL with six digits),
            // so the columns both layouts share are compared: CHROM .. FILTER, AF, GT, GQ, GL to four digits, the phased GT
            std::vector<GenotypingResult> ga(7), gb(2);
            for (size_t i = 0; i < 7; ++i) {
                if (i == 2) continue;
                ga[i].add_to_likelihood(0, 0, 0.2L); ga[i].add_to_likelihood(0, 1, 0.7L); ga[i].add_to_likelihood(1, 1, 0.1L);
            }
            ga[2].add_to_likelihood(0, 0, 0.2L); ga[2].add_to_likelihood(0, 1, 0.0L); ga[2].add_to_likelihood(0, 2, 0.2L);
            ga[2].add_to_likelihood(1, 1, 0.0L); ga[2].add_to_likelihood(1, 2, 0.5L); ga[2].add_to_likelihood(2, 2, 0.1L);
            ga[2].add_first_haplotype_allele(2); ga[2].add_second_haplotype_allele(1);
            for (auto& r : gb) { r.add_to_likelihood(0, 0, 0.1L); r.add_to_likelihood(0, 1, 0.1L); r.add_to_likelihood(1, 1, 0.8L); }
            auto cols = [](const std::string& l) { std::vector<std::string> f; std::string t; std::istringstream is(l); while (std::getline(is, t, '\t')) f.push_back(t); return f; };
            auto records_of = [&](const std::string& path) { std::vector<std::vector<std::string>> r; for (const std::string& l : lines_of(file_text(path))) if (!l.empty() && l[0] != '#') r.push_back(cols(l)); return r; };
            auto split = [](const std::string& t, char sep) { std::vector<std::string> f; std::string x; std::istringstream is(t); while (std::getline(is, x, sep)) f.push_back(x); return f; };
            std::vector<std::string> got = plain.graphs.at("chrA").genotypes_records(ga), got_b = plain.graphs.at("chrB").genotypes_records(gb);
            got.insert(got.end(), got_b.begin(), got_b.end());
            const auto want = records_of(dir + "small1-genotypes.vcf");
            CHECK(got.size() == 10 && want.size() == 10);
            for (size_t i = 0; i < got.size() && i < want.size(); ++i) {
                const std::vector<std::string> g = cols(got[i]), &w = want[i];
                bool same = g.size() == 10 && w.size() == 10;
                for (size_t c = 0; same && c < 7; ++c) same = g[c] == w[c];
                same = same && split(g[7], ';')[0] == split(w[7], ';')[0] && g[8] == "GT:GQ:GL:KC";
                if (same) {
                    const std::vector<std::string> gs = split(g[9], ':'), ws = split(w[9], ':');
                    same = gs.size() == 4 && ws.size() == 3 && gs[0] == ws[0] && gs[1] == ws[1];
                    const std::vector<std::string> gl = split(gs[2], ','), wl = split(ws[2], ',');
                    same = same && gl.size() == wl.size();
                    for (size_t j = 0; same && j < gl.size(); ++j) same = std::fabs(std::stod(gl[j]) - std::stod(wl[j])) <= 5e-4 * std::fabs(std::stod(wl[j]));
                }
                if (!same) std::printf("  record %zu: %s\n", i, got[i].c_str());
                CHECK(same);
            }
            std::vector<std::string> ph = plain.graphs.at("chrA").phasing_records(ga), ph_b = plain.graphs.at("chrB").phasing_records(gb);
            ph.insert(ph.end(), ph_b.begin(), ph_b.end());
            const auto want_ph = records_of(dir + "small1-phasing.vcf");
            CHECK(ph.size() == 10 && want_ph.size() == 10);
            for (size_t i = 0; i < ph.size() && i < want_ph.size(); ++i) {
                const std::vector<std::string> g = cols(ph[i]), &w = want_ph[i];
                bool same = g.size() == 10 && w.size() == 10;
                for (size_t c = 0; same && c < 7; ++c) same = g[c] == w[c];
                same = same && split(g[7], ';')[0] == split(w[7], ';')[0] && split(g[9], ':')[0] == w[9];
                if (!same) std::printf("  phasing record %zu: %s\n", i, ph[i].c_str());
                CHECK(same);
            }
        }
        auto refused = [&](const std::string& vcf) { try { (void)build_graphs(dir + vcf, reference, 10, false); } catch (const std::runtime_error&) { return true; } return false; };
        CHECK(refused("no-paths.vcf") && refused("malformatted-vcf1.vcf") && refused("overlapping-variants.vcf"));
        CHECK(build_graphs(dir + "no-alt-alleles.vcf", reference, 10, false).graphs.at("chrA").size() == 1);   // symbolic ALT alleles are skipped
        CHECK(build_graphs(dir + "small2.vcf", reference, 10, false).chromosomes == std::vector<std::string>({"chrB", "chrC", "chrA"}));   // by number of bubbles
        CHECK(!refused("small3.vcf"));   // `.` alleles in the panel
        {   // variant_ids2: ids of a multi-ALT record and of two ids on one allele (small1-ids.vcf)
            const BuiltGraphs ids = build_graphs(dir + "small1-ids.vcf", reference, 10, true);
            CHECK(ids.graphs.size() == 1 && ids.graphs.at("chrA").size() == 2);
            const std::vector<std::string> lines = ids.graphs.at("chrA").genotypes_records(std::vector<GenotypingResult>(2));
            CHECK(lines.size() == 2 && lines[0].find("\tGGGG,A,T\t") != std::string::npos && lines[0].find(";ID=var1:var2,var3,var4\t") != std::string::npos);
            CHECK(lines.size() == 2 && lines[1].find(";ID=var5:var6\t") != std::string::npos);
        }
        {   // write_sampled_panel (tests/GraphBuilderTest.cpp:453-504): small4.vcf, two records in one bubble, `.` haplotypes
            const BuiltGraphs four = build_graphs(dir + "small4.vcf", reference, 10, false);
            CHECK(four.chromosomes == std::vector<std::string>({"chrA"}) && four.graphs.at("chrA").size() == 1);
            const Variant& bubble = four.graphs.at("chrA").get_variant(0);
            SampledPanel panel;
            panel.unique_kmers = 14;
            for (size_t p = 0; p < bubble.nr_of_paths(); ++p) panel.path_to_allele.push_back(bubble.get_allele_on_path(p));
            const std::vector<std::string> lines = four.graphs.at("chrA").sampled_panel_records({panel});
            const std::string want1 = "chrA\t161\t.\tG\tTA,TAAA\t.\tPASS\tAF=0.375,0.416667;UK=14;MA=2\tGT\t0\t1\t1\t1\t2\t1\t2\t2\t2\t1\t1\t0\t2\t2\t.\t.\t1\t2\t2\t1\t2\t2\t1\t0";
            const std::string want2 = "chrA\t166\t.\tG\tT\t.\tPASS\tAF=0.666667;UK=14;MA=6\tGT\t.\t1\t.\t.\t.\t1\t1\t1\t1\t1\t1\t0\t1\t1\t.\t.\t1\t1\t1\t1\t1\t1\t1\t0";
            CHECK(lines.size() == 2 && lines[0] == want1 && lines[1] == want2);
            if (lines.size() == 2 && (lines[0] != want1 || lines[1] != want2)) std::printf("  %s\n  %s\n", lines[0].c_str(), lines[1].c_str());
            const std::vector<std::string> head = Graph::sampled_panel_header(3, "20240101");
            CHECK(head.size() == 8 && head.back() == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsampledHT0\tsampledHT1\tsampledHT2");
            CHECK_THROWS(four.graphs.at("chrA").sampled_panel_records({}));
        }
        {   // close_to_start: a record closer than 2 k to the chromosome start is left out (close.vcf, k = 31)
            const ReferenceSequences close_ref(dir + "close.fa");
            const BuiltGraphs close = build_graphs(dir + "close.vcf", close_ref, 31, true);
            CHECK(close.graphs.at("chr10").size() == 1 && close.skipped == 1);
            const std::vector<std::string> lines = close.graphs.at("chr10").genotypes_records(std::vector<GenotypingResult>(1));
            CHECK(lines.size() == 1 && lines[0].rfind("chr10\t79\t.\tT\tC,A\t.\tPASS\tAF=0.5,0;UK=0;MA=0;ID=testvar2,testvar3\tGT:GQ:GL:KC\t", 0) == 0);
        }
    });
    run("unique_kmers_of on the reference's small VCFs (tests/UniqueKmerComputerTest.cpp:85-150)", [] {
        const std::string dir = g_golden_dir + "/graphbuilder/";
        const ReferenceSequences reference(dir + "small1.fa");
        auto unique_kmers = [&](const std::string& vcf, const std::string& chromosome) {
            const BuiltGraphs b = build_graphs(dir + vcf, reference, 31, true);
            const std::string segments = "/tmp/pg_test_uk_segments.fa";
            { std::FILE* f = std::fopen(segments.c_str(), "w"); std::fputs(path_segments_fasta(b, reference).c_str(), f); std::fclose(f); }
            ExactKmerCounter graph_kmers(segments, 31);
            return unique_kmers_of(b.graphs.at(chromosome), graph_kmers);
        };
        // every k-mer lies on exactly one allele: the per-allele numbers add up to the total
        auto adds_up = [](const std::shared_ptr<UniqueKmers>& u) {
            size_t summed = 0;
            for (const auto& kv : u->kmers_on_alleles()) if (kv.second > 0) summed += (size_t)kv.second;
            return summed == u->size();
        };
        const ChromosomeKmers a = unique_kmers("small1.vcf", "chrA");   // 8 records, three of them within 30 bases: 6 bubbles
        CHECK(a.objects.size() == 6 && a.rows.size() == 6);
        for (const auto& u : a.objects) CHECK(adds_up(u));
        const ChromosomeKmers b = unique_kmers("small5.vcf", "chrB");   // one record with many long alleles
        CHECK(b.objects.size() == 1);
        for (const auto& u : b.objects) CHECK(adds_up(u) && u->size() <= 301 && u->size() > 0);
    });
    run("fill_read_kmercounts_all: chromosomes on worker threads = one after the other", [] {
        // the reference's small2.vcf: bubbles on chrA, chrB and chrC; the graph itself serves as reads
        const std::string dir = g_golden_dir + "/graphbuilder/", prefix = "/tmp/pg_test_fill_all";
        CHECK(build_index(dir + "small1.fa", dir + "small2.vcf", prefix, 10, true).size() == 3);
        TargetedKmerCounter reads(10);
        reads.add_targets_from_sequences(prefix + "_path_segments.fasta");
        reads.count(prefix + "_path_segments.fasta", 2);
        UniqueKmersMap serial = load_unique_kmers_map(prefix + "_UniqueKmersMap.cereal"), threaded = load_unique_kmers_map(prefix + "_UniqueKmersMap.cereal");
        for (const auto& kv : serial.unique_kmers) fill_read_kmercounts(kv.first, &serial, reads, prefix + "_" + kv.first + "_kmers.tsv.gz", 2);
        fill_read_kmercounts_all(&threaded, reads, prefix, 2, 3);
        CHECK(serialize_unique_kmers_map(serial) == serialize_unique_kmers_map(threaded));
        size_t counted = 0;
        for (const auto& kv : threaded.unique_kmers) for (const auto& u : kv.second) for (size_t i = 0; i < u->size(); ++i) counted += u->get_readcount_of(i);
        CHECK(counted > 0);
        UniqueKmersMap broken = load_unique_kmers_map(prefix + "_UniqueKmersMap.cereal");
        CHECK_THROWS(fill_read_kmercounts_all(&broken, reads, "/tmp/pg_no_such_prefix", 2, 3));
    });
    run("build_index with worker threads: the same files as with one", [] {
        // 60 kb of pseudo-random reference, 300 records (SNPs, a deletion and a two-ALT insertion now and then), 6 haplotypes
        std::string ref;
        uint64_t x = 0x2545F4914F6CDD1Dull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        for (int i = 0; i < 60000; ++i) ref += "ACGT"[rnd() & 3];
        const std::string fa = "/tmp/pg_test_mt.fa", vcf = "/tmp/pg_test_mt.vcf";
        { std::FILE* f = std::fopen(fa.c_str(), "w"); std::fprintf(f, ">chrT\n%s\n", ref.c_str()); std::fclose(f); }
        {
            std::FILE* f = std::fopen(vcf.c_str(), "w");
            std::fprintf(f, "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\ts3\n");
            for (size_t i = 0, pos = 150; i < 300; ++i, pos += 60 + rnd() % 250) {
                const char r = ref[pos];
                const std::string other(1, r == 'A' ? 'C' : 'A');
                std::string refa(1, r), alt = other;
                int n_alt = 1;
                if (i % 17 == 3) refa = ref.substr(pos, 6), alt = std::string(1, r);
                if (i % 23 == 5) { alt = refa + "GATTACA," + refa + "TT"; n_alt = 2; }
                std::string gts;
                bool used[3] = {false, false, false};
                for (int h = 0; h < 6; ++h) {
                    const int a = (int)(rnd() % (size_t)(n_alt + 1));
                    used[a] = true;
                    gts += (h % 2 ? "|" : "\t") + std::to_string(a);
                }
                if (!used[1] || (n_alt == 2 && !used[2])) gts = n_alt == 2 ? "\t1|2\t0|0\t0|1" : "\t1|0\t0|0\t0|1";
                std::fprintf(f, "chrT\t%zu\t.\t%s\t%s\t.\tPASS\t.\tGT%s\n", pos + 1, refa.c_str(), alt.c_str(), gts.c_str());
            }
            std::fclose(f);
        }
        const std::string one = "/tmp/pg_test_mt1", four = "/tmp/pg_test_mt4";
        CHECK(build_index(fa, vcf, one, 31, true, 1) == std::vector<std::string>({"chrT"}));
        CHECK(build_index(fa, vcf, four, 31, true, 4) == std::vector<std::string>({"chrT"}));
        CHECK(read_file(one + "_path_segments.fasta") == read_file(four + "_path_segments.fasta"));
        CHECK(read_file(one + "_chrT_Graph.cereal") == read_file(four + "_chrT_Graph.cereal"));
        CHECK(read_file(one + "_UniqueKmersMap.cereal") == read_file(four + "_UniqueKmersMap.cereal"));
        // ... and as with every k-mer of the graph counted in one table (the default counts, per chromosome, only the k-mers
        // the selection can ask about)
        const std::string whole = "/tmp/pg_test_mtw";
        CHECK(build_index(fa, vcf, whole, 31, true, 2, true) == std::vector<std::string>({"chrT"}));
        CHECK(read_file(one + "_UniqueKmersMap.cereal") == read_file(whole + "_UniqueKmersMap.cereal"));
        CHECK(gunzip_text(one + "_chrT_kmers.tsv.gz") == gunzip_text(whole + "_chrT_kmers.tsv.gz"));
        const std::string table = gunzip_text(one + "_chrT_kmers.tsv.gz");
        CHECK(table == gunzip_text(four + "_chrT_kmers.tsv.gz") && std::count(table.begin(), table.end(), '\n') > 200);
        const UniqueKmersMap m = load_unique_kmers_map(four + "_UniqueKmersMap.cereal");
        CHECK(m.unique_kmers.at("chrT").size() + 1 == (size_t)std::count(table.begin(), table.end(), '\n'));
        // and the table is what the count filling reads back: every row's k-mers are the object's
        UniqueKmersMap filled = m;
        ExactKmerCounter reads(one + "_path_segments.fasta", 31);   // (the graph itself as "reads": every unique k-mer is seen once)
        fill_read_kmercounts("chrT", &filled, reads, four + "_chrT_kmers.tsv.gz", 1);
        size_t ones = 0, kmers = 0;
        for (const auto& u : filled.unique_kmers.at("chrT")) for (size_t i = 0; i < u->size(); ++i) { kmers += 1; ones += u->get_readcount_of(i) == 1; }
        CHECK(kmers > 1000 && ones == kmers);
    });
    run("an ALT allele no path carries, inside a merged bubble (tests/VariantTest.cpp:559-627 through VCF and writer)", [] {
        // chr2 ...ATGA A CTG A CTG...: A>T at 4 (paths 0, 1) and G>C,T at 7 (paths 0, 2: C is on no path), k = 5
        std::string pad;
        uint64_t x = 0x853C49E6748FEA9Bull;
        for (int i = 0; i < 30; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; pad += "ACGT"[x & 3]; }
        const std::string fa = "/tmp/pg_test_unc.fa", vcf = "/tmp/pg_test_unc.vcf";
        { std::FILE* f = std::fopen(fa.c_str(), "w"); std::fprintf(f, ">chr2\n%sATGAACTGACTG%s\n", pad.c_str(), pad.c_str()); std::fclose(f); }
        { std::FILE* f = std::fopen(vcf.c_str(), "w");
          std::fprintf(f, "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts\nchr2\t35\t.\tA\tT\t.\t.\t.\tGT\t0|1\nchr2\t38\t.\tG\tC,T\t.\t.\t.\tGT\t0|2\n");
          std::fclose(f); }
        const ReferenceSequences reference(fa);
        const BuiltGraphs b = build_graphs(vcf, reference, 5, false);
        const Graph& g = b.graphs.at("chr2");
        CHECK(g.size() == 1 && g.get_variant(0).is_combined() && g.get_variant(0).nr_of_alleles() == 2 && g.get_variant(0).nr_of_records() == 2);
        GenotypingResult r;
        r.add_to_likelihood(0, 0, 0.05L); r.add_to_likelihood(0, 1, 0.05L); r.add_to_likelihood(1, 1, 0.9L);
        r.add_first_haplotype_allele(0); r.add_second_haplotype_allele(0);
        r.set_unique_kmers(12);
        const std::vector<std::string> lines = g.genotypes_records({r});
        CHECK(lines.size() == 2);
        // the likelihoods land on the record's own alleles: (0,0) (0,T) (T,T) = bins 0, 3 and 5 of G / C / T
        CHECK(lines.size() == 2 && lines[0] == "chr2\t35\t.\tA\tT\t.\tPASS\tAF=0.5;UK=12;MA=0\tGT:GQ:GL:KC\t1/1:9:-1.301,-1.301,-0.04576:0");
        CHECK(lines.size() == 2 && lines[1] == "chr2\t38\t.\tG\tC,T\t.\tPASS\tAF=0,0.5;UK=12;MA=0\tGT:GQ:GL:KC\t2/2:9:-1.301,-inf,-inf,-1.301,-inf,-0.04576:0");
        if (g_failed) for (const std::string& l : lines) std::printf("  %s\n", l.c_str());
        const std::vector<std::string> ph = g.phasing_records({r});
        CHECK(ph.size() == 2 && ph[0].substr(ph[0].rfind('\t') + 1) == "0|0:0" && ph[1].substr(ph[1].rfind('\t') + 1) == "0|0:0");
    });
    run("ReferenceSequences on the reference's FASTA known answers (tests/FastaReaderTest.cpp:9-48)", [] {
        const std::string dir = g_golden_dir + "/graphbuilder/";
        const ReferenceSequences f(dir + "simple-fasta.fa");
        CHECK(f.contains("chr01") && f.contains("chr02") && !f.contains("chr03"));
        CHECK(f.of("chr01").size() == 1688 && f.of("chr02").size() == 2135 && f.names() == std::vector<std::string>({"chr01", "chr02"}));
        CHECK_THROWS(f.of("chrNone"));
        CHECK(f.of("chr01").substr(0, 10) == "CATTTTAAAG" && f.of("chr01").substr(21, 19) == "CCCAGAGCAGGCAAAACCC");
        CHECK(f.of("chr02").substr(1, 11) == "CCAACAATTTA" && f.of("chr02").substr(71, 10) == "TCAAATCACA");
        CHECK_THROWS(ReferenceSequences(dir + "broken-fasta.fa"));   // sequence before the first header
        CHECK_THROWS(ReferenceSequences("/tmp/pg_no_such.fa"));
    });
    run("two bubbles exactly k - 1 apart: the stretch between them is shorter than k (src/stepwiseuniquekmercomputer.cpp:11-35)", [] {
        // k = 11, SNPs at 100 and 111: ten reference bases between them, not merged (src/graphbuilder.cpp:186).  The reference's
        // k-mer register, started as A's, turns those ten bases into the 11-mer 'A' + bases; one allele of the first SNP is an
        // A here, so that 11-mer exists once in the graph and becomes a flanking k-mer of BOTH bubbles
        std::string ref;
        uint64_t x = 0xA0761D6478BD642Full;
        for (int i = 0; i < 400; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; ref += "ACGT"[x & 3]; }
        const std::string fa = "/tmp/pg_test_k1.fa", vcf = "/tmp/pg_test_k1.vcf";
        { std::FILE* f = std::fopen(fa.c_str(), "w"); std::fprintf(f, ">c\n%s\n", ref.c_str()); std::fclose(f); }
        auto alt_of = [&](size_t pos) { return ref[pos] == 'A' ? 'C' : 'A'; };
        { std::FILE* f = std::fopen(vcf.c_str(), "w");
          std::fprintf(f, "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts\nc\t101\t.\t%c\t%c\t.\t.\t.\tGT\t0|1\nc\t112\t.\t%c\t%c\t.\t.\t.\tGT\t1|0\n",
                       ref[100], alt_of(100), ref[111], alt_of(111));
          std::fclose(f); }
        const std::string prefix = "/tmp/pg_test_k1_idx";
        for (const bool whole : {false, true}) {
            CHECK(build_index(fa, vcf, prefix, 11, true, 1, whole).size() == 1);
            CHECK(Graph::load(prefix + "_c_Graph.cereal").size() == 2);
            std::vector<std::string> rows;
            { std::istringstream is(gunzip_text(prefix + "_c_kmers.tsv.gz")); std::string l; while (std::getline(is, l)) if (l[0] != '#') rows.push_back(l); }
            CHECK(rows.size() == 2);
            const std::string padded = "A" + ref.substr(101, 10);
            for (const std::string& row : rows) {
                const std::string flanking = row.substr(row.rfind('\t') + 1);
                CHECK(("," + flanking + ",").find("," + padded + ",") != std::string::npos);
            }
        }
    });
    run("build_graphs on damaged VCFs: a graph or a runtime_error, nothing else", [] {
        const std::string dir = g_golden_dir + "/graphbuilder/";
        const ReferenceSequences reference(dir + "small1.fa");
        const std::vector<unsigned char> raw = read_file(dir + "small1.vcf");
        const std::string tmp = "/tmp/pg_test_damaged.vcf";
        uint64_t x = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        size_t built = 0, refused = 0, other = 0;
        for (int round = 0; round < 400; ++round) {
            std::string text(raw.begin(), raw.end());
            const int edits = 1 + (int)(rnd() % 3);
            for (int e = 0; e < edits; ++e) {
                const size_t at = rnd() % text.size();
                switch (rnd() % 5) {
                    case 0: text.erase(at, 1 + rnd() % 12); break;                        // a hole
                    case 1: text[at] = "\t\n|/.,0123ACGTN;:"[rnd() % 18]; break;          // a wrong character
                    case 2: text.insert(at, 1, "\t\n|/.,9"[rnd() % 8]); break;            // an extra one
                    case 3: text.resize(at); break;                                       // cut short
                    default: text.insert(at, text.substr(at, std::min<size_t>(40, text.size() - at))); break;   // a repeat
                }
                if (text.empty()) text = "#";
            }
            { std::FILE* f = std::fopen(tmp.c_str(), "w"); std::fwrite(text.data(), 1, text.size(), f); std::fclose(f); }
            try { const BuiltGraphs b = build_graphs(tmp, reference, 10, round % 2 == 0); (void)path_segments_fasta(b, reference); built += 1; }
            catch (const std::runtime_error&) { refused += 1; }
            catch (const std::exception& e) { other += 1; if (other < 4) std::printf("  round %d: %s\n", round, e.what()); }
            catch (...) { other += 1; }
        }
        CHECK(other == 0 && built > 0 && refused > 0);
    });
    run("build_graphs: records closer than k - 1 merge into one bubble; what the reference refuses is refused", [] {
        // a 400-base reference without repeats of length >= 5 would be ideal; a fixed pseudo-random one serves
        std::string ref;
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 400; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; ref += "ACGT"[x & 3]; }
        const std::string fa = "/tmp/pg_test_ib.fa", vcf = "/tmp/pg_test_ib.vcf";
        { std::FILE* f = std::fopen(fa.c_str(), "w"); std::fprintf(f, ">c1 some description\n%s\n>c2\n%s\n", ref.c_str(), ref.substr(0, 120).c_str()); std::fclose(f); }
        auto alt_of = [&](size_t pos) { return std::string(1, ref[pos] == 'A' ? 'C' : 'A'); };
        auto write_vcf = [&](const std::string& body) {
            std::FILE* f = std::fopen(vcf.c_str(), "w");
            std::fprintf(f, "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n%s", body.c_str());
            std::fclose(f);
        };
        auto rec = [&](size_t pos0, const std::string& alt, const std::string& gts, const std::string& info = ".") {
            return "c1\t" + std::to_string(pos0 + 1) + "\t.\t" + std::string(1, ref[pos0]) + "\t" + alt + "\t.\t.\t" + info + "\tGT\t" + gts + "\n";
        };
        const size_t k = 11;
        // records at 100, 105 (5 apart: same bubble), 130 (own bubble), 140 with a missing haplotype; one too close to the start
        write_vcf(rec(5, alt_of(5), "0|1\t1|1") + rec(100, alt_of(100), "0|1\t1|0", "ID=a1") + rec(105, alt_of(105), "1|1\t0|0", "ID=b1") +
                  rec(130, alt_of(130) + ",G" + std::string(ref[130] == 'G' ? "T" : ""), "2|0\t1|0", "AC=1;ID=x,y") + rec(160, alt_of(160), ".|1\t0|0"));
        const ReferenceSequences reference(fa);
        CHECK(reference.names() == std::vector<std::string>({"c1", "c2"}) && reference.of("c1") == ref);
        const BuiltGraphs b = build_graphs(vcf, reference, k, true);
        CHECK(b.nr_paths == 5 && b.skipped == 1 && b.chromosomes == std::vector<std::string>({"c1"}));
        const Graph& g = b.graphs.at("c1");
        CHECK(g.size() == 3);
        const Variant& merged = g.get_variant(0);
        CHECK(merged.is_combined() && merged.nr_of_records() == 2 && merged.get_start_position() == 100 && merged.get_end_position() == 106);
        // paths: reference 0|0; s1 = (0,1), (1,1); s2 = (1,0), (0,0): combinations sorted (0,0) (0,1) (1,0) (1,1)
        CHECK(merged.nr_of_alleles() == 4 && merged.nr_of_paths() == 5);
        CHECK(merged.get_allele_on_path(0) == 0 && merged.get_allele_on_path(1) == 1 && merged.get_allele_on_path(2) == 3 && merged.get_allele_on_path(3) == 2 && merged.get_allele_on_path(4) == 0);
        CHECK(merged.get_allele_string(0) == ref.substr(90, 26));   // k - 1 flanking bases on both sides
        CHECK(merged.get_allele_string(3) == ref.substr(90, 10) + alt_of(100) + ref.substr(101, 4) + alt_of(105) + ref.substr(106, 10));
        CHECK(g.get_variant(1).nr_of_alleles() == 3 && !g.get_variant(1).is_combined());
        CHECK(g.variant_ids().size() == 4 && g.variant_ids()[0] == std::vector<std::string>({"a1"}) && g.variant_ids()[2].size() == 2);
        CHECK(g.get_variant(2).nr_of_alleles() == 3 && g.get_variant(2).is_undefined_allele(2));   // REF, ALT, the missing haplotype's own allele
        // the segment file: reference up to the first bubble, its alleles, ..., the rest; then the chromosome without variants
        const std::string seg = path_segments_fasta(b, reference);
        CHECK(seg.find(">c1_reference_100\n" + ref.substr(0, 100) + "\n>c1_100_0\n" + ref.substr(90, 26) + "\n") == 0);
        CHECK(seg.find(">c1_reference_end\n" + ref.substr(161) + "\n>c2_reference_end\n" + ref.substr(0, 120) + "\n") != std::string::npos);
        // unique k-mers of the bubbles against the graph's own k-mer counts
        { std::FILE* f = std::fopen("/tmp/pg_test_ib_segments.fa", "w"); std::fputs(seg.c_str(), f); std::fclose(f); }
        ExactKmerCounter graph_kmers("/tmp/pg_test_ib_segments.fa", k);
        const ChromosomeKmers ck = unique_kmers_of(g, graph_kmers);
        CHECK(ck.rows.size() == 3 && ck.objects.size() == 3);
        CHECK(ck.rows[0].rfind("c1\t100\t106\t", 0) == 0 && ck.objects[0]->get_nr_paths() == 5 && ck.objects[0]->size() > 0);
        CHECK(ck.objects[2]->is_undefined_allele(2) && !ck.objects[2]->is_undefined_allele(1));
        // refusals
        auto refused = [&](const std::string& body) {
            write_vcf(body);
            try { (void)build_graphs(vcf, reference, k, true); } catch (const std::runtime_error&) { return true; }
            return false;
        };
        CHECK(refused(rec(100, alt_of(100), "0|1\t1|0") + "c2\t51\t.\t" + std::string(1, ref[50]) + "\t" + alt_of(50) + "\t.\t.\t.\tGT\t0|1\t1|0\n" +
                      rec(200, alt_of(200), "0|1\t1|0")));                                  // c1 again after c2
        CHECK(refused(rec(100, alt_of(100), "0/1\t1|0")));                                   // unphased
        CHECK(refused(rec(100, alt_of(100), "0|1|1\t1|0")));                                 // not diploid
        CHECK(refused(rec(100, alt_of(100), "0|2\t1|0")));                                   // allele that does not exist
        CHECK(refused(rec(100, alt_of(100), "0|1\t1|0") + rec(100, alt_of(100), "0|1\t1|0")));   // overlapping records
        CHECK(refused("c1\t101\t.\t" + alt_of(100) + "\t" + std::string(1, ref[100]) + "\t.\t.\t.\tGT\t0|1\t1|0\n"));   // REF does not match
        CHECK(refused("c9\t101\t.\tA\tC\t.\t.\t.\tGT\t0|1\t1|0\n"));                    // unknown chromosome
        write_vcf(rec(100, "<DEL>", "0|1\t1|0") + rec(200, alt_of(200) + "N", "0|1\t1|0"));   // symbolic / undefined ALT: skipped, not refused
        const BuiltGraphs none = build_graphs(vcf, reference, k, true);
        CHECK(none.skipped == 2 && none.graphs.empty());
    });
}

static void kmer_count_cpu_tests() {
    run("fill_read_kmercounts: the reference's index + k-mer table + reads give the reference's counted archive", [] {
        // tests/data/index_UniqueKmersMap.cereal (PanGenie-index output), index_chr1_kmers.tsv.gz, region-reads.fa ->
        // tests/data/region_UniqueKmersList.cereal, the archive tests/CommandsTest.cpp:59-93 feeds its HMM
        // (k-mer abundance peak 18, tests/CommandsTest.cpp:56).  Byte for byte, once the run times the reference
        // measured are copied over.
        UniqueKmersMap m = load_unique_kmers_map(g_golden_dir + "/index_UniqueKmersMap.cereal");
        const UniqueKmersMap want = load_unique_kmers_map(g_golden_dir + "/region_UniqueKmersList.cereal");
        CHECK(m.kmersize == 31 && m.unique_kmers["chr1"].size() == 2);
        CHECK(m.unique_kmers["chr1"][0]->get_readcount_of(0) == 0);  // the index carries no counts
        ExactKmerCounter counts(g_golden_dir + "/region-reads.fa", m.kmersize);
        CHECK(counts.distinct_kmers() > 1000);
        fill_read_kmercounts("chr1", &m, counts, g_golden_dir + "/index_chr1_kmers.tsv.gz", 18);
        for (size_t v = 0; v < 2; ++v) {
            UniqueKmers& got = *m.unique_kmers["chr1"][v];
            UniqueKmers& exp = *want.unique_kmers.at("chr1")[v];
            CHECK(got.size() == exp.size() && got.get_coverage() == exp.get_coverage());
            size_t same = 0;
            for (size_t k = 0; k < got.size() && k < exp.size(); ++k) same += got.get_readcount_of(k) == exp.get_readcount_of(k);
            CHECK(same == exp.size());
        }
        CHECK(m.unique_kmers["chr1"][0]->get_coverage() == 30 && m.unique_kmers["chr1"][1]->get_coverage() == 34);
        m.runtimes = want.runtimes;
        m.sampling_runtimes = want.sampling_runtimes;
        CHECK(serialize_unique_kmers_map(m) == read_file(g_golden_dir + "/region_UniqueKmersList.cereal"));
    });
    run("k-mer counters on the reference's known answers (tests/KmerCounterTest.cpp:10-32)", [] {
        const std::string reads = "/tmp/pg_kc_reads.fa", kmerfile = "/tmp/pg_kc_kmerfile.fa";   // tests/data/reads.fa, kmerfile.fa
        { FILE* f = std::fopen(reads.c_str(), "w"); std::fputs(">read1\nATGCTGTAAAAAAACGGC\n", f); std::fclose(f); }
        { FILE* f = std::fopen(kmerfile.c_str(), "w"); std::fputs(">kmers\nATGCTGTAAAA\n", f); std::fclose(f); }
        const std::string read = "ATGCTGTAAAAAAACGGC";
        ExactKmerCounter all(reads, 10);
        for (size_t i = 0; i + 10 <= read.size(); ++i) CHECK(all.getKmerAbundance(read.substr(i, 10)) == 1);
        // only the k-mers of kmerfile.fa are counted; the others answer 0 when the counter is asked to behave like Jellyfish's
        TargetedKmerCounter graph_only(10, true);
        CHECK(graph_only.add_targets_from_sequences(kmerfile) == 2);
        graph_only.count(reads, 1);
        CHECK(graph_only.getKmerAbundance("ATGCTGTAAA") == 1 && graph_only.getKmerAbundance("TGCTGTAAAA") == 1 && graph_only.targets() == 2);
        const std::string others = "GCTGTAAAAAAACGGC";
        for (size_t i = 0; i + 10 <= others.size(); ++i) CHECK(graph_only.getKmerAbundance(others.substr(i, 10)) == 0);
        CHECK(graph_only.abundance_histogram(5) == std::vector<size_t>({0, 2, 0, 0, 0, 0}));
        TargetedKmerCounter strict(10);
        strict.add_targets_from_sequences(kmerfile);
        strict.count(reads, 1);
        CHECK(strict.getKmerAbundance("TTTACAGCAT") == 1);   // (the reverse complement of the first one)
        CHECK_THROWS(strict.getKmerAbundance("GCTGTAAAAA"));
    });
    run("TargetedKmerCounter with a large target set: table built and filled by worker threads = the exact counts", [] {
        // 400 k windows of a pseudo-random sequence registered (more than the 2^18 at which the table is built in parallel),
        // reads = 6000 pieces of it, either strand, a few wrong letters and an N now and then
        std::string graph;
        uint64_t x = 0xD1B54A32D192ED03ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        for (int i = 0; i < 400030; ++i) graph += "ACGT"[rnd() & 3];
        const std::string fa = "/tmp/pg_test_big_graph.fa", reads = "/tmp/pg_test_big_reads.fa";
        { std::FILE* f = std::fopen(fa.c_str(), "w"); std::fprintf(f, ">g\n%s\n", graph.c_str()); std::fclose(f); }
        {
            std::FILE* f = std::fopen(reads.c_str(), "w");
            for (int r = 0; r < 6000; ++r) {
                std::string piece = graph.substr(rnd() % (graph.size() - 200), 100 + rnd() % 100);
                if (r % 2) { std::reverse(piece.begin(), piece.end()); for (char& c : piece) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
                if (r % 7 == 0) piece[rnd() % piece.size()] = "ACGT"[rnd() & 3];
                if (r % 31 == 0) piece[rnd() % piece.size()] = 'N';
                std::fprintf(f, ">r%d\n%s\n", r, piece.c_str());
            }
            std::fclose(f);
        }
        ExactKmerCounter exact(reads, 31);
        TargetedKmerCounter one(31), four(31);
        CHECK(one.add_targets_from_sequences(fa) == 400000 && four.add_targets_from_sequences(fa) == 400000);
        one.count(reads, 1);
        four.count(reads, 4);
        CHECK(one.targets() == four.targets() && one.targets() > 399000 && one.kmers_seen() == four.kmers_seen());
        size_t differ = 0, seen = 0;
        for (size_t i = 0; i + 31 <= graph.size(); i += 3) {
            const std::string kmer = graph.substr(i, 31);
            const size_t want = exact.getKmerAbundance(kmer);
            differ += one.getKmerAbundance(kmer) != want || four.getKmerAbundance(kmer) != want;
            seen += want > 0;
        }
        CHECK(differ == 0 && seen > 50000);
        CHECK(one.abundance_histogram(100) == four.abundance_histogram(100));
        // the same reads as FASTA with the sequence over several lines, and as FASTQ whose quality lines begin with '@', '+' and
        // '>' now and then: the same counts
        const std::string fa_lines = "/tmp/pg_test_big_reads_lines.fa", fq = "/tmp/pg_test_big_reads.fq";
        {
            std::FILE* in = std::fopen(reads.c_str(), "r");
            std::FILE* a = std::fopen(fa_lines.c_str(), "w");
            std::FILE* q = std::fopen(fq.c_str(), "w");
            char line[512];
            int r = 0;
            while (std::fgets(line, sizeof line, in)) {
                if (line[0] == '>') continue;
                std::string seq(line);
                while (!seq.empty() && (seq.back() == '\n' || seq.back() == '\r')) seq.pop_back();
                std::fprintf(a, ">read%d some text\n", r);
                for (size_t i = 0; i < seq.size(); i += 37 + r % 5) std::fprintf(a, "%s\n", seq.substr(i, 37 + r % 5).c_str());
                std::string quality(seq.size(), 'I');
                if (r % 3 == 0) quality[0] = '@';
                if (r % 3 == 1) quality[0] = '+';
                if (r % 7 == 2) quality[0] = '>';
                std::fprintf(q, "@read%d/1\n%s\n+%s\n%s\n", r, seq.c_str(), r % 2 ? "read" : "", quality.c_str());
                r += 1;
            }
            std::fclose(in); std::fclose(a); std::fclose(q);
        }
        for (const std::string& path : {fa_lines, fq}) {
            TargetedKmerCounter other(31);
            other.add_targets_from_sequences(fa);
            other.count(path, 3);
            CHECK(other.kmers_seen() == one.kmers_seen());
            size_t wrong = 0;
            for (size_t i = 0; i + 31 <= graph.size(); i += 3) wrong += other.getKmerAbundance(graph.substr(i, 31)) != one.getKmerAbundance(graph.substr(i, 31));
            CHECK(wrong == 0);
        }
    });
    run("ExactKmerCounter: canonical counts, FASTA and FASTQ, letters outside ACGT", [] {
        const std::string fa = "/tmp/pg_test_reads.fa", fq = "/tmp/pg_test_reads.fq";
        { FILE* f = std::fopen(fa.c_str(), "w"); std::fputs(">r1\nACGTAC\nGT\n>r2\nACGNACGTA\n", f); std::fclose(f); }
        { FILE* f = std::fopen(fq.c_str(), "w"); std::fputs("@r1\nACGTACGT\n+\n@@@@@@@@\n@r2\nACGNACGTA\n+r2\n>>>>>>>>>\n", f); std::fclose(f); }
        for (const std::string& path : {fa, fq}) {
            ExactKmerCounter c(path, 4);
            // r1 = ACGTACGT: ACGT (its own reverse complement) x2, CGTA / TACG (one canonical class) x2, GTAC (palindrome) x1;
            // r2 = ACG N ACGTA: ACGT x1, CGTA x1
            CHECK(c.getKmerAbundance("ACGT") == 3);
            CHECK(c.getKmerAbundance("CGTA") == 3 && c.getKmerAbundance("TACG") == 3);
            CHECK(c.getKmerAbundance("GTAC") == 1);
            CHECK(c.getKmerAbundance("AAAA") == 0 && c.getKmerAbundance("ACGN") == 0);
            bool threw = false;
            try { c.getKmerAbundance("ACG"); } catch (const std::runtime_error&) { threw = true; }
            CHECK(threw);
        }
        std::vector<std::string> fl = {"ACGT", "CGTA", "GTAC", "AAAA"};
        ExactKmerCounter c(fa, 4);
        CHECK(compute_local_coverage(fl, c, 3) == 1);   // counts 3, 3, 1, 0 within [0, 12]: (3 + 3 + 1 + 0) / 4 in integers
        CHECK(compute_local_coverage(fl, c, 8) == 3);   // [2, 32]: (3 + 3) / 2
        CHECK(compute_local_coverage(fl, c, 100) == 100);  // none within [25, 400]: the given coverage
    });
    run("TargetedKmerCounter: the index's k-mers only, any read set size; same counts as the exact counter", [] {
        // the reference's fixture again: every unique and flanking k-mer of the table registered, the reads streamed
        UniqueKmersMap m = load_unique_kmers_map(g_golden_dir + "/index_UniqueKmersMap.cereal");
        const std::string table = g_golden_dir + "/index_chr1_kmers.tsv.gz", reads = g_golden_dir + "/region-reads.fa";
        ExactKmerCounter exact(reads, m.kmersize);
        // a gzipped copy of the reads and a FASTA rendering of them (the counter reads all three forms)
        const std::string gz = "/tmp/pg_test_region_reads.fq.gz", fq = "/tmp/pg_test_region_reads_as.fa";
        {
            const std::vector<unsigned char> raw = read_file(reads);
            gzFile out = gzopen(gz.c_str(), "wb");
            CHECK(out != nullptr);
            gzwrite(out, raw.data(), (unsigned)raw.size());
            gzclose(out);
            // the fixture is FASTQ (four lines per read): a FASTA rendering of it, every sequence broken over two lines
            std::FILE* f = std::fopen(fq.c_str(), "w");
            std::string text(raw.begin(), raw.end());
            size_t at = 0, lineno = 0;
            while (at < text.size()) {
                const size_t nl = text.find('\n', at);
                const std::string line = text.substr(at, nl == std::string::npos ? std::string::npos : nl - at);
                if (lineno % 4 == 1) std::fprintf(f, ">r%zu\n%s\n%s\n", lineno / 4, line.substr(0, line.size() / 2).c_str(), line.substr(line.size() / 2).c_str());
                lineno += 1;
                if (nl == std::string::npos) break;
                at = nl + 1;
            }
            std::fclose(f);
        }
        std::vector<std::string> all;   // the table's k-mers
        {
            gzFile t = gzopen(table.c_str(), "rb");
            char buf[1 << 16]; std::string line;
            while (gzgets(t, buf, sizeof buf)) {
                line += buf;
                if (line.empty() || line.back() != '\n') continue;
                line.pop_back();
                std::string chrom; size_t start = 0; std::vector<std::string> km, fl; bool header = false;
                parse_kmer_line(line, chrom, start, km, fl, header);
                all.insert(all.end(), km.begin(), km.end());
                all.insert(all.end(), fl.begin(), fl.end());
                line.clear();
            }
            gzclose(t);
        }
        CHECK(all.size() > 100);
        for (const std::string& path : {reads, gz, fq}) {
            for (unsigned threads : {1u, 4u}) {
                TargetedKmerCounter c(m.kmersize);
                CHECK(c.add_targets_from_table(table) == 2);
                c.count(path, threads);
                CHECK(c.targets() > 100 && c.targets() <= all.size() && c.kmers_seen() > 1000);
                size_t same = 0, nonzero = 0;
                for (const std::string& k : all) { const size_t n = c.getKmerAbundance(k); same += n == exact.getKmerAbundance(k); nonzero += n > 0; }
                CHECK(same == all.size() && nonzero > 50);
            }
        }
        // the pinned fixture through the targeted counter: the reference's counted archive byte for byte
        {
            const UniqueKmersMap want = load_unique_kmers_map(g_golden_dir + "/region_UniqueKmersList.cereal");
            TargetedKmerCounter c(m.kmersize);
            c.add_targets_from_table(table);
            c.count(gz, 3);
            fill_read_kmercounts("chr1", &m, c, table, 18);
            m.runtimes = want.runtimes;
            m.sampling_runtimes = want.sampling_runtimes;
            CHECK(serialize_unique_kmers_map(m) == read_file(g_golden_dir + "/region_UniqueKmersList.cereal"));
        }
        // counts of two files add up; a k-mer that was not registered is refused, not answered with 0; no targets after counting
        {
            TargetedKmerCounter c(4);
            c.add_target("ACGT"); c.add_target("TACG"); c.add_target("ACGN");
            const std::string fa = "/tmp/pg_test_reads_t.fa";
            { FILE* f = std::fopen(fa.c_str(), "w"); std::fputs(">r1\nACGTAC\nGT\n>r2\nACGNACGTA", f); std::fclose(f); }   // (no final newline)
            c.count(fa);
            CHECK(c.getKmerAbundance("ACGT") == 3 && c.getKmerAbundance("CGTA") == 3 && c.getKmerAbundance("ACGN") == 0);
            c.count(fa, 2);
            CHECK(c.getKmerAbundance("ACGT") == 6 && c.kmers_seen() == 2 * 7);
            bool threw = false;
            try { c.getKmerAbundance("GTAC"); } catch (const std::runtime_error&) { threw = true; }
            CHECK(threw);
            threw = false;
            try { c.add_target("GTAC"); } catch (const std::runtime_error&) { threw = true; }
            CHECK(threw);
            threw = false;
            try { c.count("/tmp/pg_no_such_reads.fa"); } catch (const std::runtime_error&) { threw = true; }
            CHECK(threw);
        }
    });
    run("parse_kmer_line: columns, lists, headers, malformed rows", [] {
        std::string chrom; size_t start = 0; std::vector<std::string> km, fl; bool header = false;
        parse_kmer_line("chr7\t1234\tx\tACGT,CCCC,GGGT\tAAAA", chrom, start, km, fl, header);
        CHECK(chrom == "chr7" && start == 1234 && !header);
        CHECK(km == std::vector<std::string>({"ACGT", "CCCC", "GGGT"}) && fl == std::vector<std::string>({"AAAA"}));
        km.clear(); fl.clear();
        // tests/KmerParser.cpp:9-94 (the lists are appended to, as the reference's are: fresh ones per row)
        parse_kmer_line("chr1\t1\t2\tnan\tnan", chrom, start, km, fl, header);
        CHECK(!header && km.empty() && fl.empty() && chrom == "chr1" && start == 1);
        parse_kmer_line("chr1\t1\t2\tnan\tAAAT,TGGG", chrom, start, km, fl, header);
        CHECK(km.empty() && fl == std::vector<std::string>({"AAAT", "TGGG"}) && chrom == "chr1" && start == 1);
        km.clear(); fl.clear();
        parse_kmer_line("chr1\t1\t2\tTGTG,ATGT\tnan", chrom, start, km, fl, header);
        CHECK(km == std::vector<std::string>({"TGTG", "ATGT"}) && fl.empty());
        km.clear(); fl.clear();
        parse_kmer_line("chr1\t1\t2\tTGTG,ATGT\tTTTT,GGGG", chrom, start, km, fl, header);
        CHECK(km == std::vector<std::string>({"TGTG", "ATGT"}) && fl == std::vector<std::string>({"TTTT", "GGGG"}) && chrom == "chr1" && start == 1);
        km.clear(); fl.clear();
        parse_kmer_line("chr7\t99\tx\tnan\tnan", chrom, start, km, fl, header);
        CHECK(start == 99 && km.empty() && fl.empty() && !header);
        parse_kmer_line("chr7\t5\tx\tA,,C,\tnan\t", chrom, start, km, fl, header);  // empty item kept, trailing comma / tab ignored
        CHECK(km == std::vector<std::string>({"A", "", "C"}) && fl.empty());
        km.clear();
        parse_kmer_line("#chromosome\tstart\tend\tunique_kmers\tunique_kmers_overhang", chrom, start, km, fl, header);
        CHECK(header && km.empty() && chrom == "chr7");
        for (const char* bad : {"chr7\t5\tx\tA", "chr7\t5\tx\tA\tB\tC", ""}) {
            bool threw = false;
            try { parse_kmer_line(bad, chrom, start, km, fl, header); } catch (const std::runtime_error&) { threw = true; }
            CHECK(threw);
        }
    });
}

// What the reference's run_genotype_command does with a PanGenie-index prefix (src/commands.cpp:738-1050), on its own
// fixture tests/data/index_* + region-reads.fa (tests/CommandsTest.cpp:18-37): index + graph archives, read k-mer counts
// into the index, the HMM of every chromosome (on the device), normalised results, VCF lines.
static std::vector<std::string> genotype_index_fixture(std::vector<GenotypingResult>* results_out = nullptr) {
    UniqueKmersMap m = load_unique_kmers_map(g_golden_dir + "/index_UniqueKmersMap.cereal");
    Graph graph = Graph::load(g_golden_dir + "/index_chr1_Graph.cereal");
    ExactKmerCounter counts(g_golden_dir + "/region-reads.fa", m.kmersize);
    const size_t kmer_abundance_peak = 18;   // tests/CommandsTest.cpp:56
    fill_read_kmercounts("chr1", &m, counts, g_golden_dir + "/index_chr1_kmers.tsv.gz", kmer_abundance_peak);
    ProbabilityTable probs(kmer_abundance_peak / 4, kmer_abundance_peak * 4, 2 * kmer_abundance_peak, 0.01L);
    HMM hmm(&m.unique_kmers["chr1"], &probs, true, false, 1.26, false, 0.00001L, nullptr, false);   // src/commands.cpp:160
    std::vector<GenotypingResult> results = hmm.move_genotyping_result();
    for (auto& r : results) r.normalize();                                                           // :981-987
    if (results_out) *results_out = results;
    std::vector<std::string> lines = Graph::genotypes_header("sample");
    for (const std::string& l : graph.genotypes_records(results)) lines.push_back(l);
    return lines;
}

static void graph_cpu_tests() {
    run("Graph archive: the reference's fixture parses and re-serialises byte for byte", [] {
        // tests/data/index_chr1_Graph.cereal: what PanGenie-index wrote for tests/data/region.vcf + region.fa (k = 31, the
        // reference added as a path) and run_genotype_command reads back (tests/CommandsTest.cpp:18-37)
        const std::vector<unsigned char> raw = read_file(g_golden_dir + "/index_chr1_Graph.cereal");
        CHECK(raw.size() == 3399);
        Graph g = Graph::parse(raw);
        CHECK(g.get_chromosome() == "chr1" && g.get_kmer_size() == 31 && g.reference_added() && g.size() == 2);
        CHECK(g.serialize() == raw);
        const Variant& a = g.get_variant(0);
        const Variant& b = g.get_variant(1);
        CHECK(a.get_start_position() == 138 && b.get_start_position() == 207 && a.get_end_position() == 139 && b.get_end_position() == 209);
        CHECK(a.nr_of_alleles() == 44 && b.nr_of_alleles() == 45 && a.nr_of_paths() == 215 && !a.is_combined() && !b.is_combined());
        // tests/data/region.vcf: chr1 139 T C and chr1 208 TG CG,CA; the bubble alleles carry 30 bases of flank either side
        const std::string ref = g.reference("chr1");
        CHECK(ref.size() == 301);
        CHECK(a.get_allele_string(0) == ref.substr(138 - 30, 30 + 1 + 30));
        CHECK(a.get_allele_string(0).substr(30, 1) == "T" && a.get_allele_string(1).substr(30, 1) == "C");
        CHECK(b.get_allele_string(0).substr(30, 2) == "TG" && b.get_allele_string(1).substr(30, 2) == "CG" && b.get_allele_string(2).substr(30, 2) == "CA");
        CHECK(!a.is_undefined_allele(0) && !a.is_undefined_allele(1) && a.is_undefined_allele(2) && a.is_undefined_allele(43));
        CHECK(g.variant_ids().size() == 2 && g.variant_ids()[0].size() == 1 && g.variant_ids()[1].size() == 2);
        bool threw = false;
        try { std::vector<unsigned char> cut(raw.begin(), raw.begin() + 2000); Graph::parse(cut); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    });
    run("Graph::phasing_records: GT:KC of the Viterbi haplotypes (src/graph.cpp:280-412)", [] {
        Graph g = Graph::load(g_golden_dir + "/index_chr1_Graph.cereal");   // chr1 139 T C; chr1 208 TG CG,CA; bubble alleles 2.. / 3.. undefined
        std::vector<GenotypingResult> ph(2);
        ph[0].add_first_haplotype_allele(1); ph[0].add_second_haplotype_allele(0); ph[0].set_coverage(30); ph[0].set_unique_kmers(62);
        ph[1].add_first_haplotype_allele(2); ph[1].add_second_haplotype_allele(7); ph[1].set_coverage(34); ph[1].set_unique_kmers(0);
        const std::vector<std::string> lines = g.phasing_records(ph);
        CHECK(lines.size() == 2);
        auto last = [](const std::string& l) { return l.substr(l.rfind('\t') + 1); };
        auto format = [](const std::string& l) { const size_t e = l.rfind('\t'), b = l.rfind('\t', e - 1); return l.substr(b + 1, e - b - 1); };
        CHECK(format(lines[0]) == "GT:KC" && format(lines[1]) == "GT:KC");
        CHECK(lines[0].rfind("chr1\t139\t.\tT\tC\t.\tPASS\tAF=", 0) == 0 && lines[0].find(";UK=62;MA=42;ID=") != std::string::npos);
        // a phasing-only result holds no likelihoods: among the DEFINED alleles the haplotypes come out as 0 (the reference
        // maps them inside get_specific_likelihoods' loop over the stored genotypes, src/genotypingresult.cpp:83-91)
        CHECK(last(lines[0]) == "0|0:30");
        CHECK(last(lines[1]) == "0|.:34");              // allele 7 of the second record is undefined sequence
        CHECK(last(g.phasing_records(ph, true)[1]) == "./.:34");   // ignore_imputed and no unique k-mers
        // with likelihoods stored (a run with genotyping and phasing) the haplotypes are mapped onto the defined alleles
        ph[0].add_to_likelihood(0, 1, 0.75L); ph[0].add_to_likelihood(1, 1, 0.25L);
        CHECK(last(g.phasing_records(ph)[0]) == "1|0:30");
        const std::vector<std::string> head = Graph::phasing_header("sample", "20230821");
        CHECK(head.size() == 10 && head[1] == "##fileDate=20230821" && head[8] == "##FORMAT=<ID=KC,Number=1,Type=Float,Description=\"Local kmer coverage.\">");
        bool threw = false;
        try { ph.pop_back(); g.phasing_records(ph); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    });
    run("Variant: a combined bubble back into its records (tests/VariantTest.cpp:170-241)", [] {
        // three records of chr2 merged into one bubble: A>T at 4, GAG>ACC at 7, G>GTC at 13; paths (0,0,0) (0,0,0) (1,1,1) (1,1,0)
        Variant v = Variant::from_parts("chr2", 4, "ATGA", "GGAA", {{"A", "T"}, {"GAG", "ACC"}, {"G", "GTC"}}, {"CT", "ACT"},
                                        {{0, 0, 0}, {1, 1, 0}, {1, 1, 1}}, {0, 0, 2, 1}, false);
        CHECK(v.is_combined() && v.nr_of_alleles() == 3 && v.get_end_position() == 14);
        CHECK(v.get_allele_string(0) == "ACTGAGACTG" && v.get_allele_string(2) == "TCTACCACTGTC");
        GenotypingResult g;
        g.add_to_likelihood(0, 0, 0.05); g.add_to_likelihood(0, 1, 0.05); g.add_to_likelihood(1, 1, 0.0);
        g.add_to_likelihood(0, 2, 0.3); g.add_to_likelihood(1, 2, 0.5); g.add_to_likelihood(2, 2, 0.1);
        g.add_first_haplotype_allele(0); g.add_second_haplotype_allele(2);
        g.set_coverage(7); g.set_unique_kmers(14);
        std::vector<VcfSite> sites = v.records(&g);
        CHECK(sites.size() == 3);
        const size_t starts[3] = {4, 7, 13};
        const std::vector<std::vector<std::string>> alleles = {{"A", "T"}, {"GAG", "ACC"}, {"G", "GTC"}};
        const std::vector<std::vector<unsigned short>> paths = {{0, 0, 1, 1}, {0, 0, 1, 1}, {0, 0, 1, 0}};
        const double expected[3][3] = {{0.05, 0.35, 0.6}, {0.05, 0.35, 0.6}, {0.1, 0.8, 0.1}};
        for (size_t i = 0; i < 3 && i < sites.size(); ++i) {
            CHECK(sites[i].chromosome == "chr2" && sites[i].start == starts[i] && sites[i].alleles == alleles[i] && sites[i].paths == paths[i]);
            std::vector<long double> got = sites[i].likelihoods.get_all_likelihoods(2);
            CHECK(got.size() == 3 && close_all({(double)got[0], (double)got[1], (double)got[2]}, {expected[i][0], expected[i][1], expected[i][2]}));
            CHECK(sites[i].likelihoods.get_haplotype() == std::make_pair((unsigned short)0, (unsigned short)1));
            CHECK(sites[i].likelihoods.coverage() == 7 && sites[i].likelihoods.nr_unique_kmers() == 14);
        }
        CHECK(v.records(nullptr).size() == 3 && v.records(nullptr)[1].likelihoods.contains_no_likelihoods());
    });
    run("Graph::genotypes_records: the text of the records (src/graph.cpp:165-277)", [] {
        Variant merged = Variant::from_parts("chr2", 4, "ATGA", "GGAA", {{"A", "T"}, {"GAG", "ACC"}, {"G", "GTC"}}, {"CT", "ACT"},
                                             {{0, 0, 0}, {1, 1, 0}, {1, 1, 1}}, {0, 0, 2, 1}, true);
        // a single record with an allele no path carries and an undefined one: dropped from ALT, counted in MA
        Variant single = Variant::from_parts("chr2", 99, "AAAA", "CCCC", {{"C", "G", "CNN", "T"}}, {}, {{0}, {1}, {2}, {3}}, {0, 1, 1, 3}, true);
        Graph graph = Graph::from_parts("chr2", 4, false, {merged, single}, {{"id-T"}, {}, {"id-GTC"}, {"id-G", "id-T2"}});
        GenotypingResult g;
        g.add_to_likelihood(0, 0, 0.05L); g.add_to_likelihood(0, 1, 0.05L); g.add_to_likelihood(1, 1, 0.0L);
        g.add_to_likelihood(0, 2, 0.3L); g.add_to_likelihood(1, 2, 0.5L); g.add_to_likelihood(2, 2, 0.1L);
        g.set_coverage(7); g.set_unique_kmers(14);
        GenotypingResult empty;   // no likelihoods: 0/0 with probability 1 (src/graph.cpp:225-227)
        empty.set_coverage(3);
        std::vector<std::string> lines = graph.genotypes_records({g, empty});
        CHECK(lines.size() == 4);
        auto fields = [](const std::string& l) { std::vector<std::string> f; std::string t; std::istringstream is(l); while (std::getline(is, t, '\t')) f.push_back(t); return f; };
        const std::vector<std::vector<std::string>> want = {
            {"chr2", "5", ".", "A", "T", ".", "PASS", "AF=0.5;UK=14;MA=0;ID=id-T", "GT:GQ:GL:KC"},
            {"chr2", "8", ".", "GAG", "ACC", ".", "PASS", "AF=0.5;UK=14;MA=0", "GT:GQ:GL:KC"},
            {"chr2", "14", ".", "G", "GTC", ".", "PASS", "AF=0.25;UK=14;MA=0;ID=id-GTC", "GT:GQ:GL:KC"},
            {"chr2", "100", ".", "C", "G,T", ".", "PASS", "AF=0.5,0.25;UK=0;MA=1;ID=id-G,id-T2", "GT:GQ:GL:KC"}};
        for (size_t i = 0; i < 4 && i < lines.size(); ++i) {
            std::vector<std::string> f = fields(lines[i]);
            CHECK(f.size() == 10);
            for (size_t k = 0; k < 9 && k < f.size(); ++k) { if (f[k] != want[i][k]) std::printf("  record %zu column %zu: %s\n", i, k, f[k].c_str()); CHECK(f[k] == want[i][k]); }
        }
        // the sample columns: record 0 = likelihoods (0.05, 0.35, 0.6) -> 1/1; the empty result -> 0/0 with GL 0,-inf,...
        CHECK(fields(lines[0])[9] == "1/1:3:-1.301,-0.4559,-0.2218:7");
        CHECK(fields(lines[2])[9] == "0/1:6:-1,-0.09691,-1:7");
        CHECK(fields(lines[3])[9] == "0/0:10000:0,-inf,-inf,-inf,-inf,-inf:3");
        std::vector<std::string> header = Graph::genotypes_header("HG0", "20250226");
        CHECK(header.size() == 12 && header[0] == "##fileformat=VCFv4.2" && header[1] == "##fileDate=20250226");
        CHECK(header[11] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tHG0");
        bool threw = false;
        try { graph.genotypes_records({g}); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    });
}

static void archive_cpu_tests() {
    run("archives: every truncation and byte corruption parses or throws std::runtime_error (never crashes)", [] {
        // the three readers (index map, `-w` results, graph) on the reference's fixtures and our own archive: cut at every
        // length (step 1 near the ends, coarser in the middle), and with single bytes overwritten — a length field turned
        // into 2^64-ish, a shared-pointer id into nonsense.  What counts is that nothing but runtime_error comes out
        // (bad_alloc / length_error / out-of-range reads would be the bug); the sanitizer build of this test checks the rest.
        struct Kind { const char* name; std::vector<unsigned char> bytes; std::function<void(const std::vector<unsigned char>&)> parse; };
        std::vector<Kind> kinds;
        kinds.push_back({"index", read_file(g_golden_dir + "/index_UniqueKmersMap.cereal"), [](const std::vector<unsigned char>& b) { (void)parse_unique_kmers_map(b); }});
        kinds.push_back({"results", serialize_results(sample_results()), [](const std::vector<unsigned char>& b) { (void)parse_results(b); }});
        kinds.push_back({"graph", read_file(g_golden_dir + "/index_chr1_Graph.cereal"), [](const std::vector<unsigned char>& b) { (void)Graph::parse(b); }});
        for (Kind& k : kinds) {
            size_t other = 0, threw = 0, parsed = 0;
            auto attempt = [&](const std::vector<unsigned char>& b) {
                try { k.parse(b); parsed += 1; }
                catch (const std::runtime_error&) { threw += 1; }
                catch (...) { other += 1; }
            };
            const size_t n = k.bytes.size();
            CHECK(n > 64);
            for (size_t cut = 0; cut < n; cut += (cut < 256 || cut + 256 > n) ? 1 : 97) attempt(std::vector<unsigned char>(k.bytes.begin(), k.bytes.begin() + (long)cut));
            const size_t truncations_parsed = parsed;
            uint64_t x = 88172645463325252ull;   // xorshift: positions and values of the corruptions
            for (int it = 0; it < 600; ++it) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                std::vector<unsigned char> b = k.bytes;
                const size_t at = (size_t)(x % n);
                const unsigned char vals[4] = {0xFF, 0x00, 0x80, (unsigned char)(x >> 40)};
                b[at] = vals[(x >> 33) & 3];
                if ((x >> 36) & 1) for (size_t q = at; q < at + 8 && q < n; ++q) b[q] = 0xFF;   // a whole 64-bit field
                attempt(b);
            }
            CHECK(other == 0);
            CHECK(threw > 100);                   // (cuts almost always fail; corruptions of payload bytes parse fine)
            CHECK(truncations_parsed <= 2);       // only (nearly) complete prefixes can parse
            (void)k.name;
        }
    });
    run("cereal binary archive: Results (`-w`, what PanGenie-vcf reads) layout and round trip", [] {
        // the smallest case byte by byte: one chromosome "c", one result with one likelihood 0.5 of genotype (0, 1)
        Results one;
        GenotypingResult g;
        g.add_to_likelihood(0, 1, 0.5L); g.add_first_haplotype_allele(1); g.add_second_haplotype_allele(0); g.set_coverage(7); g.set_unique_kmers(9);
        one.result["c"] = {g};
        one.runtimes["c"] = 2.0;
        const std::vector<unsigned char> want = {
            1, 0, 0, 0, 0, 0, 0, 0,                                   // map size
            1, 0, 0, 0, 0, 0, 0, 0, 'c',                              // key
            1, 0, 0, 0, 0, 0, 0, 0,                                   // vector size
            1, 0, 0, 0, 0, 0, 0, 0,                                   // genotype_to_likelihood size
            0, 0, 1, 0,                                               // pair (0, 1)
            0, 0, 0, 0, 0, 0, 0, 0x80, 0xFE, 0x3F, 0, 0, 0, 0, 0, 0,  // 0.5L: mantissa 2^63, exponent 0x3FFE, 6 padding bytes
            1, 0, 0, 0, 7, 0, 9, 0,                                   // haplotype_1, haplotype_2, local_coverage, unique_kmers
            1, 0, 0, 0, 0, 0, 0, 0,                                   // runtimes size
            1, 0, 0, 0, 0, 0, 0, 0, 'c', 0, 0, 0, 0, 0, 0, 0, 0x40};  // "c" -> 2.0
        CHECK(serialize_results(one) == want);
        const Results r = sample_results();
        const std::vector<unsigned char> bytes = serialize_results(r);
        Results back = parse_results(bytes);
        CHECK(serialize_results(back) == bytes);
        CHECK(back.result.size() == 2 && back.result["chr1"].size() == 2 && back.result["chr10"].size() == 1);
        CHECK(back.result["chr1"][0].get_stored_likelihoods() == r.result.at("chr1")[0].get_stored_likelihoods());
        CHECK(back.result["chr1"][0].get_haplotype() == std::make_pair((unsigned short)1, (unsigned short)0));
        CHECK(back.result["chr1"][0].coverage() == 27 && back.result["chr1"][0].nr_unique_kmers() == 20);
        CHECK(back.result["chr1"][1].contains_no_likelihoods() && back.result["chr1"][1].coverage() == 3);
        CHECK(back.result["chr10"][0].get_genotype_likelihood(5, 2) == 1e-4000L && back.result["chr10"][0].nr_unique_kmers() == 301);
        CHECK(back.runtimes["chr10"] == 0.125);
        bool threw = false;
        try { std::vector<unsigned char> cut(bytes.begin(), bytes.end() - 3); parse_results(cut); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    });
    run("HMM::serialize (reference src/hmm.hpp:49-52: archive(genotyping_result)) = one chromosome's vector of the Results archive", [] {
        const Results r = sample_results();
        const std::vector<GenotypingResult>& chr1 = r.result.at("chr1");
        // the Results archive of {"chr1": v}: u64 1, u64 4 "chr1", <the vector>, u64 0 runtimes — the vector's bytes are the HMM's
        Results one;
        one.result["chr1"] = chr1;
        const std::vector<unsigned char> whole = serialize_results(one);
        const std::vector<unsigned char> vec(whole.begin() + 8 + 8 + 4, whole.end() - 8);
        HMM h = HMM::deserialize(vec);   // (no device needed)
        CHECK(h.serialize() == vec);
        const std::vector<GenotypingResult> got = h.get_genotyping_result();
        CHECK(got.size() == chr1.size());
        for (size_t i = 0; i < got.size() && i < chr1.size(); ++i) {
            CHECK(got[i].get_stored_likelihoods() == chr1[i].get_stored_likelihoods());
            CHECK(got[i].get_haplotype() == chr1[i].get_haplotype() && got[i].coverage() == chr1[i].coverage() && got[i].nr_unique_kmers() == chr1[i].nr_unique_kmers());
        }
        CHECK(HMM().serialize() == std::vector<unsigned char>(8, 0));   // an HMM that genotyped nothing: an empty vector
        bool threw = false;
        try { std::vector<unsigned char> cut(vec.begin(), vec.end() - 1); (void)HMM::deserialize(cut); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { std::vector<unsigned char> more = vec; more.push_back(0); (void)HMM::deserialize(more); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    });
    run("cereal binary archive: the reference's own fixtures parse and re-serialise byte for byte", [] {
        for (const char* name : {"region_UniqueKmersList.cereal", "region2_UniqueKmersList.cereal"}) {
            const std::string path = g_golden_dir + "/" + name;
            FILE* f = std::fopen(path.c_str(), "rb");
            CHECK(f != nullptr);
            if (!f) continue;
            std::vector<unsigned char> bytes;
            unsigned char buf[4096];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
            std::fclose(f);
            UniqueKmersMap m = parse_unique_kmers_map(bytes);
            CHECK(m.kmersize == 31 && m.unique_kmers.size() == 1 && m.unique_kmers.count("chr1") == 1 && m.add_reference);
            CHECK(m.unique_kmers["chr1"].size() == 2);
            CHECK(serialize_unique_kmers_map(m) == bytes);
        }
        // first record of the region fixture: position 138, coverage 30, 62 k-mers, 44 alleles, 215 paths
        UniqueKmersMap m = load_unique_kmers_map(g_golden_dir + "/region_UniqueKmersList.cereal");
        auto& u = *m.unique_kmers["chr1"][0];
        us ids;
        u.get_allele_ids(ids);
        CHECK(u.get_variant_position() == 138 && u.get_coverage() == 30 && u.size() == 62 && ids.size() == 44 && u.get_nr_paths() == 215);
        CHECK(u.kmers_on_allele(0) == 31 && u.kmers_on_allele(1) == 31 && u.kmers_on_allele(2) == 0 && u.is_undefined_allele(2) && !u.is_undefined_allele(1));
    });
    run("cereal binary archive: written objects read back identically", [] {
        UniqueKmersMap m;
        m.kmersize = 31; m.add_reference = true; m.runtimes["chrA"] = 0.25; m.sampling_runtimes["chrA"] = 0.5;
        auto a = bi(1000, {0, 1, 1}); kmer(a, 7, {0}); kmer(a, 9, {1}); a->set_coverage(21);
        auto b = multi(2000, {0, 2, 1, 1}); kmer(b, 3, {0}); kmer(b, 4, {1, 2}); kmer(b, 5, {2}); b->set_undefined_allele(2); b->set_coverage(18);
        m.unique_kmers["chrA"] = {a, b};
        m.unique_kmers["chrB"] = {a};   // the same object twice: a back reference in the archive
        const std::vector<unsigned char> bytes = serialize_unique_kmers_map(m);
        UniqueKmersMap r = parse_unique_kmers_map(bytes);
        CHECK(serialize_unique_kmers_map(r) == bytes);
        CHECK(r.unique_kmers["chrB"][0].get() == r.unique_kmers["chrA"][0].get());
        auto& rb = *r.unique_kmers["chrA"][1];
        CHECK(rb.get_variant_position() == 2000 && rb.get_coverage() == 18 && rb.size() == 3 && rb.is_undefined_allele(2));
        CHECK(rb.kmer_on_allele(1, 2) && rb.kmer_on_allele(1, 1) && !rb.kmer_on_allele(1, 0) && rb.get_allele(1) == 2);
        CHECK(r.runtimes["chrA"] == 0.25 && r.sampling_runtimes["chrA"] == 0.5);
    });
    run("VCF sample column (GT:GQ:GL:KC)", [] {
        GenotypingResult g;
        g.add_to_likelihood(0, 0, 0.1L); g.add_to_likelihood(0, 1, 0.8L); g.add_to_likelihood(1, 1, 0.1L);
        g.set_coverage(27);
        us defined = {0, 1};
        CHECK(genotype_field(g, defined, 2) == "0/1:6:-1,-0.09691,-1:27");   // GQ = (size_t)(-10 log10(0.2)) = 6
        GenotypingResult sure;
        sure.add_to_likelihood(1, 1, 1.0L); sure.add_to_likelihood(0, 0, 0.0L); sure.add_to_likelihood(0, 1, 0.0L);
        CHECK(genotype_field(sure, defined, 2) == "1/1:10000:-inf,-inf,0:0");
        GenotypingResult empty;   // no likelihoods: 0/0 with certainty (reference src/graph.cpp:225-227)
        CHECK(genotype_field(empty, defined, 2) == "0/0:10000:0,-inf,-inf:0");
        GenotypingResult tie;
        tie.add_to_likelihood(0, 0, 0.5L); tie.add_to_likelihood(0, 1, 0.5L);
        CHECK(genotype_field(tie, defined, 2).substr(0, 4) == ".:.:");
        // an undefined allele (2) is dropped and the rest renormalised
        GenotypingResult three;
        three.add_to_likelihood(0, 0, 0.2L); three.add_to_likelihood(0, 1, 0.2L); three.add_to_likelihood(1, 1, 0.1L);
        three.add_to_likelihood(0, 2, 0.25L); three.add_to_likelihood(1, 2, 0.25L);
        CHECK(genotype_field(three, defined, 3) == ".:.:-0.3979,-0.3979,-0.699:0");
        // a likelihood sixteen long double steps below 1 keeps its logarithm (the last record of demo/test_genotyping.vcf:
        // 1/1:180:-55.05,-18.08,-3.767e-19) — the log10 is taken in long double
        GenotypingResult close;
        close.add_to_likelihood(0, 0, 8.959757403193811914e-56L); close.add_to_likelihood(0, 1, 8.3995489827273644067e-19L);
        close.add_to_likelihood(1, 1, 1.0L - 16.0L * 0x1p-64L);
        close.set_coverage(3);
        CHECK(genotype_field(close, defined, 2) == "1/1:180:-55.05,-18.08,-3.767e-19:3");
    });
}

// a panel with duplicated paths (exact ties).  The gaps
// between the variants differ: with EQUAL neighbouring gaps "switch here" and "switch one column later" are the same
// product taken in a different order, and which one wins is decided by the rounding of the last bit — in the
// reference as anywhere else.
static vector<shared_ptr<UniqueKmers>> viterbi_panel(size_t V, size_t H, size_t dup_from) {
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    vector<shared_ptr<UniqueKmers>> a;
    size_t position = 1000;
    for (size_t v = 0; v < V; ++v) {
        position += 60 + (size_t)(rnd() % 1300);
        us p1;
        for (size_t p = 0; p < H; ++p) p1.push_back((unsigned short)(rnd() % 3 == 0));
        p1[v % H] = 1;
        for (size_t p = dup_from; p < H; ++p) p1[p] = p1[p - dup_from];  // duplicated paths: exact ties
        auto ua = bi(position, p1);
        for (int q = 0; q < 4; ++q) {
            kmer(ua, (unsigned short)(rnd() % 30), {0});
            kmer(ua, (unsigned short)(rnd() % 30), {1});
        }
        ua->set_coverage(27);
        a.push_back(ua);
    }
    return a;
}

// ----------------------------------------------------------------------------------- sampler (CPU: host-side costs)
static void sampler_cpu_tests() {
    run("SampledPaths::mask_indexes / recombination", [] {
        SampledPaths s;
        s.sampled_paths = {{0, 0, 1, 1}, {1, 1, 2, 0}};
        CHECK((s.mask_indexes(1, 2) == vector<bool>{false, false, true}));
        CHECK((s.mask_indexes(2, 2) == vector<bool>{true, false, false}));
        CHECK_THROWS(s.mask_indexes(4, 2));
        CHECK_THROWS(s.mask_indexes(2, 1));
        CHECK(!s.recombination(0, 0) && !s.recombination(1, 0) && s.recombination(2, 0) && !s.recombination(3, 0));
        CHECK(!s.recombination(0, 1) && !s.recombination(1, 1) && s.recombination(2, 1) && s.recombination(3, 1));
    });
    run("SamplingEmissions get_emission_cost", [] {
        auto u1 = bi(2000, {0, 0});
        auto u2 = bi(3000, {1, 0});
        u2->set_undefined_allele(0);
        kmer(u2, 20, {1}); kmer(u2, 1, {1});
        SamplingEmissions s1(u1), s2(u2);
        CHECK(s1.get_emission_cost(0) == 0);
        CHECK(s2.get_emission_cost(0) == 50 && s2.get_emission_cost(1) == 3);
        auto v1 = bi(2000, {0, 1});
        kmer(v1, 20, {0}); kmer(v1, 10, {0}); kmer(v1, 1, {0}); kmer(v1, 3, {1});
        auto v2 = bi(3000, {0, 1});
        v2->set_undefined_allele(0);
        kmer(v2, 1, {0}); kmer(v2, 1, {0}); kmer(v2, 20, {1}); kmer(v2, 2, {1}); kmer(v2, 0, {1});
        SamplingEmissions t1(v1), t2(v2);
        CHECK(t1.get_emission_cost(0) == 1 && t1.get_emission_cost(1) == 0);
        CHECK(t2.get_emission_cost(0) == 50 && t2.get_emission_cost(1) == 4);
        auto w = bi(2000, {0, 1});
        kmer(w, 20, {0}); kmer(w, 1, {1});
        SamplingEmissions t3(w);
        CHECK(t3.get_emission_cost(0) == 0 && t3.get_emission_cost(1) == 25);
        auto m = multi(2000, {0, 1, 2});
        m->set_undefined_allele(1);
        kmer(m, 20, {0}); kmer(m, 2, {2});
        SamplingEmissions t4(m);
        CHECK(t4.get_emission_cost(0) == 0 && t4.get_emission_cost(1) == 50 && t4.get_emission_cost(2) == 25);
        t1.penalize(0, 10); CHECK(t1.get_emission_cost(0) == 11);
        t1.penalize(0, 10); t1.penalize(0, 10); CHECK(t1.get_emission_cost(0) == 25);
    });
    run("SamplingTransitions compute_transition_cost", [] {
        SamplingTransitions s(1000000, 2000000, 1.26, 5, 0.25);
        const double recomb_prob = 0.04455105238;
        const unsigned int expected_cost = -10.0 * log10(recomb_prob);
        CHECK(s.compute_transition_cost(false) == 0);
        CHECK(s.compute_transition_cost(true) == expected_cost);
    });
}

// ----------------------------------------------------------------------------------- sampler (GPU)
static vector<shared_ptr<UniqueKmers>> sampler_panel3(size_t pos2, bool extra_kmer) {
    auto u1 = multi(1000000, {0, 1, 2});
    kmer(u1, 10, {0}); kmer(u1, 10, {0}); kmer(u1, 7, {0});
    kmer(u1, 1, {1}); kmer(u1, 2, {1});
    if (extra_kmer) kmer(u1, 1, {1});
    kmer(u1, 20, {1});
    kmer(u1, 11, {2}); kmer(u1, 10, {2}); kmer(u1, 1, {2});
    u1->set_coverage(5);
    auto u2 = bi(pos2, {0, 1, 1});
    kmer(u2, 1, {0}); kmer(u2, 1, {0}); kmer(u2, 20, {1}); kmer(u2, 22, {1});
    u2->set_coverage(5);
    return {u1, u2};
}

static void sampler_gpu_tests() {
    run("HaplotypeSampler get_column_minima", [] {
        HaplotypeSampler h(nullptr, 0);
        size_t f, s2; unsigned int fv, sv;
        vector<unsigned int> col = {10, 2, 14, 1};
        vector<bool> mask = {true, true, true, true};
        h.get_column_minima(col, mask, f, s2, fv, sv);
        CHECK(f == 3 && s2 == 1 && fv == 1 && sv == 2);
        col = {10, 2, 14, 2};
        h.get_column_minima(col, mask, f, s2, fv, sv);
        CHECK(f == 1 && s2 == 3 && fv == 2 && sv == 2);
        col = {10, 10, 10, 10};
        h.get_column_minima(col, mask, f, s2, fv, sv);
        CHECK(f == 0 && s2 == 1 && fv == 10 && sv == 10);
        col = {10, 20, 30}; mask = {true, false, true};
        h.get_column_minima(col, mask, f, s2, fv, sv);
        CHECK(f == 0 && s2 == 2 && fv == 10 && sv == 30);
        mask = {false, true, true};
        h.get_column_minima(col, mask, f, s2, fv, sv);
        CHECK(f == 1 && s2 == 2 && fv == 20 && sv == 30);
    });
    run("HaplotypeSampler size 0 leaves the panel alone", [] {
        auto u1 = bi(2000, {0, 0});
        auto u2 = bi(3000, {1, 0});
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2};
        HaplotypeSampler h(&uks, 0);
        CHECK(h.get_sampled_paths().sampled_paths.empty() && u2->get_nr_paths() == 2);
    });
    run("HaplotypeSampler Viterbi", [] {
        auto u1 = bi(1000000, {0, 1});
        kmer(u1, 10, {0}); kmer(u1, 1, {1}); u1->set_coverage(5);
        auto u2 = bi(2000000, {1, 0});
        kmer(u2, 10, {0}); kmer(u2, 1, {0}); kmer(u2, 2, {1}); u2->set_coverage(5);
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2};
        vector<unsigned int> best;
        HaplotypeSampler h(&uks, 1, 1.26, 25000.0L, &best);
        CHECK(best.size() == 1 && best[0] == 6);
        auto sp = h.get_sampled_paths();
        CHECK(sp.sampled_paths.size() == 1 && (sp.sampled_paths[0] == vector<size_t>{0, 1}));
    });
    run("HaplotypeSampler Viterbi2 / Viterbi3", [] {
        auto uks = sampler_panel3(1000010, false);
        vector<unsigned int> best;
        HaplotypeSampler h(&uks, 2, 1.26, 25000.0L, &best);
        CHECK(best.size() == 2 && best[0] == 1 && best[1] == 14);
        auto sp = h.get_sampled_paths();
        CHECK((sp.sampled_paths[0] == vector<size_t>{2, 2}) && (sp.sampled_paths[1] == vector<size_t>{1, 1}));
        auto uks3 = sampler_panel3(2000000, true);
        best.clear();
        HaplotypeSampler h3(&uks3, 2, 1.26, 25000.0L, &best);
        CHECK(best[0] == 1 && best[1] == 14);
        sp = h3.get_sampled_paths();
        CHECK((sp.sampled_paths[0] == vector<size_t>{2, 2}) && (sp.sampled_paths[1] == vector<size_t>{0, 1}));
    });
    run("HaplotypeSampler update_unique_kmers (+ reference path)", [] {
        auto uks = sampler_panel3(2000000, true);
        HaplotypeSampler h(&uks, 2, 1.26, 25000.0L, nullptr);
        auto u1 = uks[0], u2 = uks[1];
        CHECK(u1->size() == 6);
        const us c1 = {10, 10, 7, 11, 10, 1};
        for (size_t i = 0; i < c1.size(); ++i) CHECK(u1->get_readcount_of(i) == c1[i]);
        for (size_t i = 0; i < 3; ++i) CHECK(u1->kmer_on_path(i + 3, 0) && u1->kmer_on_path(i, 1));
        CHECK(u2->size() == 2 && u2->get_readcount_of(0) == 20 && u2->get_readcount_of(1) == 22);
        for (size_t i = 0; i < 2; ++i) CHECK(u2->kmer_on_path(i, 0) && u2->kmer_on_path(i, 1));
        auto uksr = sampler_panel3(2000000, true);
        HaplotypeSampler hr(&uksr, 2, 1.26, 25000.0L, nullptr, true);
        auto sp = hr.get_sampled_paths();
        CHECK(sp.sampled_paths.size() == 3 && (sp.sampled_paths[2] == vector<size_t>{0, 0}));
        u1 = uksr[0]; u2 = uksr[1];
        CHECK(u1->size() == 6 && u2->size() == 4);
        for (size_t i = 0; i < 3; ++i) CHECK(u1->kmer_on_path(i + 3, 0) && u1->kmer_on_path(i, 1) && u1->kmer_on_path(i, 2));
        const us c2 = {1, 1, 20, 22};
        for (size_t i = 0; i < c2.size(); ++i) CHECK(u2->get_readcount_of(i) == c2[i]);
        for (size_t i = 0; i < 2; ++i) CHECK(u2->kmer_on_path(i + 2, 0) && u2->kmer_on_path(i + 2, 1) && u2->kmer_on_path(i, 2));
    });
    run("HaplotypeSampler then HMM on the sampled panel", [] {
        // 40 paths over 30 variants: sample 6 (+ reference), then genotype on what is left, as the reference's
        // prepare_unique_kmers + run_genotyping do (src/commands.cpp:148-151, :155-175)
        vector<shared_ptr<UniqueKmers>> uks;
        unsigned x = 12345;
        auto rnd = [&x]() { x = x * 1664525u + 1013904223u; return x >> 8; };
        for (size_t v = 0; v < 30; ++v) {
            us p2a(40);
            for (auto& a : p2a) a = rnd() % 2;
            p2a[0] = 0;
            auto u = bi(1000 + 700 * v, p2a);
            for (int k = 0; k < 4; ++k) kmer(u, rnd() % 30, {(unsigned short)(k % 2)});
            u->set_coverage(20);
            uks.push_back(u);
        }
        HaplotypeSampler h(&uks, 6, 1.26, 25000.0L, nullptr, true);
        for (auto& u : uks) CHECK(u->get_nr_paths() == 7);
        for (size_t v = 0; v < 30; ++v) {
            std::unordered_set<size_t> seen;
            for (size_t j = 0; j < 6; ++j) seen.insert(h.get_sampled_paths().sampled_paths[j][v]);
            CHECK(seen.size() == 6);  // a pass never re-picks a path an earlier pass holds at that column
        }
        ProbabilityTable probs(5, 41, 61, 0.0L);
        HMM hmm(&uks, &probs, true, false, 1.26, false, 25000.0L);
        auto res = hmm.get_genotyping_result();
        CHECK(res.size() == 30);
        for (auto& r : res) {
            const long double sum = r.get_genotype_likelihood(0, 0) + r.get_genotype_likelihood(0, 1) + r.get_genotype_likelihood(1, 1);
            CHECK(std::fabs((double)(sum - 1.0L)) < 1e-9);
        }
    });
}

static void gpu_tests() {
    run("device visible", [] { CHECK(HMM::device_count() >= 1); });
    run("TransitionProbabilityComputer", [] {
        TransitionProbabilityComputer t(1000000, 2000000, 1.26, 5, false, 0.25);
        const double q = 0.04455105238, p = q + 0.77724473806;
        CHECK(close(t.compute_transition_prob(0, 0, 0, 0), p * p) && close(t.compute_transition_prob(0, 0, 0, 1), p * q));
        CHECK(close(t.compute_transition_prob(1, 2, 2, 1), q * q) && close(t.compute_transition_prob(1, 3, 1, 1), p * q));
        CHECK(close(t.compute_transition_prob(0), p * p) && close(t.compute_transition_prob(2), q * q));
        TransitionProbabilityComputer t10(1000000, 2000000, 1.26, 10, false, 0.25);
        const double q10 = 0.01183851532, p10 = q10 + 0.88161484678;
        CHECK(close(t10.compute_transition_prob(1), p10 * q10));
        TransitionProbabilityComputer u(1, 2, 1.26, 5, true, 0.25);
        CHECK(u.compute_transition_prob(2) == 1.0L);
    });
    run("EmissionProbabilityComputer", [] {
        vector<us> alleles = {{0}, {0}, {1}, {1}, {1}};
        us counts = {4, 6, 8, 2, 5};
        vector<CopyNumber> cns = {CopyNumber(0.01, 0.2, 0.0), CopyNumber(0.001, 0.5, 0.001), CopyNumber(0.0, 0.3, 0.02), CopyNumber(0.05, 0.6, 0.0), CopyNumber(0.01, 0.2, 0.01)};
        ProbabilityTable probs(0, 10, 10, 0.0);
        auto u = multi(1000, {0, 1, 2});
        u->set_undefined_allele(2);
        for (size_t i = 0; i < counts.size(); ++i) { u->insert_kmer(counts[i], alleles[i]); probs.modify_probability(0, counts[i], cns[i]); }
        EmissionProbabilityComputer e(u, &probs);
        CHECK(close(e.get_emission_probability(0, 0), 0.0) && close(e.get_emission_probability(0, 1), 0.0036) && close(e.get_emission_probability(1, 0), 0.0036));
        CHECK(close(e.get_emission_probability(1, 1), 0.0) && close(e.get_emission_probability(0, 2), 0.000128225) && close(e.get_emission_probability(2, 1), 0.000132565));
        CHECK(close(e.get_emission_probability(2, 2), 0.000019852));
    });
    auto std_table = [] {
        ProbabilityTable probs(5, 10, 30, 0.0L);
        probs.modify_probability(5, 10, CopyNumber(0.1, 0.9, 0.1));
        probs.modify_probability(5, 20, CopyNumber(0.01, 0.01, 0.9));
        probs.modify_probability(5, 5, CopyNumber(0.9, 0.3, 0.1));
        return probs;
    };
    run("HMM get_genotyping_result / skip_reference_position", [&] {
        auto u1 = bi(2000, {0, 1}); kmer(u1, 10, {0}); kmer(u1, 10, {1}); u1->set_coverage(5);
        auto ref_only = bi(2500, {0, 0}); kmer(ref_only, 10, {0}); kmer(ref_only, 20, {1}); ref_only->set_coverage(22);
        auto u3 = bi(3000, {0, 1}); kmer(u3, 20, {0}); kmer(u3, 5, {1}); u3->set_coverage(5);
        ProbabilityTable probs = std_table();
        vector<shared_ptr<UniqueKmers>> two = {u1, u3}, three = {u1, ref_only, u3};
        HMM a(&two, &probs, true, false, R01, false, 0.25);
        CHECK(close_all(triples(a.get_genotyping_result()), {0.0509465435, 0.9483202731, 0.0007331832, 0.9678020017, 0.031003181, 0.0011948172}));
        HMM b(&three, &probs, true, false, R01, false, 0.25);
        auto res = b.get_genotyping_result();
        CHECK(close_all(triples(res), {0.0509465435, 0.9483202731, 0.0007331832, 0.0, 0.0, 0.0, 0.9678020017, 0.031003181, 0.0011948172}));
        CHECK(res[1].coverage() == 22 && res[0].coverage() == 5 && res[1].nr_unique_kmers() == 2 && res[1].contains_no_likelihoods());
    });
    run("HMM undefined alleles", [] {
        auto u1 = bi(2000, {0, 1}); u1->set_undefined_allele(0); kmer(u1, 10, {0});
        auto u2 = bi(3000, {1, 0}); kmer(u2, 20, {0}); kmer(u2, 1, {1});
        ProbabilityTable probs(0, 1, 21, 0.0L);
        probs.modify_probability(0, 10, CopyNumber(0.1, 0.9, 0.1));
        probs.modify_probability(0, 20, CopyNumber(0.01, 0.01, 0.9));
        probs.modify_probability(0, 1, CopyNumber(0.9, 0.3, 0.1));
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2};
        HMM hmm(&uks, &probs, true, false, R01, false, 0.25);
        auto res = hmm.get_genotyping_result();
        CHECK(close_all(triples(res), {0.02396597038, 0.52185641164, 0.45417761795, 0.97855858361, 0.01875778106, 0.00268363531}));
        us defined = {1};
        CHECK(close(res[0].get_specific_likelihoods(defined).get_genotype_likelihood(0, 0), 1.0));
    });
    run("HMM no unique kmers / uniform / no alt allele", [] {
        ProbabilityTable none;
        auto a1 = bi(2000, {0, 0, 1}), a2 = bi(3000, {0, 1, 1});
        vector<shared_ptr<UniqueKmers>> uks = {a1, a2};
        HMM hmm(&uks, &none, true, false, 1070.02483182, false, 0.25);
        CHECK(close_all(triples(hmm.get_genotyping_result()), {4.0 / 9, 4.0 / 9, 1.0 / 9, 1.0 / 9, 4.0 / 9, 4.0 / 9}));
        auto b1 = bi(2000, {0, 1, 1}), b2 = bi(3000, {0, 0, 1});
        vector<shared_ptr<UniqueKmers>> uks2 = {b1, b2};
        HMM uni(&uks2, &none, true, false, 1.26, true, 0.25);
        CHECK(close_all(triples(uni.get_genotyping_result()), {1 / 9.0, 4 / 9.0, 4 / 9.0, 4 / 9.0, 4 / 9.0, 1 / 9.0}));
        auto c = bi(2000, {0, 0, 0}); kmer(c, 10, {0, 1}); kmer(c, 5, {});
        ProbabilityTable probs(0, 1, 11, 0.0L);
        probs.modify_probability(0, 10, CopyNumber(0.1, 0.2, 0.9));
        probs.modify_probability(0, 5, CopyNumber(0.3, 0.4, 0.1));
        vector<shared_ptr<UniqueKmers>> uks3 = {c};
        HMM noalt(&uks3, &probs, true, false, 1.26, false, 0.25);
        CHECK(noalt.get_genotyping_result()[0].get_likeliest_genotype() == std::make_pair(-1, -1));
    });
    run("HMM emissions_zero / underflow", [] {
        auto u1 = bi(1000, {0, 1}); kmer(u1, 10, {0}); kmer(u1, 10, {1});
        auto u2 = bi(2000, {1, 1}); kmer(u2, 0, {1}); kmer(u2, 0, {1});
        auto u3 = bi(3000, {0, 1}); kmer(u3, 10, {0}); kmer(u3, 10, {1});
        ProbabilityTable probs(0, 1, 11, 0.0L);
        probs.modify_probability(0, 10, CopyNumber(0.0, 1.0, 0.0));
        probs.modify_probability(0, 0, CopyNumber(1.0, 0.0, 0.0));
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2, u3};
        HMM hmm(&uks, &probs, true, false, R01, false, 0.25);
        CHECK(close_all(triples(hmm.get_genotyping_result()), {0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0}));
        auto v2 = bi(2000, {0, 1}); kmer(v2, 20, {0}); kmer(v2, 0, {1});
        ProbabilityTable p2(0, 1, 21, 0.0L);
        p2.modify_probability(0, 10, CopyNumber(0.0, 1.0, 0.0));
        p2.modify_probability(0, 20, CopyNumber(0.0, 0.0, 1.0));
        p2.modify_probability(0, 0, CopyNumber(1.0, 0.0, 0.0));
        vector<shared_ptr<UniqueKmers>> uks2 = {u1, v2, u3};
        HMM under(&uks2, &p2, true, false, 0.0, false, 0.25);
        CHECK(close_all(triples(under.get_genotyping_result()), {0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0}));
    });
    run("HMM only_paths / normalize / combine_likelihoods", [&] {
        us only = {0, 3};
        auto u1 = multi(2000, {0, 2, 1, 1}); kmer(u1, 10, {0}); kmer(u1, 10, {1});
        auto u2 = multi(3000, {0, 0, 2, 1}); kmer(u2, 20, {0}); kmer(u2, 1, {1});
        ProbabilityTable probs(0, 1, 21, 0.0L);
        probs.modify_probability(0, 10, CopyNumber(0.1, 0.9, 0.1));
        probs.modify_probability(0, 20, CopyNumber(0.01, 0.01, 0.9));
        probs.modify_probability(0, 1, CopyNumber(0.9, 0.3, 0.1));
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2};
        HMM hmm1(&uks, &probs, true, false, R01, false, 0.25, &only);
        vector<double> first = triples(hmm1.get_genotyping_result());
        CHECK(close_all(first, {0.0509465435, 0.9483202731, 0.0007331832, 0.9678020017, 0.031003181, 0.0011948172}));

        us only01 = {0, 1};
        auto w1 = multi(2000, {0, 1, 2}); kmer(w1, 12, {2});
        auto w2 = multi(3000, {0, 1, 2}); kmer(w2, 12, {2});
        ProbabilityTable p13(0, 1, 13, 0.0L);
        p13.modify_probability(0, 12, CopyNumber(0.05, 0.8, 0.15));
        vector<shared_ptr<UniqueKmers>> wks = {w1, w2};
        HMM raw(&wks, &p13, true, false, R01, false, 0.25, &only01, false);
        CHECK(close_all(triples(raw.get_genotyping_result()), {0.000625, 0.00125, 0.000625, 0.0125, 0.025, 0.0125}));
        raw.normalize();
        vector<double> second = triples(raw.get_genotyping_result());
        CHECK(close_all(second, {0.25, 0.5, 0.25, 0.25, 0.5, 0.25}));
        hmm1.combine_likelihoods(raw);
        vector<double> expect;
        for (size_t i = 0; i < 6; ++i) expect.push_back(first[i] + second[i]);
        CHECK(close_all(triples(hmm1.get_genotyping_result()), expect));
        vector<shared_ptr<UniqueKmers>> one = {w1};
        HMM small(&one, &p13, true, false, R01, false, 0.25, &only01);
        CHECK_THROWS(hmm1.combine_likelihoods(small));
    });
    run("HMM error behaviour", [] {
        ProbabilityTable none;
        auto u = bi(2000, {0, 1});
        vector<shared_ptr<UniqueKmers>> uks = {u};
        us nobody = {5, 6};
        CHECK_THROWS(HMM(&uks, &none, true, false, 1.26, false, 0.25, &nobody));  // column not covered by any paths
    });
    run("index archive -> HMM -> VCF sample column (the reference's region fixtures)", [] {
        // tests/CommandsTest.cpp:59-93 builds these strings from a directly constructed HMM over the archive; the
        // expected values here are the CPU oracle's for the same fixture and parameters (tests/test_cereal_io.py)
        ProbabilityTable probs(18 / 4, 18 * 4, 2 * 18, 0.01L);
        struct Case { const char* file; vector<us> defined; vector<std::string> expect; };
        vector<Case> cases = {
            {"region_UniqueKmersList.cereal", {{0, 1}, {0, 1, 2}}, {"0/1:10000:-55.77,0,-56.61:30", "0/1:10000:-30.56,0,-40.21,-85.2,-77.92,-103.2:34"}},
            {"region2_UniqueKmersList.cereal", {{0, 1}, {0, 1}}, {"0/1:10000:-55.72,0,-56.66:30", "0/1:10000:-29.55,0,-41.23:34"}}};
        for (auto& c : cases) {
            UniqueKmersMap m = load_unique_kmers_map(g_golden_dir + "/" + c.file);
            HMM hmm(&m.unique_kmers["chr1"], &probs, true, false, 1.26, false, 0.00001L);
            auto res = hmm.get_genotyping_result();
            CHECK(res.size() == 2);
            for (size_t i = 0; i < res.size() && i < 2; ++i) {
                res[i].normalize();
                us ids;
                m.unique_kmers["chr1"][i]->get_allele_ids(ids);
                const std::string got = genotype_field(res[i], c.defined[i], ids.size());
                if (got != c.expect[i]) std::printf("    got %s expected %s\n", got.c_str(), c.expect[i].c_str());
                CHECK(got == c.expect[i]);
            }
        }
    });
    run("run_contigs_multi_gpu == one HMM per task", [&] {
        // three tasks (two contigs, one of them with a path subset) through the multi-GPU job loop on the
        // devices present; results must equal those of the one-shot HMM constructor, bin for bin
        auto mk = [](size_t pos, us paths, unsigned short c0, unsigned short c1) {
            auto u = bi(pos, paths); kmer(u, c0, {0}); kmer(u, c1, {1}); u->set_coverage(5); return u; };
        vector<shared_ptr<UniqueKmers>> c1 = {mk(2000, {0, 1, 1, 0}, 10, 10), mk(3000, {0, 1, 0, 0}, 20, 5), mk(4100, {1, 1, 0, 0}, 5, 20)};
        vector<shared_ptr<UniqueKmers>> c2 = {mk(500, {0, 1, 1, 0}, 5, 5), mk(900, {1, 0, 1, 0}, 20, 10)};
        us sub = {0, 1, 3};
        ProbabilityTable probs = std_table();
        vector<ContigTask> tasks = {{&c1, nullptr}, {&c2, nullptr}, {&c1, &sub}};
        vector<int> devs;
        for (int d = 0; d < HMM::device_count(); ++d) devs.push_back(d);
        auto res = run_contigs_multi_gpu(tasks, &probs, R01, false, 0.25, devs);
        CHECK(res.size() == 3);
        for (size_t t = 0; t < tasks.size(); ++t) {
            HMM one(tasks[t].unique_kmers, &probs, true, false, R01, false, 0.25, tasks[t].only_paths, false);
            auto ref = one.get_genotyping_result();
            CHECK(ref.size() == res[t].size());
            for (size_t v = 0; v < ref.size() && v < res[t].size(); ++v) {
                CHECK(ref[v].get_stored_likelihoods() == res[t][v].get_stored_likelihoods());
                CHECK(ref[v].coverage() == res[t][v].coverage() && ref[v].nr_unique_kmers() == res[t][v].nr_unique_kmers());
            }
        }
    });
    run("HMM genotyping + phasing in one constructor (no_unique_kmers3)", [] {
        // reference tests/HMMTest.cpp:392-438: likelihoods AND the Viterbi haplotypes
        auto u1 = bi(2000, {0, 1}); kmer(u1, 10, {0}); kmer(u1, 10, {1});
        auto u2 = bi(3000, {0, 1});
        auto u3 = bi(4000, {0, 1}); kmer(u3, 10, {0}); kmer(u3, 9, {1});
        ProbabilityTable probs(0, 1, 21, 0.0L);
        probs.modify_probability(0, 10, CopyNumber(0.1, 0.9, 0.1));
        probs.modify_probability(0, 9, CopyNumber(0.1, 0.8, 0.1));
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2, u3};
        HMM hmm(&uks, &probs, true, true, R01, false, 0.25);
        auto res = hmm.get_genotyping_result();
        CHECK(close_all(triples(res), {0.00264169937, 0.99471660125, 0.00264169937, 0.02552917716, 0.94894164567, 0.02552917716,
                                       0.002961313333, 0.99407737333, 0.002961313333}));
        us h1, h2;
        for (auto& r : res) { h1.push_back(r.get_haplotype().first); h2.push_back(r.get_haplotype().second); }
        CHECK((h1 == us{0, 0, 0} && h2 == us{1, 1, 1}) || (h1 == us{1, 1, 1} && h2 == us{0, 0, 0}));
    });
    run("24 HMM constructors at once on worker threads (src/commands.cpp:949-978) == one after the other", [] {
        // the reference's calling pattern: one constructor per (contig x subset) on thread-pool workers, all sharing the
        // ProbabilityTable; mixed shapes; every worker must get exactly what it gets alone (stored likelihoods are
        // compared as long doubles, bit for bit), one worker's failure must stay its own
        ProbabilityTable probs(6, 108, 54, 0.01L);
        const size_t shapes[24][2] = {{900, 64}, {400, 16}, {300, 128}, {700, 64}, {250, 30}, {1200, 16}, {500, 64}, {350, 64},
                                      {150, 128}, {800, 16}, {450, 64}, {200, 5}, {650, 64}, {300, 32}, {1000, 64}, {120, 128},
                                      {550, 16}, {380, 64}, {270, 64}, {600, 48}, {330, 64}, {90, 2}, {1, 64}, {40, 64}};
        vector<vector<shared_ptr<UniqueKmers>>> panels;
        for (auto& sh : shapes) panels.push_back(viterbi_panel(sh[0], sh[1], sh[1] > 4 ? sh[1] - sh[1] / 6 : sh[1]));
        vector<vector<GenotypingResult>> alone(24), together(24);
        for (size_t t = 0; t < 24; ++t) alone[t] = HMM(&panels[t], &probs, true, false, 1.26, false, 1e-5L, nullptr, false).get_genotyping_result();
        vector<std::string> errors(25);
        vector<shared_ptr<UniqueKmers>> broken = viterbi_panel(50, 16, 16);
        broken.push_back(bi(999999, {0, 1}));  // its paths {0, 1} restricted to an only_paths set without them: "not covered by any paths"
        vector<std::thread> workers;
        for (size_t t = 0; t < 24; ++t)
            workers.emplace_back([&, t] {
                try { together[t] = HMM(&panels[t], &probs, true, false, 1.26, false, 1e-5L, nullptr, false).get_genotyping_result(); }
                catch (const std::exception& e) { errors[t] = e.what(); }
            });
        workers.emplace_back([&] {
            vector<unsigned short> only = {2, 3, 4};
            try { HMM h(&broken, &probs, true, false, 1.26, false, 1e-5L, &only, false); }
            catch (const std::exception& e) { errors[24] = e.what(); }
        });
        for (auto& w : workers) w.join();
        for (size_t t = 0; t < 24; ++t) {
            CHECK(errors[t].empty());
            CHECK(alone[t].size() == together[t].size());
            size_t same = 0;
            for (size_t v = 0; v < alone[t].size() && v < together[t].size(); ++v)
                same += alone[t][v].get_stored_likelihoods() == together[t][v].get_stored_likelihoods() &&
                        alone[t][v].coverage() == together[t][v].coverage() && alone[t][v].nr_unique_kmers() == together[t][v].nr_unique_kmers();
            CHECK(same == alone[t].size());
        }
        CHECK(!errors[24].empty());
        uint64_t st[3] = {0, 0, 0};
        pg_hmm_coalesce_stats(st);
        CHECK(st[2] >= 2);  // calls were merged
    });
    run("run_genotype_command on the reference's index fixture: the genotyped VCF (tests/CommandsTest.cpp:18-93)", [] {
        std::vector<GenotypingResult> results;
        const std::vector<std::string> lines = genotype_index_fixture(&results);
        CHECK(lines.size() == 12 + 2 && results.size() == 2);
        auto fields = [](const std::string& l) { std::vector<std::string> f; std::string t; std::istringstream is(l); while (std::getline(is, t, '\t')) f.push_back(t); return f; };
        // the records of tests/data/region.vcf: chr1 139 T C and chr1 208 TG CG,CA with their ids; 42 of the 44 / 45 bubble
        // alleles are undefined sequence (MA); allele frequencies over the 214 panel paths
        Graph graph = Graph::load(g_golden_dir + "/index_chr1_Graph.cereal");
        auto af = [&](size_t variant, unsigned short allele) {
            float n = 0.0f;
            for (size_t p = 0; p < 215; ++p) n += graph.get_variant(variant).get_allele_on_path(p) == allele ? 1.0f : 0.0f;
            std::ostringstream os; os << std::setprecision(6) << n / 214u; return os.str();
        };
        const std::vector<std::string> a = fields(lines[12]), b = fields(lines[13]);
        CHECK(a.size() == 10 && b.size() == 10);
        const std::vector<std::string> want_a = {"chr1", "139", ".", "T", "C", ".", "PASS",
            "AF=" + af(0, 1) + ";UK=62;MA=42;ID=chr1-49638-SNV->50027902>50027904>50027905-1", "GT:GQ:GL:KC"};
        const std::vector<std::string> want_b = {"chr1", "208", ".", "TG", "CG,CA", ".", "PASS",
            "AF=" + af(1, 1) + "," + af(1, 2) + ";UK=" + std::to_string(results[1].nr_unique_kmers()) +
            ";MA=42;ID=chr1-49707-SNV->50027911>50027913>50027914-1,chr1-49707-COMPLEX->50027911>50027913>50027915>50027916-2", "GT:GQ:GL:KC"};
        for (size_t k = 0; k < 9 && a.size() == 10 && b.size() == 10; ++k) {
            if (a[k] != want_a[k]) std::printf("  record 0 column %zu: %s\n", k, a[k].c_str());
            if (b[k] != want_b[k]) std::printf("  record 1 column %zu: %s\n", k, b[k].c_str());
            CHECK(a[k] == want_a[k] && b[k] == want_b[k]);
        }
        // the sample columns as tests/CommandsTest.cpp:59-93 forms them from a directly constructed HMM
        std::vector<std::vector<unsigned short>> defined = {{0, 1}, {0, 1, 2}};
        CHECK(a.size() == 10 && a[9] == genotype_field(results[0], defined[0], 44));
        CHECK(b.size() == 10 && b[9] == genotype_field(results[1], defined[1], 45));
        CHECK(results[0].coverage() == 30 && results[1].coverage() == 34);
    });
    run("HMM phasing only (tests/HMMTest.cpp:392-438 without the likelihoods)", [] {
        auto u1 = bi(2000, {0, 1}); kmer(u1, 10, {0}); kmer(u1, 10, {1});
        auto u2 = bi(3000, {0, 1});
        auto u3 = bi(4000, {0, 1}); kmer(u3, 10, {0}); kmer(u3, 9, {1});
        ProbabilityTable probs(0, 1, 21, 0.0L);
        probs.modify_probability(0, 10, CopyNumber(0.1, 0.9, 0.1));
        probs.modify_probability(0, 9, CopyNumber(0.1, 0.8, 0.1));
        vector<shared_ptr<UniqueKmers>> uks = {u1, u2, u3};
        HMM hmm(&uks, &probs, false, true, 446.287102628, false, 0.25);
        auto res = hmm.get_genotyping_result();
        us h1, h2;
        for (auto& r : res) { h1.push_back(r.get_haplotype().first); h2.push_back(r.get_haplotype().second); CHECK(r.contains_no_likelihoods()); }
        CHECK((h1 == us{0, 0, 0} && h2 == us{1, 1, 1}) || (h1 == us{1, 1, 1} && h2 == us{0, 0, 0}));
        CHECK(res[0].nr_unique_kmers() == 2 && res[1].nr_unique_kmers() == 0 && res[2].nr_unique_kmers() == 2);
    });
    run("HMM Viterbi (pg_viterbi.hip): phasing alone == phasing with genotyping; more than 64 paths are refused", [] {
        // 30 / 45 / 64 / 12 paths (lanes per row of states 32 / 64 / 64 / 16), duplicated paths for exact ties, three
        // transition regimes (default, strong recombination, none).  (Device vs the long double oracle: tests/test_viterbi_gpu.py.)
        ProbabilityTable probs(6, 108, 54, 0.01L);
        const size_t shapes[4][2] = {{300, 45}, {500, 30}, {200, 64}, {260, 12}};
        for (auto& sh : shapes) {
            const size_t V = sh[0], H = sh[1];
            auto a = viterbi_panel(V, H, H - H / 6);
            for (int regime = 0; regime < 3; ++regime) {
                const double rec = regime == 1 ? 446.287102628 : (regime == 2 ? 0.0 : 1.26);
                const long double N = regime == 1 ? 0.25L : 25000.0L;
                HMM dev(&a, &probs, false, true, rec, false, N);
                HMM both(&a, &probs, true, true, rec, false, N);
                auto rb = dev.get_genotyping_result(), rc = both.get_genotyping_result();
                size_t same = 0, meta = 0;
                for (size_t v = 0; v < V; ++v) {
                    same += rb[v].get_haplotype() == rc[v].get_haplotype();
                    meta += rb[v].contains_no_likelihoods();
                }
                CHECK(same == V);
                CHECK(meta == V);
            }
        }
        auto wide = viterbi_panel(20, 70, 60);
        bool threw = false;
        try { HMM h(&wide, &probs, false, true); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    });
}

// ------------------------------------------------------------------ the reference's demo (BASELINE.json configs[0]; README "Demo"):
//   PanGenie-index -r test-reference.fa -v test-variants.vcf -o preprocessing ; PanGenie -f preprocessing -i test-reads.fa -o test
// composed from the host pieces.  Kept on the test side: it is the reference's command layer (run_index_command +
// run_genotype_command, src/commands.cpp:592-1052), which this build does not ship; what it shows is that the pieces either side
// of the device path reproduce demo/test_genotyping.vcf.
//
// The k-mer abundance peak of the default run (only graph k-mers counted => the LARGEST peak, src/commands.cpp:840): the
// histogram is averaged in place over windows of three, left to right, so every mean already sees the mean before it; a peak
// is the last position of a rise before a fall; the highest one wins, the earlier one on ties.  (Test-side on purpose: the
// product takes the peak as an argument, kmer_counts.hpp.)
static size_t demo_abundance_peak(std::vector<size_t> seen) {
    for (size_t c = 1; c + 1 < seen.size(); ++c) seen[c] = (seen[c - 1] + seen[c] + seen[c + 1]) / 3;
    size_t best = 0, best_height = 0, before = 0;
    bool found = false, falling = false;
    for (size_t c = 0; c < seen.size(); before = seen[c], ++c) {
        if (seen[c] > before) falling = false;
        else if (seen[c] < before) {
            if (!falling && (!found || before > best_height)) { best = c - 1; best_height = before; found = true; }
            falling = true;
        }
    }
    if (!found) throw std::runtime_error("no peak in the k-mer abundance histogram");
    return best;
}

struct DemoSample {
    std::vector<std::string> chromosomes;   // in the order run_genotype_command visits them (the archive's map order)
    UniqueKmersMap counted;
    size_t peak = 0;
};

// PanGenie-index, then steps 1-3 of run_genotype_command (src/commands.cpp:812-876): counts of the graph's k-mers in the reads,
// the abundance peak, counts and local coverage into the index
static DemoSample sample_prepare(const std::string& prefix, const std::string& readfile, unsigned threads);
static DemoSample demo_prepare(const std::string& demo_dir, const std::string& prefix, unsigned threads) {
    build_index(demo_dir + "/test-reference.fa", demo_dir + "/test-variants.vcf", prefix, 31, true);
    return sample_prepare(prefix, demo_dir + "/test-reads.fa", threads);
}
static DemoSample sample_prepare(const std::string& prefix, const std::string& readfile, unsigned threads) {
    DemoSample d;
    d.counted = load_unique_kmers_map(prefix + "_UniqueKmersMap.cereal");
    TargetedKmerCounter reads(d.counted.kmersize);
    reads.add_targets_from_sequences(prefix + "_path_segments.fasta");
    reads.count(readfile, threads);
    d.peak = demo_abundance_peak(reads.abundance_histogram(10000));
    for (auto& kv : d.counted.unique_kmers) {
        d.chromosomes.push_back(kv.first);
        fill_read_kmercounts(kv.first, &d.counted, reads, prefix + "_" + kv.first + "_kmers.tsv.gz", d.peak);
    }
    return d;
}

// the writing part of run_genotype_command / run_vcf_command (src/commands.cpp:1019-1044, :1106-1135)
static void demo_write_vcf(const std::string& prefix, const std::map<std::string, std::vector<GenotypingResult>>& results,
                           const std::vector<std::string>& chromosomes, const std::string& out, const std::string& sample, bool phasing = false) {
    std::remove(out.c_str());
    bool header = true;
    for (const std::string& c : chromosomes) {
        const Graph graph = Graph::load(prefix + "_" + c + "_Graph.cereal");
        if (phasing) graph.write_phasing(out, results.at(c), header, sample, false);
        else graph.write_genotypes(out, results.at(c), header, sample, false);
        header = false;
    }
}

// the whole default run with the HMM on the device: one subset of all paths (<= 100 paths: no sampling; src/commands.cpp:799-803,
// :906-915), likelihoods unnormalised out of the HMM, normalised afterwards (:160, :981-987)
// With `phasing_out` also the phasing job of a `-p` run (:961-966: the Viterbi path over min(paths, 30) paths — PathSampler's
// single subset of ALL paths is all of them, whatever its random numbers) and its VCF.
static void sample_genotype_on_device(DemoSample& d, const std::string& prefix, const std::string& out, const std::string& phasing_out);
static void demo_genotype_on_device(const std::string& demo_dir, const std::string& prefix, const std::string& out, const std::string& phasing_out) {
    DemoSample d = demo_prepare(demo_dir, prefix, 2);
    sample_genotype_on_device(d, prefix, out, phasing_out);
}
static void sample_genotype_on_device(DemoSample& d, const std::string& prefix, const std::string& out, const std::string& phasing_out) {
    ProbabilityTable probs(d.peak / 4, d.peak * 4, 2 * d.peak, 0.01L);
    std::map<std::string, std::vector<GenotypingResult>> results, phasings;
    for (const std::string& c : d.chromosomes) {
        std::vector<std::shared_ptr<UniqueKmers>>& uks = d.counted.unique_kmers[c];
        if (uks.empty()) { results[c] = {}; phasings[c] = {}; continue; }
        // more than 100 paths: 15 haplotypes are sampled first (src/commands.cpp:799-803; the sampler at the end of the
        // reference's fill_read_kmercounts, :148-151, with the command line's defaults: sampling effective N 0.01, allele
        // penalty 5) and the panel shrinks to them (+ the reference path)
        if (uks[0]->get_nr_paths() > 100) HaplotypeSampler(&uks, 15, 1.26, 0.01L, nullptr, d.counted.add_reference, "", c, 5);
        std::vector<unsigned short> all_paths(uks[0]->get_nr_paths());
        for (size_t p = 0; p < all_paths.size(); ++p) all_paths[p] = (unsigned short)p;
        HMM hmm(&uks, &probs, true, false, 1.26, false, 0.00001L, &all_paths, false);
        results[c] = hmm.move_genotyping_result();
        for (GenotypingResult& r : results[c]) r.normalize();
        if (!phasing_out.empty()) {
            if (all_paths.size() > 30) throw std::runtime_error("demo: more than 30 paths (the reference would sample 30 for phasing)");
            HMM viterbi(&uks, &probs, false, true, 1.26, false, 0.00001L, &all_paths, false);
            phasings[c] = viterbi.move_genotyping_result();
        }
    }
    demo_write_vcf(prefix, results, d.chromosomes, out, "sample");
    if (!phasing_out.empty()) demo_write_vcf(prefix, phasings, d.chromosomes, phasing_out, "sample", true);
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    if (argc > 2) g_golden_dir = argv[2];
    if (mode == "cpu") { cpu_tests(); archive_cpu_tests(); graph_cpu_tests(); index_builder_cpu_tests(); kmer_count_cpu_tests(); sampler_cpu_tests(); }
    else if (mode == "gpu") { gpu_tests(); sampler_gpu_tests(); }
    else if (mode == "dump-results" && argc >= 3) {  // the archive of sample_results() for the Python reader (tests/test_cereal_io.py)
        save_results(sample_results(), argv[2]);
        return 0;
    }
    else if (mode == "write-vcf" && argc >= 4) {   // the genotyped VCF of the index fixture (tests/test_cereal_io.py compares it with the oracle)
        std::FILE* f = std::fopen(argv[3], "w");
        if (!f) return 2;
        for (const std::string& l : genotype_index_fixture()) std::fprintf(f, "%s\n", l.c_str());
        std::fclose(f);
        return 0;
    }
    else if (mode == "index" && argc >= 5) {   // PanGenie-index on any input (scale checks)
        const std::vector<std::string> chromosomes = build_index(argv[2], argv[3], argv[4], argc > 5 ? (size_t)std::atoi(argv[5]) : 31u, true, argc > 6 ? (unsigned)std::atoi(argv[6]) : 1u,
                                                                  argc > 7 && std::string(argv[7]) == "whole");
        std::printf("%zu chromosomes\n", chromosomes.size());
        return 0;
    }
    else if (mode == "demo-counts" && argc >= 4) {   // CPU: index + counted archive of the demo, peak on stdout (tests/test_demo.py)
        DemoSample d = demo_prepare(argv[2], argv[3], argc > 4 ? (unsigned)std::atoi(argv[4]) : 2u);
        save_unique_kmers_map(d.counted, std::string(argv[3]) + "_counted_UniqueKmersMap.cereal");
        std::printf("peak=%zu\n", d.peak);
        return 0;
    }
    else if (mode == "vcf" && argc >= 5) {   // CPU: PanGenie-vcf — a Results archive + the Graph archives of <prefix> -> genotyped VCF
        Results r = load_results(argv[3]);
        std::vector<std::string> chromosomes;
        for (const auto& kv : r.result) chromosomes.push_back(kv.first);
        demo_write_vcf(argv[2], r.result, chromosomes, argv[4], "sample", argc > 5 && std::string(argv[5]) == "phasing");
        return 0;
    }
    else if (mode == "counts" && argc >= 4) {   // CPU: the counted archive of any index + reads, peak on stdout
        DemoSample d = sample_prepare(argv[2], argv[3], argc > 4 ? (unsigned)std::atoi(argv[4]) : 8u);
        save_unique_kmers_map(d.counted, std::string(argv[2]) + "_counted_UniqueKmersMap.cereal");
        std::printf("peak=%zu\n", d.peak);
        return 0;
    }
    else if (mode == "genotype" && argc >= 5) {   // GPU: PanGenie -f <prefix> -i <reads> on any index (tools/pipeline_check.sh); stage times on stderr
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { const auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "genotype: %-36s %8.3f s\n", what, std::chrono::duration<double>(t - t0).count()); t0 = t; };
        DemoSample d = sample_prepare(argv[2], argv[3], argc > 5 ? (unsigned)std::atoi(argv[5]) : 8u);
        std::fprintf(stderr, "genotype: k-mer abundance peak %zu\n", d.peak);
        lap("reads counted, counts into the index");
        sample_genotype_on_device(d, argv[2], argv[4], "");
        lap("HMM on the device, VCF written");
        return 0;
    }
    else if (mode == "cohort" && argc >= 6) {   // GPU: several samples against one index in ONE device job: cohort <prefix> <out prefix> <threads> <reads>...
        const std::string prefix = argv[2], out_prefix = argv[3];
        const unsigned threads = (unsigned)std::atoi(argv[4]);
        UniqueKmersMap index = load_unique_kmers_map(prefix + "_UniqueKmersMap.cereal");
        std::vector<std::string> chromosomes;
        for (const auto& kv : index.unique_kmers) chromosomes.push_back(kv.first);
        std::vector<SampleCounts> samples;
        size_t low = ~(size_t)0, high = 0;
        for (int i = 5; i < argc; ++i) {   // per sample: graph-only counts, abundance peak, counts into the (shared) objects, copied out
            TargetedKmerCounter reads(index.kmersize);
            reads.add_targets_from_sequences(prefix + "_path_segments.fasta");
            reads.count(argv[i], threads);
            const size_t peak = demo_abundance_peak(reads.abundance_histogram(10000));
            for (const std::string& c : chromosomes) fill_read_kmercounts(c, &index, reads, prefix + "_" + c + "_kmers.tsv.gz", peak);
            samples.push_back(SampleCounts::of(index.unique_kmers));
            low = std::min(low, peak); high = std::max(high, peak);
            std::fprintf(stderr, "cohort: sample %d peak %zu\n", i - 5, peak);
        }
        ProbabilityTable probs(low / 4, high * 4, 2 * high, 0.01L);   // one table whose box spans every sample's
        auto results = genotype_cohort(index.unique_kmers, samples, &probs, 1.26, false, 0.00001L, 0);
        for (size_t s = 0; s < results.size(); ++s) {
            for (auto& kv : results[s]) for (GenotypingResult& r : kv.second) r.normalize();
            demo_write_vcf(prefix, results[s], chromosomes, out_prefix + "_" + std::to_string(s) + ".vcf", "sample");
        }
        return 0;
    }
    else if (mode == "demo" && argc >= 5) {   // GPU: the demo end to end
        demo_genotype_on_device(argv[2], argv[3], argv[4], argc > 5 ? argv[5] : "");
        return 0;
    }
    else { std::printf("usage: test_host cpu|gpu\n"); return 2; }
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
