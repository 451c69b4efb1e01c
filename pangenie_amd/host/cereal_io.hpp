// cereal_io.hpp — reader / writer of the reference's cereal *binary* archives either side of the path — without
// cereal: the UniqueKmers tables (`<prefix>_UniqueKmersMap.cereal`, the index PanGenie-index writes and PanGenie reads:
// reference src/commands.hpp:11-28, src/commands.cpp:653-705, :771-777) and the genotyping Results (`-w`, below).
//
// Layout (little endian, no framing; member order from the reference's serialize functions
// src/commands.hpp:19-22, src/biallelicuniquekmers.hpp:101-104 and :32-41,
// src/multiallelicuniquekmers.hpp:100-103, src/kmerpath.hpp:25-28, src/kmerpath16.hpp:25-28):
//   kmersize u64 · map<string, vector<shared_ptr<UniqueKmers>>>: u64 n; per entry string (u64 len + bytes) and
//   vector (u64 n; per element: polymorphic type id u32 — MSB set the first time a type occurs, then
//   followed by its registered name as a string — · shared-pointer id u32 (MSB set: new object, data follows;
//   else a back reference) · object: variant_pos u64, local_coverage f32, current_index u64, kmer_to_count
//   (u64 n + n x u16), alleles map (u64 n; key bool u8 | u16; value {offset u16, kmers u16 | u32},
//   is_undefined u8), path_to_allele (u64 n + n x (bool u8 | u16))) · runtimes, sampling_runtimes
//   map<string, f64> · add_reference u8.
// Checked byte for byte against the reference's own fixtures (tests/golden/region*_UniqueKmersList.cereal).
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "pangenie_host.hpp"

namespace pangenie {

/** reference src/commands.hpp:11-28 (without the mutex) */
struct UniqueKmersMap {
    size_t kmersize = 31;
    std::map<std::string, std::vector<std::shared_ptr<UniqueKmers>>> unique_kmers;
    std::map<std::string, double> runtimes;
    std::map<std::string, double> sampling_runtimes;
    bool add_reference = false;
};

/** throws std::runtime_error on malformed input */
UniqueKmersMap load_unique_kmers_map(const std::string& path);
UniqueKmersMap parse_unique_kmers_map(const std::vector<unsigned char>& bytes);
std::vector<unsigned char> serialize_unique_kmers_map(const UniqueKmersMap& m);
void save_unique_kmers_map(const UniqueKmersMap& m, const std::string& path);

/** reference src/commands.cpp:59-72 (without the mutex): the object PanGenie serializes with `-w` to
 *  `<out>_genotyping.cereal` (src/commands.cpp:511-516, :1012-1017) and PanGenie-vcf reads back (:1099-1104) to write
 *  the VCFs.  Layout: map<string, vector<GenotypingResult>> (u64 n; per entry string, u64 n, per element:
 *  genotype_to_likelihood — u64 n; per entry u16 a1, u16 a2 (the reference's own pair serializer,
 *  src/genotypingresult.hpp:13-29), long double as its 16 bytes in memory (x86-64: 80-bit value + 6 padding bytes,
 *  written as 0 here) — · haplotype_1 u16 · haplotype_2 u16 · local_coverage u16 · unique_kmers u16
 *  (src/genotypingresult.hpp:77-80)) · runtimes map<string, f64>. */
struct Results {
    std::map<std::string, std::vector<GenotypingResult>> result;
    std::map<std::string, double> runtimes;
};
Results load_results(const std::string& path);
Results parse_results(const std::vector<unsigned char>& bytes);
std::vector<unsigned char> serialize_results(const Results& r);
void save_results(const Results& r, const std::string& path);

}  // namespace pangenie
