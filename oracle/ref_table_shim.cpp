// TEST INFRASTRUCTURE ONLY.  C entry points around the reference's OWN ProbabilityTable and CopyNumber translation
// units (src/probabilitytable.cpp, src/copynumber.cpp — free of the <cereal/...> headers this image lacks, so they
// compile from the sources where they lie under /root/reference).  Built by `make ref` into oracle/_ref/; used by
// tests/test_oracle_golden.py to pin pg_oracle.c's table (and the product's host-side table, pg_shim.cpp) bit for bit.
#include "probabilitytable.hpp"

extern "C" void* ref_table_create(unsigned short cov_min, unsigned short cov_max, unsigned short count_max, long double regularization) {
    return new ProbabilityTable(cov_min, cov_max, count_max, regularization);
}
extern "C" void* ref_table_create_default() { return new ProbabilityTable(); }
extern "C" void ref_table_destroy(void* t) { delete static_cast<ProbabilityTable*>(t); }
extern "C" void ref_table_get(const void* t, unsigned short coverage, unsigned short count, long double out3[3]) {
    const CopyNumber cn = static_cast<const ProbabilityTable*>(t)->get_probability(coverage, count);
    for (int i = 0; i < 3; ++i) out3[i] = cn.get_probability_of(i);
}
extern "C" void ref_copynumber_regularized(long double cn0, long double cn1, long double cn2, long double reg, long double out3[3]) {
    const CopyNumber cn(cn0, cn1, cn2, reg);
    for (int i = 0; i < 3; ++i) out3[i] = cn.get_probability_of(i);
}
