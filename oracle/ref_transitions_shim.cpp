// TEST INFRASTRUCTURE ONLY.  C entry point around the reference's OWN SamplingTransitions translation
// unit (src/samplingtransitions.cpp — the one file of the sampler path that compiles without the
// <cereal/...> headers this image lacks).  Built by `make ref` into oracle/_ref/ from the sources where
// they lie under /root/reference; used by tests/test_sampler.py to pin the restatement in
// pg_sampler_oracle.c (which overload of exp/log10 the reference ends up calling).
#include <cstddef>
#include "samplingtransitions.hpp"

extern "C" unsigned int ref_sampling_transition_cost(unsigned long long from_pos, unsigned long long to_pos, double recombrate,
                                                     unsigned short nr_paths, long double effective_N) {
    SamplingTransitions t((size_t)from_pos, (size_t)to_pos, recombrate, nr_paths, effective_N);
    return t.compute_transition_cost(true);
}
