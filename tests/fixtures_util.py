"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: build batches
and tables from the golden JSON (tests/golden/reference_known_answers.json)."""
import numpy as np

from pangenie_amd.panel import BiallelicUniqueKmers, MultiallelicUniqueKmers, flatten

LD = np.longdouble


def build_variant(d):
    ctor = BiallelicUniqueKmers if d["type"] == "biallelic" else MultiallelicUniqueKmers
    u = ctor(d["pos"], d["path_to_allele"])
    for a in d["undefined"]:
        u.set_undefined_allele(a)
    for count, alleles in d["kmers"]:
        u.insert_kmer(count, alleles)
    u.set_coverage(d["coverage"])
    return u


def build_batch(variants, only_paths=None):
    return flatten([build_variant(v) for v in variants], only_paths)


def fill_table(table, spec, copynumber_regularized):
    """Apply the fixture's modify_probability calls to `table` (oracle or product)."""
    for m in spec["modify"]:
        cov, count, p = m[0], m[1], m[2]
        if len(m) == 4:  # CopyNumber(p0,p1,p2,reg) form
            p = copynumber_regularized(p[0], p[1], p[2], m[3])
        table.modify(cov, count, p[0], p[1], p[2])
    return table


def triple(result, present=(0, 1)):
    return [float(result.get_genotype_likelihood(0, 0)),
            float(result.get_genotype_likelihood(0, 1)),
            float(result.get_genotype_likelihood(1, 1))]
