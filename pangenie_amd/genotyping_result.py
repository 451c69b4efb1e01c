"""Host-side mirror of the reference's GenotypingResult (src/genotypingresult.{hpp,cpp}).

Post-processing stays on the host in 80-bit `np.longdouble`, exactly where the
reference does it (normalisation src/commands.cpp:981-987, GT/GQ
src/graph.cpp:229-273): the device only produces the unnormalised bins.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

LD = np.longdouble


class GenotypingResult:
    def __init__(self):
        self.genotype_to_likelihood: Dict[Tuple[int, int], LD] = {}
        self.haplotype_1 = 0
        self.haplotype_2 = 0
        self.local_coverage = 0
        self.unique_kmers = 0

    # src/genotypingresult.cpp:16-28
    @staticmethod
    def _genotype(a1: int, a2: int) -> Tuple[int, int]:
        return (a1, a2) if a1 < a2 else (a2, a1)

    def add_to_likelihood(self, a1: int, a2: int, value) -> None:
        g = self._genotype(a1, a2)
        self.genotype_to_likelihood[g] = self.genotype_to_likelihood.get(g, LD(0)) + LD(value)

    # :38-46
    def get_genotype_likelihood(self, a1: int, a2: int) -> LD:
        return self.genotype_to_likelihood.get(self._genotype(a1, a2), LD(0))

    # :48-67 (VCF ordering)
    def get_all_likelihoods(self, nr_alleles: int) -> List[LD]:
        n = nr_alleles * (nr_alleles + 1) // 2
        out = [LD(0)] * n
        for (a1, a2), l in self.genotype_to_likelihood.items():
            idx = (a2 * (a2 + 1)) // 2 + a1
            if idx >= n:
                raise RuntimeError("GenotypeResult::get_all_likelihoods: genotype does not match number of alleles.")
            out[idx] = l
        return out

    # :70-96
    def get_specific_likelihoods(self, alleles: Sequence[int]) -> "GenotypingResult":
        res = GenotypingResult()
        index = {a: i for i, a in enumerate(alleles)}
        s = LD(0)
        for (g1, g2), l in sorted(self.genotype_to_likelihood.items()):
            if g1 not in index or g2 not in index:
                continue
            i, j = index[g1], index[g2]
            if self.haplotype_1 == g1:
                res.haplotype_1 = i
            if self.haplotype_2 == g2:
                res.haplotype_2 = j
            res.add_to_likelihood(i, j, l)
            s += l
        if s > 0:
            res.divide_likelihoods_by(s)
        return res

    # :118-137
    def get_genotype_quality(self, a1: int, a2: int) -> int:
        s = sum(self.genotype_to_likelihood.values(), LD(0))
        if abs(s - 1) > 0.0000000001:
            raise RuntimeError("GenotypingResult::get_genotype_quality: genotype quality can only be computed from normalized likelihoods.")
        prob_wrong = LD(1) - self.get_genotype_likelihood(a1, a2)
        if prob_wrong > 0.0:
            return int(-10 * np.log10(prob_wrong))
        return 10000

    # :143-147
    def divide_likelihoods_by(self, value) -> None:
        v = LD(value)
        for g in self.genotype_to_likelihood:
            self.genotype_to_likelihood[g] = self.genotype_to_likelihood[g] / v

    # :149-180 (ties within 1e-10 -> ./.; `>=` so the LAST maximal key in map order wins)
    def get_likeliest_genotype(self) -> Tuple[int, int]:
        if not self.genotype_to_likelihood:
            return (-1, -1)
        best_value = LD(0)
        best = (0, 0)
        items = sorted(self.genotype_to_likelihood.items())
        for g, l in items:
            if l >= best_value:
                best_value, best = l, g
        for g, l in items:
            if g != best and abs(l - best_value) < 0.0000000001:
                return (-1, -1)
        return best if best_value > 0 else (-1, -1)

    # :193-198
    def combine(self, other: "GenotypingResult") -> None:
        for g, l in other.genotype_to_likelihood.items():
            self.genotype_to_likelihood[g] = self.genotype_to_likelihood.get(g, LD(0)) + l

    # :200-210
    def normalize(self) -> None:
        s = LD(0)
        for _, l in sorted(self.genotype_to_likelihood.items()):
            s += l
        if s > 0:
            self.divide_likelihoods_by(s)

    def contains_no_likelihoods(self) -> bool:
        return not self.genotype_to_likelihood

    def nr_unique_kmers(self) -> int:
        return self.unique_kmers

    def coverage(self) -> int:
        return self.local_coverage


def vcf_sample_field(result: "GenotypingResult", defined_alleles: Sequence[int], nr_alleles: int, ignore_imputed: bool = False) -> str:
    """`GT:GQ:GL:KC` of one record as Graph::write_genotypes prints it (reference src/graph.cpp:217-273);
    mirror of pangenie::genotype_field (pangenie_amd/host/pangenie_host.hpp).  `result` must be normalised."""
    tmp = GenotypingResult()
    tmp.genotype_to_likelihood = dict(result.genotype_to_likelihood)
    tmp.local_coverage, tmp.unique_kmers = result.local_coverage, result.unique_kmers
    if tmp.contains_no_likelihoods():
        tmp.add_to_likelihood(0, 0, 1.0)
    nr_missing = nr_alleles - len(defined_alleles)
    gl = tmp.get_specific_likelihoods(defined_alleles) if nr_missing > 0 else tmp
    n = len(defined_alleles)
    g = gl.get_likeliest_genotype()
    if ignore_imputed and result.nr_unique_kmers() == 0:
        g = (-1, -1)
    out = f"{g[0]}/{g[1]}:{gl.get_genotype_quality(g[0], g[1])}:" if g[0] != -1 and g[1] != -1 else ".:.:"
    liks = gl.get_all_likelihoods(n)
    if len(liks) < 3:
        raise RuntimeError(f"Graph::write_genotypes_of: too few likelihoods ({len(liks)}) computed")

    def fmt(x):  # ostream << setprecision(4) << log10(long double)
        with np.errstate(divide="ignore"):
            v = float(np.log10(LD(x)))
        return "%.4g" % v if v == v and abs(v) != float("inf") else ("-inf" if v < 0 else ("inf" if v > 0 else "nan"))
    return out + ",".join(fmt(x) for x in liks) + f":{result.coverage()}"


def results_from_flat(batch, lik_ld: np.ndarray, kept: np.ndarray, allele_present: np.ndarray,
                      n_kmers: np.ndarray, coverage: np.ndarray, haplotype_1=None, haplotype_2=None,
                      with_likelihoods: bool = True) -> List[GenotypingResult]:
    """Rebuild vector<GenotypingResult> from the flat bins (include/pangenie_hmm.h layout):
    a bin becomes a map key iff the variant is a kept column and both allele
    slots occur on a selected path (reference src/hmm.cpp:368)."""
    out: List[GenotypingResult] = []
    geno_off = batch.geno_off
    for v in range(batch.n_variants):
        r = GenotypingResult()
        r.unique_kmers = int(n_kmers[v])
        r.local_coverage = int(coverage[v])
        if haplotype_1 is not None and kept[v]:  # Viterbi path (reference src/hmm.cpp:161-162)
            r.haplotype_1 = int(haplotype_1[v])
            r.haplotype_2 = int(haplotype_2[v])
        if kept[v] and with_likelihoods:
            a0, a1 = int(batch.allele_off[v]), int(batch.allele_off[v + 1])
            A = a1 - a0
            ids = batch.allele_id[a0:a1]
            pres = allele_present[a0:a1]
            base = int(geno_off[v])
            for a in range(A):
                if not pres[a]:
                    continue
                for b in range(a, A):
                    if not pres[b]:
                        continue
                    idx = base + a * A - a * (a - 1) // 2 + (b - a)
                    r.genotype_to_likelihood[(int(ids[a]), int(ids[b]))] = LD(lik_ld[idx])
        out.append(r)
    return out


def normalized_bins(batch, lik_ld: np.ndarray) -> np.ndarray:
    """Vectorised per-variant normalisation of flat bins in long double
    (GenotypingResult::normalize over all variants)."""
    geno_off = batch.geno_off.astype(np.int64)
    lik_ld = np.asarray(lik_ld, dtype=LD)
    sums = np.add.reduceat(np.concatenate([lik_ld, np.zeros(1, LD)]), geno_off[:-1]) if lik_ld.size else np.zeros(0, LD)
    G = np.diff(geno_off)
    sums = np.where(G > 0, sums, LD(0))
    denom = np.repeat(np.where(sums > 0, sums, LD(1)), G)
    return lik_ld / denom
