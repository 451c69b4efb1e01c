"""The reference's demo (BASELINE.json configs[0]; README "Demo": PanGenie-index on demo/test-reference.fa +
test-variants.vcf, PanGenie on test-reads.fa) reproduced from the pieces either side of the device path, and compared with
the reference's own expected output demo/test_genotyping.vcf (4 records, GT:GQ:GL:KC) — SURVEY.md §8(c) "config #1".
The demo's files are kept as data under tests/golden/demo/.  The command layer that strings the pieces together lives on
the test side (tests/cpp/test_host.cpp: demo_prepare, demo_write_vcf, demo_genotype_on_device).

CPU test: index builder -> targeted k-mer counts -> abundance peak -> counts into the index (all C++ host code), the
HMM by the CPU oracle, a `-w` Results archive, the VCF by the host's Graph::write_genotypes (= the reference's PanGenie-vcf
step).  GPU test: the same with the HMM on the device, in one process."""
import subprocess
from pathlib import Path

import pytest

from pangenie_amd import cereal_io
from pangenie_amd.build import build_host, HOST_TEST
from pangenie_amd.genotyping_result import results_from_flat
from pangenie_amd.panel import flatten

DEMO = Path(__file__).resolve().parent / "golden" / "demo"


def without_date(text):
    lines = text.splitlines()
    assert lines[1].startswith("##fileDate=")
    return lines[:1] + lines[2:]


def expected():
    return without_date((DEMO / "test_genotyping.vcf").read_text())


def check_phasing(text):
    """demo/test_phasing.vcf was written by an earlier release (2023: other unique k-mer rules, hence other UK / KC values),
    so it pins what did not change: the header, the fixed columns, AF / MA and — the Viterbi path over the same panel and
    the same reads — the phased genotypes."""
    got, want = without_date(text), without_date((DEMO / "test_phasing.vcf").read_text())
    assert got[:9] == want[:9] and len(got) == len(want) == 13
    for g, w in zip(got[9:], want[9:]):
        g, w = g.split("\t"), w.split("\t")
        assert g[:7] == w[:7] and g[8] == w[8] == "GT:KC"
        gi, wi = (dict(kv.split("=") for kv in x[7].split(";")) for x in (g, w))
        assert gi["AF"] == wi["AF"] and gi["MA"] == wi["MA"] and sorted(gi) == sorted(wi)
        assert g[9].split(":")[0] == w[9].split(":")[0]
    assert [g.split("\t")[9] for g in got[9:]] == ["1|0:3", "0|1:3", "1|2:1", "1|1:3"]
    assert [g.split("\t")[7].split(";")[1] for g in got[9:]] == ["UK=32", "UK=32", "UK=48", "UK=32"]


def test_demo_index_counts_oracle_vcf(tmp_path):
    from oracle import pyoracle as orc
    build_host()
    prefix = tmp_path / "preprocessing"
    r = subprocess.run([str(HOST_TEST), "demo-counts", str(DEMO), str(prefix), "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    peak = int(r.stdout.strip().split("=")[1])
    assert peak == 3   # (the demo's reads are a thin sample: KC=3 in the expected records)
    index = cereal_io.load(str(prefix) + "_UniqueKmersMap.cereal")
    counted = cereal_io.load(str(prefix) + "_counted_UniqueKmersMap.cereal")
    assert counted.kmersize == 31 and counted.add_reference and list(counted.unique_kmers) == ["chr1"]
    uks = counted.unique_kmers["chr1"]
    # 12 samples = 24 haplotypes + the reference path; UK of the expected records
    assert [u.variant_pos for u in uks] == [15951, 16508, 16635, 18261] and all(len(u.path_to_allele) == 25 for u in uks)
    assert [len(u.kmer_to_count) for u in uks] == [32, 32, 48, 32]
    assert all(sum(u.kmer_to_count) == 0 for u in index.unique_kmers["chr1"]) and all(sum(u.kmer_to_count) > 0 for u in uks)
    assert [int(u.local_coverage) for u in uks] == [3, 3, 1, 3]

    # run_genotyping (src/commands.cpp:155-185) by the oracle: all 25 paths, unnormalised, then normalised (:981-987)
    res = cereal_io.Results()
    for chrom, objects in counted.unique_kmers.items():
        batch = flatten(objects)
        ref = orc.genotype_contig(batch, orc.OracleTable(peak // 4, peak * 4, 2 * peak, 0.01), orc.make_params(1.26, False, 1e-5))
        results = results_from_flat(batch, ref.lik, ref.kept, ref.allele_present, ref.n_kmers, ref.coverage)
        for g in results:
            g.normalize()
        res.result[chrom] = results
        res.runtimes[chrom] = 0.0
    archive = tmp_path / "test_genotyping.cereal"
    archive.write_bytes(cereal_io.dumps_results(res))
    out = tmp_path / "test_genotyping.vcf"
    r = subprocess.run([str(HOST_TEST), "vcf", str(prefix), str(archive), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert without_date(out.read_text()) == expected()

    # the phasing job of a `-p` run (src/commands.cpp:961-966): Viterbi over all 25 paths, by the oracle
    ph = cereal_io.Results()
    for chrom, objects in counted.unique_kmers.items():
        batch = flatten(objects)
        vit = orc.viterbi_contig(batch, orc.OracleTable(peak // 4, peak * 4, 2 * peak, 0.01),
                                 orc.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True))
        ph.result[chrom] = results_from_flat(batch, None, vit.kept, None, vit.n_kmers, vit.coverage, vit.hap1, vit.hap2, with_likelihoods=False)
        ph.runtimes[chrom] = 0.0
    archive.write_bytes(cereal_io.dumps_results(ph))
    out = tmp_path / "test_phasing.vcf"
    r = subprocess.run([str(HOST_TEST), "vcf", str(prefix), str(archive), str(out), "phasing"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    check_phasing(out.read_text())


@pytest.mark.gpu
def test_demo_end_to_end_on_the_device(tmp_path):
    build_host()
    out, phasing = tmp_path / "test_genotyping.vcf", tmp_path / "test_phasing.vcf"
    r = subprocess.run([str(HOST_TEST), "demo", str(DEMO), str(tmp_path / "preprocessing"), str(out), str(phasing)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert without_date(out.read_text()) == expected()
    check_phasing(phasing.read_text())
