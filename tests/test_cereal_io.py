"""The reference's cereal-binary UniqueKmers archives without cereal (pangenie_amd/cereal_io.py; the C++ host
reader is exercised by tests/cpp/test_host.cpp): the reference's own fixtures tests/data/region*_UniqueKmersList.cereal
(kept as data under tests/golden/) parse, re-serialise byte for byte, flatten into batches the oracle
accepts, and — on a GPU — give HIP results identical to the oracle's down to the VCF sample column
(`GT:GQ:GL:KC`, reference src/graph.cpp:217-273; tests/CommandsTest.cpp:59-93 builds the same strings from a
directly constructed HMM)."""
from pathlib import Path

import numpy as np
import pytest

from pangenie_amd import cereal_io
from pangenie_amd.genotyping_result import results_from_flat, vcf_sample_field
from pangenie_amd.panel import flatten

GOLDEN = Path(__file__).resolve().parent / "golden"
# tests/CommandsTest.cpp:20-35,59-60: k-mer abundance peak 18, regularization 0.01, effective_N 1e-5, recombrate 1.26
TABLE = (18 // 4, 18 * 4, 2 * 18, 0.01)
PARAMS = (1.26, False, 1e-5)


def oracle_fields(batch, uks, defined):
    from oracle import pyoracle as orc
    ref = orc.genotype_contig(batch, orc.OracleTable(*TABLE), orc.make_params(*PARAMS))
    res = results_from_flat(batch, ref.lik, ref.kept, ref.allele_present, ref.n_kmers, ref.coverage)
    out = []
    for g, d, u in zip(res, defined, uks):
        g.normalize()
        out.append(vcf_sample_field(g, d, len(u.alleles)))
    return ref, out


def test_reference_fixtures_parse_and_roundtrip():
    for name, paths, alleles in (("region_UniqueKmersList.cereal", 215, (44, 45)), ("region2_UniqueKmersList.cereal", 6, (2, 2))):
        raw = (GOLDEN / name).read_bytes()
        m = cereal_io.loads(raw)
        assert m.kmersize == 31 and list(m.unique_kmers) == ["chr1"] and m.add_reference
        uks = m.unique_kmers["chr1"]
        assert [u.variant_pos for u in uks] == [138, 207] and [len(u.path_to_allele) for u in uks] == [paths, paths]
        assert tuple(len(u.alleles) for u in uks) == alleles and not any(u.biallelic for u in uks)
        assert cereal_io.dumps(m) == raw
    u = cereal_io.load(GOLDEN / "region_UniqueKmersList.cereal").unique_kmers["chr1"][0]
    # coverage 30, 62 k-mers, alleles 0 and 1 carry 31 k-mers each, alleles 2..43 none and undefined
    assert u.local_coverage == 30.0 and len(u.kmer_to_count) == 62
    assert bin(u.alleles[0][1]).count("1") == 31 and bin(u.alleles[1][1]).count("1") == 31 and u.alleles[1][0] == 31
    assert all(u.alleles[a][1] == 0 and u.alleles[a][2] for a in range(2, 44))


def test_fixtures_through_the_oracle_give_wellformed_records():
    for name, defined in (("region_UniqueKmersList.cereal", [[0, 1], [0, 1, 2]]), ("region2_UniqueKmersList.cereal", [[0, 1], [0, 1]])):
        uks = cereal_io.load(GOLDEN / name).unique_kmers["chr1"]
        batch = flatten(uks)
        ref, fields = oracle_fields(batch, uks, defined)
        assert ref.n_columns == 2
        for f, d in zip(fields, defined):
            gt, gq, gl, kc = f.split(":")
            assert len(gl.split(",")) == len(d) * (len(d) + 1) // 2 and kc in ("30", "34")
            assert gt == "." or (int(gq) >= 0 and all(int(a) < len(d) for a in gt.split("/")))


@pytest.mark.gpu
def test_fixtures_hip_vs_oracle_down_to_the_vcf_column():
    from pangenie_amd import hmm
    from tests.parity_util import assert_parity
    for name, defined in (("region_UniqueKmersList.cereal", [[0, 1], [0, 1, 2]]), ("region2_UniqueKmersList.cereal", [[0, 1], [0, 1]])):
        uks = cereal_io.load(GOLDEN / name).unique_kmers["chr1"]
        batch = flatten(uks)
        ref, want = oracle_fields(batch, uks, defined)
        res = hmm.genotype_contig(batch, hmm.ProbabilityTable(*TABLE), hmm.make_params(*PARAMS))
        assert_parity(batch, res, ref)
        got = []
        for g, d, u in zip(res.genotyping_results(), defined, uks):
            g.normalize()
            got.append(vcf_sample_field(g, d, len(u.alleles)))
        assert got == want, (got, want)


def test_results_archive_cpp_writer_python_reader(tmp_path):
    """`-w` Results archive (<out>_genotyping.cereal, reference src/commands.cpp:59-72, :511-516): the C++ writer
    (pangenie_amd/host/cereal_io.cpp) and the independent Python reader / writer agree byte for byte, and the values
    come back as 80-bit long doubles (one of them below the double range)."""
    import subprocess
    from pangenie_amd.build import build_host, HOST_TEST
    build_host()
    path = tmp_path / "x_genotyping.cereal"
    subprocess.run([str(HOST_TEST), "dump-results", str(path)], check=True)
    raw = path.read_bytes()
    res = cereal_io.loads_results(raw)
    assert sorted(res.result) == ["chr1", "chr10"] and [len(res.result[c]) for c in ("chr1", "chr10")] == [2, 1]
    a, b = res.result["chr1"]
    LD = np.longdouble
    assert a.genotype_to_likelihood == {(0, 0): LD(0.5), (0, 1): LD(0.25), (1, 1): LD(1) / LD(3)}
    assert (a.haplotype_1, a.haplotype_2, a.local_coverage, a.unique_kmers) == (1, 0, 27, 20)
    assert b.contains_no_likelihoods() and b.local_coverage == 3
    c = res.result["chr10"][0]
    assert c.genotype_to_likelihood[(2, 5)] == LD("1e-4000") and c.genotype_to_likelihood[(2, 5)] > 0
    assert (c.haplotype_1, c.haplotype_2, c.unique_kmers) == (5, 2, 301)
    assert res.runtimes == {"chr1": 1.5, "chr10": 0.125}
    assert cereal_io.dumps_results(res) == raw


@pytest.mark.gpu
def test_genotyped_vcf_of_the_index_fixture_vs_oracle(tmp_path):
    """run_genotype_command on the reference's own index fixture (tests/CommandsTest.cpp:18-93), by the C++ host over
    the device: index + Graph archives -> read k-mer counts -> HMM on the GPU -> Graph::genotypes_records.  The fixed
    columns come from tests/data/region.vcf (what the graph was built from), the sample columns must be the ones the
    CPU oracle gives on the reference's counted archive."""
    import subprocess
    from pangenie_amd.build import build_host, HOST_TEST
    build_host()
    out = tmp_path / "fixture_genotyping.vcf"
    subprocess.run([str(HOST_TEST), "write-vcf", str(GOLDEN), str(out)], check=True)
    lines = out.read_text().splitlines()
    header, records = [l for l in lines if l.startswith("#")], [l.split("\t") for l in lines if not l.startswith("#")]
    assert header[0] == "##fileformat=VCFv4.2" and header[1].startswith("##fileDate=20") and len(header) == 12
    assert header[-1] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample"
    assert [r[:7] for r in records] == [["chr1", "139", ".", "T", "C", ".", "PASS"], ["chr1", "208", ".", "TG", "CG,CA", ".", "PASS"]]
    assert all(r[8] == "GT:GQ:GL:KC" for r in records)
    info = [dict(kv.split("=", 1) for kv in r[7].split(";")) for r in records]
    assert [i["MA"] for i in info] == ["42", "42"] and info[0]["UK"] == "62"
    assert info[0]["ID"] == "chr1-49638-SNV->50027902>50027904>50027905-1"
    assert info[1]["ID"] == "chr1-49707-SNV->50027911>50027913>50027914-1,chr1-49707-COMPLEX->50027911>50027913>50027915>50027916-2"
    uks = cereal_io.load(GOLDEN / "region_UniqueKmersList.cereal").unique_kmers["chr1"]
    batch = flatten(uks)
    # allele frequencies over the 214 panel paths (the reference path, index 0 of 215, left out)
    pa = batch.path_allele.reshape(2, 215)
    want_af = [["%.6g" % np.float32(np.float32((pa[0] == 1).sum()) / np.float32(214))],
               ["%.6g" % np.float32(np.float32((pa[1] == a).sum()) / np.float32(214)) for a in (1, 2)]]
    assert [i["AF"].split(",") for i in info] == want_af
    _, want = oracle_fields(batch, uks, [[0, 1], [0, 1, 2]])
    assert [r[9] for r in records] == want
