"""The pieces either side of the device path strung together on a small synthetic pangenome (tools/simulate_pangenome.py:
SNPs, deletions, multi-allelic insertions; the sample is a mosaic of panel haplotypes, 25x reads with wrong letters):
index builder -> graph-only k-mer counts -> abundance peak -> counts into the index -> [haplotype sampling when the panel has
more than 100 paths] -> HMM -> VCF.  On the CPU the HMM (and the sampler) are the oracle's and the genotypes are scored
against the truth; on the GPU the device's VCF must be the oracle twin's, line for line (GT, GQ, four-digit GL, KC of every
record; see assert_same_vcf for the one allowance at the last long double step below 1).  The full-size runs of the same scripts are in profiles/r03_pipeline_check*.txt."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import pipeline_cpu_check   # noqa: E402
import simulate_pangenome   # noqa: E402
from pangenie_amd.build import build_host, HOST_TEST   # noqa: E402


def prepared(tmp_path, samples, seed, records=600):
    build_host()
    p = str(tmp_path / "sim")
    simulate_pangenome.panel(300000, records, samples, seed, p)
    simulate_pangenome.sample(p, 25, seed + 1)
    subprocess.run([str(HOST_TEST), "index", p + ".fa", p + ".vcf", p + "_idx", "31", "2"], check=True, capture_output=True, timeout=300)
    return p


def without_date(path):
    return [l for l in Path(path).read_text().splitlines() if not l.startswith("##fileDate")]


def assert_same_vcf(got_path, want_path):
    """Line for line the same text, with one allowance in the sample column: the oracle's likelihoods are long doubles, the
    device's are fp64 values normalised in long double — a likelihood within a few long double steps of 1 (log10 of the order of
    1e-19) can come out as exactly 1 on one side, and the genotype quality, -10 log10 of what is missing to 1, then reads
    10000 instead of something above 150.  So: GT and KC identical, every GL equal to the four printed digits or both below
    1e-15 in magnitude, GQ identical or both at least 150 — and no more than 1 record in 200 (at least 2) may need an allowance."""
    got, want = without_date(got_path), without_date(want_path)
    assert len(got) == len(want)
    allowed = 0
    for g, w in zip(got, want):
        if g == w:
            continue
        g, w = g.split("\t"), w.split("\t")
        assert g[:9] == w[:9], (g, w)
        (ggt, ggq, ggl, gkc), (wgt, wgq, wgl, wkc) = g[9].split(":"), w[9].split(":")
        assert ggt == wgt and gkc == wkc, (g, w)
        assert ggq == wgq or (min(int(ggq), int(wgq)) >= 150), (g, w)
        for a, b in zip(ggl.split(","), wgl.split(",")):
            assert a == b or (abs(float(a)) < 1e-15 and abs(float(b)) < 1e-15), (g, w)
        allowed += 1
    assert allowed <= max(2, len(want) // 200), allowed


@pytest.mark.parametrize("samples", [8, 54])
def test_pipeline_with_the_oracle_scores_against_the_truth(tmp_path, samples):
    p = prepared(tmp_path, samples, 100 + samples)
    peak = pipeline_cpu_check.main(p + "_idx", p + "_reads.fa", p + "_cpu.vcf")
    assert 12 <= peak <= 26   # 25x reads of 150 bases, k = 31: about 20 windows over an error-free k-mer
    r = simulate_pangenome.score(p + "_truth.tsv", p + "_cpu.vcf")
    assert r["records"] == r["truth"] > 550 and r["untyped"] <= 2
    assert r["concordance"] >= 0.99 and r["nonref_concordance"] >= 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("samples", [8, 54])
def test_pipeline_on_the_device_writes_the_oracle_twins_vcf(tmp_path, samples):
    p = prepared(tmp_path, samples, 100 + samples)
    pipeline_cpu_check.main(p + "_idx", p + "_reads.fa", p + "_cpu.vcf")
    r = subprocess.run([str(HOST_TEST), "genotype", p + "_idx", p + "_reads.fa", p + "_gpu.vcf", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert_same_vcf(p + "_gpu.vcf", p + "_cpu.vcf")


@pytest.mark.gpu
def test_pipeline_on_the_device_dense_panel(tmp_path):
    """2400 records in 300 kb: most bubbles merge several records (many alleles per column, reference stretches shorter
    than k between them); the device's VCF is still the oracle twin's."""
    p = prepared(tmp_path, 10, 31, records=2400)
    pipeline_cpu_check.main(p + "_idx", p + "_reads.fa", p + "_cpu.vcf")
    r = subprocess.run([str(HOST_TEST), "genotype", p + "_idx", p + "_reads.fa", p + "_gpu.vcf", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert_same_vcf(p + "_gpu.vcf", p + "_cpu.vcf")
    score = simulate_pangenome.score(p + "_truth.tsv", p + "_gpu.vcf")
    assert score["records"] == score["truth"] > 2300 and score["concordance"] >= 0.99


@pytest.mark.gpu
def test_cohort_of_three_samples_in_one_job_writes_each_samples_own_vcf(tmp_path):
    """pangenie::genotype_cohort (pg_cohort_new: the index on the device once, every (sample, chromosome) a chain of one
    job, one probability table whose box spans the samples' peaks): three samples of different depth against the same
    index must each get the VCF the oracle twin writes for that sample alone with its own table."""
    p = prepared(tmp_path, 10, 77)                      # (leaves a 25x sample in <p>_reads.fa)
    reads = [p + "_reads_a.fa"]
    Path(p + "_reads.fa").rename(reads[0])
    for name, coverage, seed in (("b", 14, 500), ("c", 36, 501)):
        simulate_pangenome.sample(p, coverage, seed)   # (writes <p>_reads.fa)
        reads.append(p + f"_reads_{name}.fa")
        Path(p + "_reads.fa").rename(reads[-1])
    peaks = [pipeline_cpu_check.main(p + "_idx", r, p + f"_cpu{i}.vcf") for i, r in enumerate(reads)]
    assert len(set(peaks)) == 3
    r = subprocess.run([str(HOST_TEST), "cohort", p + "_idx", p + "_cohort", "4"] + reads, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for i in range(3):
        assert_same_vcf(p + f"_cohort_{i}.vcf", p + f"_cpu{i}.vcf")
