"""Which kernels run for which shape: pg_job_plan (the planner of pg_shim.cpp, as text) for one small job per shape.
Run on a GPU box (jobs are built, not run): python tools/plan_table.py > profiles/r06_plan_table.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel, synthetic_sample_counts

table = hmm.ProbabilityTable(*default_table_args())
V = 96


def cohort(H, chains, contigs=8, phasing=False, **kw):
    index = [synthetic_panel(V, H, 20, seed=10 + i, **kw) for i in range(contigs)]
    kcs, covs = zip(*[synthetic_sample_counts(ix, seed=i) for i, ix in enumerate(index)])
    job = hmm.Job.cohort(index, [(list(kcs), list(covs))] * (chains // contigs), table, hmm.make_params(1.26, False, 1e-5, run_phasing=phasing) if phasing else hmm.make_params(1.26, False, 1e-5))
    text = job.plan()
    job.close()
    return text


SHAPES = [
    ("whole genome, 24 chromosomes x 64 paths, biallelic (BASELINE configs[3]; < 64 chains: chunked)", dict(H=64, chains=24, contigs=24)),
    ("one contig x 64 paths, biallelic (a one-shot HMM constructor call)", dict(H=64, chains=1, contigs=1)),
    ("few chains x 64 paths with 3-5-allele objects", dict(H=64, chains=8, multiallelic_frac=0.2)),
    ("few chains x 128 paths with 3-5-allele objects (BASELINE configs[4], one GPU's share)", dict(H=128, chains=8, multiallelic_frac=0.2)),
    ("few chains x 128 paths with objects of more than 5 alleles (wide columns)", dict(H=128, chains=8, multiallelic_frac=0.2, wide_frac=0.05)),
    ("cohort, 512 chains x 64 paths, biallelic", dict(H=64, chains=512)),
    ("cohort, 256 chains x 64 paths with 3-5-allele objects", dict(H=64, chains=256, multiallelic_frac=0.2)),
    ("cohort, 256 chains x 64 paths with wide columns", dict(H=64, chains=256, multiallelic_frac=0.2, wide_frac=0.02)),
    ("cohort, 128 chains x 128 paths with 3-5-allele objects", dict(H=128, chains=128, multiallelic_frac=0.2)),
    ("cohort, 4096 chains x 16 paths (15 sampled + reference), biallelic", dict(H=16, chains=4096)),
    ("cohort, 4096 chains x 16 paths with 3-5-allele objects (the default production shape)", dict(H=16, chains=4096, multiallelic_frac=0.2)),
    ("cohort, 4096 chains x 16 paths with wide columns", dict(H=16, chains=4096, multiallelic_frac=0.2, wide_frac=0.02)),
    ("cohort, 128 chains x 16 paths with 3-5-allele objects (fewer than 320 such chains)", dict(H=16, chains=128, multiallelic_frac=0.2)),
    ("few chains x 16 paths", dict(H=16, chains=8, multiallelic_frac=0.2)),
    ("cohort, 1024 chains x 17 paths (-x 16 + reference: pads to 32)", dict(H=17, chains=1024, multiallelic_frac=0.2)),
    ("few chains x 30 paths, run_phasing (the reference's phasing panel)", dict(H=30, chains=8, phasing=True)),
    ("few chains x 300 paths (generic kernel)", dict(H=300, chains=8, contigs=8)),
]
for title, kw in SHAPES:
    print("## " + title)
    try:
        print(cohort(**kw))
    except Exception as e:   # a shape this build refuses: say so
        print("  (refused: %s)\n" % e)
