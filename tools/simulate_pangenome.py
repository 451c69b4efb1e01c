#!/usr/bin/env python3
"""A synthetic pangenome and a sample sequenced from it — inputs for an end-to-end run of the pieces around the device path
(tools/pipeline_check.sh): index builder -> graph-only k-mer counts -> counts into the index -> HMM on the device -> VCF,
scored against the sample's true genotypes.

  panel  <length> <records> <samples> <seed> <prefix>      ->  <prefix>.fa, <prefix>.vcf (phased, every ALT carried)
  sample <prefix> <coverage> <seed>                        ->  <prefix>_reads.fa, <prefix>_truth.tsv
  score  <prefix>_truth.tsv <genotyped.vcf>                ->  concordance of the unphased genotypes

The sample is NOT one of the panel's: each of its two haplotypes is a mosaic of panel haplotypes (a switch every ~200
records), so the genotyper has to find it through the HMM.  Reads: 150 bases, either strand, 0.3 % substitutions."""
import sys

import numpy as np

READ = 150


def panel(length, records, samples, seed, prefix):
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, length, dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    text = letters[ref].tobytes().decode()
    with open(prefix + ".fa", "w") as f:
        f.write(">chr1\n")
        for i in range(0, length, 80):
            f.write(text[i:i + 80] + "\n")
    positions = np.sort(rng.choice(np.arange(300, length - 300, 5), records, replace=False))
    with open(prefix + ".vcf", "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"s{i}" for i in range(samples)) + "\n")
        last_end = 0
        for p in positions:
            p = int(p)
            if p <= last_end:
                continue
            kind = rng.random()
            if kind < 0.8:      # SNP
                r = text[p]
                alts = ["ACGT".replace(r, "")[int(rng.integers(0, 3))]]
            elif kind < 0.9:    # deletion
                r = text[p:p + int(rng.integers(2, 21))]
                alts = [r[0]]
            else:               # insertion(s), up to three alleles
                r = text[p]
                alts = list(dict.fromkeys(r + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 61)))) for _ in range(int(rng.integers(1, 4)))))
            af = rng.random() * 0.9 + 0.05
            hap = np.where(rng.random(2 * samples) < af, rng.integers(1, len(alts) + 1, 2 * samples), 0)
            for a in range(1, len(alts) + 1):       # every ALT on some haplotype
                if not (hap == a).any():
                    hap[int(rng.integers(0, 2 * samples))] = a
            gts = "\t".join(f"{hap[2 * s]}|{hap[2 * s + 1]}" for s in range(samples))
            f.write(f"chr1\t{p + 1}\t.\t{r}\t{','.join(alts)}\t.\tPASS\t.\tGT\t{gts}\n")
            last_end = p + len(r) + 1


def read_panel(prefix):
    ref = "".join(l.strip() for l in open(prefix + ".fa") if not l.startswith(">"))
    recs = []
    for l in open(prefix + ".vcf"):
        if l.startswith("#"):
            continue
        c = l.rstrip("\n").split("\t")
        recs.append((int(c[1]) - 1, c[3], c[4].split(","), [tuple(int(x) for x in g.split("|")) for g in c[9:]]))
    return ref, recs


def sample(prefix, coverage, seed):
    rng = np.random.default_rng(seed)
    ref, recs = read_panel(prefix)
    n_hap = 2 * len(recs[0][3])
    truth = []
    haps = []
    for h in range(2):
        src = int(rng.integers(0, n_hap))
        out, at, alleles = [], 0, []
        for pos, r, alts, gts in recs:
            if rng.random() < 1 / 200:
                src = int(rng.integers(0, n_hap))
            a = gts[src // 2][src % 2]
            alleles.append(a)
            out.append(ref[at:pos])
            out.append(r if a == 0 else alts[a - 1])
            at = pos + len(r)
        out.append(ref[at:])
        haps.append(np.frombuffer("".join(out).encode(), dtype=np.uint8))
        truth.append(alleles)
    with open(prefix + "_truth.tsv", "w") as f:
        for (pos, r, alts, _), a, b in zip(recs, truth[0], truth[1]):
            f.write(f"chr1\t{pos + 1}\t{a}\t{b}\n")
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    header = np.frombuffer(b">r\n", dtype=np.uint8)
    with open(prefix + "_reads.fa", "wb") as f:
        for hap in haps:
            n = int(coverage / 2 * len(hap) / READ)
            for lo in range(0, n, 200000):
                m = min(200000, n - lo)
                start = rng.integers(0, len(hap) - READ, m)
                reads = hap[start[:, None] + np.arange(READ)[None, :]]
                flip = rng.random(m) < 0.5
                reads[flip] = comp[reads[flip][:, ::-1]]
                err = rng.random(reads.shape) < 0.003
                reads[err] = letters[rng.integers(0, 4, int(err.sum()))]
                block = np.empty((m, len(header) + READ + 1), dtype=np.uint8)
                block[:, :len(header)] = header
                block[:, len(header):-1] = reads
                block[:, -1] = ord("\n")
                f.write(block.tobytes())


def score(truth_tsv, vcf):
    truth = {}
    for l in open(truth_tsv):
        c = l.split()
        truth[(c[0], int(c[1]))] = tuple(sorted((int(c[2]), int(c[3]))))
    total = same = untyped = nonref = nonref_same = 0
    for l in open(vcf):
        if l.startswith("#"):
            continue
        c = l.rstrip("\n").split("\t")
        want = truth.get((c[0], int(c[1])))
        if want is None:
            continue
        total += 1
        gt = c[9].split(":")[0]
        if "." in gt:
            untyped += 1
            continue
        got = tuple(sorted(int(x) for x in gt.split("/")))
        same += got == want
        if want != (0, 0):
            nonref += 1
            nonref_same += got == want
    return {"records": total, "truth": len(truth), "concordance": same / max(total, 1), "nonref": nonref,
            "nonref_concordance": nonref_same / max(nonref, 1), "untyped": untyped}


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "panel":
        panel(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
    elif cmd == "sample":
        sample(sys.argv[2], float(sys.argv[3]), int(sys.argv[4]))
    elif cmd == "score":
        r = score(sys.argv[2], sys.argv[3])
        print(f"records {r['records']} of {r['truth']} in the truth; genotype concordance {r['concordance']:.4f} "
              f"(non-reference genotypes {r['nonref_concordance']:.4f} of {r['nonref']}); untyped {r['untyped']}")
    else:
        sys.exit(__doc__)
