"""Comparison rules of the parity tests (HIP path vs CPU oracle / golden vectors).

Likelihood tolerance = BASELINE.json north_star: 1e-6 RELATIVE on EVERY unnormalised genotype
likelihood, however far below its variant's largest bin it lies (the device returns one fp64
mantissa and one int32 exponent per bin, rebuilt as long double).  The only slack is the
reference's own resolution: its 80-bit long doubles turn denormal below 2^-16382 and carry an
absolute rounding quantum of 2^-16445 there, so a bin that is a sum of up to H^2 such terms is
compared with an absolute allowance of (H^2 + 8) quanta — values of that size print as `-inf`/0
in the reference's own output.  Genotype calls must be identical.
"""
import numpy as np

LD = np.longdouble
REL_TOL = 1e-6
LD_QUANTUM = np.ldexp(LD(1), -16445)  # smallest positive (denormal) x87 long double


def rel_errors(batch, got, ref):
    """Relative error of every bin; bins within the reference's denormal rounding noise count as 0."""
    got = np.asarray(got, dtype=LD)
    ref = np.asarray(ref, dtype=LD)
    denom = np.maximum(np.abs(got), np.abs(ref))
    err = np.abs(got - ref)
    rel = np.where(denom > 0, err / np.where(denom > 0, denom, LD(1)), LD(0))
    H = LD(batch.n_paths)
    noise = err <= LD_QUANTUM * (H * H + LD(8))
    return np.where(noise, LD(0), rel)


def calls(batch, lik, tie=1e-10):
    """GenotypingResult::normalize + get_likeliest_genotype for every variant
    (reference src/genotypingresult.cpp:149-210): index of the winning bin or -1."""
    geno_off = batch.geno_off.astype(np.int64)
    lik = np.asarray(lik, dtype=LD)
    out = np.full(batch.n_variants, -1, dtype=np.int64)
    for v in range(batch.n_variants):
        seg = lik[geno_off[v]:geno_off[v + 1]]
        if seg.size == 0:
            continue
        s = seg.sum()
        if s > 0:
            seg = seg / s
        best = int(np.argmax(seg))
        bv = seg[best]
        if not bv > 0:
            continue
        others = np.delete(seg, best)
        if others.size and np.any(np.abs(others - bv) < tie):
            continue
        out[v] = best
    return out


def assert_parity(batch, res, ref, tol=REL_TOL, check_calls=True):
    assert res.n_columns == ref.n_columns
    assert (res.kept == ref.kept).all()
    assert (res.allele_present == ref.allele_present).all()
    assert (res.n_kmers == ref.n_kmers).all() and (res.coverage == ref.coverage).all()
    got = res.likelihoods_ld()
    rel = rel_errors(batch, got, ref.lik)
    worst = float(rel.max()) if rel.size else 0.0
    assert worst < tol, f"max relative likelihood error {worst:.3e} >= {tol}"
    if check_calls:
        a, b = calls(batch, got), calls(batch, ref.lik)
        assert (a == b).all(), f"{int((a != b).sum())} genotype calls differ"
    return worst


def log_parity(line: str) -> None:
    """Worst-error lines of the whole-chain parity tests: printed (pytest -s) and, when PG_PARITY_LOG names a file,
    appended to it (how profiles/r05_whole_chain_parity.txt was made on the GPU box)."""
    import os
    print(line)
    path = os.environ.get("PG_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")
