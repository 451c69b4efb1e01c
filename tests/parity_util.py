"""Comparison rules of the parity tests (HIP path vs CPU oracle / golden vectors).

Likelihood tolerance = BASELINE.json north_star: 1e-6 RELATIVE on every unnormalised
genotype likelihood (the device returns lik*2^lik_exp, rebuilt as long double).  Entries
more than TINY_DECADES = 200 decades below their variant's largest bin are outside what the
device's fp64 columns resolve exactly (a stored column carries the scale of its
own emission-weighted sum, so entries that far down can be sub-normal) and are compared
absolutely against that scale — a 1e-200 share of a variant's likelihood mass cannot move a
genotype call or quality.  Genotype calls must be identical.
"""
import numpy as np

LD = np.longdouble
REL_TOL = 1e-6
TINY_DECADES = 200


def rel_errors(batch, got, ref):
    got = np.asarray(got, dtype=LD)
    ref = np.asarray(ref, dtype=LD)
    geno_off = batch.geno_off.astype(np.int64)
    G = np.diff(geno_off)
    scale = np.zeros(batch.n_variants, dtype=LD)
    nz = G > 0
    if ref.size:
        mx = np.maximum.reduceat(np.concatenate([np.abs(ref), np.zeros(1, LD)]), geno_off[:-1])
        scale = np.where(nz, mx, LD(0))
    scale_e = np.repeat(scale, G)
    denom = np.maximum(np.abs(got), np.abs(ref))
    tiny = denom <= scale_e * LD(10.0) ** LD(-TINY_DECADES)
    rel = np.where(denom > 0, np.abs(got - ref) / np.where(denom > 0, denom, LD(1)), LD(0))
    rel = np.where(tiny, np.abs(got - ref) / np.where(scale_e > 0, scale_e, LD(1)), rel)
    return rel


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
