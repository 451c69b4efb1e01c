"""Flat contig batches (the SoA of include/pangenie_hmm.h) and how to make them.

* `ContigBatch`     — numpy arrays + the ctypes view handed to the C ABI.
* `UniqueKmers`     — small Python mirror of the reference's BiallelicUniqueKmers /
                      MultiallelicUniqueKmers (src/biallelicuniquekmers.cpp,
                      src/multiallelicuniquekmers.cpp, src/kmerpath.cpp) used to
                      express the reference's unit-test fixtures as data.
* `flatten`         — UniqueKmers list (+ only_paths) -> ContigBatch; restates
                      ColumnIndexer's path selection (src/columnindexer.cpp:8-33).
* `synthetic_panel` — deterministic synthetic panels of BASELINE.json's shapes
                      (SURVEY.md §8(d)).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Iterable, Sequence

import numpy as np

from ._lib import PgContigBatch, u8p, u16p, u32p, u64p


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


@dataclass
class ContigBatch:
    n_paths: int
    variant_pos: np.ndarray      # u64 [V]
    coverage: np.ndarray         # u16 [V]
    kmer_off: np.ndarray         # u32 [V+1]
    kmer_count: np.ndarray       # u16 [sumK]
    allele_off: np.ndarray       # u32 [V+1]
    allele_id: np.ndarray        # u16 [sumA]
    allele_flags: np.ndarray     # u8  [sumA]
    allele_kmer_off: np.ndarray  # u16 [sumA]
    allele_kmer_mask: np.ndarray # u32 [sumA]
    path_allele: np.ndarray      # u16 [V*H]
    _c: PgContigBatch | None = field(default=None, repr=False)

    def __post_init__(self):
        def fix(a, dt, n=None):
            a = np.ascontiguousarray(a, dtype=dt)
            # ctypes needs a valid pointer even for empty arrays
            if a.size == 0:
                a = np.zeros(1, dtype=dt)[:0].copy()
            return a
        self.variant_pos = fix(self.variant_pos, np.uint64)
        self.coverage = fix(self.coverage, np.uint16)
        self.kmer_off = fix(self.kmer_off, np.uint32)
        self.kmer_count = fix(self.kmer_count, np.uint16)
        self.allele_off = fix(self.allele_off, np.uint32)
        self.allele_id = fix(self.allele_id, np.uint16)
        self.allele_flags = fix(self.allele_flags, np.uint8)
        self.allele_kmer_off = fix(self.allele_kmer_off, np.uint16)
        self.allele_kmer_mask = fix(self.allele_kmer_mask, np.uint32)
        self.path_allele = fix(self.path_allele, np.uint16)
        V = self.n_variants
        assert self.kmer_off.shape == (V + 1,) and self.allele_off.shape == (V + 1,)
        assert self.path_allele.size == V * self.n_paths

    @property
    def n_variants(self) -> int:
        return int(self.variant_pos.shape[0])

    @property
    def geno_off(self) -> np.ndarray:
        A = np.diff(self.allele_off.astype(np.uint64))
        out = np.zeros(self.n_variants + 1, dtype=np.uint64)
        np.cumsum(A * (A + 1) // 2, out=out[1:])
        return out

    def as_c(self) -> PgContigBatch:
        if self._c is None:
            c = PgContigBatch()
            c.n_variants = self.n_variants
            c.n_paths = self.n_paths
            c.variant_pos = _ptr(self.variant_pos, u64p)
            c.coverage = _ptr(self.coverage, u16p)
            c.kmer_off = _ptr(self.kmer_off, u32p)
            c.kmer_count = _ptr(self.kmer_count, u16p)
            c.allele_off = _ptr(self.allele_off, u32p)
            c.allele_id = _ptr(self.allele_id, u16p)
            c.allele_flags = _ptr(self.allele_flags, u8p)
            c.allele_kmer_off = _ptr(self.allele_kmer_off, u16p)
            c.allele_kmer_mask = _ptr(self.allele_kmer_mask, u32p)
            c.path_allele = _ptr(self.path_allele, u16p)
            self._c = c
        return self._c

    def slice(self, lo: int, hi: int) -> "ContigBatch":
        """Variants [lo, hi) as their own batch (used for bounded CPU-baseline samples)."""
        k0, k1 = int(self.kmer_off[lo]), int(self.kmer_off[hi])
        a0, a1 = int(self.allele_off[lo]), int(self.allele_off[hi])
        H = self.n_paths
        return ContigBatch(
            n_paths=H,
            variant_pos=self.variant_pos[lo:hi].copy(),
            coverage=self.coverage[lo:hi].copy(),
            kmer_off=(self.kmer_off[lo:hi + 1] - k0).astype(np.uint32),
            kmer_count=self.kmer_count[k0:k1].copy(),
            allele_off=(self.allele_off[lo:hi + 1] - a0).astype(np.uint32),
            allele_id=self.allele_id[a0:a1].copy(),
            allele_flags=self.allele_flags[a0:a1].copy(),
            allele_kmer_off=self.allele_kmer_off[a0:a1].copy(),
            allele_kmer_mask=self.allele_kmer_mask[a0:a1].copy(),
            path_allele=self.path_allele[lo * H:hi * H].copy(),
        )

    def with_counts(self, kmer_count: np.ndarray, coverage: np.ndarray) -> "ContigBatch":
        """The same index with another sample's read k-mer counts / local coverage (what
        fill_read_kmercounts changes per sample, reference src/commands.cpp:118-138).  Index arrays
        are shared, not copied."""
        kc = np.ascontiguousarray(kmer_count, np.uint16)
        cv = np.ascontiguousarray(coverage, np.uint16)
        assert kc.shape == self.kmer_count.shape and cv.shape == self.coverage.shape
        return ContigBatch(self.n_paths, self.variant_pos, cv, self.kmer_off, kc, self.allele_off, self.allele_id,
                           self.allele_flags, self.allele_kmer_off, self.allele_kmer_mask, self.path_allele)

    def update_paths(self, sampled_paths: np.ndarray) -> "ContigBatch":
        """UniqueKmers::update_paths on every variant at once (reference
        src/haplotypesampler.cpp:296-309, src/multiallelicuniquekmers.cpp:195-232): the panel keeps
        path j = sampled_paths[j][v] at variant v, the alleles those paths carry and the k-mers that lie on
        at least one kept allele (re-indexed in their old order).  sampled_paths: [S, V] path ids."""
        sp = np.ascontiguousarray(sampled_paths, dtype=np.int64)
        V, H = self.n_variants, self.n_paths
        S = sp.shape[0]
        assert sp.shape == (S, V) and (V == 0 or (sp.min() >= 0 and sp.max() < H))
        pa = self.path_allele.reshape(V, H)
        new_pa = pa[np.arange(V)[None, :], sp].T.copy()  # [V, S]
        A = np.diff(self.allele_off.astype(np.int64))
        K = np.diff(self.kmer_off.astype(np.int64))
        slot_v = np.repeat(np.arange(V), A)
        keep_slot = (new_pa[slot_v] == self.allele_id[:, None]).any(axis=1)
        bits = ((self.allele_kmer_mask[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool)
        local = self.allele_kmer_off.astype(np.int64)[:, None] + np.arange(32)[None, :]
        bits &= local < K[slot_v][:, None]  # bits beyond the variant's k-mers do not exist
        gidx = self.kmer_off.astype(np.int64)[slot_v][:, None] + local
        keep_kmer = np.zeros(int(self.kmer_off[-1]), bool)
        sel = bits & keep_slot[:, None]
        keep_kmer[gidx[sel]] = True
        csum = np.concatenate([[0], np.cumsum(keep_kmer)])
        new_koff = csum[self.kmer_off.astype(np.int64)]
        new_local = np.where(sel, csum[np.minimum(gidx, keep_kmer.size - 1 if keep_kmer.size else 0)] - new_koff[slot_v][:, None], 1 << 30) if keep_kmer.size else np.full(sel.shape, 1 << 30)
        first = new_local.min(axis=1)  # KmerPath offset = first index set (src/kmerpath.cpp:13-17)
        has = first < (1 << 30)
        first = np.where(has, first, 0)
        shift = np.where(sel, new_local - first[:, None], 0)
        assert shift.max(initial=0) < 32
        new_mask = (np.where(sel, np.uint64(1) << shift.astype(np.uint64), np.uint64(0))).sum(axis=1).astype(np.uint32)
        ks = keep_slot
        new_aoff = np.concatenate([[0], np.cumsum(np.bincount(slot_v[ks], minlength=V))]).astype(np.uint32)
        return ContigBatch(S, self.variant_pos.copy(), self.coverage.copy(), new_koff.astype(np.uint32), self.kmer_count[keep_kmer],
                           new_aoff, self.allele_id[ks], self.allele_flags[ks], first[ks].astype(np.uint16), new_mask[ks],
                           new_pa.reshape(-1))

    def nbytes(self) -> int:
        return sum(getattr(self, f).nbytes for f in (
            "variant_pos", "coverage", "kmer_off", "kmer_count", "allele_off", "allele_id",
            "allele_flags", "allele_kmer_off", "allele_kmer_mask", "path_allele"))


# --------------------------------------------------------------------------- #
#  Python mirror of the reference UniqueKmers objects (fixture construction)
# --------------------------------------------------------------------------- #
class UniqueKmers:
    """Behavioural mirror of {Bi,Multi}allelicUniqueKmers for building fixtures.

    reference: src/biallelicuniquekmers.cpp:8-68,107-118,195-220,
               src/multiallelicuniquekmers.cpp:7-59,98-109,168-193,
               src/kmerpath.cpp:13-48 (window = 32; kmerpath16.cpp: window = 16).
    """

    def __init__(self, variant_position: int, path_to_allele: Sequence[int], biallelic: bool):
        self.biallelic = biallelic
        self.window = 16 if biallelic else 32
        self.variant_pos = int(variant_position)
        self.local_coverage = 0
        self.kmer_to_count: list[int] = []
        self.path_to_allele = [int(a) for a in path_to_allele]
        # allele -> [offset, mask, is_undefined]
        self.alleles: dict[int, list] = {}
        for a in self.path_to_allele:
            if biallelic and a not in (0, 1):
                raise RuntimeError("BiallelicUniqueKmers: provided alleles need to be either 0 or 1 (biallelic).")
            self.alleles[a] = [0, 0, False]

    def set_coverage(self, c: int):
        self.local_coverage = int(c)

    def insert_kmer(self, readcount: int, alleles: Iterable[int]):
        index = len(self.kmer_to_count)
        self.kmer_to_count.append(int(readcount))
        for a in alleles:
            if self.biallelic and a not in (0, 1):
                raise RuntimeError("BiallelicUniqueKmers::insert_kmer: provided alleles need to be either 0 or 1 (biallelic)")
            info = self.alleles.setdefault(int(a), [0, 0, False])  # operator[] creates the allele
            if info[1] == 0:
                info[0] = index
            if index < info[0] or index >= info[0] + self.window:
                raise RuntimeError("KmerPath: index is invalid")
            info[1] |= 1 << (index - info[0])

    def set_undefined_allele(self, a: int):
        if a not in self.alleles:
            raise RuntimeError(f"set_undefined_allele: allele_id {a} does not exist.")
        self.alleles[a][2] = True

    def is_undefined_allele(self, a: int) -> bool:
        return bool(self.alleles.get(a, [0, 0, False])[2])

    def size(self) -> int:
        return len(self.kmer_to_count)

    def get_path_ids(self, only_include: Sequence[int] | None = None):
        if only_include is not None:
            p = [int(x) for x in only_include if int(x) < len(self.path_to_allele)]
        else:
            p = list(range(len(self.path_to_allele)))
        return p, [self.path_to_allele[x] for x in p]

    def get_allele(self, path_id: int) -> int:
        if path_id >= len(self.path_to_allele):
            raise RuntimeError("UniqueKmers:get_allele: index out of bounds.")
        return self.path_to_allele[path_id]

    def get_readcount_of(self, kmer_index: int) -> int:
        if kmer_index >= len(self.kmer_to_count):
            raise RuntimeError(f"get_readcount_of: requested kmer index: {kmer_index} does not exist.")
        return self.kmer_to_count[kmer_index]

    def kmer_on_allele(self, kmer_index: int, allele: int) -> bool:
        off, mask, _ = self.alleles[allele]
        return off <= kmer_index < off + self.window and bool((mask >> (kmer_index - off)) & 1)

    def kmer_on_path(self, kmer_index: int, path_index: int) -> bool:
        if path_index >= len(self.path_to_allele):
            raise RuntimeError(f"kmer_on_path: path_index {path_index} does not exist.")
        if kmer_index >= len(self.kmer_to_count):
            raise RuntimeError(f"kmer_on_path: requested kmer index: {kmer_index} does not exist.")
        return self.kmer_on_allele(kmer_index, self.path_to_allele[path_index])

    def update_paths(self, path_ids: Sequence[int]):
        """Keep only the given paths (in that order), the alleles they carry and the k-mers on those
        alleles (reference src/multiallelicuniquekmers.cpp:195-232, src/biallelicuniquekmers.cpp:223-260)."""
        new_p2a = [self.get_allele(int(p)) for p in path_ids]
        kept = {a: self.alleles[a] for a in sorted(set(new_p2a))}
        kmer_to_alleles: dict[int, list[int]] = {}
        for a in kept:
            for k in range(len(self.kmer_to_count)):
                if self.kmer_on_allele(k, a):
                    kmer_to_alleles.setdefault(k, []).append(a)
        undefined = [a for a, info in kept.items() if info[2]]
        old_counts = self.kmer_to_count
        self.path_to_allele = new_p2a
        self.alleles = {a: [0, 0, False] for a in kept}
        self.kmer_to_count = []
        for a in undefined:
            self.set_undefined_allele(a)
        for k in sorted(kmer_to_alleles):
            self.insert_kmer(old_counts[k], kmer_to_alleles[k])


def BiallelicUniqueKmers(pos, path_to_allele):
    return UniqueKmers(pos, path_to_allele, True)


def MultiallelicUniqueKmers(pos, path_to_allele):
    return UniqueKmers(pos, path_to_allele, False)


def flatten(unique_kmers: Sequence[UniqueKmers], only_paths: Sequence[int] | None = None) -> ContigBatch:
    """UniqueKmers list -> flat batch.  Path selection as ColumnIndexer does it
    (reference src/columnindexer.cpp:12-23): the selected paths are those of
    variant 0; a variant without paths is an error."""
    V = len(unique_kmers)
    paths: list[int] = []
    for v, uk in enumerate(unique_kmers):
        p, _ = uk.get_path_ids(only_paths)
        if len(p) == 0:
            raise RuntimeError(f"HMM::index_columns: column {v} is not covered by any paths.")
        if v == 0:
            paths = p
    H = len(paths)
    pos = np.zeros(V, np.uint64)
    cov = np.zeros(V, np.uint16)
    kmer_off = np.zeros(V + 1, np.uint32)
    allele_off = np.zeros(V + 1, np.uint32)
    counts: list[int] = []
    aid: list[int] = []
    aflag: list[int] = []
    aoff: list[int] = []
    amask: list[int] = []
    pa = np.zeros(V * H, np.uint16)
    for v, uk in enumerate(unique_kmers):
        pos[v] = uk.variant_pos
        cov[v] = np.uint16(int(uk.local_coverage) & 0xFFFF)
        counts.extend(uk.kmer_to_count)
        kmer_off[v + 1] = len(counts)
        for a in sorted(uk.alleles):
            off, mask, undef = uk.alleles[a]
            aid.append(a); aflag.append(1 if undef else 0); aoff.append(off); amask.append(mask)
        allele_off[v + 1] = len(aid)
        for i, p in enumerate(paths):
            pa[v * H + i] = uk.get_allele(p)
    return ContigBatch(H, pos, cov, kmer_off, np.array(counts, np.uint16), allele_off,
                       np.array(aid, np.uint16), np.array(aflag, np.uint8),
                       np.array(aoff, np.uint16), np.array(amask, np.uint32), pa)


# --------------------------------------------------------------------------- #
#  Synthetic panels
# --------------------------------------------------------------------------- #
PEAK = 27  # k-mer coverage peak used for the CPU measurements in BASELINE.md


def default_table_args(peak: int = PEAK):
    """ProbabilityTable(peak/4, 4*peak, 2*peak, 0.01) — reference src/commands.cpp:846."""
    return (peak // 4, 4 * peak, 2 * peak, 0.01)


def synthetic_panel(n_variants: int, n_paths: int, kmers_per_variant: int = 20, *,
                    seed: int = 12345, multiallelic_frac: float = 0.0,
                    undefined_frac: float = 0.01, zero_kmer_frac: float = 0.01,
                    peak: int = PEAK, max_alleles: int = 5, local_alts: int = 4,
                    wide_frac: float = 0.0, wide_alleles: tuple = (6, 12), wide_at: Sequence[int] = ()) -> ContigBatch:
    """Deterministic synthetic contig of the shapes BASELINE.json names.

    Positions step by 50+U[0,1200) bp; allele frequency f~U(.05,.95); each path carries ALT
    with probability f (multiallelic: uniform among 1..A-1, A~U{3..max_alleles}); the sample's true
    genotype follows two panel paths; K k-mers per variant split evenly over the alleles,
    each k-mer on exactly one allele (as UniqueKmerComputer produces them, reference
    src/uniquekmercomputer.cpp:59-69); read counts ~ Poisson(cn*peak/2) for cn in {1,2},
    and 0 (90%) / 1 (10%) for cn=0; local coverage = peak-3+U{0..6}.
    `wide_frac` of the objects (drawn from a generator of their own: the other draws are the same with and without
    them) get A ~ U{wide_alleles[0]..wide_alleles[1]} alleles with every path's ALT uniform among all of them — bubbles
    whose columns carry more than five distinct alleles on the selected paths (the device's "wide" columns); `wide_at`
    names variants that are such objects whatever the draw says.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    V, H, K = int(n_variants), int(n_paths), int(kmers_per_variant)
    pos = np.cumsum(50 + rng.integers(0, 1200, size=V, dtype=np.int64)).astype(np.uint64) + 10000
    cov = (peak - 3 + rng.integers(0, 7, size=V)).astype(np.uint16)

    n_all = np.full(V, 2, dtype=np.int64)
    if multiallelic_frac > 0:
        multi = rng.random(V) < multiallelic_frac
        n_all[multi] = rng.integers(3, max(6, max_alleles + 1), size=int(multi.sum()))
    if wide_frac > 0 or len(wide_at):
        rng_w = np.random.Generator(np.random.PCG64(int(seed) * 7919 + 17))
        wide = rng_w.random(V) < wide_frac
        wide[[int(i) for i in wide_at if 0 <= int(i) < V]] = True   # (tests: wide objects at chosen places — first, last, the meeting point)
        n_all[wide] = rng_w.integers(int(wide_alleles[0]), int(wide_alleles[1]) + 1, size=int(wide.sum()))
    f = rng.uniform(0.05, 0.95, size=V)
    carries_alt = rng.random((V, H)) < f[:, None]
    # (max_alleles > 5: bubbles with many alleles in the object of which the panel paths carry at
    # most `local_alts` ALTs — what a sampled panel looks like on a hypervariable site)
    alt_choice = 1 + (rng.integers(0, 1 << 30, size=(V, H)) % np.minimum(n_all[:, None] - 1, local_alts if max_alleles > 5 else 1 << 30))
    path_allele = np.where(carries_alt, alt_choice, 0).astype(np.uint16)

    # sample haplotypes: two panel paths with occasional switches
    def mosaic():
        sw = rng.random(V) < 0.002
        seg = np.cumsum(sw)
        starts = rng.integers(0, H, size=int(seg.max()) + 1)
        return starts[seg]
    h1, h2 = mosaic(), mosaic()
    rows = np.arange(V)
    g1, g2 = path_allele[rows, h1].astype(np.int64), path_allele[rows, h2].astype(np.int64)

    # k-mers: allele a owns the contiguous block [a*per, (a+1)*per), per = min(K//A, window)
    zero_k = rng.random(V) < zero_kmer_frac
    window = np.where(n_all == 2, 16, 32)
    per = np.minimum(K // n_all, window)
    per[zero_k] = 0
    Kv = per * n_all
    kmer_off = np.zeros(V + 1, np.uint32)
    np.cumsum(Kv, out=kmer_off[1:])
    sumK = int(kmer_off[-1])
    kv = np.repeat(np.arange(V), Kv)                       # variant of each k-mer
    kidx = np.arange(sumK) - np.repeat(kmer_off[:-1].astype(np.int64), Kv)
    kallele = kidx // np.maximum(per[kv], 1)
    cn = (kallele == g1[kv]).astype(np.int64) + (kallele == g2[kv]).astype(np.int64)
    lam = cn * (peak / 2.0)
    counts = rng.poisson(lam).astype(np.int64)
    noise = (rng.random(sumK) < 0.1).astype(np.int64)
    counts = np.where(cn == 0, noise, counts)
    counts = np.clip(counts, 0, 65535).astype(np.uint16)

    allele_off = np.zeros(V + 1, np.uint32)
    np.cumsum(n_all, out=allele_off[1:])
    sumA = int(allele_off[-1])
    av = np.repeat(np.arange(V), n_all)
    aidx = np.arange(sumA) - np.repeat(allele_off[:-1].astype(np.int64), n_all)
    allele_id = aidx.astype(np.uint16)
    a_per = per[av]
    allele_kmer_off = (aidx * a_per).astype(np.uint16)
    allele_kmer_mask = np.where(a_per > 0, (np.uint64(1) << a_per.astype(np.uint64)) - np.uint64(1), 0).astype(np.uint32)
    allele_kmer_off[a_per == 0] = 0
    allele_flags = np.zeros(sumA, np.uint8)
    undef_var = rng.random(V) < undefined_frac
    # the last ALT allele of the chosen variants is undefined
    last_slot = allele_off[1:].astype(np.int64) - 1
    allele_flags[last_slot[undef_var]] = 1

    return ContigBatch(H, pos, cov, kmer_off, counts, allele_off, allele_id, allele_flags,
                       allele_kmer_off, allele_kmer_mask, path_allele.reshape(-1))


def synthetic_sample_counts(index: ContigBatch, *, seed: int, peak: int = PEAK):
    """Read k-mer counts + local coverage of ANOTHER sample against the same index: a new true
    genotype (two mosaic panel paths), counts ~ Poisson(cn * peak / 2) as in synthetic_panel.
    Returns (kmer_count u16 [sumK], coverage u16 [V])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    V, H = index.n_variants, index.n_paths
    pa = index.path_allele.reshape(V, H)

    def mosaic():
        sw = rng.random(V) < 0.002
        seg = np.cumsum(sw)
        starts = rng.integers(0, H, size=int(seg.max()) + 1 if V else 1)
        return starts[seg]
    rows = np.arange(V)
    g1, g2 = pa[rows, mosaic()].astype(np.int64), pa[rows, mosaic()].astype(np.int64)
    Kv = np.diff(index.kmer_off.astype(np.int64))
    sumK = int(index.kmer_off[-1]) if V else 0
    kv = np.repeat(np.arange(V), Kv)
    kidx = np.arange(sumK) - np.repeat(index.kmer_off[:-1].astype(np.int64), Kv)
    # allele (id) that owns k-mer kidx of its variant: the one whose (offset, mask) window holds it
    A = np.diff(index.allele_off.astype(np.int64))
    av = np.repeat(np.arange(V), A)
    mask = index.allele_kmer_mask.astype(np.float64)
    alen = np.where(mask > 0, np.floor(np.log2(np.maximum(mask, 1.0))) + 1, 0).astype(np.int64)  # k-mers per allele block
    per = np.zeros(V, np.int64)
    np.maximum.at(per, av, alen)
    kallele = kidx // np.maximum(per[kv], 1)
    cn = (kallele == g1[kv]).astype(np.int64) + (kallele == g2[kv]).astype(np.int64)
    counts = rng.poisson(cn * (peak / 2.0)).astype(np.int64)
    noise = (rng.random(sumK) < 0.1).astype(np.int64)
    counts = np.clip(np.where(cn == 0, noise, counts), 0, 65535).astype(np.uint16)
    cov = (peak - 3 + rng.integers(0, 7, size=V)).astype(np.uint16)
    return counts, cov


# algorithmic HBM bytes per variant, SURVEY.md §8(d):
#   B(H,K,A) = 16 H^2 + 4 K + 2 H + 3 A + 16 + 8 G + 8   (kept columns)
#   skipped variants count only input + output bytes.
def algorithmic_bytes(batch: ContigBatch, kept: np.ndarray) -> int:
    H = batch.n_paths
    K = np.diff(batch.kmer_off.astype(np.int64))
    A = np.diff(batch.allele_off.astype(np.int64))
    G = A * (A + 1) // 2
    io = 4 * K + 2 * H + 3 * A + 16 + 8 * G + 8
    return int(io.sum() + 16 * H * H * int(np.asarray(kept, dtype=np.int64).sum()))
