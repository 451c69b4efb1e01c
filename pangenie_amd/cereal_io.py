"""Reader / writer of the reference's cereal *binary* archives of UniqueKmers tables
(`<prefix>_UniqueKmersMap.cereal`, written by PanGenie-index; reference src/commands.hpp:11-28,
src/commands.cpp:653-705) — without cereal.

Layout (little endian, no framing; SURVEY.md appendix D, member order from the reference's serialize
functions src/commands.hpp:19-22, src/biallelicuniquekmers.hpp:101-104, :32-41,
src/multiallelicuniquekmers.hpp:100-103, src/kmerpath.hpp:25-28, src/kmerpath16.hpp:25-28):

  kmersize u64 · map<string, vector<shared_ptr<UniqueKmers>>> (u64 n; per entry: string = u64 len + bytes;
  vector = u64 n; per element: polymorphic type id u32 — MSB set the first time a type occurs, then
  followed by its name as a string — · shared-pointer id u32 (MSB set = new object, data follows) ·
  object: variant_pos u64, local_coverage f32, current_index u64, kmer_to_count (u64 n + n x u16),
  alleles map (u64 n; key bool u8 | u16; value {offset u16, kmers u16 | u32}, is_undefined u8),
  path_to_allele (u64 n + n x (bool u8 | u16))) · runtimes, sampling_runtimes map<string, f64> ·
  add_reference u8.

Python side = plumbing for tests and tools; the C++ host has the same reader
(pangenie_amd/host/cereal_io.hpp).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List

from .panel import UniqueKmers

_MSB = 0x80000000
_NAMES = {True: "BiallelicUniqueKmers", False: "MultiallelicUniqueKmers"}


@dataclass
class UniqueKmersMap:
    kmersize: int = 31
    unique_kmers: Dict[str, List[UniqueKmers]] = field(default_factory=dict)
    runtimes: Dict[str, float] = field(default_factory=dict)
    sampling_runtimes: Dict[str, float] = field(default_factory=dict)
    add_reference: bool = False


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.o = data, 0

    def take(self, fmt: str):
        v = struct.unpack_from("<" + fmt, self.d, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self) -> str:
        n = self.take("Q")
        s = self.d[self.o:self.o + n].decode()
        self.o += n
        return s


def _read_object(r: _Reader, biallelic: bool) -> UniqueKmers:
    pos = r.take("Q")
    cov = r.take("f")
    r.take("Q")  # current_index (= number of k-mers inserted)
    n = r.take("Q")
    counts = list(r.take(f"{n}H")) if n > 1 else ([r.take("H")] if n == 1 else [])
    na = r.take("Q")
    alleles = {}
    for _ in range(na):
        key = r.take("B") if biallelic else r.take("H")
        off = r.take("H")
        mask = r.take("H") if biallelic else r.take("I")
        undef = r.take("B")
        alleles[int(key)] = [int(off), int(mask), bool(undef)]
    npth = r.take("Q")
    fmt = "B" if biallelic else "H"
    pta = list(r.take(f"{npth}{fmt}")) if npth > 1 else ([r.take(fmt)] if npth == 1 else [])
    u = UniqueKmers(pos, pta, biallelic)
    u.kmer_to_count = [int(c) for c in counts]
    u.alleles = alleles
    u.local_coverage = cov
    return u


def _read_str_double_map(r: _Reader) -> Dict[str, float]:
    out = {}
    for _ in range(r.take("Q")):
        k = r.string()
        out[k] = r.take("d")
    return out


def loads(data: bytes) -> UniqueKmersMap:
    r = _Reader(data)
    m = UniqueKmersMap(kmersize=r.take("Q"))
    type_of_id: Dict[int, bool] = {}
    objects: Dict[int, UniqueKmers] = {}
    for _ in range(r.take("Q")):
        name = r.string()
        lst = []
        for _ in range(r.take("Q")):
            tid = r.take("I")
            if tid & _MSB:
                tname = r.string()
                if tname not in _NAMES.values():
                    raise ValueError(f"unknown polymorphic type {tname!r}")
                type_of_id[tid & ~_MSB] = tname == _NAMES[True]
            elif tid == 0:
                lst.append(None)  # null pointer
                continue
            pid = r.take("I")
            if pid & _MSB:
                obj = _read_object(r, type_of_id[tid & ~_MSB])
                objects[pid & ~_MSB] = obj
            else:
                obj = objects[pid]
            lst.append(obj)
        m.unique_kmers[name] = lst
    m.runtimes = _read_str_double_map(r)
    m.sampling_runtimes = _read_str_double_map(r)
    m.add_reference = bool(r.take("B"))
    if r.o != len(data):
        raise ValueError(f"{len(data) - r.o} trailing bytes")
    return m


def load(path) -> UniqueKmersMap:
    with open(path, "rb") as f:
        return loads(f.read())


def dumps(m: UniqueKmersMap) -> bytes:
    out = bytearray()
    w = lambda fmt, *v: out.extend(struct.pack("<" + fmt, *v))

    def string(s: str):
        b = s.encode()
        w("Q", len(b))
        out.extend(b)
    w("Q", m.kmersize)
    w("Q", len(m.unique_kmers))
    type_ids: Dict[bool, int] = {}
    next_ptr = 1
    for name in sorted(m.unique_kmers):  # std::map order
        string(name)
        lst = m.unique_kmers[name]
        w("Q", len(lst))
        for u in lst:
            if u.biallelic in type_ids:
                w("I", type_ids[u.biallelic])
            else:
                type_ids[u.biallelic] = len(type_ids) + 1
                w("I", type_ids[u.biallelic] | _MSB)
                string(_NAMES[u.biallelic])
            w("I", next_ptr | _MSB)
            next_ptr += 1
            w("Q", u.variant_pos)
            w("f", float(u.local_coverage))
            w("Q", len(u.kmer_to_count))
            w("Q", len(u.kmer_to_count))
            for c in u.kmer_to_count:
                w("H", c)
            w("Q", len(u.alleles))
            for a in sorted(u.alleles):
                off, mask, undef = u.alleles[a]
                if u.biallelic:
                    w("BHHB", a, off, mask, 1 if undef else 0)
                else:
                    w("HHIB", a, off, mask, 1 if undef else 0)
            w("Q", len(u.path_to_allele))
            for a in u.path_to_allele:
                w("B" if u.biallelic else "H", a)
    for mp in (m.runtimes, m.sampling_runtimes):
        w("Q", len(mp))
        for k in sorted(mp):
            string(k)
            w("d", mp[k])
    w("B", 1 if m.add_reference else 0)
    return bytes(out)


def dump(m: UniqueKmersMap, path) -> None:
    with open(path, "wb") as f:
        f.write(dumps(m))


# --------------------------------------------------------------------------- #
#  Results: what PanGenie serializes with `-w` to <out>_genotyping.cereal (reference
#  src/commands.cpp:59-72, :511-516, :1012-1017) and PanGenie-vcf reads back (:1099-1104).
#  map<string, vector<GenotypingResult>> + runtimes map<string, f64>; a GenotypingResult is
#  genotype_to_likelihood (u64 n; per entry u16 a1, u16 a2, long double = its 16 bytes in memory: the
#  80-bit value + 6 padding bytes) · haplotype_1 · haplotype_2 · local_coverage · unique_kmers (u16 each)
#  (src/genotypingresult.hpp:13-29, :77-80).  Same layout as pangenie_amd/host/cereal_io.hpp.
# --------------------------------------------------------------------------- #
@dataclass
class Results:
    result: Dict[str, list] = field(default_factory=dict)   # chromosome -> [GenotypingResult]
    runtimes: Dict[str, float] = field(default_factory=dict)


def loads_results(data: bytes) -> Results:
    import numpy as np
    from .genotyping_result import GenotypingResult
    r = _Reader(data)
    out = Results()
    for _ in range(r.take("Q")):
        name = r.string()
        lst = []
        for _ in range(r.take("Q")):
            g = GenotypingResult()
            for _ in range(r.take("Q")):
                a1, a2 = r.take("HH")
                lik = np.frombuffer(r.d[r.o:r.o + 16], dtype=np.longdouble)[0]
                r.o += 16
                g.genotype_to_likelihood[(a1, a2)] = lik
            g.haplotype_1, g.haplotype_2, g.local_coverage, g.unique_kmers = r.take("HHHH")
            lst.append(g)
        out.result[name] = lst
    out.runtimes = _read_str_double_map(r)
    if r.o != len(data):
        raise ValueError("Results archive: trailing bytes")
    return out


def dumps_results(res: Results) -> bytes:
    import numpy as np
    out = bytearray()
    w = lambda fmt, *v: out.extend(struct.pack("<" + fmt, *v))

    def string(s: str):
        b = s.encode()
        w("Q", len(b))
        out.extend(b)
    w("Q", len(res.result))
    for name in sorted(res.result):  # std::map order
        string(name)
        w("Q", len(res.result[name]))
        for g in res.result[name]:
            w("Q", len(g.genotype_to_likelihood))
            for (a1, a2) in sorted(g.genotype_to_likelihood):
                w("HH", a1, a2)
                raw = np.array([g.genotype_to_likelihood[(a1, a2)]], dtype=np.longdouble).tobytes()
                out.extend(raw[:10] + b"\x00" * 6)
            w("HHHH", g.haplotype_1, g.haplotype_2, g.local_coverage, g.unique_kmers)
    w("Q", len(res.runtimes))
    for k in sorted(res.runtimes):
        string(k)
        w("d", res.runtimes[k])
    return bytes(out)
