"""CPU checks of bench.py's bookkeeping: the sweep-phase classification of kernel names (what routes the committed
rocprofv3 PMC bytes to `roofline.traffic`) and the traffic lookup itself, on the profiles the repository carries."""
import csv
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_sweep_kernel_of_the_committed_profiles_has_a_phase():
    b = _bench()
    seen = set()
    for f in sorted((ROOT / "profiles").glob("r03_*_kernel_stats.csv")):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name") or row.get("name") or ""
                if "k_sweep" in name:
                    seen.add(name)
                    assert b._sweep_phase(name) in (1, 2, 3), name
    assert any("k_sweep_leanx<1" in n for n in seen) and any("k_sweep_lean2<" in n for n in seen)
    assert any("k_sweep_small16<1" in n for n in seen) and any("k_sweep_lean_tri<1" in n for n in seen)
    assert b._sweep_phase("k_post(DevContig const*, unsigned int)") is None
    assert b._sweep_phase("k_sweep_leanx2(DevContig const*)") == 2 and b._sweep_phase("k_sweep_leanx_tri(DevContig const*)") == 1 and b._sweep_phase("k_sweep_leanx_triw(DevContig const*)") == 1
    assert b._sweep_phase("void k_sweep<128, 32, 1, false, 3>(DevContig const*, unsigned int)") == 3


def test_profiled_traffic_matches_the_algorithmic_bytes_of_the_committed_bench_lines():
    """`traffic` of a bench line comes from the PMC summary of its workload: within 15 % of the algorithmic bytes for the
    lone-chain sweeps and the triangle cohort (nothing re-read), above them for the 128-path cohort (spills + partials)."""
    b = _bench()
    for line, lo, hi in (("r03_bench_default.json", 0.95, 1.15), ("r03_bench_chr22.json", 0.95, 1.15), ("r03_bench_h128.json", 0.95, 1.15)):
        d = json.loads((ROOT / "profiles" / line).read_text())
        r = d["roofline"]
        assert r["traffic"] is not None and r["traffic_source"].startswith("r03_"), line
        assert lo <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= hi, (line, r["traffic"], r["algorithmic_bytes_per_launch"])
    d = json.loads((ROOT / "profiles" / "r03_bench_default.json").read_text())
    for key, lo, hi in (("cohort", 0.95, 1.1), ("cohort_h16", 1.0, 1.3), ("cohort_h128", 1.2, 2.0)):
        r = d[key]["roofline"]
        assert r["traffic"] is not None, key
        assert lo <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= hi, (key, r["traffic"] / r["algorithmic_bytes_per_launch"])
    # the lookup itself: phase 1 of the 128-path chain is k_sweep_leanx's writes
    t1, src = b.profiled_traffic("chr22_h128", 1)
    assert src == "r03_chr22_h128_summary.json" and 7.5e9 < t1 < 8.5e9


def test_round5_kernel_names_and_profiles():
    """The kernels of round 5 are classified (k_sweep_small16x: phases 1-3; the bins / prep kernels: no sweep phase), and the
    committed round-5 summaries give the traffic of the new cohort lines within 20 % of their algorithmic bytes."""
    b = _bench()
    assert b._sweep_phase("void k_sweep_small16x<2>(DevContig const*, unsigned int const*, unsigned int, unsigned int, double*)") == 2
    assert b._sweep_phase("void k_sweep_small16x<1>(DevContig const*, unsigned int const*, unsigned int, unsigned int, double*)") == 1
    assert b._sweep_phase("void k_sweep_small16x<3>(DevContig const*, unsigned int const*, unsigned int, unsigned int, double*)") == 3
    for n in ("k_bins_x(DevContig const*)", "k_bins_wide(DevContig const*)", "k_prep_m4(DevContig const*, DevTable)"):
        assert b._sweep_phase(n) is None
    d = json.loads((ROOT / "profiles" / "r05_bench_default.json").read_text())
    for key in ("cohort_h16m", "cohort_h16w"):
        r = d[key]["roofline"]
        assert d[key]["sweep_mode"] == "fused" and r["kernel"] == "k_sweep_phase2"
        assert r["traffic_source"].startswith("r05_") and 0.95 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.15
        assert r["frac"] >= 0.55
    assert d["cohort_h16m"]["value"] >= 0.8 * d["cohort_h16"]["value"] and d["cohort_h16w"]["value"] >= 0.7 * d["cohort_h16"]["value"]


def test_wide_objects_of_the_generator_leave_the_other_draws_alone():
    """synthetic_panel(wide_frac=...) draws its wide objects from a generator of its own: positions, coverage and the alleles of
    every other object are those of the panel without them (the bench lines with and without wide objects differ in those only)."""
    import numpy as np
    import sys
    sys.path.insert(0, str(ROOT))
    from pangenie_amd.panel import synthetic_panel
    a = synthetic_panel(1500, 16, 20, seed=5, multiallelic_frac=0.2)
    w = synthetic_panel(1500, 16, 20, seed=5, multiallelic_frac=0.2, wide_frac=0.05)
    Aa, Aw = np.diff(a.allele_off.astype(int)), np.diff(w.allele_off.astype(int))
    wide = Aw > 5
    assert 30 < int(wide.sum()) < 130 and (Aa[~wide] == Aw[~wide]).all()
    assert (a.variant_pos == w.variant_pos).all() and (a.coverage == w.coverage).all()
    pa, pw = a.path_allele.reshape(1500, 16), w.path_allele.reshape(1500, 16)
    assert (pa[~wide] == pw[~wide]).all()
    assert max(len(set(r)) for r in pw[wide]) > 5   # (wide COLUMNS: more than five alleles on the sixteen paths)
