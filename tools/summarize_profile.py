#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats + separate FETCH_SIZE / WRITE_SIZE PMC
passes) into the small summaries kept under profiles/.

usage: summarize_profile.py <gpurun_out/rNN dir> <profiles/prefix> <workload>
Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced streaming read, so reads are doubled.
"""
import csv
import json
import sys
from pathlib import Path

src, prefix, workload = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
out = {"workload": workload, "kernels": {}, "notes": "rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
       "FETCH_SIZE doubled (gfx950 wide-read correction), KiB -> bytes"}
stats = list(csv.DictReader(open(src / "kt" / "kt_kernel_stats.csv")))
lines = ["rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (workload %s)" % workload,
         "%-58s %6s %14s %12s" % ("kernel", "calls", "avg_ns", "percent")]
for r in stats:
    lines.append("%-58s %6s %14.0f %12s" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
    out["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
for name, col in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = src / name / "pmc_counter_collection.csv"
    if not f.exists():
        continue
    agg, cnt = {}, {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
        cnt[k] = cnt.get(k, 0) + 1
    for k in agg:
        per_launch = agg[k] / cnt[k] * 1024.0 * (2.0 if col == "FETCH_SIZE" else 1.0)
        key = "hbm_read_bytes" if col == "FETCH_SIZE" else "hbm_write_bytes"
        out["kernels"].setdefault(k, {})[key + "_per_launch"] = per_launch
        out["kernels"][k][key + "_total"] = per_launch * cnt[k]
        out["kernels"][k]["pmc_launches"] = cnt[k]
# optional SQ passes (tools/profile_workload.sh with SQ=1): per-launch averages of the raw counters
sq = {}
for f in sorted(src.glob("pmc_sq*/pmc_counter_collection.csv")):
    agg, cnt = {}, {}
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Counter_Name"])
        agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
        cnt[key] = cnt.get(key, 0) + 1
    for (k, c), v in agg.items():
        sq.setdefault(k, {})[c] = v / cnt[(k, c)]
for k, d in sq.items():
    out["kernels"].setdefault(k, {})["sq_per_launch"] = d
if sq:
    lines.append("")
    lines.append("SQ counters per launch (rocprofv3 --pmc, four separate passes):")
    for k, d in sq.items():
        if any(t in k for t in ("k_sweep", "k_post", "ks_", "k_bins", "k_prep", "k_vit", "k_records")):
            lines.append("  " + k[:70])
            lines.append("    " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items())))
lines.append("")
lines.append("HBM traffic per launch from PMC (bytes; reads = 2*FETCH_SIZE*1024, writes = WRITE_SIZE*1024):")
for k, v in out["kernels"].items():
    if "hbm_read_bytes_per_launch" in v or "hbm_write_bytes_per_launch" in v:
        lines.append("%-58s read %14.0f  write %14.0f" % (k[:58], v.get("hbm_read_bytes_per_launch", 0), v.get("hbm_write_bytes_per_launch", 0)))
Path(prefix + "_summary.txt").write_text("\n".join(lines) + "\n")
Path(prefix + "_summary.json").write_text(json.dumps(out, indent=1))
print("\n".join(lines))
