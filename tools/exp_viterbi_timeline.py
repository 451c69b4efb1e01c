"""k_viterbi's column step, stamped (a -DPG_VIT_TIMELINE build: pg_experiments.h VitTimeline) and ablated (-DPG_VIT_EXP=mask builds:
results wrong, timing only).  Builds the variants on the GPU box, one after the other, and prints ns per column and — for the
stamped build — cycles per segment of wave 0.  profiles/r06_viterbi.txt."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R))

SEG = ["top -> row maxima (hi) [K per lane + 4 DPP steps]", "rows: lo pass", "rows: index pass", "LDS writes", "barrier",
       "LDS reads back", "column as a whole: hi pass", "... lo pass, ballot, index", "scale, t2 / t1 products", "states (t0 product, 3 x better, emission, store)",
       "wide check, e := en"]

CHILD = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
from pangenie_amd import hmm
from pangenie_amd.panel import default_table_args, synthetic_panel
t = hmm.ProbabilityTable(*default_table_args())
prm = hmm.make_params(1.26, False, 1e-5, run_genotyping=False, run_phasing=True)
for H in (30, 16, 64):
    b = synthetic_panel(100_000, H, 20, seed=900)
    job = hmm.Job([b], t, prm)
    job.run()
    ms = []
    for _ in range(3):
        job.run(); ms.append(job.viterbi_ms())
    C = int(job.fetch(0).n_columns)
    pc = job.profile_counters(0)
    print(json.dumps({"H": H, "ns_per_column": min(ms) * 1e6 / C, "steps": int(pc[15]), "seg": [int(x) for x in pc[:11]]}))
    job.close()
""" % str(R)


def main():
    from pangenie_amd import build as b
    lib = R / "pangenie_amd/csrc/libpangenie_hmm.so"
    keep = Path("/tmp/lib_keep.so")
    shutil.copy(lib, keep)
    variants = [("product", None), ("no fast paths", ["PG_VIT_FASTG=0", "PG_VIT_FASTR=0"]), ("rows fast only", ["PG_VIT_FASTG=0"]),
                ("column fast only", ["PG_VIT_FASTR=0"]), ("4 waves", ["PG_VIT_NW=4"]), ("timeline", ["PG_VIT_TIMELINE"]),
                ("timeline, 4 waves", ["PG_VIT_TIMELINE", "PG_VIT_NW=4"])] + [("exp %d" % m, ["PG_VIT_EXP=%d" % m]) for m in (8, 16)]
    if len(sys.argv) > 1:
        variants = [v for v in variants if v[0] in sys.argv[1:]]
    try:
        for name, defs in variants:
            if defs is not None:
                b.build_hip(force=True, out=Path("/tmp/libv/libpangenie_hmm.so"), defines=defs)
                shutil.copy("/tmp/libv/libpangenie_hmm.so", lib)
            else:
                shutil.copy(keep, lib)
            out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600)
            import json
            for line in out.stdout.splitlines():
                if not line.startswith("{"):
                    continue
                d = json.loads(line)
                print("%-18s H = %2d: %7.1f ns per column" % (name, d["H"], d["ns_per_column"]), flush=True)
                if d["steps"]:
                    tot = sum(d["seg"])
                    for s, v in zip(SEG, d["seg"]):
                        print("      %-70s %7.1f cycles" % (s, v / d["steps"]))
                    print("      %-70s %7.1f" % ("sum", tot / d["steps"]))
            if out.returncode:
                print(name, "FAILED", out.stderr[-400:])
    finally:
        shutil.copy(keep, lib)


if __name__ == "__main__":
    main()
