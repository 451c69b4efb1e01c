"""Where the cycles of a lean forward step go: builds the product library with -DPG_LEANPROF (s_memtime stamps at
the seams of the step, see pg_kernels.hip) and prints cycles per column per segment.  Tooling only.
usage: python tools/prof_lean.py build (CPU)  |  python tools/prof_lean.py run (GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd())
lib = os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_leanprof.so")
if sys.argv[1] == "build":
    from pangenie_amd import build
    build.build_hip(out=lib, defines=("PG_LEANPROF=1",), force=True)
    print("built", lib)
else:
    os.environ["PANGENIE_HMM_LIB"] = lib
    from pangenie_amd import hmm
    from pangenie_amd.panel import synthetic_panel, default_table_args
    b = synthetic_panel(50000, 64, 20, seed=12345)
    job = hmm.Job([b], hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5))
    job.run(); job.run()
    ms = job.kernel_ms(); C = job.fetch(0).n_columns
    p = job.profile_counters(0).astype(float)
    pipe = os.environ.get("PG_LEAN_PIPE", "1") != "0"
    n = max(p[40] if pipe else p[39], 1)
    if pipe:
        names = ["top: records, park, issue G reads", "states 0-3", "Gj, Cn, first MFMAs issued", "states 4-7", "adds, second MFMAs issued",
                 "states 8-11", "constants, u round trip", "states 12-15, tail, barrier"]
    else:
      names = ["record reads + column sums", "u round trip", "MFMA total", "scale / constants / emission pair", "16 states + stores",
             "park sums, collect scalars", "barrier"]
    print("phase1 %.2f ms, %d columns; last forward launch: %d steps (s_memtime ticks; 100 MHz constant clock if the counter is REFCLK)" % (ms["k_sweep_phase1"], C, n))
    tot = 0
    for i, nm in enumerate(names):
        print("  %-36s %8.1f ticks/column" % (nm, p[32 + i] / n)); tot += p[32 + i] / n
    print("  %-36s %8.1f" % ("sum", tot))
