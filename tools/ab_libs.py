import os, subprocess, sys
code = ("import sys; sys.path.insert(0,'.'); from pangenie_amd import hmm; from pangenie_amd.panel import synthetic_panel, default_table_args;"
        "b=synthetic_panel(50000,64,20,seed=12345); job=hmm.Job([b],hmm.ProbabilityTable(*default_table_args()),hmm.make_params(1.26,False,1e-5));"
        "job.run(); job.run(); ms=job.kernel_ms(); print(ms['k_sweep_phase1'], ms['k_sweep_phase2'])")
for lib in sys.argv[1:]:
    for dbg in ("0", "8"):
        env = dict(os.environ, PG_DEBUG=dbg)
        if lib != "product": env["PANGENIE_HMM_LIB"] = os.path.join(os.getcwd(), lib)
        print(lib, "PG_DEBUG", dbg, subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())
