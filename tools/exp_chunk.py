import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = ("import os, sys; sys.path.insert(0,'.'); from pangenie_amd import hmm; from pangenie_amd.panel import synthetic_panel, default_table_args;"
        "b=synthetic_panel(200000,64,20,seed=12345); job=hmm.Job([b],hmm.ProbabilityTable(*default_table_args()),hmm.make_params(1.26,False,1e-5));"
        "job.run(); job.run(); ms=job.kernel_ms(); r=job.fetch(0); C=r.n_columns;"
        "print('chunk %8s: phase1 %7.2f ms = %5.0f ns/column | phase2 %7.2f ms = %5.0f ns/column | run %7.2f ms' % (os.environ.get('PG_CHUNK_COLS','default'), ms['k_sweep_phase1'], ms['k_sweep_phase1']*1e6/(C/2), ms['k_sweep_phase2'], ms['k_sweep_phase2']*1e6/(C/2), sum(ms.values())))")
for k in ("1024", "4096", "16384", "65536", "131072"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PG_CHUNK_COLS=k))
