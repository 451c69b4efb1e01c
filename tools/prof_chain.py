import os, sys
sys.path.insert(0, os.getcwd())
os.environ["PG_DEBUG"]=sys.argv[1] if len(sys.argv)>1 else "8"
H=int(sys.argv[2]) if len(sys.argv)>2 else 64
from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel, default_table_args
V=50000
b=synthetic_panel(V,H,20,seed=12345)
job=hmm.Job([b], hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26,False,1e-5))
job.run(); job.run()
ms=job.kernel_ms(); r=job.fetch(0); C=r.n_columns
p=job.profile_counters(0).astype(float)
print("H",H,"dbg",os.environ["PG_DEBUG"],"fwd ms",ms["k_forward"],"bwd ms",ms["k_backward"],"cols",C, "us/col fwd", ms["k_forward"]*1e3/C)
print(" loader cycles/col: stage-wait %.0f  barrier %.0f"%(p[8]/(C/2), p[9]/(C/2)))
for w in range(4):
    o=p[16+4*w:20+4*w]/C
    print(" wave",w,"cycles/col: pre %.0f main %.0f reduce %.0f barrier %.0f  total %.0f"%(o[0],o[1],o[2],o[3],o.sum()))
print(" wave0 pre split cycles/col: setup(rec reads) %.0f  finalize(sums) %.0f  scale+u %.0f  -> rest = row_values" % tuple(p[40:43]/C))
