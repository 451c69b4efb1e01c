import os, sys
sys.path.insert(0, os.getcwd())
os.environ["PG_DEBUG"] = "32"
from pangenie_amd import hmm
from pangenie_amd.panel import synthetic_panel, default_table_args
for (V, H) in [(800, 32), (600, 64), (300, 16)]:
    b = synthetic_panel(V, H, 20, seed=1000 + V + H)
    job = hmm.Job([b], hmm.ProbabilityTable(*default_table_args()), hmm.make_params(1.26, False, 1e-5))
    job.run()
    p = job.profile_counters(0)
    print(V, H, "fallbacks phase1/phase2:", int(p[50]), int(p[51]))
