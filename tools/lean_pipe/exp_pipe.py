"""The pipelined lean step (k_sweep_leanp) against the plain one (the default) on lone 64-path chains: agreement of
the likelihoods over chain lengths and chunk sizes that cross every boundary of the kernel (record blocks of 64, chunk
resume, a chain shorter than a block), then ns per column of both on a 50 000-variant chain.  Tooling only.
usage (GPU box): python tools/exp_pipe.py [check] [time] [variants NAME=V ...]"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from pangenie_amd import hmm  # noqa: E402
from pangenie_amd.panel import default_table_args, synthetic_panel  # noqa: E402


def run(batch, pipe, table_args=None, **env):
    (os.environ.__setitem__("PG_KERNELS", "leanpipe") if pipe else os.environ.pop("PG_KERNELS", None))
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        job = hmm.Job([batch], hmm.ProbabilityTable(*(table_args or default_table_args())), hmm.make_params(1.26, False, 1e-5))
        job.run()
        r = job.fetch(0)
        return r.likelihoods_ld(), r.n_columns
    finally:
        for k in env:
            os.environ.pop(k, None)
        os.environ.pop("PG_KERNELS", None)


def rel(a, b):
    d = np.maximum(np.abs(a), np.abs(b))
    return np.where(d > 0, np.abs(a - b) / np.where(d > 0, d, 1), 0)


def check():
    bad = 0
    for (V, chunk, seed) in ((2, 4096, 1), (3, 4096, 2), (5, 1, 3), (9, 2, 4), (70, 4096, 5), (130, 4096, 6), (131, 7, 7), (200, 64, 8), (517, 100, 9), (3000, 4096, 10), (3000, 333, 11)):
        b = synthetic_panel(V, 64, 20, seed=seed)
        a0, c0 = run(b, False, PG_SWEEP_MODE="chunked", PG_CHUNK_COLS=chunk)
        a1, c1 = run(b, True, PG_SWEEP_MODE="chunked", PG_CHUNK_COLS=chunk)
        r = rel(a0, a1)
        worst = float(r.max()) if r.size else 0.0
        first = int(np.argmax(r > 1e-9)) if (r > 1e-9).any() else -1
        ok = c0 == c1 and worst < 1e-9
        bad += 0 if ok else 1
        print("V %5d chunk %5d columns %5d/%5d: max rel diff pipelined vs plain %.3e  first entry above 1e-9: %d of %d  %s" % (V, chunk, c1, c0, worst, first, r.size, "ok" if ok else "MISMATCH"))
    # unregularised table: zero emissions, all-zero columns, uniform fall-backs in both roles
    targs = [6, 108, 54, 0.0]
    for (V, chunk, seed) in ((400, 4096, 21), (400, 50, 22), (1500, 4096, 23)):
        b = synthetic_panel(V, 64, 20, seed=seed)
        b.kmer_count[::3] = 0
        b.kmer_count[1::17] = 60000
        a0, c0 = run(b, False, tuple(targs), PG_SWEEP_MODE="chunked", PG_CHUNK_COLS=chunk)
        a1, c1 = run(b, True, tuple(targs), PG_SWEEP_MODE="chunked", PG_CHUNK_COLS=chunk)
        r = rel(a0, a1)
        worst = float(r.max()) if r.size else 0.0
        ok = c0 == c1 and worst < 1e-9
        bad += 0 if ok else 1
        print("unregularised V %5d chunk %5d: max rel diff %.3e  %s" % (V, chunk, worst, "ok" if ok else "MISMATCH"))
    print("check:", "ALL OK" if bad == 0 else "%d MISMATCHES" % bad)
    return bad


TIME_CODE = ("import os, sys; sys.path.insert(0,'.'); from pangenie_amd import hmm; from pangenie_amd.panel import synthetic_panel, default_table_args;"
             "b=synthetic_panel(50000,64,20,seed=12345); job=hmm.Job([b],hmm.ProbabilityTable(*default_table_args()),hmm.make_params(1.26,False,1e-5));"
             "job.run(); job.run(); ms=job.kernel_ms(); r=job.fetch(0); C=r.n_columns; q=job.profile_counters(0).astype(float);"
             "import hashlib; h=hashlib.sha1(r.lik.tobytes()+r.lik_exp.tobytes()).hexdigest()[:12];"
             "print('%%-34s phase1 %%7.2f ms = %%5.0f ns/column | cycles/column forward %%5.0f backward %%5.0f | phase2 %%7.2f ms = %%5.0f ns/column | total run %%7.2f ms | results %%s' %% (%r, ms['k_sweep_phase1'], ms['k_sweep_phase1']*1e6/(C/2), q[0]/(C/2), q[16]/(C/2), ms['k_sweep_phase2'], ms['k_sweep_phase2']*1e6/(C/2), sum(ms.values()), h))")


def timeit(label, lib=None, **env):
    e = dict(os.environ)
    if lib:
        e["PANGENIE_HMM_LIB"] = lib
    e.update({k: str(v) for k, v in env.items()})
    subprocess.run([sys.executable, "-c", TIME_CODE % label], env=e)


def variant_lib(v):
    return os.path.join(os.getcwd(), "tools", "_build", "libpangenie_hmm_%s.so" % v.replace("=", "").replace(",", "_"))


if __name__ == "__main__":
    args = sys.argv[1:] or ["check", "time"]
    if "build" in args:   # (here, CPU) python tools/exp_pipe.py build NAME=V ...
        from pangenie_amd import build
        for v in args[args.index("build") + 1:]:
            build.build_hip(out=variant_lib(v), defines=tuple(x for x in v.split(",") if x != "prof") + ("PG_CHAIN_PROF=1",), force=True)
            print("built", variant_lib(v))
        sys.exit(0)
    rc = 0
    if "check" in args:
        rc = check()
    if "time" in args:
        timeit("plain (product)")
        timeit("pipelined (PG_KERNELS=leanpipe)", PG_KERNELS="leanpipe")
        timeit("fused mode (triangle stores)", PG_SWEEP_MODE="fused")
        timeit("fused, full columns (PG_KERNELS=notri)", PG_SWEEP_MODE="fused", PG_KERNELS="notri")
    if "variants" in args:
        for v in args[args.index("variants") + 1:]:   # NAME or NAME@pipe (the pipelined lean step of that build)
            name, _, how = v.partition("@")
            if how == "pipe":
                timeit(name + " pipelined", lib=variant_lib(name), PG_KERNELS="leanpipe")
            else:
                timeit(name, lib=variant_lib(name))
    sys.exit(1 if rc else 0)
