"""The oracle is test infrastructure: nothing the product ships may import, link, load or execute it.  `pangenie_amd/build.py`
BUILDS it (building the checker is not using it); only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg use it."""
import ast
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "pangenie_amd"


def imports_of(path):
    tree = ast.parse(path.read_text())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom) and node.module:
            yield node.module


def test_no_python_module_of_the_package_imports_the_oracle():
    for py in PKG.rglob("*.py"):
        bad = [m for m in imports_of(py) if m == "oracle" or m.startswith("oracle.")]
        assert not bad, (py, bad)


def test_no_product_source_refers_to_the_oracle_library():
    # native sources of the product and the ctypes loaders: no path into oracle/, no oracle library name
    files = [p for p in (PKG / "csrc").iterdir() if p.suffix in (".cpp", ".hip", ".h")]
    files += [p for p in (PKG / "host").rglob("*") if p.suffix in (".cpp", ".h", ".hpp")]
    files += [PKG / "_lib.py", PKG / "hmm.py"]
    pat = re.compile(r"libpg_oracle|oracle/_ref|oracle/_build|pyoracle")
    for f in files:
        hits = [ln for ln in f.read_text(errors="replace").splitlines() if pat.search(ln)]
        assert not hits, (f, hits[:3])


def test_bench_uses_the_oracle_in_the_cpu_baseline_legs_only():
    """Every import of the oracle in bench.py sits in the function `cpu_baseline` or under an `if not args.no_cpu_baseline`
    (the CPU-port timing of the sampler / Viterbi sub-measurements and their `matches_oracle` check): never on a timed GPU path."""
    src = (ROOT / "bench.py").read_text()
    tree = ast.parse(src)
    parents = {}
    for node in ast.walk(tree):
        for ch in ast.iter_child_nodes(node):
            parents[ch] = node
    imports = [n for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module and n.module.split(".")[0] == "oracle"]
    imports += [n for n in ast.walk(tree) if isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names)]
    assert imports, "bench.py is expected to time the CPU port of the reference"
    for imp in imports:
        node, ok = imp, False
        while node in parents and not ok:
            node = parents[node]
            if isinstance(node, ast.FunctionDef) and node.name == "cpu_baseline":
                ok = True
            if isinstance(node, ast.If) and "no_cpu_baseline" in (ast.get_source_segment(src, node.test) or ""):
                ok = True
        assert ok, "oracle imported outside a CPU-baseline leg at bench.py:%d" % imp.lineno
