"""Per-kernel register / scratch / LDS usage of the product kernels (hipcc remarks; runs without a GPU).
usage: python tools/kernel_resources.py [--file pg_sampler.hip] [extra hipcc flags]"""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
SRC = "pg_kernels.hip"
if "--file" in sys.argv:
    i = sys.argv.index("--file")
    SRC = sys.argv[i + 1]
    del sys.argv[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", str(ROOT / "pangenie_amd/csrc" / SRC),
       "-o", "/tmp/pg_kernels_res.o", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    g = lambda k: r.get(k, 0)
    print(f"{name[:58]:58s} VGPR {g('VGPRs'):4d} AGPR {g('AGPRs'):3d} SGPR {g('TotalSGPRs'):4d} scratch {g('ScratchSize'):5d} "
          f"spillV {g('VGPRs Spill'):3d} occ {g('Occupancy')} LDS {g('LDS Size')}")
