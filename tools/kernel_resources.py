#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in ba_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage, device only).
usage: tools/kernel_resources.py [filter-substring ...]   (extra -D flags through VISFS_BA_EXTRA_FLAGS; KRES_TXT=<file> reuses a saved remark dump)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "visfs_amd", "csrc", "ba_kernels.hip")
if os.environ.get("KRES_TXT"):
    txt = open(os.environ["KRES_TXT"]).read()
else:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/_kres.o"]
    cmd += os.environ.get("VISFS_BA_EXTRA_FLAGS", "").split()
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark:\s+([A-Za-z][\w /\[\]]*?): (\S+) \[-Rpass", line)
    if not m: continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        cur = {"name": val}; rows.append(cur)
    elif cur is not None and val.lstrip("-").isdigit():
        cur[key] = int(val)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
flt = sys.argv[1:]
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'spillV':>6} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
for r, n in zip(rows, names):
    name = re.sub(r"\(.*\)$", "", n.replace("visfs_ba::", "").replace("void ", ""))
    if flt and not all(f in name for f in flt): continue
    print(f"{r.get('VGPRs', -1):>5} {r.get('AGPRs', -1):>5} {r.get('TotalSGPRs', -1):>5} {r.get('VGPRs Spill', -1):>6} {r.get('ScratchSize [bytes/lane]', -1):>7} {r.get('Occupancy [waves/SIMD]', -1):>3} {r.get('LDS Size [bytes/block]', -1):>6}  {name}")
