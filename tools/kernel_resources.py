#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of libludvm_hip.so as hipcc reports it (no GPU needed).
    python tools/kernel_resources.py [filter] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
flt = args[0] if args and args[0] != "--" else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
       "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(ROOT, "ludvm_amd/csrc/ludvm_hip.hip"),
       "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print(f"{'kernel':70s} VGPR AGPR scratch occ  LDS")
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.replace("ludvm::", "").split("(")[0]
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} {r.get('VGPRs', '?'):>4s} {r.get('AGPRs', '?'):>4s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>3s} {r.get('LDS Size [bytes/block]', '?'):>6s}")
