#!/usr/bin/env python3
"""Register / LDS / occupancy table of the pair kernels as hipcc compiles them for gfx950 (no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(extra=(), unit="launch.hip"):
    """unit: launch.hip instantiates every pair kernel; wake.hip / march.hip / induce.hip / flowfield.hip / order.hip hold the
    O(N) kernels of their parts of the ABI."""
    src = os.path.join(ROOT, "ludvm_amd", "csrc", unit)
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                          "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", os.devnull, *extra],
                         check=True, capture_output=True, text=True).stderr
    cur, K = None, {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            K[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            K[cur][k.strip()] = v.strip()
    return K


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else "pair_"
    for k, v in resources().items():
        if pat in k:
            name = re.sub(r"^_ZN5ludvm", "", k)
            print(name[:60].ljust(60), "VGPR", v.get("VGPRs"), "spill", v.get("VGPRs Spill"), "scratch",
                  v.get("ScratchSize [bytes/lane]"), "occ", v.get("Occupancy [waves/SIMD]"), "LDS", v.get("LDS Size [bytes/block]"))
