#!/usr/bin/env python3
"""Register and scratch use of every kernel of one HIP source (no GPU needed):
   python tools/kernel_resources.py vit-pytorch_amd/csrc/gemm_bf16.hip [name-substring] [-D...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
flags = [a for a in sys.argv[2:] if a.startswith("-")]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      *flags, "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dem:
        continue
    print(f"{dem[:110]:110s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} scratch {r.get('ScratchSize [bytes/lane]')} "
          f"vspill {r.get('VGPRs Spill')} lds {r.get('LDS Size [bytes/block]')}")
