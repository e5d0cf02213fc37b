#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy of our own kernels in one csrc/*.hip (hipcc remarks)."""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\s+(\S.*?) \[-Rpass-analysis", line)
        if not m:
            continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    if "rocprim" in name or (pat and pat not in name):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    print(f"{dem[:70]:70s} VGPR={d.get('VGPRs','?'):>4} AGPR={d.get('AGPRs','?'):>3} SGPR={d.get('TotalSGPRs', d.get('SGPRs','?')):>4} "
          f"scratch={d.get('ScratchSize [bytes/lane]','?'):>3} occ={d.get('Occupancy [waves/SIMD]','?'):>2} LDS={d.get('LDS Size [bytes/block]','?')}")
