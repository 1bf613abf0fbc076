#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel.  Usage: python tools/kernel_resources.py online-detection_amd/csrc/knm_pass_q.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["/usr/bin/c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in name:
        continue
    print("%-90s vgpr %4s agpr %4s sgpr %4s scratch %5s lds %7s occ %s" % (name[-90:], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"),
                                                                            r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                                                            r.get("Occupancy [waves/SIMD]")))
