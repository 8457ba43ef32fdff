#!/usr/bin/env python3
"""Compact per-kernel resource table of a .hip translation unit (hipcc -Rpass-analysis=kernel-resource-usage, CPU only).
usage: kernel_resources.py real3dportrait_amd/csrc/r3d_render.hip [name-filter] [extra hipcc flags...]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wno-unused-function", "-Iinclude",
       "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for l in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", l)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    if flt in r["name"]:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void r3d::", "")
        print("%-62s vgpr %3s agpr %3s sgpr %3s | spill sgpr %3s vgpr %3s scratch %4s | occ %s lds %s" % (
            name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
            r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
