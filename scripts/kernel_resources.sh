#!/bin/bash
# VGPRs / scratch bytes / LDS of every kernel in unet.hip for a set of -D flags (compile only, no GPU):
#   scripts/kernel_resources.sh [-DTRACE_BF3_WGS=3 ...]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I trace_amd/csrc -fno-slp-vectorize "$@" \
  --cuda-device-only -c trace_amd/csrc/unet.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import sys, re
name = None; row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); row = {}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m: row[key] = int(m.group(1))
    if name and "lds" in row:
        import subprocess
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = short.replace("trace::", "").replace("(trace::ConvArgs)", "").replace("void ", "")
        print("%-64s vgpr %3d agpr %3d scratch %4d occ %d" % (short[:64], row.get("vgpr", -1), row.get("agpr", 0), row.get("scratch", -1), row.get("occ", -1)))
        name = None
'
