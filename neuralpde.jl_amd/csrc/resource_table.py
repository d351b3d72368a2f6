#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel."""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    def f(key):
        m = re.search(key + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = dem.replace("pk::", "").replace("void ", "")
    rows.append((dem[:95], f("VGPRs"), f("AGPRs"), f("ScratchSize \[bytes/lane\]"), f("Occupancy \[waves/SIMD\]"), f("LDS Size \[bytes/block\]"), f("SGPRs")))
print(f"{'kernel':95s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s} {'SGPR':>5s}")
for r in rows:
    print(f"{r[0]:95s} {r[1]:5d} {r[2]:5d} {r[3]:7d} {r[4]:4d} {r[5]:7d} {r[6]:5d}")
