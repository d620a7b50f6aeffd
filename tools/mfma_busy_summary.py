#!/usr/bin/env python3
"""MFMA-busy fraction per kernel from tools/pmc_fold.py's output (tools/gpu.sh pmc_mfma -> pmc_mfma_per_kernel.txt):
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs divided by GRBM_GUI_ACTIVE / 8 XCDs.   python tools/mfma_busy_summary.py IN [TITLE] > OUT"""
import ast
import sys

rows, key, cur = [], None, {}
for line in open(sys.argv[1]):
    if line.startswith("("):
        if key is not None:
            rows.append((key, cur))
        key, cur = ast.literal_eval(line.strip()), {}
    elif line.strip() and key is not None:
        parts = line.split()
        cur[parts[0]] = float(parts[1])
if key is not None:
    rows.append((key, cur))
print("MFMA-busy fraction per kernel of one eager forward (bench.py --graph 0)%s, folded from %s:" % (
    ", " + sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[1].split("/")[-1]))
print("SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs divided by GRBM_GUI_ACTIVE / 8 XCDs (GRBM_GUI_ACTIVE is not the shader clock: the fractions "
      "understate the pipe's share of shader cycles by 10-20 %, DESIGN 4.2 round 4)\n")
for (name, grid), c in rows:
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c["GRBM_GUI_ACTIVE"] <= 0:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print("%-62s grid %-8s cycles/XCD %9.0f  mfma busy %.3f" % (name, grid, cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc))
