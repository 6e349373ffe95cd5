#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, avg/min/max us."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    rows[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
print(f"{'kernel':40s} {'grid(blocks)':>22s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for (name, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:40]:40s} {str(grid):>22s} {len(v):6d} {sum(v)/len(v):9.1f} {min(v):9.1f} {max(v):9.1f}")
