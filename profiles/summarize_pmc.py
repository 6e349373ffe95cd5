#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per (kernel, grid)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for key, cs in sorted(acc.items(), key=lambda kv: -kv[0][1]):
    if filt and filt not in key[0]:
        continue
    print(key[0][:44], key[1], " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
