#!/usr/bin/env python3
"""Rewrite the measured numbers inside the end-of-round row of profiles/README.md from the capture's files.
Usage: python profiles/tools/readme_row.py r02_k"""
import csv, json, os, re, sys
tag = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "")
J = lambda n: json.load(open(f"{P}{tag}_{n}.json"))
b, ur = J("bench"), J("bench_under_rocprof")
rows = list(csv.DictReader(open(f"{P}{tag}_kernel_stats.csv")))
k5 = [r for r in rows if "k_blur_solve" in r["Name"]]
roc = sum(float(r["TotalDurationNs"]) for r in k5) / sum(int(r["Calls"]) for r in k5) / 1e3
tj = json.load(open(P + "traffic.json"))
grid = {}
for line in open(f"{P}{tag}_kernel_trace_by_grid.txt"):
    m = re.match(r"(\S.*?)\s+\((\d+), 1, 1\)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
    if m:
        grid[(m.group(1).strip(), int(m.group(2)))] = (float(m.group(4)), float(m.group(5)))
l0 = max(g for (k, g) in grid if k.startswith("k_blur_solve<true, 1>"))
g = lambda k: grid[(k, l0)]
v = lambda n: J("bench_" + n)["value"]
text = ("Bench **%d pairs/s** (%d under rocprofv3), `k_blur_solve` %.1f µs by HIP events vs %.1f µs by rocprofv3 over the same 12 launches per step, "
        "**%.3f** of 8 TB/s, `frac_measured` %.3f (`traffic.json` %d MB per launch, signature `%s` = the committed device sources); level-0 launches "
        "%.0f / %.0f / %.0f µs (rocprofv3 averages; minima %.0f / %.0f / %.0f); 4K %d, 2880² eye %d, lanes 2: %d, B = 8: %d, 640×360: %.1f k, "
        "256² B = 256: %.1f k, B = 64: %.1f k, zoom %d, independent pairs %d.") % (
    round(b["value"]), round(ur["value"]), b["roofline"]["avg_launch_ms"] * 1e3, roc, b["roofline"]["frac"], b["roofline"]["frac_measured"],
    tj["hbm_bytes_per_launch"] / 1e6, tj["kernel_signature"],
    g("k_blur_solve<true, 1>")[0], g("k_blur_solve<true, 0>")[0], g("k_blur_solve<false, 0>")[0],
    g("k_blur_solve<true, 1>")[1], g("k_blur_solve<true, 0>")[1], g("k_blur_solve<false, 0>")[1],
    round(v("4k")), round(v("2880_eye")), round(v("lanes2")), round(v("b8")), v("640") / 1e3, v("256_b256") / 1e3, v("256_b64") / 1e3,
    round(v("zoom005")), round(v("independent")))
path = P + "README.md"
s = open(path).read()
i = s.index(f"| `{tag}_*` | **end of round 2**")
j = s.index("\n", i)
row = re.sub(r"Bench \*\*\d+ pairs/s\*\*.*?independent pairs \d+\.", text.replace("\\", "\\\\"), s[i:j], flags=re.S)
open(path, "w").write(s[:i] + row + s[j:])
print(text)
