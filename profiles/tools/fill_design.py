#!/usr/bin/env python3
"""Regenerate the numbers table of DESIGN.md section 7 (between the numbers:begin / numbers:end markers) from
profiles/tools/design_section7.tmpl and the committed profiles of one capture tag, so that every quoted number is
read from a file and not transcribed by hand.
Usage: python profiles/tools/fill_design.py r02_d [DESIGN.md]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "DESIGN.md")
P = os.path.join(ROOT, "profiles")


def J(name):
    return json.load(open(os.path.join(P, f"{tag}_{name}.json")))


b = J("bench")
r = b["roofline"]
kc = b["kernel_classes"]
stats = {row["Name"].split("(")[0].replace("void ", ""): row for row in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv")))}
k5 = [v for k, v in stats.items() if k.startswith("k_blur_solve")]
k5_calls = sum(int(v["Calls"]) for v in k5)
k5_ns = sum(float(v["TotalDurationNs"]) for v in k5)
tot_ns = sum(float(v["TotalDurationNs"]) for k, v in stats.items() if k.startswith("k_"))
grid = {}
for line in open(os.path.join(P, f"{tag}_kernel_trace_by_grid.txt")):
    m = re.match(r"(\S.*?)\s+\((\d+), 1, 1\)\s+(\d+)\s+([\d.]+)", line)
    if m:
        grid[(m.group(1).strip(), int(m.group(2)))] = float(m.group(4))
l0 = max(g for (k, g) in grid if k.startswith("k_blur_solve<true, 1>"))
tj = json.load(open(os.path.join(P, "traffic.json")))
pmc = {}
for line in open(os.path.join(P, f"{tag}_pmc_fetch_write.txt")):
    m = re.match(r"(\S.*?)\s+(\d+)\s+(FETCH_SIZE|WRITE_SIZE)=([\d.e+]+)", line)
    if m:
        pmc[(m.group(1).strip(), int(m.group(2)), m.group(3))] = float(m.group(4))


def real_gb(kernel):   # per launch: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB
    return (2 * pmc[(kernel, l0, "FETCH_SIZE")] + pmc[(kernel, l0, "WRITE_SIZE")]) * 1024 / 1e9


s256 = b["small_image"]
cpu = b["cpu_baseline"]
vals = {
    "VALUE": f"{b['value']:.0f}", "MS": f"{b['ms_per_step']:.2f}",
    "K5SHARE": f"{100 * k5_ns / tot_ns:.0f}", "K5US": f"{r['avg_launch_ms'] * 1e3:.1f}", "K5GB": f"{r['achieved']:.0f}",
    "K5FRAC": f"{r['frac']:.3f}", "K5ROC": f"{k5_ns / k5_calls / 1e3:.1f}",
    "TRAFFIC": f"{tj['hbm_bytes_per_launch'] / 1e6:.0f}", "TRATIO": f"{tj['hbm_bytes_per_launch'] / r['alg_bytes_per_launch']:.2f}",
    "FRACM": f"{(r.get('frac_measured') or tj['hbm_bytes_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e9 / 8000.0):.3f}",
    "L0A": f"{grid[('k_blur_solve<true, 1>', l0)]:.0f}", "L0B": f"{grid[('k_blur_solve<true, 0>', l0)]:.0f}",
    "L0C": f"{grid[('k_blur_solve<false, 0>', l0)]:.0f}",
    "L0A_GB": f"{real_gb('k_blur_solve<true, 1>'):.1f}", "L0B_GB": f"{real_gb('k_blur_solve<true, 0>'):.1f}",
    "L0C_GB": f"{real_gb('k_blur_solve<false, 0>'):.1f}",
    "L0A_TB": f"{real_gb('k_blur_solve<true, 1>') / grid[('k_blur_solve<true, 1>', l0)] * 1e3:.1f}",
    "L0B_TB": f"{real_gb('k_blur_solve<true, 0>') / grid[('k_blur_solve<true, 0>', l0)] * 1e3:.1f}",
    "L0C_TB": f"{real_gb('k_blur_solve<false, 0>') / grid[('k_blur_solve<false, 0>', l0)] * 1e3:.1f}",
    "KC_BS": f"{kc['k_blur_solve']['ms_per_step']:.2f}", "KC_PE": f"{kc['k_polyexp']['ms_per_step']:.2f}",
    "KC_PY": f"{kc['k_pyr_level']['ms_per_step']:.2f}", "KC_P1": f"{kc['k_pass1']['ms_per_step']:.2f}",
    "KC_UM": f"{kc['k_update_matrices']['ms_per_step']:.2f}", "KC_RAD": f"{kc['k_radial']['ms_per_step']:.2f}",
    "WP": f"{b['whole_path']['achieved_GBps'] / 1e3:.2f}", "WPF": f"{b['whole_path']['achieved_GBps'] / 8000:.3f}",
    "LANES2": f"{J('bench_lanes2')['value']:.0f}", "B8": f"{J('bench_b8')['value']:.0f}",
    "ZOOM": f"{J('bench_zoom005')['value']:.0f}", "INDEP": f"{J('bench_independent')['value']:.0f}",
    "K4": f"{J('bench_4k')['value']:.0f}", "K4F": f"{J('bench_4k')['roofline']['frac']:.3f}",
    "EYE": f"{J('bench_2880_eye')['value']:.0f}", "S640": f"{J('bench_640')['value'] / 1e3:.1f} k",
    "S256": f"{s256['value'] / 1e3:.0f} k", "S256F": f"{s256['whole_path_frac']:.2f}",
    "S256S": f"{s256['value'] * 32.9e6 / 8e12:.2f}", "S256B64": f"{J('bench_256_b64')['value'] / 1e3:.0f} k",
    "PCIE_G": f"{b['pcie_inclusive']['gray']['value']:.0f}", "PCIE_GG": f"{b['pcie_inclusive']['gray']['h2d_GBps']:.1f}",
    "PCIE_B": f"{b['pcie_inclusive']['bgr']['value']:.0f}", "PCIE_BG": f"{b['pcie_inclusive']['bgr']['h2d_GBps']:.1f}",
    "CORES": str(cpu["threads"]), "CPU": f"{cpu['value']:.1f}", "RATIO": f"{b['value'] / cpu['value']:.0f}",
}
tmpl = open(os.path.join(ROOT, "profiles", "tools", "design_section7.tmpl")).read()
missing = set(re.findall(r"@([A-Z0-9_]+)@", tmpl)) - set(vals)
assert not missing, missing
for k, v in vals.items():
    tmpl = tmpl.replace(f"@{k}@", v)
tmpl = tmpl.replace("r02_d", tag)
text = open(path).read()
a = text.index("<!-- numbers:begin")
a = text.index("\n", a) + 1
e = text.index("<!-- numbers:end -->")
open(path, "w").write(text[:a] + tmpl + text[e:])
print("filled", len(vals), "values from", tag)
