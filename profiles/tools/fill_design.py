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
l0 = max(((v, g) for (k, g), v in grid.items() if k.startswith("k_blur_solve<true, 1>")))[1]   # level 0 = the longest folded launch
tj = json.load(open(os.path.join(P, "traffic.json")))["workloads"]["1920x1080_b32"]
pmc = {}
for line in open(os.path.join(P, f"{tag}_pmc_fetch_write.txt")):
    m = re.match(r"(\S.*?)\s+(\d+)\s+(FETCH_SIZE|WRITE_SIZE)=([\d.e+]+)", line)
    if m:
        pmc[(m.group(1).strip(), int(m.group(2)), m.group(3))] = float(m.group(4))


def real_gb(kernel):   # per launch: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB
    return (2 * pmc[(kernel, l0, "FETCH_SIZE")] + pmc[(kernel, l0, "WRITE_SIZE")]) * 1024 / 1e9


s256 = b["small_image"]
cpu = b["cpu_baseline"]
li = b["large_image"]
k4, eye = li["3840x2160"], li["2880x2880_eye"]
pc = b["pcie_inclusive"]
vals = {
    "VALUE": f"{b['value']:.0f}", "MS": f"{b['ms_per_step']:.2f}",
    "K5SHARE": f"{100 * k5_ns / tot_ns:.0f}", "K5US": f"{r['avg_launch_ms'] * 1e3:.1f}", "K5GB": f"{r['achieved']:.0f}",
    "K5FRAC": f"{r['frac']:.3f}", "K5ROC": f"{k5_ns / k5_calls / 1e3:.1f}",
    "TRAFFIC": f"{tj['hbm_bytes_per_launch'] / 1e6:.0f}", "TRATIO": f"{tj['hbm_bytes_per_launch'] / r['alg_bytes_per_launch']:.2f}",
    "FRACM": f"{r['frac_measured']:.3f}",
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
    "KC_GRAY": f"{kc['k_gray']['ms_per_step']:.2f}", "KC_GRAYF": f"{kc['k_gray']['frac']:.2f}",
    "WP": f"{b['whole_path']['achieved_GBps'] / 1e3:.2f}", "WPF": f"{b['whole_path']['achieved_GBps'] / 8000:.3f}",
    "K4": f"{k4['value']:.0f}", "K4F": f"{k4['roofline']['frac']:.3f}", "K4FM": f"{k4['roofline']['frac_measured']:.3f}",
    "K4T": f"{k4['roofline']['traffic'] / k4['roofline']['alg_bytes_per_launch']:.2f}",
    "EYE": f"{eye['value']:.0f}", "EYEF": f"{eye['roofline']['frac']:.3f}", "EYEFM": f"{eye['roofline']['frac_measured']:.3f}",
    "S256": f"{s256['value'] / 1e3:.0f} k", "S256F": f"{s256['whole_path_frac']:.2f}",
    "S256S": f"{s256['value'] * 32.9e6 / 8e12:.2f}", "S256K5": f"{s256['roofline']['frac']:.2f}",
    "S256K5M": f"{s256['roofline']['frac_measured']:.2f}",
    "PCIE_G": f"{pc['gray']['value']:.0f}", "PCIE_GR": f"{pc['gray']['vs_resident']:.2f}",
    "PCIE_B": f"{pc['bgr']['value']:.0f}", "PCIE_BG": f"{pc['bgr']['h2d_GBps']:.1f}", "PCIE_BR": f"{pc['bgr']['vs_resident']:.2f}",
    "PCIE_P": f"{pc['bgr_pinned']['value']:.0f}", "PCIE_PR": f"{pc['bgr_pinned']['vs_resident']:.2f}",
    "CPU": f"{cpu['value']:.1f}", "CPUW": str(cpu["cores"]), "CPU1": f"{cpu['single_thread']:.2f}",
    "CPUPW": f"{cpu['per_worker_vs_single_at_value']:.2f}", "CPUQ": f"{cpu['cpu_quota_cores']:.0f}" if cpu.get("cpu_quota_cores") else "none",
    "CPUSWEEP": ", ".join(f"{k}: {v['pairs_per_s']:.1f}" for k, v in cpu["sweep"].items()),
    "RATIO": f"{b['value'] / cpu['value']:.0f}",
    "CPUP": f"{cpu.get('value_parity_build', float('nan')):.1f}", "CPUP1": f"{cpu.get('single_thread_parity_build', float('nan')):.2f}",
    "CPUD": f"{cpu.get('max_abs_dflow_fast_vs_parity', float('nan')):.1e}",
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
