#!/usr/bin/env python3
"""Rewrite the "end of round" row of profiles/README.md (the row that starts with `<old_tag>_*`) and the tag-dependent
numbers quoted in README.md / DESIGN.md from the files of capture <new_tag> (after adopt_capture.sh).
    python profiles/tools/readme_capture_row.py r04_d r04_e "note about replaced captures" """
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OLD, T = sys.argv[1], sys.argv[2]
NOTE = sys.argv[3] if len(sys.argv) > 3 else ""
P = os.path.join(ROOT, "profiles")
R = T.split("_")[0]
b = json.load(open(f"{P}/{T}_bench.json"))
st = {}
for r in csv.DictReader(open(f"{P}/{T}_kernel_stats.csv")):
    if "k_blur_solve" in r["Name"]:
        st[r["Name"][5:27]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]))
tot, calls = sum(v[2] for v in st.values()), sum(v[0] for v in st.values())
grid = {}
for line in open(f"{P}/{T}_kernel_trace_by_grid.txt"):
    m = re.match(r"(\S.*?)\s+\((\d+), 1, 1\)\s+(\d+)\s+([\d.]+)", line)
    if m:
        grid[(m.group(1).strip(), int(m.group(2)))] = float(m.group(4))
l0 = max(g for (k, g) in grid if k.startswith("k_blur_solve<true, 1>"))


def ws(n, f):
    t = c = 0
    for r in csv.DictReader(open(f"{P}/{f}")):
        if "k_blur_solve" in r["Name"]:
            t += int(r["TotalDurationNs"]); c += int(r["Calls"])
    ro = json.load(open(f"{P}/{T}_bench_under_rocprof_{n}.json"))["roofline"]
    us = t / c / 1e3
    return us, ro["avg_launch_ms"] * 1e3, ro["alg_bytes_per_launch"] / us / 1e6, ro["alg_bytes_per_launch"] / us / 1e6 / 8, ro["traffic"] / 1e6, ro["traffic"] / us / 1e6 / 8


k4, eye, s2 = ws("3840x2160", f"{R}_kernel_stats_4k.csv"), ws("2880x2880", f"{R}_kernel_stats_2880_eye.csv"), ws("256x256", f"{R}_kernel_stats_256_b256.csv")
pc, c, si = b["pcie_inclusive"], b["cpu_baseline"], b["small_image"]
row = (f"| `{T}_*`, `{R}_kernel_stats_4k.csv`, `{R}_kernel_stats_256_b256.csv`, `{R}_kernel_stats_2880_eye.csv`, `{R}_kernel_trace_by_grid_4k.txt`, "
       f"`{R}_kernel_trace_by_grid_256_b256.txt`, `{T}_bench_under_rocprof_*.json`, `traffic.json` | **end of round {int(R[1:])}** (`tools/capture_round.sh {T}` + "
       f"`tools/adopt_capture.sh {T}` + `tools/readme_capture_row.py`; the capture script takes a `--kernel-trace --stats` pass per workload; device-source signature "
       f"`{b['config']['kernel_signature']}`; the GPU tests green on the same sources first): bench **{b['value']:.0f} pairs/s** ({b['ms_per_step']:.2f} ms per step), "
       f"`k_blur_solve` {b['roofline']['avg_launch_ms'] * 1e3:.1f} µs by HIP events vs {tot / calls / 1e3:.1f} µs by rocprofv3 (`{T}_kernel_stats.csv`: "
       + " + ".join(f"{x[0]} × {x[1]:.1f}" for x in st.values()) + f" µs over {calls} launches), **{b['roofline']['frac']:.3f}** algorithmic, traffic "
       f"{b['roofline']['traffic'] / 1e6:.0f} MB per launch → `frac_measured` {b['roofline']['frac_measured']:.3f}; level 0 "
       f"{grid[('k_blur_solve<true, 1>', l0)]:.0f} / {grid[('k_blur_solve<true, 0>', l0)]:.0f} / {grid[('k_blur_solve<false, 0>', l0)]:.0f} µs. "
       f"**Recomputing the other workloads' fractions without HIP events:** 3840×2160 B = 32 — `k_blur_solve` 84 launches, average **{k4[0]:.1f} µs** "
       f"(`{R}_kernel_stats_4k.csv`; HIP events in the same run {k4[1]:.1f}), algorithmic 8727.1 MB per launch → {k4[2]:.2f} TB/s = {k4[3]:.3f} of 8 TB/s, measured traffic "
       f"{k4[4]:.0f} MB → {k4[5]:.3f}; 2880×2880 eye — {eye[0]:.1f} µs (events {eye[1]:.1f}) → {eye[3]:.3f} / {eye[5]:.3f}; 256×256 B = 256 — 84 launches, average "
       f"**{s2[0]:.1f} µs** (`{R}_kernel_stats_256_b256.csv`; events {s2[1]:.1f}: the event pair costs ≈ 6 µs on a 100 µs launch), algorithmic 517.6 MB → {s2[2]:.2f} TB/s = "
       f"{s2[3]:.3f}, traffic {s2[4]:.0f} MB → {s2[5]:.3f}. Bench line extras: `large_image` 4K {b['large_image']['3840x2160']['value']:.0f} / eye "
       f"{b['large_image']['2880x2880_eye']['value']:.0f} pairs/s, `small_image` {si['value'] / 1e3:.1f} k pairs/s (graphs {si['graphs']['captured']} captured / "
       f"{si['graphs']['replayed']} replayed / {si['graphs']['capture_failures']} failed; from host frames {si['pcie_inclusive_gray']['value'] / 1e3:.0f} k), `pcie_inclusive` "
       f"(one 3000-frame chunk) gray {pc['gray']['value']:.0f} ({pc['gray']['vs_resident']:.2f}×) / BGR {pc['bgr']['value']:.0f} ({pc['bgr']['vs_resident']:.2f}×) / BGR pinned "
       f"{pc['bgr_pinned']['value']:.0f} ({pc['bgr_pinned']['vs_resident']:.2f}×), `cpu_baseline` **{c['value']:.1f} pairs/s** on the `-O3 -march=native` build "
       f"({c['cores']} workers; {c['single_thread']:.2f} single) with `value_parity_build` {c['value_parity_build']:.1f} ({c['single_thread_parity_build']:.2f} single) and "
       f"max \\|Δflow\\| {c['max_abs_dflow_fast_vs_parity']:.1e} px between the builds. {NOTE} |")
p = f"{P}/README.md"
s = open(p).read()
i = s.index(f"| `{OLD}_*`")
s = s[:i] + row + s[s.index("\n", i):]
open(p, "w").write(s)
for p in (f"{ROOT}/README.md", f"{ROOT}/DESIGN.md"):
    s = open(p).read().replace(f"{OLD}_bench.json", f"{T}_bench.json").replace(f"`{OLD}_kernel_stats.csv`", f"`{T}_kernel_stats.csv`")
    if p.endswith("DESIGN.md"):
        s = re.sub(r"`k_blur_solve` averages [\d.]+ µs at 4K \(HIP events in the same run: [\d.]+\) and [\d.]+ µs at 256² B = 256 \(events [\d.]+:",
                   f"`k_blur_solve` averages {k4[0]:.1f} µs at 4K (HIP events in the same run: {k4[1]:.1f}) and {s2[0]:.1f} µs at 256² B = 256 (events {s2[1]:.1f}:", s)
        s = re.sub(r"\([\d.]+ pairs/s at the \d+-worker knee of the box\), `value_parity_build` \([\d.]+\), `single_thread` \([\d.]+ vs [\d.]+\), `max_abs_dflow_fast_vs_parity` \([\d.e-]+ px\)",
                   f"({c['value']:.1f} pairs/s at the {c['cores']}-worker knee of the box), `value_parity_build` ({c['value_parity_build']:.1f}), `single_thread` "
                   f"({c['single_thread']:.2f} vs {c['single_thread_parity_build']:.2f}), `max_abs_dflow_fast_vs_parity` ({c['max_abs_dflow_fast_vs_parity']:.1e} px)", s)
    open(p, "w").write(s)
print("row rewritten for", T)
