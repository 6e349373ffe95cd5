#!/bin/bash
# Copy the judged artefacts of a capture_round.sh run from gpurun_out/ into profiles/ under the round's fixed names.
#   bash profiles/tools/adopt_capture.sh r04_c [old_tag_to_remove]
set -e
cd "$(dirname "$0")/../.."
T=$1; OLD=$2; G=gpurun_out
[ -n "$OLD" ] && rm -f profiles/${OLD}_bench.json profiles/${OLD}_bench_under_rocprof*.json profiles/${OLD}_kernel_stats.csv profiles/${OLD}_kernel_trace_by_grid.txt profiles/${OLD}_pmc_fetch_write*.txt
for f in bench.json bench_under_rocprof.json kernel_stats.csv kernel_trace_by_grid.txt pmc_fetch_write.txt; do cp $G/${T}_$f profiles/; done
for n in 3840x2160 2880x2880 256x256; do cp $G/${T}_pmc_fetch_write_$n.txt $G/${T}_bench_under_rocprof_$n.json profiles/; done
R=${T%%_*}
cp $G/${T}_kernel_stats_3840x2160.csv profiles/${R}_kernel_stats_4k.csv
cp $G/${T}_kernel_stats_256x256.csv profiles/${R}_kernel_stats_256_b256.csv
cp $G/${T}_kernel_stats_2880x2880.csv profiles/${R}_kernel_stats_2880_eye.csv
cp $G/${T}_kernel_trace_by_grid_3840x2160.txt profiles/${R}_kernel_trace_by_grid_4k.txt
cp $G/${T}_kernel_trace_by_grid_256x256.txt profiles/${R}_kernel_trace_by_grid_256_b256.txt
cp $G/${T}_traffic.json profiles/traffic.json
python profiles/tools/fill_design.py $T
python - "$T" <<'PY'
import csv, json, sys
T = sys.argv[1]; R = T.split("_")[0]
for n, f in (("3840x2160", f"{R}_kernel_stats_4k.csv"), ("2880x2880", f"{R}_kernel_stats_2880_eye.csv"), ("256x256", f"{R}_kernel_stats_256_b256.csv")):
    tot = calls = 0
    for r in csv.DictReader(open("profiles/" + f)):
        if "k_blur_solve" in r["Name"]:
            tot += int(r["TotalDurationNs"]); calls += int(r["Calls"])
    ro = json.load(open(f"profiles/{T}_bench_under_rocprof_{n}.json"))["roofline"]
    us = tot / calls / 1e3
    print(f"{n}: k_blur_solve {calls} launches, rocprofv3 average {us:.1f} us (HIP events {ro['avg_launch_ms'] * 1e3:.1f}); algorithmic {ro['alg_bytes_per_launch'] / 1e6:.1f} MB "
          f"-> {ro['alg_bytes_per_launch'] / us / 1e6:.2f} TB/s = {ro['alg_bytes_per_launch'] / us / 1e6 / 8:.3f}; traffic {ro['traffic'] / 1e6:.0f} MB -> {ro['traffic'] / us / 1e6 / 8:.3f}")
for r in csv.DictReader(open(f"profiles/{T}_kernel_stats.csv")):
    if "k_blur_solve" in r["Name"]:
        print(r["Name"][5:25], r["Calls"], f"{float(r['AverageNs']) / 1e3:.1f} us")
PY
head -4 profiles/${T}_kernel_trace_by_grid.txt
