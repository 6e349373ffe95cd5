#!/bin/bash
# Move one capture (tools/capture_round.sh + capture_extra.sh, merged back under gpurun_out/) into profiles/ as THE
# end-of-round set: removes the previous set <old>, copies <new>, installs its traffic.json and regenerates DESIGN.md's
# numbers table.  Usage (build container): bash profiles/tools/install_capture.sh r02_f r02_g
set -e
cd "$(dirname "$0")/../.."
OLD=$1; NEW=$2
FILES="bench.json bench_under_rocprof.json kernel_stats.csv kernel_trace_by_grid.txt pmc_fetch_write.txt bench_4k.json bench_2880_eye.json bench_lanes2.json bench_b8.json bench_640.json bench_256_b256.json bench_256_b64.json bench_zoom005.json bench_independent.json"
for f in $FILES; do test -f gpurun_out/${NEW}_$f || { echo "missing gpurun_out/${NEW}_$f"; exit 1; }; done
for f in $FILES; do git rm -q --cached profiles/${OLD}_$f 2>/dev/null || true; rm -f profiles/${OLD}_$f; cp gpurun_out/${NEW}_$f profiles/; done
test -f profiles/${OLD}_batch_sweep.txt && git mv -f profiles/${OLD}_batch_sweep.txt profiles/${NEW}_batch_sweep.txt
cp gpurun_out/${NEW}_traffic.json profiles/traffic.json
sed -i "s/${OLD}/${NEW}/g" profiles/README.md
python profiles/tools/fill_design.py $NEW
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
tj = json.load(open("profiles/traffic.json"))
print("traffic.json signature", tj["kernel_signature"], "sources", bench.kernel_signature(), "match" if tj["kernel_signature"] == bench.kernel_signature() else "STALE")
d = json.load(open("profiles/${NEW}_bench.json"))
print("bench", round(d["value"]), "pairs/s  frac", round(d["roofline"]["frac"], 3), " events us", round(d["roofline"]["avg_launch_ms"] * 1e3, 1), " traffic MB", round(tj["hbm_bytes_per_launch"] / 1e6))
PY
