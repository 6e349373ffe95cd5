#!/bin/bash
# the other configurations of BASELINE.json / profiles README, one bench line each (no extras, no CPU baseline)
cd "$GRAFT_REPO_ROOT"
TAG=${1:-cap}
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-extras"
$B --width 3840 --height 2160 --steps 8 > $O/${TAG}_bench_4k.json 2>/dev/null &&
$B --width 2880 --height 2880 --steps 8 > $O/${TAG}_bench_2880_eye.json 2>/dev/null &&
$B --lanes 2 > $O/${TAG}_bench_lanes2.json 2>/dev/null &&
$B --batch 8 --steps 60 > $O/${TAG}_bench_b8.json 2>/dev/null &&
$B --width 640 --height 360 --steps 60 > $O/${TAG}_bench_640.json 2>/dev/null &&
$B --width 256 --height 256 --batch 256 --steps 40 --no-events > $O/${TAG}_bench_256_b256.json 2>/dev/null &&
$B --width 256 --height 256 --batch 64 --steps 100 --no-events > $O/${TAG}_bench_256_b64.json 2>/dev/null &&
$B --zoom 0.05 > $O/${TAG}_bench_zoom005.json 2>/dev/null &&
$B --independent > $O/${TAG}_bench_independent.json 2>/dev/null
for f in $O/${TAG}_bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['value'], 1), 'pairs/s', round(d['ms_per_step'], 3), 'ms/step', 'frac', round(d['roofline']['frac'], 3), 'checked', d['checked'])
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
