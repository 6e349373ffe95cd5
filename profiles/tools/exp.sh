#!/bin/bash
# A/B experiments on the GPU box: for every "label|bench args" argument run bench.py (short, no extras) under
# rocprofv3 --kernel-trace and print pairs/s + the per-(kernel, grid) launch durations of the big launches.
# PMC=1 adds FETCH_SIZE / WRITE_SIZE passes.  Usage: bash profiles/tools/exp.sh "base|" "order1|--tile-order 1"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/exp
mkdir -p $O
for spec in "$@"; do
  label=${spec%%|*}; args=${spec#*|}
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/$label -o t -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $args > $O/$label.json 2> $O/$label.err || { echo "$label FAILED"; tail -5 $O/$label.err; exit 1; }
  python - "$label" $O/$label.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"== {sys.argv[1]}: {d['value']:.0f} pairs/s  {d['ms_per_step']:.3f} ms/step  blur_solve {d['kernel_ms_per_step'].get('k_blur_solve', 0):.3f} ms  checked={d['checked']}")
PY
  python profiles/summarize_trace.py $O/$label/t_kernel_trace.csv | head -${TOP:-12}
  if [ -n "$PMC" ]; then
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${label}_F -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $args > /dev/null 2>&1 || exit 1
    timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${label}_W -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $args > /dev/null 2>&1 || exit 1
    python profiles/summarize_pmc.py $O/${label}_F/f_counter_collection.csv k_ | head -${TOP:-12}
    python profiles/summarize_pmc.py $O/${label}_W/w_counter_collection.csv k_ | head -${TOP:-12}
  fi
done
