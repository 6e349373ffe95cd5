#!/bin/bash
# rocprofv3 kernel stats of the other resolutions' bench command (same flags as capture_extra.sh)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-cap}
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_4k -o s -- python3 bench.py --no-cpu-baseline --no-extras --width 3840 --height 2160 --steps 8 > $O/${TAG}_bench_4k_under_rocprof.json 2>/dev/null &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_256 -o s -- python3 bench.py --no-cpu-baseline --no-extras --width 256 --height 256 --batch 256 --steps 40 --no-events > $O/${TAG}_bench_256_under_rocprof.json 2>/dev/null &&
cp $O/${TAG}_stats_4k/s_kernel_stats.csv $O/${TAG}_kernel_stats_4k.csv && cp $O/${TAG}_stats_256/s_kernel_stats.csv $O/${TAG}_kernel_stats_256_b256.csv &&
python profiles/summarize_trace.py $O/${TAG}_stats_4k/s_kernel_trace.csv > $O/${TAG}_kernel_trace_by_grid_4k.txt &&
python profiles/summarize_trace.py $O/${TAG}_stats_256/s_kernel_trace.csv > $O/${TAG}_kernel_trace_by_grid_256_b256.txt
