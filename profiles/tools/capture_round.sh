#!/bin/bash
# Capture the judged artefacts of the default bench command: bench line, rocprofv3 kernel stats, HBM
# traffic PMC passes.  Usage (on the GPU box, via gpurun): bash profiles/tools/capture_round.sh r01_g
# Writes into gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards.
set -e
TAG=${1:-cap}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o s -- python3 bench.py --no-cpu-baseline --no-extras > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_F -o f -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_W -o w -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp profiles/traffic.json /tmp/traffic_prev.json
BATCH=$(python -c "import json; print(json.load(open('$O/${TAG}_bench_under_rocprof.json'))['config']['pairs_per_step'])")
python profiles/make_traffic.py $O/${TAG}_F/f_counter_collection.csv $O/${TAG}_W/w_counter_collection.csv k_blur_solve 1920x1080 $BATCH $TAG
# the other workloads of the bench line: configs[2] / the configs[4] eye (B = 32, one lane) and the reference's 256x256
# operating point (B = 256): FETCH / WRITE passes -> their own traffic.json entries (bench.py: large_image / small_image)
for WL in "3840 2160 32" "2880 2880 32" "256 256 256"; do
  set -- $WL
  N=${1}x${2}
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_F_$N -o f -- python3 bench.py --width $1 --height $2 --batch $3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_W_$N -o w -- python3 bench.py --width $1 --height $2 --batch $3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python profiles/make_traffic.py $O/${TAG}_F_$N/f_counter_collection.csv $O/${TAG}_W_$N/w_counter_collection.csv k_blur_solve $N $3 $TAG
  # kernel durations of THIS workload (SURVEY 7 hard part 5: 4K is the HBM-roofline run; 256x256 the reference's operating
  # point): the same command under --kernel-trace --stats, so large_image / small_image roofline fractions can be
  # recomputed from profiles/ without trusting HIP events
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$N -o s -- python3 bench.py --width $1 --height $2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/${TAG}_bench_under_rocprof_$N.json 2>/dev/null
  cp $O/${TAG}_stats_$N/s_kernel_stats.csv $O/${TAG}_kernel_stats_$N.csv
  python profiles/summarize_trace.py $O/${TAG}_stats_$N/s_kernel_trace.csv > $O/${TAG}_kernel_trace_by_grid_$N.txt
  (python profiles/summarize_pmc.py $O/${TAG}_F_$N/f_counter_collection.csv k_; python profiles/summarize_pmc.py $O/${TAG}_W_$N/w_counter_collection.csv k_) > $O/${TAG}_pmc_fetch_write_$N.txt
  echo "traffic $N done"
done
cp profiles/traffic.json $O/${TAG}_traffic.json
# the bench line last, so that it already carries this capture's traffic (profiles/traffic.json, same kernel signature)
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python profiles/summarize_trace.py $O/${TAG}_stats/s_kernel_trace.csv > $O/${TAG}_kernel_trace_by_grid.txt
cp $O/${TAG}_stats/s_kernel_stats.csv $O/${TAG}_kernel_stats.csv
(python profiles/summarize_pmc.py $O/${TAG}_F/f_counter_collection.csv k_; python profiles/summarize_pmc.py $O/${TAG}_W/w_counter_collection.csv k_) > $O/${TAG}_pmc_fetch_write.txt
head -12 $O/${TAG}_kernel_stats.csv
