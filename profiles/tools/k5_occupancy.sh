# timing-only experiment: k_blur_solve register budget (waves per SIMD) vs step time
set -e
cd funscript_flow_amd/csrc
for wv in 4 5 3 2; do
  rm -f kernels_farneback.o; make EXTRA=-DFFL_K5_WAVES=$wv >/dev/null 2>&1
  (cd ../..; timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('K5_WAVES=$wv', round(d['value']), round(d['ms_per_step'],3), 'K5 avg launch ms', round(r['avg_launch_ms'],4))")
done
