cd "$GRAFT_REPO_ROOT"
for b in 16 24 32 48 64 96; do
python bench.py --no-cpu-baseline --no-extras --batch $b --steps $((640 / b)) --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$b', round(d['value']), round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3))"
done
