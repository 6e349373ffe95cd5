cd "$GRAFT_REPO_ROOT"
for l in 1 2 3; do for e in 0 1 2; do
python bench.py --no-cpu-baseline --no-extras --no-events --lanes $l --expand $e --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lanes=$l expand=$e', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
done; done
