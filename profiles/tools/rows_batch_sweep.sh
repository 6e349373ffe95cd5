cd "$GRAFT_REPO_ROOT"
for r in 4 6 8 10 12 16; do
  python bench.py --no-extras --no-cpu-baseline --steps 40 --blur-rows $r 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows $r', round(d['value']), d['ms_per_step'], d['checked'])"
done
for b in 24 32 40 48 64; do
  python bench.py --no-extras --no-cpu-baseline --steps 40 --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value']), d['ms_per_step'], d['checked'])"
done
