cd "$GRAFT_REPO_ROOT"
for r in 0 1 2 4 8; do
  python bench.py --no-extras --no-cpu-baseline --steps 40 --no-events --width 256 --height 256 --batch 256 --blur-rows $r 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x256 B=256 blur_rows $r', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
done
for b in 128 256; do
  python bench.py --no-extras --no-cpu-baseline --steps 40 --no-events --width 256 --height 256 --batch $b --lanes 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x256 B=$b lanes 2', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
done
cd "$GRAFT_REPO_ROOT"
for l in 1 2 3 4; do
  python bench.py --no-extras --no-cpu-baseline --steps 60 --no-events --width 256 --height 256 --batch 256 --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x256 B=256 lanes $l', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
done
for l in 2 3; do
  python bench.py --no-extras --no-cpu-baseline --steps 60 --no-events --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1080p B=32 lanes $l (no events)', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
done
