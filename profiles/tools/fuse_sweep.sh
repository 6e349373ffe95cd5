#!/bin/bash
# which levels take the folded first iteration: fuse_first = minimum tiles x pairs of a level (default 10000)
cd "$GRAFT_REPO_ROOT"
for f in 10000 4000 1500 500; do
  python bench.py --no-extras --no-cpu-baseline --steps 40 --fuse-first $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1080p fuse_first $f', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
  python bench.py --no-extras --no-cpu-baseline --steps 40 --no-events --width 256 --height 256 --batch 256 --fuse-first $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x256 B=256 fuse_first $f', round(d['value']), round(d['ms_per_step'],3), d['checked'])"
done
