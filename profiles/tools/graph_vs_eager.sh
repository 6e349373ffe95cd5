#!/bin/bash
# eager launches with the dominant kernel's events (the default bench) against graph replay without events
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  python bench.py --no-extras --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events/eager', round(d['value']), d['ms_per_step'])"
  python bench.py --no-extras --no-cpu-baseline --steps 60 --no-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no events/graph', round(d['value']), d['ms_per_step'])"
done
