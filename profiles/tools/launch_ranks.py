#!/usr/bin/env python3
"""SUPERSEDED in round 4 by funscript_flow_amd/launch.py: `python bench.py --gpus N` now spawns its own ranks, relays rank 0's
JSON line and stops the others on the first failure.  Kept because the round-3 rehearsal records name it.

Start N ranks of a script with the torch.distributed environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), without
torch.distributed.run: the launcher itself then never opens the GPU, so a 1-GPU rehearsal box (at most 6 processes may
have the card open) can take 6 ranks.  Exit code = the first non-zero rank exit code.

    python profiles/tools/launch_ranks.py 6 29611 bench.py --gpus 6 --rehearse-gloo --steps 10
"""
import os
import subprocess
import sys


def main():
    n, port, script = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[4:], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    sys.exit(rc)


if __name__ == "__main__":
    main()
