#!/bin/bash
# diagnostic: phase shares of the level-0 folded k_blur_solve launch from in-kernel stamps (a separate -DFFL_STAMP build)
cd "$GRAFT_REPO_ROOT"
(cd funscript_flow_amd/csrc && rm -f kernels_farneback.o && make EXTRA=-DFFL_STAMP > /dev/null 2>&1) || { echo build failed; exit 1; }
python profiles/tools/k5_stamps.py
(cd funscript_flow_amd/csrc && rm -f kernels_farneback.o && make > /dev/null 2>&1)
