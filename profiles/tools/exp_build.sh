#!/bin/bash
# compile-time variants: bash profiles/tools/exp_build.sh "label|EXTRA flags|bench args" ...   (rebuilds kernels_farneback.o per variant)
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  IFS='|' read -r label extra args <<< "$spec"
  (cd funscript_flow_amd/csrc && rm -f kernels_farneback.o && make EXTRA="$extra" > /dev/null 2>&1) || { echo "$label build FAILED"; exit 1; }
  TOP=${TOP:-8} bash profiles/tools/exp.sh "$label|$args" 2>&1 | grep -v "Traceback\|BrokenPipe\|File \"\|print(f"
done
(cd funscript_flow_amd/csrc && rm -f kernels_farneback.o && make > /dev/null 2>&1)
