#!/bin/bash
# compile-time variants of kernels_post.hip: bash profiles/tools/exp_build_post.sh "label|EXTRA flags|bench args" ...
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  IFS='|' read -r label extra args <<< "$spec"
  (cd funscript_flow_amd/csrc && rm -f kernels_post.o && make EXTRA="$extra" > /dev/null 2>&1) || { echo "$label build FAILED"; exit 1; }
  TOP=${TOP:-12} bash profiles/tools/exp.sh "$label|$args" 2>&1 | grep -E "^==|k_pass1 |k_radial "
done
(cd funscript_flow_amd/csrc && rm -f kernels_post.o && make > /dev/null 2>&1)
