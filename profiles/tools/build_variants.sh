#!/bin/bash
# Build kernel-code variants of libffl_hip.so side by side for profiles/tools/ab_libs.py (runs here: hipcc cross-compiles).
#   bash profiles/tools/build_variants.sh "name|EXTRA flags" ...   ->  profiles/tools/_ab/lib_<name>.so   (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../.."
C=funscript_flow_amd/csrc
mkdir -p profiles/tools/_ab
make -C $C -j4 libffl_hip.so > /dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result -Wno-unused-value"
for spec in "$@"; do
  IFS='|' read -r name extra <<< "$spec"
  /opt/rocm/bin/hipcc $FLAGS $extra -Rpass-analysis=kernel-resource-usage -c $C/kernels_farneback.hip -o profiles/tools/_ab/kf_$name.o 2> profiles/tools/_ab/kf_$name.log
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o profiles/tools/_ab/lib_$name.so $C/ffl_api.o profiles/tools/_ab/kf_$name.o $C/kernels_post.o $C/kernels_frontend.o
  echo "== $name: $extra"
  grep -A12 "Function Name: _Z12k_blur_solve" profiles/tools/_ab/kf_$name.log | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //' | paste - - - - - | sed 's/\[-Rpass[^]]*\]//g' | cut -c1-260
done
