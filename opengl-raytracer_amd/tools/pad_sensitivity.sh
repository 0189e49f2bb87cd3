#!/bin/bash
# What does one more instruction in the traversal loop of k_paths cost?  Builds the library with 16 extra independent
# v_nop / s_nop per inner-node step into csrc/diag/ (run this part where hipcc is), then times them against the product
# library on the GPU box with ab_render.sh.  usage: pad_sensitivity.sh build | run
set -e
cd "$(dirname "$0")/../csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -fno-slp-vectorize -Wno-unused-function -I../../include -shared"
if [ "$1" = build ]; then
  mkdir -p diag
  /opt/rocm/bin/hipcc $F '-DPT_PAD_VNOP="16"' -o diag/pad_vnop16.so rt_api.hip &
  /opt/rocm/bin/hipcc $F '-DPT_PAD_SNOP="16"' -o diag/pad_snop16.so rt_api.hip &
  wait
else
  cd ../.. && bash opengl-raytracer_amd/tools/ab_render.sh dragon80k
fi
