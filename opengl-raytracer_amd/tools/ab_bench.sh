#!/bin/bash
# A/B of render-library builds with bench.py: the product library and every library under csrc/diag/, at the driver's 20 steps
# and at 128, two rounds.  usage (GPU box, repo root): ab_bench.sh [bench.py args]
for round in 1 2; do for ST in ${AB_STEPS:-20 128}; do
for L in $PWD/opengl-raytracer_amd/csrc/librt_hip.so $PWD/opengl-raytracer_amd/csrc/diag/*.so; do
  [ -f "$L" ] || continue
  printf "%-5s %-16s " $ST "$(basename $L)"
  RT_HIP_LIB=$L timeout -k 10 200 python bench.py --steps $ST --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['region_ms'], d['kernels_ms_per_frame'])"
done; done; done
