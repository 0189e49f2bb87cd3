#!/bin/bash
# Phase-switch thresholds of the path kernel with RT_FLAG_FIX_INVDIR (packets on by rule), bench.py 64 steps, tuning build.
L=$PWD/opengl-raytracer_amd/csrc/librt_hip_tune.so
for round in 1 2; do for t in 0 16 24 32 40 48; do for sv in 8 16 24; do
  printf "trav %-3s service %-3s " $t $sv
  RT_TUNE_MIN_TRAV=$t RT_TUNE_MIN_SERVICE=$sv RT_HIP_LIB=$L timeout -k 10 200 python bench.py --fix-invdir --steps 64 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_frame'])" || exit 1
done; done; done
