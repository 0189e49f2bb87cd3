#!/bin/bash
# Phase-switch thresholds of queue launches re-swept with bench.py (128 steps) through the tuning build.
# usage (GPU box, repo root): tune_queue2.sh [bench.py args]
L=$PWD/opengl-raytracer_amd/csrc/librt_hip_tune.so
for round in 1 2; do for t in 6 8 10 12 16; do for sv in 4 6 8; do
  printf "trav %-3s service %-3s " $t $sv
  RT_TUNE_MIN_TRAV=$t RT_TUNE_MIN_SERVICE=$sv RT_HIP_LIB=$L timeout -k 10 200 python bench.py --steps 128 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_frame'])" || exit 1
done; done; done
