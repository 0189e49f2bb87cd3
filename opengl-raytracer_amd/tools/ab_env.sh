#!/bin/bash
# A/B of one tuning variable of the tuning build with bench.py, at several step counts, two rounds.
# usage (GPU box, repo root): ab_env.sh <VARIABLE> "<values>" "<step counts>" [bench.py args]
V=$1; VALS=$2; STEPS=$3; shift 3
L=$PWD/opengl-raytracer_amd/csrc/librt_hip_tune.so
for round in 1 2; do for ST in $STEPS; do for x in $VALS; do
  printf "%-5s %s=%-4s " $ST $V $x
  env $V=$x RT_HIP_LIB=$L timeout -k 10 200 python bench.py --steps $ST --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['region_ms'], d['kernels_ms_per_frame'])" || exit 1
done; done; done
