#!/bin/bash
# The ramp of the frames traced ahead of a steady frame loop, swept through the tuning build: after how many steady calls
# batches grow to 4 and to RT_TUNE_AHEAD frames.  usage (GPU box, repo root): tune_ahead.sh
L=$PWD/opengl-raytracer_amd/csrc/librt_hip_tune.so
for round in 1 2; do for ST in 20 64 256; do for cfg in "6 40 8" "4 12 8" "4 8 8" "3 6 8" "4 12 16" "6 40 16"; do
  set -- $cfg
  printf "%-4s t4 %-3s t8 %-3s max %-3s " $ST $1 $2 $3
  RT_TUNE_AHEAD_T4=$1 RT_TUNE_AHEAD_T8=$2 RT_TUNE_AHEAD=$3 RT_HIP_LIB=$L timeout -k 10 200 python bench.py --workload dragon80k_1080p_frame_loop --steps $ST --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['region_ms'])" || exit 1
done; done; done
