#!/bin/bash
# Primary rays traced tile by tile (k_primary + k_paths from the hit queues) against lane by lane (RT_TUNE_PACKETS=0; 2 forces packets where the rule would not choose them), through
# the tuning build, on every static workload.  usage (GPU box, repo root): packets_ab.sh
for round in 1 2; do
for s in "dragon80k" "teapot" "dragon80k --width 3840 --height 2160" "grid --frames 32 --warmup 16" "dragon80k --fix-invdir"; do
for cfg in "RT_TUNE_PACKETS=0" "RT_TUNE_PACKETS=2"; do
  printf "%-40s %-20s " "$s" "$cfg"
  env $cfg timeout -k 10 200 python -u opengl-raytracer_amd/tools/prof_render.py --frames 128 --warmup 32 --scene $s 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done; done; done
