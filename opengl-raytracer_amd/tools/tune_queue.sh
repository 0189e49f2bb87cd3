#!/bin/bash
# Phase-switch thresholds of the path kernel re-swept through the tuning build, for launches that start from k_primary's hit
# queues (default where packets are chosen) and for launches that trace their primary rays lane by lane (RT_TUNE_PACKETS=0).
# usage (GPU box, repo root): tune_queue.sh [scene ...]
for S in ${@:-dragon80k teapot}; do
run() { printf "%s %-60s " "$S" "$*"; env "$@" timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'; }
for round in 1 2; do
for t in 4 8 12 16 24 40; do for sv in 4 6 8 16; do
  run RT_TUNE_MIN_TRAV=$t RT_TUNE_MIN_SERVICE=$sv
  run RT_TUNE_PACKETS=0 RT_TUNE_MIN_TRAV=$t RT_TUNE_MIN_SERVICE=$sv
done; done
done
done
