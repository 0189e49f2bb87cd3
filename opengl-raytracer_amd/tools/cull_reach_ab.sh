#!/bin/bash
# The path kernel's cull pad: constant (RT_TUNE_CULL_REACH=0) against growing with the box's reach from the ray origin (=1).
set -e
for s in ${@:-dragon80k teapot grid}; do
for round in 1 2; do
for v in 0 1; do
  printf "%s reach %s " $s $v
  RT_TUNE_CULL_REACH=$v timeout -k 10 300 python -u opengl-raytracer_amd/tools/prof_render.py --scene $s --frames 64 --warmup 16 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done
done
done
