#!/bin/bash
# A/B of the true-direction cull of the inner-node step: ms per frame without it and with two slack values.
set -e
for s in dragon80k teapot; do
for v in off 2e-3 2e-5; do
  printf "%s slack %-6s " $s $v
  if [ $v = off ]; then unset RT_TUNE_CULL_SLACK; else export RT_TUNE_CULL_SLACK=$v; fi
  timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $s --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done
done
