#!/bin/bash
# A/B of the path kernel's subtree cull: ms per frame with the scene's own slack (default), without the cull (-1) and
# with smaller fixed slacks (what a tighter pad would buy).
set -e
for s in ${@:-dragon80k teapot}; do
for round in 1 2; do
for v in default -1 1e-2 2e-3 2e-5; do
  printf "%s slack %-8s " $s $v
  if [ $v = default ]; then unset RT_TUNE_CULL_SLACK; else export RT_TUNE_CULL_SLACK=$v; fi
  timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $s --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done
done
done
