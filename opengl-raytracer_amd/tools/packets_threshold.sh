#!/bin/bash
# Where do tile packets stop paying?  The same scene at falling resolutions (fewer pixels per triangle), packets forced off / on
# through the tuning build; ms per frame over 128-frame calls.  usage (GPU box, repo root): packets_threshold.sh [scene]
S=${1:-dragon80k}
for wh in "1920 1080" "1440 810" "1280 720" "960 540" "800 450" "640 360" "480 270"; do set -- $wh
  for pk in 0 2; do
    printf "%s %sx%s packets=%s  " $S $1 $2 $pk
    RT_TUNE_PACKETS=$pk timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $S --width $1 --height $2 --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
  done
done
