#!/bin/bash
# Step-vote weights of the path kernel (a triangle step runs when W_LEAF * leaf lanes > W_INNER * inner lanes).
for S in ${@:-dragon80k}; do for round in 1 2 3; do IFS=","; for w in ${VOTES:-4 7,3 8,1 3,1 4,1 5,1 8}; do IFS=" "
  set -- $w
  printf "%s w_inner %s w_leaf %s  " $S $1 $2
  RT_TUNE_W_INNER=$1 RT_TUNE_W_LEAF=$2 timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done; done; done
