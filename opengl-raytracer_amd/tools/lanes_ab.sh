#!/bin/bash
# Internal streams / staging slots (libraries under csrc/diag built with -DRT_LANES=n): frame-by-frame loop and batched render.
D=/root/repo/opengl-raytracer_amd/csrc
for round in 1 2; do for L in $D/librt_hip.so $D/diag/*.so; do
  printf "%-12s " $(basename $L .so)
  RT_HIP_LIB=$L timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene dragon80k --frames 256 --warmup 32 --per-frame 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //' | tr '\n' ' '
  RT_HIP_LIB=$L timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene dragon80k --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done; done
