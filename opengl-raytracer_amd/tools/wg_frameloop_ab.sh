#!/bin/bash
# Frame-by-frame loop and batched render with smaller workgroups (libraries under csrc/diag built with -DPT_BLOCK=...):
# finer-grained workgroups free their LDS and wave slots earlier at the end of a single-frame launch.
run() { # lib frame_blocks
  printf "%-12s frame_blocks %s: " $(basename $1 .so) $2
  RT_HIP_LIB=$1 RT_TUNE_FRAME_BLOCKS=$2 timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene dragon80k --frames 128 --warmup 32 --per-frame 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //' | tr '\n' ' '
  RT_HIP_LIB=$1 timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene dragon80k --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
}
D=/root/repo/opengl-raytracer_amd/csrc
for round in 1 2; do
run $D/librt_hip.so 1
run $D/diag/b512.so 2
run $D/diag/b512.so 1
run $D/diag/b512_l4.so 1
run $D/diag/b512_l4.so 2
run $D/diag/b256.so 4
run $D/diag/b256.so 2
run $D/diag/b256_l4.so 2
run $D/diag/b256_l4.so 4
done
