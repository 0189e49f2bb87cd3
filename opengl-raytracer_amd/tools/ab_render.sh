# A/B timing of render-library builds on one box: every library under csrc/diag/ plus the product one, two rounds each.
# usage: [AB_ARGS="--per-frame"] bash ab_render.sh [scene ...]
set -e
SCENES=${@:-dragon80k}
for round in 1 2; do
for s in $SCENES; do
for L in /root/repo/opengl-raytracer_amd/csrc/librt_hip.so /root/repo/opengl-raytracer_amd/csrc/diag/*.so; do
  [ -f "$L" ] || continue
  printf "%s %-14s " "$s" "$(basename $L)"
  RT_HIP_LIB=$L timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $s --frames 128 --warmup 32 $AB_ARGS 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done
done
done
