# Sweep of the path kernel's scheduling knobs through the RT_TUNE_* environment variables (one box).
# usage: bash tune_paths.sh [scene ...]
for S in ${@:-dragon80k}; do
run() { printf "%s %-40s " $S "$*"; env "$@" timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'; }
for round in 1 2; do
run RT_TUNE_MIN_TRAV=40
run RT_TUNE_MIN_TRAV=32
run RT_TUNE_MIN_TRAV=36
run RT_TUNE_MIN_TRAV=44
run RT_TUNE_MIN_TRAV=48
run RT_TUNE_MIN_SERVICE=8
run RT_TUNE_MIN_SERVICE=24
run RT_TUNE_W_INNER=2 RT_TUNE_W_LEAF=3
run RT_TUNE_W_INNER=1 RT_TUNE_W_LEAF=2
done
done
