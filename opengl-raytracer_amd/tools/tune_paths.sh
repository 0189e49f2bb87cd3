# Sweep of the path kernel's step-vote weights through the RT_TUNE_* developer knobs (one box).
# usage: bash tune_paths.sh [scene ...]
for S in ${@:-dragon80k}; do
run() { printf "%s %-40s " $S "$*"; env "$@" timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 128 --warmup 32 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'; }
run RT_TUNE_W_INNER=2 RT_TUNE_W_LEAF=3
run RT_TUNE_W_INNER=3 RT_TUNE_W_LEAF=5
run RT_TUNE_W_INNER=4 RT_TUNE_W_LEAF=7
run RT_TUNE_W_INNER=5 RT_TUNE_W_LEAF=9
run RT_TUNE_W_INNER=5 RT_TUNE_W_LEAF=8
run RT_TUNE_W_INNER=1 RT_TUNE_W_LEAF=2
run RT_TUNE_W_INNER=4 RT_TUNE_W_LEAF=9
run RT_TUNE_W_INNER=2 RT_TUNE_W_LEAF=5
run RT_TUNE_W_INNER=3 RT_TUNE_W_LEAF=5 RT_TUNE_MIN_TRAV=36
run RT_TUNE_W_INNER=3 RT_TUNE_W_LEAF=5 RT_TUNE_MIN_SERVICE=24
run RT_TUNE_W_INNER=2 RT_TUNE_W_LEAF=3
done
