set -e
for cfg in "256 8" "256 4" "128 8" "128 4" "256 0"; do
  set -- $cfg
  echo "== small_max $1 tiny $2"
  RT_BUILD_SMALL_MAX=$1 RT_BUILD_TINY=$2 timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_build.py --scene grid --repeat 1 --rebuild 2 2>&1 | grep -E "rebuild|Error|error"
done
