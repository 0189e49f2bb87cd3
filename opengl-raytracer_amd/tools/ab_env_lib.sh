#!/bin/bash
# As ab_env.sh, with the library named: ab_env_lib.sh <library> <VARIABLE> "<values>" "<step counts>" [bench.py args]
L=$1; V=$2; VALS=$3; STEPS=$4; shift 4
for round in 1 2; do for ST in $STEPS; do for x in $VALS; do
  printf "%-5s %s %s=%-4s " $ST $(basename $L) $V $x
  env $V=$x RT_HIP_LIB=$L timeout -k 10 200 python bench.py --steps $ST --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['region_ms'], d['kernels_ms_per_frame'])" || exit 1
done; done; done
