#!/bin/bash
# How the launches of a call share the chip, through the tuning build: bench.py at the driver's 20 steps and at 128.
#   RT_TUNE_STAGGER=0       primary passes of consecutive launches start together (1, default: one after the other)
#   RT_TUNE_QUEUE_BLOCKS=2  path launches that start from the hit queues fill the chip (1, default: one workgroup per CU)
#   RT_TUNE_PACKETS=0       primary rays traced lane by lane inside the path kernel
# usage (GPU box, repo root): overlap_ab.sh
run() { printf "%-10s %-58s " "$ST" "$*"; env RT_HIP_LIB=$PWD/opengl-raytracer_amd/csrc/librt_hip_tune.so "$@" timeout -k 10 200 python bench.py --steps $ST --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['timing']['region_ms'])"; }
for round in 1 2; do
for ST in 20 128; do
run RT_TUNE_STAGGER=1
run RT_TUNE_STAGGER=0 RT_TUNE_QUEUE_BLOCKS=2
run RT_TUNE_STAGGER=1 RT_TUNE_QUEUE_BLOCKS=2
run RT_TUNE_STAGGER=0 RT_TUNE_QUEUE_BLOCKS=1
run RT_TUNE_PACKETS=0
done; done
