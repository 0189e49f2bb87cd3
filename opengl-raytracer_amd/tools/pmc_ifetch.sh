#!/bin/bash
# Instruction-fetch side of the path kernel: fetch requests, instruction-cache hits / busy cycles, and how the waves'
# cycles split into issuing / waiting.  usage (GPU box, repo root): pmc_ifetch.sh [scene]
S=${1:-dragon80k}
OUT=$PWD/gpurun_out/pmc_ifetch_$S
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PR="python3 $OLDPWD/opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 16 --warmup 16"
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $OUT/a -o p --output-format csv -- $PR > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES -d $OUT/b -o p --output-format csv -- $PR > $OUT/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_DCACHE_REQ SQC_DCACHE_BUSY_CYCLES -d $OUT/c -o p --output-format csv -- $PR > $OUT/c.log 2>&1
cd $OLDPWD
python3 opengl-raytracer_amd/tools/rocprof_summary.py $OUT/a $OUT/b $OUT/c | grep -E "k_paths|==" | sed 's/void k_paths<false, false, 0>(DScene, DCamera, PathsArgs)/k_paths/'

