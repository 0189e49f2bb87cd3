#!/bin/bash
# PMC passes over the render of the ~9.3M-triangle grid scene (each pass its own process; run from the repo root on
# the GPU box).  usage: pmc_grid.sh TAG [extra prof_render.py args]
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_grid_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PR="python3 $OLDPWD/opengl-raytracer_amd/tools/prof_render.py --scene grid --frames 8 --warmup 2 $*"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $PR > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $OUT/sq -o p --output-format csv -- $PR > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/tcc -o p --output-format csv -- $PR > $OUT/tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/tcp -o p --output-format csv -- $PR > $OUT/tcp.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- $PR > $OUT/fetch.log 2>&1
cd $OLDPWD
grep "ms per frame" $OUT/trace.log || true
python3 opengl-raytracer_amd/tools/rocprof_summary.py $OUT/sq $OUT/tcc $OUT/tcp $OUT/fetch | grep -E "k_paths|==" 
