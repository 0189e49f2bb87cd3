#!/bin/bash
# Instruction mix, wait cycles, lanes per vector instruction and vector-memory accesses of the kernels of the default pipeline
# (k_primary, k_paths): a kernel trace and four PMC passes over the same 16-frame render, each in its own process.  With
# RT_TUNE_PACKETS=0 in the environment the tuning build traces primary rays lane by lane inside k_paths: the "before" column.
# usage (GPU box, from the repo root): [RT_TUNE_PACKETS=0] pmc_primary.sh <tag> [scene]
T=$1
S=${2:-dragon80k}
OUT=$PWD/gpurun_out/pmc_primary_$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PR="python3 $OLDPWD/opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 16 --warmup 16"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $PR > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/a -o p --output-format csv -- $PR > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVES -d $OUT/b -o p --output-format csv -- $PR > $OUT/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT/c -o p --output-format csv -- $PR > $OUT/c.log 2>&1
cd $OLDPWD
python3 opengl-raytracer_amd/tools/rocprof_summary.py $OUT/trace $OUT/a $OUT/b $OUT/c | grep -E "k_paths|k_primary|k_rebase|k_accumulate|==|kernel "
