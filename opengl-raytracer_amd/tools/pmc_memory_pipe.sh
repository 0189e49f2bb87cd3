#!/bin/bash
# How busy are the texture-address unit (TA), the vector L1 (TCP) and the LDS under the path kernel?  Three PMC passes over
# the same render, each in its own process.  usage (GPU box, from the repo root): pmc_memory_pipe.sh [scene]
S=${1:-dragon80k}
OUT=$PWD/gpurun_out/pmc_mempipe_$S
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PR="python3 $OLDPWD/opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 16 --warmup 16"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE -d $OUT/a -o p --output-format csv -- $PR > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $OUT/b -o p --output-format csv -- $PR > $OUT/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES -d $OUT/c -o p --output-format csv -- $PR > $OUT/c.log 2>&1
cd $OLDPWD
python3 opengl-raytracer_amd/tools/rocprof_summary.py $OUT/a $OUT/b $OUT/c | grep -E "k_paths|==" | sed 's/void k_paths<false, false>(DScene, DCamera, PathsArgs)/k_paths/'
