# Stall breakdown of the path kernel: three SQ counter passes (each its own run) over the same workload.
# usage: bash pmc_paths_stalls.sh <scene> <outdir>
set -e
SCENE=${1:-dragon80k}
OUT=${2:-/root/repo/gpurun_out/pmc_stalls}
cd /tmp && export TMPDIR=/tmp
PR="python3 /root/repo/opengl-raytracer_amd/tools/prof_render.py --scene $SCENE --frames 8 --warmup 8"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/a -o p --output-format csv -- $PR > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $OUT/b -o p --output-format csv -- $PR > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAIT_INST_LDS -d $OUT/c -o p --output-format csv -- $PR > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU -d $OUT/d -o p --output-format csv -- $PR > /dev/null 2>&1
cd /root/repo && python3 opengl-raytracer_amd/tools/rocprof_summary.py $OUT/a $OUT/b $OUT/c $OUT/d | grep -E "k_paths"
