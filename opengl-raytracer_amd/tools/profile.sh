#!/bin/bash
# Profiles the render loop on the GPU box: one kernel-trace run plus separate --pmc passes
# (never combined with other trace domains).  Usage: profile.sh <tag> [prof_render.py args...]
# Output: gpurun_out/prof_<tag>/..., summary in gpurun_out/prof_<tag>/summary.txt
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
R="python3 opengl-raytracer_amd/tools/prof_render.py $*"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $R > "$out/trace.log" 2>&1 || { echo "kernel-trace run failed"; tail -5 "$out/trace.log"; exit 1; }
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- $R > "$out/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -3 "$out/pmc$i.log"; }
done
python3 opengl-raytracer_amd/tools/rocprof_summary.py "$out"/trace "$out"/pmc* > "$out/summary.txt" 2>&1
# keep only the summaries (raw CSVs of many launches are large)
find "$out" -name "*.csv" -size +2M -delete
cat "$out/summary.txt"
