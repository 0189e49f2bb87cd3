#!/bin/bash
# rocprofv3 around the SAME command the driver runs (python3 bench.py ...): one kernel-trace/stats run and the two
# HBM-traffic PMC passes, each in a run of its own.  Usage: profile_bench.sh <tag> [bench.py args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_under_trace.json" 2> "$out/trace.log" || { echo "trace run failed"; tail -5 "$out/trace.log"; exit 1; }
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- python3 bench.py --no-cpu-baseline "$@" > "$out/pmc$i.json" 2> "$out/pmc$i.log" || { echo "pmc pass $i failed"; tail -3 "$out/pmc$i.log"; }
done
python3 opengl-raytracer_amd/tools/rocprof_summary.py "$out"/trace "$out"/pmc* > "$out/summary.txt" 2>&1
find "$out" -name "*.csv" -size +2M -delete
grep -E "k_paths|k_accumulate|kernel  " "$out/summary.txt"
