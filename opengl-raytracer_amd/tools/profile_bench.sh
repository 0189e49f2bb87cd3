#!/bin/bash
# rocprofv3 around the SAME command the driver runs (python3 bench.py ...): one kernel-trace/stats run and the PMC
# passes (HBM traffic: FETCH_SIZE, WRITE_SIZE; L2 view: TCP->TCC requests, TCC hits/misses; vector-memory address unit),
# each in a run of its own.
# Then writes this workload's entry of profiles/pmc_traffic.json, stamped with the build id of the profiled library, and a
# kernel summary.  The entry's key is the --workload among the bench.py args (default dragon80k_1080p), "+fix_invdir" appended
# when --fix-invdir is among them.
# Usage: profile_bench.sh <tag> <frames per launch> [bench.py args...]   (run from the repo root on the GPU box;
# give --steps and --warmup as multiples of <frames per launch> so that every launch is a full one)
set -u
tag=$1; shift
fpl=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
key=dragon80k_1080p; prev=
for a in "$@"; do [ "$prev" = "--workload" ] && key=$a; prev=$a; done
for a in "$@"; do [ "$a" = "--fix-invdir" ] && key=$key+fix_invdir; done
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_under_trace.json" 2> "$out/trace.log" || { echo "trace run failed"; tail -5 "$out/trace.log"; exit 1; }
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc$i" -- python3 bench.py --no-cpu-baseline "$@" > "$out/pmc$i.json" 2> "$out/pmc$i.log" || { echo "pmc pass $i failed"; tail -3 "$out/pmc$i.log"; exit 1; }
done
python3 opengl-raytracer_amd/tools/rocprof_summary.py "$out"/trace "$out"/pmc* > "$out/summary.txt" 2>&1
python3 opengl-raytracer_amd/tools/pmc_traffic.py "$out" "$fpl" profiles/pmc_traffic.json "$key" > "$out/pmc_traffic.log" 2>&1 || { echo "pmc_traffic failed"; tail -3 "$out/pmc_traffic.log"; exit 1; }
cp profiles/pmc_traffic.json "$out/pmc_traffic.json"
find "$out" -name "*.csv" -size +2M -delete
grep -E "k_primary|k_paths|k_accumulate|kernel  " "$out/summary.txt"
