#!/bin/bash
# Kernel trace of launches that have the chip to themselves: 12 synchronised calls of 16 frames after a warm-up, no counters,
# no overlap.  Mean / min / max of k_primary and k_paths of these are what bench.py's roofline.launch_ms should reproduce.
# usage (GPU box, repo root): standalone_trace.sh <tag> [prof_render.py args]
tag=$1; shift
out=$PWD/gpurun_out/standalone_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 opengl-raytracer_amd/tools/prof_render.py --frames 16 --warmup 32 --standalone 12 "$@" > $out/run.txt 2>&1
cat $out/run.txt | grep -E "synchronised|rror"
python3 opengl-raytracer_amd/tools/rocprof_summary.py $out/trace | grep -E "k_primary|k_paths|k_accumulate|kernel  |=="
echo "launch by launch (start order):"
python3 opengl-raytracer_amd/tools/trace_timeline.py $out/trace -39 39 | grep -E "k_primary|k_paths"
