#!/bin/bash
# The measurement record of a build, part 1 (one GPU call): the rocprofv3 kernel trace and the PMC passes of the bench command
# (profile_bench.sh, which also writes the dragon80k_1080p entry of profiles/pmc_traffic.json for this build), the bench lines
# that quote it, the other workloads, launches timed alone, the PMC views of the pipeline's kernels (packets on / off) and
# smoke().  Part 2, round_record_grid.sh, adds the 9.27 M-triangle scene.
# usage (GPU box, repo root): round_record.sh <tag>
set -e
tag=$1
out=gpurun_out/$tag
mkdir -p $out
bash opengl-raytracer_amd/tools/profile_bench.sh $tag 16 --steps 64 --warmup 16 --reps 3 > $out/profile.log 2>&1
cp profiles/pmc_traffic.json $out/pmc_traffic.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2>> $out/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-cull > $out/bench_steps20_no_cull.json 2>> $out/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --workload dragon80k_1080p_frame_loop --steps 20 --warmup 5 > $out/bench_frame_loop_steps20.json 2>> $out/bench.err
bash opengl-raytracer_amd/tools/standalone_trace.sh $tag > $out/paths_standalone_trace.txt 2>&1
bash opengl-raytracer_amd/tools/pmc_primary.sh ${tag}_packets dragon80k > $out/pmc_pipeline_packets.txt 2>&1
RT_TUNE_PACKETS=0 bash opengl-raytracer_amd/tools/pmc_primary.sh ${tag}_lanes dragon80k > $out/pmc_pipeline_lane_by_lane.txt 2>&1
bash opengl-raytracer_amd/tools/bench_all.sh > $out/bench_all.txt 2>&1
cp gpurun_out/bench_all.jsonl $out/bench_all_workloads.jsonl
timeout -k 10 300 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke.txt 2>&1
tail -1 $out/smoke.txt
cat $out/bench_all.txt | tail -14
