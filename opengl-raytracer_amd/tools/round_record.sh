#!/bin/bash
# Everything the measurement record of a build needs, in one GPU call: the rocprofv3 kernel trace and the PMC passes of
# the bench command (profile_bench.sh, which also writes profiles/pmc_traffic.json for this build), then the bench lines
# that quote it, the other workloads, the PMC views of the path kernel and smoke().
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
bash opengl-raytracer_amd/tools/bench_all.sh > $out/bench_all.txt 2>&1
cp gpurun_out/bench_all.jsonl $out/bench_all_workloads.jsonl
bash opengl-raytracer_amd/tools/pmc_memory_pipe.sh dragon80k > $out/pmc_memory_pipe.txt 2>&1
bash opengl-raytracer_amd/tools/pmc_ifetch.sh dragon80k > $out/pmc_ifetch.txt 2>&1
bash opengl-raytracer_amd/tools/pmc_insts.sh dragon80k > $out/pmc_insts.txt 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke.txt 2>&1
tail -1 $out/smoke.txt
cat $out/bench_all.txt | tail -14
