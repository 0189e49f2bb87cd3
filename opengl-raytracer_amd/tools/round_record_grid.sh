#!/bin/bash
# The measurement record of a build, part 2: the 9.27 M-triangle scene (bench.py --workload grid9m_1080p), bug-compatible and
# with --fix-invdir: kernel trace + the PMC passes of profile_bench.sh (entries grid9m_1080p and grid9m_1080p+fix_invdir of
# profiles/pmc_traffic.json), then the two bench lines that quote them, and launches timed alone.
# usage (GPU box, repo root): round_record_grid.sh <tag>      (after round_record.sh <tag>: pass its pmc_traffic.json on)
set -e
tag=$1
out=gpurun_out/$tag
mkdir -p $out
[ -f $out/pmc_traffic.json ] && cp $out/pmc_traffic.json profiles/pmc_traffic.json
bash opengl-raytracer_amd/tools/profile_bench.sh ${tag}_grid9m 16 --workload grid9m_1080p --steps 32 --warmup 16 --reps 2 > $out/profile_grid9m.log 2>&1
bash opengl-raytracer_amd/tools/profile_bench.sh ${tag}_grid9m_fix 16 --workload grid9m_1080p --fix-invdir --steps 32 --warmup 16 --reps 2 > $out/profile_grid9m_fix.log 2>&1
cp profiles/pmc_traffic.json $out/pmc_traffic.json
timeout -k 10 300 python bench.py --no-cpu-baseline --workload grid9m_1080p --steps 32 --warmup 16 > $out/bench_grid9m.json 2> $out/bench_grid.err
timeout -k 10 300 python bench.py --no-cpu-baseline --workload grid9m_1080p --fix-invdir --steps 32 --warmup 16 > $out/bench_grid9m_fix.json 2>> $out/bench_grid.err
bash opengl-raytracer_amd/tools/standalone_trace.sh ${tag}_grid9m --scene grid > $out/paths_standalone_trace_grid9m.txt 2>&1
python3 - $out <<'PY'
import json, sys
for f in ("bench_grid9m", "bench_grid9m_fix"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], "bound", r["bound"], "frac", r["frac"], "valu", r["valu"]["frac"], "hbm", r["hbm"], "l2", r["l2"])
PY
