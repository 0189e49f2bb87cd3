#!/bin/bash
# Copies what round_record.sh <tag> left under gpurun_out/ into profiles/<tag>_* (and profiles/pmc_traffic.json), removes
# the files of the record it replaces, and prints the figures the documents quote.
# usage (repository root): install_record.sh <tag> [<replaced tag>]
set -e
T=$1; OLD=$2
R=gpurun_out/$T; P=gpurun_out/prof_$T
[ -n "$OLD" ] && git rm -q --ignore-unmatch profiles/${OLD}_bench* profiles/${OLD}_paths_dragon80k_pmc_*
cp $R/pmc_traffic.json profiles/pmc_traffic.json
for f in bench_steps20 bench bench_steps20_no_cull bench_frame_loop_steps20; do cp $R/$f.json profiles/${T}_$f.json; done
cp $R/bench_all_workloads.jsonl profiles/${T}_bench_all_workloads.jsonl
cp $P/summary.txt profiles/${T}_bench_rocprofv3.txt
cp $P/bench_under_trace.json profiles/${T}_bench_under_trace.json
cp $P/trace/vm/*_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
for f in pmc_memory_pipe pmc_ifetch pmc_insts; do cp $R/$f.txt profiles/${T}_paths_dragon80k_$f.txt; done
grep _build profiles/pmc_traffic.json
python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
for f in ("bench_steps20", "bench", "bench_steps20_no_cull", "bench_frame_loop_steps20"):
    d = json.load(open(f"profiles/{t}_{f}.json")); r = d["roofline"]
    print(f, d["build"], d["value"], d["fps"], d["ms_per_step"], d["timing"]["region_ms"], "frac", r["frac"], "launch_ms", r["launch_ms"],
          "exec", r["executed"]["frac"], r["executed"]["frac_without_cull_tests"], "traffic", r["traffic"],
          r["hbm"] and (r["hbm"]["achieved"], r["hbm"]["frac"]), r["l2"] and (r["l2"]["achieved"], r["l2"]["frac"], r["l2"]["hit_rate"]),
          r["vmem_address"] and (r["vmem_address"]["l1_accesses_g_per_s"], r["vmem_address"]["ta_busy_frac"]))
    if "cpu_baseline" in d:
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_core"]["value"])
for l in open(f"profiles/{t}_bench_all_workloads.jsonl"):
    d = json.loads(l)
    print("  ", d["config"]["workload"][:34], d["config"].get("pipeline"), d["value"], d["fps"], d["ms_per_step"])
PY
grep "k_paths" profiles/${T}_bench_kernel_stats.csv | cut -c1-140
grep -h "k_paths" profiles/${T}_paths_dragon80k_pmc_insts.txt | awk '{print $(NF-4), $NF}'
