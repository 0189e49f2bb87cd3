#!/bin/bash
# Copies what round_record.sh <tag> and round_record_grid.sh <tag> left under gpurun_out/ into profiles/<tag>_* (and
# profiles/pmc_traffic.json), removes the files of the record it replaces, and prints the figures the documents quote.
# usage (repository root): install_record.sh <tag> [<replaced tag>]
set -e
T=$1; OLD=$2
R=gpurun_out/$T; P=gpurun_out/prof_$T
[ -n "$OLD" ] && git rm -q --ignore-unmatch "profiles/${OLD}_*"
# each profile_bench.sh run leaves its copy of pmc_traffic.json with its own workload's entry renewed: take every entry from the
# run that measured it (the two parts of a record start from the same tree, so neither sees the other's entry)
python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
dst = json.load(open("profiles/pmc_traffic.json"))
for suffix, key in (("", "dragon80k_1080p"), ("_grid9m", "grid9m_1080p"), ("_grid9m_fix", "grid9m_1080p+fix_invdir")):
    src = json.load(open(f"gpurun_out/prof_{t}{suffix}/pmc_traffic.json"))
    dst["workloads"][key] = src["workloads"][key]
json.dump(dst, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
for f in bench_steps20 bench bench_steps20_no_cull bench_frame_loop_steps20 bench_grid9m bench_grid9m_fix; do cp $R/$f.json profiles/${T}_$f.json; done
cp $R/bench_all_workloads.jsonl profiles/${T}_bench_all_workloads.jsonl
cp $R/smoke.txt profiles/${T}_smoke.txt
for w in "" _grid9m _grid9m_fix; do
  cp gpurun_out/prof_$T$w/summary.txt profiles/${T}_bench${w}_rocprofv3.txt
  cp gpurun_out/prof_$T$w/bench_under_trace.json profiles/${T}_bench${w}_under_trace.json
  cp gpurun_out/prof_$T$w/trace/vm/*_kernel_stats.csv profiles/${T}_bench${w}_kernel_stats.csv
done
cp $R/paths_standalone_trace.txt profiles/${T}_dragon80k_paths_standalone_trace.txt
cp $R/paths_standalone_trace_grid9m.txt profiles/${T}_grid9m_paths_standalone_trace.txt
cp $R/pmc_pipeline_packets.txt profiles/${T}_pipeline_dragon80k_pmc_packets.txt
cp $R/pmc_pipeline_lane_by_lane.txt profiles/${T}_pipeline_dragon80k_pmc_lane_by_lane.txt
sed -i "s#/tmp/code/[^ ]*/repo/##" profiles/${T}_*.txt
grep _build profiles/pmc_traffic.json
python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
for f in ("bench_steps20", "bench", "bench_steps20_no_cull", "bench_frame_loop_steps20", "bench_grid9m", "bench_grid9m_fix"):
    d = json.load(open(f"profiles/{t}_{f}.json")); r = d["roofline"]
    print(f, d["build"], d["value"], d["fps"], d["ms_per_step"], d["timing"]["region_ms"], "bound", r["bound"], "frac", r["frac"],
          "frac_executed", r.get("frac_executed"), "launch_ms", r["launch_ms"], "kernel", r["kernel"],
          "exec", r["executed"]["frac"], "traffic", r["traffic"],
          r["valu"]["frac"], r["hbm"] and (r["hbm"]["achieved"], r["hbm"]["frac"]), r["l2"] and (r["l2"]["achieved"], r["l2"]["frac"], r["l2"]["hit_rate"]),
          r["vmem_address"] and (r["vmem_address"]["l1_accesses_g_per_s"], r["vmem_address"]["ta_busy_frac"]), d["kernels_ms_per_frame"])
    if "cpu_baseline" in d:
        print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_core"]["value"])
for l in open(f"profiles/{t}_bench_all_workloads.jsonl"):
    d = json.loads(l)
    print("  ", d["config"]["workload"][:34], d["config"].get("pipeline"), d["value"], d["fps"], d["ms_per_step"])
PY
head -6 profiles/${T}_dragon80k_paths_standalone_trace.txt profiles/${T}_grid9m_paths_standalone_trace.txt
