#!/bin/bash
# Every bench.py workload once (single GPU), one JSON line each, into gpurun_out/bench_all.jsonl.
# Run from the repository root on the GPU box.
out=gpurun_out/bench_all.jsonl
: > $out
run() { echo "== bench.py $*" | tee -a gpurun_out/bench_all.progress >&2; timeout -k 10 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null >> $out || echo "{\"failed\": \"$*\"}" >> $out; }
run --steps 64 --warmup 8
run --workload teapot_1080p --steps 64 --warmup 8
run --workload dragon80k_4k --steps 256 --warmup 8
run --workload dragon80k_1080p --fix-invdir --steps 64 --warmup 8
run --workload dragon80k_1080p_frame_loop --steps 64 --warmup 8
run --workload dragon80k_1080p --reuse-primary --steps 64 --warmup 8
run --workload dragon80k_4k --reuse-primary --steps 256 --warmup 8
run --workload dragon80k_1080p --pipeline megakernel --steps 32 --warmup 4
run --workload dragon80k_1080p --pipeline wavefront --steps 32 --warmup 4
run --workload grid9m_1080p --steps 16 --warmup 4
run --workload grid9m_1080p --fix-invdir --steps 16 --warmup 4
run --workload grid9m_dynamic --steps 20 --warmup 3
run --workload grid9m_dynamic --dynamic-update refit --steps 20 --warmup 3
python3 - <<'PY'
import json
for ln in open("gpurun_out/bench_all.jsonl"):
    d = json.loads(ln)
    if "failed" in d:
        print("FAILED", d["failed"]); continue
    print(f"{d['config']['workload'][:28]:28s} {d['config'].get('pipeline',''):10s} {d['value']:10.1f} Mrays/s {d['fps']:9.1f} fps {d['ms_per_step']:8.3f} ms/step")
PY
