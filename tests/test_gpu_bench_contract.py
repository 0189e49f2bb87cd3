"""bench.py prints ONE JSON line with the fields the build driver reads; a short run checks the schema and that the
numbers are self-consistent."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_the_contract_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--reps", "3", "--cpu-seconds", "2"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 100.0 and d["ms_per_step"] > 0.0
    # value = rays of the timed frames / time: the two independent measurements agree
    assert abs(d["value"] - d["rays_per_frame"] / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # the binding ceiling of this kernel is vector-instruction issue; the fraction must be one
    assert r["bound"] == "valu" and r["unit"] == "Tlane-op/s" and abs(r["peak"] - 78.64) < 0.01
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # PMC traffic is quoted only for the build it was measured on
    if r["traffic"] is None:
        assert r["hbm"] is None and "null" in r["note"]
    else:
        assert 0.0 < r["hbm"]["frac"] <= 1.0 and r["hbm"]["peak"] == 8000.0
    t = d["timing"]
    assert t["repetitions"] == 3 and t["region_ms"]["min"] <= t["region_ms"]["median"] <= t["region_ms"]["max"]
    assert abs(d["ms_per_step"] - t["region_ms"]["median"] / d["steps"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "one_core"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0.0
    assert c["one_core"]["cores"] == 1 and c["one_core"]["value"] > 0.0


def _run_bench(extra, launcher=()):
    cmd = [sys.executable, *launcher, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--reps", "3", "--no-cpu-baseline", *extra]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    # the JSON line and nothing else on stdout -- also under the launcher, where RCCL writes its version block to descriptor 1
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_under_torch_distributed_single_rank():
    """The N > 1 code path on one GPU: launched by torch.distributed.run with one rank, bench.py initialises RCCL,
    renders with RT_FLAG_SUM (here in two row bands, the default is one), reduces every band to rank 0 and scales the sum.  The mean image must equal
    the one of the plain run (running mean) up to fp32 rounding, and the exposed reduce time must be small."""
    plain = _run_bench([])
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29655"]
    d = _run_bench(["--gpus", "1", "--reduce-bands", "2"], launcher)
    assert d["n_gpus"] == 1 and d["config"]["frames_total"] == 8
    assert d["rays_per_frame"] == plain["rays_per_frame"]
    for a, b in zip(d["image_mean_rgb"], plain["image_mean_rgb"]):
        assert abs(a - b) <= 2e-5 * abs(b)
    assert d["reduce"]["bands"] == 2 and d["reduce"]["share_of_region"] < 0.05
    one = _run_bench(["--gpus", "1", "--reduce-bands", "1"], launcher)
    assert one["image_mean_rgb"] == d["image_mean_rgb"]      # bands are bit-identical to the whole image


def test_bench_frame_loop_workload():
    """One call of the boundary per frame (the reference's own loop): same rays, same image as the batched call."""
    plain = _run_bench([])
    loop = _run_bench(["--workload", "dragon80k_1080p_frame_loop"])
    assert loop["rays_per_frame"] == plain["rays_per_frame"] and loop["image_mean_rgb"] == plain["image_mean_rgb"]
    assert loop["roofline"]["launches"] == 8
