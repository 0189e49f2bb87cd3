#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (and FPS) of the render path on the 80K-triangle scene at
1920x1080, 1 spp per frame, full diffuse bounces with Russian roulette, bug-compatible mode.

    python bench.py --gpus N --steps K --warmup W

A step is one frame (one dispatch of the reference: one sample per pixel over the whole image) per
GPU.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the frames are partitioned by
sample index -- rank g renders frameCount g, g+N, ... of the same scene and camera into a private
fp32 SUM buffer -- and RCCL reduces of the accumulators to rank 0 close the timed region; that is
weak scaling (fixed work per GPU).  The timed region (exactly K steps between barrier + synchronize)
is repeated --reps times and the line reports the median repetition.  Rays are counted on the device
(calls of the closest-hit routine, raytrace.glsl:245) in a separate untimed pass over exactly the
timed frames.

Besides the contract's fields the JSON line carries
  roofline      the dominant kernel against the ceiling that binds it, vector-instruction issue:
                algorithmic lane-operations per launch / its mean launch duration (HIP events on the
                launch stream, separate pass in which launches run one at a time) against the chip's
                fp32 lane-operations per second.  traffic, hbm and l2 come from rocprofv3 PMC passes
                of the same build (profiles/pmc_traffic.json; null when that file is of another build)
  timing        median / min / max of the repetitions
  reduce        (under torch.distributed) the exposed time of the closing reduces
  cpu_baseline  the CPU oracle (a port of the reference's GLSL path) timed on this host's cores on a
                bounded sample of the same workload, all cores and one core
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth of the 8 XCDs
VALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T fp32 lane-ops/s (157.3 TFLOP/s with FMAs; an FMA counts once here)
WORKLOADS = {
    # name: (obj, camera position, width, height)
    "dragon80k_1080p": ("dragon80k.obj", (0.0, 0.0, -5.5), 1920, 1080),
    # the same workload driven like the reference's loop: one call of the boundary per frame, two buffers
    "dragon80k_1080p_frame_loop": ("dragon80k.obj", (0.0, 0.0, -5.5), 1920, 1080),
    "teapot_1080p": ("teapot.obj", (0.0, 0.0, -2.0), 1920, 1080),
    "dragon80k_4k": ("dragon80k.obj", (0.0, 0.0, -5.5), 3840, 2160),
    # ~10M-triangle stress scene: baked grid of dragon stand-ins and teapots (generated in memory, ~15 s)
    "grid9m_1080p": (None, (0.0, 0.0, -30.0), 1920, 1080),
    # the same mesh animated: every step perturbs all vertices on the device, rebuilds the BVH on the GPU
    # (rt_rebuild_scene_device) and renders one fresh frame; N > 1 splits the frame into horizontal bands
    "grid9m_dynamic": (None, (0.0, 0.0, -30.0), 1920, 1080),
}


def algorithmic_bytes(c) -> float:
    """Bytes the reference's own access pattern moves for the counted work: 32 B per box test (two
    vec3 bounds padded to 16 B), 16 B per node pop (the child/leaf column), 48 B per triangle test
    (three padded vertices), 32 B per hit (normal + colour) and 32 B per path (accumulator read +
    write).  c = [box, tri, rays, hits, pops, paths]."""
    box, tri, rays, hits, pops, paths = [float(x) for x in c]
    return 32.0 * box + 16.0 * pops + 48.0 * tri + 32.0 * hits + 32.0 * paths


def algorithmic_lane_ops(c) -> float:
    """VALU view of the same counters: ~27 lane-ops per box test (6 sub, 6 mul, 12 min/max, 3 compares),
    ~60 per triangle test (two cross products, four dots, a division), ~150 per ray for shading."""
    box, tri, rays, hits, pops, paths = [float(x) for x in c]
    return 27.0 * box + 60.0 * tri + 150.0 * rays


def executed_lane_ops(c, culled: bool) -> float:
    """Lane-ops of the traversal as it ran: the same weights on the executed counts, plus ~22 per box test for the cull
    test that rides on every inner-node visit of the default pipeline (3 reciprocals, 12 sub/mul, 14 min/max, a compare)."""
    box, tri, rays, hits, pops, paths = [float(x) for x in c]
    return (27.0 + (22.0 if culled else 0.0)) * box + 60.0 * tri + 150.0 * rays


def pmc_key(workload: str, fix_invdir: bool) -> str:
    return workload + ("+fix_invdir" if fix_invdir else "")


def load_pmc(key: str, kernels, build: str):
    """PMC figures per frame, summed over the kernel classes `kernels`, from profiles/pmc_traffic.json -- only if its entry
    for this workload was taken with THIS build of the library (every entry carries the build id of the library that was
    profiled).  Returns (dict, None) or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing"
    e = tj.get("workloads", {}).get(key)
    if e is None:
        return None, f"profiles/pmc_traffic.json has no entry for {key}"
    if e.get("_build") != build:
        return None, f"profiles/pmc_traffic.json[{key}] is of build {e.get('_build')}, the library is {build}"
    out = {"_source": e.get("_source"), "_frames_per_launch": e.get("_frames_per_launch")}
    for k in kernels:
        c = e.get(k)
        if not c or c.get("hbm_bytes_per_frame") is None:
            return None, f"profiles/pmc_traffic.json[{key}] has no entry for {k}"
        for f_ in ("hbm_bytes_per_frame", "l2_bytes_per_frame", "l1_accesses_per_frame", "l2_hits_per_frame", "l2_misses_per_frame",
                   "ta_busy_cycles_per_frame", "busy_cu_cycles_per_frame"):
            if c.get(f_) is not None:
                out[f_] = out.get(f_, 0.0) + float(c[f_])
    if out.get("l2_hits_per_frame") is not None and out.get("l2_misses_per_frame") is not None:
        tot = out["l2_hits_per_frame"] + out["l2_misses_per_frame"]
        out["l2_hit_rate"] = round(out["l2_hits_per_frame"] / tot, 4) if tot > 0 else None
    if out.get("ta_busy_cycles_per_frame") is not None and out.get("busy_cu_cycles_per_frame"):
        out["ta_busy_frac"] = round(out["ta_busy_cycles_per_frame"] / out["busy_cu_cycles_per_frame"], 4)
    return out, None


def roofline_dict(valu, hbm, l2, rest):
    """The roofline object of the JSON line: all views, the binding one (largest fraction) on top."""
    views = {"valu": valu}
    if hbm is not None:
        views["hbm"] = hbm
    if l2 is not None:
        views["l2"] = l2
    bound = max(views, key=lambda k: views[k]["frac"])
    top = views[bound]
    out = {"bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"]}
    out.update(rest)
    out.update({"valu": valu, "hbm": hbm, "l2": l2})
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=7,
                    help="the timed region (exactly --steps steps, barrier + synchronize on both sides) is repeated this many "
                         "times; the line reports the median repetition, min and max next to it")
    ap.add_argument("--workload", default="dragon80k_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", default="paths", choices=["paths", "wavefront", "megakernel"])
    ap.add_argument("--fix-invdir", action="store_true", help="non-default: refresh invDir after bounces")
    ap.add_argument("--no-cull", action="store_true",
                    help="non-default: RT_FLAG_NO_CULL, traverse exactly the nodes the reference traverses")
    ap.add_argument("--reuse-primary", action="store_true",
                    help="non-default, a separate workload: RT_FLAG_REUSE_PRIMARY, the primary hits are traced once and every frame "
                         "starts at the first bounce; Mrays/s then counts the rays actually traced, compare by fps")
    ap.add_argument("--partition", default="samples", choices=["samples", "tiles"],
                    help="N > 1: 'samples' = each GPU renders its own frames (weak scaling, the default); 'tiles' = every "
                         "GPU renders its horizontal band of every frame (strong scaling)")
    ap.add_argument("--reduce-bands", type=int, default=1,
                    help="sample partition under torch.distributed: the image is finished in this many row bands, the reduce "
                         "of one band travelling while the next renders (1, the default = one reduce of the whole image at the end; "
                         "2 costs 1.6 %% more launches on one rank and pays off only if the collective finds free CUs next to the "
                         "persistent kernel, which no run on several GPUs has shown yet)")
    ap.add_argument("--dynamic-update", default="rebuild", choices=["rebuild", "refit"],
                    help="grid9m_dynamic: rebuild the BVH every step (default) or refit the first build's tree")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()
    args.reps = max(1, args.reps)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            return 2
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run, also with a single rank
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    pkg = entry.load_package()
    from opengl_raytracer_amd import partition
    obj, cam_pos, W, H = WORKLOADS[args.workload]
    if args.workload == "grid9m_dynamic":
        return run_dynamic(args, pkg, partition, torch, dist, dev, rank, world, local_rank, cam_pos, W, H)
    frame_loop = args.workload.endswith("_frame_loop")
    if obj is None:
        sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "tools"))
        import gen_scenes

        fx = os.path.join(ROOT, "fixtures")
        v, f = gen_scenes.grid([os.path.join(fx, "dragon80k.obj")] * 3 + [os.path.join(fx, "teapot.obj")], (6, 5, 5), 8.0, 7)
        scene = pkg.build_scene(v, f, colour_seed=1)
        obj = "generated grid 6x5x5 of dragon80k/teapot"
    else:
        scene = pkg.load_scene(os.path.join(ROOT, "fixtures", obj), colour_seed=1)
    cam = pkg.make_camera(pos=cam_pos)
    ctx = pkg.RenderContext(local_rank)
    ctx.upload_scene(scene)
    # One stream carries everything: the library's accumulate kernels, torch's fills and scales, and the point the
    # collectives are ordered against -- the context's own stream, handed to torch as an external stream.  (Creating a
    # torch.cuda.Stream instead makes torch allocate its pool of 32 streams, which then share the process's 4 hardware
    # queues with the library's internal streams.)
    stream = torch.cuda.ExternalStream(ctx.get_stream(), device=dev)
    torch.cuda.set_stream(stream)
    build_id = pkg.build_id()

    # without torch.distributed the frames accumulate with the reference's running mean; under it (also with a single
    # rank) every rank renders plain sums that one reduce to rank 0 adds up
    tiles = args.partition == "tiles" and world > 1
    sum_mode = dist is not None and not tiles
    mode = "tiles" if tiles else "samples"
    reduce_bands = args.reduce_bands if (sum_mode and not frame_loop) else 1
    base_flags = pkg.RT_FLAG_SUM if sum_mode else 0
    if args.fix_invdir:
        base_flags |= pkg.RT_FLAG_FIX_INVDIR
    if args.no_cull:
        base_flags |= pkg.RT_FLAG_NO_CULL
    if args.reuse_primary:
        base_flags |= pkg.RT_FLAG_REUSE_PRIMARY
    if args.pipeline == "megakernel":
        base_flags |= pkg.RT_FLAG_MEGAKERNEL
        reduce_bands = 1   # bands are a feature of the default pipeline
    if args.pipeline == "wavefront":
        base_flags |= pkg.RT_FLAG_WAVEFRONT
        reduce_bands = 1
    # frame_loop drives rt_raytrace_frame with two buffers, as the reference's loop does with its two textures
    bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2 if frame_loop else 1)]
    scratch = [torch.zeros_like(b) for b in bufs]
    last = {"buf": bufs[0]}

    def make_render(target, flags, bounces=100):
        def render(frame_base, stride, frames, band, rows):
            if frame_loop:  # one call of the boundary per frame (main.cpp:193), ping-pong (main.cpp:189-191,199)
                p = pkg.make_params(W, H, ray_bounces=bounces, flags=flags)
                for k in range(frames):
                    ctx.raytrace_frame(cam, p, frame_base + k * stride, target[(k + 1) & 1].data_ptr(), target[k & 1].data_ptr())
                last["buf"] = target[(frames - 1) & 1]
                return
            # all frames of the rank in ONE call: the library is free to render the frames of a call concurrently
            p = pkg.make_params(W, H, ray_bounces=bounces, frame_base=frame_base, frames=frames, frame_stride=stride, flags=flags,
                                band=band, interleaved=tiles)
            ctx.render_device(cam, p, target[0].data_ptr())
            last["buf"] = target[0]
        return render

    marks = []

    def start_reduce(rows):
        if dist is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)   # the rows' last accumulate has been enqueued: what follows until `done` is exposed reduce time
        marks.append(ev)
        return dist.reduce(last["buf"][rows[0]:rows[1]], dst=0, async_op=True)

    def finish(work):
        if work is not None:
            work.wait()     # orders the stream behind the collective; does not block the host

    def no_reduce(rows):
        return None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def one_pass(target, flags, reduce=True, bounces=100, steps=None):
        partition.render_partitioned(rank=rank, world=world, steps=args.steps if steps is None else steps, height=H,
                                     render=make_render(target, flags, bounces),
                                     start_reduce=start_reduce if reduce else no_reduce, finish=finish, mode=mode,
                                     reduce_bands=reduce_bands)

    # ---- warm-up (untimed): W steps, and one pass through the collective --------------------------------
    if args.warmup:
        one_pass(bufs, base_flags, steps=args.warmup)
    elif dist is not None:
        finish(start_reduce((0, H)))
    marks.clear()

    # ---- timed region: exactly K steps + the closing reduce, repeated --reps times -----------------------
    times, reduce_ms = [], []
    for _ in range(args.reps):
        barrier()
        t0 = time.perf_counter()
        if sum_mode:
            for b in bufs:
                b.zero_()
        one_pass(bufs, base_flags)
        done = torch.cuda.Event(enable_timing=True)
        done.record(stream)
        if rank == 0 and sum_mode:
            last["buf"].mul_(1.0 / float(args.steps * world))  # sum -> mean image
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            reduce_ms.append(float(marks[-1].elapsed_time(done)))
            marks.clear()
        times.append(elapsed)
    order = sorted(times)
    elapsed = order[len(order) // 2] if len(order) % 2 else 0.5 * (order[len(order) // 2 - 1] + order[len(order) // 2])
    image_mean = last["buf"][..., :3].mean(dim=(0, 1)).tolist() if rank == 0 else None

    # ---- untimed: count the rays of exactly those frames ---------------------------------------------
    def count_pass(flags, bounces=100):
        ctx.reset_counters()
        one_pass(scratch, flags | pkg.RT_FLAG_COUNTERS, reduce=False, bounces=bounces)
        return ctx.counters().astype(np.float64)

    # `counters`: the reference's traversal (its numBoxTests / numTriangleTests: the definition of the algorithmic work);
    # `executed`: the traversal as the timed region ran it (the default pipeline skips subtrees a bounce ray cannot reach)
    def all_ranks(c):
        if dist is None:
            return c
        tot = torch.tensor(c, dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        return tot.cpu().numpy()

    counters = count_pass(base_flags | pkg.RT_FLAG_NO_CULL)
    executed = counters if args.no_cull else count_pass(base_flags)
    culled = bool(executed[0] < counters[0])
    counters_all, executed_all = all_ranks(counters), all_ranks(executed)
    total_rays = float(counters_all[2])
    frames_total = args.steps if tiles else args.steps * world

    # ---- untimed: per-kernel durations with HIP events on the launch stream ------------------------
    # (profiled launches run one at a time on the context's stream; in the timed region consecutive launches overlap)
    ctx.set_profiling(True)
    one_pass(scratch, base_flags, reduce=False)
    ms, launches = ctx.kernel_times()
    ctx.set_profiling(False)
    names = pkg.KERNEL_CLASS_NAMES
    dominant = int(np.argmax(ms))
    # The default pipeline may trace a launch's primary rays tile by tile in k_primary ahead of the path kernel: the two
    # kernels then share the traversal work the counters describe and are rated together, durations added.
    i_paths, i_primary = names.index("paths"), names.index("primary")
    pair = names[dominant] in ("paths", "primary") and launches[i_primary] > 0
    if pair:
        dominant = i_paths
    # algorithmic work of the dominant kernel class alone: for the wavefront extend kernel that is the
    # traversal work of the first RT_WF_BOUNCES(=2) bounces, which a 2-bounce counting pass isolates
    if names[dominant] == "extend":
        dom_counters = count_pass(base_flags | pkg.RT_FLAG_NO_CULL, bounces=2)
        dom_counters[5] = 0.0  # the accumulator traffic belongs to the shade kernel
        dom_counters[3] = 0.0  # so do the normal/colour fetches
    else:
        dom_counters = counters.copy()
    dom_executed = executed if names[dominant] == "paths" else dom_counters
    n_launch = max(1, int(launches[dominant]))
    dom_ms = (float(ms[dominant]) + (float(ms[i_primary]) if pair else 0.0)) / n_launch
    dom_kernels = ["primary", "paths"] if pair else [names[dominant]]
    dom_s = dom_ms * 1e-3 if dom_ms > 0 else float("inf")
    dom_bytes_per_launch = algorithmic_bytes(dom_counters) / n_launch
    dom_ops_per_launch = algorithmic_lane_ops(dom_counters) / n_launch
    frames_per_launch = args.steps / n_launch  # whole-image frames (a launch over a band counts by its share of the rows)

    if args.reuse_primary or args.no_cull or args.pipeline != "paths":
        pmc, pmc_why = None, "no PMC record is kept for this variant"
    else:
        pmc, pmc_why = load_pmc(pmc_key(args.workload, args.fix_invdir), dom_kernels, build_id)
    traffic = hbm = l2 = vmem = None
    if pmc is not None:
        traffic = round(pmc["hbm_bytes_per_frame"] * frames_per_launch)
        hbm = {"achieved": round(traffic / dom_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(traffic / dom_s / 1e9 / HBM_PEAK_GBS, 4)}
        if pmc.get("l1_accesses_per_frame") is not None:
            acc = pmc["l1_accesses_per_frame"] * frames_per_launch
            vmem = {"l1_accesses_g_per_s": round(acc / dom_s / 1e9, 1), "ta_busy_frac": pmc.get("ta_busy_frac"),
                    "note": "every lane reads its own 64-byte record as four 16-byte quads, one L1 access each; ta_busy_frac = "
                            "TA_TA_BUSY_sum / SQ_BUSY_CU_CYCLES: the busiest unit of the kernel, but occupied rather than saturated "
                            "(with no records staged in LDS the same unit serves 58 % more accesses per second)"}
        if pmc.get("l2_bytes_per_frame") is not None:
            l2b = pmc["l2_bytes_per_frame"] * frames_per_launch
            l2 = {"achieved": round(l2b / dom_s / 1e9, 1), "peak": L2_PEAK_GBS, "unit": "GB/s",
                  "frac": round(l2b / dom_s / 1e9 / L2_PEAK_GBS, 4), "bytes_per_launch": round(l2b),
                  "hit_rate": pmc.get("l2_hit_rate")}

    if rank == 0:
        value = total_rays / elapsed / 1e6
        tlane = dom_ops_per_launch / dom_s / 1e12
        out = {
            "metric": ("Mrays/s, 80K-tri dragon stand-in 1080p 1spp" if args.workload.startswith("dragon80k_1080p")
                       else f"Mrays/s, {args.workload}") + (" (primary hits reused: rays actually traced)" if args.reuse_primary else ""),
            "value": round(value, 2),
            "unit": "Mrays/s",
            "fps": round(frames_total / elapsed, 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if tiles else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {obj} ({scene.num_tris} triangles, {scene.num_nodes} BVH nodes"
                            + (", procedural stand-in for the reference's unpublished dragon" if str(obj).startswith("dragon80k") else "")
                            + f"), {W}x{H}, 1 spp per frame, "
                            f"100 bounces, Russian roulette, {'fix_invdir' if args.fix_invdir else 'bug-compatible invDir'}"
                            + (", RT_FLAG_NO_CULL (the reference's literal traversal)" if args.no_cull else "")
                            + (", one call of the boundary per frame with ping-pong buffers (the reference's loop, main.cpp:183-203)"
                               if frame_loop else "")
                            + (", RT_FLAG_REUSE_PRIMARY: primary hits traced once per camera, not per frame (NOT the reference's "
                               "work per frame; separate workload)" if args.reuse_primary else ""),
                "pipeline": args.pipeline,
                "frames_total": frames_total,
                "parallelism": (f"tile-partition x{world}" if tiles else f"sample-partition x{world}") if world > 1 else "single GPU",
            },
            "timing": {
                "repetitions": args.reps,
                "region_ms": {"median": round(elapsed * 1e3, 4), "min": round(order[0] * 1e3, 4), "max": round(order[-1] * 1e3, 4)},
                "value_best": round(total_rays / order[0] / 1e6, 2),
                "note": "every repetition times exactly `steps` steps between barrier + synchronize; value and ms_per_step are "
                        "the median repetition's",
            },
            "build": build_id,
            "rays_per_frame": round(total_rays / frames_total, 1),
            "work_per_ray": {
                "box_tests": round(counters_all[0] / total_rays, 3),
                "tri_tests": round(counters_all[1] / total_rays, 3),
                "node_pops": round(counters_all[4] / total_rays, 3),
                "hit_fraction": round(counters_all[3] / total_rays, 4),
                "rays_per_path": round(total_rays / counters_all[5], 4),
                "algorithmic_bytes": round(algorithmic_bytes(counters_all) / total_rays, 1),
                "algorithmic_lane_ops": round(algorithmic_lane_ops(counters_all) / total_rays, 1),
                "note": "the reference's own traversal of these rays (RT_FLAG_NO_CULL counting pass): the definition of the "
                        "algorithmic work",
            },
            "executed_per_ray": {
                "box_tests": round(executed_all[0] / total_rays, 3),
                "tri_tests": round(executed_all[1] / total_rays, 3),
                "node_pops": round(executed_all[4] / total_rays, 3),
                "lane_ops": round(executed_lane_ops(executed_all, culled) / total_rays, 1),
                "culled": culled,
                "note": "the traversal as timed: same rays, same hits, same image; subtrees that a bounce ray's true line "
                        "cannot reach are skipped (the reference walks them because its slab tests use the primary ray's "
                        "stale invDir); lane_ops counts the cull test too",
            },
            "kernels_ms_per_frame": {names[k]: round(float(ms[k]) / args.steps, 4) for k in range(len(names)) if launches[k]},
            "image_mean_rgb": [round(x, 5) for x in image_mean],
            # Three candidate ceilings, all reported; `bound` names the one with the largest fraction and achieved / peak /
            # unit / frac are that view's.  valu: the arithmetic of the REFERENCE's traversal of these rays (27 lane-ops per
            # box test, 60 per triangle test, 150 per ray, on the reference's own counts) per second against the chip's fp32
            # lane-operation rate; frac_executed prices what the kernels ran instead (fewer box tests: subtrees a bounce ray
            # cannot reach are skipped).  hbm / l2: PMC traffic of this build against the HBM / aggregate L2 peak.
            "roofline": roofline_dict(
                valu={"achieved": round(tlane, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2), "unit": "Tlane-op/s",
                      "frac": round(tlane * 1e12 / VALU_PEAK_LANE_OPS, 4)},
                hbm=hbm, l2=l2,
                rest={
                    "kernel": "+".join(dom_kernels),
                    "frac_executed": round(algorithmic_lane_ops(dom_executed) / n_launch / dom_s / VALU_PEAK_LANE_OPS, 4),
                    "traffic": traffic,
                    "launch_ms": round(dom_ms, 4),
                    "launches": n_launch,
                    "frames_per_launch": round(frames_per_launch, 3),
                    "lane_ops_per_launch": round(dom_ops_per_launch, 0),
                    "frac_end_to_end": round(algorithmic_lane_ops(counters_all) / elapsed / VALU_PEAK_LANE_OPS / world, 4),
                    "executed": {"achieved": round(executed_lane_ops(dom_executed, culled) / n_launch / dom_s / 1e12, 3),
                                 "frac": round(executed_lane_ops(dom_executed, culled) / n_launch / dom_s / VALU_PEAK_LANE_OPS, 4),
                                 "frac_without_cull_tests": round(algorithmic_lane_ops(dom_executed) / n_launch / dom_s / VALU_PEAK_LANE_OPS, 4),
                                 "note": "frac prices the REFERENCE's traversal of these rays (its box and triangle test counts) per "
                                         "second; these are the lane-ops the kernels actually ran, with and without the arithmetic of "
                                         "the cull test itself (frac_without_cull_tests = roofline.frac_executed)"},
                    "vmem_address": vmem,
                    "algorithmic_bytes": {"per_launch": round(dom_bytes_per_launch, 0), "gb_per_s": round(dom_bytes_per_launch / dom_s / 1e9, 1),
                                          "note": "bytes the reference's access pattern would move; served by L1/L2/LDS here, not a "
                                                  "fraction of any peak"},
                    "note": ("launch_ms = " + " + ".join("k_" + k for k in dom_kernels) + " of one launch, each timed alone with HIP events on "
                             "the launch stream; " if pair else "")
                            + (("traffic = PMC (2*FETCH_SIZE + WRITE_SIZE) KiB per frame of this build (" + str(pmc.get("_source"))
                                + ") x frames per launch of this run") if pmc is not None else f"traffic: null -- {pmc_why}"),
                }),
        }
        if dist is not None:
            rs = sorted(reduce_ms)
            out["reduce"] = {"bands": reduce_bands, "exposed_ms_median": round(rs[len(rs) // 2], 4),
                             "share_of_region": round(rs[len(rs) // 2] / (elapsed * 1e3), 4),
                             "note": "time from the last band's final accumulate to the completion of all reduces (rank 0)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, scene, cam, W, H, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


def run_dynamic(args, pkg, partition, torch, dist, dev, rank, world, local_rank, cam_pos, W, H) -> int:
    """Animated mesh: a step = perturb every vertex (device-resident, a torch elementwise op standing in for the
    caller's animation) + rebuild the binned-SAH BVH on the GPU + render one frame of the new scene.  With N > 1 every
    rank rebuilds the whole tree (the scene is replicated) and renders only its horizontal band; one sum-reduce of the
    otherwise zero frame buffers per step gathers the frame on rank 0 (strong scaling of the render part)."""
    sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "tools"))
    import gen_scenes

    fx = os.path.join(ROOT, "fixtures")
    v, f = gen_scenes.grid([os.path.join(fx, "dragon80k.obj")] * 3 + [os.path.join(fx, "teapot.obj")], (6, 5, 5), 8.0, 7)
    v, f = v.astype(np.float32), f.astype(np.uint32)
    col = pkg.make_colours(1, f.shape[0])
    cam = pkg.make_camera(pos=cam_pos)
    ctx = pkg.RenderContext(local_rank)
    stream = torch.cuda.ExternalStream(ctx.get_stream(), device=dev)
    torch.cuda.set_stream(stream)
    ctx.build_scene_device(v, f, col, download=False)
    d_rest = torch.from_numpy(v).to(dev)
    d_v = torch.empty_like(d_rest)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    band = (rank, world) if world > 1 else None
    flags = pkg.RT_FLAG_FIX_INVDIR if args.fix_invdir else 0

    def step(k: int, extra_flags: int = 0):
        torch.sin(d_rest * 3.0 + 0.37 * k, out=d_v)
        d_v.mul_(0.01).add_(d_rest)
        if args.dynamic_update == "refit":
            _, build_ms = ctx.refit_scene_device(d_v.data_ptr())
        else:
            _, build_ms = ctx.rebuild_scene_device(d_v.data_ptr())
        ctx.render_device(cam, pkg.make_params(W, H, frames=1, flags=flags | extra_flags, band=band, interleaved=True), frame.data_ptr())
        if dist is not None:
            dist.reduce(frame, dst=0)
            if rank == 0 and world > 1:  # keep only this rank's band for the next step's reduce
                keep = torch.zeros(H, dtype=torch.bool, device=dev)
                for r0, r1 in partition.band_row_ranges(rank, world, H, True):
                    keep[r0:r1] = True
                frame[~keep] = 0.0
        return build_ms

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    build_ms = [step(k) for k in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # untimed: rays of the same steps
    ctx.reset_counters()
    for k in range(args.steps):
        step(k, pkg.RT_FLAG_COUNTERS)
    counters = ctx.counters().astype(np.float64)
    if dist is not None:
        tot = torch.tensor(counters, dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        counters = tot.cpu().numpy()
    # untimed: render-only time of one step's scene
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(8):
        ctx.render_device(cam, pkg.make_params(W, H, frames=1, flags=flags, band=band, interleaved=True), frame.data_ptr())
    torch.cuda.synchronize(dev)
    render_ms = (time.perf_counter() - t1) / 8 * 1e3
    if rank == 0:
        total_rays = float(counters[2])
        out = {
            "metric": "Mrays/s, grid9m_dynamic (per-frame GPU BVH rebuild + render)",
            "value": round(total_rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "fps": round(args.steps / elapsed, 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"grid9m_dynamic: generated grid 6x5x5 of dragon80k/teapot ({f.shape[0]} triangles), every vertex "
                            f"displaced per step, binned-SAH {args.dynamic_update} on the GPU, {W}x{H}, 1 spp, 100 bounces, Russian roulette, "
                            f"{'fix_invdir' if args.fix_invdir else 'bug-compatible invDir'}",
                "parallelism": f"replicated rebuild + tile-partition x{world}" if world > 1 else "single GPU",
            },
            "build_ms_per_step": round(float(np.mean(build_ms)), 3),
            "render_ms_per_step": round(render_ms, 3),
            "build_mtris_per_s": round(f.shape[0] / (float(np.mean(build_ms)) * 1e-3) / 1e6, 1),
            "rays_per_frame": round(total_rays / args.steps, 1),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


def cpu_baseline(pkg, scene, cam, W, H, budget_s):
    """The oracle on this host: two thirds of budget_s on whole frames with all the cores this process may use, one
    third on one core (stripes of 8 rows spread over frame 0)."""
    oracle = entry.load_oracle()
    # OpenMP's default team: OMP_NUM_THREADS when the host sets it (a GPU box gives a job a share of its cores), else
    # every core the process may run on
    threads = oracle.max_threads()
    nrm = oracle.normals(scene)
    img = np.zeros((H, W, 4), dtype=np.float32)
    p = pkg.make_params(W, H, frames=1)
    rays = 0.0
    frames = 0
    t0 = time.perf_counter()
    while True:
        c = oracle.frame_rows(scene, nrm, cam, p, frames, img, 0, H, threads)
        rays += float(c[2])
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s * 2.0 / 3.0 or frames >= 64:
            break
    # one core: 8-row stripes of frame 0 in an order that covers the image evenly at every stopping point
    stripes = [r for off in (0, 32, 16, 48, 8, 40, 24, 56) for r in range(off, H, 64)]
    rays1, rows1 = 0.0, 0
    t1 = time.perf_counter()
    for r in stripes:
        rays1 += float(oracle.frame_rows(scene, nrm, cam, p, 0, img, r, min(r + 8, H), 1)[2])
        rows1 += min(r + 8, H) - r
        dt1 = time.perf_counter() - t1
        if dt1 >= budget_s / 3.0:
            break
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{frames} full {W}x{H} frames (frameCount 0..{frames - 1}) of the same scene and camera, "
                  f"{dt:.1f} s wall, OpenMP over image rows, {threads} threads (the OpenMP default team of this process; "
                  f"the host shows {os.cpu_count()} CPUs)",
        "fps": round(frames / dt, 3),
        "one_core": {"value": round(rays1 / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                     "sample": f"{rows1} of {H} rows of frame 0 in 8-row stripes spread over the image, {dt1:.1f} s wall"},
    }


def _only_the_json_line_on_stdout():
    """Libraries loaded later write banners to file descriptor 1 (RCCL prints its version block there at start-up): from here on
    descriptor 1 is the terminal's stderr, and sys.stdout -- what main() prints its one JSON line to -- keeps the real one."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(real, "w", buffering=1)


if __name__ == "__main__":
    _only_the_json_line_on_stdout()
    sys.exit(main())
