#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (and FPS) of the render path on the 80K-triangle scene at
1920x1080, 1 spp per frame, full diffuse bounces with Russian roulette, bug-compatible mode.

    python bench.py --gpus N --steps K --warmup W

A step is one frame (one dispatch of the reference: one sample per pixel over the whole image) per
GPU.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the frames are partitioned by
sample index -- rank g renders frameCount g, g+N, ... of the same scene and camera into a private
fp32 SUM buffer -- and one RCCL reduce of the accumulators to rank 0 closes the timed region; that is
weak scaling (fixed work per GPU).  Rays are counted on the device (calls of the closest-hit routine,
raytrace.glsl:245) in a separate untimed pass over exactly the timed frames.

Besides the contract's fields the JSON line carries
  roofline      algorithmic bytes of the dominant kernel / its mean launch duration (HIP events on
                the launch stream, separate profiled pass in which launches run one at a time; in
                the timed region the library overlaps consecutive launches) against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference's GLSL path) timed on this host's cores on a
                bounded sample of the same workload
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
VALU_PEAK_LANE_OPS = 256 * 128 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T fp32 lane-ops/s (157.3 TFLOP/s with FMAs; an FMA counts once here)
WORKLOADS = {
    # name: (obj, camera position, width, height)
    "dragon80k_1080p": ("dragon80k.obj", (0.0, 0.0, -5.5), 1920, 1080),
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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dragon80k_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", default="paths", choices=["paths", "wavefront", "megakernel"])
    ap.add_argument("--fix-invdir", action="store_true", help="non-default: refresh invDir after bounces")
    ap.add_argument("--partition", default="samples", choices=["samples", "tiles"],
                    help="N > 1: 'samples' = each GPU renders its own frames (weak scaling, the default); 'tiles' = every "
                         "GPU renders its horizontal band of every frame (strong scaling)")
    ap.add_argument("--dynamic-update", default="rebuild", choices=["rebuild", "refit"],
                    help="grid9m_dynamic: rebuild the BVH every step (default) or refit the first build's tree")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

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
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    # N = 1 accumulates with the reference's running mean; N > 1 needs plain sums for the reduce
    tiles = args.partition == "tiles" and world > 1
    base_flags = pkg.RT_FLAG_SUM if (world > 1 and not tiles) else 0
    if args.fix_invdir:
        base_flags |= pkg.RT_FLAG_FIX_INVDIR
    if args.pipeline == "megakernel":
        base_flags |= pkg.RT_FLAG_MEGAKERNEL
    if args.pipeline == "wavefront":
        base_flags |= pkg.RT_FLAG_WAVEFRONT
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)

    def render_steps(first_step: int, nsteps: int, flags: int, target, bounces: int = 100):
        """nsteps steps of this rank (frameCount (first_step+k)*world + rank) in ONE call of the boundary:
        the library is free to render the frames of a call concurrently."""
        if tiles:  # every rank renders the same frames, restricted to its band; bands are disjoint, so the
            # closing sum-reduce of the (elsewhere zero) buffers is the gather
            p = pkg.make_params(W, H, ray_bounces=bounces, frame_base=first_step, frames=nsteps, flags=flags, band=(rank, world),
                                interleaved=True)
        else:
            base, stride, n = partition.rank_frames(rank, world, first_step, nsteps)
            p = pkg.make_params(W, H, ray_bounces=bounces, frame_base=base, frames=n, frame_stride=stride, flags=flags)
        ctx.render_device(cam, p, target.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up (untimed) ------------------------------------------------------------------------
    if args.warmup:
        render_steps(0, args.warmup, base_flags, accum)
    if dist is not None:
        warm = accum.clone()
        dist.reduce(warm, dst=0)
        del warm
    accum.zero_()

    # ---- timed region: exactly K steps + the closing reduce -----------------------------------------
    barrier()
    t0 = time.perf_counter()
    render_steps(0, args.steps, base_flags, accum)
    if dist is not None:
        dist.reduce(accum, dst=0)
    if rank == 0 and world > 1 and not tiles:
        accum.mul_(1.0 / float(args.steps * world))  # sum -> mean image
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    image_mean = accum[..., :3].mean(dim=(0, 1)).tolist() if rank == 0 else None

    # ---- untimed: count the rays of exactly those frames ---------------------------------------------
    scratch = torch.zeros_like(accum)

    def count_pass(flags, bounces=100):
        ctx.reset_counters()
        render_steps(0, args.steps, flags | pkg.RT_FLAG_COUNTERS, scratch, bounces)
        return ctx.counters().astype(np.float64)

    counters = count_pass(base_flags)
    rays_local = counters[2]
    if dist is not None:
        tot = torch.tensor(counters, dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        counters_all = tot.cpu().numpy()
    else:
        counters_all = counters
    total_rays = float(counters_all[2])
    frames_total = args.steps if tiles else args.steps * world

    # ---- untimed: per-kernel durations with HIP events on the launch stream ------------------------
    ctx.set_profiling(True)
    render_steps(0, args.steps, base_flags, scratch)
    ms, launches = ctx.kernel_times()
    ctx.set_profiling(False)
    names = pkg.KERNEL_CLASS_NAMES
    dominant = int(np.argmax(ms))
    # algorithmic work of the dominant kernel class alone: for the wavefront extend kernel that is the
    # traversal work of the first RT_WF_BOUNCES(=2) bounces, which a 2-bounce counting pass isolates
    if names[dominant] == "extend":
        dom_counters = count_pass(base_flags, bounces=2)
        dom_counters[5] = 0.0  # the accumulator traffic belongs to the shade kernel
        dom_counters[3] = 0.0  # so do the normal/colour fetches
    else:
        dom_counters = counters.copy()
    dom_bytes_per_launch = algorithmic_bytes(dom_counters) / max(1, int(launches[dominant]))
    dom_ms = float(ms[dominant]) / max(1, int(launches[dominant]))
    achieved_gbs = dom_bytes_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            # PMC bytes per frame of the profiled workload, scaled to this run's launches of the dominant kernel
            with open(tpath) as f:
                tj = json.load(f)
            per_frame = tj.get(names[dominant], {}).get("hbm_bytes_per_frame")
            if per_frame is not None and tj.get("_workload") == args.workload and not args.fix_invdir:
                traffic = round(per_frame * args.steps / max(1, int(launches[dominant])))
        except Exception:
            traffic = None

    if rank == 0:
        value = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s, 80K-tri dragon stand-in 1080p 1spp" if args.workload == "dragon80k_1080p"
            else f"Mrays/s, {args.workload}",
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
                "workload": f"{args.workload}: {obj} ({scene.num_tris} triangles, {scene.num_nodes} BVH nodes, "
                            f"procedural stand-in for the reference's unpublished dragon), {W}x{H}, 1 spp per frame, "
                            f"100 bounces, Russian roulette, {'fix_invdir' if args.fix_invdir else 'bug-compatible invDir'}",
                "pipeline": args.pipeline,
                "frames_total": frames_total,
                "parallelism": (f"tile-partition x{world}" if tiles else f"sample-partition x{world}") if world > 1 else "single GPU",
            },
            "rays_per_frame": round(total_rays / frames_total, 1),
            "work_per_ray": {
                "box_tests": round(counters_all[0] / total_rays, 3),
                "tri_tests": round(counters_all[1] / total_rays, 3),
                "node_pops": round(counters_all[4] / total_rays, 3),
                "hit_fraction": round(counters_all[3] / total_rays, 4),
                "rays_per_path": round(total_rays / counters_all[5], 4),
                "algorithmic_bytes": round(algorithmic_bytes(counters_all) / total_rays, 1),
            },
            "kernels_ms_per_frame": {names[k]: round(float(ms[k]) / args.steps, 4) for k in range(len(names)) if launches[k]},
            "image_mean_rgb": [round(x, 5) for x in image_mean],
            "roofline": {
                "bound": "hbm",
                "kernel": names[dominant],
                "achieved": round(achieved_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "launch_ms": round(dom_ms, 4),
                "algorithmic_bytes_per_launch": round(dom_bytes_per_launch, 0),
                "note": "scene is L2/Infinity-Cache resident, so algorithmic bytes/s may exceed the HBM peak; "
                        "the binding ceiling is VALU issue (see alu)",
                "alu": {
                    "achieved_lane_ops_per_s": round(algorithmic_lane_ops(counters_all) / elapsed, 0),
                    "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                    "frac": round(algorithmic_lane_ops(counters_all) / elapsed / VALU_PEAK_LANE_OPS / world, 4),
                },
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, scene, cam, W, H, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    _ = rays_local
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
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
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
    """The oracle on this host's cores, on whole frames of the same workload until about budget_s
    seconds of wall time are spent (at least one frame)."""
    oracle = entry.load_oracle()
    threads = min(oracle.max_threads(), 16)
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
        if dt >= budget_s or frames >= 64:
            break
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{frames} full {W}x{H} frames (frameCount 0..{frames - 1}) of the same scene and camera, "
                  f"{dt:.1f} s wall, OpenMP over image rows",
        "fps": round(frames / dt, 3),
    }


if __name__ == "__main__":
    sys.exit(main())
