"""Worker of tests/test_cpu_dist.py: one rank of a world-size-N gloo job.  It runs the SHIPPED partition code
(partition.render_partitioned, the function bench.py drives on RCCL) with the CPU oracle injected as the renderer and
gloo as the collective, and rank 0 checks the reduced image against the sequential render of the same frames."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    steps = int(sys.argv[1])
    out_path = sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "samples"
    reduce_bands = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    launch_frames = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    from opengl_raytracer_amd import partition

    scene = pkg.load_scene(os.path.join(ROOT, "fixtures", "teapot.obj"), colour_seed=1)
    cam = pkg.make_camera()
    W, H = 96, 54
    nrm = oracle.normals(scene)
    local = np.zeros((H, W, 4), dtype=np.float32)
    shared = torch.from_numpy(local)      # same memory
    rays_local = [0.0]
    calls = []

    def render(frame_base, stride, frames, band, rows):
        """The oracle standing in for rt_render_device: the same frames, bands and accumulation mode."""
        calls.append((frame_base, stride, frames, band, rows))
        if mode == "tiles":
            assert band == (rank, world) and rows is None
            ranges = partition.band_row_ranges(rank, world, H, True)
            flags = 0
        else:
            assert band is None or partition.band_rows(band[0], band[1], H) == rows
            ranges = [rows]
            flags = pkg.RT_FLAG_SUM
        p = pkg.make_params(W, H, frames=1, flags=flags)
        for k in range(frames):
            for r0, r1 in ranges:
                rays_local[0] += float(oracle.frame_rows(scene, nrm, cam, p, frame_base + k * stride, local, r0, r1, 2)[2])

    def start_reduce(rows):
        return dist.reduce(shared[rows[0]:rows[1]], dst=0, async_op=True)

    reduced = partition.render_partitioned(rank=rank, world=world, steps=steps, height=H, render=render, start_reduce=start_reduce,
                                           finish=lambda work: work.wait(), mode=mode, reduce_bands=reduce_bands,
                                           launch_frames=launch_frames)
    rays = torch.tensor([rays_local[0]], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        covered = sorted(reduced)
        assert covered[0][0] == 0 and covered[-1][1] == H and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        if mode == "tiles":
            want, cnt_all = oracle.render(scene, cam, pkg.make_params(W, H, frames=steps), threads=2)
            same = bool(np.array_equal(local.view(np.uint32), want.view(np.uint32)))
            np.savez(out_path, same=same, rays=float(rays.item()), rays_seq=float(cnt_all[2]), calls=len(calls))
        else:
            image = local * np.float32(1.0 / float(steps * world))
            want, cnt_all = oracle.render(scene, cam, pkg.make_params(W, H, frames=steps * world), threads=2)
            err = float(np.sqrt(((image[..., :3] - want[..., :3]) ** 2).sum() / (want[..., :3] ** 2).sum()))
            # the banded and the unbanded decomposition sum the same numbers in the same order: identical bits
            whole = np.zeros_like(local)
            for g in range(world):
                base, stride, n = partition.rank_frames(g, world, 0, steps)
                part, _ = oracle.render(scene, cam, pkg.make_params(W, H, frame_base=base, frame_stride=stride, frames=n,
                                                                     flags=pkg.RT_FLAG_SUM), np.zeros_like(local), threads=2)
                whole += part
            np.savez(out_path, err=err, rays=float(rays.item()), rays_seq=float(cnt_all[2]), calls=len(calls),
                     same_as_unbanded=bool(np.array_equal(whole.view(np.uint32), local.view(np.uint32))))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
