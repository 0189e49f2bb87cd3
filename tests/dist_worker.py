"""Worker of tests/test_cpu_dist.py: one rank of a world-size-N gloo job rendering its share of the frames
with the CPU oracle in SUM mode, reducing to rank 0 and checking against the sequential running mean."""
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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    from opengl_raytracer_amd import partition

    scene = pkg.load_scene(os.path.join(ROOT, "fixtures", "teapot.obj"), colour_seed=1)
    cam = pkg.make_camera()
    W, H = 96, 54
    if mode == "tiles":
        # image-tile partition: every rank renders the same frames, restricted to its interleaved band; the bands
        # are disjoint, so the sum-reduce of the otherwise zero buffers is the gather
        nrm = oracle.normals(scene)
        local = np.zeros((H, W, 4), dtype=np.float32)
        p = pkg.make_params(W, H, frames=1)
        rays_local = 0.0
        for frame in range(steps):
            for r0, r1 in partition.band_row_ranges(rank, world, H, True):
                rays_local += float(oracle.frame_rows(scene, nrm, cam, p, frame, local, r0, r1, 2)[2])
        t = torch.from_numpy(local)
        dist.reduce(t, dst=0)
        rays = torch.tensor([rays_local], dtype=torch.float64)
        dist.all_reduce(rays)
        if rank == 0:
            want, cnt_all = oracle.render(scene, cam, pkg.make_params(W, H, frames=steps), threads=2)
            same = bool(np.array_equal(t.numpy().view(np.uint32), want.view(np.uint32)))
            np.savez(out_path, same=same, rays=float(rays.item()), rays_seq=float(cnt_all[2]))
        dist.barrier()
        dist.destroy_process_group()
        return
    base, stride, n = partition.rank_frames(rank, world, 0, steps)
    local = np.zeros((H, W, 4), dtype=np.float32)
    p = pkg.make_params(W, H, frame_base=base, frame_stride=stride, frames=n, flags=pkg.RT_FLAG_SUM)
    local, cnt = oracle.render(scene, cam, p, local, threads=2)
    t = torch.from_numpy(local)
    dist.reduce(t, dst=0)
    rays = torch.tensor([float(cnt[2])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        image = t.numpy() / float(steps * world)
        want, cnt_all = oracle.render(scene, cam, pkg.make_params(W, H, frames=steps * world), threads=2)
        err = float(np.sqrt(((image[..., :3] - want[..., :3]) ** 2).sum() / (want[..., :3] ** 2).sum()))
        np.savez(out_path, err=err, rays=float(rays.item()), rays_seq=float(cnt_all[2]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
