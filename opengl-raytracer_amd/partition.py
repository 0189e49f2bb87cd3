"""Sample partition of a progressive render over the GPUs of one node.

The path shards by samples: the RNG seed is a pure function of (pixel, frameCount)
(raytrace.glsl:277) and accumulation is a sum, so rank g of N renders frameCount g, g+N, g+2N, ...
into a private SUM buffer (RT_FLAG_SUM, frame_stride = N); one reduce to rank 0 and a division by the
total frame count give the image.  No collective is needed while rendering.
"""
from __future__ import annotations


def rank_frames(rank: int, world: int, first_step: int, steps: int):
    """(frame_base, frame_stride, frames) of `steps` steps of this rank starting at step `first_step`;
    step s of rank g is frameCount s*world + g."""
    if not (0 <= rank < world) or steps < 0 or first_step < 0:
        raise ValueError("bad partition arguments")
    return first_step * world + rank, world, steps


def all_frames(world: int, first_step: int, steps: int):
    """The frameCounts covered by all ranks together, for checking that the partition is exact."""
    out = []
    for g in range(world):
        base, stride, n = rank_frames(g, world, first_step, steps)
        out.extend(base + k * stride for k in range(n))
    return sorted(out)


def band_rows(index: int, count: int, height: int):
    """Pixel rows [r0, r1) of horizontal band `index` of `count` (the image-tile partition): bands are runs of
    8-pixel tile rows, band i = tile rows [i*R//count, (i+1)*R//count) with R = ceil(height / 8)."""
    if not (0 <= index < count):
        raise ValueError("bad band")
    R = (height + 7) // 8
    t0, t1 = index * R // count, (index + 1) * R // count
    return min(t0 * 8, height), min(t1 * 8, height)


def band_row_ranges(index: int, count: int, height: int, interleaved: bool = False):
    """Pixel-row ranges [(r0, r1), ...] of band `index` of `count`.  Contiguous: the one range of band_rows().
    Interleaved: the stripes of 16 pixel rows number index, index + count, index + 2*count, ... -- an object in the
    middle of the image is then shared by all bands instead of landing on the middle ones."""
    if not interleaved or count == 1:
        r0, r1 = band_rows(index, count, height)
        return [(r0, r1)] if r1 > r0 else []
    if not (0 <= index < count):
        raise ValueError("bad band")
    out = []
    r = index * 16
    while r < height:
        out.append((r, min(r + 16, height)))
        r += 16 * count
    return out


def render_partitioned(*, rank: int, world: int, steps: int, height: int, render, start_reduce, finish,
                       mode: str = "samples", reduce_bands: int = 1, first_step: int = 0, launch_frames: int = 16):
    """This rank's share of a partitioned accumulation and the collective that closes it; the renderer and the
    collective are passed in, so the same code runs on RCCL with the HIP library and on gloo with a CPU renderer.

      render(frame_base, frame_stride, frames, band, rows)   enqueue these frames into the rank's buffer; band is None
                                                             or (index, count) of a CONTIGUOUS band, rows its pixel-row
                                                             range (r0, r1); in 'samples' mode the buffer receives SUMS
      start_reduce(rows) -> handle                           start the sum-reduce of buffer[r0:r1] to rank 0 (may be
                                                             asynchronous)
      finish(handle)                                         wait for it

    mode 'samples' (weak scaling): rank g renders frameCount (first_step+s)*world + g for s in [0, steps).  The closing
    reduce costs time in proportion to the rows reduced, whatever the number of frames summed in them.  With
    reduce_bands > 1 only the frames of the rank's LAST launch (the library renders launch_frames frames per launch) are
    rendered band by band: everything before them goes out as whole-image launches, then the image is finished in
    `reduce_bands` contiguous row bands one after the other, the reduce of band b travelling while band b+1 renders, so
    that only the last band's reduce (1/reduce_bands of the bytes) is exposed -- at the price of reduce_bands short
    launches in place of one.  Bands are bit-identical to the whole image (seeds use global pixel indices, sums add
    the same numbers in the same order).  Rank 0 must scale the reduced sums by 1 / (steps * world) afterwards.

    mode 'tiles' (strong scaling): every rank renders the same `steps` frames, restricted to its INTERLEAVED band
    (band_row_ranges(rank, world, height, True)), running mean; the buffers are zero elsewhere, so one sum-reduce of
    the whole image is the gather.  The band passed to render is then (rank, world) and rows is None.

    Returns the list of row ranges reduced."""
    if mode == "tiles":
        render(first_step, 1, steps, (rank, world), None)
        finish(start_reduce((0, height)))
        return [(0, height)]
    if mode != "samples":
        raise ValueError("mode must be 'samples' or 'tiles'")
    base, stride, frames = rank_frames(rank, world, first_step, steps)
    count = max(1, min(int(reduce_bands), (height + 7) // 8))
    # the frames before the rank's last launch: whole image, no reduce yet
    head = 0
    if count > 1 and frames > 0 and launch_frames > 0:
        head = frames - ((frames - 1) % launch_frames + 1)
        if head > 0:
            render(base, stride, head, None, (0, height))
    pending, done = [], []
    for b in range(count):
        rows = band_rows(b, count, height)
        if rows[1] <= rows[0]:
            continue
        render(base + head * stride, stride, frames - head, None if count == 1 else (b, count), rows)
        pending.append(start_reduce(rows))
        done.append(rows)
    for h in pending:
        finish(h)
    return done
