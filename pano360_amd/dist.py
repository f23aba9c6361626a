"""Multi-GPU stitching: one process per GPU.

Two ways to use N GPUs, both driven by ``bench.py --gpus N``:

* **image sets** (throughput; ``assign_sets``): a stitching service sees a stream of
  independent image sets (one panorama per time step of a camera rig).  Sets are
  dealt out round-robin, every rank stitches its own sets with the single-GPU path
  and nothing crosses a GPU - the path partitions by object, so there is no
  data-path collective (weak scaling: per-GPU work is fixed as N grows).
* **column strips** (latency of ONE panorama; ``ShardedStitcher``): the mosaic is
  split into column strips, described below.

The reference is single-process; this is new design.  What
shards: the *mosaic*, by columns.  Rank r produces columns [c_r, c_{r+1}) of
the final mosaic from the frames whose patches reach that strip, using exactly
the single-GPU kernels restricted to the strip (``Engine.multiband_fused``,
``strip=``): ownership is evaluated on the strip grown by the blur radius, the
patches' rectangles are cut to the strip, so no partial sums ever cross a GPU
and each strip equals the same columns of the single-GPU mosaic bit for bit.
The only exchange is the composition of the finished uint8 strips, a gather
onto rank 0 (RCCL point-to-point sends over xGMI, all seven links into rank 0
busy at once - not a ring, which would be bound by one link).

Frames are resident where they are needed: rank r holds the contiguous run of
frames whose patch rectangles come within two blur radii of its strip, so
neighbouring ranks both hold the few frames that straddle their boundary.
"""

from . import engine as _eng


def assign_sets(n_sets, rank, world):
    """Indices of the image sets rank ``rank`` stitches: round-robin, so that a
    stream of sets keeps every GPU equally busy whatever its length."""
    return list(range(rank, n_sets, world))


def max_over_ranks(seconds, device="cpu", group=None):
    """The slowest rank's time: what a job that waits for every rank took."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.item()


def strip_bounds(width, world):
    """Column boundaries c_0 = 0 <= c_1 <= ... <= c_world = width, equal widths."""
    return [int(round(width * r / world)) for r in range(world + 1)]


def frames_for_strip(rects, strip, margin):
    """Indices of the frames whose patch rectangle intersects the strip grown by
    ``margin`` columns (a superset of what the strip's windows can touch)."""
    lo, hi = strip[0] - margin, strip[1] + margin
    return [i for i, (_, _, x0, x1) in enumerate(rects) if x0 < hi and x1 > lo]


def pack_strip(mosaic, strip, width):
    """Columns ``strip`` of an [H][W][3] mosaic as a dense [H][width][3] tensor
    (zero padded on the right), the unit of the gather."""
    import torch
    c0, c1 = strip
    out = torch.zeros((mosaic.shape[0], width, 3), dtype=mosaic.dtype, device=mosaic.device)
    out[:, :c1 - c0] = mosaic[:, c0:c1]
    return out


def gather_strips(packed, bounds, rank, world, group=None):
    """Compose the mosaic on rank 0 from every rank's packed strip.  Returns the
    [H][W][3] mosaic on rank 0, None elsewhere.  world == 1 needs no process
    group."""
    import torch
    width = bounds[-1]
    if world == 1:
        return packed[:, :width].contiguous()
    import torch.distributed as dist
    device = packed.device
    if dist.get_backend(group) == "gloo":      # CPU rendezvous (tests, 1-GPU dry runs)
        packed = packed.cpu()
    parts = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
    dist.gather(packed, parts, dst=0, group=group)
    if rank != 0:
        return None
    mosaic = torch.empty((packed.shape[0], width, 3), dtype=packed.dtype, device=device)
    for r in range(world):
        c0, c1 = bounds[r], bounds[r + 1]
        mosaic[:, c0:c1] = parts[r][:, :c1 - c0]
    return mosaic


class ShardedStitcher:
    """Strong-scaling driver used by ``bench.py --gpus N`` and the tests.

    ``my_frames``: the camera indices whose frames must be uploaded on this
    rank before ``step`` is called (fixed by the cameras, decided here once).
    """

    def __init__(self, eng, shapes, rots, intrs, n_levels, rank, world, max_resolution=10 ** 9,
                 group=None):
        self.eng, self.rank, self.world, self.group = eng, rank, world, group
        self.shapes, self.rots, self.intrs = shapes, rots, intrs
        self.n_levels, self.max_resolution = n_levels, max_resolution
        plan = _eng.Plan(shapes, rots, intrs, True, max_resolution)
        radius = max([_eng.gaussian_ksize(s) // 2 for s in _eng.level_sigmas(n_levels)],
                     default=0)
        self.bounds = strip_bounds(plan.shape[1], world)
        self.strip = (self.bounds[rank], self.bounds[rank + 1])
        self.pack_width = max(b - a for a, b in zip(self.bounds[:-1], self.bounds[1:]))
        # windows reach one radius past the strip for A's owners and one more for V
        self.my_frames = frames_for_strip(plan.rects, self.strip, 2 * radius)

    def step(self, frames):
        """frames[j] = device tensor of camera my_frames[j].  Returns
        (plan, mosaic on rank 0 / None elsewhere, this rank's patches)."""
        plan = _eng.Plan(self.shapes, self.rots, self.intrs, True, self.max_resolution)
        self.eng.upload_plan(plan)
        mosaic, _, _, patches = self.eng.multiband_fused(
            frames, plan, self.n_levels, frame_ids=self.my_frames, strip=self.strip)
        packed = pack_strip(mosaic, self.strip, self.pack_width)
        full = gather_strips(packed, self.bounds, self.rank, self.world, self.group)
        return plan, full, list(patches)          # window geometry only, not the arenas


def emulate_on_one_device(eng, imgs, rots, intrs, n_levels, world, max_resolution=10 ** 9):
    """Run every rank's ``step`` one after the other on a single GPU and compose
    the strips locally - the test double of an N-GPU run."""
    import torch
    shapes = [im.shape[:2] for im in imgs]
    strips, bounds = [], None
    for rank in range(world):
        st = ShardedStitcher(eng, shapes, rots, intrs, n_levels, rank, world, max_resolution)
        st.world = 1                       # no process group: keep the packed strip
        frames = eng.upload_frames([imgs[i] for i in st.my_frames])
        plan = _eng.Plan(shapes, rots, intrs, True, max_resolution)
        eng.upload_plan(plan)
        mosaic, _, _, _ = eng.multiband_fused(frames, plan, n_levels, frame_ids=st.my_frames,
                                              strip=st.strip)
        strips.append(mosaic[:, st.strip[0]:st.strip[1]])
        bounds = st.bounds
    return torch.cat(strips, dim=1), bounds
