"""Multi-GPU stitching: one process per GPU.

The reference is single-process; this is new design.  What shards is the
*mosaic*, by columns (``ShardedStitcher``): rank r produces columns [c_r, c_{r+1}) of the
final mosaic from the frames whose patches reach that strip, using exactly the single-GPU
kernels restricted to the strip (``Engine.multiband_fused``, ``strip=``).  Ownership is
evaluated on the strip grown by the blur radius and the patches' rectangles are cut to the
strip, so no partial sums ever cross a GPU and each strip equals the same columns of the
single-GPU mosaic bit for bit.  Frames are resident where they are needed: rank r holds the
contiguous run of frames whose patch rectangles come within two blur radii of its strip.

The one exchange is the composition of the finished uint8 strips on rank 0
(``StripExchange``), in either of two forms:

* ``"gather"`` (default): every rank sends its packed strip - 1/N of the mosaic - to rank 0
  (RCCL point-to-point, all xGMI links into rank 0 busy at once);
* ``"reduce"``: every rank writes its strip into a zero-initialised full-width mosaic and one
  RCCL ``reduce(sum)`` onto rank 0 composites them - the supports are disjoint, so the sum
  IS the composition, byte for byte.  It moves N times the bytes of the gather (every rank
  contributes a whole mosaic), which a ring over point-to-point xGMI links pays in full.

Either way the exchange of stitch k is asynchronous (RCCL's own stream) and overlaps the
kernels of stitch k + 1; buffers are allocated once and cycled.

``assign_sets`` deals independent image sets out to the ranks (replicas of the single-GPU
path, no collective): the secondary throughput figure of ``bench.py``.
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


def describe_job(device="cpu", reduce_device="cpu", group=None):
    """What the process group saw, for the bench line: world size, backend, every rank's
    device (index and name, gathered) and the checksum of a one-element all-reduce - the sum
    of (rank + 1) over the ranks, which must be world (world + 1) / 2 on every rank.  A job
    without a process group describes itself alone."""
    import torch
    import torch.distributed as dist
    name = str(device)
    if str(device).startswith("cuda") and torch.cuda.is_available():
        idx = torch.device(device).index or 0
        name = f"cuda:{idx} {torch.cuda.get_device_name(idx)}"
    if not (dist.is_available() and dist.is_initialized()):
        return dict(world_size=1, backend=None, devices=[name], allreduce_checksum=1,
                    allreduce_expected=1)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=reduce_device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    names = [None] * world
    dist.all_gather_object(names, name, group=group)
    return dict(world_size=world, backend=dist.get_backend(group), devices=names,
                allreduce_checksum=int(round(t.item())),
                allreduce_expected=world * (world + 1) // 2)


def preflight(device="cpu", reduce_device="cpu", group=None, log=None, nbytes=1 << 20):
    """The first real multi-rank exchange of a job, made cheap to diagnose: a record of who is
    here (``log``: rank 0's sink, called BEFORE the first collective - world, rank, backend, this
    rank's device), then the exchange's two collectives on one megabyte each - ``dist.gather`` of
    uint8 pieces onto rank 0 and ``dist.reduce(sum)`` of a uint8 buffer, the calls
    ``StripExchange.submit`` makes, on ``reduce_device`` - and their checks.  Returns the record
    ({"gather_ok", "reduce_ok", "ms"}); raises ``RuntimeError`` when a check fails.  Callers run it
    under their watchdog: a communicator that cannot move a megabyte hangs HERE."""
    import time
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"world_size": 1, "skipped": "no process group"}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if log is not None:
        import json
        log("bench preflight " + json.dumps({
            "stage": "before first collective", "world_size": world, "rank": rank,
            "backend": dist.get_backend(group), "device": str(device),
            "exchange_device": str(reduce_device)}))
    t0 = time.perf_counter()
    piece = torch.full((nbytes,), rank + 1, dtype=torch.uint8, device=reduce_device)
    parts = ([torch.empty_like(piece) for _ in range(world)] if rank == 0 else None)
    dist.gather(piece, parts, dst=0, group=group)
    gather_ok = True
    if rank == 0:
        gather_ok = all(bool((p == r + 1).all().item()) for r, p in enumerate(parts))
    # disjoint supports, as the strips are: rank r's slice holds r + 1, the sum is the composition
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=reduce_device)
    lo, hi = nbytes * rank // world, nbytes * (rank + 1) // world
    buf[lo:hi] = rank + 1
    dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, group=group)
    reduce_ok = True
    if rank == 0:
        want = torch.zeros(nbytes, dtype=torch.uint8, device=reduce_device)
        for r in range(world):
            want[nbytes * r // world:nbytes * (r + 1) // world] = r + 1
        reduce_ok = bool(torch.equal(buf, want))
    if str(reduce_device).startswith("cuda"):
        torch.cuda.synchronize()
    out = {"world_size": world, "bytes": nbytes, "gather_ok": gather_ok, "reduce_ok": reduce_ok,
           "ms": (time.perf_counter() - t0) * 1e3}
    if log is not None:
        import json
        log("bench preflight " + json.dumps(dict(out, stage="after gather + reduce")))
    if not (gather_ok and reduce_ok):
        raise RuntimeError(f"preflight exchange failed: {out}")
    return out


def strip_bounds(width, world):
    """Column boundaries c_0 = 0 <= c_1 <= ... <= c_world = width, equal widths."""
    return [int(round(width * r / world)) for r in range(world + 1)]


def balanced_strip_bounds(cost, world, align=8, min_width=64):
    """Column boundaries that give every rank the same share of ``cost`` (one non-negative
    number per mosaic column, ``Engine.column_costs``) instead of the same width: cuts at the
    world-quantiles of the running sum, moved to multiples of ``align`` (the interior block), every
    strip at least ``min_width`` wide.  Equal widths when the mosaic is too narrow for that."""
    import numpy as np
    cost = np.maximum(np.asarray(cost, dtype=np.float64), 0.0)
    width = len(cost)
    if world < 1 or width < world * max(min_width, align) or not np.isfinite(cost).all() \
            or cost.sum() <= 0.0:
        return strip_bounds(width, max(world, 1))
    run = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = np.searchsorted(run, run[-1] * np.arange(1, world) / world, side="left")
    cuts = [int(round(c / align)) * align for c in cuts]
    bounds = [0] + cuts + [width]
    for r in range(1, world):                       # monotone, no strip thinner than min_width
        bounds[r] = max(bounds[r], bounds[r - 1] + min_width)
    for r in range(world - 1, 0, -1):
        bounds[r] = min(bounds[r], bounds[r + 1] - min_width)
    return bounds


def frames_for_strip(rects, strip, margin):
    """Indices of the frames whose patch rectangle intersects the strip grown by
    ``margin`` columns (a superset of what the strip's windows can touch)."""
    lo, hi = strip[0] - margin, strip[1] + margin
    return [i for i, (_, _, x0, x1) in enumerate(rects) if x0 < hi and x1 > lo]


class StripExchange:
    """Composes the ranks' finished strips into the mosaic on rank 0, one stitch behind the
    kernels: ``target()`` hands out the full-size [H][W][3] uint8 buffer the collapse writes
    its strip into, ``submit()`` starts the (asynchronous) exchange of that buffer and
    returns at once, ``collect()`` returns the oldest finished mosaic (rank 0; ``None``
    elsewhere) - ``depth`` sets of buffers are cycled, so up to ``depth - 1`` exchanges are
    in flight behind the stitch being computed.  ``world == 1`` needs no process group and
    issues no collective - unless ``force_collective`` is set: then a one-rank process group
    must be up and the very same ``dist.gather`` / ``dist.reduce`` calls run as at world N
    (device buffers, ``async_op=True``, view lists, ``work.wait()``), which is how a one-GPU
    box executes the RCCL branch (tests/test_gpu_parity.py)."""

    MODES = ("gather", "reduce")

    def __init__(self, shape, bounds, rank, world, device, mode="gather", group=None, depth=2,
                 force_collective=False):
        import torch
        if mode not in self.MODES:
            raise ValueError(f"exchange {mode!r}: one of {self.MODES}")
        self.H, self.W = shape
        self.bounds, self.rank, self.world, self.group = list(bounds), rank, world, group
        self.strip = (self.bounds[rank], self.bounds[rank + 1])
        self.mode, self.depth, self.device = mode, max(int(depth), 1), device
        self.host_staged = False
        self.collective = world > 1 or bool(force_collective)
        if self.collective:
            import torch.distributed as dist
            if dist.get_world_size(group) != world:
                raise ValueError(f"process group has {dist.get_world_size(group)} ranks, "
                                 f"the strips were cut for {world}")
            # gloo moves host memory only (CPU tests, 1-GPU dry runs): stage through the host
            self.host_staged = (dist.get_backend(group) == "gloo"
                                and torch.device(device).type != "cpu")
        widths = [b - a for a, b in zip(self.bounds[:-1], self.bounds[1:])]
        self.pack_w = max(widths)
        self.even = len(set(widths)) == 1
        u8 = dict(dtype=torch.uint8, device=device)
        # the collapse writes only its strip's columns: everything else stays zero for good
        # (reduce: rank 0's copy receives the sum in place and is wiped again in collect)
        self.full = [torch.zeros((self.H, self.W, 3), **u8)
                     for _ in range(self.depth if mode == "reduce" or not self.collective else 1)]
        if mode == "gather" and self.collective:
            self.packed = [torch.zeros((self.H, self.pack_w, 3), **u8) for _ in range(self.depth)]
            self.parts = ([torch.empty((world, self.H, self.pack_w, 3), **u8)
                           for _ in range(self.depth)] if rank == 0 else None)
            self.mosaic = ([torch.empty((self.H, self.W, 3), **u8) for _ in range(self.depth)]
                           if rank == 0 else None)
        if mode == "reduce" and self.collective:
            # rank 0 hands the sum out as a copy in a ring of its own, as the gather does: the
            # buffer that received it is wiped by recycle() at the very next step
            self.mosaic = ([torch.empty((self.H, self.W, 3), **u8) for _ in range(self.depth)]
                           if rank == 0 else None)
        self.slot = 0
        self.inflight = []              # [(slot, work or None)], oldest first

    def target(self):
        """The buffer the next stitch's collapse writes its strip into."""
        if len(self.inflight) >= self.depth:
            raise RuntimeError("collect() the finished mosaic before starting another stitch")
        return self.full[self.slot % len(self.full)]

    def submit(self):
        """Start the exchange of the buffer ``target()`` handed out."""
        import torch
        slot = self.slot
        self.slot = (self.slot + 1) % self.depth
        work = None
        if self.collective:
            import torch.distributed as dist
            if self.mode == "reduce":
                buf = self.full[slot]
                if self.host_staged:
                    host = buf.cpu()
                    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM, group=self.group)
                    if self.rank == 0:
                        buf.copy_(host)
                else:
                    work = dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, group=self.group,
                                       async_op=True)
            else:
                c0, c1 = self.strip
                packed = self.packed[slot]
                packed[:, :c1 - c0].copy_(self.full[0][:, c0:c1])
                parts = self.parts[slot] if self.rank == 0 else None
                if self.host_staged:
                    hp = packed.cpu()
                    hparts = ([torch.empty_like(hp) for _ in range(self.world)]
                              if self.rank == 0 else None)
                    dist.gather(hp, hparts, dst=0, group=self.group)
                    if self.rank == 0:
                        parts.copy_(torch.stack(hparts))
                else:
                    work = dist.gather(packed, list(parts.unbind(0)) if self.rank == 0 else None,
                                       dst=0, group=self.group, async_op=True)
        self.inflight.append((slot, work))

    def collect(self):
        """The oldest stitch's mosaic, [H][W][3] uint8 on rank 0 (valid until its buffers
        come round again, ``depth`` stitches later), ``None`` on the other ranks."""
        slot, work = self.inflight.pop(0)
        if work is not None:
            work.wait()                  # orders the current stream behind the collective
        if self.rank != 0:
            return None
        if not self.collective:
            return self.full[slot]
        if self.mode == "reduce":
            self.mosaic[slot].copy_(self.full[slot])
            return self.mosaic[slot]
        parts, mosaic = self.parts[slot], self.mosaic[slot]
        if self.even:
            mosaic.view(self.H, self.world, self.pack_w, 3).copy_(parts.permute(1, 0, 2, 3))
        else:
            for r in range(self.world):
                c0, c1 = self.bounds[r], self.bounds[r + 1]
                mosaic[:, c0:c1].copy_(parts[r, :, :c1 - c0])
        return mosaic

    def recycle(self):
        """reduce mode: on rank 0 the sum landed in this rank's strip buffer, and a backend may
        use the other ranks' buffers as workspace (gloo does): wipe the columns outside the
        strip before the buffer takes the next one (a memset, ~20 us for 100 MB)."""
        if self.mode != "reduce" or not self.collective:
            return
        c0, c1 = self.strip
        buf = self.full[self.slot % len(self.full)]
        if c0 > 0:
            buf[:, :c0].zero_()
        if c1 < self.W:
            buf[:, c1:].zero_()

    def drain(self):
        """Finish every exchange in flight; returns the last mosaic (rank 0) or None."""
        last = None
        while self.inflight:
            last = self.collect()
        return last


class ShardedStitcher:
    """One panorama over ``world`` GPUs (strong scaling); used by ``bench.py --gpus N`` and
    the tests.

    ``my_frames``: the camera indices whose frames must be uploaded on this rank before
    ``step`` is called (fixed by the cameras, decided here once).  ``step`` runs this rank's
    strip of one stitch and starts its exchange; the composed mosaic of the PREVIOUS step
    comes back from the same call (rank 0), the last one from ``finish()``."""

    def __init__(self, eng, shapes, rots, intrs, n_levels, rank, world, max_resolution=10 ** 9,
                 group=None, exchange="gather", depth=2, cache_plan=True,
                 force_collective=False, lane_groups="shared", trust_layouts=None,
                 keep_geometry=False, balance=True):
        # ``eng``: one engine, or a list of them - "lanes": consecutive stitches then alternate
        # between the engines, each on a stream of its own with its own exchange buffers, so
        # that one stitch's kernels cover the other's host round trip (on a column strip of a
        # world-8 run the kernels are 0.45 ms and that round trip plus the launches' gaps 0.1).
        # Every buffer of a lane is only ever touched in its lane's stream order.
        # ``lane_groups``: which communicator a lane's exchange runs on.
        # * "shared" (default): all lanes use ``group``.  The lanes are driven in lock step -
        #   ``step`` takes them round-robin from one thread, the same on every rank - so every
        #   rank issues its collectives in ONE order on ONE communicator, and torch runs them on
        #   that communicator's own stream in that order (each waits for its lane's stream, the
        #   lane's stream waits for it in ``collect``): the canonical RCCL usage; nothing of two
        #   communicators can wait for each other on the device.
        # * "own": a communicator per lane (``dist.new_group``), so that a lane's collective can
        #   only pair with the SAME lane's on the other ranks whatever order the lanes are driven
        #   in (tests/test_dist_cpu.py drives them from skewed threads, on gloo).  On RCCL
        #   collectives of different communicators run side by side, and the library asks for
        #   one issue order across ranks all the same; the lock-step order satisfies it, but
        #   nothing is gained over "shared" while the lanes are driven from one thread.  EVERY
        #   rank of the default group must then construct the stitcher (``dist.new_group`` is
        #   collective over it), in the same order, and ``close()`` it to give them back.
        engines = list(eng) if isinstance(eng, (list, tuple)) else [eng]
        self.eng, self.rank, self.world, self.group = engines[0], rank, world, group
        self.shapes, self.rots, self.intrs = shapes, rots, intrs
        self.n_levels, self.max_resolution = n_levels, max_resolution
        # cache_plan (default): the host geometry of the cameras comes out of the engine's
        # content-keyed memo (Engine.cached_plan / engine.PlanMemo: same shapes, rotations,
        # calibrations, padding, cap and table columns -> the plan the reference's per-stitch
        # recomputation, stitcher.py:276-302, would produce again, bit for bit; a camera moved by
        # one ulp misses).  At eight ranks the 0.36 ms of float64 NumPy - the same on every rank -
        # are as long as a strip's kernels.  False: ``Plan(...)`` per stitch.
        self.cache_plan = cache_plan
        plan = _eng.Plan(shapes, rots, intrs, True, max_resolution)
        radius = max([_eng.gaussian_ksize(s) // 2 for s in _eng.level_sigmas(n_levels)],
                     default=0)
        # ``balance`` (default): strips of equal WORK - the engine's cost of every column, from the
        # geometry alone (Engine.column_costs), cut at its quantiles; rank 0's cut is broadcast when
        # a process group is up, so that every rank uses the same whatever its GPU computed.
        # Engines without a device (CPU tests) and ``balance=False``: equal widths.
        self.bounds = strip_bounds(plan.shape[1], world)
        self.balanced = False
        if balance and world > 1 and hasattr(engines[0], "column_costs"):
            self.bounds = self._balanced_bounds(engines[0], plan, n_levels, world, group,
                                                bool(exchange))
            self.balanced = True
        self.strip = (self.bounds[rank], self.bounds[rank + 1])
        # windows reach one radius past the strip for A's owners and one more for V
        self.my_frames = frames_for_strip(plan.rects, self.strip, 2 * radius)
        # mosaic columns whose sin / cos this rank's kernels read: the strip, the reach of the
        # interior test around it (ownership) and one radius more (windows V), generously
        self.table_cols = (self.strip[0] - 4 * radius - 64, self.strip[1] + 4 * radius + 64)
        self.depth = max(int(depth), 1)
        lane_depth = self.depth          # (every lane can hold the whole pipeline: any pattern of lanes)
        self.lanes = []
        self._lane_groups = []
        self._restore = []
        for i, use in enumerate(engines):
            stream = None
            if i > 0 and str(getattr(use, "device", "cpu")).startswith("cuda"):
                import torch
                stream = torch.cuda.Stream(use.device)
            # exchange=None: geometry only (emulation of the ranks on one device)
            lane_group = group
            if lane_groups not in ("shared", "own"):
                raise ValueError(f"lane_groups {lane_groups!r}: 'shared' or 'own'")
            if (lane_groups == "own" and exchange and (world > 1 or force_collective)
                    and len(engines) > 1):
                import torch.distributed as dist
                # (collective: every rank builds its lanes in the same order)
                lane_group = dist.new_group(ranks=(dist.get_process_group_ranks(group)
                                                   if group is not None else None))
                self._lane_groups.append(lane_group)
            ex = (StripExchange(plan.shape, self.bounds, rank, world, use.device, exchange,
                                lane_group, lane_depth, force_collective) if exchange else None)
            if stream is not None:
                # the exchange buffers were zero-filled on the constructing stream: the lane's
                # first write must come behind those fills
                import torch
                stream.wait_stream(torch.cuda.current_stream(use.device))
            # with the plan out of the memo a lane's stitch repeats its previous one's Plan
            # object: the engine then queues it with the verified layout and does not wait
            # (Engine.trust_layouts; checked when the mosaic is collected)
            # (``trust_layouts``: None = whenever the plan comes out of the memo; False = never)
            # (``keep_geometry``: a trusted repeat also re-uses the owner map, masks, record table
            # and work list its lane's previous stitch left on the device - Engine.keep_geometry)
            if exchange and hasattr(use, "trust_layouts"):
                # (what the engine was set to before: close() puts it back)
                self._restore.append((use, bool(getattr(use, "trust_layout", False)),
                                      bool(getattr(use, "keep_geometry", False)),
                                      use.get_option(_eng._lib.OPT_STITCH_ASYNC)
                                      if hasattr(use, "get_option") else None))
                use.trust_layouts(bool(cache_plan) if trust_layouts is None
                                  else bool(trust_layouts and cache_plan),
                                  keep_geometry=bool(keep_geometry))
            self.lanes.append((use, stream, ex))
        self.exchange = self.lanes[0][2]
        self.count = 0
        self.order = []                 # lanes of the stitches whose mosaics are still to come

    @staticmethod
    def _balanced_bounds(eng, plan, n_levels, world, group, collective):
        import torch
        import torch.distributed as dist
        bounds = balanced_strip_bounds(eng.column_costs(plan, n_levels), world)
        if collective and dist.is_available() and dist.is_initialized() \
                and dist.get_world_size(group) == world:
            staged = dist.get_backend(group) == "gloo"
            t = torch.tensor(bounds, dtype=torch.int64, device="cpu" if staged else eng.device)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0,
                           group=group)
            bounds = [int(v) for v in t.cpu().tolist()]
        return bounds

    @staticmethod
    def _on(stream):
        import contextlib
        if stream is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(stream)

    def _collect_oldest(self):
        use, stream, ex = self.lanes[self.order.pop(0)]
        if getattr(use, "trust_layout", False):
            use.verify_trusted()        # (that stitch's layout kernel finished a stitch ago)
        with self._on(stream):
            mosaic = ex.collect()
        if stream is not None:
            # collect()'s copies were queued on the lane's private stream: order the caller's
            # current stream behind them, so that `mosaic.cpu()` or a kernel on the caller's
            # stream never reads a half-written buffer
            import torch
            torch.cuda.current_stream(use.device).wait_stream(stream)
        return mosaic

    def step(self, frames):
        """frames[j] = device tensor of camera my_frames[j].  Returns (plan, the mosaic of the
        stitch ``depth - 1`` steps back on rank 0 / None, this rank's patches).

        Stream contract: every lane works on a stream of its own; the mosaic handed back is
        ordered behind the CALLER's current stream (the caller's stream waits for the lane's
        copies before this returns), so work the caller queues on its current stream - or a
        ``.cpu()`` - sees the finished mosaic.  The buffer stays valid until its lane has
        taken ``depth`` more stitches."""
        if self.exchange is None:
            raise RuntimeError("ShardedStitcher(exchange=None) holds the strip geometry only "
                               "(emulate_on_one_device); step() needs an exchange mode")
        lane = self.count % len(self.lanes)
        self.count += 1
        use, stream, ex = self.lanes[lane]
        with self._on(stream):
            if self.cache_plan:
                plan = use.cached_plan(self.shapes, self.rots, self.intrs, True,
                                       self.max_resolution, self.table_cols)
            else:
                plan = _eng.Plan(self.shapes, self.rots, self.intrs, True, self.max_resolution,
                                 table_cols=self.table_cols)
                use.upload_plan(plan)
            ex.recycle()
            _, _, _, patches = use.multiband_fused(
                frames, plan, self.n_levels, frame_ids=self.my_frames, strip=self.strip,
                mosaic_out=ex.target())
            ex.submit()
        self.order.append(lane)
        # the exchange just started runs behind the next stitch's kernels; what is handed
        # back is the stitch depth - 1 steps before it (depth 1: this one, synchronously)
        previous = self._collect_oldest() if len(self.order) > self.depth - 1 else None
        return plan, previous, list(patches)          # window geometry only, not the arenas

    def finish(self):
        """Completes the exchanges in flight; the last stitch's mosaic on rank 0 (same stream
        contract as ``step``: ordered behind the caller's current stream)."""
        last = None
        while self.order:
            last = self._collect_oldest()
        return last

    def close(self):
        """Finishes what is in flight and destroys the lanes' communicators (collective: every
        rank closes its stitcher, in the same order it built it).  The stitcher cannot step
        afterwards."""
        self.finish()
        # the engines go back to the trust / layout settings they came with: a later direct use
        # must not stay in trusted mode with nobody calling verify_trusted
        restore, self._restore = self._restore, []
        for use, trust, keep, async_opt in restore:
            use.trust_layouts(trust, keep_geometry=keep)
            if async_opt is not None:
                use.set_option(_eng._lib.OPT_STITCH_ASYNC, async_opt)
        groups, self._lane_groups = self._lane_groups, []
        if groups:
            import torch.distributed as dist
            for g in groups:
                dist.destroy_process_group(g)
        self.exchange = None


def emulate_on_one_device(eng, imgs, rots, intrs, n_levels, world, max_resolution=10 ** 9,
                          balance=True):
    """Run every rank's strip one after the other on a single GPU and compose
    the strips locally - the test double of an N-GPU run."""
    import torch
    shapes = [im.shape[:2] for im in imgs]
    strips, bounds = [], None
    for rank in range(world):
        st = ShardedStitcher(eng, shapes, rots, intrs, n_levels, rank, world, max_resolution,
                             exchange=None, balance=balance)
        frames = eng.upload_frames([imgs[i] for i in st.my_frames])
        plan = _eng.Plan(shapes, rots, intrs, True, max_resolution, table_cols=st.table_cols)
        eng.upload_plan(plan)
        mosaic, _, _, _ = eng.multiband_fused(frames, plan, n_levels, frame_ids=st.my_frames,
                                              strip=st.strip)
        strips.append(mosaic[:, st.strip[0]:st.strip[1]])
        bounds = st.bounds
    return torch.cat(strips, dim=1), bounds
