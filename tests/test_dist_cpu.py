"""CPU, world_size 2 and 3 over gloo: the only exchange of the multi-GPU path (the
composition of the ranks' strips on rank 0: point-to-point gather, or sum-reduce of
zero-padded mosaics; pipelined one stitch deep) driven through the same ShardedStitcher
steps bench.py --gpus N runs, and the host-side sharding logic.  The kernels themselves
need a GPU; their strip-restricted form is checked against the whole-mosaic run in
tests/test_gpu_parity.py::test_column_strips_compose_the_single_gpu_mosaic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pano360_amd import dist as pdist
from pano360_amd import engine, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _HostEngine:
    """Stands in for the HIP engine on a GPU-less host: "stitches" strip [c0, c1) of step k
    by writing the known mosaic of step k into those columns of the target buffer and
    nothing else - what Engine.multiband_fused(strip=, mosaic_out=) does with kernels."""
    device = "cpu"

    def __init__(self, shape, clock=None):
        self.shape = shape
        self.clock = clock if clock is not None else [0]     # shared by the engines of one stitcher

    @property
    def k(self):
        return self.clock[0]

    @k.setter
    def k(self, value):
        self.clock[0] = value

    def truth(self, k):
        H, W = self.shape
        base = torch.arange(H * W * 3, dtype=torch.int64).reshape(H, W, 3)
        return (base * 7 + 13 * k).remainder(251).to(torch.uint8)

    def upload_plan(self, plan):
        return plan

    def cached_plan(self, shapes, rots, intrs, padded, max_resolution, table_cols=None):
        self.cached = getattr(self, "cached", 0) + 1        # Engine.cached_plan's signature
        return engine.Plan(shapes, rots, intrs, padded, max_resolution, table_cols)

    def multiband_fused(self, frames, plan, n_levels, frame_ids=None, strip=None,
                        mosaic_out=None, **_):
        assert tuple(mosaic_out.shape[:2]) == self.shape == tuple(plan.shape)
        c0, c1 = strip
        mosaic_out[:, c0:c1] = self.truth(self.k)[:, c0:c1]
        self.k += 1
        return mosaic_out, None, None, []


def _scene():
    rots, intrs = synth.make_cameras(5, 160, 90, sweep_deg=80.0)
    return [(90, 160)] * 5, rots, intrs


def _worker(rank, world, port, mode, depth, result, cache_plan=False, lanes=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes, rots, intrs = _scene()
        shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape
        eng = _HostEngine(shape)
        engines = [eng] + [_HostEngine(shape, eng.clock) for _ in range(lanes - 1)]
        # the bench's strips step: ShardedStitcher over the process group, K steps, finish
        st = pdist.ShardedStitcher(engines if lanes > 1 else eng, shapes, rots, intrs, 5, rank,
                                   world, exchange=mode, depth=depth, cache_plan=cache_plan)
        assert st.exchange.world == dist.get_world_size() == world and len(st.lanes) == lanes
        got = []
        for _ in range(5):
            _, previous, _ = st.step(frames=None)
            if previous is not None:
                got.append(previous.clone())
        last = st.finish()
        if last is not None:
            got.append(last.clone())
        assert sum(getattr(e, "cached", 0) for e in engines) == (5 if cache_plan else 0)
        if rank == 0:
            ok = len(got) == 5 and all(torch.equal(m, eng.truth(k)) for k, m in enumerate(got))
            result.put(bool(ok))
        else:
            assert not got
        # a process group of another size is refused
        if world > 1:
            with pytest.raises(ValueError):
                pdist.StripExchange(shape, pdist.strip_bounds(shape[1], world + 1), 0, world + 1,
                                    "cpu", mode)
        # max-over-ranks timing reduction used by bench.py
        assert pdist.max_over_ranks(float(rank + 1)) == world
        # what the bench line reports about the process group
        seen = pdist.describe_job("cpu", "cpu")
        assert seen["world_size"] == world and seen["backend"] == "gloo"
        assert seen["allreduce_checksum"] == seen["allreduce_expected"] == world * (world + 1) // 2
        assert seen["devices"] == ["cpu"] * world
        # the bench's pre-flight: who is here (logged before any collective of its own), then one
        # gather and one uint8 sum-reduce of a small buffer, checked on rank 0
        said = []
        pre = pdist.preflight("cpu", "cpu", log=said.append if rank == 0 else None, nbytes=4099)
        assert pre["world_size"] == world and pre["gather_ok"] and pre["reduce_ok"]
        if rank == 0:
            assert len(said) == 2 and "before first collective" in said[0] and len(said[0]) < 1024
            assert f'"world_size": {world}' in said[0] and '"backend": "gloo"' in said[0]
        # image sets of a stream, dealt out over the ranks: each set exactly once
        mine = torch.zeros(7, dtype=torch.int64)
        mine[pdist.assign_sets(7, rank, world)] = 1
        dist.all_reduce(mine)
        assert mine.tolist() == [1] * 7
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,depth,world", [("gather", 2, 2), ("reduce", 2, 2), ("gather", 1, 2),
                                              ("reduce", 1, 3), ("gather", 2, 3)])
def test_sharded_steps_over_gloo(mode, depth, world):
    """The strips step of bench.py --gpus N at world 2 and 3 (mosaic widths that do and do
    not divide evenly), both exchanges - point-to-point gather of packed strips, sum-reduce
    of zero-padded full-width mosaics - synchronous and pipelined one stitch deep: rank 0
    composes exactly the known mosaic of every step, in order."""
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    mp.spawn(_worker, args=(world, _free_port(), mode, depth, result), nprocs=world, join=True)
    assert result.get() is True


@pytest.mark.parametrize("mode,depth,world,lanes", [("gather", 2, 2, 2), ("reduce", 2, 3, 2),
                                                    ("gather", 1, 2, 2), ("reduce", 2, 2, 3),
                                                    ("gather", 1, 3, 3)])
def test_sharded_steps_with_stitches_in_flight_over_gloo(mode, depth, world, lanes):
    """The same steps with consecutive stitches alternating between several engines, each with
    exchange buffers of its own (bench.py's strips with two stitches in flight per rank): rank 0
    still composes the known mosaic of every step, in order, whatever the pipeline depth."""
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    mp.spawn(_worker, args=(world, _free_port(), mode, depth, result, False, lanes), nprocs=world,
             join=True)
    assert result.get() is True


def test_sharded_steps_with_the_cached_plan_over_gloo():
    """The same steps with the host geometry kept from stitch to stitch (bench.py's
    `plan cached` figure): every step asks the engine's cache, the mosaics are the same."""
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    mp.spawn(_worker, args=(2, _free_port(), "gather", 2, result, True), nprocs=2, join=True)
    assert result.get() is True


def test_exchange_without_a_mode_is_refused_in_step():
    shapes, rots, intrs = _scene()
    shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape
    st = pdist.ShardedStitcher(_HostEngine(shape), shapes, rots, intrs, 5, 0, 2, exchange=None)
    with pytest.raises(RuntimeError, match="geometry only"):
        st.step(None)


def test_exchange_world_1_needs_no_process_group():
    shapes, rots, intrs = _scene()
    shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape
    for mode in pdist.StripExchange.MODES:
        eng = _HostEngine(shape)
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, 5, 0, 1, exchange=mode, depth=2)
        assert st.step(None)[1] is None
        assert torch.equal(st.step(None)[1], eng.truth(0))
        assert torch.equal(st.finish(), eng.truth(1))


def test_trusted_layouts_and_kept_geometry_reach_the_lanes_engines():
    """ShardedStitcher switches its lanes' engines to trusted stitches when the plan comes out of
    the memo, hands `keep_geometry` through, and checks a trusted lane's layout when it collects
    that lane's mosaic (Engine.trust_layouts / verify_trusted; the GPU side:
    test_kept_geometry_stitches_equal_waiting_ones)."""
    shapes, rots, intrs = _scene()
    shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape

    class Trusting(_HostEngine):
        def trust_layouts(self, on=True, keep_geometry=False):
            self.trust_layout, self.keep_geometry = bool(on), bool(on and keep_geometry)
            self.calls = getattr(self, "calls", []) + [(bool(on), bool(keep_geometry))]

        def verify_trusted(self):
            self.verified = getattr(self, "verified", 0) + 1

    for cache_plan, keep, want in [(True, True, (True, True)), (True, False, (True, False)),
                                   (False, True, (False, False))]:
        eng = Trusting(shape)
        lanes = [eng, Trusting(shape, eng.clock)]
        st = pdist.ShardedStitcher(lanes, shapes, rots, intrs, 5, 0, 1, exchange="gather", depth=2,
                                   cache_plan=cache_plan, keep_geometry=keep)
        assert all((e.trust_layout, e.keep_geometry) == want for e in lanes)
        got = [m.clone() for m in (st.step(None)[1] for _ in range(4)) if m is not None]
        got.append(st.finish().clone())
        assert all(torch.equal(m, eng.truth(k)) for k, m in enumerate(got))
        assert sum(getattr(e, "verified", 0) for e in lanes) == (len(got) if cache_plan else 0)
        # close() hands the engines back as they came: untrusted (a later direct use of the engine
        # has nobody calling verify_trusted)
        st.close()
        assert all((e.trust_layout, e.keep_geometry) == (False, False) for e in lanes)


def test_balanced_strip_bounds_cut_equal_work():
    """Strips of equal work: the cuts sit at the quantiles of the column costs (aligned, no strip
    thinner than the minimum), cover the mosaic, and fall back to equal widths when they cannot."""
    rng = np.random.default_rng(3)
    for world in (2, 3, 4, 8):
        cost = rng.random(13760) + np.where((np.arange(13760) > 3000) & (np.arange(13760) < 11000), 3.0, 0.0)
        b = pdist.balanced_strip_bounds(cost, world)
        assert b[0] == 0 and b[-1] == 13760 and len(b) == world + 1
        assert all(b[r + 1] - b[r] >= 64 for r in range(world)) and all(v % 8 == 0 for v in b[1:-1])
        work = [cost[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(work) / (cost.sum() / world) < 1.01          # (one aligned column either way)
        even = pdist.strip_bounds(13760, world)
        assert max(cost[even[r]:even[r + 1]].sum() for r in range(world)) >= max(work)
    # all the cost in a few columns: the minimum width holds, the order too
    b = pdist.balanced_strip_bounds(np.r_[np.zeros(5000), np.ones(100), np.zeros(5000)], 4)
    assert b == sorted(b) and b[0] == 0 and b[-1] == 10100 and min(np.diff(b)) >= 64
    # too narrow, no cost, nonsense: equal widths
    assert pdist.balanced_strip_bounds(np.ones(100), 8) == pdist.strip_bounds(100, 8)
    assert pdist.balanced_strip_bounds(np.zeros(4096), 4) == pdist.strip_bounds(4096, 4)
    assert pdist.balanced_strip_bounds(np.full(4096, np.nan), 4) == pdist.strip_bounds(4096, 4)


def _balanced_worker(rank, world, port, result):
    """The ranks of a process group take rank 0's cut, whatever their own engine computed."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes, rots, intrs = _scene()
        shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape

        class Costed(_HostEngine):
            def column_costs(self, plan, n_levels):
                cost = np.ones(plan.shape[1])
                cost[:plan.shape[1] // 3] += 5.0 + rank       # (a rank-dependent answer on purpose)
                return cost

        eng = Costed(shape)
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, 5, rank, world, exchange="gather", depth=2)
        want = pdist.balanced_strip_bounds(np.where(np.arange(shape[1]) < shape[1] // 3, 6.0, 1.0), world)
        ok = st.balanced and st.bounds == want and st.bounds != pdist.strip_bounds(shape[1], world)
        got = []
        for _ in range(4):
            previous = st.step(frames=None)[1]
            if previous is not None:
                got.append(previous.clone())
        last = st.finish()
        if rank == 0:
            got.append(last.clone())
            ok = ok and len(got) == 4 and all(torch.equal(m, eng.truth(k)) for k, m in enumerate(got))
        gathered = [None] * world
        dist.all_gather_object(gathered, bool(ok))
        if rank == 0:
            result.put(all(gathered))
    finally:
        dist.destroy_process_group()


def test_balanced_strips_over_gloo_take_rank_0s_cut():
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    mp.spawn(_balanced_worker, args=(2, _free_port(), result), nprocs=2, join=True)
    assert result.get() is True


def _forced_world1_worker(rank, world, port, result):
    """One rank, a real (gloo) process group, the collectives FORCED: the very calls of a world-N
    run - `dist.gather` on views of one buffer / `dist.reduce(uint8, SUM)`, `async_op=True`,
    `work.wait()` - with two lanes on communicators of their own, then `close()`."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes, rots, intrs = _scene()
        shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape
        ok = True
        for mode, lane_groups in [(m, g) for m in pdist.StripExchange.MODES for g in ("own", "shared")]:
            eng = _HostEngine(shape)
            st = pdist.ShardedStitcher([eng, _HostEngine(shape, eng.clock)], shapes, rots, intrs, 5,
                                       0, 1, exchange=mode, depth=2, force_collective=True,
                                       lane_groups=lane_groups)
            assert all(ex.collective for _, _, ex in st.lanes)
            assert len(st._lane_groups) == (2 if lane_groups == "own" else 0)
            assert all((ex.group is not None) == (lane_groups == "own") for _, _, ex in st.lanes)
            got = []
            for _ in range(6):
                previous = st.step(None)[1]
                if previous is not None:
                    got.append(previous.clone())
            got.append(st.finish().clone())
            ok &= len(got) == 6 and all(torch.equal(m, eng.truth(k)) for k, m in enumerate(got))
            st.close()
            assert not st._lane_groups
            with pytest.raises(RuntimeError):
                st.step(None)
        result.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_forced_collectives_at_world_1_and_close():
    """`force_collective=True` makes a one-rank run issue the world-N collectives (what the GPU
    test does on RCCL, tests/test_a_rccl_world1.py); `close()` gives the lane communicators back."""
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    mp.spawn(_forced_world1_worker, args=(1, _free_port(), result), nprocs=1, join=True)
    assert result.get() is True


def test_strip_bounds_cover_the_mosaic():
    for width in (1, 7, 64, 13760):
        for world in (1, 2, 3, 8):
            b = pdist.strip_bounds(width, world)
            assert b[0] == 0 and b[-1] == width and len(b) == world + 1
            assert all(x <= y for x, y in zip(b[:-1], b[1:]))
            if width >= world:
                widths = [y - x for x, y in zip(b[:-1], b[1:])]
                assert max(widths) - min(widths) <= 1


def test_frames_for_strip_is_a_superset_of_what_windows_touch():
    """Every frame whose patch can contribute to a strip (its rectangle reaches
    the strip grown by the radius) is selected, for every rank; together the
    ranks hold every frame."""
    cfg = synth.CONFIGS["cfg3"]
    rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                     sweep_deg=cfg["sweep_deg"])
    shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    radius = 43
    for world in (2, 8):
        bounds = pdist.strip_bounds(plan.shape[1], world)
        seen = set()
        for r in range(world):
            strip = (bounds[r], bounds[r + 1])
            got = pdist.frames_for_strip(plan.rects, strip, 2 * radius)
            for i, (_, _, x0, x1) in enumerate(plan.rects):
                touches = x0 < strip[1] + radius and x1 > strip[0] - radius
                assert (i in got) or not touches
            assert got == list(range(got[0], got[-1] + 1))      # a contiguous run
            assert len(got) < cfg["n"] or world == 1
            seen.update(got)
        assert seen == set(range(cfg["n"]))


def test_windows_for_strip_clipping():
    rect = (0, 100, 1000, 1400)                     # patch columns 1000..1399
    box = (0, 99, 1100, 1199)                       # owns columns 1100..1199
    area, win = engine.windows_for(box, rect, 43)
    assert area == (0, 100, 57, 243)
    area, win = engine.windows_for(box, rect, 43, strip=(1150, 1300))
    # V: A grown by the radius, its columns rounded outwards to multiples of 4
    assert area == (0, 100, 150, 243) and win[2] == (150 - 43) & ~3 and win[3] == (243 + 43 + 3) & ~3
    assert engine.windows_for(box, rect, 43, strip=(1243, 2000)) is None
    assert engine.windows_for(box, rect, 43, strip=(0, 1057)) is None
    area, _ = engine.windows_for(box, rect, 43, strip=(0, 1058))
    assert area == (0, 100, 57, 58)


def test_assign_sets_round_robin():
    for n_sets in (0, 1, 8, 13):
        for world in (1, 2, 8):
            got = [pdist.assign_sets(n_sets, r, world) for r in range(world)]
            assert sorted(i for g in got for i in g) == list(range(n_sets))
            assert max(map(len, got)) - min(map(len, got)) <= 1
    assert pdist.max_over_ranks(1.25) == 1.25          # no process group: identity


def _skewed_lanes_worker(rank, world, port, mode, result):
    """Two lanes, each driven by a thread of its own, the threads skewed against each other
    differently on every rank: rank r starts lane (r % 2) first and the other lane 0.3 s later,
    so the ranks' submissions reach the backend in opposite lane orders."""
    import threading
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes, rots, intrs = _scene()
        shape = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape
        eng = _HostEngine(shape)
        st = pdist.ShardedStitcher([eng, _HostEngine(shape, eng.clock)], shapes, rots, intrs, 5,
                                   rank, world, exchange=mode, depth=2, lane_groups="own")
        groups = [ex.group for _, _, ex in st.lanes]
        assert groups[0] is not groups[1] and None not in groups       # a communicator per lane
        truth = eng.truth
        got = {0: [], 1: []}
        errors = []

        def drive(lane, delay):
            try:
                time.sleep(delay)
                ex = st.lanes[lane][2]
                c0, c1 = ex.strip
                for k in range(3):
                    ex.recycle()
                    ex.target()[:, c0:c1] = truth(10 * lane + k)[:, c0:c1]
                    ex.submit()
                    mosaic = ex.collect()
                    if mosaic is not None:
                        got[lane].append(mosaic.clone())
                    time.sleep(0.05 * ((rank + lane) % 2))
            except Exception as err:       # noqa: BLE001 - reported through the queue
                errors.append(repr(err))
        first = rank % 2
        threads = [threading.Thread(target=drive, args=(first, 0.0)),
                   threading.Thread(target=drive, args=(1 - first, 0.3))]
        for th in threads:
            th.start()
        for th in threads:
            th.join(60)
        assert not errors and not any(th.is_alive() for th in threads), errors
        if rank == 0:
            ok = all(len(got[lane]) == 3 and
                     all(torch.equal(m, truth(10 * lane + k)) for k, m in enumerate(got[lane]))
                     for lane in (0, 1))
            result.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("gather", 2), ("reduce", 2), ("gather", 3)])
def test_lanes_exchange_independently_when_driven_from_skewed_threads(mode, world):
    """Each lane of a ShardedStitcher exchanges over a communicator of its own, so the order in
    which different ranks issue lane 0's and lane 1's collectives does not matter: here every
    rank drives its two lanes from two threads, and odd ranks start the lanes in the opposite
    order of even ranks.  (On one shared communicator this pattern pairs lane 0's gather on one
    rank with lane 1's on another - or waits for ever.)"""
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    mp.spawn(_skewed_lanes_worker, args=(world, _free_port(), mode, result), nprocs=world, join=True)
    assert result.get() is True
