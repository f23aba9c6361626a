#!/usr/bin/env python3
"""What one rank of an N-GPU strips run costs per stitch when only its strip's kernels run:
rank r of world N emulated on this one GPU (no exchange).  The time that does not shrink with
N - the plan, the launches, the one synchronisation - is the floor of strong scaling.
    python tools/strip_floor.py [cfg3] [world ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import dist as pdist  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

import json  # noqa: E402
argv = [a for a in sys.argv[1:] if not a.startswith("--json=")]
JSON_OUT = next((a[7:] for a in sys.argv[1:] if a.startswith("--json=")), None)
name = argv[0] if argv else "cfg3"
worlds = [int(v) for v in argv[1:]] or [1, 2, 4, 8]
ROWS = []
cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine()
LANES = [(eng, torch.cuda.current_stream(eng.device))]
for _ in range(max(1, int(os.environ.get("PANO_SETS_IN_FLIGHT", "1"))) - 1):
    _s = torch.cuda.Stream(eng.device)
    with torch.cuda.stream(_s):
        LANES.append((engine.Engine(eng.device), _s))
if os.environ.get("PANO_PLAN_CACHED", "0") != "0" and os.environ.get("PANO_TRUST_LAYOUT", "1") != "0":
    for _e, _ in LANES:      # as ShardedStitcher does with the plan out of the memo
        _e.trust_layouts(True, keep_geometry=os.environ.get("PANO_KEEP_GEOMETRY", "0") != "0")
BALANCE = os.environ.get("PANO_STRIP_BALANCE", "1") != "0"     # strips of equal work (default) / width
COUNT = [0, False]
pool = {}
PER_RANK = {}
import gc  # noqa: E402
gc.collect()
gc.freeze()          # the cyclic collector's 37 ms pause would land in one of the 20 timed steps
DISTINCT = int(os.environ.get("PANO_DISTINCT_FRAMES", "0"))    # cfg5: cycle a few 8K frames
for world in worlds:
    worst = (0.0, None)
    # every rank of the world (up to 8; beyond that the first, the middle and the last)
    for rank in ((range(world) if world <= 8 else sorted({0, world // 2, world - 1}))
                 if not os.environ.get("PANO_STRIP_RANK") else [int(os.environ["PANO_STRIP_RANK"])]):
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, cfg["n_levels"], rank, world,
                                   exchange=None, balance=BALANCE)
        for i in st.my_frames:
            k = i % DISTINCT if DISTINCT else i
            if k not in pool:
                pool[k] = eng.upload_frames([synth.make_frame(k, cfg["width"], cfg["height"], "A")])[0]
        frames = [pool[i % DISTINCT if DISTINCT else i] for i in st.my_frames]
        out = torch.zeros(engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape + (3,),
                          dtype=torch.uint8, device=eng.device)

        # PANO_SETS_IN_FLIGHT=2: consecutive stitches alternate between two engines / streams
        # (timing(True), the per-kernel pass, goes back to one)
        def step():
            lane = COUNT[0] % len(LANES) if not COUNT[1] else 0
            use, stream = LANES[lane]
            COUNT[0] += 1
            # max(2, lanes) stitches in flight, not more: the oldest is waited for before another is
            # queued (what collecting its mosaic does in ShardedStitcher.step, depth = max(2, lanes));
            # a host that queues a trusted stitch in 0.05 ms would otherwise run twenty ahead
            t_wait = time.perf_counter()
            while len(DONE) >= max(2, len(LANES)):
                DONE.pop(0).synchronize()
            WAITED[0] += time.perf_counter() - t_wait
            with torch.cuda.stream(stream):
                if os.environ.get("PANO_PLAN_CACHED", "0") != "0":
                    plan = use.cached_plan(shapes, rots, intrs, True, 10 ** 9, st.table_cols)
                else:
                    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9, table_cols=st.table_cols)
                    use.upload_plan(plan)
                use.multiband_fused(frames, plan, cfg["n_levels"], frame_ids=st.my_frames,
                                    strip=st.strip, mosaic_out=OUTS[(COUNT[0] - 1) % len(LANES)] if not COUNT[1] else out)
                DONE.append(torch.cuda.Event())
                DONE[-1].record(stream)
        OUTS = [out] + [torch.zeros_like(out) for _ in LANES[1:]]
        DONE = []
        WAITED = [0.0]
        COUNT[1] = False
        for _ in range(3 * len(LANES)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        WAITED[0] = 0.0
        for _ in range(20):
            step()
        # the host's share: queueing alone (without its waits for the oldest stitch)
        host_ms = (time.perf_counter() - t0 - WAITED[0]) / 20 * 1e3
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        COUNT[1] = True
        eng.timing(True)                 # second pass: the kernels' own times (events serialise
        for _ in range(20):              # what the first pass ran side by side)
            step()
        torch.cuda.synchronize()
        times = eng.kernel_times()
        kern = sum(v[0] for v in times.values()) / 20
        eng.timing(False)
        PER_RANK.setdefault(world, []).append(round(ms, 4))
        if ms > worst[0]:
            worst = (ms, rank, kern, len(frames), {k: round(v[0] / 20, 3) for k, v in times.items()}, host_ms)
    ROWS.append(dict(world=world, rank=worst[1], ms_per_stitch=worst[0], kernel_ms=worst[2],
                     frames_resident=worst[3], kernels=worst[4], host_ms_per_stitch=worst[5],
                     ms_per_stitch_by_rank=PER_RANK[world], bounds=list(st.bounds)))
    sys.stdout.flush()
    print(f"world {world}: by rank {PER_RANK[world]}")
    print(f"world {world}: slowest of ranks sampled = rank {worst[1]}: {worst[0]:.3f} ms per stitch "
          f"(host {worst[5]:.3f} ms of it, timed kernels {worst[2]:.3f} ms, {worst[3]} frames resident) {worst[4]}")
if os.environ.get("PANO_HOST_PROFILE"):
    # where the host's share goes: the last (world, rank) again under cProfile
    import cProfile
    import pstats
    COUNT[1] = False
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
t0 = time.perf_counter()
for _ in range(50):
    engine.Plan(shapes, rots, intrs, True, 10 ** 9)
print("Plan alone: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
if JSON_OUT:
    # appended to: one visit runs this tool once per (lanes, plan) setting
    entry = dict(config=name, lanes=len(LANES), plan_cached=os.environ.get("PANO_PLAN_CACHED", "0") != "0",
                 stitch_async=os.environ.get("PANO_STITCH_ASYNC", "0"),
                 trusted_layouts=bool(LANES[0][0].trust_layout),
                 kept_geometry=bool(LANES[0][0].keep_geometry), balanced_strips=BALANCE,
                 what="rank r of world N emulated on ONE GPU: its strip's kernels only, no exchange; "
                      "wall ms per stitch of 20 stitches, slowest of all ranks (of ranks 0, N/2, N-1 above world 8)",
                 rows=ROWS)
    have = []
    if os.path.exists(JSON_OUT):
        with open(JSON_OUT) as fid:
            have = json.load(fid)
    have.append(entry)
    with open(JSON_OUT, "w") as fid:
        json.dump(have, fid, indent=1)
