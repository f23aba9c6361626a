#!/usr/bin/env python3
"""Headline benchmark: blended megapixels/s of warp + multiband blend.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full stitch of the workload: uint8 frames + cameras resident
in HBM -> uint8 mosaic in HBM (the boundary of the reference's own timer,
stitcher.py:441-444): host geometry, spherical warp, ownership, the L-1
Gaussian blurs per frame, band-pass collapse.  value = sum of patch pixels the
reference algorithm warps and blends ("P") / step time.

Workload (BASELINE.json): the metric is quoted on N x 4K frames, so the default
is config 3 - 32 synthetic 3840x2160 frames, 5 deg yaw steps, hfov 60 deg,
native resolution, 5 levels.  With N GPUs a step stitches N such image sets,
one per GPU (independent panoramas: weak scaling, no data-path collective);
the same launch then also times ONE image set split into column strips over
the N GPUs (strong scaling, strips gathered over RCCL) and reports it under
"strips".  ``--mode strips`` makes that the headline instead; ``--workload
cfg2`` runs the 8 x 1080p configuration.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_PEAK_TFLOPS = 157.3         # f32 vector == f32 MFMA peak on gfx950
NATIVE = 10 ** 9                # MAX_RESOLUTION that never caps


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg5", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="sets", choices=["sets", "strips"],
                    help="N > 1: 'sets' = one independent image set per GPU and step "
                         "(default, weak scaling); 'strips' = one image set, its mosaic "
                         "split into column strips (strong scaling)")
    ap.add_argument("--no-strips", action="store_true",
                    help="N > 1, mode sets: skip the secondary column-strip measurement")
    ap.add_argument("--strips-timeout", type=float, default=120.0)
    return ap.parse_args()


def workload(name):
    from pano360_amd import synth
    if name == "tiny":
        return dict(n=8, width=320, height=180, sweep_deg=140.0, n_levels=5)
    return dict(synth.CONFIGS[name])


def pmc_traffic(name, workload=None):
    """HBM bytes per launch of kernel `name` from the committed PMC summary of this
    same command (profiles/<round>/pmc_traffic.json, written from tools/pmc.sh
    output: FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes, per MI355X_MICROARCH.md)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")))[::-1]:
        with open(path) as fid:
            table = json.load(fid)
        if workload is not None and table.get("workload") != workload:
            continue                       # counters of another workload say nothing here
        if name in table.get("bytes_per_launch", {}):
            return table["bytes_per_launch"][name], os.path.relpath(path, ROOT)
    return None, None


def measured_traffic(times, steps, workload):
    """HBM bytes per step over all timed kernels, from the committed PMC summary of this
    workload (per-launch bytes x launches per step); None when there is no summary."""
    total, seen = 0.0, False
    for name, (_, launches) in times.items():
        per_launch, _ = pmc_traffic({"tile_flags_kernel": "tile_flags32_kernel"}.get(name, name),
                                    workload)
        if per_launch is not None:
            total += per_launch * launches / steps
            seen = True
    return total if seen else None


def roofline_for(times, plan, patches, n_levels, steps, px_active, workload=None):
    """Roofline entry of the kernel with the largest share of the timed region.
    Algorithmic work per launch counts the pixels that launch really produced
    (windows V / rectangles A of the patches, only the column tiles that were
    not skipped as interior - DESIGN.md §Kernels), not the reference's
    whole-patch count P."""
    from pano360_amd import engine
    name = max(times, key=lambda k: times[k][0])
    total_ms, launches = times[name]
    avg_s = total_ms / launches * 1e-3
    M = plan.shape[0] * plan.shape[1]
    win = [((p.window[1] - p.window[0]), (p.window[3] - p.window[2]),
            (p.area[1] - p.area[0]), (p.area[3] - p.area[2])) for p in patches]
    px_rows = sum(vh * aw for vh, vw, ah, aw in win)      # row pass: rows of V x cols of A
    px_cols = px_active or sum(ah * aw for vh, vw, ah, aw in win)   # column pass / gather
    px_warp = sum(vh * vw for vh, vw, ah, aw in win)      # warp: V
    taps = [engine.gaussian_ksize(s) for s in engine.level_sigmas(n_levels)]
    if name in ("blur_rows_kernel", "blur_cols_kernel"):
        # one launch = one level of one patch, 4 channels: taps FMAs per output
        px = px_rows if name == "blur_rows_kernel" else px_cols
        flop = steps * sum(2.0 * t * 4 * px for t in taps)
        achieved = flop / launches / avg_s / 1e12
        traffic, source = pmc_traffic(name, workload)
        return dict(kernel=name, bound="mfma", achieved=achieved, peak=F32_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=achieved / F32_PEAK_TFLOPS, traffic=traffic,
                    traffic_source=source,
                    note="f32 FMA on the vector ALU; gfx950 f32 MFMA peak equals the "
                         "f32 vector peak (157.3 TFLOP/s), no MFMA is issued",
                    avg_launch_ms=avg_s * 1e3, launches=launches)
    per_step = {
        # 3 float planes written + the frame bytes under the window (about 1:1 scale)
        "warp_spherical_kernel": 12.0 * px_warp + 3.0 * px_warp,
        # no pixel data read: owner (2 B) + valid (1 B) written per mosaic pixel
        "ownership_cameras_kernel": 3.0 * M,
        "owned_boxes_kernel": 2.0 * M,
        # warped colour + L-1 blurred RGBA per gathered pixel; owner/valid read, u8 out
        "multiband_compose_kernel": (12.0 + 16.0 * (n_levels - 1)) * px_cols + 6.0 * M,
        # fused row + column pass of every level on the matrix cores: colour planes (12 B)
        # and owner map (2 B) read over V, L-1 blurred RGBA copies written over the active
        # tiles; the intermediate image never reaches memory
        "blur_mfma_kernel": 14.0 * px_warp + 16.0 * (n_levels - 1) * px_cols,
    }.get(name, 0.0)
    achieved = per_step * steps / launches / avg_s / 1e9
    traffic, source = pmc_traffic(name, workload)
    out = dict(kernel=name, bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS,
               unit="GB/s", frac=achieved / HBM_PEAK_GBPS, traffic=traffic,
               traffic_source=source, avg_launch_ms=avg_s * 1e3, launches=launches)
    if name == "blur_mfma_kernel":
        flop = steps * sum(2.0 * t * 4 * (px_rows + px_cols) for t in taps)
        out["note"] = ("split-float16 Toeplitz products on the matrix cores (3 MFMAs per "
                       "float32-accurate product); %.1f TFLOP/s of useful float32-equivalent "
                       "FMA work" % (flop / launches / avg_s / 1e12))
    return out


def cpu_baseline(cfg):
    """The CPU oracle (C + OpenMP restatement of the reference path) timed on
    this host on a bounded sample: the first 4 frames of the same sweep."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pano_oracle as po
    from pano360_amd import synth
    n = min(4, cfg["n"])
    step = cfg.get("step_deg") or cfg["sweep_deg"] / (cfg["n"] - 1)
    imgs, rots, intrs = synth.make_scene(n, cfg["width"], cfg["height"], step_deg=step,
                                         seed=0, kind="A")
    plan = po.Plan([im.shape[:2] for im in imgs], rots, intrs, True, NATIVE)
    px = sum((r[1] - r[0]) * (r[3] - r[2]) for r in plan.rects)
    t0 = time.time()
    po.stitch(imgs, rots, intrs, "multiband", cfg["n_levels"], max_resolution=NATIVE)
    dt = time.time() - t0
    return dict(value=px / dt / 1e6, unit="MP/s", cores=po.max_threads(), kind="port",
                sample=f"first {n} of the {cfg['n']} frames ({cfg['width']}x{cfg['height']}, "
                       f"{step:.2f} deg/step), multiband L={cfg['n_levels']}, native "
                       f"resolution, {px / 1e6:.1f} MP of patches in {dt:.1f} s; oracle = "
                       f"oracle/pano_oracle.c (gcc -O2 -fopenmp), {os.cpu_count()} host CPUs")


def timed_steps(eng, step, steps, warmup, fence):
    """W untimed steps, then exactly K steps between two fences.  Returns
    (seconds on this rank, last step's result, per-kernel HIP-event times)."""
    for _ in range(warmup):
        step()
    fence()
    eng.timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    times = eng.kernel_times()
    eng.timing(False)
    return elapsed, result, times


def main():
    args = parse()
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for N > 1")
    # PANO_DIST_BACKEND=gloo lets several ranks share one GPU for a dry run of the
    # multi-rank path on a 1-GPU box; the real launch is one rank per GPU over RCCL
    backend = os.environ.get("PANO_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)

    cfg = workload(args.workload)
    n_levels = cfg["n_levels"]
    rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                     sweep_deg=cfg.get("sweep_deg"),
                                     step_deg=cfg.get("step_deg"))
    shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
    eng = engine.Engine(f"cuda:{local}")
    reduce_device = eng.device if backend == "nccl" else "cpu"

    def upload(set_id, which):      # one frame at a time: 120 x 8K is 12 GB
        return [eng.upload_frames([synth.make_frame(set_id * cfg["n"] + i, cfg["width"],
                                                    cfg["height"], "A")])[0] for i in which]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    strips = args.mode == "strips" and world > 1
    if strips:
        # ONE panorama (image set 0) split into column strips, one per rank; the
        # finished strips are gathered on rank 0 (strong scaling)
        runner = pdist.ShardedStitcher(eng, shapes, rots, intrs, n_levels, rank, world)
        frames = upload(0, runner.my_frames)

        def step():
            return runner.step(frames)
    else:
        # one image set per rank and step (rank r holds set r): the sets of a step
        # are independent panoramas, nothing crosses a GPU (weak scaling)
        my_set = pdist.assign_sets(world, rank, world)[0]
        frames = upload(my_set, range(cfg["n"]))

        def step():
            plan = engine.Plan(shapes, rots, intrs, True, NATIVE)
            mosaic, _, _, patches = eng.stitch(frames, plan, "multiband", n_levels)
            # keep only the window geometry: holding the arenas across steps would make
            # the allocator carve out fresh gigabytes every step
            return plan, mosaic, list(patches)

    for _ in range(3):            # setup: first-touch allocations of the workspaces
        step()
    fence()
    # Python's cyclic collector walks every live object (all of torch and numpy) when its
    # oldest generation comes due - 37 ms, ten stitches, in the middle of a timed step.
    # Everything alive after setup is long-lived: park it where the collector does not look.
    import gc
    gc.collect()
    gc.freeze()
    elapsed, (plan, mosaic, patches), times = timed_steps(eng, step, args.steps, args.warmup,
                                                          fence)
    elapsed = pdist.max_over_ranks(elapsed, reduce_device)
    sets_per_step = 1 if strips else world

    out = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        P, M = plan.patch_pixels, plan.shape[0] * plan.shape[1]
        S = cfg["n"] * cfg["width"] * cfg["height"]
        algo_bytes = sets_per_step * (3.0 * S + (33 + 64 * n_levels) * P
                                      + (16 * n_levels + 3) * M)
        if strips:
            how = (f"one image set per step, its mosaic split into {world} column strips "
                   f"(one per GPU), finished uint8 strips gathered on rank 0 over RCCL")
        elif world > 1:
            how = (f"{world} independent image sets per step, one per GPU, no data-path "
                   f"collective (the column-strip split of ONE mosaic is timed under 'strips')")
        else:
            how = "one image set per step on one GPU"
        out = {
            "metric": "blended megapixels/sec (multiband)",
            "value": sets_per_step * P / (ms * 1e-3) / 1e6,
            "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong" if strips else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {cfg['n']} synthetic {cfg['width']}x"
                            f"{cfg['height']} frames per image set, spherical warp + "
                            f"multiband blend L={n_levels}, native resolution "
                            f"(MAX_RESOLUTION uncapped)",
                "image_sets_per_step": sets_per_step,
                "frames": cfg["n"], "mosaic": list(plan.shape),
                "patch_megapixels": P / 1e6, "source_megapixels": S / 1e6,
                "mosaic_megapixels": M / 1e6,
                "parallelism": how,
            },
            "pipeline": {
                "algorithmic_GB_per_step": algo_bytes / 1e9,
                "algorithmic_GBps": algo_bytes / (ms * 1e-3) / 1e9,
                "frac_of_hbm_peak_all_gpus": algo_bytes / (ms * 1e-3) / 1e9
                                             / (HBM_PEAK_GBPS * world),
                "input_MPps": sets_per_step * S / (ms * 1e-3) / 1e6,
                "mosaic_MPps": sets_per_step * M / (ms * 1e-3) / 1e6,
            },
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in sorted(times.items())},
            "hbm_traffic": (lambda b: None if b is None else {
                "GB_per_step_per_gpu": b / 1e9,
                "GBps_per_gpu": b / (ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "source": "profiles/*/pmc_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE per launch) "
                          "x launches per step, this rank's kernels"})(
                measured_traffic(times, args.steps, args.workload)),
            "roofline": roofline_for(times, plan, patches, n_levels, args.steps,
                                     eng.active_tile_pixels(), args.workload),
            "active_megapixels": {
                "warped": sum((p.window[1] - p.window[0]) * (p.window[3] - p.window[2])
                              for p in patches) / 1e6,
                "blurred": sum((p.area[1] - p.area[0]) * (p.area[3] - p.area[2])
                               for p in patches) / 1e6},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg)

    if world > 1 and not strips and not args.no_strips:
        # Secondary measurement, same launch: ONE panorama over all GPUs (latency mode).
        # It is the only part of this program with a data exchange, so a watchdog makes
        # sure the headline line above is printed whatever happens here.
        import threading

        def give_up():
            if rank == 0:
                out["strips"] = {"error": f"no result within {args.strips_timeout} s"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(args.strips_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            runner = pdist.ShardedStitcher(eng, shapes, rots, intrs, n_levels, rank, world)
            sframes = upload(0, runner.my_frames)
            for _ in range(3):
                runner.step(sframes)
            fence()
            s_elapsed, (splan, _, _), s_times = timed_steps(
                eng, lambda: runner.step(sframes), args.steps, args.warmup, fence)
            s_elapsed = pdist.max_over_ranks(s_elapsed, reduce_device)
            if rank == 0:
                s_ms = s_elapsed / args.steps * 1e3
                out["strips"] = {
                    "what": f"one {cfg['n']}-frame image set per step, mosaic split into "
                            f"{world} column strips, uint8 strips gathered on rank 0 "
                            f"({backend}); strong scaling",
                    "ms_per_step": s_ms,
                    "value": splan.patch_pixels / (s_ms * 1e-3) / 1e6, "unit": "MP/s",
                    "frames_on_rank0": len(runner.my_frames),
                    "kernel_ms_per_step_rank0": {k: v[0] / args.steps
                                                 for k, v in sorted(s_times.items())}}
        except Exception as err:       # noqa: BLE001 - reported, the headline stands
            if rank == 0:
                out["strips"] = {"error": repr(err)[:300]}
        dog.cancel()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
