#!/usr/bin/env python3
"""Steady-state per-kernel durations from a rocprofv3 --kernel-trace CSV of bench.py.

    tools/trace_stats.py KERNEL_TRACE.csv --steps K [--drop N] [--out FILE.csv]

bench.py runs setup and warm-up stitches, K timed steps (two stitches in flight on config 3:
the kernels of two streams overlap and stretch each other) and then the SAME K steps once
more, one at a time, with HIP events around every launch - the pass its `kernel_ms_per_step`
and `roofline` come from.  rocprofv3's own *_kernel_stats.csv averages all of that, cold
first-touch launches included.  This tool reports, per kernel: every dispatch; everything
after the first N (default 10) dispatches; and the LAST K steps' dispatches (= the
instrumented, one-at-a-time pass: the figure that must agree with the bench's event time)."""
import argparse
import csv
import re
import statistics
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True, help="bench.py --steps of the profiled command")
    ap.add_argument("--total-steps", type=int, default=0,
                    help="all stitches the command ran (setup + warm-up + 2 x steps); default: derived "
                         "from the dispatch count of the kernel launched once per stitch")
    ap.add_argument("--drop", type=int, default=10)
    ap.add_argument("--out")
    args = ap.parse_args()
    rows = defaultdict(list)
    with open(args.trace) as fid:
        for r in csv.DictReader(fid):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            name = re.sub(r"\s+", " ", name)
            rows[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for v in rows.values():
        v.sort()
    # stitches the command ran: the dispatch count of a once-per-stitch kernel
    once = [k for k in rows if k.startswith("multiband_compose_kernel")] or \
           [k for k in rows if k.startswith("gray_u8_kernel")]
    total = args.total_steps or (len(rows[once[0]]) if once else 0)
    out = []
    print("%-44s %6s %9s | %9s %9s %9s | %6s %9s %9s" % (
        "kernel", "calls", "avg_us", "steady_n", "avg_us", "median_us", "last_n", "avg_us", "median_us"))
    for name in sorted(rows, key=lambda k: -sum(e - s for s, e in rows[k])):
        d = [(e - s) / 1e3 for s, e in rows[name]]
        per_step = len(d) / total if total else 0
        steady = d[args.drop:] if len(d) > args.drop else d
        n_last = int(round(per_step * args.steps)) if per_step else 0
        last = d[-n_last:] if 0 < n_last <= len(d) else steady
        rec = dict(kernel=name, calls=len(d), avg_us=statistics.fmean(d),
                   steady_calls=len(steady), steady_avg_us=statistics.fmean(steady),
                   steady_median_us=statistics.median(steady),
                   last_calls=len(last), last_avg_us=statistics.fmean(last),
                   last_median_us=statistics.median(last), min_us=min(d), max_us=max(d),
                   launches_per_stitch=per_step)
        out.append(rec)
        if rec["avg_us"] * rec["calls"] > 1.0:
            print("%-44s %6d %9.1f | %9d %9.1f %9.1f | %6d %9.1f %9.1f" % (
                name[:44], rec["calls"], rec["avg_us"], rec["steady_calls"], rec["steady_avg_us"],
                rec["steady_median_us"], rec["last_calls"], rec["last_avg_us"], rec["last_median_us"]))
    if args.out:
        with open(args.out, "w", newline="") as fid:
            w = csv.DictWriter(fid, fieldnames=list(out[0]))
            w.writeheader()
            for rec in out:
                w.writerow({k: (round(v, 3) if isinstance(v, float) else v) for k, v in rec.items()})


if __name__ == "__main__":
    main()
