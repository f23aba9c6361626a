#!/usr/bin/env python3
"""How the kernels of the stitches in flight share the GPU: from a rocprofv3 --kernel-trace CSV of
bench.py, over the TIMED steps (the K stitches in front of the last K, which are the instrumented
one-at-a-time pass).

    tools/trace_overlap.py KERNEL_TRACE.csv --steps K

Prints the window's length, the time at least one / exactly one / two and more kernels ran, per
kernel its mean duration in the window against its mean in the one-at-a-time pass (the stretch),
and for the large kernels the time each pair of them ran together."""
import argparse
import csv
import re
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--once", default="multiband_compose_kernel")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as fid:
        for r in csv.DictReader(fid):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            name = re.sub(r"\s+", " ", name).split("<")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    once = [r for r in rows if r[2].startswith(args.once)]
    k = args.steps
    assert len(once) >= 2 * k, (len(once), k)
    solo_from = once[-k][0] - 1                 # the instrumented pass starts before its first collapse...
    # ...with the kernels of that stitch in front of it: take the pass from the end of the collapse before
    t_solo = once[-k - 1][1]
    t_win0, t_win1 = once[-2 * k - 1][1], once[-k - 1][1]
    win = [r for r in rows if r[0] >= t_win0 and r[1] <= t_win1]
    solo = [r for r in rows if r[0] >= t_solo]
    length = t_win1 - t_win0
    # sweep
    ev = []
    for a, b, n in win:
        ev.append((a, 1, n))
        ev.append((b, -1, n))
    ev.sort(key=lambda e: (e[0], e[1]))
    running = defaultdict(int)
    last = t_win0
    depth_time = defaultdict(int)
    pair = defaultdict(int)
    alone = defaultdict(int)
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            names = sorted(x for x, c in running.items() if c > 0)
            depth = sum(running.values())
            depth_time[min(depth, 3)] += dt
            if depth == 1:
                alone[names[0]] += dt
            for i, x in enumerate(names):
                for y in names[i + (0 if running[x] > 1 else 1):]:
                    pair[(x, y)] += dt
        running[n] += d
        last = t
    print("window %.3f ms for %d stitches = %.4f ms per stitch" % (length / 1e6, k, length / 1e6 / k))
    print("idle %.1f %%   one kernel %.1f %%   two %.1f %%   three and more %.1f %%" % tuple(
        100.0 * depth_time[d] / length for d in (0, 1, 2, 3)))
    def mean(rs, n):
        v = [b - a for a, b, m in rs if m == n]
        return (sum(v) / len(v) / 1e3, len(v)) if v else (0.0, 0)
    names = sorted({r[2] for r in win}, key=lambda n: -mean(win, n)[0] * mean(win, n)[1])
    print("%-28s %10s %10s %8s %12s" % ("kernel", "in flight", "alone", "stretch", "ms / stitch"))
    for n in names[:10]:
        a, ca = mean(win, n)
        b, cb = mean(solo, n)
        print("%-28s %8.1f us %8.1f us %8.2f %12.4f" % (n[:28], a, b, a / b if b else 0, a * ca / 1e3 / k))
    big = names[:4]
    print("time two of the large kernels ran together (ms per stitch):")
    for i, x in enumerate(big):
        for y in big[i:]:
            key = (x, y) if (x, y) in pair else (y, x)
            print("  %-26s + %-26s %.4f" % (x[:26], y[:26], pair.get(key, 0) / 1e6 / k))
    print("time a large kernel ran alone (ms per stitch):", {n[:20]: round(alone[n] / 1e6 / k, 4) for n in big})


if __name__ == "__main__":
    main()
