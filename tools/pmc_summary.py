#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc CSVs written by tools/pmc.sh, and
pmc_traffic.json: HBM bytes per launch.

Reads: the L2's memory-side read requests by size, 32 x TCC_EA0_RDREQ_32B + 64 x
TCC_EA0_RDREQ_64B + 128 x TCC_EA0_RDREQ_128B (the `rdreq` pass).  FETCH_SIZE tallies every
request at 64 B (MI355X_MICROARCH.md, "HBM"), so it reads half of a 128-B request; the
calibration of profiles/r04/fetch_calib.json (tools/fetch_calib.sh: kernels that read a known
1 GiB once) shows the sized requests give the known bytes exactly for 16-, 4-, 2- and 1-byte
per lane streams alike, where FETCH_SIZE x 2 is right only as long as every request is a
128-B one.  Each kernel's ratio sized / FETCH_SIZE is kept as `fetch_size_factor`.
Without the `rdreq` pass (older visits) the figure falls back to FETCH_SIZE x 2.
Writes: WRITE_SIZE x 1024 (exact for every store shape of the calibration)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def main(root, workload="cfg3", by_grid=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fid:
            for row in csv.DictReader(fid):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                name = re.sub(r"<.*", "", name)[:40]
                if by_grid:                      # kernels launched at many sizes: one line per size
                    tmpl = re.search(r"<([^>]*)>", row["Kernel_Name"])
                    name += f"<{tmpl.group(1)}>" if tmpl else ""
                    name += f" grid {row.get('Grid_Size', '?')}"
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    traffic, detail = {}, {}
    for name in sorted(acc):
        print(f"== {name}  (dispatches: {max(len(v) for v in acc[name].values())})")
        for c in counters:
            vals = acc[name].get(c)
            if vals:
                print(f"   {c:24s} avg {sum(vals) / len(vals):16.1f}   sum {sum(vals):18.1f}")
        f, w = acc[name].get("FETCH_SIZE"), acc[name].get("WRITE_SIZE")
        avg = lambda c: (sum(acc[name][c]) / len(acc[name][c])) if acc[name].get(c) else 0.0
        if f and w:
            fetch = sum(f) / len(f) * 1024.0
            if acc[name].get("TCC_EA0_RDREQ_128B_sum") or acc[name].get("TCC_EA0_RDREQ_64B_sum"):
                read = (32.0 * avg("TCC_EA0_RDREQ_32B_sum") + 64.0 * avg("TCC_EA0_RDREQ_64B_sum")
                        + 128.0 * avg("TCC_EA0_RDREQ_128B_sum"))
                how = "sized requests"
            else:
                read, how = 2.0 * fetch, "FETCH_SIZE x 2"
            write = sum(w) / len(w) * 1024.0
            traffic[name] = read + write
            detail[name] = dict(read_bytes=read, write_bytes=write, read_from=how,
                                fetch_size_bytes=fetch,
                                fetch_size_factor=read / fetch if fetch else None)
    with open(os.path.join(root, "pmc_traffic.json"), "w") as fid:
        json.dump({"command": "bench.py (see tools/pmc.sh)", "workload": workload,
                   "unit": "bytes per launch",
                   "formula": "reads: 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B (L2 "
                              "memory-side requests by size; falls back to FETCH_SIZE x 2 x 1024 "
                              "without that pass); writes: WRITE_SIZE x 1024",
                   "bytes_per_launch": traffic, "detail": detail}, fid, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
