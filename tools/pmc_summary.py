#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc CSVs written by tools/pmc.sh."""
import collections
import csv
import glob
import os
import sys


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fid:
            for row in csv.DictReader(fid):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    for name in sorted(acc):
        print(f"== {name}  (dispatches: {max(len(v) for v in acc[name].values())})")
        for c in counters:
            vals = acc[name].get(c)
            if vals:
                print(f"   {c:24s} avg {sum(vals) / len(vals):16.1f}   sum {sum(vals):18.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
