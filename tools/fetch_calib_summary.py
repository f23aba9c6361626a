#!/usr/bin/env python3
"""known bytes / counted bytes per access shape (tools/fetch_calib.sh) -> fetch_calib.json"""
import collections
import csv
import glob
import json
import os
import sys


def main(root):
    known = json.load(open(os.path.join(root, "known.json")))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fid:
            for row in csv.DictReader(fid):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {"footprint_bytes": known["footprint_bytes"], "unit": "known bytes / (counter x 1024)", "shapes": {}}
    print("%-16s %14s %14s %10s %10s %8s  requests" % ("shape", "read B", "write B", "rd factor", "wr factor", "GB/s"))
    for name, k in known["kernels"].items():
        c = {cn: sum(v) / len(v) for cn, v in acc.get(name, {}).items()}
        f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        row = dict(k)
        row["FETCH_SIZE_KiB"] = f
        row["WRITE_SIZE_KiB"] = w
        row["read_factor"] = k["read_bytes"] / (f * 1024) if f and k["read_bytes"] else None
        row["write_factor"] = k["write_bytes"] / (w * 1024) if w and k["write_bytes"] else None
        for cn in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum",
                   "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
            if cn in c:
                row[cn] = c[cn]
        if k["read_bytes"] and "TCC_EA0_RDREQ_sum" in c and c["TCC_EA0_RDREQ_sum"]:
            row["bytes_per_rdreq"] = k["read_bytes"] / c["TCC_EA0_RDREQ_sum"]
        if k["read_bytes"] and "TCC_EA0_RDREQ_128B_sum" in c:
            # the memory-side read requests by size: what the L2 really asked the fabric for
            sized = (32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0)
                     + 128 * c["TCC_EA0_RDREQ_128B_sum"])
            row["sized_request_bytes"] = sized
            row["bytes_per_sized"] = k["read_bytes"] / sized if sized else None
        if k["write_bytes"] and "TCC_EA0_WRREQ_sum" in c and c["TCC_EA0_WRREQ_sum"]:
            row["bytes_per_wrreq"] = k["write_bytes"] / c["TCC_EA0_WRREQ_sum"]
        out["shapes"][name.replace("calib_", "")] = row
        print("%-16s %14.0f %14.0f %10s %10s %8.0f  %s" % (
            name, k["read_bytes"], k["write_bytes"],
            "%.3f" % row["read_factor"] if row["read_factor"] else "-",
            "%.3f" % row["write_factor"] if row["write_factor"] else "-", k["GBps"],
            {kk: round(vv, 1) for kk, vv in row.items() if kk.startswith("bytes_per")}))
    json.dump(out, open(os.path.join(root, "fetch_calib.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
