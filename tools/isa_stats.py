#!/usr/bin/env python3
"""Static instruction statistics of one kernel of a gfx950 assembly listing (hipcc -S).

    tools/isa_stats.py FILE.s KERNEL_SUBSTRING [--blocks]

Prints the kernel's register metadata, the instruction mix of the whole body and of the basic
blocks that hold matrix products (the step loops), so that SGPR spill traffic (v_readlane /
v_writelane) inside the loops can be told from spill traffic in prologues."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "readlane" if op.startswith("v_readlane") else "readfirstlane"
    if op.startswith("v_writelane"):
        return "writelane"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^[A-Za-z_][\w.$]*:", l) and key in l and not l.startswith(".L"):
            start = i
        elif start is not None and l.strip().startswith("s_endpgm"):
            end = i
            break
    if start is None:
        sys.exit("kernel not found")
    name = lines[start].split(":")[0]
    meta = {}
    inmeta = False
    for l in lines:
        if ".name:" in l:
            inmeta = name in l
        m = re.match(r"\s*-?\s*\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|agpr_count|group_segment_fixed_size|private_segment_fixed_size):\s*(\d+)", l)
        if m:
            last = (m.group(1), int(m.group(2)))
            meta.setdefault("_pending", []).append(last)
        if ".name:" in l:
            if inmeta:
                pass
    # simpler: the metadata block that contains the name
    text = "\n".join(lines)
    for blk in text.split("  - .agpr_count:")[1:]:
        if name in blk:
            blk = ".agpr_count:" + blk
            meta = dict(re.findall(r"\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|agpr_count|private_segment_fixed_size):\s*(\d+)", blk))
            break
    print("kernel", name)
    print("meta", meta)
    blocks = []
    cur = ["<entry>", Counter(), start]
    labels = {}
    for i in range(start + 1, end + 1):
        l = lines[i].strip()
        if not l or l.startswith(";") or l.startswith("."):
            m = re.match(r"^(\.LBB[\w]+):", l)
            if m:
                blocks.append(cur)
                cur = [m.group(1), Counter(), i]
                labels[m.group(1)] = len(blocks)
            continue
        op = l.split()[0]
        cur[1][classify(op)] += 1
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            cur.append(l.split()[-1])
    blocks.append(cur)
    total = Counter()
    for b in blocks:
        total.update(b[1])
    n = sum(total.values())
    print("body: %d instructions" % n, dict(total))
    nm = total["mfma"]
    if nm:
        non = n - nm
        print("static non-MFMA per MFMA: %.2f   readlane+writelane per MFMA: %.3f" % (non / nm, (total["readlane"] + total["writelane"]) / nm))
    hot = Counter()
    for b in blocks:
        if b[1]["mfma"]:
            hot.update(b[1])
    hn = sum(hot.values())
    print("blocks with MFMA: %d instructions" % hn, dict(hot))
    if hot["mfma"]:
        print("  non-MFMA per MFMA there: %.2f   readlane+writelane per MFMA: %.3f" % ((hn - hot["mfma"]) / hot["mfma"], (hot["readlane"] + hot["writelane"]) / hot["mfma"]))
    # loops: a block range closed by a backward branch; report those holding MFMAs
    idx = {b[0]: k for k, b in enumerate(blocks)}
    loops = []
    for k, b in enumerate(blocks):
        for tgt in b[3:]:
            if tgt in idx and idx[tgt] <= k:
                c = Counter()
                for bb in blocks[idx[tgt]:k + 1]:
                    c.update(bb[1])
                if c["mfma"]:
                    loops.append((idx[tgt], k, c))
    # innermost only: drop loops that strictly contain another
    inner = [l for l in loops if not any((o[0] >= l[0] and o[1] <= l[1] and o != l) for o in loops)]
    print("innermost loops with MFMA: %d" % len(inner))
    for a, z, c in inner:
        t = sum(c.values())
        print("  blocks %d-%d (%s): %d instr, mfma %d, valu %d, salu %d, lds %d, vmem %d, readlane %d, writelane %d, s_nop %d, waitcnt %d, branch %d -> non-MFMA/MFMA %.2f" % (
            a, z, blocks[a][0], t, c["mfma"], c["valu"], c["salu"], c["lds"], c["vmem"], c["readlane"], c["writelane"], c["s_nop"], c["waitcnt"], c["branch"], (t - c["mfma"]) / c["mfma"]))
    if show_blocks:
        for b in blocks:
            if b[1]["mfma"]:
                print(b[0], dict(b[1]))


if __name__ == "__main__":
    main()
