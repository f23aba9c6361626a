#!/usr/bin/env python3
"""Instruction breakdown of warp_windows_kernel's fast path (every tap inside the frame) from a
`hipcc -S` listing:   hipcc ... -S --cuda-device-only -o warp.s csrc/warp.hip; tools/warp_isa.py warp.s
The fast path = the head (inverse map and tap selection of the thread's four rows), the block
that loads and unpacks the taps, the block that looks the colours up, interpolates and stores."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
a = next(i for i, l in enumerate(lines) if l.startswith("_Z19warp_windows_kernel"))
b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[a:b]
blocks, cur = [], ["entry", []]
for l in body:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = [t.split(":")[0], []]
    elif t and not t.startswith((";", ".")) and not t.endswith(":"):
        cur[1].append(t.split()[0])
blocks.append(cur)


def kind(op):
    table = [
        ("f64 chain (K R ray)", ("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_cvt_f64")),
        ("f64 -> f32", ("v_cvt_f32_f64",)),
        ("two IEEE divisions", ("v_div_", "v_rcp_f32", "v_fma_f32", "v_fmac_f32")),
        ("cvRound, >> 5, saturate, fractions", ("v_rndne", "v_cvt_i32_f32", "v_med3", "v_ashrrev", "v_cndmask", "v_cmp_lt_f32", "v_cmp_gt_f32", "v_cmp_le_f32", "v_cmp_nlt", "v_cmp_ngt", "v_bfrev")),
        ("byte -> table offset (SDWA)", ("v_lshlrev_b32_sdwa", "v_and_b32_sdwa")),
        ("LDS table look-ups", ("ds_read",)),
        ("memory", ("global_", "buffer_", "flat_", "s_load", "ds_write")),
        ("multiplies (weights, products)", ("v_mul_f32",)),
        ("adds / subs (weights, sums, centre)", ("v_add_f32", "v_sub_f32")),
        ("weights: byte -> float", ("v_cvt_f32_ubyte",)),
        ("integer address / index", ("v_mul_lo", "v_mad_", "v_add_u32", "v_sub_u32", "v_subrev", "v_lshl", "v_and_b32", "v_lshrrev", "v_add_co", "v_addc", "v_mov", "v_min_i32", "v_max_i32", "v_or_b32", "v_bfe", "v_xad", "v_xor", "v_mul_hi", "v_cvt_f32_u32", "v_cvt_u32", "v_rcp_iflag", "v_cmp_le_u32", "v_cmp_gt_i32", "v_cmp_gt_u32", "v_cmp_lt_u32", "v_cmp_lt_i32", "v_cmp_eq")),
        ("waits / nops", ("s_waitcnt", "s_nop", "s_barrier")),
        ("scalar", ("s_",)),
    ]
    for name, pre in table:
        if op.startswith(pre):
            return name
    return "other: " + op


# the fast path: the largest block before the taps' loads (head), then the two blocks at the
# end of the kernel (loads + unpack; look-up + interpolation + stores)
big = sorted(range(len(blocks)), key=lambda i: -len(blocks[i][1]))
head = min(i for i in big[:3])
fast = [head] + [i for i in range(len(blocks)) if i > head and len(blocks[i][1]) >= 60][-2:]
total = collections.Counter()
for i in fast:
    c = collections.Counter(kind(op) for op in blocks[i][1])
    total.update(c)
    print(f"block {blocks[i][0]}: {len(blocks[i][1])} instructions")
rows = 4
print(f"\nfast path, per thread (= {rows} pixels) and per pixel:")
vec = 0
for name, n in total.most_common():
    is_vec = name not in ("waits / nops", "scalar", "memory", "LDS table look-ups")
    vec += n if is_vec else 0
    print(f"  {name:42s} {n:5d}  {n / rows:6.1f}" + ("" if is_vec else "   (not vector ALU)"))
print(f"  {'vector-ALU instructions':42s} {vec:5d}  {vec / rows:6.1f}")
print(f"  {'all instructions':42s} {sum(total.values()):5d}  {sum(total.values()) / rows:6.1f}")
