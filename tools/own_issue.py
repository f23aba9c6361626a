#!/usr/bin/env python3
"""The ownership kernel against ITS roof - the vector ALU's issue slots, not HBM - from a counter
summary (tools/pmc.sh -> pmc_summary.py's summary.txt):

    tools/own_issue.py <summary.txt> <workload> <out.json>

SQ_ACTIVE_INST_VALU counts, in quad-cycles, the time waves spent executing vector-ALU instructions
(MI355X_MICROARCH.md: SQ_ACTIVE_INST_* are quad-cycles); a SIMD executes one such instruction at a
time, the chip has 256 CUs x 4 SIMDs, and GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs.  So

    valu_busy = 4 SQ_ACTIVE_INST_VALU / 1024  /  (GRBM_GUI_ACTIVE / 8)

is the share of the kernel's SIMD-cycles in which a vector instruction was executing: the kernel's
fraction of its issue roof, clock-free.  1 / valu_busy is how far above its issue bound it runs
(latency chains: profiles/r05/notes.md section 2).  HBM is not its roof: 0.1 GB of algorithmic
bytes, counter traffic 1.05 x that."""
import json
import re
import sys

SIMDS, XCDS = 1024, 8


def main(summary, workload, out, kernel="ownership_cameras_kernel"):
    vals, on = {}, False
    with open(summary) as fid:
        for line in fid:
            if line.startswith("== "):
                on = line[3:].split()[0] == kernel
                continue
            m = re.match(r"\s+(\S+)\s+avg\s+([0-9.eE+-]+)", line)
            if on and m:
                vals[m.group(1)] = float(m.group(2))
    need = ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")
    missing = [k for k in need if k not in vals]
    if missing:
        raise SystemExit(f"{summary}: no {missing} for {kernel}")
    simd_cycles = vals["GRBM_GUI_ACTIVE"] / XCDS * SIMDS
    busy = 4.0 * vals["SQ_ACTIVE_INST_VALU"] / simd_cycles
    rec = {
        "kernel": kernel, "workload": workload, "bound": "valu",
        "valu_busy": busy,
        "times_its_issue_bound": 1.0 / busy,
        "valu_insts_per_launch": vals["SQ_INSTS_VALU"],
        "cycles_per_valu_inst": 4.0 * vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_INSTS_VALU"],
        "gui_active_cycles_per_xcd": vals["GRBM_GUI_ACTIVE"] / XCDS,
        "sq_active_inst_valu_quadcycles": vals["SQ_ACTIVE_INST_VALU"],
        "sq_busy_cycles": vals.get("SQ_BUSY_CYCLES"),
        "source": summary,
        "how": "valu_busy = 4 SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), per-launch "
               "averages of separate --pmc passes (tools/pmc.sh)",
    }
    with open(out, "w") as fid:
        json.dump(rec, fid, indent=1)
    print(f"{kernel} on {workload}: VALU busy {busy:.3f} of its SIMD-cycles = {1 / busy:.2f} x its issue "
          f"bound; {vals['SQ_INSTS_VALU'] / 1e6:.1f} M vector instructions per launch at "
          f"{rec['cycles_per_valu_inst']:.2f} cycles each")


if __name__ == "__main__":
    main(*sys.argv[1:4])
