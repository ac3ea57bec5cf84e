#!/usr/bin/env python3
"""Turns a summary of profiles/run_pmc.sh (bench workload, one k_tick launch) into the two records bench.py quotes:
profiles/pmc_traffic.json (HBM bytes per launch) and profiles/pmc_issue.json (VALU issue per launch, shader clock).
usage: pmc_to_json.py <summary.txt> <kernel ms of that profiled launch> <label>"""
import json, os, sys
src, kernel_ms, label = sys.argv[1], float(sys.argv[2]), sys.argv[3]
v = {}
for line in open(src):
    f = line.split()
    if len(f) >= 5 and f[2] == "per":
        v[f[0]] = float(f[4])
here = os.path.dirname(os.path.abspath(__file__))
fetch_kb, write_kb = v["FETCH_SIZE"], v["WRITE_SIZE"]
json.dump({
    "source": "%s (rocprofv3 --pmc, passes 3 and 4 of profiles/run_pmc.sh, one k_tick_spec launch of the bench workload)" % os.path.basename(src),
    "collected": label,
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950 FETCH_SIZE tallies 128-B read requests at 64 B: doubled; WRITE_SIZE as read. The doubling is calibrated for wide "
                  "streaming reads; k_tick reads 4-8 B per lane, so the figure is an upper estimate.",
    "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
}, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
xcds = 8
cycles = v["GRBM_GUI_ACTIVE"] / xcds                 # the counter is summed over the 8 XCDs
mhz = cycles / (kernel_ms * 1e-3) / 1e6
json.dump({
    "source": os.path.basename(src), "collected": label,
    "SQ_INSTS_VALU_per_launch": v["SQ_INSTS_VALU"], "SQ_INSTS_SALU_per_launch": v["SQ_INSTS_SALU"], "SQ_INSTS_LDS_per_launch": v["SQ_INSTS_LDS"],
    "GRBM_GUI_ACTIVE_per_xcd": cycles, "profiled_kernel_ms": kernel_ms, "shader_clock_mhz": mhz,
    "valu_busy_frac": v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles),
    "exec_mask_occupancy": v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]),
}, open(os.path.join(here, "pmc_issue.json"), "w"), indent=1)
print(open(os.path.join(here, "pmc_issue.json")).read())
