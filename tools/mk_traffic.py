#!/usr/bin/env python3
"""Per-kernel HBM traffic (FETCH_SIZE + WRITE_SIZE, raw KB) and VALU occupancy from the PMC passes of tools/prof_round.sh."""
import glob, json, os, sqlite3, sys
root, n = sys.argv[1], int(sys.argv[2])
val = {}
for db in glob.glob(os.path.join(root, "pmc_*", "pmc_results.db")):
    c = sqlite3.connect(db)
    for kn, cn, v in c.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        val.setdefault(kn.split("(")[0], {})[cn] = v
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE / SQ_THREAD_CYCLES_VALU (separate passes), python bench.py "
                 f"--zmws {n} --steps 1 --warmup 0, tools/prof_round.sh",
       "note": "hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on this gfx950 stack FETCH_SIZE reports exactly half of the bytes of a "
               "coalesced streaming read at 1, 4 and 16 bytes per lane and WRITE_SIZE is exact (profiles/r01_counter_calibration.txt, "
               "tools/calib; MI355X_MICROARCH.md HBM section)",
       "zmws": n, "workload": "10 passes x 10 kb", "kernels": {},
       "valu_busy_note": "SQ_THREAD_CYCLES_VALU/64 quad-cycles x4 / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs)"}
for k, d in sorted(val.items()):
    e = {}
    if "FETCH_SIZE" in d: e["fetch_size_kb_per_zmw"] = d["FETCH_SIZE"] / n
    if "WRITE_SIZE" in d: e["write_size_kb_per_zmw"] = d["WRITE_SIZE"] / n
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d: e["hbm_bytes_per_zmw"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / n)
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("GRBM_GUI_ACTIVE"):
        e["valu_busy_frac"] = round(d["SQ_THREAD_CYCLES_VALU"] / 64 * 4 / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
