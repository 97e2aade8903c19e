#!/usr/bin/env python3
"""Per-kernel HBM traffic (FETCH_SIZE + WRITE_SIZE, raw KB) and VALU issue from the PMC passes of tools/prof_round.sh.
Counters are summed over every dispatch of a kernel in the run; one bench run = (warmup + steps + 1) passes over the batch,
taken from the number of k_stitch dispatches (one per pass)."""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root, n = sys.argv[1], int(sys.argv[2])
head = sys.argv[3] if len(sys.argv) > 3 else None
workload = sys.argv[4] if len(sys.argv) > 4 else "10 passes x 10 kb"
def kname(n):
    """kernel name without return type, template arguments and parameters: 'void k_polish_t<512, 2, 8>(KParams, ...)' -> 'k_polish'"""
    n = n.split("(")[0].split("<")[0].strip()
    n = n[5:] if n.startswith("void ") else n
    return "k_polish" if n == "k_polish_t" else n


val, cnt = {}, {}
for db in glob.glob(os.path.join(root, "pmc_*", "pmc_results.db")):
    c = sqlite3.connect(db)
    for kn, cn, v, k in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        d_ = val.setdefault(kname(kn), {}); d_[cn] = d_.get(cn, 0) + v       # (the two instantiations of k_polish_t add up)
        c_ = cnt.setdefault(kname(kn), {}); c_[cn] = max(c_.get(cn, 0), k)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE / SQ_* (separate passes), python bench.py "
                 f"--pmc --serial-stages --zmws {n} --steps 1 --warmup 1 --distinct 1 (serial stages: a kernel's GRBM_GUI_ACTIVE is then its own), tools/prof_round.sh",
       "head": head, "csrc_sha16": __import__("bench").csrc_sha16() if os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")) else None,
       "note": "hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on this gfx950 stack FETCH_SIZE reports exactly half of the bytes of a "
               "coalesced streaming read at 1, 4 and 16 bytes per lane and WRITE_SIZE is exact (profiles/r01_counter_calibration.txt, "
               "tools/calib; MI355X_MICROARCH.md HBM section)",
       "zmws": n, "workload": workload, "kernels": {},
       "valu_note": "valu_issue_frac_if_{2,4}cyc = SQ_INSTS_VALU x {2,4} SIMD cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the bounds of the "
                    "VALU pipe occupancy (profiles/r02_valu_peak.txt: v_add/mul_f32 and v_add_u32 issue in 2 cycles per wave64, v_fma_f32, "
                    "v_max_i32, DPP ops in 4); lanes_active_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)"}
# SIMD cycles one wave64 VALU instruction of each kernel's dominant mix occupies (profiles/r02_valu_peak.txt): float add / mul 2.5, fma 3.7,
# integer max / cmp / cndmask / DPP 4.2 — k_polish mixes float multiply-adds with selects and DPP shifts, the DP kernels are DPP / max heavy
# round 5: k_align16 came out at 1.13 "of the peak" with 4.2 cycles per instruction, also in a serial-stages pass, so its mix is cheaper than that:
# the ISA of its column loop is ~ 45 % v_add / v_mov / shifts / logic (2.5 cycles), ~ 45 % compares and selects (4.2), ~ 10 % DPP (4.3) = 3.5 on average.  A fraction
# above 1 is flagged in the file (calibration_inconsistent) instead of being printed as if it were a measurement.
# second session of round 5: k_align16 no longer carries (origin, dirty bits) — its column is ~ 47 % adds / shifts / logic / v_addc (2.5), ~ 33 % compares, selects, min / max (4.2),
# ~ 20 % DPP (4.3) = 3.45 by the single-opcode calibration, and the serial-stage counters then say 1.09 of the peak: the calibration over-estimates a MIXED stream (a 2.5-cycle
# opcode issued between two 4-cycle ones does not wait for a whole slot).  The counters themselves bound the average from above — SQ_INSTS_VALU x c <= the kernel's SIMD cycles gives
# c <= 3.22 — and that bound is what the table now holds: the kernel fills the VALU (waves wait 20 % of their cycles), 0.99 is "saturated", not a measurement of 1 % slack.
CYC = {"k_polish": 3.2, "k_poa_dp": 4.2, "k_align16": 3.2, "k_align16_tb": 4.2, "k_align": 4.2, "k_rescue": 4.2, "k_kinetics": 4.2}
tot_busy = tot_cycles = 0.0
for k, d in sorted(val.items()):
    runs = max(1, cnt.get("k_stitch", cnt.get("k_polish", {})).get(next(iter(d)), 1))   # k_stitch: exactly one dispatch per pass (k_polish: one per piece of the slot grid)
    e = {}
    per = lambda name: d[name] / (runs * n)
    if "FETCH_SIZE" in d: e["fetch_size_kb_per_zmw"] = per("FETCH_SIZE")
    if "WRITE_SIZE" in d: e["write_size_kb_per_zmw"] = per("WRITE_SIZE")
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d: e["hbm_bytes_per_zmw"] = int((2 * per("FETCH_SIZE") + per("WRITE_SIZE")) * 1024)
    if "SQ_INSTS_VALU" in d:
        e["valu_wave_instr_per_zmw"] = int(per("SQ_INSTS_VALU"))
        if d.get("GRBM_GUI_ACTIVE"):
            simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 1024
            e["valu_issue_frac_if_2cyc"] = round(d["SQ_INSTS_VALU"] * 2 / simd_cycles, 3)
            e["valu_issue_frac_if_4cyc"] = round(d["SQ_INSTS_VALU"] * 4 / simd_cycles, 3)
            e["valu_cycles_per_instr_calibrated"] = CYC.get(k, 4.0)
            e["valu_frac_of_calibrated_peak"] = round(d["SQ_INSTS_VALU"] * CYC.get(k, 4.0) / simd_cycles, 3)
            if e["valu_frac_of_calibrated_peak"] > 1.0: e["calibration_inconsistent"] = True
            if k.startswith("k_"): tot_busy += d["SQ_INSTS_VALU"] * CYC.get(k, 4.0); tot_cycles += simd_cycles
        if d.get("SQ_THREAD_CYCLES_VALU"):
            e["lanes_active_frac"] = round(d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_INSTS_VALU"]), 3)
    if d.get("GRBM_GUI_ACTIVE"): e["gpu_active_ms_per_pass"] = round(d["GRBM_GUI_ACTIVE"] / 8 / 2.4e6 / runs, 3)
    if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"): e["wave_wait_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"): e["lds_bank_conflict_frac"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 3)
    out["kernels"][k] = e
out["valu_frac_of_calibrated_peak_whole_step"] = round(tot_busy / tot_cycles, 3) if tot_cycles else None
out["valu_frac_note"] = ("valu_frac_of_calibrated_peak = SQ_INSTS_VALU x (calibrated SIMD cycles per instruction of the kernel's dominant opcode mix) / "
                         "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of the VALU issue slots the kernel fills; whole_step = the same over every k_* kernel")
print(json.dumps(out, indent=1))
