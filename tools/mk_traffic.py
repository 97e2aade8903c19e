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
                    "VALU pipe occupancy (profiles/r06_valu_peak.txt: VOP2 float / add ops issue in ~ 2.5 cycles per wave64, integer max, compares, selects "
                    "and DPP ops in ~ 4.3); lanes_active_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)"}
# Round 6: no per-kernel "calibrated cycles per instruction" fitted to the counters it is then compared with.  The two bounds
# (every VALU instruction 2 cycles / 4 cycles) are what the counters say; beside them the kernel's OWN opcode histogram (tools/isa_histogram.py over hipcc -S, inner
# loops weighted) priced with the single-opcode table gives one estimate — static, a mixed stream overlaps more than the sum of its parts, so a result above 1
# is FLAGGED (isa_histogram_over_1), never clipped.
try:
    ISA = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", os.environ.get("ISA_HISTOGRAM", "r06_isa_histogram.json"))))
except Exception:
    ISA = {}
ISA_NAME = {"k_polish": "k_polish_t"}
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
            c_isa = ISA.get(ISA_NAME.get(k, k), {}).get("cycles_per_valu")
            if c_isa:
                e["valu_cycles_from_isa_histogram"] = c_isa
                e["valu_issue_frac_from_isa_histogram"] = round(d["SQ_INSTS_VALU"] * c_isa / simd_cycles, 3)
                if e["valu_issue_frac_from_isa_histogram"] > 1.0: e["isa_histogram_over_1"] = True
                if k.startswith("k_"): tot_busy += d["SQ_INSTS_VALU"] * c_isa; tot_cycles += simd_cycles
        if d.get("SQ_THREAD_CYCLES_VALU"):
            e["lanes_active_frac"] = round(d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_INSTS_VALU"]), 3)
    if d.get("GRBM_GUI_ACTIVE"): e["gpu_active_ms_per_pass"] = round(d["GRBM_GUI_ACTIVE"] / 8 / 2.4e6 / runs, 3)
    if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"): e["wave_wait_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"): e["lds_bank_conflict_frac"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 3)
    out["kernels"][k] = e
out["valu_issue_frac_from_isa_histogram_whole_step"] = round(tot_busy / tot_cycles, 3) if tot_cycles else None
out["valu_frac_note"] = ("valu_issue_frac_from_isa_histogram = SQ_INSTS_VALU x (cycles per VALU instruction of the kernel's own opcode histogram, tools/isa_histogram.py: "
                         + str(ISA.get("_method")) + ") / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); an estimate beside the two bounds valu_issue_frac_if_{2,4}cyc, flagged when above 1")
print(json.dumps(out, indent=1))
