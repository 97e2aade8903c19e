#!/usr/bin/env python3
"""SIMD cycles per VALU instruction of every kernel from its OWN instruction stream (never fit the figure to the counters it is compared
with).  `hipcc -S` of ccsx_kernels.hip with the product's flags; per kernel the VALU opcodes are counted with a weight of 8^(loop depth) — the compiler's
"in Loop: ... Depth=N" annotations — as a stand-in for the dynamic mix (inner loops dominate; the weight is the one free choice here and is printed), and each
opcode is priced from the single-opcode table measured on the device at 4-8 waves per SIMD (profiles/r06_valu_peak.txt).  Output: JSON on stdout,
{kernel: {"cycles_per_valu": c, "valu_weighted": n, "top": [[opcode, share], ...]}}; tools/mk_traffic.py multiplies c with SQ_INSTS_VALU and divides by the
kernel's SIMD cycles — a result above 1 is flagged there, not hidden.
    python tools/isa_histogram.py > profiles/r06_isa_histogram.json"""
import json, os, re, subprocess, sys, tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-falign-loops=64"]
DEPTH_WEIGHT = 8.0

# SIMD cycles per wave64 instruction, profiles/r06_valu_peak.txt (4 and 8 waves per SIMD)
FLOAT_FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_max_f32", "v_min_f32", "v_mac_f32")
INT_FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
            "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_mul_u32_u24", "v_mul_i32_i24", "v_bfrev_b32", "v_ffbh_u32", "v_ffbl_b32", "v_bcnt_u32_b32")


def price(op, line):
    if "dpp" in op or "row_" in line or "wave_sh" in line or "sdwa" in op: return 4.3
    base = re.sub(r"_(e32|e64)$", "", op)
    if base.startswith("v_cmp"): return 4.2
    if base.startswith("v_cndmask"): return 4.3
    if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"): return 6.1
    if base in FLOAT_FAST: return 2.5
    if base in INT_FAST: return 2.8
    if base.startswith(("v_max_", "v_min_", "v_med3", "v_lshl_add", "v_add3", "v_lshl_or", "v_and_or", "v_or3", "v_bfe", "v_bfi", "v_perm", "v_alignbit", "v_mad_", "v_add_lshl", "v_xad", "v_pk_", "v_cvt", "v_mbcnt", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")): return 4.4
    if base.startswith(("v_mul_lo", "v_mul_hi", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_div_", "v_ldexp", "v_frexp", "v_trunc", "v_floor", "v_rndne", "v_fract")): return 8.0
    return 4.4                                                 # anything not measured: the slow class


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(R, "include"), "-I" + os.path.join(R, "ccs_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                        os.path.join(R, "ccs_amd", "csrc", "ccsx_kernels.hip")], check=True, capture_output=True)
        lines = open(out).read().splitlines()
    res, cur, depth = {}, None, 0
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            cur = m.group(1); depth = 0; res[cur] = {}
            continue
        if cur is None: continue
        if ln.startswith(".Lfunc_end"): cur = None; continue
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", ln):
            d = re.search(r"Depth=(\d+)", ln)
            depth = int(d.group(1)) if d else 0
            continue
        d = re.search(r"Loop Header: Depth=(\d+)", ln)
        if d: depth = int(d.group(1)); continue
        t = ln.strip().split()
        if not t or not t[0].startswith("v_") or t[0].startswith(("v_nop", "v_accvgpr")): continue
        w = DEPTH_WEIGHT ** depth
        e = res[cur].setdefault(t[0], [0.0, 0.0])
        e[0] += w; e[1] += w * price(t[0], ln)
    outj = {"_method": f"static VALU opcode histogram of hipcc -S, weight {DEPTH_WEIGHT:g}^(loop depth), priced with profiles/r06_valu_peak.txt (4-8 waves per SIMD)"}
    for k, h in res.items():
        n = sum(v[0] for v in h.values()); c = sum(v[1] for v in h.values())
        if n <= 0: continue
        mm = re.match(r"^_Z(\d+)", k)                          # Itanium mangling: _Z <length> <name> ...
        name = k[mm.end():mm.end() + int(mm.group(1))] if mm else k
        top = sorted(h.items(), key=lambda kv: -kv[1][0])[:8]
        outj[name] = {"cycles_per_valu": round(c / n, 3), "valu_weighted": round(n, 1), "top": [[o, round(v[0] / n, 3)] for o, v in top]}
    json.dump(outj, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
