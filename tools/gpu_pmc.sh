# PMC counters per kernel on one MI355X (run through gpurun): one rocprofv3 pass per counter set over `bench.py --pmc` (the headline
# workload and nothing else), summed per kernel and printed per ZMW.  Replaces round 3's r03_polish_pmc / r03_pmc_poa / r03_lds /
# prof_lds / prof_sq / prof_valu / pmc_sq / count_valu one-offs.
#   Z=8192 (ZMWs per step)   KERNELS='k_polish%' (sqlite LIKE pattern; default every k_ kernel)   SERIAL=1 (--serial-stages)
#   SETS="SQ_INSTS_VALU,SQ_WAVE_CYCLES FETCH_SIZE WRITE_SIZE"   (space separated passes, comma separated counters; defaults below)
#   usage: gpurun --timeout 1500 -- 'Z=8192 KERNELS=k_polish% bash tools/gpu_pmc.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
Z=${Z:-8192}
SETS=${SETS:-"SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_WAIT_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_THREAD_CYCLES_VALU,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE,FETCH_SIZE WRITE_SIZE"}
flag=""; [ "${SERIAL:-1}" = 1 ] && flag="--serial-stages"
i=0
for set in $SETS; do
  i=$((i + 1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc ${set//,/ } -d $GRAFT_REPO_ROOT/$O/pass$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc --zmws $Z --steps 1 --warmup 1 --distinct 1 $flag $BENCH_ARGS > $GRAFT_REPO_ROOT/$O/b$i.json 2> $GRAFT_REPO_ROOT/$O/b$i.err) || { echo "pass $i ($set) failed"; tail -3 $O/b$i.err; }
done
Z=$Z KERNELS="${KERNELS:-k_%}" python - <<'PY' | tee gpurun_out/pmc/summary.txt
import glob, sqlite3, os
val = {}
for db in glob.glob("gpurun_out/pmc/pass*/**/*results.db", recursive=True):
    c = sqlite3.connect(db)
    for kn, cn, v in c.execute("select kernel_name, counter_name, sum(value) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (os.environ["KERNELS"],)):
        val.setdefault(kn.split("(")[0], {})[cn] = v
Z, runs = int(os.environ["Z"]), 2          # bench.py --pmc --steps 1 --warmup 1 = two launches of every kernel
for k in sorted(val):
    d = val[k]
    print(k, "per ZMW:", "  ".join("%s %.4g" % (cn, d[cn] / runs / Z) for cn in sorted(d)))
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"): print("   LDS bank-conflict fraction %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_WAIT_INST_ANY" in d and d.get("SQ_WAVE_CYCLES"): print("   waves waiting %.3f   VALU-active %.3f" % (d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / d["SQ_WAVE_CYCLES"]))
PY
rm -rf $O/pass*
