# PMC counters of the POA kernels (separate passes), 4096 ZMWs x 1 step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_pmc_poa; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
Z=${Z:-4096}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc --zmws $Z --steps 1 --warmup 1 --distinct 1 --serial-stages > $GRAFT_REPO_ROOT/$O/b_$tag.json 2> $GRAFT_REPO_ROOT/$O/b_$tag.err)
done
python - <<'PY'
import glob, sqlite3, os
val = {}
for db in glob.glob("gpurun_out/r03_pmc_poa/pmc_*/**/*results.db", recursive=True):
    c = sqlite3.connect(db)
    for kn, cn, v, k in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        val.setdefault(kn.split("(")[0], {})[cn] = (v, k)
Z = int(os.environ.get("Z", "4096"))
for k in sorted(val):
    if not k.startswith("k_"): continue
    d = val[k]
    runs = 2
    line = [k]
    for cn in sorted(d):
        line.append("%s %.4g" % (cn, d[cn][0] / runs / Z))
    print("  per ZMW: " + "  ".join(line))
PY
rm -rf $O/pmc_*
