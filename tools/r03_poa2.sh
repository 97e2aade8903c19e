# quick loop for POA kernel work: parity subset, bench at two batch sizes, VALU counters of the POA kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_poa2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for z in 8192 16384; do
  timeout 400 python bench.py --no-cpu-baseline --extra '' --steps 6 --warmup 2 --zmws $z --serial-stages > $O/bench_$z.json 2> $O/bench_$z.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$z.json"))
    print("$z", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["stage_ms"])
except Exception as e: print("bench $z failed", e); print(open("$O/bench_$z.err").read()[-1500:])
PY
done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/$O/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc --zmws 4096 --steps 1 --warmup 1 --distinct 1 --serial-stages > $GRAFT_REPO_ROOT/$O/b_pmc.json 2> $GRAFT_REPO_ROOT/$O/b_pmc.err)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --pmc --zmws 16384 --steps 3 --warmup 1 --distinct 2 --serial-stages > $GRAFT_REPO_ROOT/$O/b_trace.json 2> $GRAFT_REPO_ROOT/$O/b_trace.err)
python tools/profsum.py $O | grep -E "k_poa|k_align16|k_polish|kernel " 
python - <<'PY'
import glob, sqlite3
val = {}
for db in glob.glob("gpurun_out/r03_poa2/pmc/**/*results.db", recursive=True):
    c = sqlite3.connect(db)
    for kn, cn, v, k in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        val.setdefault(kn.split("(")[0], {})[cn] = v
for k in sorted(val):
    if k.startswith("k_poa"): print(k, {cn: round(v / 2 / 4096) for cn, v in sorted(val[k].items())})
PY
rm -rf $O/pmc $O/trace
