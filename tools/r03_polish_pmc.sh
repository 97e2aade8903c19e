# k_polish: parity subset, serial-stage timing, LDS bank-conflict counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_polish; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for mode in serial two; do
  flag=""; [ $mode = serial ] && flag="--serial-stages"
  timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 10 --warmup 3 $flag > $O/b_$mode.json 2> $O/b_$mode.err
  python -c "
import json; d=json.load(open('$O/b_$mode.json')); print('$mode', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['stage_ms'])"
done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/$O/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc --zmws 2048 --steps 1 --warmup 1 --distinct 1 --serial-stages > /dev/null 2>&1)
python - <<'PY'
import glob, sqlite3
for db in glob.glob("gpurun_out/r03_polish/pmc/**/*results.db", recursive=True):
    c = sqlite3.connect(db)
    d = {cn: v for kn, cn, v in c.execute("select kernel_name, counter_name, sum(value) from counters_collection where kernel_name like 'k_polish%' group by kernel_name, counter_name")}
    print("k_polish per ZMW:", {k: round(v / 2 / 2048) for k, v in d.items()}, "conflict frac %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf $O/pmc
