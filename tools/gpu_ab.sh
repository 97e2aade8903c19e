# same-box A/B on one MI355X (run through gpurun): every variant is built, checked against the oracle (parity subset) and timed with
# the serial-stage and the two-stage bench.  Replaces round 3's one-off r03_oldnew* / r03_variant* / r03_ab scripts.
#   VARIANTS="name:-DFLAG=1,-DOTHER=2 name2:"      compile-time variants of the working tree (CCSX_EXTRA_FLAGS); "name:" = no flags
#   OLD=tools/_old_kernels.hip                     additionally: the kernel file of an earlier commit (git show REV:ccs_amd/csrc/ccsx_kernels.hip > tools/_old_kernels.hip)
#   BENCH_ARGS="--workload c4"                     extra bench.py arguments;  PARITY=0 skips the parity subset;  MODES=--serial-stages REPS=1: one serial-stage line only
#   every result line is also appended to gpurun_out/ab_all.txt (survives several invocations in one gpurun call)
#   usage: gpurun --timeout 1500 -- 'VARIANTS="head: lds48:-DPW_LDS_BYTES=49152" bash tools/gpu_ab.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
run() {
  name=$1
  if [ "${PARITY:-1}" != 0 ]; then
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/pytest_$name.txt 2>&1
    tail -1 $O/pytest_$name.txt
    if ! grep -q " passed" $O/pytest_$name.txt || grep -q " failed" $O/pytest_$name.txt; then echo "parity $name FAILED"; grep -m3 "Error\|error\|FAILED" $O/pytest_$name.txt; return; fi
  fi
  for mode in ${MODES:-"--serial-stages" ""}; do
    for rep in $(seq ${REPS:-2}); do
      timeout 400 python bench.py --no-cpu-baseline --extra '' --steps ${STEPS:-8} --warmup 3 $mode $BENCH_ARGS > $O/b.json 2> $O/b.err
      python - <<PY
import json
try:
    d = json.load(open("$O/b.json")); s = d["stage_ms"]
    print("%-12s %-15s %8.0f ZMW/s %7.1f ms/step | draft %.1f align %.1f polish %.1f total %.1f" % ("$name", "$mode" or "two-stage", d["value"], d["ms_per_step"], s["draft_ms"], s["align_ms"], s["polish_ms"], s["total_ms"]))
except Exception as e:
    print("$name bench failed", e); print(open("$O/b.err").read()[-800:])
PY
    done
  done
}
for v in $VARIANTS; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  CCSX_EXTRA_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; continue; }
  run $name
done 2>&1 | tee -a $O/ab.txt gpurun_out/ab_all.txt
if [ -n "$OLD" ] && [ -f "$OLD" ]; then
  cp ccs_amd/csrc/ccsx_kernels.hip $O/new.hip
  cp $OLD ccs_amd/csrc/ccsx_kernels.hip
  { python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_old.log 2>&1 && PARITY=0 run old || tail -5 $O/build_old.log; } 2>&1 | tee -a $O/ab.txt gpurun_out/ab_all.txt
  cp $O/new.hip ccs_amd/csrc/ccsx_kernels.hip; rm $O/new.hip
fi
python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
