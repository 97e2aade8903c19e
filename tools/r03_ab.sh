# round 3, first GPU call: GPU tests, then the two-stage queue A/B (serial stages vs two compute streams) and a kernel trace of both
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab; rm -rf $O; mkdir -p $O
nproc > $O/host.txt; free -g >> $O/host.txt; cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/memory.max >> $O/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for mode in two serial; do
  flag=""; [ $mode = serial ] && flag="--serial-stages"
  timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 13 --warmup 2 $flag > $O/bench_$mode.json 2> $O/bench_$mode.err
  python - <<PY
import json
d=json.load(open("$O/bench_$mode.json"))
print("$mode", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["stage_ms"])
PY
done
export TMPDIR=/tmp
for mode in two serial; do
  flag=""; [ $mode = serial ] && flag="--serial-stages"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$mode -o trace -- python $GRAFT_REPO_ROOT/bench.py --pmc --steps 6 --warmup 1 --distinct 3 $flag > $GRAFT_REPO_ROOT/$O/trace_$mode.json 2> $GRAFT_REPO_ROOT/$O/trace_$mode.err)
  python tools/trace_overlap.py $O/trace_$mode > $O/overlap_$mode.txt 2>&1
  cat $O/overlap_$mode.txt
  rm -rf $O/trace_$mode
done
