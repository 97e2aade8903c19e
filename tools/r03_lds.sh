# experiment: k_polish LDS per workgroup vs the two-stage queue (does leaving LDS for the draft-stage kernels pay?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_lds; rm -rf $O; mkdir -p $O
for v in ${LDS_LIST:-52992 40448 32256}; do
  CCSX_EXTRA_FLAGS="-DPW_LDS_BYTES=$v" python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_$v.log 2>&1 || { echo "build $v failed"; tail -5 $O/build_$v.log; continue; }
  for mode in two serial; do
    flag=""; [ $mode = serial ] && flag="--serial-stages"
    timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 10 --warmup 2 $flag > $O/b_${v}_$mode.json 2> $O/b_${v}_$mode.err
    python - <<PY
import json
try:
    d=json.load(open("$O/b_${v}_$mode.json")); print("PW_LDS_BYTES $v $mode: %.0f ZMW/s, %.1f ms/step, kernels %.1f" % (d["value"], d["ms_per_step"], d["kernels_ms_per_step"]), d["stage_ms"])
except Exception as e: print("$v $mode failed", e)
PY
  done
done
python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
