# quick GPU loop (run through gpurun): parity tests, then serial-stage and two-stage bench lines.
#   usage: gpurun --timeout 1500 -- 'bash tools/gpu_quick.sh [full|subset|none] [extra bench.py args]'
# full = the whole -m gpu suite, subset = tests/test_gpu_parity.py + tests/test_golden.py, none = bench only.  Output: gpurun_out/quick/
cd $GRAFT_REPO_ROOT
WHAT=${1:-subset}; shift
O=gpurun_out/quick; rm -rf $O; mkdir -p $O
case $WHAT in
  full)   timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt;;
  subset) timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt;;
  none)   echo "pytest skipped" > $O/pytest.txt;;
esac
tail -5 $O/pytest.txt
for mode in "--serial-stages" ""; do
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 8 --warmup 3 $mode "$@" > $O/b.json 2> $O/b.err
    python - <<PY
import json
try:
    d = json.load(open("$O/b.json")); s = d["stage_ms"]
    print("%-16s %8.0f ZMW/s %7.1f ms/step | draft %.1f align %.1f polish %.1f total %.1f" % ("$mode" or "two-stage", d["value"], d["ms_per_step"], s["draft_ms"], s["align_ms"], s["polish_ms"], s["total_ms"]))
except Exception as e:
    print("bench failed", e); print(open("$O/b.err").read()[-1500:])
PY
  done
done | tee $O/bench.txt
