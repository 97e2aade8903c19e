# quick loop: parity subset + one bench line (serial stages, 8192)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_quick; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout 400 python bench.py --no-cpu-baseline --extra '' --steps 6 --warmup 2 --serial-stages $BENCH_ARGS > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench.json"))
    print(d["value"], d["ms_per_step"], d["stage_ms"])
except Exception as e: print("bench failed", e); print(open("$O/bench.err").read()[-1500:])
PY
