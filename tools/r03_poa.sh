# round 3: the four-graphs-per-wave POA — parity tests, then the bench at 8192 and 16384 ZMWs per step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_poa; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -15 $O/pytest.txt
for z in 8192 16384; do
  timeout 400 python bench.py --no-cpu-baseline --extra '' --steps 8 --warmup 2 --zmws $z --serial-stages > $O/bench_$z.json 2> $O/bench_$z.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$z.json"))
    print("$z", d["value"], d["ms_per_step"], d["stage_ms"])
except Exception as e: print("bench $z failed", e); print(open("$O/bench_$z.err").read()[-1500:])
PY
done
