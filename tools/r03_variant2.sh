# same-box A/B of compile-time variants with a parity check each: VARIANTS="name:-DFLAG=..,-DFLAG2 name2:"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_variant; rm -rf $O; mkdir -p $O
for v in $VARIANTS; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  CCSX_EXTRA_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; continue; }
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/pytest_$name.txt 2>&1; tail -1 $O/pytest_$name.txt; grep -q " passed" $O/pytest_$name.txt && ! grep -q failed $O/pytest_$name.txt || { echo "parity $name FAILED"; grep -m3 "Error\|error\|FAILED" $O/pytest_$name.txt; continue; }
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 8 --warmup 3 --serial-stages > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.load(open('$O/b.json')); s=d['stage_ms']; print('$name serial: draft %.1f align %.1f polish %.1f total %.1f' % (s['draft_ms'], s['align_ms'], s['polish_ms'], s['total_ms']))"
  done
  timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 10 --warmup 3 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('$name two-stage: %.0f ZMW/s %.1f ms/step' % (d['value'], d['ms_per_step']))"
done
python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
