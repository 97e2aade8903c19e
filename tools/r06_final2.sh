# the round's last GPU check: full -m gpu suite and the default bench line (with profiles/r06_traffic*.json in place), smoke()
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 < /dev/null; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 < /dev/null; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; python -c "
import json; d = json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['head'], d['roofline']['traffic'], d['roofline'].get('traffic_matches_these_sources'), d['roofline']['valu'], {k: v.get('value') for k, v in d['extra'].items() if isinstance(v, dict) and 'value' in v})"
