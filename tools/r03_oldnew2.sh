# same-box A/B: HEAD's kernels + API (tools/_old_kernels.hip, tools/_old_api.cpp, untracked) vs the working tree's
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_oldnew; rm -rf $O; mkdir -p $O
run() {
  name=$1
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 8 --warmup 3 --serial-stages > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.load(open('$O/b.json')); s=d['stage_ms']; print('$name serial: draft %.1f align %.1f polish %.1f total %.1f' % (s['draft_ms'], s['align_ms'], s['polish_ms'], s['total_ms']))"
  done
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 10 --warmup 3 > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('$name two-stage: %.0f ZMW/s %.1f ms/step' % (d['value'], d['ms_per_step']))"
  done
}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run new
cp ccs_amd/csrc/ccsx_kernels.hip $O/new.hip; cp ccs_amd/csrc/ccsx_api.cpp $O/new_api.cpp
cp tools/_old_kernels.hip ccs_amd/csrc/ccsx_kernels.hip; cp tools/_old_api.cpp ccs_amd/csrc/ccsx_api.cpp
python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_old.log 2>&1 || tail -5 $O/build_old.log
run old
cp $O/new.hip ccs_amd/csrc/ccsx_kernels.hip; cp $O/new_api.cpp ccs_amd/csrc/ccsx_api.cpp; rm $O/new.hip $O/new_api.cpp
