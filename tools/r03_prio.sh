# two-stage queue: does a stream priority for one of the stages change the throughput?  (CCSX_STAGE_PRIO experiment knob)
cd $GRAFT_REPO_ROOT
for p in none draft polish none draft polish; do
  if [ $p = none ]; then unset CCSX_STAGE_PRIO; else export CCSX_STAGE_PRIO=$p; fi
  timeout 300 python bench.py --no-cpu-baseline --extra '' --steps 10 --warmup 3 > /tmp/b.json 2> /tmp/b.err
  python -c "
import json; d=json.load(open('/tmp/b.json')); s=d['stage_ms']; print('prio $p: %.0f ZMW/s %.1f ms/step  (overlapped stage ms: draft %.1f align %.1f polish %.1f)' % (d['value'], d['ms_per_step'], s['draft_ms'], s['align_ms'], s['polish_ms']))"
done
