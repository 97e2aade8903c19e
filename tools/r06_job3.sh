# round 6 GPU job 3: trace-back by runs (parity subset + c4/c2 bench), QV calibration at scale (default cap and max_qv 93)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
timeout 1400 bash tools/gpu_quick.sh subset > $O/quick.txt 2>&1 < /dev/null; tail -6 $O/quick.txt
for rep in 1 2; do timeout 600 python bench.py --no-cpu-baseline --extra '' --workload c4 --zmws 4096 --steps 4 --warmup 3 --serial-stages > $O/c4_$rep.json 2> $O/c4_$rep.err < /dev/null; python -c "
import json; d = json.load(open('$O/c4_$rep.json')); print('c4 serial', d['value'], d['stage_ms'])"; done
timeout 600 python bench.py --no-cpu-baseline --extra '' --workload c4 --steps 4 --warmup 3 > $O/c4_two.json 2> $O/c4_two.err < /dev/null; python -c "
import json; d = json.load(open('$O/c4_two.json')); print('c4 two-stage', d['value'], d['stage_ms'])"
sleep 5
timeout 1800 python tools/qv_calibration.py 1024 > $O/qv_calibration.txt 2> $O/qv_calibration.err < /dev/null
cp gpurun_out/r06_qv_calibration.json $O/ 2>/dev/null
grep "^##" $O/qv_calibration.txt
sleep 5
MAX_QV=93 timeout 1800 python tools/qv_calibration.py 512 > $O/qv_calibration_maxqv93.txt 2> $O/qv_calibration_maxqv93.err < /dev/null
cp gpurun_out/r06_qv_calibration_maxqv93.json $O/ 2>/dev/null
grep "^##" $O/qv_calibration_maxqv93.txt
