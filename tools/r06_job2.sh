# round 6 GPU job 2: full -m gpu suite, tract floor by passes (new rule), QV calibration at scale, k_polish attribution by instructions
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 < /dev/null; echo "pytest rc $?" >> $O/pytest.txt
tail -4 $O/pytest.txt
timeout 900 python tools/tract_floor_study.py 384 3000 > $O/tract_floor.txt 2> $O/tract_floor.err < /dev/null
cat $O/tract_floor.txt
timeout 1500 python tools/qv_calibration.py 1024 > $O/qv_calibration.txt 2> $O/qv_calibration.err < /dev/null
cp gpurun_out/r06_qv_calibration.json $O/ 2>/dev/null
grep "^##" $O/qv_calibration.txt
timeout 1500 bash tools/polish_instr.sh > $O/polish_instr.txt 2>&1 < /dev/null
cat gpurun_out/pinstr/summary.txt
