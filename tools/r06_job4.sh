cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
timeout 1800 python tools/qv_calibration.py 1024 > $O/qv_calibration.txt 2> $O/qv_calibration.err < /dev/null
cp gpurun_out/r06_qv_calibration.json $O/ 2>/dev/null
grep "^##" $O/qv_calibration.txt
MAX_QV=93 timeout 1800 python tools/qv_calibration.py 512 > $O/qv_calibration_maxqv93.txt 2> $O/qv_calibration_maxqv93.err < /dev/null
cp gpurun_out/r06_qv_calibration_maxqv93.json $O/ 2>/dev/null
grep "^##" $O/qv_calibration_maxqv93.txt
