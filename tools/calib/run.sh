cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/calib; rm -rf $D; mkdir -p $D
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d $D/pmc_$c -o pmc -- $R/tools/calib/calib_fetch > $D/out_$c.txt 2>&1; done
python $R/tools/profsum.py $D > $D/summary.txt; cat $D/summary.txt; tail -1 $D/out_FETCH_SIZE.txt
