# round 6 final evidence on one MI355X: full -m gpu suite, corruption fuzz (1200 batches), parity sweep, prof_round (bench default, kernel trace, PMC c2 / c4 / c5), CLI numbers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 < /dev/null; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
( echo "# HEAD 3ac1a9337f0d (final kernels of round 6: SPEC v8, POA by position, per-column fill table, trace-back by runs), seed0 1000000"; timeout 2400 python tools/corruption_fuzz.py 1200 1000000 ) > $O/corruption_fuzz.txt 2>&1 < /dev/null; tail -3 $O/corruption_fuzz.txt
( echo "# HEAD 3ac1a9337f0d"; timeout 1500 python tools/parity_sweep.py 24 ) > $O/parity_sweep.txt 2>&1 < /dev/null; tail -3 $O/parity_sweep.txt
timeout 3000 bash tools/prof_round.sh r06 3ac1a9337f0d > $O/prof_round.log 2>&1 < /dev/null; tail -5 $O/prof_round.log
timeout 1500 bash tools/final_numbers.sh r06_cli > $O/final_numbers.log 2>&1 < /dev/null; tail -30 gpurun_out/r06_cli/cli.txt
