# round 6 GPU job: full -m gpu suite, POA traffic (FETCH/WRITE passes), tract-floor-by-passes study
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 < /dev/null; echo "pytest rc $?" >> $O/pytest.txt
tail -4 $O/pytest.txt
Z=16384 KERNELS='k_poa%' SETS="FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE" timeout 900 bash tools/gpu_pmc.sh > $O/pmc_poa.txt 2>&1 < /dev/null
cp gpurun_out/pmc/summary.txt $O/pmc_poa_summary.txt; cat $O/pmc_poa_summary.txt
timeout 900 python tools/tract_floor_study.py 384 3000 > $O/tract_floor.txt 2> $O/tract_floor.err < /dev/null
cat $O/tract_floor.txt
