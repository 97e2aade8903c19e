# round-2 first GPU probe: VALU issue calibration, GPU tests, phase shares, LDS conflicts, reference-binary probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_probe; rm -rf $O; mkdir -p $O
( command -v ccs && ccs --version ) > $O/ccs_probe.txt 2>&1 || echo "ccs: not on PATH" >> $O/ccs_probe.txt
( command -v samtools pbindex pbmerge; ls /opt/conda/bin 2>/dev/null | head ) >> $O/ccs_probe.txt 2>&1
nproc >> $O/ccs_probe.txt; cat /sys/fs/cgroup/cpu.max >> $O/ccs_probe.txt 2>&1; lscpu | head -20 >> $O/ccs_probe.txt
timeout 300 tools/valu_peak/valu_peak > $O/valu_peak.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
CCSX_LIB=$PWD/ccs_amd/libccsx_phases.so timeout 300 python bench.py --no-cpu-baseline --steps 2 > $O/bench_phases.json 2> $O/phases.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/$O/pmc_lds -o pmc -- python $R/bench.py --zmws 2048 --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/bench_pmc_lds.json 2> $R/$O/bench_pmc_lds.err
python $R/tools/profsum.py $R/$O > $R/$O/summary.txt
rm -rf $R/$O/pmc_lds
tail -3 $R/$O/pytest_gpu.log; cat $R/$O/valu_peak.txt | head -45; cat $R/$O/phases.txt | tail -20; cat $R/$O/bench_default.json
