# SQ counter pass (instruction counts, wait cycles, LDS conflicts) of the default workload at 2048 ZMWs; summary -> gpurun_out/<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-pmc}; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $O/pmc_sq -o pmc -- python $R/bench.py --zmws 2048 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $O/pmc_sq2 -o pmc -- python $R/bench.py --zmws 2048 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
python $R/tools/profsum.py $O > $O/summary.txt
rm -rf $O/pmc_sq $O/pmc_sq2
grep -v "rocclr\|k_post\|k_setup" $O/summary.txt
