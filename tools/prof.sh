cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o trace -- python $R/bench.py --zmws 4096 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_trace.log 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof/pmc_$tag -o pmc -- python $R/bench.py --zmws 1024 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof/bench_$tag.log 2>&1
done
find $R/gpurun_out/prof -name "*.csv" | head -50; du -sh $R/gpurun_out/prof
