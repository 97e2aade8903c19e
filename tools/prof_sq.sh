cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-4096}
rm -rf $R/gpurun_out/prof_sq; mkdir -p $R/gpurun_out/prof_sq
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SENDMSG"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_sq/pmc_$tag -o pmc -- python $R/bench.py --zmws $N --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_sq/bench_$tag.log 2>&1
done
