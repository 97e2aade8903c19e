# VALU occupancy of every kernel at the default bench size (separate PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_valu
rm -rf $D; mkdir -p $D
for set in "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $D/pmc_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $D/bench_$tag.json 2> $D/bench_$tag.err
done
python $R/tools/profsum.py $D > $D/summary.txt
python $R/tools/mk_traffic.py $D 8192 > $D/valu.json
