# Round-end evidence on one MI355X: kernel trace of the default bench + separate PMC passes (HBM bytes, VALU occupancy).
# usage (on the GPU box): bash tools/prof_round.sh <tag>   -> gpurun_out/prof_<tag>/, summary via tools/profsum.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r01}
D=$R/gpurun_out/prof_$T
rm -rf $D; mkdir -p $D
python $R/bench.py > $D/bench_default.json 2> $D/bench_default.err
rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- python $R/bench.py --no-cpu-baseline > $D/bench_trace.json 2> $D/bench_trace.err
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $D/pmc_$tag -o pmc -- python $R/bench.py --zmws 2048 --steps 1 --warmup 0 --no-cpu-baseline > $D/bench_$tag.json 2> $D/bench_$tag.err
done
python $R/tools/profsum.py $D > $D/summary.txt
python $R/tools/mk_traffic.py $D 2048 > $D/traffic.json
ls -la $D; du -sh $D
