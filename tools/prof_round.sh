# Round-end evidence on one MI355X: kernel trace of the default bench + separate PMC passes (HBM bytes, VALU / LDS counters).
# usage (on the GPU box): bash tools/prof_round.sh <tag> [head]  -> gpurun_out/prof_<tag>/{summary.txt,traffic.json,bench_default.json}
# The counter passes run `bench.py --pmc` (the headline workload and nothing else) so profiles/<tag>_traffic.json describes the very
# command the bench line comes from; <head> (the commit, passed in from the build container: the box has no .git) is stored in it.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04}
H=${2:-unknown}
Z=${Z:-16384}
D=$R/gpurun_out/prof_$T
rm -rf $D; mkdir -p $D
timeout 900 python $R/bench.py > $D/bench_default.json 2> $D/bench_default.err < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- python $R/bench.py --pmc --steps 6 --warmup 1 --distinct 3 > $D/bench_trace.json 2> $D/bench_trace.err < /dev/null
python $R/tools/trace_overlap.py $D/trace > $D/overlap.txt 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  # the HEADLINE step: 16384 ZMWs per batch (8192 until round 4's second session), two-stage queue on (round 3 counted a 2048-ZMW step, where k_poa_dp has 0.5 waves per SIMD)
  # round 5: SERIAL stages for the counter passes — under the two-stage queue a kernel's GRBM_GUI_ACTIVE also counts cycles in which the
  # other stage's kernel holds part of the chip, and a "fraction of the VALU peak" came out above 1 for k_align16
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $D/pmc_$tag -o pmc -- python $R/bench.py --pmc --serial-stages --zmws $Z --steps 1 --warmup 1 --distinct 1 > $D/bench_$tag.json 2> $D/bench_$tag.err < /dev/null
done
python $R/tools/profsum.py $D > $D/summary.txt
python $R/tools/mk_traffic.py $D $Z $H > $D/traffic.json
rm -rf $D/pmc_*
# the same counters for configs[3] (30 passes x 20 kb; VERDICT r04 item 6d: its alignment and polish stages attributed, not guessed)
if [ "${C4:-1}" != 0 ]; then
  Z4=${Z4:-4096}
  for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 900 rocprofv3 --kernel-trace --pmc $set -d $D/pmc_$tag -o pmc -- python $R/bench.py --pmc --serial-stages --workload c4 --zmws $Z4 --steps 1 --warmup 1 --distinct 1 > $D/bench_c4_$tag.json 2> $D/bench_c4_$tag.err < /dev/null
  done
  python $R/tools/mk_traffic.py $D $Z4 $H "30 passes x 20 kb" > $D/traffic_c4.json
  rm -rf $D/pmc_*
fi
# round 6: the same counters for configs[4] (3-50 passes x 1-25 kb): its stage balance attributed, not guessed
if [ "${C5:-1}" != 0 ]; then
  Z5=${Z5:-8192}
  for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 900 rocprofv3 --kernel-trace --pmc $set -d $D/pmc_$tag -o pmc -- python $R/bench.py --pmc --serial-stages --workload c5 --zmws $Z5 --steps 1 --warmup 1 --distinct 1 > $D/bench_c5_$tag.json 2> $D/bench_c5_$tag.err < /dev/null
  done
  python $R/tools/mk_traffic.py $D $Z5 $H "3-50 passes x 1-25 kb" > $D/traffic_c5.json
  rm -rf $D/pmc_*
fi
rm -rf $D/trace
ls -la $D; cat $D/traffic.json
