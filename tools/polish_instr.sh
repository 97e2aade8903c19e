# k_polish by phase in INSTRUCTIONS (not only time): the compiled-out variants of profiles/r04_polish_attribution.txt under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU
# SQ_INSTS_LDS SQ_WAVE_CYCLES, one 8192-ZMW step each, serial stages.   usage (through gpurun): bash tools/polish_instr.sh  -> gpurun_out/pinstr/summary.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/pinstr; rm -rf $O; mkdir -p $O
for v in ${PINSTR_VARIANTS:-"base:" "one:-DCCSX_EXP_ONE_ROUND" "one_nofill:-DCCSX_EXP_ONE_ROUND,-DCCSX_EXP_NO_FILL,-DCCSX_EXP_ALL_VALID" "one_noscore:-DCCSX_EXP_ONE_ROUND,-DCCSX_EXP_NO_SCORE" "one_nomask:-DCCSX_EXP_ONE_ROUND,-DCCSX_EXP_NO_BANDMASK" "prologue:-DCCSX_EXIT_AFTER_PROLOGUE"}; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  CCSX_EXTRA_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_$name.log 2>&1 || { echo "build $name failed"; continue; }
  ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $GRAFT_REPO_ROOT/$O/p_$name -o pmc -- python $GRAFT_REPO_ROOT/bench.py --pmc --serial-stages --zmws 8192 --steps 1 --warmup 1 --distinct 1 > $GRAFT_REPO_ROOT/$O/b_$name.json 2> $GRAFT_REPO_ROOT/$O/b_$name.err < /dev/null )
  python - <<PY
import sqlite3, glob
db = glob.glob("$O/p_$name/**/pmc_results.db", recursive=True) or glob.glob("$O/p_$name/pmc_results.db")
c = sqlite3.connect(db[0])
rows = dict((cn, (v, k)) for cn, v, k in c.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like '%k_polish%' group by counter_name"))
n = rows["SQ_INSTS_VALU"][1]
print("%-12s per ZMW (8192 per dispatch, %d dispatches): VALU %9.0f  SALU %9.0f  LDS %9.0f  wave-cycles %11.0f  lanes active %.3f" % ("$name", n, rows["SQ_INSTS_VALU"][0] / n / 8192, rows["SQ_INSTS_SALU"][0] / n / 8192,
      rows["SQ_INSTS_LDS"][0] / n / 8192, rows["SQ_WAVE_CYCLES"][0] / n / 8192, rows["SQ_THREAD_CYCLES_VALU"][0] / 64 / rows["SQ_INSTS_VALU"][0]))
PY
  rm -rf $O/p_$name
done 2>&1 | tee $O/summary.txt
python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
