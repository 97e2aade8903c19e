# kernel trace of a short bench run: per-kernel durations + the timeline between batches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_trace; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --pmc --steps 5 --warmup 1 --distinct 3 --serial-stages $BENCH_ARGS > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err)
python tools/profsum.py $O > $O/summary.txt; head -40 $O/summary.txt
python tools/trace_gaps.py $O/trace > $O/gaps.txt 2>&1; cat $O/gaps.txt
python - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/r03_trace/trace/**/*results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
# the last full batch: list every dispatch with start / duration
setups = [i for i, r in enumerate(rows) if r[0].startswith("k_setup")]
a, b = setups[-2], setups[-1]
prev_end = None
for n, s, e in rows[a:b]:
    print("%-28s start %9.3f ms  dur %8.3f ms  gap-before %7.3f" % (n.split("(")[0][:28], (s - t0) / 1e6, (e - s) / 1e6, 0 if prev_end is None else (s - prev_end) / 1e6))
    prev_end = e
PY
rm -rf $O/trace
