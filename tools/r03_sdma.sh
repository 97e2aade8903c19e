# do the H2D uploads run on SDMA engines or as shader blit kernels?  A/B over the runtime's switches.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_sdma; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
run() {   # tag, env...
  tag=$1; shift
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/$O/$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --pmc --steps 6 --warmup 1 --distinct 6 --serial-stages > $GRAFT_REPO_ROOT/$O/$tag.json 2> $GRAFT_REPO_ROOT/$O/$tag.err)
  python - <<PY
import glob, sqlite3, json
tag = "$tag"
try:
    d = json.load(open("$O/%s.json" % tag)); line = "value %.0f ZMW/s, ms_per_step %.1f, kernels %.1f" % (d["value"], d["ms_per_step"], d["kernels_ms_per_step"])
except Exception as e: line = "bench failed: %s" % e
db = glob.glob("$O/%s/**/*results.db" % tag, recursive=True)
out = []
if db:
    c = sqlite3.connect(db[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    k = c.execute("select count(*), sum(end-start) from kernels where name like '%copyBuffer%'").fetchone()
    out.append("blit copy kernels: %d calls, %.1f ms" % (k[0], (k[1] or 0) / 1e6))
    for t in tabs:
        if "memory_cop" in t.lower() and not t.startswith("rocpd_info"):
            try:
                cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
                n = c.execute("select count(*) from %s" % t).fetchone()[0]
                if "size" in cols and "start" in cols:
                    rows = c.execute("select count(*), sum(size), sum(end-start) from %s where size > 1000000" % t).fetchone()
                    out.append("%s: %d records; > 1 MB: %d copies, %.2f GB, %.1f ms -> %.1f GB/s" % (t, n, rows[0], (rows[1] or 0) / 1e9, (rows[2] or 0) / 1e6, (rows[1] or 0) / max(1, rows[2] or 1)))
                else: out.append("%s: %d records, columns %s" % (t, n, cols[:12]))
            except Exception as e: out.append("%s: %s" % (t, e))
print(tag, "|", line, "|", " ; ".join(out))
PY
  rm -rf $O/$tag
}
run default X=1
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
